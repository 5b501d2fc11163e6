// C ABI of libcmblens_hip.so (see include/cmblens.h): entry points, argument checks and error plumbing.  No kernel is instantiated in this
// translation unit -- the typed bodies do_*<T> live in api_body.hpp and are compiled by tu_main_{f32,f64}.hip (api_decl.hpp has the map).
#include "api_decl.hpp"

namespace cmbl { thread_local std::string g_last_error; }
using namespace cmbl;

// Driver scratch lives in the dataset, keyed by the flow it was built for (it holds a Flow<T>&): a flow that goes away must take its
// entries with it -- the buffers would otherwise leak until the dataset is destroyed, and a NEW flow allocated at the same address would
// silently inherit scratch built around the old one.  Datasets register here so that cmbl_lenseflow_destroy can find them.
static std::mutex g_ds_mtx;
static std::vector<cmbl_dataset*> g_datasets;
static void registry_add(cmbl_dataset* d) { std::lock_guard<std::mutex> l(g_ds_mtx); g_datasets.push_back(d); }
static void registry_remove(cmbl_dataset* d) { std::lock_guard<std::mutex> l(g_ds_mtx); g_datasets.erase(std::remove(g_datasets.begin(), g_datasets.end(), d), g_datasets.end()); }
static void registry_drop_flow(const void* f32, const void* f64) {
  std::lock_guard<std::mutex> l(g_ds_mtx);
  for (cmbl_dataset* d : g_datasets) { if (f32) d->drv32.erase(f32); if (f64) d->drv64.erase(f64); }
}

template <typename F>
static int guard(F&& f) {
  try { f(); return CMBL_OK; }
  catch (const Error& e) { g_last_error = e.msg; return e.code; }
  catch (const std::exception& e) { g_last_error = e.what(); return CMBL_ERR_ARG; }
  catch (...) { g_last_error = "unknown error"; return CMBL_ERR_ARG; }
}
#define NOTNULL(p) CMBL_REQUIRE((p) != nullptr, ERR_ARG, "null pointer argument: " #p)
#define BASIS_OK(b) CMBL_REQUIRE((b) >= 0 && (b) <= 2, ERR_ARG, "bad basis: " #b)
#define POLB_OK(P, B) CMBL_REQUIRE((P) >= 1 && (P) <= 3 && (B) >= 1, ERR_SHAPE, "npol must be 1..3 and nbatch >= 1")
#define BY_DTYPE(ctx, expr32, expr64) do { if ((ctx)->p->dtype == CMBL_F32) { expr32; } else { expr64; } } while (0)

#ifdef CMBL_STAMPS
namespace cmbl {
#define CMBL_X(unit) int stamps_read_##unit(unsigned long long* out_host, int n);
CMBL_X(main_f32) CMBL_X(main_f64) CMBL_X(gen_f32) CMBL_X(gen_f64) CMBL_X(small_f32) CMBL_X(small_f64) CMBL_X(cty_f32_a) CMBL_X(cty_f32_b) CMBL_X(cty_f64_a) CMBL_X(cty_f64_b) \
    CMBL_X(ctx_f32_a) CMBL_X(ctx_f32_b) CMBL_X(ctx_f64_a) CMBL_X(ctx_f64_b)
#undef CMBL_X
}
#endif
extern "C" {

const char* cmbl_last_error(void) { return g_last_error.c_str(); }
int cmbl_version(void) { return 100; }
int cmbl_abi_version(void) { return CMBL_ABI_VERSION; }

int cmbl_ctx_create(int Ny, int Nx, double theta, int dtype, int device, void* stream, cmbl_ctx** out) {
  return guard([&] {
    NOTNULL(out);
    CMBL_REQUIRE(dtype == CMBL_F32 || dtype == CMBL_F64, ERR_ARG, "dtype must be CMBL_F32 or CMBL_F64");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) fail(ERR_HIP, "no HIP device available (this library has no CPU fallback)");
    CMBL_REQUIRE(device >= 0 && device < ndev, ERR_ARG, "device index out of range");
    auto h = std::make_unique<cmbl_ctx>();
    if (dtype == CMBL_F32) h->p.reset(do_ctx_create<float>(Ny, Nx, theta, device, stream));
    else h->p.reset(do_ctx_create<double>(Ny, Nx, theta, device, stream));
    *out = h.release();
  });
}
int cmbl_ctx_destroy(cmbl_ctx* ctx) { return guard([&] { delete ctx; }); }
int cmbl_ctx_synchronize(cmbl_ctx* ctx) { return guard([&] { NOTNULL(ctx); CMBL_HIP(hipStreamSynchronize(ctx->p->stream)); }); }

int cmbl_ctx_set_option(cmbl_ctx* ctx, const char* name, int value) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(name);
    int* o = ctx->p->opt_ptr(name);
    CMBL_REQUIRE(o != nullptr, ERR_ARG, std::string("unknown option: ") + name);
    *o = value;
  });
}
int cmbl_ctx_get_option(cmbl_ctx* ctx, const char* name, int* value_host) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(name); NOTNULL(value_host);
    int* o = ctx->p->opt_ptr(name);
    CMBL_REQUIRE(o != nullptr, ERR_ARG, std::string("unknown option: ") + name);
    *value_host = *o;
  });
}
int cmbl_prof_enable(cmbl_ctx* ctx, int on) {
  return guard([&] { NOTNULL(ctx); if (!on) ctx->p->prof_collect(); ctx->p->prof_on = on != 0; });
}
int cmbl_prof_reset(cmbl_ctx* ctx) { return guard([&] { NOTNULL(ctx); ctx->p->prof_collect(); ctx->p->prof_reset(); }); }
int cmbl_prof_count(void) { return K_COUNT; }
const char* cmbl_prof_name(int k) { return (k >= 0 && k < K_COUNT) ? kKernelNames[k] : ""; }
int cmbl_prof_get(cmbl_ctx* ctx, int k, double* total_ms, long* launches) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(total_ms); NOTNULL(launches);
    CMBL_REQUIRE(k >= 0 && k < K_COUNT, ERR_ARG, "bad kernel class");
    ctx->p->prof_collect();
    *total_ms = ctx->p->prof_ms[k]; *launches = ctx->p->prof_n[k];
  });
}

int cmbl_timer_report(cmbl_ctx* ctx, char* buf, size_t buflen) {
  int need = 0;
  const int rc = guard([&] {
    NOTNULL(ctx);
    ctx->p->prof_collect();
    std::string r = "kernel_class launches total_ms mean_us\n";
    for (int k = 0; k < K_COUNT; ++k) {
      if (!ctx->p->prof_n[k]) continue;
      char line[160];
      std::snprintf(line, sizeof line, "%s %ld %.6f %.3f\n", kKernelNames[k], ctx->p->prof_n[k], ctx->p->prof_ms[k], 1e3 * ctx->p->prof_ms[k] / ctx->p->prof_n[k]);
      r += line;
    }
    need = (int)r.size();
    if (buf && buflen) { const size_t m = std::min(buflen - 1, r.size()); std::memcpy(buf, r.data(), m); buf[m] = 0; }
  });
  return rc == CMBL_OK ? need : -rc;
}
int cmbl_device_malloc(cmbl_ctx* ctx, size_t bytes, void** out) {
  return guard([&] { NOTNULL(ctx); NOTNULL(out); CMBL_HIP(hipSetDevice(ctx->p->device)); hipError_t e = hipMalloc(out, bytes); if (e != hipSuccess) { *out = nullptr; (void)hipGetLastError(); fail(ERR_ALLOC, std::string("hipMalloc: ") + hipGetErrorString(e)); } });
}
int cmbl_device_free(cmbl_ctx* ctx, void* p) { return guard([&] { NOTNULL(ctx); CMBL_HIP(hipSetDevice(ctx->p->device)); if (p) CMBL_HIP(hipFree(p)); }); }
int cmbl_copy_to_device(cmbl_ctx* ctx, void* dst, const void* src, size_t bytes) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(dst); NOTNULL(src);
    CMBL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->p->stream)); CMBL_HIP(hipStreamSynchronize(ctx->p->stream));
  });
}
int cmbl_copy_to_host(cmbl_ctx* ctx, void* dst, const void* src, size_t bytes) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(dst); NOTNULL(src);
    CMBL_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->p->stream)); CMBL_HIP(hipStreamSynchronize(ctx->p->stream));
  });
}

int cmbl_ctx_geometry_host(cmbl_ctx* ctx, int which, double* out, size_t n) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(out);
    const CtxBase& c = *ctx->p;
    const std::vector<double>* v = which == 0 ? &c.h_lx : which == 1 ? &c.h_ly : which == 2 ? &c.h_lam : which == 3 ? &c.h_sin2
                                 : which == 4 ? &c.h_cos2 : which == 5 ? &c.h_lmag : nullptr;
    CMBL_REQUIRE(v != nullptr, ERR_ARG, "bad geometry selector");
    CMBL_REQUIRE(n == v->size(), ERR_SHAPE, "geometry output has the wrong length");
    std::memcpy(out, v->data(), n * sizeof(double));
  });
}


int cmbl_convert(cmbl_ctx* ctx, int bi, const void* in, int bo, void* out, int P, int B) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(in); NOTNULL(out); BASIS_OK(bi); BASIS_OK(bo); POLB_OK(P, B);
    BY_DTYPE(ctx, do_convert<float>(ctx, bi, in, bo, out, P, B), do_convert<double>(ctx, bi, in, bo, out, P, B));
  });
}
int cmbl_rfft(cmbl_ctx* ctx, const void* map, void* fourier, int P, int B) { return cmbl_convert(ctx, CMBL_MAP, map, CMBL_FOURIER, fourier, P, B); }
int cmbl_irfft(cmbl_ctx* ctx, const void* fourier, void* map, int P, int B) { return cmbl_convert(ctx, CMBL_FOURIER, fourier, CMBL_MAP, map, P, B); }


int cmbl_diag_apply(cmbl_ctx* ctx, int kind, int bd, const void* diag, int bi, const void* in, int bo, void* out, int P, int B) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(diag); NOTNULL(in); NOTNULL(out); BASIS_OK(bi); BASIS_OK(bo); POLB_OK(P, B);
    CMBL_REQUIRE(kind == CMBL_DIAG_MUL || kind == CMBL_DIAG_DIV_NAN2ZERO, ERR_ARG, "kind must be CMBL_DIAG_MUL or CMBL_DIAG_DIV_NAN2ZERO");
    CMBL_REQUIRE(bd == CMBL_FOURIER || bd == CMBL_HARMONIC, ERR_ARG, "operator must be diagonal in FOURIER or HARMONIC");
    BY_DTYPE(ctx, do_diag<float>(ctx, kind, bd, diag, P, false, bi, in, bo, out, P, B), do_diag<double>(ctx, kind, bd, diag, P, false, bi, in, bo, out, P, B));
  });
}
int cmbl_blockdiag_ieb_apply(cmbl_ctx* ctx, const void* te_bb, int transpose, int bi, const void* in, int bo, void* out, int B) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(te_bb); NOTNULL(in); NOTNULL(out); BASIS_OK(bi); BASIS_OK(bo); POLB_OK(3, B);
    BY_DTYPE(ctx, do_diag<float>(ctx, 2, B_HARMONIC, te_bb, 5, transpose != 0, bi, in, bo, out, 3, B),
             do_diag<double>(ctx, 2, B_HARMONIC, te_bb, 5, transpose != 0, bi, in, bo, out, 3, B));
  });
}

int cmbl_dot(cmbl_ctx* ctx, int basis, const void* a, const void* b, int P, int B, double* out) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(a); NOTNULL(b); NOTNULL(out); BASIS_OK(basis); POLB_OK(P, B);
    BY_DTYPE(ctx, do_dot<float>(ctx, basis, a, b, P, B, out), do_dot<double>(ctx, basis, a, b, P, B, out));
  });
}
int cmbl_norm(cmbl_ctx* ctx, int basis, const void* a, int P, int B, double* out) {
  const int rc = cmbl_dot(ctx, basis, a, a, P, B, out);
  if (rc == CMBL_OK) for (int i = 0; i < B; ++i) out[i] = std::sqrt(out[i]);
  return rc;
}
int cmbl_logdet_diag(cmbl_ctx* ctx, int basis, const void* d, int P, int B, double* out) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(d); NOTNULL(out); BASIS_OK(basis); POLB_OK(P, B);
    BY_DTYPE(ctx, do_diag_reduce<float>(ctx, 0, basis, d, P, B, out), do_diag_reduce<double>(ctx, 0, basis, d, P, B, out));
  });
}
int cmbl_tr_diag(cmbl_ctx* ctx, int basis, const void* d, int P, int B, double* out) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(d); NOTNULL(out); BASIS_OK(basis); POLB_OK(P, B);
    BY_DTYPE(ctx, do_diag_reduce<float>(ctx, 1, basis, d, P, B, out), do_diag_reduce<double>(ctx, 1, basis, d, P, B, out));
  });
}
int cmbl_set_sum_accuracy_mode(cmbl_ctx* ctx, int mode) {
  return guard([&] {
    NOTNULL(ctx);
    CMBL_REQUIRE(mode == CMBL_SUM_WORKING || mode == CMBL_SUM_FLOAT64 || mode == CMBL_SUM_KAHAN, ERR_ARG, "mode must be CMBL_SUM_WORKING, _FLOAT64 or _KAHAN");
    ctx->p->sum_mode = mode;
  });
}
int cmbl_logdet(cmbl_ctx* ctx, const void* d, int nplanes, double* out) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(d); NOTNULL(out); CMBL_REQUIRE(nplanes >= 1, ERR_ARG, "nplanes >= 1");
    BY_DTYPE(ctx, do_logdet<float>(ctx, d, nplanes, out), do_logdet<double>(ctx, d, nplanes, out));
  });
}

// ---- LenseFlow -------------------------------------------------------------------------------------
int cmbl_lenseflow_create(cmbl_ctx* ctx, int nsteps, cmbl_flow** out) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(out);
    auto h = std::make_unique<cmbl_flow>();
    h->ctx = ctx;
    BY_DTYPE(ctx, do_flow_create<float>(h.get(), nsteps), do_flow_create<double>(h.get(), nsteps));
    *out = h.release();
  });
}
int cmbl_lenseflow_destroy(cmbl_flow* L) { return guard([&] { if (L) registry_drop_flow(L->f32.get(), L->f64.get()); delete L; }); }
int cmbl_lenseflow_set_phi(cmbl_flow* L, int basis, const void* phi, int nb) {
  return guard([&] {
    NOTNULL(L); NOTNULL(phi); BASIS_OK(basis); CMBL_REQUIRE(nb >= 1, ERR_SHAPE, "nbatch_phi >= 1");
    BY_DTYPE(L->ctx, do_flow_set_phi<float>(L, basis, phi, nb), do_flow_set_phi<double>(L, basis, phi, nb));
  });
}
int cmbl_lenseflow_apply(cmbl_flow* L, int mode, int bi, const void* in, int bo, void* out, int P, int B) {
  return guard([&] {
    NOTNULL(L); NOTNULL(in); NOTNULL(out); BASIS_OK(bi); BASIS_OK(bo); POLB_OK(P, B);
    CMBL_REQUIRE(mode >= 0 && mode <= 3, ERR_ARG, "bad flow mode");
    BY_DTYPE(L->ctx, do_flow_apply<float>(L, mode, bi, in, bo, out, P, B), do_flow_apply<double>(L, mode, bi, in, bo, out, P, B));
  });
}
int cmbl_lenseflow_grad(cmbl_flow* L, int mode, const void* f_end, int bdel, const void* delta, void* dphi, int bdf, void* df,
                        void* f_start, int P, int B, int quirk) {
  return guard([&] {
    NOTNULL(L); NOTNULL(f_end); NOTNULL(delta); NOTNULL(dphi); NOTNULL(df); BASIS_OK(bdel); BASIS_OK(bdf); POLB_OK(P, B);
    CMBL_REQUIRE(mode == CMBL_FLOW_FWD || mode == CMBL_FLOW_INV, ERR_ARG, "grad mode must be CMBL_FLOW_FWD or CMBL_FLOW_INV");
    BY_DTYPE(L->ctx, do_flow_grad<float>(L, mode, f_end, bdel, delta, dphi, bdf, df, f_start, P, B, quirk),
             do_flow_grad<double>(L, mode, f_end, bdel, delta, dphi, bdf, df, f_start, P, B, quirk));
  });
}

int cmbl_max_lensing_step(cmbl_flow* L, int basis, const void* phi, const void* eta, int nb, double* out) {
  return guard([&] {
    NOTNULL(L); NOTNULL(phi); NOTNULL(eta); NOTNULL(out); BASIS_OK(basis); CMBL_REQUIRE(nb >= 1, ERR_SHAPE, "nbatch >= 1");
    BY_DTYPE(L->ctx, do_max_lensing_step<float>(L, basis, phi, eta, nb, out), do_max_lensing_step<double>(L, basis, phi, eta, nb, out));
  });
}

// ---- small linear algebra / quadratic-estimate helpers ---------------------------------------------------
int cmbl_axpby(cmbl_ctx* ctx, int basis, const double* a, const void* x, const double* b, const void* y, void* out, int P, int B) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(a); NOTNULL(x); NOTNULL(out); BASIS_OK(basis); POLB_OK(P, B);
    CMBL_REQUIRE(y == nullptr || b != nullptr, ERR_ARG, "b is required when y is given");
    const long n = (basis == B_MAP ? ctx->p->npix() : 2 * ctx->p->plane()) * P;
    BY_DTYPE(ctx, do_axpby<float>(ctx, a, x, b, y, out, n, B), do_axpby<double>(ctx, a, x, b, y, out, n, B));
  });
}
int cmbl_qe_leg(cmbl_ctx* ctx, const void* in_fourier, int n, int p1, int p2, void* out_map, int B) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(in_fourier); NOTNULL(out_map); CMBL_REQUIRE(B >= 1 && n >= 0 && p1 >= 0 && p2 >= 0, ERR_ARG, "bad leg indices");
    BY_DTYPE(ctx, do_qe_leg<float>(ctx, in_fourier, n, p1, p2, out_map, B), do_qe_leg<double>(ctx, in_fourier, n, p1, p2, out_map, B));
  });
}
int cmbl_fourier_lmul(cmbl_ctx* ctx, const void* in_map, int p1, int p2, int take_abs, void* out_fourier, int B) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(in_map); NOTNULL(out_fourier); CMBL_REQUIRE(B >= 1 && p1 >= 0 && p2 >= 0, ERR_ARG, "bad exponents");
    BY_DTYPE(ctx, do_fourier_lmul<float>(ctx, in_map, p1, p2, take_abs, out_fourier, B), do_fourier_lmul<double>(ctx, in_map, p1, p2, take_abs, out_fourier, B));
  });
}
int cmbl_map_fma(cmbl_ctx* ctx, const void* a, const void* b, double scale, void* out, int accumulate, int nslices) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(a); NOTNULL(b); NOTNULL(out); CMBL_REQUIRE(nslices >= 1, ERR_SHAPE, "nslices >= 1");
    const long n = ctx->p->npix() * nslices;
    BY_DTYPE(ctx, do_map_fma<float>(ctx, a, b, scale, out, accumulate, n), do_map_fma<double>(ctx, a, b, scale, out, accumulate, n));
  });
}

int cmbl_randn(cmbl_ctx* ctx, const uint64_t* seeds, int nslots, uint64_t stream, void* out, long n_per_slot) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(seeds); NOTNULL(out);
    CMBL_REQUIRE(nslots >= 1 && n_per_slot >= 1, ERR_SHAPE, "nslots >= 1 and n_per_slot >= 1");
    BY_DTYPE(ctx, do_randn<float>(ctx, seeds, nslots, stream, out, n_per_slot), do_randn<double>(ctx, seeds, nslots, stream, out, n_per_slot));
  });
}

// ---- dataset ---------------------------------------------------------------------------------------
int cmbl_dataset_create(cmbl_ctx* ctx, int npol, cmbl_dataset** out) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(out);
    auto h = std::make_unique<cmbl_dataset>();
    h->ctx = ctx;
    BY_DTYPE(ctx, do_dataset_create<float>(h.get(), npol), do_dataset_create<double>(h.get(), npol));
    registry_add(h.get());
    *out = h.release();
  });
}
int cmbl_dataset_destroy(cmbl_dataset* ds) { return guard([&] { if (ds) registry_remove(ds); delete ds; }); }
int cmbl_dataset_set_op(cmbl_dataset* ds, int which, const void* planes, int nplanes) {
  return guard([&] { NOTNULL(ds); NOTNULL(planes); BY_DTYPE(ds->ctx, do_dataset_set_op<float>(ds, which, planes, nplanes), do_dataset_set_op<double>(ds, which, planes, nplanes)); });
}
int cmbl_dataset_set_data(cmbl_dataset* ds, const void* d, int B) {
  return guard([&] { NOTNULL(ds); NOTNULL(d); CMBL_REQUIRE(B >= 1, ERR_SHAPE, "nbatch >= 1"); BY_DTYPE(ds->ctx, do_dataset_set_data<float>(ds, d, B), do_dataset_set_data<double>(ds, d, B)); });
}
int cmbl_dataset_set_logdet(cmbl_dataset* ds, double v) {
  return guard([&] { NOTNULL(ds); BY_DTYPE(ds->ctx, ds->f32->logdet_sum = v, ds->f64->logdet_sum = v); });
}

int cmbl_gradientf_logpdf(cmbl_dataset* ds, cmbl_flow* L, const void* f, const void* d, int zero_d, void* out, int B) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(L); NOTNULL(f); NOTNULL(out); CMBL_REQUIRE(B >= 1, ERR_SHAPE, "nbatch >= 1");
    CMBL_REQUIRE(ds->ctx == L->ctx, ERR_ARG, "dataset and flow belong to different contexts");
    BY_DTYPE(ds->ctx, do_gradf<float>(ds, L, f, d, zero_d, out, B), do_gradf<double>(ds, L, f, d, zero_d, out, B));
  });
}

int cmbl_wiener_cg(cmbl_dataset* ds, cmbl_flow* L, const void* d, const void* fstart, double tol, int maxit, void* f_out,
                   double* hist, int* nit, int B) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(L); NOTNULL(f_out); NOTNULL(hist); NOTNULL(nit);
    CMBL_REQUIRE(B >= 1 && maxit >= 1, ERR_ARG, "nbatch >= 1 and maxit >= 1");
    CMBL_REQUIRE(ds->ctx == L->ctx, ERR_ARG, "dataset and flow belong to different contexts");
    BY_DTYPE(ds->ctx, do_cg<float>(ds, L, d, fstart, tol, maxit, f_out, hist, nit, B),
             do_cg<double>(ds, L, d, fstart, tol, maxit, f_out, hist, nit, B));
  });
}

int cmbl_logpdf_mixed(cmbl_dataset* ds, cmbl_flow* L, const void* fo, const void* phio, double* lp, int B) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(L); NOTNULL(fo); NOTNULL(phio); NOTNULL(lp); CMBL_REQUIRE(B >= 1, ERR_SHAPE, "nbatch >= 1");
    CMBL_REQUIRE(ds->ctx == L->ctx, ERR_ARG, "dataset and flow belong to different contexts");
    BY_DTYPE(ds->ctx, do_lpm<float>(ds, L, fo, phio, lp, nullptr, nullptr, B, 0), do_lpm<double>(ds, L, fo, phio, lp, nullptr, nullptr, B, 0));
  });
}
int cmbl_grad_logpdf_mixed(cmbl_dataset* ds, cmbl_flow* L, const void* fo, const void* phio, double* lp, void* gfo, void* gphio, int B, int quirk) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(L); NOTNULL(fo); NOTNULL(phio); NOTNULL(lp); NOTNULL(gfo); NOTNULL(gphio); CMBL_REQUIRE(B >= 1, ERR_SHAPE, "nbatch >= 1");
    CMBL_REQUIRE(ds->ctx == L->ctx, ERR_ARG, "dataset and flow belong to different contexts");
    BY_DTYPE(ds->ctx, do_lpm<float>(ds, L, fo, phio, lp, gfo, gphio, B, quirk), do_lpm<double>(ds, L, fo, phio, lp, gfo, gphio, B, quirk));
  });
}

int cmbl_hmc_step(cmbl_dataset* ds, cmbl_flow* L, const void* fo, const void* phio, const void* mass, const void* white_p, const double* log_u_host,
                  const uint64_t* seeds_host, uint64_t step, int nleap, double eps, int always_accept, int alias_quirk, int B,
                  void* phio_out, double* dH_host, int* accept_host) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(L); NOTNULL(fo); NOTNULL(phio); NOTNULL(mass); NOTNULL(phio_out); NOTNULL(dH_host); NOTNULL(accept_host);
    CMBL_REQUIRE(B >= 1 && B <= MAXBATCH && nleap >= 1, ERR_ARG, "1 <= nbatch <= 256 and nleap >= 1");
    CMBL_REQUIRE(ds->ctx == L->ctx, ERR_ARG, "dataset and flow belong to different contexts");
    BY_DTYPE(ds->ctx, do_hmc<float>(ds, L, fo, phio, mass, white_p, log_u_host, seeds_host, step, nleap, eps, always_accept, alias_quirk, B, phio_out, dH_host, accept_host),
             do_hmc<double>(ds, L, fo, phio, mass, white_p, log_u_host, seeds_host, step, nleap, eps, always_accept, alias_quirk, B, phio_out, dH_host, accept_host));
  });
}
int cmbl_map_joint_step(cmbl_dataset* ds, cmbl_flow* L, const void* phi, const void* fstart, const void* hinv, double alpha_max, double alpha_tol,
                        double cg_tol, int cg_maxit, int alias_quirk, int B, void* f_out, void* phi_out, double* logpdf_host, double* alpha_host,
                        int* ncg_host, int* nls_host) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(L); NOTNULL(phi); NOTNULL(hinv); NOTNULL(f_out); NOTNULL(phi_out); NOTNULL(logpdf_host); NOTNULL(alpha_host); NOTNULL(ncg_host); NOTNULL(nls_host);
    CMBL_REQUIRE(B >= 1 && B <= MAXBATCH && cg_maxit >= 1 && alpha_max > 0 && alpha_tol > 0, ERR_ARG, "1 <= nbatch <= 256, cg_maxit >= 1, alpha_max > 0, alpha_tol > 0");
    CMBL_REQUIRE(ds->ctx == L->ctx, ERR_ARG, "dataset and flow belong to different contexts");
    BY_DTYPE(ds->ctx, do_map_step<float>(ds, L, phi, fstart, hinv, alpha_max, alpha_tol, cg_tol, cg_maxit, alias_quirk, B, f_out, phi_out, logpdf_host, alpha_host, ncg_host, nls_host),
             do_map_step<double>(ds, L, phi, fstart, hinv, alpha_max, alpha_tol, cg_tol, cg_maxit, alias_quirk, B, f_out, phi_out, logpdf_host, alpha_host, ncg_host, nls_host));
  });
}

int cmbl_quadratic_estimate(cmbl_dataset* ds, int which, const double* Cf_host, const double* Cftilde_host, const double* Cn_host, const double* TF_host,
                            const double* Cphi_host, int wiener_filtered, const double* AL_in_host, void* phiqe_out, double* AL_out_host, int B) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(Cf_host); NOTNULL(Cftilde_host); NOTNULL(Cn_host); NOTNULL(TF_host); NOTNULL(Cphi_host); NOTNULL(phiqe_out);
    CMBL_REQUIRE(which >= 0 && which <= 2, ERR_ARG, "which: 0 = TT, 1 = EE, 2 = EB (src/quadratic_estimate.jl:41: the others are not implemented by the reference either)");
    CMBL_REQUIRE(B >= 1, ERR_SHAPE, "nbatch >= 1");
    BY_DTYPE(ds->ctx, do_qe<float>(ds, which, Cf_host, Cftilde_host, Cn_host, TF_host, Cphi_host, wiener_filtered, AL_in_host, phiqe_out, AL_out_host, B),
             do_qe<double>(ds, which, Cf_host, Cftilde_host, Cn_host, TF_host, Cphi_host, wiener_filtered, AL_in_host, phiqe_out, AL_out_host, B));
  });
}

#ifdef CMBL_STAMPS
// phase timestamps of the last stamped launch (tools/gpu_stamps*.py) of the translation unit CMBL_STAMPS_TU (kernels_fft.hpp CMBL_STAMPS_READER)
int cmbl_debug_stamps(unsigned long long* out_host, int n) {
  return guard([&] {
    CMBL_HIP(hipDeviceSynchronize());
    const char* e = std::getenv("CMBL_STAMPS_TU");
    const std::string u = e ? e : "main_f32";
    int rc = -1;
#define CMBL_X(unit) if (u == #unit) rc = cmbl::stamps_read_##unit(out_host, n);
    CMBL_X(main_f32) CMBL_X(main_f64) CMBL_X(gen_f32) CMBL_X(gen_f64) CMBL_X(small_f32) CMBL_X(small_f64) CMBL_X(cty_f32_a) CMBL_X(cty_f32_b) CMBL_X(cty_f64_a) CMBL_X(cty_f64_b) \
    CMBL_X(ctx_f32_a) CMBL_X(ctx_f32_b) CMBL_X(ctx_f64_a) CMBL_X(ctx_f64_b)
#undef CMBL_X
    CMBL_REQUIRE(rc == 0, ERR_ARG, "CMBL_STAMPS_TU: main_{f32,f64} | gen_{f32,f64} | small_{f32,f64} | cty_{f32,f64}_{a,b} | ctx_{f32,f64}_{a,b}");
  });
}
#endif

}  // extern "C"
