// C ABI of libcmblens_hip.so (see include/cmblens.h).  Single translation unit: all kernels are templates.
#include "engine.hpp"
#include "drivers.hpp"
#include "../../include/cmblens.h"

namespace cmbl { thread_local std::string g_last_error; }
using namespace cmbl;

struct cmbl_ctx { std::unique_ptr<CtxBase> p; };
struct cmbl_flow { cmbl_ctx* ctx; std::unique_ptr<Flow<float>> f32; std::unique_ptr<Flow<double>> f64; };
struct cmbl_dataset {
  cmbl_ctx* ctx; std::unique_ptr<Dataset<float>> f32; std::unique_ptr<Dataset<double>> f64;
  std::map<const void*, std::unique_ptr<Drivers<float>>> drv32;        // driver scratch per (dataset, flow) pair
  std::map<const void*, std::unique_ptr<Drivers<double>>> drv64;
  std::vector<std::unique_ptr<DevBuf>> qe_pool;                        // legs and products of cmbl_quadratic_estimate, reused between calls
};

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

template <typename T> static Ctx<T>* C(cmbl_ctx* c) { return static_cast<Ctx<T>*>(c->p.get()); }
#define BY_DTYPE(ctx, expr32, expr64) do { if ((ctx)->p->dtype == CMBL_F32) { expr32; } else { expr64; } } while (0)

template <typename T> static void do_convert(cmbl_ctx* ctx, int bi, const void* in, int bo, void* out, int P, int B) {
  Ctx<T>* c = C<T>(ctx);
  const long sl = (long)P * B;
  c->tmpA.ensure(sizeof(cx<T>) * sl * c->plane());
  cx<T>* F = c->tmpA.template as<cx<T>>();
  // carry the data in the basis of whichever side is a Fourier basis; map<->map is a copy
  if (bi == B_MAP && bo == B_MAP) { CMBL_HIP(hipMemcpyAsync(out, in, sizeof(T) * sl * c->npix(), hipMemcpyDeviceToDevice, c->stream)); return; }
  const int carry = (bi == B_MAP) ? (bo == B_HARMONIC ? B_HARMONIC : B_FOURIER) : bi;
  c->to_F(bi, in, F, carry, P, B);
  c->from_F(F, carry, bo, out, P, B);
}
template <typename T>
static void do_diag(cmbl_ctx* ctx, int kind, int bd, const void* diag, int nplanes, bool transpose, int bi, const void* in, int bo, void* out, int P, int B) {
  Ctx<T>* c = C<T>(ctx);
  const long sl = (long)P * B;
  c->tmpA.ensure(sizeof(cx<T>) * sl * c->plane());
  c->tmpB.ensure(sizeof(T) * nplanes * c->plane());
  cx<T>* F = c->tmpA.template as<cx<T>>();
  T* dF = c->tmpB.template as<T>();
  c->ref2F_real((const T*)diag, dF, nplanes);
  const T* d[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  for (int k = 0; k < nplanes; ++k) d[k] = dF + (size_t)k * c->plane();
  c->to_F(bi, in, F, bd, P, B);                       // B(f): convert to the operator's basis (src/specialops.jl:9)
  c->harm(F, F, P, B, kind, d, transpose, false, false);
  c->from_F(F, bd, bo, out, P, B);
}
template <typename T> static void do_dot(cmbl_ctx* ctx, int basis, const void* a, const void* b, int P, int B, double* out) {
  Ctx<T>* c = C<T>(ctx);
  if (basis == B_MAP) { c->dot_map((const T*)a, (const T*)b, P, B, out); return; }
  // Fourier-type bases: the weighted sum is invariant under the internal permutation, but lam depends on ky, so
  // bring both operands to F layout (rotation QU<->EB is orthogonal entry by entry, any Fourier basis works as is)
  const long sl = (long)P * B;
  c->tmpA.ensure(sizeof(cx<T>) * 2 * sl * c->plane());
  cx<T>* Fa = c->tmpA.template as<cx<T>>(); cx<T>* Fb = Fa + sl * c->plane();
  c->ref2F((const cx<T>*)a, Fa, sl); c->ref2F((const cx<T>*)b, Fb, sl);
  c->dot_F(Fa, Fb, P, B, out);
}
// logdet / tr of Diagonal(field): which = 0 logdet, 1 tr
template <typename T> static void do_diag_reduce(cmbl_ctx* ctx, int which, int basis, const void* d, int P, int B, double* out) {
  Ctx<T>* c = C<T>(ctx);
  if (basis == B_MAP) {
    if (which == 0) c->logdet_map((const T*)d, P, B, out); else c->tr_map((const T*)d, P, B, out);
    return;
  }
  const long sl = (long)P * B;
  c->tmpA.ensure(sizeof(cx<T>) * sl * c->plane());
  cx<T>* F = c->tmpA.template as<cx<T>>();
  c->ref2F((const cx<T>*)d, F, sl);                    // the lam-weighted sums are invariant under the internal permutation of x
  if (which == 0) c->logdet_Fc(F, P, B, out); else c->tr_Fc(F, P, B, out);
}
template <typename T> static void do_logdet(cmbl_ctx* ctx, const void* d, int nplanes, double* out) {
  Ctx<T>* c = C<T>(ctx);
  c->tmpB.ensure(sizeof(T) * nplanes * c->plane());
  c->ref2F_real((const T*)d, c->tmpB.template as<T>(), nplanes);
  c->logdet_F(c->tmpB.template as<T>(), nplanes, out);
}
template <typename T>
static void do_gradf(Dataset<T>& ds, Flow<T>& L, const void* f, const void* d, int zero_d, void* out, int B) {
  Ctx<T>* c = ds.c;
  const long n = ds.fsize(B);
  ds.cvt.ensure(sizeof(cx<T>) * 3 * n);
  cx<T>* fF = ds.cvt.template as<cx<T>>(); cx<T>* dF = fF + n; cx<T>* oF = dF + n;
  c->ref2F((const cx<T>*)f, fF, (long)ds.P * B);
  const cx<T>* dd = nullptr;
  if (!zero_d) {
    if (d) { c->ref2F((const cx<T>*)d, dF, (long)ds.P * B); dd = dF; }
    else { CMBL_REQUIRE(ds.Bd == B, ERR_SHAPE, "dataset data batch size differs from nbatch"); dd = ds.d_h.template as<cx<T>>(); }
  }
  ds.gradientf(L, fF, dd, oF, B);
  c->F2ref(oF, (cx<T>*)out, (long)ds.P * B);
}
template <typename T>
static void do_cg(Dataset<T>& ds, Flow<T>& L, const void* d, const void* fstart, double tol, int maxit, void* f_out, double* hist, int* nit, int B) {
  Ctx<T>* c = ds.c;
  const long n = ds.fsize(B);
  ds.cvt.ensure(sizeof(cx<T>) * 3 * n);
  cx<T>* dF = ds.cvt.template as<cx<T>>(); cx<T>* sF = dF + n; cx<T>* oF = sF + n;
  const cx<T>* dd;
  if (d) { c->ref2F((const cx<T>*)d, dF, (long)ds.P * B); dd = dF; }
  else { CMBL_REQUIRE(ds.Bd == B, ERR_SHAPE, "dataset data batch size differs from nbatch"); dd = ds.d_h.template as<cx<T>>(); }
  const cx<T>* fs = nullptr;
  if (fstart) { c->ref2F((const cx<T>*)fstart, sF, (long)ds.P * B); fs = sF; }
  *nit = ds.wiener_cg(L, dd, fs, tol, maxit, oF, hist, B);
  c->F2ref(oF, (cx<T>*)f_out, (long)ds.P * B);
  CMBL_HIP(hipStreamSynchronize(c->stream));
}
template <typename T>
static void do_lpm(Dataset<T>& ds, Flow<T>& L, const void* fo, const void* phio, double* lp, void* gfo, void* gphio, int B, int quirk) {
  Ctx<T>* c = ds.c;
  const long pl = c->plane();
  ds.cvt.ensure(sizeof(cx<T>) * 2 * B * pl);
  cx<T>* pF = ds.cvt.template as<cx<T>>(); cx<T>* gF = pF + (long)B * pl;
  c->ref2F((const cx<T>*)phio, pF, B);
  ds.logpdf_mixed(L, (const T*)fo, pF, lp, (T*)gfo, gfo ? gF : nullptr, B, quirk != 0);
  if (gfo) c->F2ref(gF, (cx<T>*)gphio, B);
  CMBL_HIP(hipStreamSynchronize(c->stream));
}

template <typename T> static Drivers<T>& drivers_of(std::map<const void*, std::unique_ptr<Drivers<T>>>& m, Dataset<T>& ds, Flow<T>& L) {
  auto& p = m[&L];
  if (!p) p = std::make_unique<Drivers<T>>(ds, L);
  return *p;
}
template <typename T>
static void do_hmc(Drivers<T>& dr, const void* fo, const void* phio, const void* mass, const void* white_p, const double* log_u, const uint64_t* seeds,
                   uint64_t step, int nleap, double eps, int always, int quirk, int B, void* phio_out, double* dH, int* accept) {
  Dataset<T>& ds = dr.ds;
  Ctx<T>* c = ds.c;
  const long pl = c->plane(), np = c->npix();
  ds.cvt.ensure(sizeof(cx<T>) * 2 * B * pl + sizeof(T) * (pl + (long)B * np));
  cx<T>* pF = ds.cvt.template as<cx<T>>(); cx<T>* oF = pF + (long)B * pl;
  T* mF = reinterpret_cast<T*>(oF + (long)B * pl); T* w = mF + pl;
  c->ref2F((const cx<T>*)phio, pF, B);
  c->ref2F_real((const T*)mass, mF, 1);
  const T* wp = (const T*)white_p;
  if (!wp) {                                                             // randn!(rng, ...) with the drivers' stream convention (rng.py)
    CMBL_REQUIRE(seeds != nullptr, ERR_ARG, "white_p == NULL needs seeds_host");
    c->randn(w, seeds, B, stream_id(STREAM_P, step), np);
    wp = w;
  }
  std::vector<double> lu(B);
  for (int b = 0; b < B; ++b) {
    if (log_u) lu[b] = log_u[b];
    else { CMBL_REQUIRE(seeds != nullptr, ERR_ARG, "log_u_host == NULL needs seeds_host"); lu[b] = std::log(philox_uniform(seeds[b], stream_id(STREAM_U, step))); }
  }
  dr.hmc_step((const T*)fo, pF, mF, wp, lu.data(), nleap, eps, always != 0, quirk != 0, B, oF, dH, accept);
  c->F2ref(oF, (cx<T>*)phio_out, B);
  CMBL_HIP(hipStreamSynchronize(c->stream));
}
template <typename T>
static void do_map_step(Drivers<T>& dr, const void* phi, const void* fstart, const void* hinv, double amax, double atol, double cg_tol, int cg_maxit, int quirk,
                        int B, void* f_out, void* phi_out, double* logpdf, double* alpha, int* ncg, int* nls) {
  Dataset<T>& ds = dr.ds;
  Ctx<T>* c = ds.c;
  const long pl = c->plane(), n = ds.fsize(B);
  CMBL_REQUIRE(ds.Bd == B, ERR_SHAPE, "dataset data batch size differs from nbatch");
  ds.cvt.ensure(sizeof(cx<T>) * (2 * B * pl + 2 * n) + sizeof(T) * 2 * pl);
  cx<T>* pF = ds.cvt.template as<cx<T>>(); cx<T>* oF = pF + (long)B * pl; cx<T>* sF = oF + (long)B * pl; cx<T>* fF = sF + n;
  T* hF = reinterpret_cast<T*>(fF + n); T* ones = hF + pl;
  c->ref2F((const cx<T>*)phi, pF, B);
  c->ref2F_real((const T*)hinv, hF, 1);
  const cx<T>* fs = nullptr;
  if (fstart) { c->ref2F((const cx<T>*)fstart, sF, (long)ds.P * B); fs = sF; }
  // G = I for the duration of the step (src/maximization.jl:146), whatever G the dataset carries
  std::vector<T> h1(pl, T(1));
  CMBL_HIP(hipMemcpyAsync(ones, h1.data(), sizeof(T) * pl, hipMemcpyHostToDevice, c->stream));
  CMBL_HIP(hipStreamSynchronize(c->stream));
  // (a dataset that never set G works as well: the slot is a one-plane diagonal for the duration of the call and is put back as it was)
  auto& g = ds.ops[OP_G_INV];
  struct Swap { decltype(g)& o; const T* d0; int np, kind; ~Swap() { o.d[0] = d0; o.nplanes = np; o.kind = kind; } } sw{g, g.d[0], g.nplanes, g.kind};
  if (g.nplanes == 0) { g.nplanes = 1; g.kind = 1; }
  g.d[0] = ones;
  std::vector<double> hist((size_t)cg_maxit * B);
  dr.map_joint_step(pF, fs, hF, amax, atol, cg_tol, cg_maxit, quirk != 0, B, fF, oF, logpdf, alpha, ncg, nls, hist.data());
  c->F2ref(fF, (cx<T>*)f_out, (long)ds.P * B);
  c->F2ref(oF, (cx<T>*)phi_out, B);
  CMBL_HIP(hipStreamSynchronize(c->stream));
}

template <typename T>
static void do_qe(Dataset<T>& ds, std::vector<std::unique_ptr<DevBuf>>& pool, int which, const double* Cf, const double* Cft, const double* Cn, const double* TF, const double* Cphi, int wiener,
                  const double* AL_in, void* phiqe_out, double* AL_out, int B) {
  Ctx<T>* c = ds.c;
  const long pl = c->plane();
  CMBL_REQUIRE(ds.Bd == B, ERR_SHAPE, "dataset data batch size differs from nbatch");
  // data components the estimator uses: TT -> T; EE -> E; EB -> E, B  (component index inside the dataset's I / EB / IEB data)
  const int P = ds.P;
  int comp[2] = {0, 0};
  if (which == 0) { CMBL_REQUIRE(P == 1 || P == 3, ERR_ARG, "TT needs a dataset with temperature"); comp[0] = 0; }
  else { CMBL_REQUIRE(P >= 2, ERR_ARG, "EE / EB need a dataset with polarisation"); comp[0] = P - 2; comp[1] = P - 1; }
  const int ncomp = which == 2 ? 2 : 1;
  ds.cvt.ensure(sizeof(cx<T>) * (long)ncomp * B * pl);
  cx<T>* dr[2] = {ds.cvt.template as<cx<T>>(), ds.cvt.template as<cx<T>>() + (long)B * pl};
  for (int k = 0; k < ncomp; ++k)
    for (int b = 0; b < B; ++b) c->F2ref(ds.d_h.template as<cx<T>>() + ((long)b * P + comp[k]) * pl, dr[k] + (long)b * pl, 1);
  const cx<T>* drc[2] = {dr[0], dr[1]};
  quadratic_estimate<T>(c, pool, which, B, drc, Cf, Cft, Cn, TF, Cphi, wiener != 0, AL_in, (cx<T>*)phiqe_out, AL_out);
  // the legs and products stay allocated for the next call (a one-off estimator otherwise spends half its time in hipMalloc), unless
  // they are large relative to the device: ~100 maps, 3 GB at 2048^2 in double precision
  size_t held = 0, free_b = 0, total_b = 0;
  for (const auto& b : pool) held += b->bytes;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); total_b = (size_t)16 << 30; }
  if (held > total_b / 16) pool.clear();                                 // 18 GB on a 288 GB part (round 4 dropped the pool above 1 GB: re-allocated on every call at 2048^2)
}

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
    if (dtype == CMBL_F32) h->p = std::make_unique<Ctx<float>>(Ny, Nx, theta, device, stream);
    else h->p = std::make_unique<Ctx<double>>(Ny, Nx, theta, device, stream);
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
    BY_DTYPE(ctx, h->f32 = std::make_unique<Flow<float>>(C<float>(ctx), nsteps), h->f64 = std::make_unique<Flow<double>>(C<double>(ctx), nsteps));
    *out = h.release();
  });
}
int cmbl_lenseflow_destroy(cmbl_flow* L) { return guard([&] { if (L) registry_drop_flow(L->f32.get(), L->f64.get()); delete L; }); }
int cmbl_lenseflow_set_phi(cmbl_flow* L, int basis, const void* phi, int nb) {
  return guard([&] {
    NOTNULL(L); NOTNULL(phi); BASIS_OK(basis); CMBL_REQUIRE(nb >= 1, ERR_SHAPE, "nbatch_phi >= 1");
    BY_DTYPE(L->ctx, L->f32->set_phi(basis, phi, nb), L->f64->set_phi(basis, phi, nb));
  });
}
int cmbl_lenseflow_apply(cmbl_flow* L, int mode, int bi, const void* in, int bo, void* out, int P, int B) {
  return guard([&] {
    NOTNULL(L); NOTNULL(in); NOTNULL(out); BASIS_OK(bi); BASIS_OK(bo); POLB_OK(P, B);
    CMBL_REQUIRE(mode >= 0 && mode <= 3, ERR_ARG, "bad flow mode");
    BY_DTYPE(L->ctx, L->f32->apply(mode, bi, in, bo, out, P, B), L->f64->apply(mode, bi, in, bo, out, P, B));
  });
}
int cmbl_lenseflow_grad(cmbl_flow* L, int mode, const void* f_end, int bdel, const void* delta, void* dphi, int bdf, void* df,
                        void* f_start, int P, int B, int quirk) {
  return guard([&] {
    NOTNULL(L); NOTNULL(f_end); NOTNULL(delta); NOTNULL(dphi); NOTNULL(df); BASIS_OK(bdel); BASIS_OK(bdf); POLB_OK(P, B);
    CMBL_REQUIRE(mode == CMBL_FLOW_FWD || mode == CMBL_FLOW_INV, ERR_ARG, "grad mode must be CMBL_FLOW_FWD or CMBL_FLOW_INV");
    BY_DTYPE(L->ctx, L->f32->grad(mode, f_end, bdel, delta, dphi, bdf, df, f_start, P, B, quirk != 0),
             L->f64->grad(mode, f_end, bdel, delta, dphi, bdf, df, f_start, P, B, quirk != 0));
  });
}

int cmbl_max_lensing_step(cmbl_flow* L, int basis, const void* phi, const void* eta, int nb, double* out) {
  return guard([&] {
    NOTNULL(L); NOTNULL(phi); NOTNULL(eta); NOTNULL(out); BASIS_OK(basis); CMBL_REQUIRE(nb >= 1, ERR_SHAPE, "nbatch >= 1");
    BY_DTYPE(L->ctx, L->f32->max_lensing_step(basis, phi, eta, nb, out), L->f64->max_lensing_step(basis, phi, eta, nb, out));
  });
}

// ---- small linear algebra / quadratic-estimate helpers ---------------------------------------------------
int cmbl_axpby(cmbl_ctx* ctx, int basis, const double* a, const void* x, const double* b, const void* y, void* out, int P, int B) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(a); NOTNULL(x); NOTNULL(out); BASIS_OK(basis); POLB_OK(P, B);
    CMBL_REQUIRE(y == nullptr || b != nullptr, ERR_ARG, "b is required when y is given");
    const long n = (basis == B_MAP ? ctx->p->npix() : 2 * ctx->p->plane()) * P;
    BY_DTYPE(ctx, C<float>(ctx)->lincomb((float*)out, (const float*)x, (const float*)y, a, b, n, B),
             C<double>(ctx)->lincomb((double*)out, (const double*)x, (const double*)y, a, b, n, B));
  });
}
int cmbl_qe_leg(cmbl_ctx* ctx, const void* in_fourier, int n, int p1, int p2, void* out_map, int B) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(in_fourier); NOTNULL(out_map); CMBL_REQUIRE(B >= 1 && n >= 0 && p1 >= 0 && p2 >= 0, ERR_ARG, "bad leg indices");
    BY_DTYPE(ctx, C<float>(ctx)->qe_leg((const cx<float>*)in_fourier, (float*)out_map, n, p1, p2, B),
             C<double>(ctx)->qe_leg((const cx<double>*)in_fourier, (double*)out_map, n, p1, p2, B));
  });
}
int cmbl_fourier_lmul(cmbl_ctx* ctx, const void* in_map, int p1, int p2, int take_abs, void* out_fourier, int B) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(in_map); NOTNULL(out_fourier); CMBL_REQUIRE(B >= 1 && p1 >= 0 && p2 >= 0, ERR_ARG, "bad exponents");
    BY_DTYPE(ctx, C<float>(ctx)->fourier_lmul((const float*)in_map, (cx<float>*)out_fourier, p1, p2, take_abs != 0, B),
             C<double>(ctx)->fourier_lmul((const double*)in_map, (cx<double>*)out_fourier, p1, p2, take_abs != 0, B));
  });
}
int cmbl_map_fma(cmbl_ctx* ctx, const void* a, const void* b, double scale, void* out, int accumulate, int nslices) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(a); NOTNULL(b); NOTNULL(out); CMBL_REQUIRE(nslices >= 1, ERR_SHAPE, "nslices >= 1");
    const long n = ctx->p->npix() * nslices;
    BY_DTYPE(ctx, C<float>(ctx)->map_fma((float*)out, (const float*)a, (const float*)b, scale, accumulate != 0, n),
             C<double>(ctx)->map_fma((double*)out, (const double*)a, (const double*)b, scale, accumulate != 0, n));
  });
}

int cmbl_randn(cmbl_ctx* ctx, const uint64_t* seeds, int nslots, uint64_t stream, void* out, long n_per_slot) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(seeds); NOTNULL(out);
    CMBL_REQUIRE(nslots >= 1 && n_per_slot >= 1, ERR_SHAPE, "nslots >= 1 and n_per_slot >= 1");
    BY_DTYPE(ctx, C<float>(ctx)->randn((float*)out, seeds, nslots, stream, n_per_slot),
             C<double>(ctx)->randn((double*)out, seeds, nslots, stream, n_per_slot));
  });
}

// ---- dataset ---------------------------------------------------------------------------------------
int cmbl_dataset_create(cmbl_ctx* ctx, int npol, cmbl_dataset** out) {
  return guard([&] {
    NOTNULL(ctx); NOTNULL(out);
    auto h = std::make_unique<cmbl_dataset>();
    h->ctx = ctx;
    BY_DTYPE(ctx, h->f32 = std::make_unique<Dataset<float>>(C<float>(ctx), npol), h->f64 = std::make_unique<Dataset<double>>(C<double>(ctx), npol));
    registry_add(h.get());
    *out = h.release();
  });
}
int cmbl_dataset_destroy(cmbl_dataset* ds) { return guard([&] { if (ds) registry_remove(ds); delete ds; }); }
int cmbl_dataset_set_op(cmbl_dataset* ds, int which, const void* planes, int nplanes) {
  return guard([&] { NOTNULL(ds); NOTNULL(planes); BY_DTYPE(ds->ctx, ds->f32->set_op(which, planes, nplanes), ds->f64->set_op(which, planes, nplanes)); });
}
int cmbl_dataset_set_data(cmbl_dataset* ds, const void* d, int B) {
  return guard([&] { NOTNULL(ds); NOTNULL(d); CMBL_REQUIRE(B >= 1, ERR_SHAPE, "nbatch >= 1"); BY_DTYPE(ds->ctx, ds->f32->set_data(d, B), ds->f64->set_data(d, B)); });
}
int cmbl_dataset_set_logdet(cmbl_dataset* ds, double v) {
  return guard([&] { NOTNULL(ds); BY_DTYPE(ds->ctx, ds->f32->logdet_sum = v, ds->f64->logdet_sum = v); });
}

int cmbl_gradientf_logpdf(cmbl_dataset* ds, cmbl_flow* L, const void* f, const void* d, int zero_d, void* out, int B) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(L); NOTNULL(f); NOTNULL(out); CMBL_REQUIRE(B >= 1, ERR_SHAPE, "nbatch >= 1");
    CMBL_REQUIRE(ds->ctx == L->ctx, ERR_ARG, "dataset and flow belong to different contexts");
    BY_DTYPE(ds->ctx, do_gradf<float>(*ds->f32, *L->f32, f, d, zero_d, out, B), do_gradf<double>(*ds->f64, *L->f64, f, d, zero_d, out, B));
  });
}

int cmbl_wiener_cg(cmbl_dataset* ds, cmbl_flow* L, const void* d, const void* fstart, double tol, int maxit, void* f_out,
                   double* hist, int* nit, int B) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(L); NOTNULL(f_out); NOTNULL(hist); NOTNULL(nit);
    CMBL_REQUIRE(B >= 1 && maxit >= 1, ERR_ARG, "nbatch >= 1 and maxit >= 1");
    CMBL_REQUIRE(ds->ctx == L->ctx, ERR_ARG, "dataset and flow belong to different contexts");
    BY_DTYPE(ds->ctx, do_cg<float>(*ds->f32, *L->f32, d, fstart, tol, maxit, f_out, hist, nit, B),
             do_cg<double>(*ds->f64, *L->f64, d, fstart, tol, maxit, f_out, hist, nit, B));
  });
}

int cmbl_logpdf_mixed(cmbl_dataset* ds, cmbl_flow* L, const void* fo, const void* phio, double* lp, int B) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(L); NOTNULL(fo); NOTNULL(phio); NOTNULL(lp); CMBL_REQUIRE(B >= 1, ERR_SHAPE, "nbatch >= 1");
    CMBL_REQUIRE(ds->ctx == L->ctx, ERR_ARG, "dataset and flow belong to different contexts");
    BY_DTYPE(ds->ctx, do_lpm<float>(*ds->f32, *L->f32, fo, phio, lp, nullptr, nullptr, B, 0), do_lpm<double>(*ds->f64, *L->f64, fo, phio, lp, nullptr, nullptr, B, 0));
  });
}
int cmbl_grad_logpdf_mixed(cmbl_dataset* ds, cmbl_flow* L, const void* fo, const void* phio, double* lp, void* gfo, void* gphio, int B, int quirk) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(L); NOTNULL(fo); NOTNULL(phio); NOTNULL(lp); NOTNULL(gfo); NOTNULL(gphio); CMBL_REQUIRE(B >= 1, ERR_SHAPE, "nbatch >= 1");
    CMBL_REQUIRE(ds->ctx == L->ctx, ERR_ARG, "dataset and flow belong to different contexts");
    BY_DTYPE(ds->ctx, do_lpm<float>(*ds->f32, *L->f32, fo, phio, lp, gfo, gphio, B, quirk), do_lpm<double>(*ds->f64, *L->f64, fo, phio, lp, gfo, gphio, B, quirk));
  });
}

int cmbl_hmc_step(cmbl_dataset* ds, cmbl_flow* L, const void* fo, const void* phio, const void* mass, const void* white_p, const double* log_u_host,
                  const uint64_t* seeds_host, uint64_t step, int nleap, double eps, int always_accept, int alias_quirk, int B,
                  void* phio_out, double* dH_host, int* accept_host) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(L); NOTNULL(fo); NOTNULL(phio); NOTNULL(mass); NOTNULL(phio_out); NOTNULL(dH_host); NOTNULL(accept_host);
    CMBL_REQUIRE(B >= 1 && B <= MAXBATCH && nleap >= 1, ERR_ARG, "1 <= nbatch <= 256 and nleap >= 1");
    CMBL_REQUIRE(ds->ctx == L->ctx, ERR_ARG, "dataset and flow belong to different contexts");
    BY_DTYPE(ds->ctx, do_hmc<float>(drivers_of(ds->drv32, *ds->f32, *L->f32), fo, phio, mass, white_p, log_u_host, seeds_host, step, nleap, eps, always_accept, alias_quirk, B, phio_out, dH_host, accept_host),
             do_hmc<double>(drivers_of(ds->drv64, *ds->f64, *L->f64), fo, phio, mass, white_p, log_u_host, seeds_host, step, nleap, eps, always_accept, alias_quirk, B, phio_out, dH_host, accept_host));
  });
}
int cmbl_map_joint_step(cmbl_dataset* ds, cmbl_flow* L, const void* phi, const void* fstart, const void* hinv, double alpha_max, double alpha_tol,
                        double cg_tol, int cg_maxit, int alias_quirk, int B, void* f_out, void* phi_out, double* logpdf_host, double* alpha_host,
                        int* ncg_host, int* nls_host) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(L); NOTNULL(phi); NOTNULL(hinv); NOTNULL(f_out); NOTNULL(phi_out); NOTNULL(logpdf_host); NOTNULL(alpha_host); NOTNULL(ncg_host); NOTNULL(nls_host);
    CMBL_REQUIRE(B >= 1 && B <= MAXBATCH && cg_maxit >= 1 && alpha_max > 0 && alpha_tol > 0, ERR_ARG, "1 <= nbatch <= 256, cg_maxit >= 1, alpha_max > 0, alpha_tol > 0");
    CMBL_REQUIRE(ds->ctx == L->ctx, ERR_ARG, "dataset and flow belong to different contexts");
    BY_DTYPE(ds->ctx, do_map_step<float>(drivers_of(ds->drv32, *ds->f32, *L->f32), phi, fstart, hinv, alpha_max, alpha_tol, cg_tol, cg_maxit, alias_quirk, B, f_out, phi_out, logpdf_host, alpha_host, ncg_host, nls_host),
             do_map_step<double>(drivers_of(ds->drv64, *ds->f64, *L->f64), phi, fstart, hinv, alpha_max, alpha_tol, cg_tol, cg_maxit, alias_quirk, B, f_out, phi_out, logpdf_host, alpha_host, ncg_host, nls_host));
  });
}

int cmbl_quadratic_estimate(cmbl_dataset* ds, int which, const double* Cf_host, const double* Cftilde_host, const double* Cn_host, const double* TF_host,
                            const double* Cphi_host, int wiener_filtered, const double* AL_in_host, void* phiqe_out, double* AL_out_host, int B) {
  return guard([&] {
    NOTNULL(ds); NOTNULL(Cf_host); NOTNULL(Cftilde_host); NOTNULL(Cn_host); NOTNULL(TF_host); NOTNULL(Cphi_host); NOTNULL(phiqe_out);
    CMBL_REQUIRE(which >= 0 && which <= 2, ERR_ARG, "which: 0 = TT, 1 = EE, 2 = EB (src/quadratic_estimate.jl:41: the others are not implemented by the reference either)");
    CMBL_REQUIRE(B >= 1, ERR_SHAPE, "nbatch >= 1");
    BY_DTYPE(ds->ctx, do_qe<float>(*ds->f32, ds->qe_pool, which, Cf_host, Cftilde_host, Cn_host, TF_host, Cphi_host, wiener_filtered, AL_in_host, phiqe_out, AL_out_host, B),
             do_qe<double>(*ds->f64, ds->qe_pool, which, Cf_host, Cftilde_host, Cn_host, TF_host, Cphi_host, wiener_filtered, AL_in_host, phiqe_out, AL_out_host, B));
  });
}

#ifdef CMBL_STAMPS
int cmbl_debug_stamps(unsigned long long* out_host, int n) {   // phase timestamps of the last k_delta_y launch (tools/gpu_stamps.py)
  return guard([&] { CMBL_HIP(hipDeviceSynchronize()); CMBL_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(cmbl::g_stamps), sizeof(unsigned long long) * n)); });
}
#endif

}  // extern "C"
