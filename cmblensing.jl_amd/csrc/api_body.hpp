// The typed bodies behind the C ABI (declared in api_decl.hpp); included by tu_main_{f32,f64}.hip only, which instantiate them.
#pragma once
#include "api_decl.hpp"

namespace cmbl {

template <typename T> std::unique_ptr<Flow<T>>& flow_of(cmbl_flow* L) { if constexpr (sizeof(T) == 4) return L->f32; else return L->f64; }
template <typename T> std::unique_ptr<Dataset<T>>& ds_of(cmbl_dataset* d) { if constexpr (sizeof(T) == 4) return d->f32; else return d->f64; }
template <typename T> std::map<const void*, std::unique_ptr<Drivers<T>>>& drv_of(cmbl_dataset* d) { if constexpr (sizeof(T) == 4) return d->drv32; else return d->drv64; }
template <typename T> Ctx<T>* C(cmbl_ctx* c) { return static_cast<Ctx<T>*>(c->p.get()); }

template <typename T> void do_convert(cmbl_ctx* ctx, int bi, const void* in, int bo, void* out, int P, int B) {
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
void do_diag(cmbl_ctx* ctx, int kind, int bd, const void* diag, int nplanes, bool transpose, int bi, const void* in, int bo, void* out, int P, int B) {
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
template <typename T> void do_dot(cmbl_ctx* ctx, int basis, const void* a, const void* b, int P, int B, double* out) {
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
template <typename T> void do_diag_reduce(cmbl_ctx* ctx, int which, int basis, const void* d, int P, int B, double* out) {
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
template <typename T> void do_logdet(cmbl_ctx* ctx, const void* d, int nplanes, double* out) {
  Ctx<T>* c = C<T>(ctx);
  c->tmpB.ensure(sizeof(T) * nplanes * c->plane());
  c->ref2F_real((const T*)d, c->tmpB.template as<T>(), nplanes);
  c->logdet_F(c->tmpB.template as<T>(), nplanes, out);
}
template <typename T>
void do_gradf(cmbl_dataset* dsh, cmbl_flow* Lh, const void* f, const void* d, int zero_d, void* out, int B) {
  Dataset<T>& ds = *ds_of<T>(dsh); Flow<T>& L = *flow_of<T>(Lh);
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
void do_cg(cmbl_dataset* dsh, cmbl_flow* Lh, const void* d, const void* fstart, double tol, int maxit, void* f_out, double* hist, int* nit, int B) {
  Dataset<T>& ds = *ds_of<T>(dsh); Flow<T>& L = *flow_of<T>(Lh);
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
void do_lpm(cmbl_dataset* dsh, cmbl_flow* Lh, const void* fo, const void* phio, double* lp, void* gfo, void* gphio, int B, int quirk) {
  Dataset<T>& ds = *ds_of<T>(dsh); Flow<T>& L = *flow_of<T>(Lh);
  Ctx<T>* c = ds.c;
  const long pl = c->plane();
  ds.cvt.ensure(sizeof(cx<T>) * 2 * B * pl);
  cx<T>* pF = ds.cvt.template as<cx<T>>(); cx<T>* gF = pF + (long)B * pl;
  c->ref2F((const cx<T>*)phio, pF, B);
  ds.logpdf_mixed(L, (const T*)fo, pF, lp, (T*)gfo, gfo ? gF : nullptr, B, quirk != 0);
  if (gfo) c->F2ref(gF, (cx<T>*)gphio, B);
  CMBL_HIP(hipStreamSynchronize(c->stream));
}

template <typename T> Drivers<T>& drivers_of(std::map<const void*, std::unique_ptr<Drivers<T>>>& m, Dataset<T>& ds, Flow<T>& L) {
  auto& p = m[&L];
  if (!p) p = std::make_unique<Drivers<T>>(ds, L);
  return *p;
}
template <typename T>
void do_hmc(cmbl_dataset* dsh, cmbl_flow* Lh, const void* fo, const void* phio, const void* mass, const void* white_p, const double* log_u, const uint64_t* seeds,
                   uint64_t step, int nleap, double eps, int always, int quirk, int B, void* phio_out, double* dH, int* accept) {
  Drivers<T>& dr = drivers_of(drv_of<T>(dsh), *ds_of<T>(dsh), *flow_of<T>(Lh));
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
void do_map_step(cmbl_dataset* dsh, cmbl_flow* Lh, const void* phi, const void* fstart, const void* hinv, double amax, double atol, double cg_tol, int cg_maxit, int quirk,
                        int B, void* f_out, void* phi_out, double* logpdf, double* alpha, int* ncg, int* nls) {
  Drivers<T>& dr = drivers_of(drv_of<T>(dsh), *ds_of<T>(dsh), *flow_of<T>(Lh));
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
void do_qe(cmbl_dataset* dsh, int which, const double* Cf, const double* Cft, const double* Cn, const double* TF, const double* Cphi, int wiener,
                  const double* AL_in, void* phiqe_out, double* AL_out, int B) {
  Dataset<T>& ds = *ds_of<T>(dsh); auto& pool = dsh->qe_pool;
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


// ---- the members the entry points used to call directly (api.hip must not instantiate a launching member: see api_decl.hpp) ----------
template <typename T> CtxBase* do_ctx_create(int Ny, int Nx, double theta, int device, void* stream) { return new Ctx<T>(Ny, Nx, theta, device, stream); }
template <typename T> void do_axpby(cmbl_ctx* ctx, const double* a, const void* x, const double* b, const void* y, void* out, long n, int B) {
  C<T>(ctx)->lincomb((T*)out, (const T*)x, (const T*)y, a, b, n, B);
}
template <typename T> void do_qe_leg(cmbl_ctx* ctx, const void* in_fourier, int n, int p1, int p2, void* out_map, int B) {
  C<T>(ctx)->qe_leg((const cx<T>*)in_fourier, (T*)out_map, n, p1, p2, B);
}
template <typename T> void do_fourier_lmul(cmbl_ctx* ctx, const void* in_map, int p1, int p2, int take_abs, void* out_fourier, int B) {
  C<T>(ctx)->fourier_lmul((const T*)in_map, (cx<T>*)out_fourier, p1, p2, take_abs != 0, B);
}
template <typename T> void do_map_fma(cmbl_ctx* ctx, const void* a, const void* b, double scale, void* out, int accumulate, long n) {
  C<T>(ctx)->map_fma((T*)out, (const T*)a, (const T*)b, scale, accumulate != 0, n);
}
template <typename T> void do_randn(cmbl_ctx* ctx, const uint64_t* seeds, int nslots, uint64_t stream, void* out, long n_per_slot) {
  C<T>(ctx)->randn((T*)out, seeds, nslots, stream, n_per_slot);
}
template <typename T> void do_flow_create(cmbl_flow* h, int nsteps) { flow_of<T>(h) = std::make_unique<Flow<T>>(C<T>(h->ctx), nsteps); }
template <typename T> void do_flow_set_phi(cmbl_flow* L, int basis, const void* phi, int nb) { flow_of<T>(L)->set_phi(basis, phi, nb); }
template <typename T> void do_flow_apply(cmbl_flow* L, int mode, int bi, const void* in, int bo, void* out, int P, int B) { flow_of<T>(L)->apply(mode, bi, in, bo, out, P, B); }
template <typename T> void do_flow_grad(cmbl_flow* L, int mode, const void* f_end, int bdel, const void* delta, void* dphi, int bdf, void* df, void* f_start, int P, int B, int quirk) {
  flow_of<T>(L)->grad(mode, f_end, bdel, delta, dphi, bdf, df, f_start, P, B, quirk != 0);
}
template <typename T> void do_max_lensing_step(cmbl_flow* L, int basis, const void* phi, const void* eta, int nb, double* out) { flow_of<T>(L)->max_lensing_step(basis, phi, eta, nb, out); }
template <typename T> void do_dataset_create(cmbl_dataset* h, int npol) { ds_of<T>(h) = std::make_unique<Dataset<T>>(C<T>(h->ctx), npol); }
template <typename T> void do_dataset_set_op(cmbl_dataset* ds, int which, const void* planes, int nplanes) { ds_of<T>(ds)->set_op(which, planes, nplanes); }
template <typename T> void do_dataset_set_data(cmbl_dataset* ds, const void* d, int B) { ds_of<T>(ds)->set_data(d, B); }

}  // namespace cmbl
