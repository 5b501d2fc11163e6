// Small-map flows of Flow<T> (declared in engine.hpp): the launches of kernels_small.hpp.  Defined out of class and instantiated explicitly by
// tu_small_{f32,f64}.hip (api_decl.hpp has the map of the build).
#pragma once
#include "engine.hpp"
#include "kernels_small.hpp"

namespace cmbl {

// shapes with a small-map kernel: 32 <= Ny, Nx <= 128, powers of two, the half plane pair within 160 KB of LDS in this precision
#ifndef CMBL_SMALL_LIST
#define CMBL_SMALL_LIST(X) X(5, 5) X(5, 6) X(5, 7) X(6, 5) X(6, 6) X(6, 7) X(7, 5) X(7, 6) X(7, 7)
#endif

// Option small_flow: 0 = never; 1 (default) = where it measured faster at EVERY batch size -- up to 64 x 64 pixels, by shape alone so that a
// batch slot's result never depends on the batch it rides in (profiles/r06_ab_small_flow.txt: 64^2 L*f 0.204 -> 0.147 ms at B = 1, 0.337 ->
// 0.148 ms at B = 64; 64 x 128 is 18 % slower at B = 1 and 1.6 x faster at B = 64; 128^2 3 x slower at B = 1, equal at B = 64);
// 2 = wherever a kernel exists (batched small-map workloads, tests).
template <typename T>
bool Flow<T>::small_ok() const {
  if (!c->opts.small_flow || c->generic || !use_pcache) return false;
  if (c->opts.small_flow == 1 && c->npix() > 4096) return false;
  const int lgny = c->lgM + 1, lgnx = c->lgNx;
#define CMBL_X(a, b) if (lgny == a && lgnx == b) return SmallGeom<T, a, b>::fits;
  CMBL_SMALL_LIST(CMBL_X)
#undef CMBL_X
  return false;
}

template <typename T>
static SmallArgs<T> small_args(const Flow<T>& L, const void* in, void* out, int P, int k0, int dir) {
  const Ctx<T>* c = L.c;
  SmallArgs<T> a{};
  a.in = (const T*)in; a.out = (T*)out; a.pcache = L.pcache.template as<T>();
  a.tw = (c->Ny >= c->Nx ? c->twY : c->twX).template as<cx<T>>();
  a.lx_r = c->lx_r.template as<T>(); a.ly = c->ly.template as<T>();
  a.n = L.n; a.P = P; a.Bphi = L.Bphi; a.k0 = k0; a.dir = dir;
  const double h = (double)dir / L.n;                                       // rounded like Flow::coef
  a.hhalf = (T)(h / 2); a.hfull = (T)h; a.h6 = (T)(h / 6);
  return a;
}

template <typename T>
void Flow<T>::small_flow_map(const T* in, T* out, int P, int B, bool inverse) {
  const long slices = (long)P * B;
  const SmallArgs<T> a = small_args(*this, in, out, P, inverse ? 2 * n : 0, inverse ? -1 : 1);
  const int lgny = c->lgM + 1, lgnx = c->lgNx;
#define CMBL_X(ly_, lx_) if (lgny == ly_ && lgnx == lx_) { using G = SmallGeom<T, ly_, lx_>; if constexpr (G::fits) { \
    CMBL_LAUNCH_NT(c, K_FLOW_Y, G::NT, (k_small_flow<T, ly_, lx_>), dim3((unsigned)slices), G::lds, c->stream, a); return; } }
  CMBL_SMALL_LIST(CMBL_X)
#undef CMBL_X
  fail(ERR_STATE, "no small-map kernel for this shape");
}

template <typename T>
void Flow<T>::small_flow_adj(const cx<T>* in, cx<T>* out, int P, int B, bool inverse) {
  const long slices = (long)P * B;
  const SmallArgs<T> a = small_args(*this, in, out, P, inverse ? 0 : 2 * n, inverse ? 1 : -1);
  const int lgny = c->lgM + 1, lgnx = c->lgNx;
#define CMBL_X(ly_, lx_) if (lgny == ly_ && lgnx == lx_) { using G = SmallGeom<T, ly_, lx_>; if constexpr (G::fits) { \
    CMBL_LAUNCH_NT(c, K_ADJ_Y, G::NT, (k_small_adj<T, ly_, lx_>), dim3((unsigned)slices), G::lds, c->stream, a); return; } }
  CMBL_SMALL_LIST(CMBL_X)
#undef CMBL_X
  fail(ERR_STATE, "no small-map kernel for this shape");
}

// the delta flow in one launch: up to 64 x 64 pixels in single, 32 x 64 in double precision (the register state of both parts; SmallGeom::delta_fits)
template <typename T>
bool Flow<T>::small_delta_ok() const { return small_ok() && c->npix() * (long)sizeof(T) <= 4096 * 4; }

template <typename T>
void Flow<T>::small_flow_delta(T* f, cx<T>* df, cx<T>* dphi, int P, int B, bool forward_primal, bool alias_quirk, const DphiTail<T>* tail) {
  const long slices = (long)P * B, pl = c->plane(), np = c->npix();
  const int nst = 4 * n;
  Wst.ensure(sizeof(T) * (size_t)nst * 2 * slices * np);
  U5.ensure(sizeof(T) * 5 * B * np); F5.ensure(sizeof(cx<T>) * 5 * B * pl); tcbuf.ensure(sizeof(T) * 2 * nst);
  const double t0 = forward_primal ? 1.0 : 0.0, h = (forward_primal ? -1.0 : 1.0) / n;
  tc_host.resize(2 * (size_t)nst);
  int it = 0;
  for (int step = 0; step < n; ++step)
    for (int stage = 1; stage <= 4; ++stage, ++it) {                      // the (t_s, c_s) table of the quadrature, as Flow::flow_delta fills it
      const RKCoef<T> rk = coef(step, stage, t0, h, step == n - 1 && stage == 4);
      tc_host[2 * it] = rk.t;
      tc_host[2 * it + 1] = (T)((stage == 1 || stage == 4 ? 1.0 : 2.0) * h / 6);
    }
  SmallDeltaArgs<T> d{};
  d.a = small_args(*this, f, f, P, forward_primal ? 2 * n : 0, forward_primal ? -1 : 1);
  d.df = df; d.wst = Wst.as<T>(); d.slices = slices;
  const int lgny = c->lgM + 1, lgnx = c->lgNx;
  bool done = false;
#define CMBL_X(ly_, lx_) if (!done && lgny == ly_ && lgnx == lx_) { using G = SmallGeom<T, ly_, lx_>; if constexpr (G::fits && G::delta_fits) { \
    CMBL_LAUNCH_NT(c, K_DELTA_Y, G::NT, (k_small_delta<T, ly_, lx_>), dim3((unsigned)slices), G::lds, c->stream, d); done = true; } }
  CMBL_SMALL_LIST(CMBL_X)
#undef CMBL_X
  if (!done) fail(ERR_STATE, "no small-map delta kernel for this shape");
  dphi_finish(dphi, P, B, nst, alias_quirk, tail);
}

#define CMBL_INSTANTIATE_SMALL(T)                                                     \
  template bool Flow<T>::small_ok() const;                                            \
  template void Flow<T>::small_flow_map(const T*, T*, int, int, bool);                \
  template void Flow<T>::small_flow_adj(const cx<T>*, cx<T>*, int, int, bool);       \
  template bool Flow<T>::small_delta_ok() const;                                      \
  template void Flow<T>::small_flow_delta(T*, cx<T>*, cx<T>*, int, int, bool, bool, const DphiTail<T>*);

}  // namespace cmbl
