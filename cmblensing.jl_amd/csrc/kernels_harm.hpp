// Fused harmonic-space work between the flows: chains of Fourier-diagonal operators (C_l, beam, Fourier mask, mixing operators, the
// T-E block), the QU<->EB rotation, the data residual and the quadratic forms of the posterior, applied INSIDE the row pass that
// carries a field from the y-transformed ("mixed") layout through the x transform and back.
//
// Reference call sites (one array pass, one reduction pass each): the data model  M B L f  (src/dataset.jl:59-66), its transpose in
// gradientf_logpdf (src/dataset.jl:76-80), the Gaussian terms (src/distributions.jl:11-15), DiagOp `*` / `\` (src/specialops.jl:9-10),
// BlockDiagIEB (src/specialops.jl:80-83), the basis rotations (src/proj_lambert.jl:253-271), dot (src/proj_lambert.jl:322-325), and the
// vector updates / scalars of conjugate_gradient (src/numerical_algorithms.jl:95-125).
//
// Two carriers:
//   k_x_pw   : a row workgroup holds RPW ky-rows of ALL P pol slices of one batch slot in LDS:  [fft_x] -> pointwise functor on the
//              P components of every mode -> [ifft_x].  Input mixed or F, output mixed and / or whatever the functor stores (F layout).
//   k_pw_flat: the same functor interface on F-layout arrays without a transform (CG vector updates, the phi prior).
// Both can accumulate per-batch sums of lambda-weighted products (the Fourier inner product) and finish them in the launch itself:
// every block leaves a partial, the block that arrives last adds the partials in a fixed order (deterministic), scales, and runs the
// functor's scalar epilogue (e.g. alpha = res / pAp, the stop test of the CG) -- no reduction launches, no host round trip.
#pragma once
#include "kernels_fft.hpp"
#include "kernels_pointwise.hpp"

namespace cmbl {

// ---- accumulation in the mode of set_sum_accuracy_mode! (src/util.jl:288-316), selected at run time -----------------------------
template <typename T> struct SumAccRt {
  double d = 0; T s = 0, c = 0;
  __device__ __forceinline__ void add(int mode, T x) {
    if (mode == SUM_FLOAT64) d += (double)x;
    else if (mode == SUM_WORKING) s += x;
    else { const T t = s + x; c += (fabs(s) >= fabs(x)) ? ((s - t) + x) : ((x - t) + s); s = t; }
  }
  __device__ __forceinline__ double partial(int mode) const { return mode == SUM_FLOAT64 ? d : (mode == SUM_WORKING ? (double)s : (double)s + (double)c); }
};

// sum over the NT threads of the block; `red` = NT/64 + 1 values of scratch in LDS; the result is returned to EVERY thread.
template <typename A, int NT>
__device__ __forceinline__ A block_sum_all(A v, A* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) { A r = 0; for (int w = 0; w < NT / 64; ++w) r += red[w]; red[NT / 64] = r; }
  __syncthreads();
  const A r = red[NT / 64];
  __syncthreads();
  return r;
}

// Sums of a launch: every block leaves one partial per accumulator, part[(k * B + b) * nblk + g]; whoever needs the total adds the
// nblk partials in a fixed order (sum_partials) -- the consumer kernel in its prologue, or k_finish_parts.  No atomics, no ticket:
// 1024 same-address device-scope atomics cost more than the kernel they would finish (measured: 35 us for an 8 us kernel).
struct DotOut { double* part; int nblk; int mode; };

template <typename T, int NT, int NACC>
__device__ __forceinline__ void store_partials(const DotOut& o, SumAccRt<T> (&acc)[NACC], int b, int g, int B, double* scratch) {
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    double r;
    if (o.mode == SUM_WORKING) r = (double)block_sum_all<T, NT>((T)acc[k].partial(o.mode), reinterpret_cast<T*>(scratch));
    else r = block_sum_all<double, NT>(acc[k].partial(o.mode), scratch);
    if (threadIdx.x == 0) o.part[((size_t)k * B + b) * o.nblk + g] = r;
  }
}
// total of nblk partials, scaled, rounded as the accumulation mode prescribes (k_reduce_final); returned to every thread
template <typename T, int NT>
__device__ __forceinline__ double sum_partials(const double* __restrict__ p, int nblk, double scale, int mode, double* scratch) {
  if (mode == SUM_WORKING) {
    T a = 0;
    for (int i = threadIdx.x; i < nblk; i += NT) a += (T)p[i];
    return (double)(block_sum_all<T, NT>(a, reinterpret_cast<T*>(scratch)) * (T)scale);
  }
  double a = 0;
  for (int i = threadIdx.x; i < nblk; i += NT) a += p[i];
  const double r = block_sum_all<double, NT>(a, scratch) * scale;
  return mode == SUM_KAHAN ? (double)((T)r) : r;
}
// out[j] = total of region j: grid (nregions * B); regions may have different block counts
struct PartRegions { const double* part[4]; int nblk[4]; double* out[4]; };
template <typename T>
__global__ __launch_bounds__(NTP) void k_finish_parts(PartRegions r, int B, double scale, int mode) {
  __shared__ double scratch[NTP / 64 + 2];
  const int j = blockIdx.x / B, b = blockIdx.x % B;
  const double v = sum_partials<T, NTP>(r.part[j] + (size_t)b * r.nblk[j], r.nblk[j], scale, mode, scratch);
  if (threadIdx.x == 0) r.out[j][b] = v;
}

// ---- operators at one Fourier mode --------------------------------------------------------------------------------------------------
// kind 0 identity; 1 diagonal multiply d[p]; 2 IEB block (a b; c d) on (I, E), e on B (P == 3; transpose swaps b <-> c);
// 3 diagonal "\": nan2zero(v / d[p])  (src/specialops.jl:10).  Planes are real, F layout, shared by all batch slots.
template <typename T> struct OpRef { const T* d[5]; int kind; int transpose; };
template <typename T> __host__ __device__ inline OpRef<T> no_op() { OpRef<T> o{}; o.kind = 0; return o; }

template <typename T, int P>
__device__ __forceinline__ void op_apply(const OpRef<T>& o, long i, cx<T> (&v)[P]) {
  if (o.kind == 1) {
#pragma unroll
    for (int p = 0; p < P; ++p) v[p] = o.d[p][i] * v[p];
  } else if (o.kind == 3) {
#pragma unroll
    for (int p = 0; p < P; ++p) { const T d = o.d[p][i]; v[p] = mk<T>(nan2zero(v[p].x / d), nan2zero(v[p].y / d)); }
  } else if (o.kind == 2) {
    if constexpr (P == 3) {
      const T a = o.d[0][i], b = o.d[o.transpose ? 2 : 1][i], c = o.d[o.transpose ? 1 : 2][i], d = o.d[3][i], e = o.d[4][i];
      const cx<T> I = v[0], E = v[1];
      v[0] = a * I + b * E; v[1] = c * I + d * E; v[2] = e * v[2];
    }
  }
}
// QU -> EB: E = -Q c - U s, B = Q s - U c ; EB -> QU: Q = -E c + B s, U = -E s - B c   (src/proj_lambert.jl:253-271)
template <typename T, int P> __device__ __forceinline__ void rot_qu2eb(cx<T> (&v)[P], T c, T s) {
  if constexpr (P >= 2) {
    const cx<T> Q = v[P - 2], U = v[P - 1];
    v[P - 2] = mk<T>(-Q.x * c - U.x * s, -Q.y * c - U.y * s);
    v[P - 1] = mk<T>(Q.x * s - U.x * c, Q.y * s - U.y * c);
  }
}
template <typename T, int P> __device__ __forceinline__ void rot_eb2qu(cx<T> (&v)[P], T c, T s) {
  if constexpr (P >= 2) {
    const cx<T> E = v[P - 2], Bm = v[P - 1];
    v[P - 2] = mk<T>(-E.x * c + Bm.x * s, -E.y * c + Bm.y * s);
    v[P - 1] = mk<T>(-E.x * s - Bm.x * c, -E.y * s - Bm.y * c);
  }
}
template <typename T> struct HarmChain {
  static constexpr int MAXOPS = 4;
  int nops = 0;
  OpRef<T> op[MAXOPS];
  __host__ void push(const OpRef<T>& o) { if (o.kind != 0) op[nops++] = o; }
};
template <typename T, int P> __device__ __forceinline__ void chain_apply(const HarmChain<T>& ch, long i, cx<T> (&v)[P]) {
  for (int k = 0; k < ch.nops; ++k) op_apply<T, P>(ch.op[k], i, v);
}
template <typename T, int P> __device__ __forceinline__ T dot_term(const cx<T> (&a)[P], const cx<T> (&b)[P], int p, T lam) { return (a[p].x * b[p].x + a[p].y * b[p].y) * lam; }

// geometry a functor needs at a mode
template <typename T> struct ModeGeom { const T* cos2; const T* sin2; const T* lam; long plane; int Nx; };

// ---- functors ---------------------------------------------------------------------------------------------------------------------
// Interface: `Local prologue(b, B, scratch)` once per block (all threads; may sum a producer's partials), then per mode
//   rows: operator()(local, b, i, ky, v, acc, mode) with v = the P components held in LDS;   flat: operator()(local, b, i, ky, acc, mode)
// i = ky * Nx + x slot (index within an F-layout plane).  NACC sums per batch slot leave the launch as partials (DotOut).
struct NoLocal {};

// generic chain:  v <- scale * Rout( chain( Rin(v) [+ zscale * z] ) ), optionally stored to outF (F layout, after Rout)
template <typename T, int P> struct PwChain {
  static constexpr int NACC = 0;
  using Local = NoLocal;
  ModeGeom<T> g; HarmChain<T> ch;
  int in_qu, out_qu;
  const cx<T>* z; T zscale;                   // harmonic-basis array added after the input rotation (nullable)
  cx<T>* outF;                                // nullable
  T scale;
  __device__ __forceinline__ Local prologue(int, int, double*) const { return {}; }
  __device__ __forceinline__ void operator()(const Local&, int b, long i, int, cx<T> (&v)[P], SumAccRt<T>*, int) const {
    T c = 0, s = 0;
    if (P >= 2 && (in_qu || out_qu)) { c = g.cos2[i]; s = g.sin2[i]; }
    if (in_qu) rot_qu2eb<T, P>(v, c, s);
    if (z) {
#pragma unroll
      for (int p = 0; p < P; ++p) v[p] = v[p] + zscale * z[((long)b * P + p) * g.plane + i];
    }
    chain_apply<T, P>(ch, i, v);
    if (out_qu) rot_eb2qu<T, P>(v, c, s);
#pragma unroll
    for (int p = 0; p < P; ++p) { v[p] = scale * v[p]; if (outF) outF[((long)b * P + p) * g.plane + i] = v[p]; }
  }
};

// fhat (QU Fourier) -> f = D^-1 fhat (stored, harmonic), Cf^-1 f (stored), f' Cf^-1 f (sum 0) -> f in QU Fourier for the next flow
// (src/dataset.jl:111-117 unmix, src/distributions.jl:11-15)
template <typename T, int P> struct PwUnmixPrior {
  static constexpr int NACC = 1;
  using Local = NoLocal;
  ModeGeom<T> g; OpRef<T> Dinv, Cfinv;
  cx<T>* f_h; cx<T>* cfif;
  __device__ __forceinline__ Local prologue(int, int, double*) const { return {}; }
  __device__ __forceinline__ void operator()(const Local&, int b, long i, int ky, cx<T> (&v)[P], SumAccRt<T>* acc, int mode) const {
    T c = 0, s = 0;
    if (P >= 2) { c = g.cos2[i]; s = g.sin2[i]; }
    rot_qu2eb<T, P>(v, c, s);
    op_apply<T, P>(Dinv, i, v);
    cx<T> w[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { w[p] = v[p]; f_h[((long)b * P + p) * g.plane + i] = v[p]; }
    op_apply<T, P>(Cfinv, i, w);
    const T lam = g.lam[ky];
#pragma unroll
    for (int p = 0; p < P; ++p) { cfif[((long)b * P + p) * g.plane + i] = w[p]; acc[0].add(mode, dot_term<T, P>(v, w, p, lam)); }
    rot_eb2qu<T, P>(v, c, s);
  }
};

// data-space core of the model and of its transpose:  x (QU Fourier) -> Rin -> pre chain -> z = x - d -> w = Cn^-1 z -> z'w (sum 0)
// -> post chain -> scale -> Rout.   d == nullptr: no data term.  (src/dataset.jl:59-66,76-80)
template <typename T, int P> struct PwResid {
  static constexpr int NACC = 1;
  using Local = NoLocal;
  ModeGeom<T> g; HarmChain<T> pre, post; OpRef<T> Cninv;
  const cx<T>* d; int dB;                     // data (harmonic F); dB = 1: one data set for all batch slots
  T scale;
  cx<T>* outF;                                // nullable: the result (QU Fourier) also stored in the F layout
  __device__ __forceinline__ Local prologue(int, int, double*) const { return {}; }
  __device__ __forceinline__ void operator()(const Local&, int b, long i, int ky, cx<T> (&v)[P], SumAccRt<T>* acc, int mode) const {
    T c = 0, s = 0;
    if (P >= 2) { c = g.cos2[i]; s = g.sin2[i]; }
    rot_qu2eb<T, P>(v, c, s);
    chain_apply<T, P>(pre, i, v);
    if (d) {
#pragma unroll
      for (int p = 0; p < P; ++p) v[p] = v[p] - d[((long)(dB == 1 ? 0 : b) * P + p) * g.plane + i];
    }
    cx<T> z[P];
#pragma unroll
    for (int p = 0; p < P; ++p) z[p] = v[p];
    op_apply<T, P>(Cninv, i, v);
    const T lam = g.lam[ky];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[0].add(mode, dot_term<T, P>(z, v, p, lam));
    chain_apply<T, P>(post, i, v);
    rot_eb2qu<T, P>(v, c, s);
#pragma unroll
    for (int p = 0; p < P; ++p) { v[p] = scale * v[p]; if (outF) outF[((long)b * P + p) * g.plane + i] = v[p]; }
  }
};

// phi = G^-1 phi°, Cphi^-1 phi, phi' Cphi^-1 phi (sum 0)   (src/dataset.jl:111, src/distributions.jl:11-15); spin 0, flat
template <typename T> struct PwPhiPrior {
  static constexpr int NACC = 1;
  using Local = NoLocal;
  const T* lam; long plane; const T* Ginv; const T* Cpinv;
  const cx<T>* phio; cx<T>* phi; cx<T>* cpip;
  __device__ __forceinline__ Local prologue(int, int, double*) const { return {}; }
  __device__ __forceinline__ void operator()(const Local&, int b, long i, int ky, SumAccRt<T>* acc, int mode) const {
    const cx<T> v = Ginv[i] * phio[(long)b * plane + i], w = Cpinv[i] * v;
    phi[(long)b * plane + i] = v; cpip[(long)b * plane + i] = w;
    acc[0].add(mode, (v.x * w.x + v.y * w.y) * lam[ky]);
  }
};
// out = G^-1' (a + b - cpip)   (the last line of the posterior gradient); spin 0, flat
template <typename T> struct PwPhiGrad {
  static constexpr int NACC = 0;
  using Local = NoLocal;
  long plane; const T* Ginv; const cx<T>* a; const cx<T>* b2; const cx<T>* cpip; cx<T>* out;
  __device__ __forceinline__ Local prologue(int, int, double*) const { return {}; }
  __device__ __forceinline__ void operator()(const Local&, int b, long i, int, SumAccRt<T>*, int) const {
    const long o = (long)b * plane + i;
    out[o] = Ginv[i] * (a[o] + b2[o] - cpip[o]);
  }
};

// Conjugate gradient (src/numerical_algorithms.jl:73-134), scalars on the device.  Three launches per iteration after the flows; a sum
// is finished by the launch that consumes it:
//   PwCgAp : A p = Rin(Y) - Cf^-1 p ; partials of p'Ap                                                              (:100-101)
//   PwCgXr : [alpha = res / p'Ap]  x += alpha p ; r -= alpha A p ; z = Pinv r ; partials of r'z                     (:102-105)
//   PwCgP  : [res' = r'z ; beta = res' / res ; best-iterate / history / stop bookkeeping]  p = Pinv r + beta p ; bestx = x   (:106-125)
// The state (res, best, done, better) is double-buffered by iteration parity: block (0, 0) of PwCgP writes the next parity while the
// other blocks still read the current one.  `done` latches; from then on the launches change nothing.
struct CgScal {
  double *res, *best;                          // [2][MAXBATCH] (parity)
  double* hist;                                // [maxit][B]
  int *done, *better;                          // [2] (parity)
  int *nan, *nh;
};
template <typename T, int P> struct PwCgAp {
  static constexpr int NACC = 1;
  using Local = NoLocal;
  ModeGeom<T> g; OpRef<T> Cfinv;
  const cx<T>* Y; const cx<T>* p; cx<T>* Ap;
  __device__ __forceinline__ Local prologue(int, int, double*) const { return {}; }
  __device__ __forceinline__ void operator()(const Local&, int b, long i, int ky, SumAccRt<T>* acc, int mode) const {
    cx<T> v[P], w[P], pp[P];
#pragma unroll
    for (int q = 0; q < P; ++q) { v[q] = Y[((long)b * P + q) * g.plane + i]; w[q] = p[((long)b * P + q) * g.plane + i]; pp[q] = w[q]; }
    T c = 0, sn = 0;
    if (P >= 2) { c = g.cos2[i]; sn = g.sin2[i]; }
    rot_qu2eb<T, P>(v, c, sn);
    op_apply<T, P>(Cfinv, i, w);
    const T lam = g.lam[ky];
#pragma unroll
    for (int q = 0; q < P; ++q) { v[q] = v[q] - w[q]; Ap[((long)b * P + q) * g.plane + i] = v[q]; acc[0].add(mode, dot_term<T, P>(pp, v, q, lam)); }
  }
};
template <typename T, int P> struct PwCgXr {
  static constexpr int NACC = 1;
  struct Local { T alpha; int done; };
  ModeGeom<T> g; OpRef<T> Pinv;
  cx<T>* x; cx<T>* r; const cx<T>* p; const cx<T>* Ap;
  CgScal s; int par;
  DotOut pAp; double scale;                    // PwCgAp's partials
  __device__ __forceinline__ Local prologue(int b, int B, double* scratch) const {
    const double v = sum_partials<T, NTP>(pAp.part + (size_t)b * pAp.nblk, pAp.nblk, scale, pAp.mode, scratch);
    return Local{(T)(s.res[par * MAXBATCH + b] / v), s.done[par]};
  }
  __device__ __forceinline__ void operator()(const Local& l, int b, long i, int ky, SumAccRt<T>* acc, int mode) const {
    if (l.done) return;
    cx<T> rv[P], z[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const long o = ((long)b * P + q) * g.plane + i;
      x[o] = x[o] + l.alpha * p[o];
      rv[q] = r[o] - l.alpha * Ap[o]; r[o] = rv[q]; z[q] = rv[q];
    }
    op_apply<T, P>(Pinv, i, z);
    const T lam = g.lam[ky];
#pragma unroll
    for (int q = 0; q < P; ++q) acc[0].add(mode, dot_term<T, P>(rv, z, q, lam));
  }
};
template <typename T, int P> struct PwCgP {
  static constexpr int NACC = 0;
  struct Local { T beta; int skip, better; };
  ModeGeom<T> g; OpRef<T> Pinv;
  const cx<T>* x; const cx<T>* r; cx<T>* p; cx<T>* bestx;
  CgScal s; int par; double tol;
  DotOut rz; double scale;                     // PwCgXr's partials
  const double* rz_fin;                        // r'z of every slot, already summed (k_finish_parts), or nullptr: summed here
  __device__ __forceinline__ Local prologue(int b, int B, double* scratch) const {
    const int nx = par ^ 1;
    const bool writer = blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
    if (s.done[par]) {                         // latched: carry the state over, change nothing
      if (writer) { s.done[nx] = 1; s.better[nx] = 0; for (int bb = 0; bb < B; ++bb) { s.res[nx * MAXBATCH + bb] = s.res[par * MAXBATCH + bb]; s.best[nx * MAXBATCH + bb] = s.best[par * MAXBATCH + bb]; } }
      return Local{T(0), 1, 0};
    }
    int better = 1, done = 1, nanf = 0;
    double mine = 0;
    for (int bb = 0; bb < B; ++bb) {           // every block needs `better`, which looks at all batch slots (:111)
      const double r2 = rz_fin ? rz_fin[bb] : sum_partials<T, NTP>(rz.part + (size_t)bb * rz.nblk, rz.nblk, scale, rz.mode, scratch);
      nanf |= isnan(r2);
      better &= (r2 < s.best[par * MAXBATCH + bb]); done &= (r2 < tol);
      if (bb == b) mine = r2;
      if (writer) { s.res[nx * MAXBATCH + bb] = r2; s.hist[(long)(*s.nh) * B + bb] = r2; }
    }
    if (writer) {
      for (int bb = 0; bb < B; ++bb) s.best[nx * MAXBATCH + bb] = better ? s.res[nx * MAXBATCH + bb] : s.best[par * MAXBATCH + bb];
      *s.nh += 1; s.better[nx] = better; *s.nan |= nanf; s.done[nx] = done || nanf;
    }
    return Local{(T)(mine / s.res[par * MAXBATCH + b]), 0, better};
  }
  __device__ __forceinline__ void operator()(const Local& l, int b, long i, int, SumAccRt<T>*, int) const {
    if (l.skip) return;
    cx<T> z[P];
#pragma unroll
    for (int q = 0; q < P; ++q) z[q] = r[((long)b * P + q) * g.plane + i];
    op_apply<T, P>(Pinv, i, z);
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const long o = ((long)b * P + q) * g.plane + i;
      p[o] = z[q] + l.beta * p[o];
      if (l.better) bestx[o] = x[o];
    }
  }
};
// start of the solve (:78-93): p = z = Pinv r ; bestx = x ; partials of res = r'z (finished by k_cg_init)
template <typename T, int P> struct PwCgStart {
  static constexpr int NACC = 1;
  using Local = NoLocal;
  ModeGeom<T> g; OpRef<T> Pinv;
  const cx<T>* x; const cx<T>* r; cx<T>* p; cx<T>* bestx;
  __device__ __forceinline__ Local prologue(int, int, double*) const { return {}; }
  __device__ __forceinline__ void operator()(const Local&, int b, long i, int ky, SumAccRt<T>* acc, int mode) const {
    cx<T> rv[P], z[P];
#pragma unroll
    for (int q = 0; q < P; ++q) { rv[q] = r[((long)b * P + q) * g.plane + i]; z[q] = rv[q]; }
    op_apply<T, P>(Pinv, i, z);
    const T lam = g.lam[ky];
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const long o = ((long)b * P + q) * g.plane + i;
      p[o] = z[q]; bestx[o] = x[o];
      acc[0].add(mode, dot_term<T, P>(rv, z, q, lam));
    }
  }
};
template <typename T>
__global__ __launch_bounds__(NTP) void k_cg_init(CgScal s, DotOut d, double scale, int B) {
  __shared__ double scratch[NTP / 64 + 2];
  int nanf = 0;
  for (int b = 0; b < B; ++b) {
    const double v = sum_partials<T, NTP>(d.part + (size_t)b * d.nblk, d.nblk, scale, d.mode, scratch);
    if (threadIdx.x == 0) { s.res[b] = v; s.best[b] = v; s.hist[b] = v; }
    nanf |= isnan(v);
  }
  if (threadIdx.x == 0) { *s.nh = 1; s.done[0] = 0; s.better[0] = 1; *s.nan = nanf; }
}

// ---- carriers ------------------------------------------------------------------------------------------------------------------------
// flat: grid (nblk, B), NTP threads
template <typename T, typename PW>
__global__ __launch_bounds__(NTP) void k_pw_flat(PW pw, DotOut o, long plane, int Nx, int B) {
  __shared__ double scratch[NTP / 64 + 2];
  const int b = blockIdx.y;
  const typename PW::Local loc = pw.prologue(b, B, scratch);
  SumAccRt<T> acc[PW::NACC > 0 ? PW::NACC : 1];
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < plane; i += (long)gridDim.x * NTP) pw(loc, b, i, (int)((unsigned)i / (unsigned)Nx), acc, o.mode);
  if constexpr (PW::NACC > 0) store_partials<T, NTP, PW::NACC>(o, reinterpret_cast<SumAccRt<T> (&)[PW::NACC]>(acc), b, blockIdx.x, B, scratch);
}

// rows: grid = B * ceil(Nyh / RPW) row groups (row_group: the batch slot takes the place of the slice); P * RPW sequences (pol slice a,
// row r -> sequence a * RPW + r) of 128 threads each, so that a workgroup that holds all pol slices of its rows is no slower than a
// single-slice row pass.
//   IN_F = false: in = mixed layout, forward x transform first;  true: in = F layout (spectrum as is)
//   out_mixed != nullptr: inverse x transform of the functor's result, written in the mixed layout (x 1/Nx); herm: the imaginary parts
//   of the ky = 0 and ky = Ny/2 rows are dropped on the way out, so that the array IS the y transform of the real map c2r gives of it
struct XPwIo { int Nyh; int herm; };
__host__ __device__ constexpr int xpw_nt(int P, int rpw) { return P * rpw * ROW_RT; }
// rows per workgroup: four sequences where possible (P = 1: 4 rows, P = 2: 2 rows; P = 3: 2 rows = six sequences), fewer when LDS is short
template <typename T> __host__ __device__ constexpr int xpw_rpw(int lgnx, int P) {
  for (int rpw = (P == 1 ? 4 : 2); rpw >= 1; rpw >>= 1)
    if (((size_t)row_tw<T>(1 << lgnx) + (size_t)P * rpw * row_ld(1 << lgnx)) * sizeof(cx<T>) <= 160 * 1024 && xpw_nt(P, rpw) <= 1024) return rpw;
  return 0;
}
template <typename T, int LGNX, int RPW, int P, bool IN_F, typename PW>
__global__ __launch_bounds__(xpw_nt(P, RPW)) void k_x_pw(const cx<T>* __restrict__ in, cx<T>* __restrict__ out_mixed, const cx<T>* __restrict__ twX,
                                                         XPwIo io, PW pw, DotOut o, int B) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int Nx = 1 << LGNX, LD = row_ld(Nx), NT = xpw_nt(P, RPW), NT1 = row_nt(RPW), XLG = row_xlg(LGNX);
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + row_tw<T>(Nx);
  const int Nyh = io.Nyh, NyhP = mixed_rows(Nyh);
  const RowGroup rg = row_group<RPW>(blockIdx.x, Nyh, gridDim.x);             // rg.sl = batch slot
  const int b = rg.sl;
  const int set = threadIdx.x / NT1, tid = threadIdx.x % NT1;               // this thread's pol slice during loads / stores
  const size_t mpl = (size_t)NyhP * Nx, fpl = (size_t)Nyh * Nx;
  TwStage<T, NT, row_tw<T>(Nx)> twr;
  twr.issue(twX);
  if constexpr (IN_F) rows_load_F<T, LGNX, RPW>(s + set * RPW * LD, in + ((size_t)b * P + set) * fpl + (size_t)rg.ky0 * Nx, rg.nr, tid);
  else {
    cx<T>* const sa[1] = {s + set * RPW * LD};
    const cx<T>* const ga[1] = {in + ((size_t)b * P + set) * mpl};
    rows_load_mixed_dif<T, LGNX, RPW, 1>(sa, ga, twX, NyhP, rg.ky0, rg.nr, tid);
  }
  twr.commit(tw);
  __syncthreads();
  const WorkSeqs<ROW_RT, RPW, row_tw_quarter<T>(LGNX)> wk{rg.nr};
  if constexpr (!IN_F) { fft_dif_w<T, LD, LGNX, LGNX, XLG, 1>(s, wk, tw); __syncthreads(); }
  const typename PW::Local loc = pw.prologue(b, B, reinterpret_cast<double*>(smem));
  SumAccRt<T> acc[PW::NACC > 0 ? PW::NACC : 1];
  for (int u = threadIdx.x; u < RPW * Nx; u += NT) {
    const int r = u >> LGNX, x = u & (Nx - 1);
    if (r < rg.nr) {
      cx<T> v[P];
#pragma unroll
      for (int p = 0; p < P; ++p) v[p] = s[(p * RPW + r) * LD + pad(x)];
      pw(loc, b, (long)(rg.ky0 + r) * Nx + x, rg.ky0 + r, v, acc, o.mode);
#pragma unroll
      for (int p = 0; p < P; ++p) s[(p * RPW + r) * LD + pad(x)] = v[p];
    }
  }
  __syncthreads();
  if (out_mixed) {
    fft_dit_w<T, LD, LGNX, LGNX, XLG, 1>(s, wk, tw);
    __syncthreads();
    rows_store_mixed_dit<T, LGNX, RPW>(s + set * RPW * LD, out_mixed + ((size_t)b * P + set) * mpl, tw, NyhP, rg.ky0, rg.nr, T(1) / T(Nx), io.herm ? Nyh - 1 : -1, tid);
  }
  if constexpr (PW::NACC > 0) {
    __syncthreads();
    store_partials<T, NT, PW::NACC>(o, reinterpret_cast<SumAccRt<T> (&)[PW::NACC]>(acc), b, rg.ky0 / RPW, B, reinterpret_cast<double*>(smem));
  }
}

// delta-phi epilogue on rows (src/lenseflow.jl:198-206, see k_dphi_combine): the x transforms of the five y-transformed reduced maps
// and their combination  i lx F1 + i ly F2 - lx^2 FA - lx ly FB - ly^2 FC  in one launch; optionally the last line of the posterior
// gradient on top:  out = Ginv * (that + add1 - sub1).   in: mixed [5][B] slices; out: F [B].
template <typename T> struct DphiTail { const cx<T>* add1; const cx<T>* sub1; const T* Ginv; };
template <typename T, int LGNX, int RPW>
__global__ __launch_bounds__(row_nt(RPW)) void k_x_dphi(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, const cx<T>* __restrict__ twX,
                                                        const T* __restrict__ lx_r, const T* __restrict__ ly, int Nyh, int B, DphiTail<T> tail) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int Nx = 1 << LGNX, LD = row_ld(Nx), NT = row_nt(RPW), XLG = row_xlg(LGNX);
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + row_tw<T>(Nx);
  const int NyhP = mixed_rows(Nyh);
  const RowGroup rg = row_group<RPW>(blockIdx.x, Nyh, gridDim.x);
  const int b = rg.sl;
  const size_t mpl = (size_t)NyhP * Nx, fpl = (size_t)Nyh * Nx;
  TwStage<T, NT, row_tw<T>(Nx)> twr;
  twr.issue(twX);
  {
    cx<T>* sa[5]; const cx<T>* ga[5];
#pragma unroll
    for (int p = 0; p < 5; ++p) { sa[p] = s + p * RPW * LD; ga[p] = in + ((size_t)p * B + b) * mpl; }
    rows_load_mixed_dif<T, LGNX, RPW, 5>(reinterpret_cast<cx<T>* const (&)[5]>(sa), reinterpret_cast<const cx<T>* const (&)[5]>(ga), twX, NyhP, rg.ky0, rg.nr);
  }
  twr.commit(tw);
  __syncthreads();
  fft_dif_w<T, LD, LGNX, LGNX, XLG, 1>(s, WorkRows<ROW_RT, RPW, row_tw_quarter<T>(LGNX)>{5, rg.nr}, tw);
  __syncthreads();
  for (int u = threadIdx.x; u < RPW * Nx; u += NT) {
    const int r = u >> LGNX, x = u & (Nx - 1);
    if (r < rg.nr) {
      const T lx = lx_r[x], l_y = ly[rg.ky0 + r];
      const cx<T> f1 = s[(0 * RPW + r) * LD + pad(x)], f2 = s[(1 * RPW + r) * LD + pad(x)], fa = s[(2 * RPW + r) * LD + pad(x)],
                  fb = s[(3 * RPW + r) * LD + pad(x)], fc = s[(4 * RPW + r) * LD + pad(x)];
      const T re = -lx * f1.y - l_y * f2.y - lx * lx * fa.x - lx * l_y * fb.x - l_y * l_y * fc.x;
      const T im = lx * f1.x + l_y * f2.x - lx * lx * fa.y - lx * l_y * fb.y - l_y * l_y * fc.y;
      const size_t oi = (size_t)b * fpl + (size_t)(rg.ky0 + r) * Nx + x;
      cx<T> v = mk<T>(re, im);
      if (tail.Ginv) v = tail.Ginv[(size_t)(rg.ky0 + r) * Nx + x] * (v + tail.add1[oi] - tail.sub1[oi]);
      out[oi] = v;
    }
  }
}

}  // namespace cmbl
