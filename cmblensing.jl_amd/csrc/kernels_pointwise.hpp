// Pointwise kernels on the internal layouts: harmonic-basis operators (Cl / mask / beam / TE block applies
// fused with the QU<->EB rotation), linear combinations with per-batch scalars, per-batch reductions.
#pragma once
#include "common.hpp"

namespace cmbl {

constexpr int MAXB = 16;                       // per-batch scalars are passed by value in chunks of MAXB
constexpr int MAXBATCH = 256;                  // most batch slots (chains per GPU) a reduction / CG / posterior call takes: sizes the per-slot scalar buffers
template <typename T> struct BScal { T v[MAXB]; };

// ---------------------------------------------------------------------------------------------
// Harmonic operator application in F layout (src/specialops.jl:9-10,80-83; src/proj_lambert.jl:253-271).
//   out = alpha * z + beta * R_out( Op( R_in(in) ) )
// R_in : QU->EB rotation if in_qu (E=-Qc-Us, B=Qs-Uc), R_out: EB->QU if out_qu (Q=-Ec+Bs, U=-Es-Bc).
// Op   : kind 0 identity; kind 1 diagonal multiply d[p]; kind 2 IEB block (a b; c d) on (I,E), e on B  (P==3);
//        kind 3 diagonal "\" : nan2zero(in / d[p])   (src/specialops.jl:10)
// transpose swaps b<->c.  Operator arrays are real, F layout [ky][xr], shared by all batch slots.
template <typename T> struct HarmOpArgs {
  const cx<T>* in; cx<T>* out; const cx<T>* z;
  const T* cos2; const T* sin2;
  const T* d[5];
  int kind, in_qu, out_qu, transpose;
  T alpha, beta;
  long plane;                                  // Nyh*Nx
  int B;
};

template <typename T> __device__ __forceinline__ T nan2zero(T v) { return isfinite(v) ? v : T(0); }

template <typename T, int P>
__global__ __launch_bounds__(NTP) void k_harm_apply(HarmOpArgs<T> a) {
  const long i = (long)blockIdx.x * NTP + threadIdx.x;
  if (i >= a.plane) return;
  T c = 0, s = 0;
  if (P >= 2 && (a.in_qu || a.out_qu)) { c = a.cos2[i]; s = a.sin2[i]; }
  T d[5];
  const int nd = (a.kind == 2) ? 5 : ((a.kind == 1 || a.kind == 3) ? P : 0);
#pragma unroll
  for (int k = 0; k < 5; ++k) d[k] = (k < nd) ? a.d[k][i] : T(0);
  if (a.kind == 2 && a.transpose) { T t = d[1]; d[1] = d[2]; d[2] = t; }
  for (int b = 0; b < a.B; ++b) {
    const long base = (long)b * P * a.plane + i;
    cx<T> v[P];
#pragma unroll
    for (int p = 0; p < P; ++p) v[p] = a.in[base + p * a.plane];
    if (P >= 2 && a.in_qu) {
      cx<T> Q = v[P - 2], U = v[P - 1];
      v[P - 2] = mk<T>(-Q.x * c - U.x * s, -Q.y * c - U.y * s);
      v[P - 1] = mk<T>(Q.x * s - U.x * c, Q.y * s - U.y * c);
    }
    if (a.kind == 1) {
#pragma unroll
      for (int p = 0; p < P; ++p) v[p] = d[p] * v[p];
    } else if (a.kind == 3) {
#pragma unroll
      for (int p = 0; p < P; ++p) v[p] = mk<T>(nan2zero(v[p].x / d[p]), nan2zero(v[p].y / d[p]));
    } else if (a.kind == 2 && P == 3) {
      cx<T> I = v[0], E = v[1];
      v[0] = d[0] * I + d[1] * E;
      v[1] = d[2] * I + d[3] * E;
      v[2] = d[4] * v[2];
    }
    if (P >= 2 && a.out_qu) {
      cx<T> E = v[P - 2], Bm = v[P - 1];
      v[P - 2] = mk<T>(-E.x * c + Bm.x * s, -E.y * c + Bm.y * s);
      v[P - 1] = mk<T>(-E.x * s - Bm.x * c, -E.y * s - Bm.y * c);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      cx<T> r = a.beta * v[p];
      if (a.z) r = r + a.alpha * a.z[base + p * a.plane];
      a.out[base + p * a.plane] = r;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// out[b][i] = a[b]*x[b][i] + c[b]*y[b][i]   on real views (complex arrays are passed as 2n reals).  grid (blocks, nb)
template <typename T>
__global__ __launch_bounds__(NTP) void k_lincomb(T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ y,
                                                BScal<T> a, BScal<T> c, long n, int b0) {
  const int b = blockIdx.y;
  const T av = a.v[b], cv = c.v[b];
  const long off = (long)(b0 + b) * n;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < n; i += (long)gridDim.x * NTP) {
    T r = av * x[off + i];
    if (y) r += cv * y[off + i];
    out[off + i] = r;
  }
}

// out[b][p][i] = m[i] * in[b][p][i]  (pixel mask, src/dataset.jl:281)   grid (blocks, slices)
template <typename T>
__global__ __launch_bounds__(NTP) void k_mask_mul(T* __restrict__ out, const T* __restrict__ in, const T* __restrict__ m, long n) {
  const long off = (long)blockIdx.y * n;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < n; i += (long)gridDim.x * NTP) out[off + i] = m[i] * in[off + i];
}

// ---------------------------------------------------------------------------------------------
// Reductions: per-batch-slot sums, deterministic two-pass (fixed partition, fixed tree).
//
// `sum_dropdims` (src/util.jl:297-316) sums the already-broadcast array -- every term is formed in the working precision T --
// in one of three ways selected by `set_sum_accuracy_mode!` (src/util.jl:288-292):
//   SUM_WORKING : plain `sum` in T (the reference's default)          -> per-thread running sum, tree and final sum in T
//   SUM_FLOAT64 : `sum(T64.(A))`                                      -> accumulation in double  (the engine's default)
//   SUM_KAHAN   : `sum_kbn`, compensated summation in T               -> per-thread Kahan-Babuska-Neumaier in T, the (sum,
//                 compensation) pairs combined in double and the result rounded to T: within one ulp(T) of the exact sum
// The summation ORDER of the reference (Julia's pairwise sum on the CPU, a tree on the GPU) is not reproducible in any mode.
enum SumMode { SUM_WORKING = 0, SUM_FLOAT64 = 1, SUM_KAHAN = 2 };

template <typename T, int MODE> struct SumAcc;
template <typename T> struct SumAcc<T, SUM_WORKING> {
  using P = T; T s = 0;
  __device__ __forceinline__ void add(T x) { s += x; }
  __device__ __forceinline__ P partial() const { return s; }
};
template <typename T> struct SumAcc<T, SUM_FLOAT64> {
  using P = double; double s = 0;
  __device__ __forceinline__ void add(T x) { s += (double)x; }
  __device__ __forceinline__ P partial() const { return s; }
};
template <typename T> struct SumAcc<T, SUM_KAHAN> {
  using P = double; T s = 0, c = 0;
  __device__ __forceinline__ void add(T x) {
    const T t = s + x;
    c += (fabs(s) >= fabs(x)) ? ((s - t) + x) : ((x - t) + s);
    s = t;
  }
  __device__ __forceinline__ P partial() const { return (double)s + (double)c; }
};

template <typename A>
__device__ __forceinline__ A block_sum(A v) {
  __shared__ A red[NTP / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  A r = 0;
  if (threadIdx.x == 0) { for (int w = 0; w < NTP / 64; ++w) r += red[w]; }
  __syncthreads();
  return r;
}

// Element functors: value of term i of batch slot bt, formed in T exactly like the reference's broadcast.
//   Fourier dot  z = real(conj(a) b) .* lam   (src/proj_lambert.jl:322-325; the 1/(Ny Nx) is applied to the sum)
template <typename T> struct TermDotF {
  const cx<T>* a; const cx<T>* b; const T* lam; long n; int Nx, Nyh;
  __device__ __forceinline__ T operator()(int bt, long i) const {
    const cx<T> u = a[(long)bt * n + i], v = b[(long)bt * n + i];
    return (u.x * v.x + u.y * v.y) * lam[(int)(((unsigned)i / (unsigned)Nx) % (unsigned)Nyh)];
  }
};
//   Map dot  z = a .* b   (src/proj_lambert.jl:318-321)
template <typename T> struct TermDotMap {
  const T* a; const T* b; long n;
  __device__ __forceinline__ T operator()(int bt, long i) const { return a[(long)bt * n + i] * b[(long)bt * n + i]; }
};
//   logdet of a Fourier diagonal, real planes or complex field: nan2zero(log|d| * lam)   (src/proj_lambert.jl:331-336)
template <typename T> struct TermLogdetF {
  const T* d; const T* lam; long n; int Nx, Nyh;
  __device__ __forceinline__ T operator()(int bt, long i) const {
    const T v = log(fabs(d[(long)bt * n + i])) * lam[(int)(((unsigned)i / (unsigned)Nx) % (unsigned)Nyh)];
    return isfinite(v) ? v : T(0);
  }
};
template <typename T> struct TermLogdetFc {
  const cx<T>* d; const T* lam; long n; int Nx, Nyh;
  __device__ __forceinline__ T operator()(int bt, long i) const {
    const cx<T> z = d[(long)bt * n + i];
    const T v = log(hypot(z.x, z.y)) * lam[(int)(((unsigned)i / (unsigned)Nx) % (unsigned)Nyh)];
    return isfinite(v) ? v : T(0);
  }
};
//   tr of a Fourier diagonal: real(d .* lam)   (src/proj_lambert.jl:346-350)
template <typename T> struct TermTrFc {
  const cx<T>* d; const T* lam; long n; int Nx, Nyh;
  __device__ __forceinline__ T operator()(int bt, long i) const { return d[(long)bt * n + i].x * lam[(int)(((unsigned)i / (unsigned)Nx) % (unsigned)Nyh)]; }
};
//   tr / logdet of a Map diagonal (src/proj_lambert.jl:337-342,351-353): d, and log|d| (the sign term is k_sign_map's)
template <typename T> struct TermTrMap {
  const T* d; long n;
  __device__ __forceinline__ T operator()(int bt, long i) const { return d[(long)bt * n + i]; }
};
template <typename T> struct TermLogAbsMap {
  const T* d; long n;
  __device__ __forceinline__ T operator()(int bt, long i) const { return log(fabs(d[(long)bt * n + i])); }
};

// pass 1: grid (nblk, B).  part[bt][blk] holds the block's partial sum (a T value in SUM_WORKING)
template <typename T, int MODE, typename F>
__global__ __launch_bounds__(NTP) void k_reduce_terms(F f, double* __restrict__ part, long n) {
  const int bt = blockIdx.y;
  SumAcc<T, MODE> acc;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < n; i += (long)gridDim.x * NTP) acc.add(f(bt, i));
  const auto r = block_sum<typename SumAcc<T, MODE>::P>(acc.partial());
  if (threadIdx.x == 0) part[(long)bt * gridDim.x + blockIdx.x] = (double)r;
}
// pass 2: grid (B).  out[bt] = scale * sum of the partials; in SUM_WORKING the sum and the scaling are done in T
template <typename T, int MODE>
__global__ __launch_bounds__(NTP) void k_reduce_final(const double* __restrict__ part, double* __restrict__ out, int nblk, double scale) {
  const int bt = blockIdx.x;
  using P = typename SumAcc<T, MODE>::P;
  P acc = 0;
  for (int i = threadIdx.x; i < nblk; i += NTP) acc += (P)part[(long)bt * nblk + i];
  const P r = block_sum<P>(acc);
  if (threadIdx.x == 0) {
    if (MODE == SUM_WORKING) out[bt] = (double)((T)r * (T)scale);
    else if (MODE == SUM_KAHAN) out[bt] = (double)((T)(r * scale));
    else out[bt] = (double)r * scale;
  }
}
// sign bookkeeping of logdet(Diagonal(::Map)): log(prod(sign.(d))) -- 0 for an even number of negative entries, NaN for an odd
// number (Julia's log(-1.0) throws; here the slot's result is NaN), -Inf when an entry is zero.  part2[bt][blk] = (#neg, #zero)
template <typename T>
__global__ __launch_bounds__(NTP) void k_sign_map(const T* __restrict__ d, unsigned long long* __restrict__ part2, long n) {
  const int bt = blockIdx.y;
  unsigned long long neg = 0, zero = 0;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < n; i += (long)gridDim.x * NTP) {
    const T v = d[(long)bt * n + i];
    neg += v < T(0); zero += v == T(0);
  }
  neg = block_sum<unsigned long long>(neg); zero = block_sum<unsigned long long>(zero);
  if (threadIdx.x == 0) { part2[2 * ((long)bt * gridDim.x + blockIdx.x)] = neg; part2[2 * ((long)bt * gridDim.x + blockIdx.x) + 1] = zero; }
}
static __global__ __launch_bounds__(NTP) void k_sign_final(const unsigned long long* __restrict__ part2, double* __restrict__ inout, int nblk) {
  const int bt = blockIdx.x;
  unsigned long long neg = 0, zero = 0;
  for (int i = threadIdx.x; i < nblk; i += NTP) { neg += part2[2 * ((long)bt * nblk + i)]; zero += part2[2 * ((long)bt * nblk + i) + 1]; }
  neg = block_sum<unsigned long long>(neg); zero = block_sum<unsigned long long>(zero);
  if (threadIdx.x == 0) {
    if (zero) inout[bt] = -INFINITY;                           // log|0| already made the sum -Inf; log(prod sign) = log(0) = -Inf
    else if (neg & 1ull) inout[bt] = NAN;                      // log(-1)
  }
}

// ---------------------------------------------------------------------------------------------
// conjugate_gradient (src/numerical_algorithms.jl:73-134) with its scalars resident on the device: the host enqueues iterations
// without waiting for any reduction.  `done` latches when every batch slot has res < tol; from then on the state-updating kernels
// below are no-ops, so an iteration the host enqueued before it saw the flag changes nothing (same iterates, same history as
// a loop that tests the residual on the host after every iteration).
struct CgState {
  double *res, *pAp, *res2, *alpha, *beta, *best, *hist;       // [B] each; hist [maxit][B]
  int *done, *nan, *nh, *better;
};
// after res = dot(r, z) of the start vector
static __global__ void k_cg_start(CgState s, int B) {
  if (threadIdx.x != 0) return;
  int nanf = 0;
  for (int b = 0; b < B; ++b) { s.best[b] = s.res[b]; s.hist[b] = s.res[b]; nanf |= isnan(s.res[b]); }
  *s.nh = 1; *s.done = 0; *s.better = 1; *s.nan = nanf;
}
// alpha = res / pAp  (:102)
static __global__ void k_cg_alpha(CgState s, int B) {
  if (threadIdx.x != 0 || *s.done) return;
  for (int b = 0; b < B; ++b) s.alpha[b] = s.res[b] / s.pAp[b];
}
// x += alpha p ; r -= alpha Ap  (:103-104), complex arrays viewed as reals; grid (blocks, B)
template <typename T>
__global__ __launch_bounds__(NTP) void k_cg_xr(T* __restrict__ x, T* __restrict__ r, const T* __restrict__ p, const T* __restrict__ Ap,
                                              CgState s, long n) {
  if (*s.done) return;
  const int b = blockIdx.y;
  const T al = (T)s.alpha[b];
  const long off = (long)b * n;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < n; i += (long)gridDim.x * NTP) {
    const T pv = p[off + i], av = Ap[off + i];
    x[off + i] += al * pv;
    r[off + i] -= al * av;
  }
}
// beta = res' / res ; res = res' ; best-iterate bookkeeping, history, stop test  (:106-125)
static __global__ void k_cg_beta(CgState s, int B, double tol) {
  if (threadIdx.x != 0) return;
  if (*s.done) { *s.better = 0; return; }
  int better = 1, done = 1, nanf = 0;
  for (int b = 0; b < B; ++b) {
    const double r2 = s.res2[b];
    nanf |= isnan(r2);
    s.beta[b] = r2 / s.res[b];
    s.res[b] = r2;
    better &= (r2 < s.best[b]); done &= (r2 < tol);
    s.hist[(long)(*s.nh) * B + b] = r2;
  }
  if (better) for (int b = 0; b < B; ++b) s.best[b] = s.res[b];
  *s.nh += 1; *s.better = better; *s.nan |= nanf;
  *s.done = done || nanf;
}
// p = z + beta p  (:107)
template <typename T>
__global__ __launch_bounds__(NTP) void k_cg_p(T* __restrict__ p, const T* __restrict__ z, CgState s, long n) {
  const int b = blockIdx.y;
  const T be = (T)s.beta[b];
  const long off = (long)b * n;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < n; i += (long)gridDim.x * NTP) p[off + i] = z[off + i] + be * p[off + i];
}
// bestx = x when this iteration improved on the best residual (:111-114)
template <typename T>
__global__ __launch_bounds__(NTP) void k_cg_keep_best(T* __restrict__ bx, const T* __restrict__ x, CgState s, long ntot) {
  if (!*s.better) return;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < ntot; i += (long)gridDim.x * NTP) bx[i] = x[i];
}

// ---------------------------------------------------------------------------------------------
// quadratic_estimate building blocks (src/quadratic_estimate.jl:83-91): one "leg"
//   out = nan2zero( in * (i lx)^p1 * (i ly)^p2 / |l|^n )      in F layout (S0, batched); caller inverse-transforms it.
template <typename T>
__global__ __launch_bounds__(NTP) void k_qe_leg(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, const T* __restrict__ lx_r,
                                                const T* __restrict__ ly, int Nx, long plane, int B, int n, int p1, int p2, int take_abs) {
  const long i = (long)blockIdx.x * NTP + threadIdx.x;
  if (i >= plane) return;
  const T lx = lx_r[(unsigned)i % (unsigned)Nx], l_y = ly[(unsigned)i / (unsigned)Nx];
  T mag = T(1);
  for (int k = 0; k < p1; ++k) mag *= lx;
  for (int k = 0; k < p2; ++k) mag *= l_y;
  if (n > 0) { const T lm = sqrt(lx * lx + l_y * l_y); for (int k = 0; k < n; ++k) mag /= lm; }
  const int rot = (p1 + p2) & 3;                          // i^(p1+p2)
  for (int b = 0; b < B; ++b) {
    const cx<T> v = in[(long)b * plane + i];
    cx<T> r = mk<T>(mag * v.x, mag * v.y);
    if (rot == 1) r = mul_i(r); else if (rot == 2) r = mk<T>(-r.x, -r.y); else if (rot == 3) r = mul_mi(r);
    r = mk<T>(nan2zero(r.x), nan2zero(r.y));
    if (take_abs) r = mk<T>(sqrt(r.x * r.x + r.y * r.y), T(0));
    out[(long)b * plane + i] = r;
  }
}

// quadratic_estimate weight planes (src/quadratic_estimate.jl:52-62,100-110), all of one component in one pass, evaluated in double like
// the reference's host algebra and rounded once:  S = TF^2 Cf~ + Cn;  W0 = TF^2 / S, W1 = W0 Cf, W2 = W0 Cf^2 as S0 Fourier planes (w, 0);
// fil = TF / S and filC = fil Cf as (w, w) pairs (a real plane times a complex field = an elementwise product of the interleaved reals).
__device__ __forceinline__ double finite0(double v) { return isfinite(v) ? v : 0.0; }
template <typename T>
__global__ __launch_bounds__(NTP) void k_qe_weights(const double* __restrict__ Cf, const double* __restrict__ Cft, const double* __restrict__ Cn,
                                                    const double* __restrict__ TF, cx<T>* __restrict__ W0, cx<T>* __restrict__ W1, cx<T>* __restrict__ W2,
                                                    cx<T>* __restrict__ fil, cx<T>* __restrict__ filC, long plane) {
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < plane; i += (long)gridDim.x * NTP) {
    const double tf = TF[i], S = tf * tf * Cft[i] + Cn[i], iS = finite0(1.0 / S), C = Cf[i];
    W0[i] = mk<T>((T)(tf * tf * iS), T(0)); W1[i] = mk<T>((T)(tf * tf * C * iS), T(0)); W2[i] = mk<T>((T)(tf * tf * C * C * iS), T(0));
    const T f = (T)finite0(tf / S), fc = (T)finite0(tf / S * C);
    fil[i] = mk<T>(f, f); filC[i] = mk<T>(fc, fc);
  }
}
// normalisation and the Wiener weight (src/quadratic_estimate.jl:44-46,177-187):  AL = pinv(tot) (or the caller's plane);
// wf = (wiener ? Cphi / (Cphi + AL) : 1) AL as a (w, w) pair
template <typename T>
__global__ __launch_bounds__(NTP) void k_qe_norm(const cx<T>* __restrict__ tot, const double* __restrict__ AL_in, const double* __restrict__ Cphi, int wiener,
                                                 double* __restrict__ AL, cx<T>* __restrict__ wf, long plane) {
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < plane; i += (long)gridDim.x * NTP) {
    const double al = AL_in ? AL_in[i] : finite0(1.0 / (double)tot[i].x);
    AL[i] = al;
    const T w = (T)(wiener ? finite0(Cphi[i] / (Cphi[i] + al)) * al : al);
    wf[i] = mk<T>(w, w);
  }
}
// out = (accumulate ? out : 0) + scale * a * b   (maps)
template <typename T>
__global__ __launch_bounds__(NTP) void k_map_fma(T* __restrict__ out, const T* __restrict__ a, const T* __restrict__ b, T scale, int accumulate, long n) {
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < n; i += (long)gridDim.x * NTP) {
    const T v = scale * a[i] * b[i];
    out[i] = accumulate ? out[i] + v : v;
  }
}

// White noise for `simulate` / `randn!` (src/specialops.jl:6,93; src/base_fields.jl:169-170): counter-based Philox4x32-10
// (Salmon et al. 2011), key = seed, counter = (c, stream): counter c gives elements 4c..4c+3 of the slot, so the draw does not
// depend on the launch geometry or on how chains are spread over GPUs.  Box-Muller in fp64 for both dtypes:
//   u1 = (w0 + 0.5)/2^32, u2 = (w1 + 0.5)/2^32,  r = sqrt(-2 ln u1):  (r cos 2 pi u2, r sin 2 pi u2), same for (w2, w3).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
template <typename T>
__global__ __launch_bounds__(NTP) void k_randn(T* __restrict__ out, long n, uint64_t seed, uint64_t stream) {
  const long nc = (n + 3) >> 2;
  for (long c = (long)blockIdx.x * NTP + threadIdx.x; c < nc; c += (long)gridDim.x * NTP) {
    uint32_t w[4];
    philox4x32_10((uint32_t)c, (uint32_t)((uint64_t)c >> 32), (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), w);
    double z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const double u1 = ((double)w[2 * h] + 0.5) * 0x1p-32, u2 = ((double)w[2 * h + 1] + 0.5) * 0x1p-32;
      const double r = sqrt(-2.0 * log(u1));
      double sn, cs;
      sincospi(2.0 * u2, &sn, &cs);
      z[2 * h] = r * cs; z[2 * h + 1] = r * sn;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (4 * c + j < n) out[4 * c + j] = (T)z[j];
  }
}

// get_max_lensing_step (src/lenseflow.jl:242-256): per pixel the two roots alpha of det(I + H(phi) + alpha H(eta)) = 0,
// minimum over the positive ones.  hp/he: [5][B][npix] maps from gradhess (gx, gy, Hxx, Hyx, Hyy); part: per-block minima.
template <typename T>
__global__ __launch_bounds__(NTP) void k_max_step(const T* __restrict__ hp, const T* __restrict__ he, double* __restrict__ part, long npix, long comp_stride) {
  const int bt = blockIdx.y;
  double best = 1e300;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < npix; i += (long)gridDim.x * NTP) {
    const long o = (long)bt * npix + i;
    const double p11 = hp[2 * comp_stride + o], p12 = hp[3 * comp_stride + o], p22 = hp[4 * comp_stride + o];
    const double e11 = he[2 * comp_stride + o], e12 = he[3 * comp_stride + o], e22 = he[4 * comp_stride + o];
    const double a = e11 * e22 - e12 * e12;
    const double b = e11 * (1 + p22) + e22 * (1 + p11) - 2 * e12 * p12;
    const double c = (1 + p11) * (1 + p22) - p12 * p12;
    const double sq = sqrt(b * b - 4 * a * c);
    const double a1 = (-b + sq) / (2 * a), a2 = (-b - sq) / (2 * a);
    if (a1 > 0 && a1 < best) best = a1;
    if (a2 > 0 && a2 < best) best = a2;
  }
  __shared__ double red[NTP / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = fmin(best, __shfl_down(best, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = red[0];
    for (int w = 1; w < NTP / 64; ++w) r = fmin(r, red[w]);
    part[(long)bt * gridDim.x + blockIdx.x] = r;
  }
}
static __global__ __launch_bounds__(NTP) void k_min_final(const double* __restrict__ part, double* __restrict__ out, int nblk) {
  __shared__ double red[NTP / 64];
  const int bt = blockIdx.x;
  double best = 1e300;
  for (int i = threadIdx.x; i < nblk; i += NTP) best = fmin(best, part[(long)bt * nblk + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = fmin(best, __shfl_down(best, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) { double r = red[0]; for (int w = 1; w < NTP / 64; ++w) r = fmin(r, red[w]); out[bt] = r; }
}

}  // namespace cmbl
