// Pointwise kernels on the internal layouts: harmonic-basis operators (Cl / mask / beam / TE block applies
// fused with the QU<->EB rotation), linear combinations with per-batch scalars, per-batch reductions.
#pragma once
#include "common.hpp"

namespace cmbl {

constexpr int MAXB = 16;                       // per-batch scalars are passed by value in chunks of MAXB
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
// Reductions, deterministic two-pass (fixed partition, fixed tree), accumulated in double.
template <int DUMMY = 0>
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double red[NTP / 64];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0) { for (int w = 0; w < NTP / 64; ++w) r += red[w]; }
  __syncthreads();
  return r;
}

// Fourier dot in F layout: sum lam[ky] * Re(conj(a) b)   (src/proj_lambert.jl:322-325); n = P*Nyh*Nx per batch
template <typename T>
__global__ __launch_bounds__(NTP) void k_dot_F(const cx<T>* __restrict__ a, const cx<T>* __restrict__ b,
                                              const T* __restrict__ lam, double* __restrict__ part,
                                              long n, int lgNx, int Nyh) {
  const int bt = blockIdx.y;
  const long off = (long)bt * n;
  double acc = 0;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < n; i += (long)gridDim.x * NTP) {
    const int ky = (int)((i >> lgNx) % Nyh);
    cx<T> u = a[off + i], v = b[off + i];
    acc += (double)lam[ky] * ((double)u.x * (double)v.x + (double)u.y * (double)v.y);
  }
  double r = block_sum(acc);
  if (threadIdx.x == 0) part[(long)bt * gridDim.x + blockIdx.x] = r;
}

// Map dot: sum a*b  (src/proj_lambert.jl:318-321)
template <typename T>
__global__ __launch_bounds__(NTP) void k_dot_map(const T* __restrict__ a, const T* __restrict__ b, double* __restrict__ part, long n) {
  const int bt = blockIdx.y;
  const long off = (long)bt * n;
  double acc = 0;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < n; i += (long)gridDim.x * NTP)
    acc += (double)a[off + i] * (double)b[off + i];
  double r = block_sum(acc);
  if (threadIdx.x == 0) part[(long)bt * gridDim.x + blockIdx.x] = r;
}

// logdet of a real diagonal in F layout: sum lam * log|d|, non-finite -> 0  (src/proj_lambert.jl:331-336)
template <typename T>
__global__ __launch_bounds__(NTP) void k_logdet_F(const T* __restrict__ d, const T* __restrict__ lam, double* __restrict__ part,
                                                 long n, int lgNx, int Nyh) {
  const int bt = blockIdx.y;
  const long off = (long)bt * n;
  double acc = 0;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < n; i += (long)gridDim.x * NTP) {
    const int ky = (int)((i >> lgNx) % Nyh);
    double v = log(fabs((double)d[off + i])) * (double)lam[ky];
    acc += isfinite(v) ? v : 0.0;
  }
  double r = block_sum(acc);
  if (threadIdx.x == 0) part[(long)bt * gridDim.x + blockIdx.x] = r;
}

__global__ __launch_bounds__(NTP) void k_reduce_final(const double* __restrict__ part, double* __restrict__ out, int nblk, double scale) {
  const int bt = blockIdx.x;
  double acc = 0;
  for (int i = threadIdx.x; i < nblk; i += NTP) acc += part[(long)bt * nblk + i];
  double r = block_sum(acc);
  if (threadIdx.x == 0) out[bt] = r * scale;
}

// ---------------------------------------------------------------------------------------------
// quadratic_estimate building blocks (src/quadratic_estimate.jl:83-91): one "leg"
//   out = nan2zero( in * (i lx)^p1 * (i ly)^p2 / |l|^n )      in F layout (S0, batched); caller inverse-transforms it.
template <typename T>
__global__ __launch_bounds__(NTP) void k_qe_leg(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, const T* __restrict__ lx_r,
                                                const T* __restrict__ ly, int lgNx, long plane, int B, int n, int p1, int p2, int take_abs) {
  const long i = (long)blockIdx.x * NTP + threadIdx.x;
  if (i >= plane) return;
  const T lx = lx_r[i & ((1 << lgNx) - 1)], l_y = ly[i >> lgNx];
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
__global__ __launch_bounds__(NTP) void k_min_final(const double* __restrict__ part, double* __restrict__ out, int nblk) {
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
