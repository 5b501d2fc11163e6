// Fused LenseFlow RK-stage kernels.
//
// Reference algorithm: src/lenseflow.jl:150-214 (velocity, velocityᴴ, negδvelocityᴴ) driven by
// src/numerical_algorithms.jl:11-24 (RK4).  The reference evaluates every stage as
//   rfft2 -> (i lx, i ly) multiply -> 2 x irfft2 -> p . grad f      (and mirror images for the adjoint)
// with one array pass per arrow.  Here each stage is ONE row kernel + ONE column kernel:
//   * p(t) = M^-1(t)' grad(phi), M(t) = I + t H(phi) is recomputed in registers from the five
//     time-independent maps (gx, gy, Hxx, Hyx, Hyy) instead of caching 6(2n+1) maps
//     (src/lenseflow.jl:131-142 caches them; identical arithmetic incl. pinv and quirk Q2);
//   * the i*ly multiply, the c2r/r2c y-transforms, the velocity product and the RK4 axpy's are fused
//     into the column kernel, which also emits the y-transform of the NEXT stage's input;
//   * the i*lx multiply sits between a forward and an inverse x-FFT inside the row kernel.
// Non-Hermitian Nyquist content (the reference does NOT zero i*l at Nyquist, src/proj_lambert.jl:63-64)
// is carried exactly as FFTW/cuFFT carry it: the c2r drops Im of the ky=0 / ky=Ny/2 rows after the x pass.
#pragma once
#include "kernels_fft.hpp"

namespace cmbl {

template <typename T> __device__ __forceinline__ T pinv_s(T v) { T r = T(1) / v; return isfinite(r) ? r : T(0); }

// p(t) and M^-1(t) at one pixel  (src/lenseflow.jl:138-139, src/field_vectors.jl:86-94,46-47)
template <typename T>
__device__ __forceinline__ void flow_pm(T t, T gx, T gy, T hxx, T hyx, T hyy, T& px, T& py, T& m11, T& m12, T& m22) {
  const T a = T(1) + t * hxx, c = t * hyx, d = T(1) + t * hyy;   // b := c  (pinv! reads A[2,1] twice)
  const T idet = pinv_s(a * d - c * c);
  m11 = idet * d; m12 = -idet * c; m22 = idet * a;               // m21 == m12
  px = m11 * gx + m12 * gy;
  py = m12 * gx + m22 * gy;
}

template <typename T> struct PhiMaps { const T *gx, *gy, *hxx, *hyx, *hyy; int Bphi; };

template <typename T> struct RKCoef { T t, cnext, h6; int stage, last; };   // stage 1..4

// RK4 bookkeeping on one value (src/numerical_algorithms.jl:15-21):
//   stage 1: acc = k ; 2,3: acc += 2k ; 4: y0 += h/6 (acc + k).  Returns the next stage input.
template <typename V, typename T>
__device__ __forceinline__ V rk_update(const RKCoef<T>& rk, V k, V& y0, V& acc) {
  if (rk.stage == 1) { acc = k; return y0 + rk.cnext * k; }
  if (rk.stage < 4) { acc = acc + T(2) * k; return y0 + rk.cnext * k; }
  y0 = y0 + rk.h6 * (acc + k);
  return y0;
}

// ---------------------------------------------------------------------------------------------
// Forward / inverse flow, column kernel.  grid (Nx/C, P*B).  C*M == R*NT.
//   in : A  = rfft_y(f_s)  (mixed), Gx = d/dx f_s y-transformed (mixed; from k_x_fft<MODE 2>)
//   out: y0/acc updated, Anext = rfft_y(f_{s+1}) (mixed)
template <typename T> struct FlowYArgs {
  const cx<T>* A; const cx<T>* Gx; cx<T>* Anext;
  T* y0; T* acc;
  PhiMaps<T> ph;
  const cx<T>* twY; const T* ly;
  int Nx, lgM, C, lgC, P;
  RKCoef<T> rk;
};

template <typename T, int R>
__global__ __launch_bounds__(NT) void k_flow_y_fwd(FlowYArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int M = 1 << a.lgM, LD = M + 1, Nyh = M + 1, Nx = a.Nx, C = a.C;
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;
  const int x0 = blockIdx.x * C;
  const size_t sl = blockIdx.y;
  const int bphi = a.ph.Bphi == 1 ? 0 : (int)(sl / a.P);
  const T invNy = T(1) / T(2 * M);
  load_twiddles(tw, a.twY, M);

  // d/dx f
  tile_load_mixed(s, a.Gx + sl * (size_t)Nyh * Nx, Nx, x0, C, a.lgC, a.lgM, [](cx<T> v, int) { return v; });
  __syncthreads();
  c2r_pre(s, C, LD, a.lgM, tw);
  fft_dit(s, C, LD, a.lgM, tw, a.lgM + 1);
  cx<T> k[R], pyr[R];
  const size_t pbase = ((size_t)bphi * Nx + x0) * M;            // in units of pairs
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int e = threadIdx.x + i * NT, c = e >> a.lgM, j = e & (M - 1);
    const size_t gi = pbase + (size_t)c * M + j;
    const cx<T> gx = reinterpret_cast<const cx<T>*>(a.ph.gx)[gi], gy = reinterpret_cast<const cx<T>*>(a.ph.gy)[gi];
    const cx<T> hxx = reinterpret_cast<const cx<T>*>(a.ph.hxx)[gi], hyx = reinterpret_cast<const cx<T>*>(a.ph.hyx)[gi];
    const cx<T> hyy = reinterpret_cast<const cx<T>*>(a.ph.hyy)[gi];
    T px0, py0, px1, py1, m11, m12, m22;
    flow_pm(a.rk.t, gx.x, gy.x, hxx.x, hyx.x, hyy.x, px0, py0, m11, m12, m22);
    flow_pm(a.rk.t, gx.y, gy.y, hxx.y, hyx.y, hyy.y, px1, py1, m11, m12, m22);
    const cx<T> d = s[c * LD + j];
    k[i] = mk<T>(px0 * (d.x * invNy), px1 * (d.y * invNy));
    pyr[i] = mk<T>(py0, py1);
  }
  __syncthreads();

  // d/dy f : i*ly applied on load
  const T* ly = a.ly;
  tile_load_mixed(s, a.A + sl * (size_t)Nyh * Nx, Nx, x0, C, a.lgC, a.lgM,
                  [ly](cx<T> v, int kk) { const T l = ly[kk]; return mk<T>(-l * v.y, l * v.x); });
  __syncthreads();
  c2r_pre(s, C, LD, a.lgM, tw);
  fft_dit(s, C, LD, a.lgM, tw, a.lgM + 1);
  cx<T>* y0p = reinterpret_cast<cx<T>*>(a.y0) + (sl * Nx + x0) * (size_t)M;
  cx<T>* accp = reinterpret_cast<cx<T>*>(a.acc) + (sl * Nx + x0) * (size_t)M;
  cx<T> fn[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int e = threadIdx.x + i * NT, c = e >> a.lgM, j = e & (M - 1);
    const cx<T> d = s[c * LD + j];
    const cx<T> kv = mk<T>(k[i].x + pyr[i].x * (d.x * invNy), k[i].y + pyr[i].y * (d.y * invNy));
    const size_t gi = (size_t)c * M + j;
    cx<T> y0 = y0p[gi];
    cx<T> acc = a.rk.stage == 1 ? mk<T>(0, 0) : accp[gi];
    fn[i] = rk_update(a.rk, kv, y0, acc);
    if (a.rk.stage == 4) y0p[gi] = y0; else accp[gi] = acc;
  }
  if (a.rk.last) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int e = threadIdx.x + i * NT, c = e >> a.lgM, j = e & (M - 1);
    s[c * LD + j] = fn[i];
  }
  __syncthreads();
  fft_dif(s, C, LD, a.lgM, tw, a.lgM + 1);
  r2c_post(s, C, LD, a.lgM, tw);
  tile_store_mixed(s, a.Anext + sl * (size_t)Nyh * Nx, Nx, x0, C, a.lgC, a.lgM, [](cx<T> v, int) { return v; });
}

// ---------------------------------------------------------------------------------------------
// Adjoint flow, column kernel: H = ifft_x(Y) (mixed)  ->  Wx = rfft_y(px*y), Wy' = i*ly*rfft_y(py*y) (mixed)
//   (src/lenseflow.jl:163-174).  Optionally (delta flow) multiplies y by grad f and writes w-partials.
template <typename T> struct AdjYArgs {
  const cx<T>* H; cx<T>* Wx; cx<T>* Wy;
  PhiMaps<T> ph;
  const cx<T>* twY; const T* ly;
  int Nx, lgM, C, lgC, P;
  T t;
};

template <typename T, int R>
__global__ __launch_bounds__(NT) void k_adj_y(AdjYArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int M = 1 << a.lgM, LD = M + 1, Nyh = M + 1, Nx = a.Nx, C = a.C;
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;
  const int x0 = blockIdx.x * C;
  const size_t sl = blockIdx.y;
  const int bphi = a.ph.Bphi == 1 ? 0 : (int)(sl / a.P);
  const T invNy = T(1) / T(2 * M);
  load_twiddles(tw, a.twY, M);
  tile_load_mixed(s, a.H + sl * (size_t)Nyh * Nx, Nx, x0, C, a.lgC, a.lgM, [](cx<T> v, int) { return v; });
  __syncthreads();
  c2r_pre(s, C, LD, a.lgM, tw);
  fft_dit(s, C, LD, a.lgM, tw, a.lgM + 1);
  cx<T> wy[R];
  const size_t pbase = ((size_t)bphi * Nx + x0) * M;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int e = threadIdx.x + i * NT, c = e >> a.lgM, j = e & (M - 1);
    const size_t gi = pbase + (size_t)c * M + j;
    const cx<T> gx = reinterpret_cast<const cx<T>*>(a.ph.gx)[gi], gy = reinterpret_cast<const cx<T>*>(a.ph.gy)[gi];
    const cx<T> hxx = reinterpret_cast<const cx<T>*>(a.ph.hxx)[gi], hyx = reinterpret_cast<const cx<T>*>(a.ph.hyx)[gi];
    const cx<T> hyy = reinterpret_cast<const cx<T>*>(a.ph.hyy)[gi];
    T px0, py0, px1, py1, m11, m12, m22;
    flow_pm(a.t, gx.x, gy.x, hxx.x, hyx.x, hyy.x, px0, py0, m11, m12, m22);
    flow_pm(a.t, gx.y, gy.y, hxx.y, hyx.y, hyy.y, px1, py1, m11, m12, m22);
    const cx<T> d = s[c * LD + j];
    const cx<T> yv = mk<T>(d.x * invNy, d.y * invNy);
    wy[i] = mk<T>(py0 * yv.x, py1 * yv.y);
    s[c * LD + j] = mk<T>(px0 * yv.x, px1 * yv.y);            // own slot: no hazard
  }
  __syncthreads();
  fft_dif(s, C, LD, a.lgM, tw, a.lgM + 1);
  r2c_post(s, C, LD, a.lgM, tw);
  tile_store_mixed(s, a.Wx + sl * (size_t)Nyh * Nx, Nx, x0, C, a.lgC, a.lgM, [](cx<T> v, int) { return v; });
  __syncthreads();
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int e = threadIdx.x + i * NT, c = e >> a.lgM, j = e & (M - 1);
    s[c * LD + j] = wy[i];
  }
  __syncthreads();
  fft_dif(s, C, LD, a.lgM, tw, a.lgM + 1);
  r2c_post(s, C, LD, a.lgM, tw);
  const T* ly = a.ly;
  tile_store_mixed(s, a.Wy + sl * (size_t)Nyh * Nx, Nx, x0, C, a.lgC, a.lgM,
                   [ly](cx<T> v, int kk) { const T l = ly[kk]; return mk<T>(-l * v.y, l * v.x); });
}

// Adjoint flow, row kernel:  k = i*lx*fft_x(Wx) + fft_x(Wy')  -> RK update of the Fourier state (F layout)
//   -> Hnext = ifft_x(next stage input) (mixed).     rows = slices*Nyh, grid ceil(rows/RX).  LDS: twX + 2*RX*Nx
template <typename T> struct AdjXArgs {
  const cx<T>* Wx; const cx<T>* Wy; cx<T>* Y0; cx<T>* acc; cx<T>* Hnext;
  const cx<T>* twX; const T* lx_r;
  int lgNx, RX; long rows;
  RKCoef<T> rk;
};

template <typename T>
__global__ __launch_bounds__(NT) void k_adj_x(AdjXArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Nx = 1 << a.lgNx;
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + (Nx >> 1);
  const long r0 = (long)blockIdx.x * a.RX;
  const int nr = (int)min((long)a.RX, a.rows - r0);
  cx<T>* s2 = s + (size_t)a.RX * Nx;
  load_twiddles(tw, a.twX, Nx >> 1);
  const int n = nr * Nx;
  for (int e = threadIdx.x; e < n; e += NT) { s[e] = a.Wx[r0 * Nx + e]; s2[e] = a.Wy[r0 * Nx + e]; }
  __syncthreads();
  // both row sets in one go: they are adjacent in LDS when nr == RX; otherwise two calls
  if (nr == a.RX) fft_dif(s, 2 * nr, Nx, a.lgNx, tw, a.lgNx);
  else { fft_dif(s, nr, Nx, a.lgNx, tw, a.lgNx); fft_dif(s2, nr, Nx, a.lgNx, tw, a.lgNx); }
  const T inv = T(1) / T(Nx);
  for (int e = threadIdx.x; e < n; e += NT) {
    const T l = a.lx_r[e & (Nx - 1)];
    const cx<T> u = s[e], v = s2[e];
    const cx<T> kv = mk<T>(-l * u.y + v.x, l * u.x + v.y);
    const long gi = r0 * Nx + e;
    cx<T> y0 = a.Y0[gi];
    cx<T> acc = a.rk.stage == 1 ? mk<T>(0, 0) : a.acc[gi];
    cx<T> fn = rk_update(a.rk, kv, y0, acc);
    if (a.rk.stage == 4) a.Y0[gi] = y0; else a.acc[gi] = acc;
    s[e] = inv * fn;
  }
  if (a.rk.last) return;
  __syncthreads();
  fft_dit(s, nr, Nx, a.lgNx, tw, a.lgNx);
  for (int e = threadIdx.x; e < n; e += NT) a.Hnext[r0 * Nx + e] = s[e];
}

// ---------------------------------------------------------------------------------------------
// delta flow (src/lenseflow.jl:176-214), per-(pol,batch) column kernel: does the f part (== k_flow_y_fwd),
// the delta-f part (== k_adj_y) and writes the spin-adjoint partial products
//   w1p = L(df) * d/dx f,  w2p = L(df) * d/dy f     (maps, one pair per pol; summed over pol by k_dphi_y)
template <typename T> struct DeltaYArgs {
  FlowYArgs<T> f;           // f part
  const cx<T>* H; cx<T>* Wx; cx<T>* Wy;   // delta-f part
  T* w1p; T* w2p;           // (P*B, Nx, Ny)
};

template <typename T, int R>
__global__ __launch_bounds__(NT) void k_delta_y(DeltaYArgs<T> d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const FlowYArgs<T>& a = d.f;
  const int M = 1 << a.lgM, LD = M + 1, Nyh = M + 1, Nx = a.Nx, C = a.C;
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;
  const int x0 = blockIdx.x * C;
  const size_t sl = blockIdx.y;
  const int bphi = a.ph.Bphi == 1 ? 0 : (int)(sl / a.P);
  const T invNy = T(1) / T(2 * M);
  const size_t moff = sl * (size_t)Nyh * Nx;
  load_twiddles(tw, a.twY, M);

  // L(delta f) = irfft2(delta f)
  tile_load_mixed(s, d.H + moff, Nx, x0, C, a.lgC, a.lgM, [](cx<T> v, int) { return v; });
  __syncthreads();
  c2r_pre(s, C, LD, a.lgM, tw);
  fft_dit(s, C, LD, a.lgM, tw, a.lgM + 1);
  cx<T> ldf[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int e = threadIdx.x + i * NT, c = e >> a.lgM, j = e & (M - 1);
    const cx<T> v = s[c * LD + j];
    ldf[i] = mk<T>(v.x * invNy, v.y * invNy);
  }
  __syncthreads();
  // d/dx f
  tile_load_mixed(s, a.Gx + moff, Nx, x0, C, a.lgC, a.lgM, [](cx<T> v, int) { return v; });
  __syncthreads();
  c2r_pre(s, C, LD, a.lgM, tw);
  fft_dit(s, C, LD, a.lgM, tw, a.lgM + 1);
  cx<T> k[R];
  const size_t pbase = ((size_t)bphi * Nx + x0) * M;
  const size_t mbase = (sl * Nx + x0) * (size_t)M;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int e = threadIdx.x + i * NT, c = e >> a.lgM, j = e & (M - 1);
    const size_t gi = pbase + (size_t)c * M + j;
    const cx<T> gx = reinterpret_cast<const cx<T>*>(a.ph.gx)[gi], gy = reinterpret_cast<const cx<T>*>(a.ph.gy)[gi];
    const cx<T> hxx = reinterpret_cast<const cx<T>*>(a.ph.hxx)[gi], hyx = reinterpret_cast<const cx<T>*>(a.ph.hyx)[gi];
    const cx<T> hyy = reinterpret_cast<const cx<T>*>(a.ph.hyy)[gi];
    T px0, py0, px1, py1, m11, m12, m22;
    flow_pm(a.rk.t, gx.x, gy.x, hxx.x, hyx.x, hyy.x, px0, py0, m11, m12, m22);
    flow_pm(a.rk.t, gx.y, gy.y, hxx.y, hyx.y, hyy.y, px1, py1, m11, m12, m22);
    const cx<T> v = s[c * LD + j];
    const cx<T> dx = mk<T>(v.x * invNy, v.y * invNy);
    k[i] = mk<T>(px0 * dx.x, px1 * dx.y);
    reinterpret_cast<cx<T>*>(d.w1p)[mbase + (size_t)c * M + j] = mk<T>(ldf[i].x * dx.x, ldf[i].y * dx.y);
  }
  __syncthreads();
  // d/dy f
  const T* ly = a.ly;
  tile_load_mixed(s, a.A + moff, Nx, x0, C, a.lgC, a.lgM,
                  [ly](cx<T> v, int kk) { const T l = ly[kk]; return mk<T>(-l * v.y, l * v.x); });
  __syncthreads();
  c2r_pre(s, C, LD, a.lgM, tw);
  fft_dit(s, C, LD, a.lgM, tw, a.lgM + 1);
  cx<T>* y0p = reinterpret_cast<cx<T>*>(a.y0) + mbase;
  cx<T>* accp = reinterpret_cast<cx<T>*>(a.acc) + mbase;
  cx<T> fn[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int e = threadIdx.x + i * NT, c = e >> a.lgM, j = e & (M - 1);
    const size_t gi = pbase + (size_t)c * M + j;
    const cx<T> gx = reinterpret_cast<const cx<T>*>(a.ph.gx)[gi], gy = reinterpret_cast<const cx<T>*>(a.ph.gy)[gi];
    const cx<T> hxx = reinterpret_cast<const cx<T>*>(a.ph.hxx)[gi], hyx = reinterpret_cast<const cx<T>*>(a.ph.hyx)[gi];
    const cx<T> hyy = reinterpret_cast<const cx<T>*>(a.ph.hyy)[gi];
    T px0, py0, px1, py1, m11, m12, m22;
    flow_pm(a.rk.t, gx.x, gy.x, hxx.x, hyx.x, hyy.x, px0, py0, m11, m12, m22);
    flow_pm(a.rk.t, gx.y, gy.y, hxx.y, hyx.y, hyy.y, px1, py1, m11, m12, m22);
    const cx<T> v = s[c * LD + j];
    const cx<T> dy = mk<T>(v.x * invNy, v.y * invNy);
    const cx<T> kv = mk<T>(k[i].x + py0 * dy.x, k[i].y + py1 * dy.y);
    const size_t li = (size_t)c * M + j;
    reinterpret_cast<cx<T>*>(d.w2p)[mbase + li] = mk<T>(ldf[i].x * dy.x, ldf[i].y * dy.y);
    cx<T> y0 = y0p[li];
    cx<T> acc = a.rk.stage == 1 ? mk<T>(0, 0) : accp[li];
    fn[i] = rk_update(a.rk, kv, y0, acc);
    if (a.rk.stage == 4) y0p[li] = y0; else accp[li] = acc;
    // reuse k[] / ldf[] for the delta-f products  (px*Ldf -> k, py*Ldf -> ldf)
    k[i] = mk<T>(px0 * ldf[i].x, px1 * ldf[i].y);
    ldf[i] = mk<T>(py0 * ldf[i].x, py1 * ldf[i].y);
  }
  __syncthreads();
  // next-stage f : rfft_y
  if (!a.rk.last) {
#pragma unroll
    for (int i = 0; i < R; ++i) { const int e = threadIdx.x + i * NT, c = e >> a.lgM, j = e & (M - 1); s[c * LD + j] = fn[i]; }
    __syncthreads();
    fft_dif(s, C, LD, a.lgM, tw, a.lgM + 1);
    r2c_post(s, C, LD, a.lgM, tw);
    tile_store_mixed(s, a.Anext + moff, Nx, x0, C, a.lgC, a.lgM, [](cx<T> v, int) { return v; });
    __syncthreads();
  }
  // Wx = rfft_y(px*Ldf)
#pragma unroll
  for (int i = 0; i < R; ++i) { const int e = threadIdx.x + i * NT, c = e >> a.lgM, j = e & (M - 1); s[c * LD + j] = k[i]; }
  __syncthreads();
  fft_dif(s, C, LD, a.lgM, tw, a.lgM + 1);
  r2c_post(s, C, LD, a.lgM, tw);
  tile_store_mixed(s, d.Wx + moff, Nx, x0, C, a.lgC, a.lgM, [](cx<T> v, int) { return v; });
  __syncthreads();
  // Wy' = i*ly*rfft_y(py*Ldf)
#pragma unroll
  for (int i = 0; i < R; ++i) { const int e = threadIdx.x + i * NT, c = e >> a.lgM, j = e & (M - 1); s[c * LD + j] = ldf[i]; }
  __syncthreads();
  fft_dif(s, C, LD, a.lgM, tw, a.lgM + 1);
  r2c_post(s, C, LD, a.lgM, tw);
  tile_store_mixed(s, d.Wy + moff, Nx, x0, C, a.lgC, a.lgM,
                   [ly](cx<T> v, int kk) { const T l = ly[kk]; return mk<T>(-l * v.y, l * v.x); });
}

// delta-phi part, column kernel (one per batch slot): w = sum_pol partials; u = M^-1 w (quirk Q1 optional);
//   Z0 = i*ly*Y(u2) - ly^2*Y(t*py*u2) ; Z1 = Y(u1) + i*ly*Y(t*(py*u1 + px*u2)) ; Z2 = Y(t*px*u1)   (mixed, S0)
// so that d(dphi)/dt = fft_x(Z0) + i*lx*fft_x(Z1) - lx^2*fft_x(Z2)        (src/lenseflow.jl:198-206)
template <typename T> struct DphiYArgs {
  const T* w1p; const T* w2p;   // (B*P, Nx, Ny)
  cx<T>* Z0; cx<T>* Z1; cx<T>* Z2;   // (B, Nyh, Nx) mixed
  PhiMaps<T> ph;
  const cx<T>* twY; const T* ly;
  int Nx, lgM, C, lgC, P, alias_quirk;
  T t;
};

template <typename T, int R>
__global__ __launch_bounds__(NT) void k_dphi_y(DphiYArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int M = 1 << a.lgM, LD = M + 1, Nyh = M + 1, Nx = a.Nx, C = a.C;
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;
  const int x0 = blockIdx.x * C;
  const size_t b = blockIdx.y;
  const int bphi = a.ph.Bphi == 1 ? 0 : (int)b;
  load_twiddles(tw, a.twY, M);
  const size_t pbase = ((size_t)bphi * Nx + x0) * M;
  cx<T> u1[R], u2[R], pxr[R], pyr[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int e = threadIdx.x + i * NT, c = e >> a.lgM, j = e & (M - 1);
    const size_t gi = pbase + (size_t)c * M + j;
    const cx<T> gx = reinterpret_cast<const cx<T>*>(a.ph.gx)[gi], gy = reinterpret_cast<const cx<T>*>(a.ph.gy)[gi];
    const cx<T> hxx = reinterpret_cast<const cx<T>*>(a.ph.hxx)[gi], hyx = reinterpret_cast<const cx<T>*>(a.ph.hyx)[gi];
    const cx<T> hyy = reinterpret_cast<const cx<T>*>(a.ph.hyy)[gi];
    T px0, py0, px1, py1, a11, a12, a22, b11, b12, b22;
    flow_pm(a.t, gx.x, gy.x, hxx.x, hyx.x, hyy.x, px0, py0, a11, a12, a22);
    flow_pm(a.t, gx.y, gy.y, hxx.y, hyx.y, hyy.y, px1, py1, b11, b12, b22);
    cx<T> w1 = mk<T>(0, 0), w2 = mk<T>(0, 0);
    for (int p = 0; p < a.P; ++p) {                       // spin-adjoint product: sum over pol (src/proj_lambert.jl:423-430)
      const size_t mi = (((size_t)b * a.P + p) * Nx + x0) * M + (size_t)c * M + j;
      w1 = w1 + reinterpret_cast<const cx<T>*>(a.w1p)[mi];
      w2 = w2 + reinterpret_cast<const cx<T>*>(a.w2p)[mi];
    }
    // u = M^-1 w   (src/field_vectors.jl:48-49; with the reference's aliasing, v[2] sees the updated v[1])
    cx<T> v1 = mk<T>(a11 * w1.x + a12 * w2.x, b11 * w1.y + b12 * w2.y);
    cx<T> in1 = a.alias_quirk ? v1 : w1;
    cx<T> v2 = mk<T>(a12 * in1.x + a22 * w2.x, b12 * in1.y + b22 * w2.y);
    u1[i] = v1; u2[i] = v2; pxr[i] = mk<T>(px0, px1); pyr[i] = mk<T>(py0, py1);
  }
  const T* ly = a.ly; const T t = a.t;
  const size_t moff = b * (size_t)Nyh * Nx;
  constexpr int RZ = R + 1;                                // a thread stores at most R+1 half-spectrum entries (C*(M+1) = R*NT + C)
  cx<T> zr[RZ];
  auto fwd = [&](auto&& gen) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R; ++i) { const int e = threadIdx.x + i * NT, c = e >> a.lgM, j = e & (M - 1); s[c * LD + j] = gen(i); }
    __syncthreads();
    fft_dif(s, C, LD, a.lgM, tw, a.lgM + 1);
    r2c_post(s, C, LD, a.lgM, tw);
  };
  auto slot = [&](int i, int& kk) -> cx<T> {               // this thread's i-th half-spectrum entry
    const int e = threadIdx.x + i * NT; const int c = e & (C - 1); kk = e >> a.lgC;
    return (e < C * (M + 1)) ? s[c * LD + hslot(kk, M, a.lgM)] : mk<T>(0, 0);
  };
  auto put = [&](cx<T>* Z, int i, cx<T> v) {
    const int e = threadIdx.x + i * NT; const int c = e & (C - 1), kk = e >> a.lgC;
    if (e < C * (M + 1)) Z[moff + (size_t)kk * Nx + x0 + c] = v;
  };
  // Z2 = Y(t*px*u1)
  fwd([&](int i) { return mk<T>(t * pxr[i].x * u1[i].x, t * pxr[i].y * u1[i].y); });
#pragma unroll
  for (int i = 0; i < RZ; ++i) { int kk; cx<T> v = slot(i, kk); put(a.Z2, i, v); }
  // Z1 = Y(u1) + i*ly*Y(t*(py*u1 + px*u2))
  fwd([&](int i) { return u1[i]; });
#pragma unroll
  for (int i = 0; i < RZ; ++i) { int kk; zr[i] = slot(i, kk); }
  fwd([&](int i) { return mk<T>(t * (pyr[i].x * u1[i].x + pxr[i].x * u2[i].x), t * (pyr[i].y * u1[i].y + pxr[i].y * u2[i].y)); });
#pragma unroll
  for (int i = 0; i < RZ; ++i) { int kk; cx<T> v = slot(i, kk); const T l = (threadIdx.x + i * NT < C * (M + 1)) ? ly[kk] : T(0);
    put(a.Z1, i, mk<T>(zr[i].x - l * v.y, zr[i].y + l * v.x)); }
  // Z0 = i*ly*Y(u2) - ly^2*Y(t*py*u2)
  fwd([&](int i) { return u2[i]; });
#pragma unroll
  for (int i = 0; i < RZ; ++i) { int kk; cx<T> v = slot(i, kk); const T l = (threadIdx.x + i * NT < C * (M + 1)) ? ly[kk] : T(0);
    zr[i] = mk<T>(-l * v.y, l * v.x); }
  fwd([&](int i) { return mk<T>(t * pyr[i].x * u2[i].x, t * pyr[i].y * u2[i].y); });
#pragma unroll
  for (int i = 0; i < RZ; ++i) { int kk; cx<T> v = slot(i, kk); const T l = (threadIdx.x + i * NT < C * (M + 1)) ? ly[kk] : T(0);
    put(a.Z0, i, mk<T>(zr[i].x - l * l * v.x, zr[i].y - l * l * v.y)); }
}

// delta-phi part, row kernel: k = fft_x(Z0) + i*lx*fft_x(Z1) - lx^2*fft_x(Z2) ; RK update of the S0 Fourier state
template <typename T> struct DphiXArgs {
  const cx<T>* Z0; const cx<T>* Z1; const cx<T>* Z2; cx<T>* Y0; cx<T>* acc;
  const cx<T>* twX; const T* lx_r;
  int lgNx, RX; long rows;
  RKCoef<T> rk;
};

template <typename T>
__global__ __launch_bounds__(NT) void k_dphi_x(DphiXArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Nx = 1 << a.lgNx;
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + (Nx >> 1);
  const long r0 = (long)blockIdx.x * a.RX;
  const int nr = (int)min((long)a.RX, a.rows - r0);
  const size_t st = (size_t)a.RX * Nx;
  load_twiddles(tw, a.twX, Nx >> 1);
  const int n = nr * Nx;
  for (int e = threadIdx.x; e < n; e += NT) {
    s[e] = a.Z0[r0 * Nx + e]; s[st + e] = a.Z1[r0 * Nx + e]; s[2 * st + e] = a.Z2[r0 * Nx + e];
  }
  __syncthreads();
  if (nr == a.RX) fft_dif(s, 3 * nr, Nx, a.lgNx, tw, a.lgNx);
  else { fft_dif(s, nr, Nx, a.lgNx, tw, a.lgNx); fft_dif(s + st, nr, Nx, a.lgNx, tw, a.lgNx); fft_dif(s + 2 * st, nr, Nx, a.lgNx, tw, a.lgNx); }
  for (int e = threadIdx.x; e < n; e += NT) {
    const T l = a.lx_r[e & (Nx - 1)];
    const cx<T> z0 = s[e], z1 = s[st + e], z2 = s[2 * st + e];
    const cx<T> kv = mk<T>(z0.x - l * z1.y - l * l * z2.x, z0.y + l * z1.x - l * l * z2.y);
    const long gi = r0 * Nx + e;
    cx<T> y0 = a.Y0[gi];
    cx<T> acc = a.rk.stage == 1 ? mk<T>(0, 0) : a.acc[gi];
    (void)rk_update(a.rk, kv, y0, acc);
    if (a.rk.stage == 4) a.Y0[gi] = y0; else a.acc[gi] = acc;
  }
}

// gradient / hessian multipliers for precompute (src/specialops.jl:184-188): F layout in, five F-layout outputs
//   out[0]=i lx phi, out[1]=i ly phi, out[2]=-lx^2 phi, out[3]=(i lx)(i ly) phi, out[4]=-ly^2 phi
template <typename T>
__global__ __launch_bounds__(NT) void k_gradhess_mult(const cx<T>* __restrict__ phi, cx<T>* __restrict__ out,
                                                      const T* __restrict__ lx_r, const T* __restrict__ ly,
                                                      int lgNx, int Nyh, int B) {
  const long plane = (long)Nyh << lgNx;
  const long i = (long)blockIdx.x * NT + threadIdx.x;
  if (i >= plane) return;
  const T lx = lx_r[i & ((1 << lgNx) - 1)], l_y = ly[i >> lgNx];
  for (int b = 0; b < B; ++b) {
    const cx<T> v = phi[(long)b * plane + i];
    const cx<T> gx = mk<T>(-lx * v.y, lx * v.x), gy = mk<T>(-l_y * v.y, l_y * v.x);
    cx<T>* o = out + (long)b * plane + i;                      // out[comp][b][plane]
    const long cs = (long)B * plane;
    o[0] = gx; o[cs] = gy;
    o[2 * cs] = mk<T>(-lx * gx.y, lx * gx.x);                 // d/dx gx
    o[3 * cs] = mk<T>(-lx * gy.y, lx * gy.x);                 // H[2,1] = d/dx gy
    o[4 * cs] = mk<T>(-l_y * gy.y, l_y * gy.x);               // d/dy gy
  }
}

}  // namespace cmbl
