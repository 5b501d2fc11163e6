// Layout converters and the 2-D real FFT passes.
//
// Layouts (P*B "slices" always outermost):
//   map      : real   [slice][x][y]            (== reference (Ny,Nx,P,B) column-major, src/proj_cartesian.jl:13-36)
//   ref      : cplx   [slice][x][ky]           (== reference half-plane (Ny/2+1,Nx,P,B))
//   mixed    : cplx   [slice][ky][x]           y-transformed only, x natural      (internal)
//   F        : cplx   [slice][ky][xr]          fully transformed, xr = bitrev(kx) (internal Fourier layout)
// The y pass ("column kernel") owns the transposition: it reads/writes whole contiguous columns on the map
// side and C-wide segments on the [ky][x] side.  The x pass ("row kernel") then works on contiguous rows.
#pragma once
#include "fft_lds.hpp"

namespace cmbl {

// ---------------------------------------------------------------------------------------------
// ref <-> F  (transpose + bit reversal of x), V = cx<T> or T.   grid (Nx/32, ceil(Nyh/32), slices), block 256
template <typename V>
__global__ __launch_bounds__(NT) void k_ref2F(const V* __restrict__ in, V* __restrict__ out, int Nx, int lgNx, int Nyh) {
  __shared__ V tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t sl = blockIdx.z;
  const int xr0 = blockIdx.x * 32, ky0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int xl = ty + 8 * i, x = brev(xr0 + xl, lgNx), ky = ky0 + tx;
    if (ky < Nyh) tile[xl][tx] = in[(sl * Nx + x) * Nyh + ky];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kl = ty + 8 * i, ky = ky0 + kl;
    if (ky < Nyh) out[(sl * Nyh + ky) * Nx + xr0 + tx] = tile[tx][kl];
  }
}

template <typename V>
__global__ __launch_bounds__(NT) void k_F2ref(const V* __restrict__ in, V* __restrict__ out, int Nx, int lgNx, int Nyh) {
  __shared__ V tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t sl = blockIdx.z;
  const int xr0 = blockIdx.x * 32, ky0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kl = ty + 8 * i, ky = ky0 + kl;
    if (ky < Nyh) tile[tx][kl] = in[(sl * Nyh + ky) * Nx + xr0 + tx];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int xl = ty + 8 * i, x = brev(xr0 + xl, lgNx), ky = ky0 + tx;
    if (ky < Nyh) out[(sl * Nx + x) * Nyh + ky] = tile[xl][tx];
  }
}

// ---------------------------------------------------------------------------------------------
// tile <-> mixed-layout global helpers for column kernels.  Tile: C sequences x LD slots, half-spectrum at hslot(k).
// lanes run over c fastest so each wave touches (64/C) segments of C contiguous complex values.
template <typename T, typename F>
__device__ __forceinline__ void tile_load_mixed(cx<T>* __restrict__ s, const cx<T>* __restrict__ g /*slice base*/,
                                                int Nx, int x0, int C, int lgC, int lgM, F&& f) {
  const int M = 1 << lgM, LD = M + 1;
  for (int e = threadIdx.x; e < (C * (M + 1)); e += NT) {
    const int c = e & (C - 1), k = e >> lgC;
    s[c * LD + hslot(k, M, lgM)] = f(g[(size_t)k * Nx + x0 + c], k);
  }
}
template <typename T, typename F>
__device__ __forceinline__ void tile_store_mixed(const cx<T>* __restrict__ s, cx<T>* __restrict__ g,
                                                 int Nx, int x0, int C, int lgC, int lgM, F&& f) {
  const int M = 1 << lgM, LD = M + 1;
  for (int e = threadIdx.x; e < (C * (M + 1)); e += NT) {
    const int c = e & (C - 1), k = e >> lgC;
    g[(size_t)k * Nx + x0 + c] = f(s[c * LD + hslot(k, M, lgM)], k);
  }
}

// ---------------------------------------------------------------------------------------------
// y pass, forward: map -> mixed.   grid (Nx/C, slices).  LDS: twY[M] + C*(M+1) cplx
template <typename T>
__global__ __launch_bounds__(NT) void k_y_r2c(const T* __restrict__ in, cx<T>* __restrict__ out,
                                              const cx<T>* __restrict__ twY, int Nx, int lgM, int C, int lgC) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int M = 1 << lgM, LD = M + 1, Nyh = M + 1;
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;
  const int x0 = blockIdx.x * C;
  const size_t sl = blockIdx.y;
  load_twiddles(tw, twY, M);
  const cx<T>* src = reinterpret_cast<const cx<T>*>(in) + (sl * Nx + x0) * (size_t)M;
  for (int e = threadIdx.x; e < C * M; e += NT) {
    const int c = e >> lgM, j = e & (M - 1);
    s[c * LD + j] = src[(size_t)c * M + j];
  }
  __syncthreads();
  fft_dif(s, C, LD, lgM, tw, lgM + 1);
  r2c_post(s, C, LD, lgM, tw);
  tile_store_mixed(s, out + sl * (size_t)Nyh * Nx, Nx, x0, C, lgC, lgM, [](cx<T> v, int) { return v; });
}

// y pass, inverse: mixed -> map, scaled by `scale` (1/Ny; the x pass already carries 1/Nx)
template <typename T>
__global__ __launch_bounds__(NT) void k_y_c2r(const cx<T>* __restrict__ in, T* __restrict__ out,
                                              const cx<T>* __restrict__ twY, int Nx, int lgM, int C, int lgC, T scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int M = 1 << lgM, LD = M + 1, Nyh = M + 1;
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;
  const int x0 = blockIdx.x * C;
  const size_t sl = blockIdx.y;
  load_twiddles(tw, twY, M);
  tile_load_mixed(s, in + sl * (size_t)Nyh * Nx, Nx, x0, C, lgC, lgM, [](cx<T> v, int) { return v; });
  __syncthreads();
  c2r_pre(s, C, LD, lgM, tw);
  fft_dit(s, C, LD, lgM, tw, lgM + 1);
  cx<T>* dst = reinterpret_cast<cx<T>*>(out) + (sl * Nx + x0) * (size_t)M;
  for (int e = threadIdx.x; e < C * M; e += NT) {
    const int c = e >> lgM, j = e & (M - 1);
    dst[(size_t)c * M + j] = scale * s[c * LD + j];
  }
}

// ---------------------------------------------------------------------------------------------
// x pass on contiguous rows.  `rows` = slices*Nyh rows of Nx.  grid ceil(rows/RX).  LDS: twX[Nx/2] + RX*Nx cplx
//   MODE 0: forward  (mixed -> F)
//   MODE 1: inverse  (F -> mixed), scaled by 1/Nx
//   MODE 2: x-derivative  (mixed -> mixed):  ifft_x( i*lx * fft_x(row) ) / Nx        (src/proj_lambert.jl:146-159, coord 1)
template <typename T, int MODE>
__global__ __launch_bounds__(NT) void k_x_fft(const cx<T>* __restrict__ in, cx<T>* __restrict__ out,
                                              const cx<T>* __restrict__ twX, const T* __restrict__ lx_r,
                                              int lgNx, long rows, int RX) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Nx = 1 << lgNx;
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + (Nx >> 1);
  const long r0 = (long)blockIdx.x * RX;
  const int nr = (int)min((long)RX, rows - r0);
  load_twiddles(tw, twX, Nx >> 1);
  const cx<T>* src = in + r0 * Nx;
  for (int e = threadIdx.x; e < nr * Nx; e += NT) s[e] = src[e];
  __syncthreads();
  const T inv = T(1) / T(Nx);
  if (MODE == 0 || MODE == 2) fft_dif(s, nr, Nx, lgNx, tw, lgNx);
  if (MODE == 2) {
    for (int e = threadIdx.x; e < nr * Nx; e += NT) {
      const T l = lx_r[e & (Nx - 1)] * inv;
      cx<T> v = s[e];
      s[e] = mk<T>(-l * v.y, l * v.x);
    }
    __syncthreads();
  }
  if (MODE == 1 || MODE == 2) fft_dit(s, nr, Nx, lgNx, tw, lgNx);
  cx<T>* dst = out + r0 * Nx;
  if (MODE == 1) { for (int e = threadIdx.x; e < nr * Nx; e += NT) dst[e] = inv * s[e]; }
  else           { for (int e = threadIdx.x; e < nr * Nx; e += NT) dst[e] = s[e]; }
}

}  // namespace cmbl
