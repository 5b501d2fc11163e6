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

#ifndef CMBL_XLG
#define CMBL_XLG 4      // row kernels: radix-16 stages (radix-8 = 3 measured 5 % faster for L*f alone but 3 % slower for the gradient step)
#endif

#ifndef CMBL_YLGN
#define CMBL_YLGN 4     // column kernels, N-point (pair) transforms: cap on fused radix-2 levels per LDS round trip
#endif
#ifndef CMBL_YLGM
#define CMBL_YLGM 4     // column kernels, N/2-point (packed real) transforms
#endif

#ifndef CMBL_ROW_WAVES
#define CMBL_ROW_WAVES 1
#endif

namespace cmbl {

// debug builds (-DCMBL_STAMPS): per-workgroup phase timestamps, read back with cmbl_debug_stamps (tools/gpu_stamps.py)
#ifdef CMBL_STAMPS
__device__ unsigned long long g_stamps[8192 * 16];
#define CMBL_STAMP(i) do { if (threadIdx.x == 0) g_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = clock64(); } while (0)
#else
#define CMBL_STAMP(i) do {} while (0)
#endif
#ifdef CMBL_STAMPS_X
#define CMBL_XSTAMP(i) CMBL_STAMP(i)
#else
#define CMBL_XSTAMP(i) do {} while (0)
#endif

// register budget of the row kernels (waves per SIMD the compiler must leave room for; fp32 only)
template <typename T> constexpr int row_min_waves() { return sizeof(T) == 4 ? CMBL_ROW_WAVES : 1; }

// Column tiles that are neighbours in x share 64/128-byte lines of the [ky][x] arrays.  Workgroup b is observed to run
// on XCD b % 8 (speed only, never correctness), so give every XCD a contiguous range of tiles: its private L2 then sees
// both halves of each shared line.
__device__ __forceinline__ int xcd_tile(int b, int nb) { return (nb & 7) ? b : (b & 7) * (nb >> 3) + (b >> 3); }

// ---------------------------------------------------------------------------------------------
// ref <-> F  (transpose + bit reversal of x), V = cx<T> or T.   grid (Nx/32, ceil(Nyh/32), slices), block 256
template <typename V>
__global__ __launch_bounds__(NTP) void k_ref2F(const V* __restrict__ in, V* __restrict__ out, int Nx, int lgNx, int Nyh) {
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
__global__ __launch_bounds__(NTP) void k_F2ref(const V* __restrict__ in, V* __restrict__ out, int Nx, int lgNx, int Nyh) {
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
// Column-tile geometry, all compile time: C columns per workgroup of NT threads, M = Ny/2, R = C*M/NT packed pairs/thread.
constexpr int ilog2c(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }
template <int R, int NT, int LGM> struct ColTile {
  static constexpr int M = 1 << LGM, N = 2 * M, LGN = LGM + 1, Nyh = M + 1;
  static constexpr int C = (R * NT) >> LGM, LGC = ilog2c(C);
  static constexpr int LDN = tile_ld(N);          // tile row stride when the buffer also carries N-point (pair) transforms
  static constexpr int LDM = tile_ld(M);          // stride for kernels that only do packed M-point transforms
  static constexpr int RZ = (C * (M + 1) + NT - 1) / NT;   // half-spectrum entries per thread (R, or R+1)
  static_assert(C >= 1 && (C << LGM) == R * NT, "tile shape must satisfy C*M == R*NT");
};

// ---- staged global -> LDS loads -------------------------------------------------------------------------------------------
// The compiler will not move a global load above the LDS store of an earlier loop iteration, so a plain "s[..] = g[..]" loop
// exposes one full memory latency per iteration (measured with in-kernel timestamps: the first phase of k_delta_y took 20k
// cycles for seven dependent round trips).  Every tile load is therefore split into issue() -- all of the thread's loads, back
// to back, into registers -- and commit() -- the LDS stores -- so that a kernel issues EVERYTHING it needs from HBM and then
// waits once.
template <typename T, int NT, int NH> struct TwStage {                       // twiddle table, NH entries
  static constexpr int K = (NH + NT - 1) / NT;
  cx<T> v[K];
  __device__ __forceinline__ void issue(const cx<T>* __restrict__ g) {
#pragma unroll
    for (int i = 0; i < K; ++i) { const int j = threadIdx.x + i * NT; if (NH % NT == 0 || j < NH) v[i] = g[j]; }
  }
  __device__ __forceinline__ void commit(cx<T>* __restrict__ lds) const {
#pragma unroll
    for (int i = 0; i < K; ++i) { const int j = threadIdx.x + i * NT; if (NH % NT == 0 || j < NH) lds[j] = v[i]; }
  }
};

// tile <-> mixed-layout global helpers for column kernels.  Tile: C sequences x LD slots, half-spectrum at hslot(k).
// lanes run over c fastest so each wave touches (64/C) segments of C contiguous complex values.
template <typename T, int NT, int LGM, int LGC> struct TileStage {           // half-spectrum tile: C columns x (M+1) rows
  static constexpr int M = 1 << LGM, C = 1 << LGC, TOT = C * (M + 1), K = (TOT + NT - 1) / NT;
  cx<T> v[K];
  __device__ __forceinline__ void issue(const cx<T>* __restrict__ g /*slice base*/, int Nx, int x0) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int e = threadIdx.x + i * NT;
      if (e < TOT) v[i] = g[(size_t)(e >> LGC) * Nx + x0 + (e & (C - 1))];
    }
  }
  template <int LD> __device__ __forceinline__ void commit(cx<T>* __restrict__ s) const {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int e = threadIdx.x + i * NT;
      if (e < TOT) s[(e & (C - 1)) * LD + hslot<LGM>(e >> LGC)] = v[i];
    }
  }
};
template <typename T, int NT, int LD, int LGM, int LGC>
__device__ __forceinline__ void tile_store_mixed(const cx<T>* __restrict__ s, cx<T>* __restrict__ g, int Nx, int x0) {
  constexpr int M = 1 << LGM, C = 1 << LGC;
  for (int e = threadIdx.x; e < (C * (M + 1)); e += NT) {
    const int c = e & (C - 1), k = e >> LGC;
    g[(size_t)k * Nx + x0 + c] = s[c * LD + hslot<LGM>(k)];
  }
}

// Two real sequences per complex transform.  Z = X + iY with X, Y the (Hermitian-extended) half spectra:
//   Z[k] = X[k] + i Y[k],  Z[N-k] = conj(X[k]) + i conj(Y[k])  (0<k<M);  Z[0], Z[M] from the real parts only (c2r semantics).
// After the N-point DIT the tile holds N*(x[n] + i y[n]).  Frequency k sits at slot pad(brev_N(k)).
// Here X = gX, Y = i*ly[k]*gY (the d/dy multiply rides along).
template <typename T, int NT, int LGN, int LGC> struct PairStage {
  static constexpr int N = 1 << LGN, M = N >> 1, C = 1 << LGC, TOT = C * (M + 1), K = (TOT + NT - 1) / NT;
  cx<T> X[K], Y[K];
  T l[K];
  __device__ __forceinline__ void issue(const cx<T>* __restrict__ gX, const cx<T>* __restrict__ gY, const T* __restrict__ ly, int Nx, int x0) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int e = threadIdx.x + i * NT;
      if (e < TOT) {
        const int k = e >> LGC;
        const size_t gi = (size_t)k * Nx + x0 + (e & (C - 1));
        X[i] = gX[gi]; Y[i] = gY[gi]; l[i] = ly[k];
      }
    }
  }
  template <int LD> __device__ __forceinline__ void commit(cx<T>* __restrict__ s) const {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int e = threadIdx.x + i * NT;
      if (e < TOT) {
        const int c = e & (C - 1), k = e >> LGC;
        const cx<T> x = X[i], y = mk<T>(-l[i] * Y[i].y, l[i] * Y[i].x);
        cx<T>* p = s + c * LD;
        if (k == 0 || k == M) {
          p[pad(brevc<LGN>(k))] = mk<T>(x.x, y.x);
        } else {
          p[pad(brevc<LGN>(k))] = mk<T>(x.x - y.y, x.y + y.x);
          p[pad(brevc<LGN>(N - k))] = mk<T>(x.x + y.y, y.x - x.y);
        }
      }
    }
  }
};
// Row kernels: `rows` rows are dealt to `nblk` workgroups as evenly as possible (the first rows % nblk workgroups take one more).
// LDS row capacity of a workgroup = ceil(rows / nblk).
__device__ __forceinline__ void row_range(long rows, int nblk, long blk, long& r0, int& nr) {
  const long base = rows / nblk, extra = rows - base * nblk;
  r0 = blk * base + (blk < extra ? blk : extra);
  nr = (int)base + (blk < extra ? 1 : 0);
}
// rows of Nx contiguous values -> row tiles (row kernels): the loads of one row are issued together
template <typename T, int NT, int LGNX, int NA>
__device__ __forceinline__ void rows_load(cx<T>* const (&s)[NA], const cx<T>* const (&g)[NA], int nr) {
  constexpr int Nx = 1 << LGNX, LD = tile_ld(Nx);
  if constexpr (Nx >= NT) {
    constexpr int PF = Nx / NT;
    for (int r = 0; r < nr; ++r) {
      cx<T> v[NA][PF];
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int i = 0; i < PF; ++i) v[a][i] = g[a][(size_t)r * Nx + threadIdx.x + i * NT];
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int i = 0; i < PF; ++i) s[a][r * LD + pad(threadIdx.x + i * NT)] = v[a][i];
    }
  } else {
    for (int e = threadIdx.x; e < nr * Nx; e += NT) {
      const int si = (e >> LGNX) * LD + pad(e & (Nx - 1));
#pragma unroll
      for (int a = 0; a < NA; ++a) s[a][si] = g[a][e];
    }
  }
}
// After the N-point DIF of a + i b (a, b real): A[k] = (Z[k] + conj Z[N-k])/2, B[k] = (Z[k] - conj Z[N-k])/(2i), k = 0..M.
// f(k, c, A, B) consumes the pair (stores it, or combines it with something held in registers).
template <typename T, int NT, int LD, int LGN, int LGC, int RZ, typename F>
__device__ __forceinline__ void pair_split(const cx<T>* __restrict__ s, F&& f) {
  constexpr int N = 1 << LGN, M = N >> 1, C = 1 << LGC;
#pragma unroll
  for (int i = 0; i < RZ; ++i) {                    // RZ = ceil(C*(M+1)/NT): compile-time trip count keeps f's captures in registers
    const int e = threadIdx.x + i * NT;
    if (e < C * (M + 1)) {
      const int c = e & (C - 1), k = e >> LGC;
      const cx<T>* p = s + c * LD;
      const cx<T> zk = p[pad(brevc<LGN>(k))], zn = p[pad(brevc<LGN>((N - k) & (N - 1)))];
      f(i, k, c, mk<T>(T(0.5) * (zk.x + zn.x), T(0.5) * (zk.y - zn.y)), mk<T>(T(0.5) * (zk.y + zn.y), T(0.5) * (zn.x - zk.x)));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// y pass, forward: map -> mixed.   grid (Nx/C, slices).  LDS: twY[M] + C*tile_ld(M) cplx
template <typename T, int R, int NT, int LGM>
__global__ __launch_bounds__(NT) void k_y_r2c(const T* __restrict__ in, cx<T>* __restrict__ out, const cx<T>* __restrict__ twY, int Nx) {
  using G = ColTile<R, NT, LGM>;
  constexpr int M = G::M, LD = G::LDM, C = G::C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;
  const int x0 = xcd_tile(blockIdx.x, gridDim.x) * C;
  const size_t sl = blockIdx.y;
  TwStage<T, NT, M> twr;
  twr.issue(twY);
  const cx<T>* src = reinterpret_cast<const cx<T>*>(in) + (sl * Nx + x0) * (size_t)M;
  cx<T> v[R];
#pragma unroll
  for (int i = 0; i < R; ++i) v[i] = src[threadIdx.x + i * NT];
  twr.commit(tw);
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int e = threadIdx.x + i * NT, c = e >> LGM, j = e & (M - 1);
    s[c * LD + pad(j)] = v[i];
  }
  __syncthreads();
  fft_dif<T, NT, LD, LGM, LGM + 1, CMBL_YLGM>(s, C, tw);
  r2c_post<T, NT, LD, LGM>(s, C, tw);
  tile_store_mixed<T, NT, LD, LGM, G::LGC>(s, out + sl * (size_t)G::Nyh * Nx, Nx, x0);
}

// y pass, inverse: mixed -> map, scaled by `scale` (1/Ny; the x pass already carries 1/Nx)
template <typename T, int R, int NT, int LGM>
__global__ __launch_bounds__(NT) void k_y_c2r(const cx<T>* __restrict__ in, T* __restrict__ out, const cx<T>* __restrict__ twY, int Nx, T scale) {
  using G = ColTile<R, NT, LGM>;
  constexpr int M = G::M, LD = G::LDM, C = G::C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;
  const int x0 = xcd_tile(blockIdx.x, gridDim.x) * C;
  const size_t sl = blockIdx.y;
  TwStage<T, NT, M> twr;
  TileStage<T, NT, LGM, G::LGC> tl;
  twr.issue(twY);
  tl.issue(in + sl * (size_t)G::Nyh * Nx, Nx, x0);
  twr.commit(tw);
  tl.template commit<LD>(s);
  __syncthreads();
  c2r_pre<T, NT, LD, LGM>(s, C, tw);
  fft_dit<T, NT, LD, LGM, LGM + 1, CMBL_YLGM>(s, C, tw);
  cx<T>* dst = reinterpret_cast<cx<T>*>(out) + (sl * Nx + x0) * (size_t)M;
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const int e = threadIdx.x + i * NT, c = e >> LGM, j = e & (M - 1);
    dst[e] = scale * s[c * LD + pad(j)];
  }
}

// y pass of the pixel-mask sandwich  rfft2( m .* irfft2(x) )  (M = Mfourier * Mpix, src/dataset.jl:279-285): mixed -> map (in LDS /
// registers only) -> x mask -> mixed.  One launch and no HBM round trip for the map instead of y_c2r, mask multiply, y_r2c.
template <typename T, int R, int NT, int LGM>
__global__ __launch_bounds__(NT) void k_y_mask(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, const T* __restrict__ mask,
                                               const cx<T>* __restrict__ twY, int Nx, T scale) {
  using G = ColTile<R, NT, LGM>;
  constexpr int M = G::M, LD = G::LDM, C = G::C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;
  const int x0 = xcd_tile(blockIdx.x, gridDim.x) * C;
  const size_t sl = blockIdx.y;
  TwStage<T, NT, M> twr;
  TileStage<T, NT, LGM, G::LGC> tl;
  twr.issue(twY);
  tl.issue(in + sl * (size_t)G::Nyh * Nx, Nx, x0);
  const cx<T>* mk2 = reinterpret_cast<const cx<T>*>(mask) + (size_t)x0 * M;       // the mask is one (Nx, Ny) map for all slices
  cx<T> mv[R];
#pragma unroll
  for (int i = 0; i < R; ++i) mv[i] = mk2[threadIdx.x + i * NT];
  twr.commit(tw);
  tl.template commit<LD>(s);
  __syncthreads();
  c2r_pre<T, NT, LD, LGM>(s, C, tw);
  fft_dit<T, NT, LD, LGM, LGM + 1, CMBL_YLGM>(s, C, tw);
#pragma unroll
  for (int i = 0; i < R; ++i) {                       // each thread scales the packed pairs it owns: same slots read and written
    const int e = threadIdx.x + i * NT, c = e >> LGM, j = e & (M - 1);
    const cx<T> v = s[c * LD + pad(j)];
    s[c * LD + pad(j)] = mk<T>(scale * mv[i].x * v.x, scale * mv[i].y * v.y);
  }
  __syncthreads();
  fft_dif<T, NT, LD, LGM, LGM + 1, CMBL_YLGM>(s, C, tw);
  r2c_post<T, NT, LD, LGM>(s, C, tw);
  tile_store_mixed<T, NT, LD, LGM, G::LGC>(s, out + sl * (size_t)G::Nyh * Nx, Nx, x0);
}

// ---------------------------------------------------------------------------------------------
// x pass on contiguous rows.  `rows` = slices*Nyh rows of Nx.  grid nblk.  LDS: twX[Nx/2] + ceil(rows/nblk)*tile_ld(Nx) cplx
//   MODE 0: forward  (mixed -> F)
//   MODE 1: inverse  (F -> mixed), scaled by 1/Nx
//   MODE 2: x-derivative  (mixed -> mixed):  ifft_x( i*lx * fft_x(row) ) / Nx        (src/proj_lambert.jl:146-159, coord 1)
template <typename T, int MODE, int NT, int LGNX>
__global__ __launch_bounds__(NT, row_min_waves<T>()) void k_x_fft(const cx<T>* __restrict__ in, cx<T>* __restrict__ out,
                                              const cx<T>* __restrict__ twX, const T* __restrict__ lx_r, long rows, int nblk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int Nx = 1 << LGNX, LD = tile_ld(Nx);
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + (Nx >> 1);
  long r0; int nr;
  row_range(rows, nblk, blockIdx.x, r0, nr);
  CMBL_XSTAMP(0);
  TwStage<T, NT, (Nx >> 1)> twr;
  twr.issue(twX);
  const T inv = T(1) / T(Nx);
  const T dl = MODE == 2 ? lx_r[1] * T(-2) * inv * inv : T(0);      // lx_r[1] = lx(kx = Nx/2) = -(Nx/2) dlx
  {
    cx<T>* const sa[1] = {s};
    const cx<T>* const ga[1] = {in + r0 * Nx};
    rows_load<T, NT, LGNX, 1>(sa, ga, nr);
  }
  twr.commit(tw);
  __syncthreads();
  CMBL_XSTAMP(1);
  if (MODE == 0 || MODE == 2) fft_dif<T, NT, LD, LGNX, LGNX, CMBL_XLG>(s, nr, tw);
  CMBL_XSTAMP(2);
  if (MODE == 2) {
    // i*lx/Nx multiply fused into the loads of the first inverse stage: slot i holds kx = bitrev(i)
    fft_dit<T, NT, LD, LGNX, LGNX, CMBL_XLG>(s, nr, tw, [dl](cx<T> v, int i) {
      const int kx = brevc<LGNX>(i);
      return mk<T>(-(dl * T(kx < (Nx >> 1) ? kx : kx - Nx)) * v.y, (dl * T(kx < (Nx >> 1) ? kx : kx - Nx)) * v.x);
    });
  }
  if (MODE == 1) fft_dit<T, NT, LD, LGNX, LGNX, CMBL_XLG>(s, nr, tw);
  CMBL_XSTAMP(3);
  cx<T>* dst = out + r0 * Nx;
  if (MODE == 1) { for (int e = threadIdx.x; e < nr * Nx; e += NT) dst[e] = inv * s[(e >> LGNX) * LD + pad(e & (Nx - 1))]; }
  else           { for (int e = threadIdx.x; e < nr * Nx; e += NT) dst[e] = s[(e >> LGNX) * LD + pad(e & (Nx - 1))]; }
  CMBL_XSTAMP(4);
}

}  // namespace cmbl
