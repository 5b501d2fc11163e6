// Layout converters and the 2-D real FFT passes.
//
// Layouts (P*B "slices" always outermost):
//   map      : real   [slice][x][y]            (== reference (Ny,Nx,P,B) column-major, src/proj_cartesian.jl:13-36)
//   ref      : cplx   [slice][x][ky]           (== reference half-plane (Ny/2+1,Nx,P,B))
//   mixed    : cplx   [slice][x/4][ky][x%4]    y-transformed only, x natural      (internal, tiled: see mix_idx)
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

__host__ __device__ constexpr int ilog2c(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

// debug builds (-DCMBL_STAMPS): per-workgroup phase timestamps, read back with cmbl_debug_stamps (tools/gpu_stamps.py)
// (the buffer is per translation unit -- device symbols are not shared without relocatable device code -- so each unit exports a reader,
//  CMBL_STAMPS_READER below, and cmbl_debug_stamps picks the unit: CMBL_STAMPS_TU = main_f32 (default) | main_f64 | gen_f32 | cty_f32_a | ctx_f32_b | ...)
#ifdef CMBL_STAMPS
__device__ unsigned long long g_stamps[8192 * 16];
#define CMBL_STAMPS_READER(unit) namespace cmbl { int stamps_read_##unit(unsigned long long* out_host, int n) { \
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * n); } }
#else
#define CMBL_STAMPS_READER(unit)
#endif
#if defined(CMBL_STAMPS) && !defined(CMBL_STAMPS_ROWS)      // CMBL_STAMPS_ROWS: only k_delta_rows writes (its own slots 14 / 15)
#define CMBL_STAMP(i) do { if (threadIdx.x == 0) g_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = clock64(); } while (0)
// slots 14 / 15: entry / exit on the 100 MHz wall clock, which all XCDs share (launch timeline across the chip)
#define CMBL_WSTAMP(i) do { if (threadIdx.x == 0) g_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define CMBL_STAMP(i) do {} while (0)
#define CMBL_WSTAMP(i) do {} while (0)
#endif
// -DCMBL_STAMPS_WAVES: every wavefront of a k_x_fft workgroup records when it starts and ends its transform chain (second half of
// g_stamps: [block][wave][2]; tools/gpu_stamps_x.py prints the spread over the waves of a workgroup)
#ifdef CMBL_STAMPS_WAVES
#define CMBL_WVSTAMP(i) do { if ((threadIdx.x & 63) == 0) g_stamps[4096 * 16 + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + (i)] = clock64(); } while (0)
#else
#define CMBL_WVSTAMP(i) do {} while (0)
#endif
#ifdef CMBL_STAMPS_X
#define CMBL_XSTAMP(i) CMBL_STAMP(i)
#define CMBL_XWSTAMP(i) CMBL_WSTAMP(i)
#else
#define CMBL_XSTAMP(i) do {} while (0)
#define CMBL_XWSTAMP(i) do {} while (0)
#endif

// register budget of the row kernels (waves per SIMD the compiler must leave room for; fp32 only)
template <typename T> constexpr int row_min_waves() { return sizeof(T) == 4 ? CMBL_ROW_WAVES : 1; }

// Column tiles that are neighbours in x share 64/128-byte lines of the [ky][x] arrays.  Workgroup b is observed to run
// on XCD b % 8 (speed only, never correctness), so give every XCD a contiguous range of tiles: its private L2 then sees
// both halves of each shared line.
__device__ __forceinline__ int xcd_tile(int b, int nb) { return (nb & 7) ? b : (b & 7) * (nb >> 3) + (b >> 3); }

// The "mixed" arrays (y-transformed, x natural) are what column kernels hand to row kernels and back, so one side always reads
// them across its natural direction.  As [ky][x] the column tiles (C = 4 columns) pull 32-byte pieces 8 KB apart: measured
// (tools/micro/ldbench.hip) 11.0 us for the 16.8 MB of a launch's (Gx, A) tiles against 2.9 us for the same bytes as contiguous
// blocks -- the largest single cost in the round-1 column kernels (first LDS commit at 14k of a 39k-cycle workgroup).  Hence the
// tiled layout  [x/4][ky][x%4]: a column tile of 4 columns is ONE contiguous block of (Ny/2+1)*4 values, and a row workgroup that
// takes 4 adjacent ky rows gathers whole 128-byte lines (4 rows x 4 values), which is as fast as contiguous rows (2.9 us / 2.9 us).
// The row count of a block is padded to a multiple of 4 (Ny/2+1 is odd): blocks and the gathered 4 x 4 lines then start on 128-byte
// boundaries (unpadded, every line of a row workgroup straddled two: +35 % measured traffic in the row kernels).
// Block width per precision (round 6): 4 columns of 8-byte elements, CMBL_MIXW64 (2) columns of 16-byte ones -- a block row is then 32 bytes in both
// precisions, 4 padded rows x one block row = one 128-byte line, and the TWO-column tile the double-precision delta-flow kernel runs on (engine.hpp
// tileY_delta: the four-column tile spills) is ONE contiguous block of whole lines instead of the 32-byte halves of a four-column block's 64-byte
// rows (those half lines were fetched by both neighbours: +18 % L2 read requests at 2048^2 fp64, profiles/r05_pmc_tcc_requests.txt).
// -DCMBL_MIXW64=4 restores the round-2..5 layout for A/B.
#ifndef CMBL_MIXW64
#define CMBL_MIXW64 4
#endif
template <int BYTES> __host__ __device__ constexpr int mixw_b() { return BYTES == 16 ? CMBL_MIXW64 : 4; }      // BYTES = sizeof(cx<T>)
template <typename T> __host__ __device__ constexpr int mixw() { return mixw_b<(int)sizeof(cx<T>)>(); }
template <int W> __host__ __device__ constexpr int lg_mixw() { return W == 4 ? 2 : W == 2 ? 1 : 0; }
static_assert(CMBL_MIXW64 == 1 || CMBL_MIXW64 == 2 || CMBL_MIXW64 == 4, "block width of the double-precision mixed layout");
__host__ __device__ constexpr int mixed_rows(int Nyh) { return (Nyh + 3) & ~3; }
template <int W> __device__ __forceinline__ size_t mix_idx(int ky, int x, int NyhP) { return ((size_t)(x >> lg_mixw<W>()) * NyhP + ky) * W + (x & (W - 1)); }
// Addressing: uniform 64-bit base (scalar registers) + 32-bit unsigned byte offset (one vector register) is the form global_load /
// global_store take directly, with no 64-bit vector arithmetic per access (the fused kernels are partly VALU-issue bound, and a
// third of their vector instructions was address arithmetic).  Offsets inside one slice of a field stay far below 4 GB.
template <typename V> __device__ __forceinline__ const V& at32(const V* base, unsigned idx) {
  return *reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + idx * (unsigned)sizeof(V));
}
template <typename V> __device__ __forceinline__ V& at32(V* base, unsigned idx) {
  return *reinterpret_cast<V*>(reinterpret_cast<char*>(base) + idx * (unsigned)sizeof(V));
}
// Hand-off stores.  The mixed-layout arrays a column kernel hands to the next row kernel (and back) are read by workgroups on all
// eight XCDs, i.e. through the fabric, whatever the producer's L2 holds.  Written with plain stores they stay dirty in the producing
// XCD's L2 until the write-back at the kernel boundary, and the dependent launch waits for it (MI355X_MICROARCH.md: + B / 6 TB/s for B
// dirty bytes); written through (`sc1`) they leave while the kernel still computes.  Measured at 1024^2 QU, per launch
// (profiles/r04_ab_write_through.txt): x_grad 7.6 -> 6.7 us, adj_y 12.0 -> 10.8 us, flow_y_fwd 13.5 -> 12.7 us, delta_cols 20.3 -> 19.5 us.
// NOT for arrays the same workgroup index re-reads in the next stage (the RK state): sc1 drops the line from the L2 and those
// re-reads then miss (adj_x +14 %).  One store per complex value (8 bytes in single precision): 16-byte forms measured the same.
#ifndef CMBL_WT_STORES
#define CMBL_WT_STORES 1
#endif
typedef float wt_f2 __attribute__((ext_vector_type(2)));
typedef float wt_f4 __attribute__((ext_vector_type(4)));
template <int BYTES> __device__ __forceinline__ void store_wt(void* q, const void* v) {
  static_assert(BYTES == 4 || BYTES == 8 || BYTES == 16, "hand-off stores are 4, 8 or 16 bytes");
#if CMBL_WT_STORES
  // The compiler cannot see that the asm is a store of more than 64 bits, so it does not keep the two wait states gfx950 needs before a
  // vector instruction overwrites the store's data registers (it scheduled `v_or_b32 v2, ...` right behind `global_store_dwordx4 .., v[2:5]`
  // in k_delta_rows: wrong Gx in single AND double precision).  The s_nop supplies them.
  // (A relaxed agent-scope __hip_atomic_store compiles to the same `global_store_dword[x2] ... sc1` without asm; measured equal for the
  // 4 / 8-byte sites, but there is no 16-byte form and two 8-byte stores cost the row storers 2.5 us per launch: r04_ab_write_through.txt)
  if constexpr (BYTES == 16) { const wt_f4 d = *reinterpret_cast<const wt_f4*>(v); asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(q), "v"(d) : "memory"); }
  else if constexpr (BYTES == 8) { const wt_f2 d = *reinterpret_cast<const wt_f2*>(v); asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(q), "v"(d) : "memory"); }
  else { const float d = *reinterpret_cast<const float*>(v); asm volatile("global_store_dword %0, %1, off sc1" : : "v"(q), "v"(d) : "memory"); }
#else
  if constexpr (BYTES == 16) *reinterpret_cast<wt_f4*>(q) = *reinterpret_cast<const wt_f4*>(v);
  else if constexpr (BYTES == 8) *reinterpret_cast<wt_f2*>(q) = *reinterpret_cast<const wt_f2*>(v);
  else *reinterpret_cast<float*>(q) = *reinterpret_cast<const float*>(v);
#endif
}
// Only for transforms whose lines (a column of the column kernels, a row of the row kernels) are at most 16 KB.  Above that -- 2048^2 in
// double, 4096^2 in single precision -- sc1 LOSES: delta_cols 198 -> 236 us at 2048^2 fp64 and 497 -> 590 us at 4096^2 fp32 with it in
// the column kernels, and with it in the row storers alone grad lnP +3.4 % at 4096^2 fp32 (the consumers of the written-through rows
// slow down).  Every shape up to 16 KB gains or is neutral: 512^2 .. 2048^2 fp32, 512^2 and 1024^2 fp64
// (profiles/r04_ab_write_through.txt).
template <typename T> __host__ __device__ constexpr bool wt_line(long n) { return (size_t)n * sizeof(cx<T>) <= 16 * 1024; }
template <typename T> __host__ __device__ constexpr bool wt_cols(int /*C*/, int M) { return wt_line<T>(2 * M); }
template <typename T, bool WT> __device__ __forceinline__ void handoff_store(cx<T>* base, unsigned idx, cx<T> v) {
  if constexpr (WT) store_wt<(int)sizeof(cx<T>)>(reinterpret_cast<char*>(base) + idx * (unsigned)sizeof(cx<T>), &v);
  else at32(base, idx) = v;
}

// A column tile of C = MIXW columns is ONE contiguous block of the mixed layout: entry (ky, c) of the tile at x0 sits at
// tile_base(g, x0) + ky * MIXW + c.  Other widths go through mix_idx.
template <typename V> __device__ __forceinline__ V* tile_base(V* g, int x0, int NyhP) {
  constexpr int W = mixw_b<(int)sizeof(V)>();
  return g + (size_t)(x0 >> lg_mixw<W>()) * NyhP * W;
}
template <typename T, int C> __device__ __forceinline__ unsigned tile_off(int ky, int c, int x0, int NyhP) {
  constexpr int W = mixw<T>();
  if constexpr (C == W) return (unsigned)(ky * W + c);
  else return (unsigned)(mix_idx<W>(ky, x0 + c, NyhP) - (size_t)(x0 >> lg_mixw<W>()) * NyhP * W);
}

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
  cx<T> v[K] = {};            // zero-initialised: a conditionally written register array otherwise lands in scratch memory (seen in fp64)
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
  cx<T> v[K] = {};
  __device__ __forceinline__ void issue(const cx<T>* __restrict__ g /*slice base*/, int Nx, int x0) {
    const cx<T>* tg = tile_base(g, x0, mixed_rows(M + 1));
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int e = threadIdx.x + i * NT;
      if (e < TOT) v[i] = at32(tg, tile_off<T, C>(e >> LGC, e & (C - 1), x0, mixed_rows(M + 1)));
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
  cx<T>* tg = tile_base(g, x0, mixed_rows(M + 1));
  for (int e = threadIdx.x; e < (C * (M + 1)); e += NT) {
    const int c = e & (C - 1), k = e >> LGC;
    at32(tg, tile_off<T, C>(k, c, x0, mixed_rows(M + 1))) = s[c * LD + hslot<LGM>(k)];
  }
}

#ifndef CMBL_COL_SPLIT
#define CMBL_COL_SPLIT 1
#endif
// threads that run the sub-stages of one column of a packed-real (M-point) transform: NT/C (two wavefronts, one per half-column; a
// radix-8 stage then has 32 butterflies per wave, half the lanes idle) or 64 (one wavefront per column, every lane busy, the other
// wavefronts of the workgroup wait at the barrier and leave the LDS / VALU pipes to the co-resident workgroup)
#ifndef CMBL_MPT_RT
#define CMBL_MPT_RT 64
#endif
template <int NT, int C> constexpr int mpt_rt() { return CMBL_MPT_RT ? CMBL_MPT_RT : NT / C; }
template <int R, int NT, int LGM> struct PairMap {
  static constexpr int M = 1 << LGM, MH = M >> 1, C = (R * NT) >> LGM;
  static constexpr bool split = CMBL_COL_SPLIT && (R % 2 == 0) && (NT % MH == 0) && (NT % C == 0) && (NT / C == 64 || NT / C == 128) && M >= 32;
  static constexpr int XLG = LGM + 1 >= 11 ? 4 : 3;            // sub-stage radix of the split N-point transforms
  __device__ static __forceinline__ int e(int i, int tid = (int)threadIdx.x) {
    if constexpr (split) {
      const int g = tid / MH, q = tid % MH;
      return ((g * (R / 2) + (i >> 1)) << LGM) + q + MH * (i & 1);
    } else return tid + i * NT;
  }
};

// ---- packed-real (M-point) transforms of a column tile, with everything around the butterflies fused ------------------------
// HalfStage: the half-spectrum tile loader that pairs A[k] with A[M-k] and does the c2r preparation while committing
//   Z[k] = (A[k] + conj A[M-k]) + i conj(w^k) (A[k] - conj A[M-k]),  Z[M-k] = conj(...)      (Im A[0], Im A[M] dropped: FFTW c2r)
// straight to the bit-reversed slots of the M-point inverse transform (twiddles w^k = exp(-2 pi i k / N) read from the global table:
// the LDS copy is not synchronised yet).  Replaces TileStage + c2r_pre: one LDS pass and one barrier less.
template <typename T, int NT, int LGM, int LGC> struct HalfStage {
  static constexpr int M = 1 << LGM, C = 1 << LGC, NP = (M >> 1) + 1, TOT = C * NP, K = (TOT + NT - 1) / NT, NyhP = mixed_rows(M + 1);
  cx<T> a[K] = {}, b[K] = {}, w[K] = {};
  __device__ __forceinline__ void issue(const cx<T>* __restrict__ g /*slice base*/, const cx<T>* __restrict__ twg, int x0, int tid = (int)threadIdx.x) {
    const cx<T>* tg = tile_base(g, x0, NyhP);
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int u = tid + i * NT;
      if (TOT % NT == 0 || u < TOT) {
        const int k = u >> LGC, c = u & (C - 1);
        a[i] = at32(tg, tile_off<T, C>(k, c, x0, NyhP)); b[i] = at32(tg, tile_off<T, C>(M - k, c, x0, NyhP)); w[i] = at32(twg, (unsigned)k);
      }
    }
  }
  template <int LD> __device__ __forceinline__ void commit(cx<T>* __restrict__ s, int tid = (int)threadIdx.x) const {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int u = tid + i * NT;
      if (TOT % NT == 0 || u < TOT) {
        const int k = u >> LGC, k2 = M - k;
        cx<T>* p = s + (u & (C - 1)) * LD;
        if (k == 0) p[0] = mk<T>(a[i].x + b[i].x, a[i].x - b[i].x);
        else {
          const cx<T> e = mk<T>(a[i].x + b[i].x, a[i].y - b[i].y), o = mk<T>(a[i].x - b[i].x, a[i].y + b[i].y);
          const cx<T> wo = mul_i(cmulconj(o, w[i]));
          p[pad(brevc<LGM>(k))] = e + wo;
          if (k2 != k) p[pad(brevc<LGM>(k2))] = conj(e - wo);
        }
      }
    }
  }
};
// the r2c finish fused into the store: Z (bit-reversed slots, after the M-point forward transform) -> A[k], A[M-k] -> mixed layout
template <typename T, int NT, int LD, int LGM, int LGC>
__device__ __forceinline__ void half_store(const cx<T>* __restrict__ s, cx<T>* __restrict__ g, const cx<T>* __restrict__ tw, int x0, int tid = (int)threadIdx.x) {
  constexpr int M = 1 << LGM, C = 1 << LGC, NP = (M >> 1) + 1, NyhP = mixed_rows(M + 1);
  cx<T>* tg = tile_base(g, x0, NyhP);
  for (int u = tid; u < C * NP; u += NT) {
    const int k = u >> LGC, k2 = M - k, c = u & (C - 1);
    const cx<T>* p = s + c * LD;
    if (k == 0) {
      const cx<T> z = p[0];
      handoff_store<T, wt_cols<T>(C, M)>(tg, tile_off<T, C>(0, c, x0, NyhP), mk<T>(z.x + z.y, 0)); handoff_store<T, wt_cols<T>(C, M)>(tg, tile_off<T, C>(M, c, x0, NyhP), mk<T>(z.x - z.y, 0));
    } else {
      const cx<T> a = p[pad(brevc<LGM>(k))], b = p[pad(brevc<LGM>(k2))];
      const cx<T> e = mk<T>(T(0.5) * (a.x + b.x), T(0.5) * (a.y - b.y)), o = mk<T>(T(0.5) * (a.x - b.x), T(0.5) * (a.y + b.y));
      const cx<T> wo = mul_mi(o * tw[k]);
      handoff_store<T, wt_cols<T>(C, M)>(tg, tile_off<T, C>(k, c, x0, NyhP), e + wo);
      if (k2 != k) handoff_store<T, wt_cols<T>(C, M)>(tg, tile_off<T, C>(k2, c, x0, NyhP), conj(e - wo));
    }
  }
}
// M-point inverse (input committed to bit-reversed slots and synchronised) -> the thread's packed pairs z[jj] = f[2jj] + i f[2jj+1],
// scaled.  Split tiles: sub-stages per half-column (WorkRows, no barriers) and the last level while reading:
//   z[jj] = u[jj] + conj(W_M^jj) v[jj],  z[jj + M/2] = u[jj] - conj(W_M^jj) v[jj]         (W_M^jj = tw[2 jj], tw = exp(-2 pi i k / N))
template <typename T, int R, int NT, int LGM, int LD>
__device__ __forceinline__ void mpt_inverse_read(cx<T>* s, const cx<T>* tw, T scale, cx<T> (&z)[R], int tid = (int)threadIdx.x) {
  using PM = PairMap<R, NT, LGM>;
  using V = typename vreg<T>::type;
  constexpr int M = 1 << LGM, MH = M >> 1, C = PM::C;
  if constexpr (PM::split) {
    fft_dit_w<T, LD, LGM, LGM + 1, 3, 1>(s, WorkRows<mpt_rt<NT, C>(), C>{1, C, tid}, tw);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R; i += 2) {
      const int e = PM::e(i, tid), c = e >> LGM, jj = e & (M - 1);
      const cx<T>* p = s + c * LD + pad(jj);
      const V u = vload(p), t = vmulc(vload(p + pad(MH)), vload(tw + 2 * jj));
      z[i] = vcx(vscale(vadd(u, t), scale)); z[i + 1] = vcx(vscale(vsub(u, t), scale));
    }
  } else {
    fft_dit<T, NT, LD, LGM, LGM + 1, CMBL_YLGM>(s, C, tw);
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int e = PM::e(i, tid), c = e >> LGM, jj = e & (M - 1);
      z[i] = scale * s[c * LD + pad(jj)];
    }
  }
}
// packed pairs z(i) of the thread -> M-point forward transform (bit-reversed slots, synchronised); the tile must be free
template <typename T, int R, int NT, int LGM, int LD, typename ZF>
__device__ __forceinline__ void mpt_write_forward(cx<T>* s, const cx<T>* tw, ZF&& zf, int tid = (int)threadIdx.x) {
  using PM = PairMap<R, NT, LGM>;
  using V = typename vreg<T>::type;
  constexpr int M = 1 << LGM, MH = M >> 1, C = PM::C;
  if constexpr (PM::split) {
#pragma unroll
    for (int i = 0; i < R; i += 2) {
      const int e = PM::e(i, tid), c = e >> LGM, jj = e & (M - 1);
      cx<T>* p = s + c * LD + pad(jj);
      const V za = vfrom(zf(i)), zb = vfrom(zf(i + 1));
      vstore(p, vadd(za, zb));
      vstore(p + pad(MH), vmul(vsub(za, zb), vload(tw + 2 * jj)));
    }
    __syncthreads();
    fft_dif_w<T, LD, LGM, LGM + 1, 3, 1>(s, WorkRows<mpt_rt<NT, C>(), C>{1, C, tid}, tw);
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int e = PM::e(i, tid), c = e >> LGM, jj = e & (M - 1);
      s[c * LD + pad(jj)] = zf(i);
    }
    __syncthreads();
    fft_dif<T, NT, LD, LGM, LGM + 1, CMBL_YLGM>(s, C, tw);
  }
}

// Two real sequences per complex transform.  Z = X + iY with X, Y the (Hermitian-extended) half spectra:
//   Z[k] = X[k] + i Y[k],  Z[N-k] = conj(X[k]) + i conj(Y[k])  (0<k<M);  Z[0], Z[M] from the real parts only (c2r semantics).
// After the N-point DIT the tile holds N*(x[n] + i y[n]).  Frequency k sits at slot pad(brev_N(k)).
// Here X = gX, Y = i*ly[k]*gY (the d/dy multiply rides along).
template <typename T, int NT, int LGN, int LGC> struct PairStage {
  static constexpr int N = 1 << LGN, M = N >> 1, C = 1 << LGC, TOT = C * (M + 1), K = (TOT + NT - 1) / NT;
  cx<T> X[K] = {}, Y[K] = {};
  T l[K] = {};
  __device__ __forceinline__ void issue(const cx<T>* __restrict__ gX, const cx<T>* __restrict__ gY, const T* __restrict__ ly, int Nx, int x0, int tid = (int)threadIdx.x) {
    const cx<T>* tX = tile_base(gX, x0, mixed_rows(M + 1));
    const cx<T>* tY = tile_base(gY, x0, mixed_rows(M + 1));
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int e = tid + i * NT;
      if (e < TOT) {
        const int k = e >> LGC;
        const unsigned gi = tile_off<T, C>(k, e & (C - 1), x0, mixed_rows(M + 1));
        X[i] = at32(tX, gi); Y[i] = at32(tY, gi); l[i] = at32(ly, (unsigned)k);
      }
    }
  }
  // the two tiles alone (l = ly[k] depends on the entry index only: a workgroup that walks several tiles keeps it from its first issue)
  __device__ __forceinline__ void issue_xy(const cx<T>* __restrict__ gX, const cx<T>* __restrict__ gY, int x0, int tid = (int)threadIdx.x) {
    const cx<T>* tX = tile_base(gX, x0, mixed_rows(M + 1));
    const cx<T>* tY = tile_base(gY, x0, mixed_rows(M + 1));
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int e = tid + i * NT;
      if (e < TOT) {
        const unsigned gi = tile_off<T, C>(e >> LGC, e & (C - 1), x0, mixed_rows(M + 1));
        X[i] = at32(tX, gi); Y[i] = at32(tY, gi);
      }
    }
  }
  template <int LD> __device__ __forceinline__ void commit(cx<T>* __restrict__ s, int tid = (int)threadIdx.x) const {
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int e = tid + i * NT;
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
// ---- row workgroups --------------------------------------------------------------------------------------------------------------
// A row workgroup takes RPW adjacent ky rows of one slice, all x; RT = 128 consecutive threads (two wavefronts) own a row.  The
// TOP radix-2 level of a row transform (span Nx/2) is done while the row moves between memory and LDS -- the thread that loads
// x[a] also loads x[a + Nx/2] and stores their sum and twiddled difference (forward), or combines the two halves on the way out
// (inverse) -- and each wavefront then runs the remaining levels on ITS half of the row without any barrier (WorkRows in
// fft_lds.hpp): 1024 points = 2 x 512 = 2 waves x 3 radix-8 stages of exactly one butterfly per lane.  (One wave per row with
// radix-16 stages measured 5.0k cycles per 1024-point transform -- a single wave's sequential instruction stream; lower radices
// with workgroup barriers were no faster.)
// The mixed-layout side is a gather / scatter of RPW x 4 values per x/4 (one 128-byte line for RPW = 4, fp32), the F-layout side
// contiguous rows.  Row r of the group sits at LDS offset r * row_ld(Nx); row_ld = 4 (mod 16) slots, so that the 16 lanes that
// write one gathered line (4 rows x 4 values) hit 16 different bank pairs.
#ifndef CMBL_ROW_ROT
#define CMBL_ROW_ROT(ky0) 0        // per-workgroup starting point of the x/4 walk, e.g. (((ky0) * 29) >> 2): measured, no effect
#endif
constexpr int ROW_RT = 128;                                              // threads per row
__host__ __device__ constexpr int row_nt(int rpw) { return ROW_RT * rpw; }   // workgroup size
__host__ __device__ constexpr int row_ld(int n) { return pad(n) + ((4 - pad(n) % 16) + 16) % 16; }
// rows per workgroup: as many of 4, 2, 1 as fit the 160 KB of LDS next to the twiddle table (NA row sets: 1, or 2 for the adjoint pass)
#ifndef CMBL_RPW_SMALL
#define CMBL_RPW_SMALL 4
#endif
#ifndef CMBL_RPW_BIG
#define CMBL_RPW_BIG 4        // Nx >= 1024; 2 measured slower for every row kernel in the timed mode (profiles/r04_probe_short_row_groups.txt)
#endif
#ifndef CMBL_XLG_SMALL
#define CMBL_XLG_SMALL 3      // Nx < 1024: radix-8 stages keep 64+ butterflies per stage for the two waves of a row (512²: step 2.58 -> 2.32 ms)
#endif
// Twiddle-table entries a row kernel keeps in LDS: the first half of the circle.  With the stage twiddles formed from ONE table read
// (stage_twiddles: index j * 2^sh < Nx / 2) and the fused top level (index < Nx / 2) nothing beyond it is read; the 4 KB saved at
// Nx = 1024 are 4 KB less to load per workgroup.
static_assert(CMBL_TW_REC == 2, "row kernels load half of the twiddle circle: needs the one-read stage twiddles in both precisions");
// 2048-point rows in double precision keep a QUARTER (8 KB instead of 16; the second quarter is -i times the first: tw_read, fft_lds.hpp).
// It was built so that two row groups of two rows (2 x 78 KB) could share a CU where one group of four rows (156 KB) is alone -- two
// out-of-phase workgroups per CU.  Measured (profiles/r05_ab_fp64_rows_2wg_rejected.txt): the d/dx pass does not change, the adjoint row
// pass on one-row groups is 7-9 % SLOWER, so the group heights stay (CMBL_F64_ROWS_2WG = 0); the smaller table stays as well (8 KB less to
// load per workgroup, results identical).
template <typename T> __host__ __device__ constexpr bool row_tw_quarter(int lgnx) { return sizeof(T) == 8 && lgnx == 11; }
template <typename T> __host__ __device__ constexpr int row_tw(int nx) { return row_tw_quarter<T>(ilog2c(nx)) ? nx >> 2 : nx >> 1; }
template <typename T, int LGNX> constexpr int row_qlg() { return row_tw_quarter<T>(LGNX) ? LGNX - 2 : 0; }
// fused radix-2 levels per stage of a row transform
__host__ __device__ constexpr int row_xlg(int lgnx) { return lgnx >= 10 ? CMBL_XLG : CMBL_XLG_SMALL; }
#ifndef CMBL_F64_ROWS_2WG
#define CMBL_F64_ROWS_2WG 0   // 1: quarter-table shapes take the tallest row group of which TWO fit a CU -- measured slower (profiles/r05_ab_fp64_rows_2wg_rejected.txt)
#endif
template <typename T> __host__ __device__ constexpr int row_rpw(int lgnx, int na) {
  const size_t budget = (CMBL_F64_ROWS_2WG && row_tw_quarter<T>(lgnx)) ? 80 * 1024 : 160 * 1024;
  for (int pass = 0; pass < 2; ++pass)                                   // second pass: nothing fits twice -> whatever fits once
    for (int rpw = (lgnx >= 10 ? CMBL_RPW_BIG : CMBL_RPW_SMALL); rpw >= 1; rpw >>= 1)
      if (((size_t)row_tw<T>(1 << lgnx) + (size_t)na * rpw * row_ld(1 << lgnx)) * sizeof(cx<T>) <= (pass == 0 ? budget : (size_t)160 * 1024)) return rpw;
  return 0;
}
// Row-group heights compiled BESIDES the LDS-fit maximum.  Below Nx = 1024 a launch over groups of four rows has fewer workgroups than
// the chip has CUs (512^2 QU: 2 x 65 = 130 on 256 CUs, 256^2: 66), and a launch lasts as long as one workgroup's chain whatever the
// number of idle CUs: heights 2 and 1 are compiled as well and the launch site takes the tallest group that still fills the chip
// (Ctx::pick_rpw; profiles/r05_ab_occupancy_tiles.txt).  From Nx = 1024 on the tallest group always fills it.
__host__ __device__ constexpr bool row_rpw_variants(int lgnx) { return lgnx < 10; }
// Block -> (slice, first row).  nblk = row groups of the launch (slices * ceil(Nyh / RPW)).  The full groups of all slices come
// first and the short ones (Nyh = Ny/2 + 1 leaves one Nyquist row per slice) LAST: the launches are one residency wave, so the
// blocks beyond the CU count share a CU with an earlier block -- a one-row group there costs its host little, a second full group
// would slow both down by the ratio of their VALU work (measured with in-kernel stamps: 6.3k instead of 3.9k cycles per transform
// chain, and the launch ends with its slowest workgroup).
struct RowGroup { int sl, ky0, nr; };
template <int RPW> __device__ __forceinline__ RowGroup row_group(unsigned blk, int Nyh, unsigned nblk) {      // 32-bit: one short division sequence
  const unsigned G = (unsigned)(Nyh + RPW - 1) / RPW, Gf = (unsigned)Nyh / RPW, rem = (unsigned)Nyh - Gf * RPW;
  const unsigned slices = nblk / G, nfull = slices * Gf;
  RowGroup g;
  if (blk < nfull) { const unsigned sl = blk / Gf; g.sl = (int)sl; g.ky0 = (int)(blk - sl * Gf) * RPW; g.nr = RPW; }
  else { g.sl = (int)(blk - nfull); g.ky0 = (int)Gf * RPW; g.nr = (int)rem; }
  return g;
}
// 16-byte global accesses: 2 single-precision or 1 double-precision complex values
template <typename T> struct alignas(16) CxVec { cx<T> v[16 / sizeof(cx<T>)]; };
template <typename T> __device__ __forceinline__ const CxVec<T>& vec32(const cx<T>* base, unsigned idx) {
  return *reinterpret_cast<const CxVec<T>*>(reinterpret_cast<const char*>(base) + idx * (unsigned)sizeof(cx<T>));
}
// 16-byte load into a register-staged array element.  With one value per vector (double precision) the element is assigned as a
// value: the aggregate copy kept the whole array in scratch memory, with a full wait after every load.
template <typename T> __device__ __forceinline__ void vload32(CxVec<T>& dst, const cx<T>* base, unsigned idx) {
  if constexpr (sizeof(cx<T>) == 16) dst.v[0] = at32(base, idx);
  else dst = vec32(base, idx);
}
template <typename T> __device__ __forceinline__ CxVec<T>& vec32(cx<T>* base, unsigned idx) {
  return *reinterpret_cast<CxVec<T>*>(reinterpret_cast<char*>(base) + idx * (unsigned)sizeof(cx<T>));
}

// mixed layout (slice bases g[a]) -> the NA row sets in LDS, through the top DIF level:
//   s[a] <- x[a] + x[a + N/2],   s[a + N/2] <- (x[a] - x[a + N/2]) * W_N^a        (twg: the global twiddle table exp(-2 pi i k / N))
// All loads of the thread are issued before the first LDS store.
// (CMBL_ROW_ROT lets every workgroup walk x/4 from its own starting point, so that the workgroups of a launch do not touch the same
// 16 KB window at the same moment; measured on MI355X: no difference, the default is no rotation)
template <typename T, int LGNX, int RPW, int NA> struct RowsMixedStage {
  using V = typename vreg<T>::type;
  static constexpr int Nx = 1 << LGNX, NH = Nx >> 1, LD = row_ld(Nx), VE = 16 / (int)sizeof(cx<T>), UPG = mixw<T>() / VE, NT = row_nt(RPW);
  static constexpr int TOT = RPW * NH / VE, K = (TOT + NT - 1) / NT;
  CxVec<T> va[NA][K] = {}, vb[NA][K] = {}, w[K] = {};
  // tid: index of the thread among the NT that share this load (threadIdx.x unless several row sets are loaded side by side)
  __device__ __forceinline__ void issue(const cx<T>* const (&g)[NA], const cx<T>* __restrict__ twg, int NyhP, int ky0, int nr, int tid = threadIdx.x) {
    const int rot = CMBL_ROW_ROT(ky0);
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int u = tid + i * NT, xt = (u / (UPG * RPW) + rot) & (NH / mixw<T>() - 1), r = (u / UPG) % RPW, c = (u % UPG) * VE;
      if ((TOT % NT == 0 || u < TOT) && r < nr) {
        vload32(w[i], twg, (unsigned)(xt * mixw<T>() + c));
        const unsigned o = (unsigned)((xt * NyhP + r) * mixw<T>() + c), ob = (unsigned)(((xt + NH / mixw<T>()) * NyhP + r) * mixw<T>() + c);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
          const cx<T>* ga = g[a] + (size_t)ky0 * mixw<T>();                    // uniform part of the address
          vload32(va[a][i], ga, o);
          vload32(vb[a][i], ga, ob);
        }
      }
    }
  }
  // row set a -> the LDS rows at s
  __device__ __forceinline__ void commit(int a, cx<T>* __restrict__ s, int ky0, int nr, int tid = threadIdx.x) const {
    const int rot = CMBL_ROW_ROT(ky0);
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int u = tid + i * NT, xt = (u / (UPG * RPW) + rot) & (NH / mixw<T>() - 1), r = (u / UPG) % RPW, c = (u % UPG) * VE;
      if ((TOT % NT == 0 || u < TOT) && r < nr) {
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const V xa = vfrom(va[a][i].v[e]), xb = vfrom(vb[a][i].v[e]);
          cx<T>* p = s + r * LD + pad(xt * mixw<T>() + c) + e;                 // c even: pad(x + 1) == pad(x) + 1; pad(x + N/2) == pad(x) + pad(N/2)
          vstore(p, vadd(xa, xb));
          vstore(p + pad(NH), vmul(vsub(xa, xb), vfrom(w[i].v[e])));
        }
      }
    }
  }
};
template <typename T, int LGNX, int RPW, int NA>
__device__ __forceinline__ void rows_load_mixed_dif(cx<T>* const (&s)[NA], const cx<T>* const (&g)[NA], const cx<T>* __restrict__ twg, int NyhP, int ky0, int nr, int tid = threadIdx.x) {
  RowsMixedStage<T, LGNX, RPW, NA> st;
  st.issue(g, twg, NyhP, ky0, nr, tid);
#pragma unroll
  for (int a = 0; a < NA; ++a) st.commit(a, s[a], ky0, nr, tid);
}
// LDS rows -> mixed layout through the last DIT level:  x[a] = u[a] + conj(W_N^a) v[a],  x[a + N/2] = u[a] - conj(W_N^a) v[a]  with
// u, v the two halves of the row in LDS; values scaled by `scale`.  tw: the LDS twiddle table.
template <typename T, int LGNX, int RPW>
__device__ __forceinline__ void rows_store_mixed_dit(const cx<T>* __restrict__ s, cx<T>* __restrict__ g, const cx<T>* __restrict__ tw, int NyhP, int ky0, int nr, T scale,
                                                     int nyq = -1 /* >= 0: drop Im of the ky = 0 and ky = nyq rows (what c2r ignores) */, int tid = threadIdx.x) {
  using V = typename vreg<T>::type;
  constexpr int Nx = 1 << LGNX, NH = Nx >> 1, LD = row_ld(Nx), VE = 16 / (int)sizeof(cx<T>), UPG = mixw<T>() / VE, NT = row_nt(RPW), TOT = RPW * NH / VE;
  const int rot = CMBL_ROW_ROT(ky0);
  for (int u = tid; u < TOT; u += NT) {
    const int xt = (u / (UPG * RPW) + rot) & (NH / mixw<T>() - 1), r = (u / UPG) % RPW, c = (u % UPG) * VE;
    if (r < nr) {
      CxVec<T> oa, ob;
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const cx<T>* p = s + r * LD + pad(xt * mixw<T>() + c) + e;
        const V uu = vload(p), t = vmulc(vload(p + pad(NH)), tw_read<T, row_qlg<T, LGNX>()>(tw, xt * mixw<T>() + c + e));
        oa.v[e] = vcx(vscale(vadd(uu, t), scale)); ob.v[e] = vcx(vscale(vsub(uu, t), scale));
        if (nyq >= 0 && (ky0 + r == 0 || ky0 + r == nyq)) { oa.v[e].y = T(0); ob.v[e].y = T(0); }
      }
      cx<T>* gk = g + (size_t)ky0 * mixw<T>();                                  // uniform part of the address
      const unsigned ia = (unsigned)((xt * NyhP + r) * mixw<T>() + c), ib = (unsigned)(((xt + NH / mixw<T>()) * NyhP + r) * mixw<T>() + c);
      if constexpr (wt_line<T>(Nx)) {                                   // hand-off: see handoff_store
        store_wt<16>(reinterpret_cast<char*>(gk) + ia * (unsigned)sizeof(cx<T>), &oa);
        store_wt<16>(reinterpret_cast<char*>(gk) + ib * (unsigned)sizeof(cx<T>), &ob);
      } else { vec32(gk, ia) = oa; vec32(gk, ib) = ob; }
    }
  }
}
// F layout: contiguous rows (g = first row of the group).  The F side of a transform is its bit-reversed end: no level is fused here.
template <typename T, int LGNX, int RPW>
__device__ __forceinline__ void rows_load_F(cx<T>* __restrict__ s, const cx<T>* __restrict__ g, int nr, int tid = threadIdx.x) {
  constexpr int Nx = 1 << LGNX, LD = row_ld(Nx), VE = 16 / (int)sizeof(cx<T>), NT = row_nt(RPW), TOT = RPW * Nx / VE, K = (TOT + NT - 1) / NT;
  // every thread loads unconditionally (rows beyond nr re-read the group's last row, entries beyond TOT the first): an array written
  // under a predicate stays in scratch memory in the double-precision instantiations, with a full wait after each load
  CxVec<T> v[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int u0 = tid + i * NT, u = (TOT % NT == 0 || u0 < TOT) ? u0 : 0, r = (u * VE) >> LGNX, x = (u * VE) & (Nx - 1);
    v[i] = vec32(g, (unsigned)(((r < nr ? r : nr - 1) << LGNX) + x));
  }
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int u = tid + i * NT, r = (u * VE) >> LGNX, x = (u * VE) & (Nx - 1);
    if ((TOT % NT == 0 || u < TOT) && r < nr) {
#pragma unroll
      for (int e = 0; e < VE; ++e) s[r * LD + pad(x) + e] = v[i].v[e];
    }
  }
}
template <typename T, int LGNX, int RPW>
__device__ __forceinline__ void rows_store_F(const cx<T>* __restrict__ s, cx<T>* __restrict__ g, int nr, T scale) {
  constexpr int Nx = 1 << LGNX, LD = row_ld(Nx), VE = 16 / (int)sizeof(cx<T>), NT = row_nt(RPW), TOT = RPW * Nx / VE;
  for (int u = threadIdx.x; u < TOT; u += NT) {
    const int r = (u * VE) >> LGNX, x = (u * VE) & (Nx - 1);
    if (r < nr) {
      CxVec<T> v;
#pragma unroll
      for (int e = 0; e < VE; ++e) v.v[e] = scale * s[r * LD + pad(x) + e];
      vec32(g, (unsigned)(u * VE)) = v;
    }
  }
}
// i * lx * v for the value at slot b0 + j of a bit-reversed row (kx = bitrev(slot), lx = dl * signed kx); b0 = first slot of a
// butterfly of the last stage (low LG bits zero), j < 2^LG a compile-time constant after unrolling.  bitrev(b0 + j) = bitrev(b0) +
// bitrev(j) without carries, and bitrev(b0) < Nx / 2^LG: one bit reversal and one conversion per butterfly, an exact float addition
// of a constant and the multiply per slot (the same lx, bit for bit, as dl * float(signed kx)).
template <typename T, int LGNX, int LG> __device__ __forceinline__ cx<T> mul_il_slot(cx<T> v, int b0, int j, T dl) {
  constexpr int Nx = 1 << LGNX;
  const T k0 = T(brevc<LGNX>(b0));
  const int cj = brevc<LG>(j) << (LGNX - LG);
  const T l = dl * (k0 + T(cj < (Nx >> 1) ? cj : cj - Nx));
  return mk<T>(-l * v.y, l * v.x);
}
// After the N-point DIF of a + i b (a, b real): A[k] = (Z[k] + conj Z[N-k])/2, B[k] = (Z[k] - conj Z[N-k])/(2i), k = 0..M.
// f(k, c, A, B) consumes the pair (stores it, or combines it with something held in registers).
template <typename T, int NT, int LD, int LGN, int LGC, int RZ, typename F>
__device__ __forceinline__ void pair_split(const cx<T>* __restrict__ s, F&& f, int tid = (int)threadIdx.x) {
  constexpr int N = 1 << LGN, M = N >> 1, C = 1 << LGC;
#pragma unroll
  for (int i = 0; i < RZ; ++i) {                    // RZ = ceil(C*(M+1)/NT): compile-time trip count keeps f's captures in registers
    const int e = tid + i * NT;
    if (e < C * (M + 1)) {
      const int c = e & (C - 1), k = e >> LGC;
      const cx<T>* p = s + c * LD;
      const cx<T> zk = p[pad(brevc<LGN>(k))], zn = p[pad(brevc<LGN>((N - k) & (N - 1)))];
      f(i, k, c, mk<T>(T(0.5) * (zk.x + zn.x), T(0.5) * (zk.y - zn.y)), mk<T>(T(0.5) * (zk.y + zn.y), T(0.5) * (zn.x - zk.x)));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// y pass, forward: map -> mixed.   grid (Nx/C, slices).  LDS: twY[M] (half the circle: all that the one-read stage twiddles and the fused levels touch) + C*tile_ld(M) cplx
template <typename T, int R, int NT, int LGM>
__global__ __launch_bounds__(NT) void k_y_r2c(const T* __restrict__ in, cx<T>* __restrict__ out, const cx<T>* __restrict__ twY, int Nx) {
  using G = ColTile<R, NT, LGM>;
  constexpr int M = G::M, LD = G::LDM, C = G::C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;                            // half of the twiddle circle (see row_tw)
  const int x0 = xcd_tile(blockIdx.x, gridDim.x) * C;
  const size_t sl = blockIdx.y;
  TwStage<T, NT, M> twr;
  twr.issue(twY);
  using PM = PairMap<R, NT, LGM>;
  const cx<T>* src = reinterpret_cast<const cx<T>*>(in) + (sl * Nx + x0) * (size_t)M;
  cx<T> v[R];
#pragma unroll
  for (int i = 0; i < R; ++i) v[i] = at32(src, (unsigned)PM::e(i));
  twr.commit(tw);
  __syncthreads();
  mpt_write_forward<T, R, NT, LGM, LD>(s, tw, [&](int i) { return v[i]; });
  half_store<T, NT, LD, LGM, G::LGC>(s, out + sl * (size_t)mixed_rows(G::Nyh) * Nx, tw, x0);
}

// y pass, inverse: mixed -> map, scaled by `scale` (1/Ny; the x pass already carries 1/Nx)
template <typename T, int R, int NT, int LGM>
__global__ __launch_bounds__(NT) void k_y_c2r(const cx<T>* __restrict__ in, T* __restrict__ out, const cx<T>* __restrict__ twY, int Nx, T scale) {
  using G = ColTile<R, NT, LGM>;
  constexpr int M = G::M, LD = G::LDM, C = G::C;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;                            // half of the twiddle circle (see row_tw)
  const int x0 = xcd_tile(blockIdx.x, gridDim.x) * C;
  const size_t sl = blockIdx.y;
  using PM = PairMap<R, NT, LGM>;
  TwStage<T, NT, M> twr;
  HalfStage<T, NT, LGM, G::LGC> tl;
  twr.issue(twY);
  tl.issue(in + sl * (size_t)mixed_rows(G::Nyh) * Nx, twY, x0);
  twr.commit(tw);
  tl.template commit<LD>(s);
  __syncthreads();
  cx<T> z[R];
  mpt_inverse_read<T, R, NT, LGM, LD>(s, tw, scale, z);
  cx<T>* dst = reinterpret_cast<cx<T>*>(out) + (sl * Nx + x0) * (size_t)M;
#pragma unroll
  for (int i = 0; i < R; ++i) at32(dst, (unsigned)PM::e(i)) = z[i];
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
  cx<T>* s = tw + M;                            // half of the twiddle circle (see row_tw)
  const int x0 = xcd_tile(blockIdx.x, gridDim.x) * C;
  const size_t sl = blockIdx.y;
  using PM = PairMap<R, NT, LGM>;
  TwStage<T, NT, M> twr;
  HalfStage<T, NT, LGM, G::LGC> tl;
  twr.issue(twY);
  tl.issue(in + sl * (size_t)mixed_rows(G::Nyh) * Nx, twY, x0);
  const cx<T>* mk2 = reinterpret_cast<const cx<T>*>(mask) + (size_t)x0 * M;       // the mask is one (Nx, Ny) map for all slices
  cx<T> mv[R];
#pragma unroll
  for (int i = 0; i < R; ++i) mv[i] = at32(mk2, (unsigned)PM::e(i));
  twr.commit(tw);
  tl.template commit<LD>(s);
  __syncthreads();
  cx<T> z[R];
  mpt_inverse_read<T, R, NT, LGM, LD>(s, tw, scale, z);
  __syncthreads();
  mpt_write_forward<T, R, NT, LGM, LD>(s, tw, [&](int i) { return mk<T>(mv[i].x * z[i].x, mv[i].y * z[i].y); });
  half_store<T, NT, LD, LGM, G::LGC>(s, out + sl * (size_t)mixed_rows(G::Nyh) * Nx, tw, x0);
}

// ---------------------------------------------------------------------------------------------
// x pass.  grid = slices * ceil(Nyh / RPW) row groups, 128 * RPW threads.  LDS: twX[Nx] + RPW * row_ld(Nx) cplx
//   MODE 0: forward  (mixed -> F)
//   MODE 1: inverse  (F -> mixed), scaled by 1/Nx
//   MODE 2: x-derivative  (mixed -> mixed):  ifft_x( i*lx * fft_x(row) ) / Nx        (src/proj_lambert.jl:146-159, coord 1)
template <typename T, int MODE, int LGNX, int RPW>
__global__ __launch_bounds__(row_nt(RPW), row_min_waves<T>()) void k_x_fft(const cx<T>* __restrict__ in, cx<T>* __restrict__ out,
                                                                          const cx<T>* __restrict__ twX, T dlx_over_Nx, int Nyh) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int Nx = 1 << LGNX, LD = row_ld(Nx), NT = row_nt(RPW);
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + row_tw<T>(Nx);
  const RowGroup rg = row_group<RPW>(blockIdx.x, Nyh, gridDim.x);
  const int NyhP = mixed_rows(Nyh);
  const size_t mo = (size_t)rg.sl * NyhP * Nx, fo = ((size_t)rg.sl * Nyh + rg.ky0) * Nx;
  CMBL_XWSTAMP(14);
  CMBL_XSTAMP(0);
  TwStage<T, NT, row_tw<T>(Nx)> twr;
  twr.issue(twX);
  if (MODE == 1) rows_load_F<T, LGNX, RPW>(s, in + fo, rg.nr);
  else {
    cx<T>* const sa[1] = {s};
    const cx<T>* const ga[1] = {in + mo};
    rows_load_mixed_dif<T, LGNX, RPW, 1>(sa, ga, twX, NyhP, rg.ky0, rg.nr);
  }
  twr.commit(tw);
  CMBL_XSTAMP(6);
  __syncthreads();
  CMBL_XSTAMP(1);
  CMBL_WVSTAMP(0);
  const WorkRows<ROW_RT, RPW, row_tw_quarter<T>(LGNX)> wk{1, rg.nr};
  if (MODE == 0) fft_dif_w<T, LD, LGNX, LGNX, row_xlg(LGNX), 1>(s, wk, tw);
  CMBL_XSTAMP(2);
  if (MODE == 2) {
    // i*lx/Nx multiply between the last forward and the first inverse butterfly, in registers: slot i holds kx = bitrev(i), lx = dlx * signed(kx)
    const T dl = dlx_over_Nx;
    fft_dif_mid_dit_w<T, LD, LGNX, LGNX, row_xlg(LGNX), 1>(s, wk, tw, [dl](int, int b0, int j, typename vreg<T>::type v) {
      constexpr int LGL = stage_lg(LGNX - 1, num_stages(LGNX - 1, row_xlg(LGNX)) - 1, row_xlg(LGNX));
      return vfrom(mul_il_slot<T, LGNX, LGL>(vcx(v), b0, j, dl));
    });
  }
  if (MODE == 1) fft_dit_w<T, LD, LGNX, LGNX, row_xlg(LGNX), 1>(s, wk, tw);
  CMBL_XSTAMP(5);
  CMBL_WVSTAMP(1);
  __syncthreads();
  CMBL_XSTAMP(3);
  if (MODE == 0) rows_store_F<T, LGNX, RPW>(s, out + fo, rg.nr, T(1));
  else rows_store_mixed_dit<T, LGNX, RPW>(s, out + mo, tw, NyhP, rg.ky0, rg.nr, MODE == 1 ? T(1) / T(Nx) : T(1));
  CMBL_XSTAMP(4);
  CMBL_XWSTAMP(15);
}

}  // namespace cmbl
