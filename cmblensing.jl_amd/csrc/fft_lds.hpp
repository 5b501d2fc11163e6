// In-LDS FFT building blocks (workgroup-cooperative, NT threads, compile-time sizes).
//
// Conventions (index math prototyped in tools/fft_proto.py):
//   forward  = e^{-i}, in-place radix-2 DIF network : natural order in  -> bit-reversed order out
//   inverse  = e^{+i}, in-place radix-2 DIT network : bit-reversed in   -> natural order out (unnormalised)
// so a forward/pointwise/inverse chain never needs a reordering pass: frequency k lives at slot brev(k).
// Up to four radix-2 levels are fused per LDS round trip: a thread pulls 2^LG elements into registers, runs LG levels of
// the in-place network on them (radix-16 for LG=4) and writes them back, so a 1024-point transform is 3 round trips / 3
// barriers.  Element i of a sequence lives at LDS index pad(i) = i + (i >> CMBL_PAD_SHIFT): the padding spreads the
// power-of-two strides of bit-reversed and butterfly accesses over the 64 banks.  All sizes are template parameters, so
// every LDS address inside a stage is `pad(base) + immediate` and every twiddle index is a constant shift.
//
// Real data, two flavours:
//   * single: a length-N real sequence as an N/2-point complex transform (r2c_post / c2r_pre); half spectrum A[0..M] at
//     slots brev(k) (k<M) and M.
//   * pair:   two real sequences a, b as ONE N-point complex transform of a + i b (pair_* helpers in kernels_fft.hpp).
// c2r drops Im A[0] and Im A[M] exactly like FFTW / pocketfft / cuFFT do (the src/util_fft.jl:21-25 path relies on it:
// the reference feeds irfft non-Hermitian input, src/proj_lambert.jl:63-64).
//
// tw[] is an LDS table exp(-2*pi*i*k/Ntw), k < Ntw (Ntw = 2^LGNTW, the FULL circle: the external twiddles W_n^(jk) of a radix-16
// stage run over nearly all of it); a transform of length n <= Ntw uses stride Ntw/n.
#pragma once
#include "common.hpp"
#include "fft_core.hpp"

#ifndef CMBL_STAGE_SYNC
#define CMBL_STAGE_SYNC() __syncthreads()
#endif

namespace cmbl {

// One spare slot per 2^CMBL_PAD_SHIFT.  A third of the LDS cycles of the fused kernels are bank conflicts (SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE = 0.32, all on the ds_read_b64 side: 32-lane groups on 64 banks; the writes stay inside their transfer time).
// A spare slot per 8 removes the read conflicts of the two lower stages of a wave-private 512-point transform (bank model of the
// stage accesses: 96 -> 64 LDS cycles per transform, 48 ideal) -- and changes no kernel time on the GPU (A/B at 1024^2), so the
// smaller footprint stays: the conflict cycles hide behind the waits of the dependent chain.
#ifndef CMBL_PAD_SHIFT
#define CMBL_PAD_SHIFT 4
#endif
__device__ __host__ __forceinline__ constexpr int pad(int i) { return i + (i >> CMBL_PAD_SHIFT); }
// leading dimension (in complex slots) of a tile row holding n elements (+1 spare slot for the packed-real Nyquist term)
__device__ __host__ __forceinline__ constexpr int tile_ld(int n) { return pad(n) + 1; }
template <int LG> __device__ __forceinline__ int brevc(int i) { return LG == 0 ? 0 : (int)(__brev((unsigned)i) >> (32 - (LG == 0 ? 1 : LG))); }

template <typename T, int NT>
__device__ __forceinline__ void load_twiddles(cx<T>* tw_lds, const cx<T>* __restrict__ tw_g, int nhalf) {
  for (int i = threadIdx.x; i < nhalf; i += NT) tw_lds[i] = tw_g[i];
}

// ---- stage schedule: levels per stage as even as possible over ceil(lgN/4) stages (9 -> 3,3,3 ; 10 -> 4,3,3 ; 5 -> 3,2)
// MAXLG caps the radix (2^MAXLG): 4 = radix-16 (fewest barriers), 3 = radix-8 (half the registers, one more round trip).
constexpr int stage_levels(int remaining, int maxlg = 4) { return (remaining + ((remaining + maxlg - 1) / maxlg) - 1) / ((remaining + maxlg - 1) / maxlg); }
constexpr int num_stages(int lgN, int maxlg = 4) { int n = 0; while (lgN > 0) { lgN -= stage_levels(lgN, maxlg); ++n; } return n; }
constexpr int stage_lg(int lgN, int idx, int maxlg = 4) { int lg = 0; for (int i = 0; i <= idx; ++i) { lg = stage_levels(lgN, maxlg); lgN -= lg; } return lg; }
constexpr int levels_after(int lgN, int idx, int maxlg = 4) { int tot = 0; for (int i = 0; i <= idx; ++i) tot += stage_lg(lgN, i, maxlg); return lgN - tot; }

// ---- who does what in a stage ------------------------------------------------------------------------------------------------
// WorkCoop: the NT threads of the workgroup share S sequences (column tiles); stages are separated by workgroup barriers.
template <int NT> struct WorkCoop {
  int S;
  static constexpr bool twq = false;
  template <int LGNB, typename F> __device__ __forceinline__ void each(F&& f) const {
    for (int q = threadIdx.x; q < (S << LGNB); q += NT) f(q >> LGNB, q & ((1 << LGNB) - 1));
  }
  __device__ __forceinline__ void sync() const { CMBL_STAGE_SYNC(); }
};
// WorkRows: RT consecutive threads own row `threadIdx.x / RT` of each of NA arrays (sequence a*RPW + row); rows >= nr are absent.
// The row kernels run the TOP radix-2 level of a row transform while loading / storing (kernels_fft.hpp), so the stages here act on
// the two halves of a row independently.  With RT = 128 wave w of a row takes the items of half w in every stage (items
// [w*I/2, (w+1)*I/2) of the I items of a stage are exactly those that touch half w), with RT = 64 one wave has the whole row: either
// way a wave only ever reads LDS slots it wrote itself, LDS operations of a wave execute in order, and no barrier is needed
// between stages.  Other RT fall back to workgroup barriers.
// TWQ: the twiddle table in LDS holds a QUARTER of the circle (row kernels of 2048-point double-precision rows: row_tw_quarter), the second
// quarter is -i times the first (stage_twiddles)
template <int RT, int RPW, bool TWQ = false> struct WorkRows {
  int NA, nr;
  int tid = (int)threadIdx.x;             // (a member so that a kernel that walks several tiles can pass an opaque copy per tile: kernels_flow.hpp delta_y_body_pipelined)
  static constexpr bool twq = TWQ;
  static constexpr bool wave_private = RT <= 128;
  template <int LGNB, typename F> __device__ __forceinline__ void each(F&& f) const {
    const int row = tid / RT, t = tid % RT;
    if (row >= nr) return;
    if constexpr (RT == 128) {
      constexpr int I = 1 << LGNB, HI = I >> 1;                       // items of the stage, items per half
      const int w = t >> 6, lane = t & 63;
      for (int a = 0; a < NA; ++a) {
        if constexpr (HI >= 64) {
#pragma unroll
          for (int i = 0; i < HI / 64; ++i) f(a * RPW + row, w * HI + lane + 64 * i);
        } else if constexpr (HI >= 1) {
          if (lane < HI) f(a * RPW + row, w * HI + lane);
        } else {
          if (t == 0) f(a * RPW + row, 0);
        }
      }
    } else {
      for (int a = 0; a < NA; ++a)
        for (int rr = t; rr < (1 << LGNB); rr += RT) f(a * RPW + row, rr);
    }
  }
  __device__ __forceinline__ void sync() const {
    if constexpr (wave_private) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
  }
};

// WorkSeqs: like WorkRows with every row set on its own threads -- RT consecutive threads own sequence `threadIdx.x / RT` of NA * RPW
// sequences (sequence a*RPW + row; rows >= nr absent).  Used by the row carriers that hold all pol slices of a batch slot.
template <int RT, int RPW, bool TWQ = false> struct WorkSeqs {
  int nr;
  static constexpr bool twq = TWQ;
  static constexpr bool wave_private = RT <= 128;
  template <int LGNB, typename F> __device__ __forceinline__ void each(F&& f) const {
    static_assert(RT == 128, "WorkSeqs: two wavefronts per sequence");
    const int seq = threadIdx.x / RT, t = threadIdx.x % RT;
    if (seq % RPW >= nr) return;
    constexpr int I = 1 << LGNB, HI = I >> 1;
    const int w = t >> 6, lane = t & 63;
    if constexpr (HI >= 64) {
#pragma unroll
      for (int i = 0; i < HI / 64; ++i) f(seq, w * HI + lane + 64 * i);
    } else if constexpr (HI >= 1) {
      if (lane < HI) f(seq, w * HI + lane);
    } else {
      if (t == 0) f(seq, 0);
    }
  }
  __device__ __forceinline__ void sync() const { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
};

// External twiddles of a stage, W^(j k) for k = 1..r-1.  CMBL_TW_REC = 1 (single precision only): ONE table read, W^j, and the powers by
// multiplication (depth-3 product tree) instead of r-1 table reads -- the transforms are bound by LDS throughput at the CU and the
// twiddle reads are a quarter of a radix-8 stage's LDS traffic; the products cost 2 packed instructions each on a VALU that has room.
#ifndef CMBL_TW_REC
#define CMBL_TW_REC 2
#endif
// table entry i < 2^(QLG+1) of a table that keeps the first 2^QLG entries (a quarter of the circle): W^(i + N/4) = -i W^i.  QLG = 0: plain read
template <typename T, int QLG> __device__ __forceinline__ typename vreg<T>::type tw_read(const cx<T>* __restrict__ tw, int i) {
  if constexpr (QLG == 0) return vload(tw + i);
  else {
    const typename vreg<T>::type w = vload(tw + (i & ((1 << QLG) - 1)));
    return (i >> QLG) ? vmul_mi(w) : w;
  }
}
template <typename T, int r, int QLG = 0, typename V>
__device__ __forceinline__ void stage_twiddles(const cx<T>* __restrict__ tw, int j, int sh, V (&w)[r]) {
  if constexpr (QLG != 0) {
    static_assert(CMBL_TW_REC == 2, "quarter-circle tables need the one-read stage twiddles");
    w[1] = tw_read<T, QLG>(tw, j << sh);
#pragma unroll
    for (int k = 2; k < r; ++k) w[k] = vmul(w[k >> 1], w[k - (k >> 1)]);
  } else if constexpr ((CMBL_TW_REC == 2 || (CMBL_TW_REC == 1 && sizeof(T) == 4)) && r >= 4) {
    w[1] = vload(tw + (j << sh));
#pragma unroll
    for (int k = 2; k < r; ++k) w[k] = vmul(w[k >> 1], w[k - (k >> 1)]);
  } else {
#pragma unroll
    for (int k = 1; k < r; ++k) w[k] = vload(tw + ((j * k) << sh));
  }
}

// One fused DIF stage = LG radix-2 levels with spans h = 2^LGH (top) ... hmin = 2^(LGH-LG+1), evaluated as ONE r-point DFT per
// thread (r = 2^LG, fft_core.hpp) followed by the external twiddles: with a[m] = x[b0 + m hmin], n = r hmin,
//     x[b0 + brev_LG(k) hmin] <- W_n^(j k) * sum_m a[m] W_r^(m k),     W_n = exp(-2 pi i / n),  j = b0 mod hmin
// (tools/fft_proto.py: identical to the level-by-level network).  tw[] covers the full circle, so W_n^(jk) is one table read.
template <typename T, int LD, int LGN, int LGNTW, int LGH, int LG, typename W>
__device__ __forceinline__ void dif_stage(cx<T>* __restrict__ s, const W& wk, const cx<T>* __restrict__ tw) {
  using V = typename vreg<T>::type;
  constexpr int r = 1 << LG, lghmin = LGH - LG + 1, hmin = 1 << lghmin, lgnb = LGN - LG, sh = LGNTW - (LGH + 1);
  wk.template each<lgnb>([&](int seq, int rr) {
    const int blk = rr >> lghmin, j = rr & (hmin - 1);
    cx<T>* p = s + seq * LD + pad((blk << (LGH + 1)) + j);
    V v[r];
#pragma unroll
    for (int m = 0; m < r; ++m) v[m] = vload(p + pad(m << lghmin));     // pad(base + m*hmin) == pad(base) + pad(m*hmin) here
    V w[r];
    if constexpr (hmin > 1) stage_twiddles<T, r, (W::twq ? LGNTW - 2 : 0)>(tw, j, sh, w);
    dft<T, LG, false>(v);
#pragma unroll
    for (int k = 0; k < r; ++k) {
      V x = v[dft_loc<LG>(k)];
      if (hmin > 1 && k > 0) x = vmul(x, w[k]);
      vstore(p + pad(brevc<LG>(k) << lghmin), x);
    }
  });
  wk.sync();
}

// One fused DIT stage = LG levels with spans hmin = 2^LGH (bottom) ... hmin * 2^(LG-1): the transpose of the above,
//     x[b0 + m hmin] <- sum_k conj(W_r^(m k)) * conj(W_n^(j k)) * x[b0 + brev_LG(k) hmin]
struct NoPre { template <typename V> __device__ __forceinline__ V operator()(V v, int) const { return v; } };

template <typename T, int LD, int LGN, int LGNTW, int LGH, int LG, typename W, typename PRE = NoPre>
__device__ __forceinline__ void dit_stage(cx<T>* __restrict__ s, const W& wk, const cx<T>* __restrict__ tw, PRE pre = PRE()) {
  using V = typename vreg<T>::type;
  constexpr int r = 1 << LG, hmin = 1 << LGH, lgnb = LGN - LG, sh = LGNTW - (LGH + LG);
  wk.template each<lgnb>([&](int seq, int rr) {
    const int blk = rr >> LGH, j = rr & (hmin - 1);
    const int b0 = (blk << (LGH + LG)) + j;                          // logical (unpadded) index of element m = 0
    cx<T>* p = s + seq * LD + pad(b0);
    V v[r], w[r];
    if constexpr (hmin > 1) stage_twiddles<T, r, (W::twq ? LGNTW - 2 : 0)>(tw, j, sh, w);
#pragma unroll
    for (int k = 0; k < r; ++k) {                                    // frequency k of the group sits at position brev(k)
      const int m = brevc<LG>(k);
      V x = vfrom(pre(*(p + pad(m << LGH)), b0 + (m << LGH)));       // pre: pointwise op fused into the first stage
      if (hmin > 1 && k > 0) x = vmulc(x, w[k]);
      v[k] = x;
    }
    dft<T, LG, true>(v);
#pragma unroll
    for (int m = 0; m < r; ++m) vstore(p + pad(m << LGH), v[dft_loc<LG>(m)]);
  });
  wk.sync();
}

// Last forward stage + pointwise operation + first inverse stage in ONE LDS round trip.  The last DIF stage (spans 2^(LG-1) .. 1) and
// the first DIT stage act on the same 2^LG adjacent slots, so a forward / multiply / inverse chain keeps them in registers:
//     v = DFT_r(slots b0 .. b0+r-1);  X_k (at slot b0 + j, j = brev(k)) <- mid(seq, b0, j, X_k);  slots <- IDFT_r(X)
// (one stage's LDS traffic and one stage's latency less per chain; LG = levels of the LAST stage of the schedule).
template <typename T, int LD, int LGN, int LG, typename W, typename MID>
__device__ __forceinline__ void dif_mid_dit_stage(cx<T>* __restrict__ s, const W& wk, MID&& mid) {
  using V = typename vreg<T>::type;
  constexpr int r = 1 << LG, lgnb = LGN - LG;
  wk.template each<lgnb>([&](int seq, int rr) {
    cx<T>* p = s + seq * LD + pad(rr << LG);                          // r <= 16 adjacent slots: pad(b0 + m) == pad(b0) + pad(m)
    V v[r], u[r];
#pragma unroll
    for (int m = 0; m < r; ++m) v[m] = vload(p + pad(m));
    dft<T, LG, false>(v);
#pragma unroll
    for (int k = 0; k < r; ++k) u[k] = mid(seq, rr << LG, brevc<LG>(k), v[dft_loc<LG>(k)]);   // slot = b0 + j, j a constant after unrolling
    dft<T, LG, true>(u);
#pragma unroll
    for (int m = 0; m < r; ++m) vstore(p + pad(m), u[dft_loc<LG>(m)]);
  });
  wk.sync();
}
// ---- forward, DIF: natural -> bit-reversed -------------------------------------------------------
// SKIP = 1: the top level (span N/2) is done by the caller; the stages cover the remaining LGN - 1 levels, i.e. both halves
// DROP = 1: the last stage is left to the caller (dif_mid_dit_stage fuses it with the first inverse stage)
template <typename T, int LD, int LGN, int LGNTW, int MAXLG, int SKIP, typename W, int I = 0, int DROP = 0>
__device__ __forceinline__ void fft_dif_w(cx<T>* __restrict__ s, const W& wk, const cx<T>* __restrict__ tw) {
  if constexpr (I < num_stages(LGN - SKIP, MAXLG) - DROP) {
    constexpr int LG = stage_lg(LGN - SKIP, I, MAXLG);
    constexpr int LGH = levels_after(LGN - SKIP, I, MAXLG) + LG - 1;      // top span index of this stage
    dif_stage<T, LD, LGN, LGNTW, LGH, LG>(s, wk, tw);
    fft_dif_w<T, LD, LGN, LGNTW, MAXLG, SKIP, W, I + 1, DROP>(s, wk, tw);
  }
}
template <typename T, int NT, int LD, int LGN, int LGNTW, int MAXLG = 4>
__device__ __forceinline__ void fft_dif(cx<T>* __restrict__ s, int S, const cx<T>* __restrict__ tw) {
  fft_dif_w<T, LD, LGN, LGNTW, MAXLG, 0>(s, WorkCoop<NT>{S}, tw);
}

// ---- inverse, DIT: bit-reversed -> natural (unnormalised); the forward schedule replayed backwards ----
template <typename T, int LD, int LGN, int LGNTW, int MAXLG, int SKIP, typename W, typename PRE = NoPre, int I = num_stages(LGN - SKIP, MAXLG) - 1>
__device__ __forceinline__ void fft_dit_w(cx<T>* __restrict__ s, const W& wk, const cx<T>* __restrict__ tw, PRE pre = PRE()) {
  if constexpr (I >= 0) {
    constexpr int LG = stage_lg(LGN - SKIP, I, MAXLG);
    constexpr int LGH = levels_after(LGN - SKIP, I, MAXLG);        // bottom span index of this stage
    if constexpr (I == num_stages(LGN - SKIP, MAXLG) - 1) dit_stage<T, LD, LGN, LGNTW, LGH, LG, W, PRE>(s, wk, tw, pre);   // pre applies to the bit-reversed input
    else dit_stage<T, LD, LGN, LGNTW, LGH, LG>(s, wk, tw);
    fft_dit_w<T, LD, LGN, LGNTW, MAXLG, SKIP, W, NoPre, I - 1>(s, wk, tw);
  }
}
template <typename T, int NT, int LD, int LGN, int LGNTW, int MAXLG = 4, typename PRE = NoPre>
__device__ __forceinline__ void fft_dit(cx<T>* __restrict__ s, int S, const cx<T>* __restrict__ tw, PRE pre = PRE()) {
  fft_dit_w<T, LD, LGN, LGNTW, MAXLG, 0, WorkCoop<NT>, PRE>(s, WorkCoop<NT>{S}, tw, pre);
}

// forward transform, mid(seq, b0, j, value) on the bit-reversed spectrum, inverse transform (unnormalised), SKIP top levels left to the caller
template <typename T, int LD, int LGN, int LGNTW, int MAXLG, int SKIP, typename W, typename MID>
__device__ __forceinline__ void fft_dif_mid_dit_w(cx<T>* __restrict__ s, const W& wk, const cx<T>* __restrict__ tw, MID&& mid) {
  constexpr int NS = num_stages(LGN - SKIP, MAXLG);
  fft_dif_w<T, LD, LGN, LGNTW, MAXLG, SKIP, W, 0, 1>(s, wk, tw);
  dif_mid_dit_stage<T, LD, LGN, stage_lg(LGN - SKIP, NS - 1, MAXLG)>(s, wk, mid);
  fft_dit_w<T, LD, LGN, LGNTW, MAXLG, SKIP, W, NoPre, NS - 2>(s, wk, tw);
}

// ---- packed real <-> half spectrum, in place on the tile ------------------------------------------
// LDS index of half-spectrum entry k (0..M); the Nyquist entry k = M uses the spare slot M
template <int LGM> __device__ __forceinline__ int hslot(int k) { return pad(k < (1 << LGM) ? brevc<LGM>(k) : (1 << LGM)); }

// after fft_dif on z[j] = f[2j] + i f[2j+1]:  A[k] for k = 0..M   (twN: exp(-2 pi i k/N), N = 2M, k < M)
template <typename T, int NT, int LD, int LGM>
__device__ __forceinline__ void r2c_post(cx<T>* __restrict__ s, int S, const cx<T>* __restrict__ twN) {
  constexpr int M = 1 << LGM, np = (M >> 1) + 1;       // pairs k = 0..M/2
  for (int q = threadIdx.x; q < S * np; q += NT) {
    const int seq = q / np, k = q - seq * np;
    cx<T>* p = s + seq * LD;
    if (k == 0) {
      cx<T> z = p[0];
      p[0] = mk<T>(z.x + z.y, 0);
      p[pad(M)] = mk<T>(z.x - z.y, 0);
    } else {
      const int k2 = M - k, i1 = pad(brevc<LGM>(k)), i2 = pad(brevc<LGM>(k2));
      cx<T> a = p[i1], b = p[i2];
      cx<T> e = mk<T>(T(0.5) * (a.x + b.x), T(0.5) * (a.y - b.y));   // (a + conj b)/2
      cx<T> o = mk<T>(T(0.5) * (a.x - b.x), T(0.5) * (a.y + b.y));   // (a - conj b)/2
      cx<T> wo = mul_mi(o * twN[k]);                                 // -i w^k o
      p[i1] = e + wo;
      if (k2 != k) p[i2] = conj(e - wo);
    }
  }
  __syncthreads();
}

// before fft_dit: Z[k] from A[k]; imaginary parts of A[0], A[M] are dropped (FFTW c2r semantics).
// Result of fft_dit is then  (f[2j] + i f[2j+1]) * N   (unnormalised, like FFTW's backward transform).
template <typename T, int NT, int LD, int LGM>
__device__ __forceinline__ void c2r_pre(cx<T>* __restrict__ s, int S, const cx<T>* __restrict__ twN) {
  constexpr int M = 1 << LGM, np = (M >> 1) + 1;
  for (int q = threadIdx.x; q < S * np; q += NT) {
    const int seq = q / np, k = q - seq * np;
    cx<T>* p = s + seq * LD;
    if (k == 0) {
      T a0 = p[0].x, am = p[pad(M)].x;
      p[0] = mk<T>(a0 + am, a0 - am);
    } else {
      const int k2 = M - k, i1 = pad(brevc<LGM>(k)), i2 = pad(brevc<LGM>(k2));
      cx<T> a = p[i1], b = p[i2];
      cx<T> e = mk<T>(a.x + b.x, a.y - b.y);                         // a + conj b
      cx<T> o = mk<T>(a.x - b.x, a.y + b.y);                         // a - conj b
      cx<T> wo = mul_i(cmulconj(o, twN[k]));                         // +i conj(w^k) o
      p[i1] = e + wo;
      if (k2 != k) p[i2] = conj(e - wo);
    }
  }
  __syncthreads();
}

}  // namespace cmbl
