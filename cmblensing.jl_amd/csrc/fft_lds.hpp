// In-LDS FFT building blocks (workgroup-cooperative, NT threads, compile-time sizes).
//
// Conventions (index math prototyped in tools/fft_proto.py):
//   forward  = e^{-i}, in-place radix-2 DIF network : natural order in  -> bit-reversed order out
//   inverse  = e^{+i}, in-place radix-2 DIT network : bit-reversed in   -> natural order out (unnormalised)
// so a forward/pointwise/inverse chain never needs a reordering pass: frequency k lives at slot brev(k).
// Up to four radix-2 levels are fused per LDS round trip: a thread pulls 2^LG elements into registers, runs LG levels of
// the in-place network on them (radix-16 for LG=4) and writes them back, so a 1024-point transform is 3 round trips / 3
// barriers.  Element i of a sequence lives at LDS index pad(i) = i + (i >> 4): the one-in-sixteen padding spreads the
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
// tw[] is an LDS table exp(-2*pi*i*k/Ntw), k < Ntw/2 (Ntw = 2^LGNTW); a transform of length n <= Ntw uses stride Ntw/n.
#pragma once
#include "common.hpp"

#ifndef CMBL_STAGE_SYNC
#define CMBL_STAGE_SYNC() __syncthreads()
#endif

namespace cmbl {

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

// exp(-2 pi i k/16), k = 0..7 (compile-time after unrolling)
template <typename T> __device__ __forceinline__ cx<T> root16(int k) {
  constexpr double c[8] = {1.0, 0.92387953251128674, 0.70710678118654752, 0.38268343236508977, 0.0,
                           -0.38268343236508977, -0.70710678118654752, -0.92387953251128674};
  constexpr double s[8] = {0.0, -0.38268343236508977, -0.70710678118654752, -0.92387953251128674, -1.0,
                           -0.92387953251128674, -0.70710678118654752, -0.38268343236508977};
  return mk<T>((T)c[k], (T)s[k]);
}
template <typename T> __device__ __forceinline__ cx<T> mul_root16(cx<T> a, int k) {       // a * exp(-2 pi i k/16)
  if (k == 0) return a;
  if (k == 4) return mul_mi(a);
  return a * root16<T>(k);
}
template <typename T> __device__ __forceinline__ cx<T> mul_root16c(cx<T> a, int k) {      // a * exp(+2 pi i k/16)
  if (k == 0) return a;
  if (k == 4) return mul_i(a);
  return cmulconj(a, root16<T>(k));
}

// ---- stage schedule: levels per stage as even as possible over ceil(lgN/4) stages (9 -> 3,3,3 ; 10 -> 4,3,3 ; 5 -> 3,2)
// MAXLG caps the radix (2^MAXLG): 4 = radix-16 (fewest barriers), 3 = radix-8 (half the registers, one more round trip).
constexpr int stage_levels(int remaining, int maxlg = 4) { return (remaining + ((remaining + maxlg - 1) / maxlg) - 1) / ((remaining + maxlg - 1) / maxlg); }
constexpr int num_stages(int lgN, int maxlg = 4) { int n = 0; while (lgN > 0) { lgN -= stage_levels(lgN, maxlg); ++n; } return n; }
constexpr int stage_lg(int lgN, int idx, int maxlg = 4) { int lg = 0; for (int i = 0; i <= idx; ++i) { lg = stage_levels(lgN, maxlg); lgN -= lg; } return lg; }
constexpr int levels_after(int lgN, int idx, int maxlg = 4) { int tot = 0; for (int i = 0; i <= idx; ++i) tot += stage_lg(lgN, i, maxlg); return lgN - tot; }

// One fused DIF stage: LG radix-2 levels with spans h = 2^LGH (top) ... hmin = 2^(LGH-LG+1).
template <typename T, int NT, int LD, int LGN, int LGNTW, int LGH, int LG>
__device__ __forceinline__ void dif_stage(cx<T>* __restrict__ s, int S, const cx<T>* __restrict__ tw) {
  constexpr int r = 1 << LG, lghmin = LGH - LG + 1, hmin = 1 << lghmin, lgnb = LGN - LG;
  for (int q = threadIdx.x; q < (S << lgnb); q += NT) {
    const int seq = q >> lgnb, rr = q & ((1 << lgnb) - 1);
    const int blk = rr >> lghmin, j = rr & (hmin - 1);
    cx<T>* p = s + seq * LD + pad((blk << (LGH + 1)) + j);
    cx<T> v[r];
#pragma unroll
    for (int m = 0; m < r; ++m) v[m] = p[pad(m << lghmin)];       // pad(base + m*hmin) == pad(base) + pad(m*hmin) here
#pragma unroll
    for (int t = 0; t < LG; ++t) {
      // level t: span h_t = h >> t ; pairs (m, m + r/2^(t+1)) inside groups of r/2^t
      const cx<T> w = tw[j << (LGNTW - (LGH - t) - 1)];            // W_{2 h_t}^j
      const int half = r >> (t + 1);
#pragma unroll
      for (int m = 0; m < r; ++m) {
        if ((m & half) == 0) {
          const int mp = m & (half - 1);                            // position inside the half group
          const cx<T> a = v[m], b = v[m + half];
          v[m] = a + b;
          // twiddle W_{2h_t}^{j + mp*hmin} = w * W_{r/2^t}^{mp} = w * root16^(mp * 16 / (r >> t))
          v[m + half] = mul_root16((a - b) * w, mp << (4 - (LG - t)));
        }
      }
    }
#pragma unroll
    for (int m = 0; m < r; ++m) p[pad(m << lghmin)] = v[m];
  }
  CMBL_STAGE_SYNC();
}

// One fused DIT stage: LG levels with spans hmin = 2^LGH (bottom) ... hmin * 2^(LG-1).
struct NoPre { template <typename V> __device__ __forceinline__ V operator()(V v, int) const { return v; } };

template <typename T, int NT, int LD, int LGN, int LGNTW, int LGH, int LG, typename PRE = NoPre>
__device__ __forceinline__ void dit_stage(cx<T>* __restrict__ s, int S, const cx<T>* __restrict__ tw, PRE pre = PRE()) {
  constexpr int r = 1 << LG, hmin = 1 << LGH, lgnb = LGN - LG;
  for (int q = threadIdx.x; q < (S << lgnb); q += NT) {
    const int seq = q >> lgnb, rr = q & ((1 << lgnb) - 1);
    const int blk = rr >> LGH, j = rr & (hmin - 1);
    const int b0 = (blk << (LGH + LG)) + j;                          // logical (unpadded) index of element m = 0
    cx<T>* p = s + seq * LD + pad(b0);
    cx<T> v[r];
#pragma unroll
    for (int m = 0; m < r; ++m) v[m] = pre(p[pad(m << LGH)], b0 + (m << LGH));   // pre: pointwise op fused into the first stage
#pragma unroll
    for (int t = 0; t < LG; ++t) {
      // level t: span h_t = hmin << t ; pairs (m, m + 2^t) inside groups of 2^(t+1)
      const cx<T> w = tw[j << (LGNTW - (LGH + t) - 1)];            // W_{2 h_t}^j  (conjugated below)
      const int half = 1 << t;
#pragma unroll
      for (int m = 0; m < r; ++m) {
        if ((m & half) == 0) {
          const int mp = m & (half - 1);
          const cx<T> a = v[m];
          // conj( W_{2h_t}^{j + mp*hmin} ) = conj(w) * conj(W_{2^(t+1)}^{mp})
          const cx<T> b = mul_root16c(cmulconj(v[m + half], w), mp << (4 - (t + 1)));
          v[m] = a + b;
          v[m + half] = a - b;
        }
      }
    }
#pragma unroll
    for (int m = 0; m < r; ++m) p[pad(m << LGH)] = v[m];
  }
  CMBL_STAGE_SYNC();
}

// ---- forward, DIF: natural -> bit-reversed -------------------------------------------------------
template <typename T, int NT, int LD, int LGN, int LGNTW, int MAXLG = 4, int I = 0>
__device__ __forceinline__ void fft_dif(cx<T>* __restrict__ s, int S, const cx<T>* __restrict__ tw) {
  if constexpr (I < num_stages(LGN, MAXLG)) {
    constexpr int LG = stage_lg(LGN, I, MAXLG);
    constexpr int LGH = levels_after(LGN, I, MAXLG) + LG - 1;      // top span index of this stage
    dif_stage<T, NT, LD, LGN, LGNTW, LGH, LG>(s, S, tw);
    fft_dif<T, NT, LD, LGN, LGNTW, MAXLG, I + 1>(s, S, tw);
  }
}

// ---- inverse, DIT: bit-reversed -> natural (unnormalised); the forward schedule replayed backwards ----
template <typename T, int NT, int LD, int LGN, int LGNTW, int MAXLG = 4, int I = num_stages(LGN, MAXLG) - 1, typename PRE = NoPre>
__device__ __forceinline__ void fft_dit(cx<T>* __restrict__ s, int S, const cx<T>* __restrict__ tw, PRE pre = PRE()) {
  if constexpr (I >= 0) {
    constexpr int LG = stage_lg(LGN, I, MAXLG);
    constexpr int LGH = levels_after(LGN, I, MAXLG);               // bottom span index of this stage
    if constexpr (I == num_stages(LGN, MAXLG) - 1) dit_stage<T, NT, LD, LGN, LGNTW, LGH, LG, PRE>(s, S, tw, pre);   // pre applies to the bit-reversed input
    else dit_stage<T, NT, LD, LGN, LGNTW, LGH, LG>(s, S, tw);
    fft_dit<T, NT, LD, LGN, LGNTW, MAXLG, I - 1>(s, S, tw);
  }
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
