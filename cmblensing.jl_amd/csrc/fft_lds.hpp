// In-LDS FFT building blocks (workgroup-cooperative, NT threads, runtime sizes).
//
// Conventions (prototype + index-math check: tools/fft_proto.py):
//   forward  = e^{-i}, in-place radix-2^2 DIF : natural order in  -> bit-reversed order out
//   inverse  = e^{+i}, in-place radix-2^2 DIT : bit-reversed in   -> natural order out (unnormalised)
// so a forward/pointwise/inverse chain never needs a reordering pass; frequency-domain data simply
// lives at LDS slot brev(k).  Real transforms of length N=2M use the packed trick (M complex points);
// the half-spectrum A[0..M] sits at slots brev(k) for k<M and slot M for k=M (tile leading dim LD>=M+1).
// c2r drops Im A[0] and Im A[M] exactly like FFTW / pocketfft / cuFFT do (src/util_fft.jl:21-25 path).
//
// A "tile" is S sequences of LD complex slots each.  tw[] is an LDS table exp(-2*pi*i*k/Ntw), k<Ntw/2;
// a transform of length n uses stride Ntw/n into it.
#pragma once
#include "common.hpp"

namespace cmbl {

template <typename T>
__device__ __forceinline__ void load_twiddles(cx<T>* tw_lds, const cx<T>* __restrict__ tw_g, int nhalf) {
  for (int i = threadIdx.x; i < nhalf; i += NT) tw_lds[i] = tw_g[i];
}

// ---- forward, DIF ------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void fft_dif(cx<T>* __restrict__ s, int S, int LD, int lgN,
                                        const cx<T>* __restrict__ tw, int lgNtw) {
  const int N = 1 << lgN;
  int lgh = lgN - 1;                                   // h = 2^lgh : span of the first radix-2 level
  // fused pairs of levels (spans h and h/2)
  for (; lgh >= 1; lgh -= 2) {
    const int h = 1 << lgh, hh = h >> 1;
    const int tws1 = lgNtw - (lgh + 1);                // W_{2h}^j = tw[j << tws1]
    const int nq = N >> 2;
    for (int q = threadIdx.x; q < S * nq; q += NT) {
      const int seq = q >> (lgN - 2), r = q & (nq - 1);
      const int blk = r >> (lgh - 1), j = r & (hh - 1);
      cx<T>* p = s + seq * LD + (blk << (lgh + 1)) + j;
      cx<T> x0 = p[0], x1 = p[hh], x2 = p[h], x3 = p[h + hh];
      const cx<T> w1 = tw[j << tws1];
      const cx<T> w2 = tw[j << (tws1 + 1)];
      cx<T> u0 = x0 + x2, u2 = (x0 - x2) * w1;
      cx<T> u1 = x1 + x3, u3 = mul_mi((x1 - x3) * w1);      // W_{2h}^{j+h/2} = -i W_{2h}^j
      p[0] = u0 + u1;       p[hh] = (u0 - u1) * w2;
      p[h] = u2 + u3;       p[h + hh] = (u2 - u3) * w2;
    }
    __syncthreads();
  }
  if (lgh == 0) {                                      // odd log2: last plain radix-2 level, h = 1, w = 1
    const int nb = N >> 1;
    for (int q = threadIdx.x; q < S * nb; q += NT) {
      const int seq = q >> (lgN - 1), r = q & (nb - 1);
      cx<T>* p = s + seq * LD + (r << 1);
      cx<T> a = p[0], b = p[1];
      p[0] = a + b; p[1] = a - b;
    }
    __syncthreads();
  }
}

// ---- inverse, DIT ------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void fft_dit(cx<T>* __restrict__ s, int S, int LD, int lgN,
                                        const cx<T>* __restrict__ tw, int lgNtw) {
  const int N = 1 << lgN;
  int lgh = 0;
  if (lgN & 1) {                                       // odd log2: first plain radix-2 level, h = 1
    const int nb = N >> 1;
    for (int q = threadIdx.x; q < S * nb; q += NT) {
      const int seq = q >> (lgN - 1), r = q & (nb - 1);
      cx<T>* p = s + seq * LD + (r << 1);
      cx<T> a = p[0], b = p[1];
      p[0] = a + b; p[1] = a - b;
    }
    __syncthreads();
    lgh = 1;
  }
  for (; lgh + 1 < lgN + 0 && lgh + 2 <= lgN; lgh += 2) {   // fused levels with spans h and 2h
    const int h = 1 << lgh;
    const int tws1 = lgNtw - (lgh + 1);                // conj W_{2h}^j
    const int nq = N >> 2;
    for (int q = threadIdx.x; q < S * nq; q += NT) {
      const int seq = q >> (lgN - 2), r = q & (nq - 1);
      const int blk = r >> lgh, j = r & (h - 1);
      cx<T>* p = s + seq * LD + (blk << (lgh + 2)) + j;
      cx<T> x0 = p[0], x1 = p[h], x2 = p[2 * h], x3 = p[3 * h];
      const cx<T> w1 = tw[j << tws1];
      const cx<T> w2 = tw[j << (tws1 - 1)];            // W_{4h}^j
      cx<T> t = cmulconj(x1, w1); cx<T> u0 = x0 + t, u1 = x0 - t;
      t = cmulconj(x3, w1);       cx<T> u2 = x2 + t, u3 = x2 - t;
      t = cmulconj(u2, w2);       p[0] = u0 + t;  p[2 * h] = u0 - t;
      t = mul_i(cmulconj(u3, w2));                      // conj W_{4h}^{j+h} = +i conj W_{4h}^j
      p[h] = u1 + t;  p[3 * h] = u1 - t;
    }
    __syncthreads();
  }
}

// ---- packed real <-> half spectrum, in place on the tile ------------------------------------------
// slot of half-spectrum index k (0..M)
__device__ __forceinline__ int hslot(int k, int M, int lgM) { return k < M ? brev(k, lgM) : M; }

// after fft_dif on z[j] = f[2j] + i f[2j+1]:  A[k] for k = 0..M   (twN: exp(-2 pi i k/N), N = 2M, k < M)
template <typename T>
__device__ __forceinline__ void r2c_post(cx<T>* __restrict__ s, int S, int LD, int lgM,
                                         const cx<T>* __restrict__ twN) {
  const int M = 1 << lgM, np = (M >> 1) + 1;           // pairs k = 0..M/2
  for (int q = threadIdx.x; q < S * np; q += NT) {
    const int seq = q / np, k = q - seq * np;
    cx<T>* p = s + seq * LD;
    if (k == 0) {
      cx<T> z = p[0];
      p[0] = mk<T>(z.x + z.y, 0);
      p[M] = mk<T>(z.x - z.y, 0);
    } else {
      const int k2 = M - k, i1 = brev(k, lgM), i2 = brev(k2, lgM);
      cx<T> a = p[i1], b = p[i2];
      cx<T> e = mk<T>(T(0.5) * (a.x + b.x), T(0.5) * (a.y - b.y));   // (a + conj b)/2
      cx<T> o = mk<T>(T(0.5) * (a.x - b.x), T(0.5) * (a.y + b.y));   // (a - conj b)/2
      cx<T> wo = mul_mi(o * twN[k]);                                 // -i w^k o
      p[i1] = e + wo;
      if (k2 != k) p[i2] = conj(e - wo);                             // A[M-k] = conj(e) - (-i w^{M-k}) ... = conj(e - wo)
    }
  }
  __syncthreads();
}

// before fft_dit: Z[k] from A[k]; imaginary parts of A[0], A[M] are dropped (FFTW c2r semantics).
// Result of fft_dit is then  (f[2j] + i f[2j+1]) * N   (unnormalised, like FFTW's backward transform).
template <typename T>
__device__ __forceinline__ void c2r_pre(cx<T>* __restrict__ s, int S, int LD, int lgM,
                                        const cx<T>* __restrict__ twN) {
  const int M = 1 << lgM, np = (M >> 1) + 1;
  for (int q = threadIdx.x; q < S * np; q += NT) {
    const int seq = q / np, k = q - seq * np;
    cx<T>* p = s + seq * LD;
    if (k == 0) {
      T a0 = p[0].x, am = p[M].x;
      p[0] = mk<T>(a0 + am, a0 - am);
    } else {
      const int k2 = M - k, i1 = brev(k, lgM), i2 = brev(k2, lgM);
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
