// Any-size path: Ny, Nx that are not powers of two (the reference takes any size through FFTW plans, src/util_fft.jl:32-35).
//
// The fused row / column kernels are compiled for power-of-two sizes (bit-reversed slot maps, compile-time strides).  Other sizes run
// the reference's own pass structure (src/lenseflow.jl:150-214: transform, multiply, transform, pointwise) on two building blocks:
//   * k_gen_dft: batched 1-D DFT of any length N <= 4096 along either axis as a chirp-z (Bluestein) convolution,
//         X[k] = w[k] sum_n (x[n] w[n]) conj(w[k-n]),   w[n] = exp(-i pi n^2 / N),
//     evaluated with the power-of-two in-LDS transforms of fft_lds.hpp at length L >= 2N-1: forward DIF (natural -> bit-reversed),
//     multiply with the precomputed transform of the chirp filter (stored bit-reversed), inverse DIT (bit-reversed -> natural); no
//     reordering pass, any prime factors.  Real input / Hermitian input with FFTW's c2r semantics (imaginary parts of the ky = 0 and
//     Nyquist entries dropped AFTER the x pass, src/util_fft.jl:21-25) / real output are options of the same kernel.
//   * pointwise kernels for the multiplies, products and RK4 bookkeeping of a stage.
// The internal F layout keeps its meaning ([slice][ky][kx]) with kx in natural order instead of bit-reversed, so every
// table-driven pointwise kernel (operators, reductions, QE legs, gradhess, the delta-phi quadrature) is shared with the fast path.
#pragma once
#include "kernels_flow.hpp"

namespace cmbl {

// in [slice][R][C] -> out [slice][C][R]; grid (ceil(C/32), ceil(R/32), slices)
template <typename V>
__global__ __launch_bounds__(NTP) void k_transpose(const V* __restrict__ in, V* __restrict__ out, int R, int C) {
  __shared__ V tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t sl = blockIdx.z;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < R && c < C) tile[ty + 8 * i][tx] = in[(sl * R + r) * C + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < R && c < C) out[(sl * C + c) * R + r] = tile[tx][ty + 8 * i];
  }
}

template <typename T>
__device__ __forceinline__ void gen_p(const PhiMaps<T>& ph, size_t gi, T t, T& px, T& py) {
  if (ph.pcx) { px = ph.pcx[gi]; py = ph.pcy[gi]; }
  else { T m11, m12, m22; flow_pm(t, ph.gx[gi], ph.gy[gi], ph.hxx[gi], ph.hyx[gi], ph.hyy[gi], px, py, m11, m12, m22); }
}
// Pointwise work of a flow stage done in the FETCH of the real-input y transform that consumes its result (each map element is
// fetched exactly once), instead of in a launch of its own that writes a map the transform then reads back:
//   mode 1: the value fetched is the next stage input f_{s+1} = RK4 update with k = p . grad f (== k_gen_vel_rk), y0 / acc updated on the way;
//   mode 2: the same for the f part of a delta-flow stage, plus the per-stage products w = L(df) grad f (== the f half of k_gen_delta);
//   mode 3: the pair (p_x L(df), p_y L(df)) (== the delta-f half of k_gen_delta), for the pair r2c.
template <typename T> struct GenPro {
  int mode;                          // 0 = plain fetch
  PhiMaps<T> ph; RKCoef<T> rk;
  const T *gx, *gy, *Ldf;
  T *y0, *acc, *w1p, *w2p;
  long npix; int P;
};

// One launch = `nseq` sequences per slice (blockIdx.y), S per workgroup.  Element n of sequence q of slice s is at
// in[s*in_slice + q*in_seq + n*in_elem] (real T or cx<T>), likewise for the output.
template <typename T> struct GenDft {
  const void* in; void* out;
  const cx<T>* chirp;                // w[n], n < N
  const cx<T>* bhat;                 // DFT_L of the chirp filter / L, bit-reversed slot order
  const cx<T>* tw;                   // exp(-2 pi i k / L), k < L
  int N, nin, nout, nseq, S;
  long in_seq, in_elem, in_slice, out_seq, out_elem, out_slice;
  int in_real, out_real, inverse, herm;
  T scale;
  // Separable-derivative / pair options (all off by default):
  //   lmul_in  : half-spectrum input element m is multiplied by i*lmul_in[m] as it is fetched (d/dy from the y transform alone)
  //   lmul_out : output element k is multiplied by i*lmul_out[k] (d/dx: forward x pass with the i*lx multiply in its store)
  //   in2 / out2 (pair): TWO real sequences per complex transform.  herm (c2r): z = ext(in) + i ext(in2) -> out = Re, out2 = Im with
  //              scales (scale, scale2); lmul_in applies to the SECOND member only.  in_real (r2c): z = in + i in2 ->
  //              out[k] = (Z[k] + conj Z[N-k]) / 2, out2[k] = (Z[k] - conj Z[N-k]) / 2i, k < nout.
  const T* lmul_in; const T* lmul_out;
  const void* in2; void* out2;
  T scale2;
  // lmul_mid (mixed-radix kernel only): forward transform, multiply element k by i*lmul_mid[k], INVERSE transform, all in LDS -- the
  // d/dx pass of a stage in one launch (unnormalised: the caller's scale carries 1/N)
  const T* lmul_mid;
  GenPro<T> pro;
  // slice window (Flow::gen_window: one launch chain per group of slices): blockIdx.y = y covers slices sl0 + y % sln of each of the
  // grid.y / sln parts of the batch, which are slstride slices apart (a pair launch over 2 * slices has two parts).  sln = 0: slice y.
  int sl0, sln, slstride;
  // Fused y passes of a flow stage (k_ct_dft only; kernels_ct.hpp): yy = 1 -- this launch is the pair c2r of a forward stage (herm, in2,
  // lmul_in, inverse, scale, scale2 as usual), but (d/dx f, d/dy f) stay in LDS: the stage's velocity and RK update (pro: ph, rk, y0, acc)
  // are applied there and the NEXT stage's rfft_y(f) is transformed and written to yy_out ([ky][x] like the inputs, yy_nout entries);
  // yy_last: the flow ends here (only y0 is updated).  One launch instead of two and no round trip of the two gradient maps.
  // yy = 2: the same for a delta-flow stage (k_ct_delta_y): a third half plane yy_in3 = ifft_x(delta f) gives L(df) (scale yy_scale3) next to
  // the gradient pair; the stage's products go to pro.w1p / w2p, the f part is updated like above (-> yy_out), and the pair
  // (p_x L(df), p_y L(df)) is transformed and split into yy_out2 / yy_out3 (what the pair r2c of gen_adj_update writes).
  int yy, yy_last, yy_nout;
  void* yy_out;
  const void* yy_in3; void* yy_out2; void* yy_out3;
  T yy_scale3;
  // TILED hand-off arrays (round 6; compile-time-plan kernels only): the half planes the column and row launches of the fused any-size stages pass
  // to each other as [slice][x / 4][ky (tile_np rows)][x % 4] -- the tiled mixed layout of the power-of-two path (kernels_fft.hpp mix_idx) -- instead
  // of [slice][ky][x].  A column workgroup's 4 (8) columns are then ONE (two) contiguous block(s) instead of 32-byte pieces a row apart, a row
  // workgroup of 4 adjacent ky gathers whole 128-byte lines.  in_tiled / out_tiled: 0 = strided (in_seq / in_elem), 1 = the sequence index is x (y
  // kernels), 2 = the element index is x (x kernels); in2, yy_in3 follow in_tiled, out2 / yy_out* follow out_tiled (addressing: CtSide, kernels_ct.hpp).
  int in_tiled, out_tiled, tile_np;
};
template <typename T> __device__ __forceinline__ size_t gen_slice(const GenDft<T>& a, unsigned y = blockIdx.y) {
  return a.sln ? (size_t)a.sl0 + (y % (unsigned)a.sln) + (size_t)(y / (unsigned)a.sln) * (unsigned)a.slstride : (size_t)y;
}


// input element n of a sequence: real / complex / Hermitian-extended half spectrum, conjugated for the e^{+i} transform
// Hermitian extension of a half spectrum with FFTW's c2r rule; `lm` != nullptr: the stored entries are multiplied by i*lm[m] first
template <typename T>
__device__ __forceinline__ cx<T> gen_herm(const cx<T>* p, size_t base, long elem, int n, int nin, int N, const T* lm) {
  const int m = n < nin ? n : N - n;
  cx<T> v = p[base + (size_t)m * elem];
  if (lm) v = mul_il(v, lm[m]);
  if (n == 0 || 2 * n == N) v.y = T(0);                               // FFTW c2r: these imaginary parts are never read
  return n < nin ? v : conj(v);
}
template <typename T>
__device__ __forceinline__ cx<T> gen_fetch(const GenDft<T>& a, size_t sl, int seq, int n) {
  const size_t base = sl * a.in_slice + (size_t)seq * a.in_seq;
  cx<T> v;
  if (a.in_real && a.pro.mode) {
    const GenPro<T>& e = a.pro;
    const size_t o = base + (size_t)n * a.in_elem, i = o - sl * (size_t)e.npix, pb = (size_t)(e.ph.Bphi == 1 ? 0 : sl / e.P) * e.npix;
    T px, py; gen_p(e.ph, pb + i, e.rk.t, px, py);
    if (e.mode == 3) { const T l = e.Ldf[o]; v = mk<T>(px * l, py * l); }
    else {
      const T ax = e.gx[o], ay = e.gy[o];
      if (e.mode == 2) { const T l = e.Ldf[o]; e.w1p[o] = l * ax; e.w2p[o] = l * ay; }
      const T k = px * ax + py * ay;
      T y = e.y0[o], ac = e.rk.stage == 1 ? T(0) : e.acc[o];
      const T nxt = rk_update(e.rk, k, y, ac);
      if (e.rk.stage == 4) e.y0[o] = y; else e.acc[o] = ac;
      v = mk<T>(nxt, T(0));
    }
  } else if (a.in_real) {
    v = mk<T>(reinterpret_cast<const T*>(a.in)[base + (size_t)n * a.in_elem], T(0));
    if (a.in2) v.y = reinterpret_cast<const T*>(a.in2)[base + (size_t)n * a.in_elem];
  } else if (!a.herm) v = reinterpret_cast<const cx<T>*>(a.in)[base + (size_t)n * a.in_elem];
  else if (!a.in2) v = gen_herm(reinterpret_cast<const cx<T>*>(a.in), base, a.in_elem, n, a.nin, a.N, a.lmul_in);
  else {
    const cx<T> u = gen_herm(reinterpret_cast<const cx<T>*>(a.in), base, a.in_elem, n, a.nin, a.N, (const T*)nullptr);
    const cx<T> w = gen_herm(reinterpret_cast<const cx<T>*>(a.in2), base, a.in_elem, n, a.nin, a.N, a.lmul_in);
    v = mk<T>(u.x - w.y, u.y + w.x);                                  // u + i w
  }
  return a.inverse ? conj(v) : v;                                     // e^{+i} transform = conj(forward(conj x))
}
#ifndef CMBL_WT_GEN
#define CMBL_WT_GEN 1
#endif
template <typename V> __device__ __forceinline__ void gen_wt(V* p, V v, bool wt) {
#if CMBL_WT_GEN
  if (wt) store_wt<(int)sizeof(V)>(p, &v); else *p = v;
#else
  *p = v;
#endif
}
// y = Z[k]; yr = Z[(N - k) % N] (only read for a real pair)
template <typename T>
__device__ __forceinline__ void gen_put(const GenDft<T>& a, size_t sl, int seq, int k, cx<T> y, cx<T> yr = cx<T>{}) {
  if (a.inverse) y = conj(y);
  const size_t o = sl * a.out_slice + (size_t)seq * a.out_seq + (size_t)k * a.out_elem;
  // every output of a transform launch is read by workgroups of the next launch on other XCDs: written through (see handoff_store, wt_line)
  const bool wt = wt_line<T>(a.N);
  if (a.out_real) {
    gen_wt(reinterpret_cast<T*>(a.out) + o, T(a.scale * y.x), wt);
    if (a.out2) gen_wt(reinterpret_cast<T*>(a.out2) + o, T(a.scale2 * y.y), wt);
    return;
  }
  if (a.in_real && a.in2) {                                           // split the transform of in + i in2
    const cx<T> c = conj(yr);
    const cx<T> x1 = mk<T>(T(0.5) * (y.x + c.x), T(0.5) * (y.y + c.y)), d = mk<T>(T(0.5) * (y.x - c.x), T(0.5) * (y.y - c.y));
    gen_wt(reinterpret_cast<cx<T>*>(a.out) + o, mk<T>(a.scale * x1.x, a.scale * x1.y), wt);
    gen_wt(reinterpret_cast<cx<T>*>(a.out2) + o, mk<T>(a.scale2 * d.y, -a.scale2 * d.x), wt);            // d / i
    return;
  }
  if (a.lmul_out) y = mul_il(y, a.lmul_out[k]);
  gen_wt(reinterpret_cast<cx<T>*>(a.out) + o, mk<T>(a.scale * y.x, a.scale * y.y), wt);
}

template <typename T, int LGL>
__global__ __launch_bounds__(NTP) void k_gen_dft(GenDft<T> a) {
  constexpr int L = 1 << LGL, LD = tile_ld(L);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* s = reinterpret_cast<cx<T>*>(smem);
  const int S = a.S, seq0 = blockIdx.x * S;
  const size_t sl = gen_slice(a);
  // consecutive threads walk whichever of (element, sequence) is contiguous in memory
  const bool in_by_seq = a.in_elem != 1 && S > 1, out_by_seq = a.out_elem != 1 && S > 1;
  for (int q = threadIdx.x; q < S * L; q += NTP) {
    int sq, n;
    if (in_by_seq) { sq = q % S; n = q / S; } else { sq = q >> LGL; n = q & (L - 1); }
    const int seq = seq0 + sq;
    cx<T> v = mk<T>(T(0), T(0));
    if (seq < a.nseq && n < a.N) v = gen_fetch(a, sl, seq, n) * a.chirp[n];
    s[sq * LD + pad(n)] = v;
  }
  __syncthreads();
  fft_dif<T, NTP, LD, LGL, LGL>(s, S, a.tw);
  for (int q = threadIdx.x; q < S * L; q += NTP) {
    const int sq = q >> LGL, n = q & (L - 1);
    cx<T>* p = s + sq * LD + pad(n);
    *p = *p * a.bhat[n];
  }
  __syncthreads();
  fft_dit<T, NTP, LD, LGL, LGL>(s, S, a.tw);
  for (int q = threadIdx.x; q < S * a.nout; q += NTP) {
    int sq, k;
    if (out_by_seq) { sq = q % S; k = q / S; } else { sq = q / a.nout; k = q - sq * a.nout; }
    const int seq = seq0 + sq;
    if (seq >= a.nseq) continue;
    const int kr = k ? a.N - k : 0;
    gen_put(a, sl, seq, k, s[sq * LD + pad(k)] * a.chirp[k], s[sq * LD + pad(kr)] * a.chirp[kr]);
  }
}


// ---- mixed-radix transforms for N = 2^a 3^b 5^c 7^d 11^e 13^f ----------------------------------------------------------------
// Stockham autosort network (natural order in and out, ping-pong between two LDS buffers): a stage of radix R with Ns = product of
// the earlier radices does, for every j < N/R with k = j mod Ns,
//     v[m] = X[j + m N/R] W_{Ns R}^{m k},   v <- DFT_R(v),   Y[(j div Ns) Ns R + k + m Ns] = v[m].
// Radices 2, 3, 4, 5 are written out; 7, 11, 13 are direct sums with table twiddles.  Sizes with larger prime factors use k_gen_dft.
struct GenPlan { int nf; int radix[14]; };

template <typename T, int R> struct Bfly;
template <typename T> struct Bfly<T, 2> { static __device__ __forceinline__ void run(cx<T>* v, const cx<T>*, int) { const cx<T> a = v[0], b = v[1]; v[0] = a + b; v[1] = a - b; } };
template <typename T> struct Bfly<T, 4> {
  static __device__ __forceinline__ void run(cx<T>* v, const cx<T>*, int) {
    const cx<T> a = v[0] + v[2], b = v[0] - v[2], c = v[1] + v[3], d = mul_mi(v[1] - v[3]);
    v[0] = a + c; v[1] = b + d; v[2] = a - c; v[3] = b - d;
  }
};
template <typename T> struct Bfly<T, 3> {
  static __device__ __forceinline__ void run(cx<T>* v, const cx<T>*, int) {
    const cx<T> t = v[1] + v[2], m = v[0] - T(0.5) * t, sd = T(0.86602540378443864676) * (v[1] - v[2]);
    v[0] = v[0] + t; v[1] = m + mul_mi(sd); v[2] = m + mul_i(sd);
  }
};
template <typename T> struct Bfly<T, 5> {
  static __device__ __forceinline__ void run(cx<T>* v, const cx<T>*, int) {
    constexpr T c1 = T(0.30901699437494742410), c2 = T(-0.80901699437494742410), s1 = T(0.95105651629515357212), s2 = T(0.58778525229247312917);
    const cx<T> t1 = v[1] + v[4], t2 = v[2] + v[3], t3 = v[1] - v[4], t4 = v[2] - v[3];
    const cx<T> a1 = v[0] + c1 * t1 + c2 * t2, a2 = v[0] + c2 * t1 + c1 * t2;
    const cx<T> b1 = s1 * t3 + s2 * t4, b2 = s2 * t3 - s1 * t4;
    v[0] = v[0] + t1 + t2;
    v[1] = a1 + mul_mi(b1); v[4] = a1 + mul_i(b1); v[2] = a2 + mul_mi(b2); v[3] = a2 + mul_i(b2);
  }
};
// direct R-point sum; tw is the full-circle table W_N, step = N / R
template <typename T, int R> struct Bfly {
  static __device__ __forceinline__ void run(cx<T>* v, const cx<T>* tw, int step) {
    cx<T> o[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      cx<T> acc = v[0];
#pragma unroll
      for (int m = 1; m < R; ++m) acc = acc + v[m] * tw[((m * k) % R) * step];
      o[k] = acc;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = o[k];
  }
};

template <typename T, int R>
__device__ __forceinline__ void mr_stage(const cx<T>* __restrict__ X, cx<T>* __restrict__ Y, int N, int Ns, const cx<T>* __restrict__ tw, int S) {
  const int nb = N / R, tstep = N / (Ns * R), nt = blockDim.x;
  for (int q = threadIdx.x; q < S * nb; q += nt) {
    const int sq = q / nb, j = q - sq * nb, blk = j / Ns, k = j - blk * Ns;
    const cx<T>* x = X + sq * N + j;
    cx<T> v[R];
#pragma unroll
    for (int m = 0; m < R; ++m) v[m] = x[m * nb];
    if (Ns > 1) {
#pragma unroll
      for (int m = 1; m < R; ++m) v[m] = v[m] * tw[m * k * tstep];
    }
    Bfly<T, R>::run(v, tw, nb);
    cx<T>* y = Y + sq * N + blk * Ns * R + k;
#pragma unroll
    for (int m = 0; m < R; ++m) y[m * Ns] = v[m];
  }
  __syncthreads();
}

// BIG: the plan contains a radix above 5 (direct-sum butterflies: many registers), compiled apart so that the common
// 2/3/4/5 kernel keeps a small register footprint and can run 1024 threads.  Launched with a multiple of 64 threads.
template <typename T, bool BIG>
__global__ __launch_bounds__(BIG ? NTP : 1024) void k_gen_dft_mr(GenDft<T> a, GenPlan plan, int tw_in_lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int N = a.N, S = a.S, seq0 = blockIdx.x * S, nt = blockDim.x;
  cx<T>* X = reinterpret_cast<cx<T>*>(smem);
  cx<T>* Y = X + (size_t)S * N;
  const cx<T>* tw = a.tw;                                              // W_N, N entries
  if (tw_in_lds) {
    cx<T>* t = Y + (size_t)S * N;
    for (int i = threadIdx.x; i < N; i += nt) t[i] = a.tw[i];
    tw = t;
  }
  const size_t sl = gen_slice(a);
  const bool in_by_seq = a.in_elem != 1 && S > 1, out_by_seq = a.out_elem != 1 && S > 1;
  for (int q = threadIdx.x; q < S * N; q += nt) {
    int sq, n;
    if (in_by_seq) { sq = q % S; n = q / S; } else { sq = q / N; n = q - sq * N; }
    const int seq = seq0 + sq;
    X[sq * N + n] = seq < a.nseq ? gen_fetch(a, sl, seq, n) : mk<T>(T(0), T(0));
  }
  __syncthreads();
  const int npass = a.lmul_mid ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  if (pass == 1) {                                                    // X = conj(i l X): the inverse transform is conj(forward(conj .))
    for (int q = threadIdx.x; q < S * N; q += nt) {
      const int sq = q / N, k = q - sq * N;
      X[q] = conj(mul_il(X[q], a.lmul_mid[k]));
    }
    __syncthreads();
  }
  int Ns = 1;
  for (int f = 0; f < plan.nf; ++f) {
    const int R = plan.radix[f];
    if constexpr (BIG) {
      switch (R) {
        case 7: mr_stage<T, 7>(X, Y, N, Ns, tw, S); break;
        case 11: mr_stage<T, 11>(X, Y, N, Ns, tw, S); break;
        case 13: mr_stage<T, 13>(X, Y, N, Ns, tw, S); break;
        default: break;
      }
    }
    switch (R) {
      case 2: mr_stage<T, 2>(X, Y, N, Ns, tw, S); break;
      case 3: mr_stage<T, 3>(X, Y, N, Ns, tw, S); break;
      case 4: mr_stage<T, 4>(X, Y, N, Ns, tw, S); break;
      case 5: mr_stage<T, 5>(X, Y, N, Ns, tw, S); break;
      default: break;
    }
    Ns *= R;
    cx<T>* t = X; X = Y; Y = t;
  }
  }
  for (int q = threadIdx.x; q < S * a.nout; q += nt) {
    int sq, k;
    if (out_by_seq) { sq = q % S; k = q / S; } else { sq = q / a.nout; k = q - sq * a.nout; }
    const int seq = seq0 + sq;
    if (seq < a.nseq) gen_put(a, sl, seq, k, a.lmul_mid ? conj(X[sq * N + k]) : X[sq * N + k], X[sq * N + (k ? N - k : 0)]);
  }
}

// ---- pointwise pieces of a flow stage ----------------------------------------------------------------------------------------

// (Fx, Fy) = (i lx F, i ly F)        F layout, grid (blocks, slices)     (src/lenseflow.jl:155, src/specialops.jl:184-188)
template <typename T>
__global__ __launch_bounds__(NTP) void k_gen_lmul2(const cx<T>* __restrict__ F, cx<T>* __restrict__ Fx, cx<T>* __restrict__ Fy,
                                                  const T* __restrict__ lx_r, const T* __restrict__ ly, int Nx, long plane) {
  const long i = (long)blockIdx.x * NTP + threadIdx.x;
  if (i >= plane) return;
  const size_t o = (size_t)blockIdx.y * plane + i;
  const T lx = lx_r[(unsigned)i % (unsigned)Nx], l_y = ly[(unsigned)i / (unsigned)Nx];
  const cx<T> v = F[o];
  Fx[o] = mul_il(v, lx); Fy[o] = mul_il(v, l_y);
}

// velocity k = p_x gx + p_y gy and the RK4 bookkeeping of the Map state (src/lenseflow.jl:150-161, src/numerical_algorithms.jl:15-21)
template <typename T>
__global__ __launch_bounds__(NTP) void k_gen_vel_rk(const T* __restrict__ gx, const T* __restrict__ gy, PhiMaps<T> ph, T* __restrict__ y0,
                                                   T* __restrict__ acc, T* __restrict__ ys, RKCoef<T> rk, long npix, int P, int sl0) {
  const size_t sl = blockIdx.y + (size_t)sl0, pb = (size_t)(ph.Bphi == 1 ? 0 : sl / P) * npix;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < npix; i += (long)gridDim.x * NTP) {
    const size_t o = sl * npix + i;
    T px, py; gen_p(ph, pb + i, rk.t, px, py);
    const T k = px * gx[o] + py * gy[o];
    T y = y0[o], a = rk.stage == 1 ? T(0) : acc[o];
    const T nxt = rk_update(rk, k, y, a);
    if (rk.stage == 4) y0[o] = y; else acc[o] = a;
    ys[o] = nxt;
  }
}

// (Wx, Wy) = (p_x y, p_y y)          (src/lenseflow.jl:166-170)
template <typename T>
__global__ __launch_bounds__(NTP) void k_gen_pmul(const T* __restrict__ y, PhiMaps<T> ph, T t, T* __restrict__ Wx, T* __restrict__ Wy, long npix, int P, int sl0) {
  const size_t sl = blockIdx.y + (size_t)sl0, pb = (size_t)(ph.Bphi == 1 ? 0 : sl / P) * npix;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < npix; i += (long)gridDim.x * NTP) {
    const size_t o = sl * npix + i;
    T px, py; gen_p(ph, pb + i, t, px, py);
    const T v = y[o];
    Wx[o] = px * v; Wy[o] = py * v;
  }
}

// adjoint velocity k = i lx Fx + i ly Fy and the RK4 bookkeeping of the Fourier state (src/lenseflow.jl:163-174)
template <typename T>
__global__ __launch_bounds__(NTP) void k_gen_adj_rk(const cx<T>* __restrict__ Fx, const cx<T>* __restrict__ Fy, const T* __restrict__ lx_r,
                                                   const T* __restrict__ ly, int Nx, cx<T>* __restrict__ Y0, cx<T>* __restrict__ acc,
                                                   cx<T>* __restrict__ Ys, RKCoef<T> rk, long plane, int sl0) {
  const long i = (long)blockIdx.x * NTP + threadIdx.x;
  if (i >= plane) return;
  const size_t o = ((size_t)blockIdx.y + (size_t)sl0) * plane + i;
  const T lx = lx_r[(unsigned)i % (unsigned)Nx], l_y = ly[(unsigned)i / (unsigned)Nx];
  const cx<T> k = mul_il(Fx[o], lx) + mul_il(Fy[o], l_y);
  cx<T> y = Y0[o], a = rk.stage == 1 ? mk<T>(T(0), T(0)) : acc[o];
  const cx<T> nxt = rk_update(rk, k, y, a);
  if (rk.stage == 4) Y0[o] = y; else acc[o] = a;
  Ys[o] = nxt;
}

// pointwise part of a delta-flow stage (src/lenseflow.jl:184-200): products for the delta-f velocity, the f velocity with its RK
// update, and the per-slice spin-adjoint partial products w_k = L(df) d_k f for the end-of-flow delta-phi quadrature
template <typename T>
__global__ __launch_bounds__(NTP) void k_gen_delta(const T* __restrict__ Ldf, const T* __restrict__ gfx, const T* __restrict__ gfy, PhiMaps<T> ph,
                                                  T* __restrict__ Wx, T* __restrict__ Wy, T* __restrict__ w1p, T* __restrict__ w2p,
                                                  T* __restrict__ y0, T* __restrict__ acc, T* __restrict__ ys, RKCoef<T> rk, long npix, int P, int sl0) {
  const size_t sl = blockIdx.y + (size_t)sl0, pb = (size_t)(ph.Bphi == 1 ? 0 : sl / P) * npix;
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < npix; i += (long)gridDim.x * NTP) {
    const size_t o = sl * npix + i;
    T px, py; gen_p(ph, pb + i, rk.t, px, py);
    const T l = Ldf[o], ax = gfx[o], ay = gfy[o];
    Wx[o] = px * l; Wy[o] = py * l;
    w1p[o] = l * ax; w2p[o] = l * ay;
    const T k = px * ax + py * ay;
    T y = y0[o], a = rk.stage == 1 ? T(0) : acc[o];
    const T nxt = rk_update(rk, k, y, a);
    if (rk.stage == 4) y0[o] = y; else acc[o] = a;
    ys[o] = nxt;
  }
}

}  // namespace cmbl
