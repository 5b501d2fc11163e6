// Any-size path, compile-time plans: batched 1-D DFTs of the lengths N = 2^a 3^b 5^c listed in CMBL_CT_LIST (the common survey patch
// sides 3 * 2^k and 5 * 2^k, and the sizes of the reference-style tables: 360, 1000), as a drop-in for k_gen_dft_mr (kernels_generic.hpp)
// behind the same GenDft argument block -- same element order in and out (natural), same fetch / store options, same results to rounding.
//
// What is different from the run-time-planned kernel (whose launches are ~1 TB/s: a fetch loop that exposes one memory latency per
// element, five barrier-separated stages with run-time divisions, 45 % LDS bank conflicts; profiles/r04_anysize_stamps_and_ilp_rejected.txt):
//   * the plan is a template parameter: power-of-two radices up to 16 first (register butterflies of fft_core.hpp, one-read stage
//     twiddles), then the 3s and 5s, as a Stockham autosort network IN PLACE -- every index is a shift, a mask or a constant division;
//   * ONE WAVEFRONT owns a sequence: a lane pulls all of its butterflies of a stage into registers and writes them back; LDS operations
//     of a wave execute in order, so every read of a stage precedes every write of it and no workgroup barrier separates the stages
//     (the WorkRows idea of the fused row kernels).  S = 64 bytes / sizeof(element) sequences per workgroup = as many wavefronts;
//   * every thread issues ALL of its global loads (up to 12 elements with all their operands) before the first LDS store: one exposed
//     memory latency per workgroup instead of one per element.  The fetch variants are separate straight-line instantiations (KIND)
//     selected by one uniform switch at the top of the kernel;
//   * padded LDS rows (fft_lds.hpp pad()), row stride = 8 (mod 32) slots so that the transposed side of a y pass spreads over the banks;
//   * workgroups that are neighbours in the strided direction share 128-byte lines: they are mapped to the same XCD (xcd_tile);
//   * (late round 6) the half planes the column and row launches of the fused flow stages hand to each other can be TILED, [x / 4][ky][x % 4]
//     (GenDft::in_tiled / out_tiled, CtSide below): a column workgroup reads and writes contiguous blocks, a row workgroup whole or half
//     128-byte lines; every fetch / store loop runs on one offset per thread + a scalar step per element (ct_map0, CtSide).
#pragma once
#include "kernels_generic.hpp"

// (two halves: the launches of a length are compiled by the translation units tu_cty*_{a,b} / tu_ctx*_{a,b} -- engine_ct.hpp -- so that the build is as
//  long as half of the list; 576 = 9 * 2^6, 1152, 2304, 2560 = 5 * 2^9 and 3072 = 3 * 2^10 since late round 6 -- the last three in half-size groups, ct_Smax: 8
//  rows of them exceed the LDS; their delta stages run in quarter-width column groups, ct_S2)
#ifndef CMBL_CT_LIST_A
#define CMBL_CT_LIST_A(X) X(96) X(160) X(192) X(320) X(360) X(384) X(480) X(576) X(640) X(720) X(768) X(960) X(1000)
#endif
#ifndef CMBL_CT_LIST_B
#define CMBL_CT_LIST_B(X) X(1152) X(1280) X(1536) X(1920) X(2304) X(2560) X(3072)
#endif
#define CMBL_CT_LIST(X) CMBL_CT_LIST_A(X) CMBL_CT_LIST_B(X)

namespace cmbl {

constexpr int ct_count(int N, int p) { int c = 0; while (N % p == 0) { N /= p; ++c; } return c; }
constexpr int ct_np2(int N) { return num_stages(ct_count(N, 2)); }                                   // power-of-two stages
constexpr int ct_nstages(int N) { return ct_np2(N) + ct_count(N, 3) + ct_count(N, 5); }
constexpr int ct_radix(int N, int i) {
  if (i < ct_np2(N)) return 1 << stage_lg(ct_count(N, 2), i);
  return i - ct_np2(N) < ct_count(N, 3) ? 3 : 5;
}
constexpr int ct_ns(int N, int i) { int ns = 1; for (int k = 0; k < i; ++k) ns *= ct_radix(N, k); return ns; }   // product of the earlier radices
constexpr int ct_ld(int N) { return pad(N) + ((8 - pad(N) % 32) + 32) % 32; }                        // row stride, 8 (mod 32) slots
template <typename T> constexpr int ct_S() { return 64 / (int)sizeof(cx<T>); }                       // sequences (= wavefronts) per workgroup
template <typename T> constexpr size_t ct_lds(int N, int rowsets = 1, int S = ct_S<T>()) { return ((size_t)(N / 2) + (size_t)rowsets * S * ct_ld(N)) * sizeof(cx<T>); }
// the most sequences per workgroup whose rows fit the LDS: ct_S up to 1920 points; 2304 and 3072 take half groups in every launch
template <typename T> constexpr int ct_Smax(int N) { return ct_lds<T>(N) <= 160 * 1024 ? ct_S<T>() : ct_S<T>() / 2; }
// the fused row update (k_ct_adj_x: a whole row AND its RK operands in registers) spills 600-800 bytes per lane in double precision beyond 1920 points:
// those shapes run the x transform and the RK update as launches of their own (Ctx::gen_ct_x)
template <typename T> constexpr bool ct_rowfuse_ok(int N) { return !(sizeof(T) == 8 && N > 1920); }
// columns per workgroup of the delta-stage kernel (two LDS rows per column): the usual count, or half of it where that does not fit (Ny > ~1150;
// the transposed side then moves 32-byte pieces)
#ifndef CMBL_CT_S2_HALF
#define CMBL_CT_S2_HALF 0      // 1: always half (A/B: shorter transform phases, 32-byte pieces on the transposed side)
#endif
// (a quarter from 2304 points on: 2 columns in single, 1 in double precision)
template <typename T> constexpr int ct_S2(int N) {
  return (!CMBL_CT_S2_HALF && ct_lds<T>(N, 2) <= 160 * 1024) ? ct_S<T>() : ct_lds<T>(N, 2, ct_S<T>() / 2) <= 160 * 1024 ? ct_S<T>() / 2 : ct_S<T>() / 4;
}

// fetch variants
enum { CT_C = 0, CT_R1, CT_R2, CT_H1, CT_H2, CT_P1, CT_P2, CT_P3 };
template <typename T> inline int ct_kind(const GenDft<T>& a) {
  if (a.in_real) return a.pro.mode == 1 ? CT_P1 : a.pro.mode == 2 ? CT_P2 : a.pro.mode == 3 ? CT_P3 : (a.in2 ? CT_R2 : CT_R1);
  if (a.herm) return a.in2 ? CT_H2 : CT_H1;
  return CT_C;
}

// r-point butterflies on register values; output X_m is left at v[ct_loc<R>(m)]
template <int R> constexpr int ct_loc(int m) { return R == 16 ? dft_loc<4>(m) : R == 8 ? dft_loc<3>(m) : m; }
template <typename T, int R, typename V> __device__ __forceinline__ void ct_bfly(V (&v)[R]) {
  if constexpr (R == 16) dft<T, 4, false>(v);
  else if constexpr (R == 8) dft<T, 3, false>(v);
  else if constexpr (R == 4) dft<T, 2, false>(v);
  else if constexpr (R == 2) dft<T, 1, false>(v);
  else if constexpr (R == 3) {
    const V t = vadd(v[1], v[2]), m = vsub(v[0], vscale(t, T(0.5))), sd = vscale(vsub(v[1], v[2]), T(0.86602540378443864676));
    v[0] = vadd(v[0], t); v[1] = vsubi(m, sd); v[2] = vaddi(m, sd);
  } else {
    static_assert(R == 5, "radices 2, 4, 8, 16, 3, 5");
    constexpr T c1 = T(0.30901699437494742410), c2 = T(-0.80901699437494742410), s1 = T(0.95105651629515357212), s2 = T(0.58778525229247312917);
    const V t1 = vadd(v[1], v[4]), t2 = vadd(v[2], v[3]), t3 = vsub(v[1], v[4]), t4 = vsub(v[2], v[3]);
    const V a1 = vadd(v[0], vadd(vscale(t1, c1), vscale(t2, c2))), a2 = vadd(v[0], vadd(vscale(t1, c2), vscale(t2, c1)));
    const V b1 = vadd(vscale(t3, s1), vscale(t4, s2)), b2 = vsub(vscale(t3, s2), vscale(t4, s1));
    v[0] = vadd(v[0], vadd(t1, t2));
    v[1] = vsubi(a1, b1); v[4] = vaddi(a1, b1); v[2] = vsubi(a2, b2); v[3] = vaddi(a2, b2);
  }
}

// W_N^i, i < N, from the table of the first half of the circle (N even): W^(i + N/2) = -W^i
template <typename T, int N> __device__ __forceinline__ typename vreg<T>::type ct_tw(const cx<T>* __restrict__ tw, int i) {
  const bool hi = i >= N / 2;
  return vscale(vload(tw + (hi ? i - N / 2 : i)), hi ? T(-1) : T(1));
}

// Stage I of the plan of N on the sequence at s (one wavefront; lane = its lane id), in place:
//   v[m] = X[j + m N/R] W_{Ns R}^{m k},  v <- DFT_R(v),  X[(j div Ns) Ns R + k + m Ns] = v[m]      (j < N/R, k = j mod Ns)
template <typename T, int N, int I>
__device__ __forceinline__ void ct_stage(cx<T>* __restrict__ s, const cx<T>* __restrict__ tw, int lane) {
  using V = typename vreg<T>::type;
  constexpr int R = ct_radix(N, I), Ns = ct_ns(N, I), nb = N / R, tstep = N / (Ns * R), B = (nb + 63) / 64;
  // LDS addresses as ONE per-butterfly base + compile-time offsets wherever the padding allows it (a quarter of the ~600 vector instructions of a
  // 768-point transform were pad() arithmetic, one add-shift-add per element and direction): pad(a + c) == pad(a) + pad(c) when c is a multiple
  // of 16, or when a mod 16 + c mod 16 cannot carry -- loads: c = m nb; stores: c = m Ns with k < Ns, Ns | 16, 16 | Ns R
  constexpr bool LD_IMM = nb % 16 == 0, ST_IMM = Ns % 16 == 0 || (16 % Ns == 0 && (Ns * R) % 16 == 0);
  V v[B][R];
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const int j0 = lane + 64 * b, j = (nb % 64 == 0 || j0 < nb) ? j0 : nb - 1;       // spare lanes re-read the last butterfly (and write nothing)
    const cx<T>* p = s + pad(j);
#pragma unroll
    for (int m = 0; m < R; ++m) v[b][m] = vload(LD_IMM ? p + pad(m * nb) : s + pad(j + m * nb));
  }
  // The stage is IN PLACE with no workgroup barrier: lanes read slots that other lanes of the wave overwrite below.  LDS operations of a wave
  // execute in issue order, so what has to hold is that every load above is ISSUED before the first store below -- made explicit here (a
  // scheduling fence; nothing crosses it) instead of resting on the compiler's inability to prove the accesses disjoint
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int b = 0; b < B; ++b) {
    const int j = lane + 64 * b;
    const int blk = j / Ns, k = j - blk * Ns;
    if constexpr (Ns > 1) {
      // External twiddles W^(m k tstep).  Radices 3 and 5: every one a table read.  Powers of two: W^1 and W^(4q) are table reads, the
      // rest ONE product each (W^2 = W^1 W^1, W^3 = W^1 W^2, W^(4q+r) = W^(4q) W^r) -- the angle error of a power grows with the exponent
      // when everything is derived from W^1 alone (15 roundings of W^1 in W^15: single-precision transforms 2-4 x less accurate than
      // with table twiddles, measured at 768^2), and R/4 + 1 reads per butterfly are still far from R - 1.
      V w[R];
      const int kt = k * tstep;
      if constexpr (R == 3 || R == 5) {
#pragma unroll
        for (int m = 1; m < R; ++m) w[m] = ct_tw<T, N>(tw, m * kt);
      } else {
        w[1] = vload(tw + kt);                                                        // kt < N / R
        if constexpr (R >= 4) { w[2] = vmul(w[1], w[1]); w[3] = vmul(w[1], w[2]); }
#pragma unroll
        for (int q = 1; q < R / 4; ++q) {
          w[4 * q] = ct_tw<T, N>(tw, 4 * q * kt);
#pragma unroll
          for (int r = 1; r < 4; ++r) w[4 * q + r] = vmul(w[4 * q], w[r]);
        }
      }
#pragma unroll
      for (int m = 1; m < R; ++m) v[b][m] = vmul(v[b][m], w[m]);
    }
    ct_bfly<T, R>(v[b]);
    if (nb % 64 == 0 || j < nb) {
      cx<T>* q = s + pad(blk * (Ns * R) + k);
#pragma unroll
      for (int m = 0; m < R; ++m) vstore(ST_IMM ? q + pad(m * Ns) : s + pad(blk * (Ns * R) + k + m * Ns), v[b][ct_loc<R>(m)]);
    }
    // many values in flight: keep the compiler from interleaving the butterflies of a lane (their twiddles and temporaries would all
    // be live at once: spills at 720 / 1000 / 1280 under the two-workgroups-per-CU register cap)
    if constexpr (B * R > 24) __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
template <typename T, int N, int I = 0>
__device__ __forceinline__ void ct_transform(cx<T>* __restrict__ s, const cx<T>* __restrict__ tw, int lane) {
  if constexpr (I < ct_nstages(N)) {
    ct_stage<T, N, I>(s, tw, lane);
    ct_transform<T, N, I + 1>(s, tw, lane);
  }
}

// ---- fetch: operands of CH elements into registers (load), then their values (value) ------------------------------------------------
template <typename T, int CH> struct CtRegs { cx<T> u[CH], w[CH]; T r[7][CH]; };

// element i of the chunk: sequence seq (clamped to a valid one by the caller), entry n < N.  Addresses are a wave-uniform base (the
// slice) + ONE 32-bit element offset shared by all arrays of the element (at32, kernels_fft.hpp): no 64-bit vector arithmetic, one
// address register per element in flight.
template <typename T, int KIND, bool FLAG /*H: lmul_in given; P: p(t) from the cache*/, int CH>
__device__ __forceinline__ void ct_load(CtRegs<T, CH>& g, int i, const GenDft<T>& a, size_t sl, int n, unsigned o /*offset of (seq, n)*/, unsigned Fs, unsigned mn) {
  const size_t sb = sl * a.in_slice;
  if constexpr (KIND == CT_C) g.u[i] = at32(reinterpret_cast<const cx<T>*>(a.in) + sb, o);
  else if constexpr (KIND == CT_R1 || KIND == CT_R2) {
    g.r[0][i] = at32(reinterpret_cast<const T*>(a.in) + sb, o);
    if constexpr (KIND == CT_R2) g.r[1][i] = at32(reinterpret_cast<const T*>(a.in2) + sb, o);
  } else if constexpr (KIND == CT_H1 || KIND == CT_H2) {
    const int m = n < a.nin ? n : a.N - n;
    const unsigned oh = Fs + (unsigned)m * mn;                            // (half spectra are fetched along ky: never the tiled x index, G is linear)
    g.u[i] = at32(reinterpret_cast<const cx<T>*>(a.in) + sb, oh);
    if constexpr (KIND == CT_H2) g.w[i] = at32(reinterpret_cast<const cx<T>*>(a.in2) + sb, oh);
    if constexpr (FLAG) g.r[0][i] = at32(a.lmul_in, (unsigned)m);
  } else {
    const GenPro<T>& e = a.pro;                                          // real maps [slice][npix]: in_slice == npix
    const size_t pb = (size_t)(e.ph.Bphi == 1 ? 0 : sl / e.P) * e.npix;
    if constexpr (FLAG) { g.r[0][i] = at32(e.ph.pcx + pb, o); g.r[1][i] = at32(e.ph.pcy + pb, o); }
    else {
      g.r[0][i] = at32(e.ph.gx + pb, o); g.r[1][i] = at32(e.ph.gy + pb, o);
      g.u[i] = mk<T>(at32(e.ph.hxx + pb, o), at32(e.ph.hyx + pb, o)); g.w[i].x = at32(e.ph.hyy + pb, o);
    }
    if constexpr (KIND == CT_P2 || KIND == CT_P3) g.r[2][i] = at32(e.Ldf + sb, o);
    if constexpr (KIND != CT_P3) {
      g.r[3][i] = at32(e.gx + sb, o); g.r[4][i] = at32(e.gy + sb, o); g.r[5][i] = at32(e.y0 + sb, o);
      g.r[6][i] = at32(e.acc + sb, o);                                   // (not read by stage 1: whatever the buffer holds)
    }
  }
}
// Hermitian extension with FFTW's c2r rule (gen_herm)
template <typename T> __device__ __forceinline__ cx<T> ct_herm(cx<T> v, int n, int nin, int N) {
  if (n == 0 || 2 * n == N) v.y = T(0);
  return n < nin ? v : conj(v);
}
// The pointwise kinds also produce what the stage writes back to memory (so[0..1]: the delta-phi products, so[2]: the new y0 / acc); the
// caller stores them AFTER its last load has returned -- loads and stores share one in-order counter (vmcnt), so a store between two
// chunks of loads makes the second chunk wait for the store's acknowledgement (measured: 14k cycles of fetch instead of 4k).
template <typename T, int KIND, bool FLAG, int CH>
__device__ __forceinline__ cx<T> ct_value(const CtRegs<T, CH>& g, int i, const GenDft<T>& a, int n, T (&so)[3]) {
  cx<T> v;
  if constexpr (KIND == CT_C) v = g.u[i];
  else if constexpr (KIND == CT_R1) v = mk<T>(g.r[0][i], T(0));
  else if constexpr (KIND == CT_R2) v = mk<T>(g.r[0][i], g.r[1][i]);
  else if constexpr (KIND == CT_H1) {
    v = g.u[i];
    if constexpr (FLAG) v = mul_il(v, g.r[0][i]);
    v = ct_herm(v, n, a.nin, a.N);
  } else if constexpr (KIND == CT_H2) {
    cx<T> w = g.w[i];
    if constexpr (FLAG) w = mul_il(w, g.r[0][i]);
    const cx<T> u = ct_herm(g.u[i], n, a.nin, a.N);
    w = ct_herm(w, n, a.nin, a.N);
    v = mk<T>(u.x - w.y, u.y + w.x);
  } else {
    const GenPro<T>& e = a.pro;
    T px, py;
    if constexpr (FLAG) { px = g.r[0][i]; py = g.r[1][i]; }
    else { T m11, m12, m22; flow_pm(e.rk.t, g.r[0][i], g.r[1][i], g.u[i].x, g.u[i].y, g.w[i].x, px, py, m11, m12, m22); }
    if constexpr (KIND == CT_P3) v = mk<T>(px * g.r[2][i], py * g.r[2][i]);
    else {
      const T ax = g.r[3][i], ay = g.r[4][i];
      if constexpr (KIND == CT_P2) { so[0] = g.r[2][i] * ax; so[1] = g.r[2][i] * ay; }
      const T k = px * ax + py * ay;
      T y = g.r[5][i], ac = e.rk.stage == 1 ? T(0) : g.r[6][i];
      const T nxt = rk_update(e.rk, k, y, ac);
      so[2] = e.rk.stage == 4 ? y : ac;
      v = mk<T>(nxt, T(0));
    }
  }
  return a.inverse ? conj(v) : v;
}

// elements a thread requests before it consumes the first: as many as fit a budget of operand registers (in units of T)
#ifndef CMBL_CT_FETCH_WORDS
#define CMBL_CT_FETCH_WORDS 64
#endif
constexpr int ct_words(int kind, bool flag) {
  return kind == CT_C ? 2 : kind == CT_R1 ? 1 : kind == CT_R2 ? 2 : kind == CT_H1 ? 2 + flag : kind == CT_H2 ? 4 + flag
       : (flag ? 2 : 5) + (kind == CT_P1 ? 4 : kind == CT_P2 ? 5 : 1);
}
constexpr int ct_chunk(int E, int kind, bool flag) {
  int ch = CMBL_CT_FETCH_WORDS / ct_words(kind, flag);
  ch = ch < 2 ? 2 : ch;
  ch = ch > E ? E : ch;
  const int nch = (E + ch - 1) / ch;
  return (E + nch - 1) / nch;
}
// thread -> (sequence sq of the workgroup, first element nb); its j-th element is nb + 64 j in every mode.  mode 0: a wavefront walks its own
// sequence; 1: consecutive threads walk the S sequences at one element (transposed side: S elements = one 64-byte piece, or -- tiled arrays --
// whole contiguous blocks); 2: tiled ROWS (x kernels on tiled arrays): consecutive threads walk the 4 x of a block row, then the S rows, then the
// blocks -- 4 rows x 4 x = one 128-byte line
template <int S> __device__ __forceinline__ void ct_map0(int mode, int tid, int& sq, int& nb) {
  constexpr int LGS = ilog2c(S);
  const int sh = mode == 2 ? 2 : 0, mk = mode == 2 ? 3 : 0;              // (uniform selects; modes 1 and 2 share one form)
  sq = mode ? ((tid >> sh) & (S - 1)) : (tid >> 6);
  nb = mode ? (((tid >> (sh + LGS)) << sh) + (tid & mk)) : (tid & 63);
}
// One side's addressing as scalars.  Strided arrays: off(seq, n) = seq sseq + n selem; tiled arrays (GenDft::in_tiled): ((x >> 2) np + ky) 4 + (x & 3)
// with x the sequence (1) or the element (2) index.  Either way off(seq, n) = F(seq) + G(n) and G(n + 64 j) = G(n) + j step -- a thread's elements
// are ONE offset + multiples of a scalar (the index arithmetic of a general (seq, n) per element cost the fused stages 5 % of their time)
struct CtSide {
  unsigned sh_s, mk_s, ms, sh_n, mk_n, mn, step;
  __device__ __forceinline__ CtSide(int tiled, int np, long sseq, long selem) {
    const bool t1 = tiled == 1, t2 = tiled == 2;
    sh_s = t1 ? 2u : 0u; mk_s = t1 ? 3u : 0u; ms = t1 ? 4u * (unsigned)np : (t2 ? 4u : (unsigned)sseq);
    sh_n = t2 ? 2u : 0u; mk_n = t2 ? 3u : 0u; mn = t2 ? 4u * (unsigned)np : (t1 ? 4u : (unsigned)selem);
    step = t2 ? 16u * mn : 64u * mn;
  }
  __device__ __forceinline__ unsigned F(int seq) const { return ((unsigned)seq >> sh_s) * ms + ((unsigned)seq & mk_s); }
  __device__ __forceinline__ unsigned G(int n) const { return ((unsigned)n >> sh_n) * mn + ((unsigned)n & mk_n); }
};
template <typename T, int N, int KIND, bool FLAG, int S = ct_S<T>()>
__device__ __forceinline__ void ct_fetch(const GenDft<T>& a, cx<T>* __restrict__ s, size_t sl, int seq0, int by_seq, int tid = threadIdx.x) {
  constexpr int LD = ct_ld(N), E = (N + 63) / 64, CH = ct_chunk(E, KIND, FLAG), NCH = (E + CH - 1) / CH;
  constexpr bool WB = KIND == CT_P1 || KIND == CT_P2;                    // kinds that write back to memory
  constexpr int WBN = E <= 16 ? NCH : 1;                                 // chunks whose write-backs are held back (all of them up to 16 elements per thread)
  int sq, nb;
  ct_map0<S>(by_seq, tid, sq, nb);
  const int seq = min(seq0 + sq, a.nseq - 1);
  const CtSide sd(a.in_tiled, a.tile_np, a.in_seq, a.in_elem);
  const unsigned Fs = sd.F(seq), base = Fs + sd.G(nb);
  // element j of the thread: n0 = nb + 64 j; only the last one can lie beyond the sequence (clamped: the general form)
  auto off = [&](int j, int n) { return (N % 64 == 0 || j < E - 1) ? base + (unsigned)j * sd.step : Fs + sd.G(n); };
  T so[WB ? WBN * CH : 1][3];
  auto write_back = [&](int j0, int cnt_base) {                          // elements j0 .. j0 + WBN * CH - 1 of the thread
    if constexpr (WB) {
      const GenPro<T>& e = a.pro;
      const size_t sb = sl * a.in_slice;
      T* const dst = (e.rk.stage == 4 ? e.y0 : e.acc) + sb;
#pragma unroll
      for (int jj = 0; jj < WBN * CH; ++jj) {
        const int j = j0 + jj, n0 = nb + 64 * j;
        if (n0 < N && j < E && seq0 + sq < a.nseq) {                     // the element exists: only then anything is written
          const unsigned o = base + (unsigned)j * sd.step;
          if constexpr (KIND == CT_P2) { at32(e.w1p + sb, o) = so[jj][0]; at32(e.w2p + sb, o) = so[jj][1]; }
          at32(dst, o) = so[jj][2];
        }
      }
    }
    (void)cnt_base;
  };
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    CtRegs<T, CH> g;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int j = c * CH + i, n0 = nb + 64 * j, n = n0 < N ? n0 : N - 1;
      ct_load<T, KIND, FLAG, CH>(g, i, a, sl, n, off(j, n), Fs, sd.mn);
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int j = c * CH + i, n0 = nb + 64 * j, n = n0 < N ? n0 : N - 1;
      const cx<T> v = ct_value<T, KIND, FLAG, CH>(g, i, a, n, so[WB ? (WBN > 1 ? c * CH + i : i) : 0]);
      if (n0 < N && j < E) s[sq * LD + pad(n0)] = v;
    }
    if constexpr (WB && WBN == 1) write_back(c * CH, 0);
  }
  if constexpr (WB && WBN > 1) write_back(0, 0);
}

// gen_put (kernels_generic.hpp) with the addressing of the fetch: wave-uniform slice base in scalar registers + one 32-bit element
// offset (the 64-bit multiplies of a general index cost the store phase 4.4k cycles per launch: 12 elements per thread)
template <int BYTES> __device__ __forceinline__ void ct_store(void* sbase, unsigned byte_off, const void* v, bool wt) {
#if CMBL_WT_GEN
  if (wt) {
    if constexpr (BYTES == 16) { const wt_f4 d = *reinterpret_cast<const wt_f4*>(v); asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" : : "v"(byte_off), "v"(d), "s"(sbase) : "memory"); }
    else if constexpr (BYTES == 8) { const wt_f2 d = *reinterpret_cast<const wt_f2*>(v); asm volatile("global_store_dwordx2 %0, %1, %2 sc1" : : "v"(byte_off), "v"(d), "s"(sbase) : "memory"); }
    else { const float d = *reinterpret_cast<const float*>(v); asm volatile("global_store_dword %0, %1, %2 sc1" : : "v"(byte_off), "v"(d), "s"(sbase) : "memory"); }
    return;
  }
#endif
  char* q = reinterpret_cast<char*>(sbase) + byte_off;
  if constexpr (BYTES == 16) *reinterpret_cast<wt_f4*>(q) = *reinterpret_cast<const wt_f4*>(v);
  else if constexpr (BYTES == 8) *reinterpret_cast<wt_f2*>(q) = *reinterpret_cast<const wt_f2*>(v);
  else *reinterpret_cast<float*>(q) = *reinterpret_cast<const float*>(v);
}
template <typename T>
__device__ __forceinline__ void ct_put(const GenDft<T>& a, size_t sl, unsigned o /*offset of (seq, k)*/, int k, cx<T> y, cx<T> yr) {
  if (a.inverse) y = conj(y);
  const size_t sb = sl * a.out_slice;
  const bool wt = wt_line<T>(a.N);
  if (a.out_real) {
    const T v1 = a.scale * y.x;
    ct_store<(int)sizeof(T)>(reinterpret_cast<T*>(a.out) + sb, o * (unsigned)sizeof(T), &v1, wt);
    if (a.out2) { const T v2 = a.scale2 * y.y; ct_store<(int)sizeof(T)>(reinterpret_cast<T*>(a.out2) + sb, o * (unsigned)sizeof(T), &v2, wt); }
    return;
  }
  if (a.in_real && a.in2) {                                             // split the transform of in + i in2
    const cx<T> c = conj(yr);
    const cx<T> x1 = mk<T>(T(0.5) * (y.x + c.x), T(0.5) * (y.y + c.y)), d = mk<T>(T(0.5) * (y.x - c.x), T(0.5) * (y.y - c.y));
    const cx<T> o1 = mk<T>(a.scale * x1.x, a.scale * x1.y), o2 = mk<T>(a.scale2 * d.y, -a.scale2 * d.x);              // d / i
    ct_store<(int)sizeof(cx<T>)>(reinterpret_cast<cx<T>*>(a.out) + sb, o * (unsigned)sizeof(cx<T>), &o1, wt);
    ct_store<(int)sizeof(cx<T>)>(reinterpret_cast<cx<T>*>(a.out2) + sb, o * (unsigned)sizeof(cx<T>), &o2, wt);
    return;
  }
  if (a.lmul_out) y = mul_il(y, a.lmul_out[k]);
  const cx<T> o1 = mk<T>(a.scale * y.x, a.scale * y.y);
  ct_store<(int)sizeof(cx<T>)>(reinterpret_cast<cx<T>*>(a.out) + sb, o * (unsigned)sizeof(cx<T>), &o1, wt);
}

// Stores: all LDS reads of the thread first, then the global stores (a read-store loop waits for LDS once per element)
template <typename T, int N, int S = ct_S<T>()>
__device__ __forceinline__ void ct_store_rows(const GenDft<T>& a, const cx<T>* __restrict__ s, size_t sl, int seq0, int out_by_seq, int wave, int lane, bool mid, int tid = threadIdx.x) {
  constexpr int LD = ct_ld(N);
  const bool split = a.in_real && a.in2;
  constexpr int E = (N + 63) / 64, NPC = (E + 11) / 12, PCH = (E + NPC - 1) / NPC;
  int sq, kb;
  ct_map0<S>(out_by_seq, tid, sq, kb);
  const int seq = seq0 + sq;
  const CtSide sd(a.out_tiled, a.tile_np, a.out_seq, a.out_elem);          // (real maps are never tiled: out_tiled == 0 with out_real)
  const unsigned base = sd.F(seq) + sd.G(kb);
  const cx<T>* p = s + sq * LD;
#pragma unroll
  for (int c = 0; c < NPC; ++c) {
    cx<T> y[PCH], yr[PCH];
#pragma unroll
    for (int ii = 0; ii < PCH; ++ii) {
      const int k = min(kb + 64 * (c * PCH + ii), a.nout - 1);
      y[ii] = p[pad(k)];
      yr[ii] = p[pad(split && k ? N - k : 0)];                           // Z[N - k]: only the pair split reads it
    }
#pragma unroll
    for (int ii = 0; ii < PCH; ++ii) {
      const int i = c * PCH + ii, k = kb + 64 * i;
      if (i < E && k < a.nout && seq < a.nseq) ct_put(a, sl, base + (unsigned)i * sd.step, k, mid ? conj(y[ii]) : y[ii], yr[ii]);
    }
  }
  (void)wave; (void)lane;
}

// Fused y passes of a forward flow stage, on the wavefront's own column `seq` (row = its LDS row, holding the fetched pair
// conj(ext(Gx) + i ext(i ly A))):  inverse transform -> (d/dx f, d/dy f) at the column's pixels -> velocity p . grad f and the RK4
// bookkeeping (src/lenseflow.jl:150-161, src/numerical_algorithms.jl:15-21; y0 / acc in memory like k_gen_vel_rk) -> the next stage
// input as a real sequence -> forward transform.  Everything between the two workgroup barriers of the launch is wave-private.
template <typename T, int N>
__device__ __forceinline__ void ct_flow_stage(const GenDft<T>& a, cx<T>* __restrict__ row, const cx<T>* __restrict__ tw, size_t sl, int seq, int lane) {
  constexpr int E = (N + 63) / 64;
  const GenPro<T>& e = a.pro;
  const size_t mb = sl * (size_t)e.npix, pb = (size_t)(e.ph.Bphi == 1 ? 0 : sl / e.P) * e.npix;
  const bool pc = e.ph.pcx != nullptr;
  // the pixels' operands: requested before the transform so that they arrive under it (up to 12 pixels per lane, p(t) from the cache;
  // longer columns and the five-map form request them after it: the registers are the transform's)
  constexpr bool EARLY = E <= 12;
  constexpr int CHP = EARLY ? E : 8, NCP = (E + CHP - 1) / CHP;            // pixels per chunk of the late form (the operands of a chunk in flight together)
  T px[CHP], py[CHP], y0v[CHP], acv[CHP];
  if constexpr (EARLY) {
#pragma unroll
    for (int i = 0; i < E; ++i) {
      const int n = min(lane + 64 * i, N - 1);
      const unsigned o = (unsigned)seq * (unsigned)N + (unsigned)n;
      y0v[i] = at32(e.y0 + mb, o); acv[i] = at32(e.acc + mb, o);
      if (pc) { px[i] = at32(e.ph.pcx + pb, o); py[i] = at32(e.ph.pcy + pb, o); }
    }
  }
  ct_transform<T, N>(row, tw, lane);
#pragma unroll
  for (int c = 0; c < NCP; ++c) {
    if constexpr (!EARLY) {
      // long columns (> 12 pixels per lane; 30 at 1920 points): the operands of ALL pixels held at once spilled 100+ registers -- chunks of 8,
      // kept apart by a scheduling fence
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ii = 0; ii < CHP; ++ii) {
        const int n = min(lane + 64 * (c * CHP + ii), N - 1);
        const unsigned o = (unsigned)seq * (unsigned)N + (unsigned)n;
        y0v[ii] = at32(e.y0 + mb, o); acv[ii] = at32(e.acc + mb, o);
        if (pc) { px[ii] = at32(e.ph.pcx + pb, o); py[ii] = at32(e.ph.pcy + pb, o); }
      }
    }
#pragma unroll
    for (int ii = 0; ii < CHP; ++ii) {
      const int i = c * CHP + ii, n0 = lane + 64 * i, n = min(n0, N - 1);
      if (i < E) {
        const unsigned o = (unsigned)seq * (unsigned)N + (unsigned)n;
        if (!pc) {
          T m11, m12, m22;
          flow_pm(e.rk.t, at32(e.ph.gx + pb, o), at32(e.ph.gy + pb, o), at32(e.ph.hxx + pb, o), at32(e.ph.hyx + pb, o), at32(e.ph.hyy + pb, o), px[ii], py[ii], m11, m12, m22);
        }
        const cx<T> z = row[pad(n)];                                       // the e^{+i} transform is conj(forward(conj .)): y = conj(z)
        const T gx = a.scale * z.x, gy = -a.scale2 * z.y;
        const T k = px[ii] * gx + py[ii] * gy;
        T y = y0v[ii], ac = e.rk.stage == 1 ? T(0) : acv[ii];
        const T nxt = rk_update(e.rk, k, y, ac);
        if (N % 64 == 0 || n0 < N) {
          if (e.rk.stage == 4) at32(e.y0 + mb, o) = y; else at32(e.acc + mb, o) = ac;
          row[pad(n)] = mk<T>(nxt, T(0));
        }
      }
    }
  }
  if (a.yy_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  ct_transform<T, N>(row, tw, lane);
}

// debug builds (-DCMBL_STAMPS -DCMBL_STAMPS_ROWS -DCMBL_STAMPS_CT): phase timestamps of the launches whose kind (+ 8 for the d/dx pass)
// equals CMBL_CT_STAMP_KIND, read back with cmbl_debug_stamps (tools/gpu_stamps_ct.py)
#ifdef CMBL_STAMPS_CT
#define CMBL_CT_STAMP(i) do { if (stamp && threadIdx.x == 0) g_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = (i) >= 14 ? wall_clock64() : clock64(); } while (0)
#else
#define CMBL_CT_STAMP(i) do {} while (0)
#endif

#ifndef CMBL_DY_PROBE
#define CMBL_DY_PROBE 0      // timing-only probes of k_ct_delta_y (results wrong): 1 no transforms, 2 no product stores, 4 no operand loads, 8 no final stores
#endif
template <typename T> constexpr int ct_min_waves() { return sizeof(T) == 4 ? 4 : 2; }       // two workgroups per CU
// ... of the x-pass kernels: two workgroups of S wavefronts per CU, and ONE from 1000 points on -- 16-24 elements per lane under the 128-register
// cap spilled 80-170 registers (1000 = 8 5 5 5: 300 bytes of scratch per lane in the d/dx pass), while a launch of <= 256 row groups has one
// workgroup per CU whatever the cap allows
template <typename T, int N, int S> constexpr int ct_min_waves_x() {
  const int w = ct_min_waves<T>() * S / ct_S<T>() / (N >= 1000 ? 2 : 1);
  return w < 1 ? 1 : w;
}

template <typename T, int N, bool CONLY = false /*complex in, complex out only (x passes): no other fetch variant is compiled in*/, int S = ct_S<T>()>
__device__ __forceinline__ void ct_dft_body(const GenDft<T>& a, int kind, unsigned ysl, int ngroups = 0) {
  constexpr int LGS = ilog2c(S), NT = 64 * S, LD = ct_ld(N), NTW = N / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + NTW;
  const int in_by_seq = a.in_tiled == 2 ? 2 : (a.in_elem != 1 ? 1 : 0), out_by_seq = a.out_tiled == 2 ? 2 : (a.out_elem != 1 ? 1 : 0);   // ct_map modes
  // (a transposed or tiled side: neighbouring groups share lines -- the same XCD's L2 should see both; the host rounds the grid of the tiled x
  //  launches up to a multiple of 8 for it, hence the exit)
  // (ngroups: the groups of THIS transform where the launch has more -- the second part of k_ct_adj_x_dx: the XCD ranges are those of its own count)
  if (ngroups && (int)blockIdx.x >= ngroups) return;
  const int seq0 = ((in_by_seq || out_by_seq) ? xcd_tile(blockIdx.x, ngroups ? ngroups : (int)gridDim.x) : (int)blockIdx.x) * S;
  if (seq0 >= a.nseq) return;
  const size_t sl = gen_slice(a, ysl);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef CMBL_STAMPS_CT
  const bool stamp = (kind >> 8) != 0;
  kind &= 255;
#endif
  CMBL_CT_STAMP(14); CMBL_CT_STAMP(0);
  TwStage<T, NT, NTW> twr;
  twr.issue(a.tw);
  if constexpr (CONLY) ct_fetch<T, N, CT_C, false, S>(a, s, sl, seq0, in_by_seq);
  else switch (kind) {                                                   // uniform: one straight-line fetch per variant
    case CT_C: ct_fetch<T, N, CT_C, false, S>(a, s, sl, seq0, in_by_seq); break;
    case CT_R1: ct_fetch<T, N, CT_R1, false, S>(a, s, sl, seq0, in_by_seq); break;
    case CT_R2: ct_fetch<T, N, CT_R2, false, S>(a, s, sl, seq0, in_by_seq); break;
    case CT_H1: if (a.lmul_in) ct_fetch<T, N, CT_H1, true, S>(a, s, sl, seq0, in_by_seq); else ct_fetch<T, N, CT_H1, false, S>(a, s, sl, seq0, in_by_seq); break;
    case CT_H2: if (a.lmul_in) ct_fetch<T, N, CT_H2, true, S>(a, s, sl, seq0, in_by_seq); else ct_fetch<T, N, CT_H2, false, S>(a, s, sl, seq0, in_by_seq); break;
    case CT_P1: if (a.pro.ph.pcx) ct_fetch<T, N, CT_P1, true, S>(a, s, sl, seq0, in_by_seq); else ct_fetch<T, N, CT_P1, false, S>(a, s, sl, seq0, in_by_seq); break;
    case CT_P2: if (a.pro.ph.pcx) ct_fetch<T, N, CT_P2, true, S>(a, s, sl, seq0, in_by_seq); else ct_fetch<T, N, CT_P2, false, S>(a, s, sl, seq0, in_by_seq); break;
    default: if (a.pro.ph.pcx) ct_fetch<T, N, CT_P3, true, S>(a, s, sl, seq0, in_by_seq); else ct_fetch<T, N, CT_P3, false, S>(a, s, sl, seq0, in_by_seq); break;
  }
  twr.commit(tw);
  CMBL_CT_STAMP(1);
  __syncthreads();
  CMBL_CT_STAMP(2);
  cx<T>* row = s + wave * LD;
  const bool live = seq0 + wave < a.nseq;                                // wave-uniform
  if (live) {
    ct_transform<T, N>(row, tw, lane);
    CMBL_CT_STAMP(3);
    if (a.lmul_mid) {                                                    // X <- conj(i l X), forward again: the inverse transform is conj(forward(conj .))
#pragma unroll
      for (int i = 0; i < (N + 63) / 64; ++i) {
        const int k = lane + 64 * i;
        if (N % 64 == 0 || k < N) row[pad(k)] = conj(mul_il(row[pad(k)], a.lmul_mid[k]));
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      ct_transform<T, N>(row, tw, lane);
    }
  }
  CMBL_CT_STAMP(4);
  __syncthreads();
  CMBL_CT_STAMP(5);
  ct_store_rows<T, N, S>(a, s, sl, seq0, out_by_seq, wave, lane, a.lmul_mid != nullptr);
  CMBL_CT_STAMP(6); CMBL_CT_STAMP(15);
}
template <typename T, int N>
// (the all-kinds kernel at 960 points, 15 elements per lane: 17 registers over the 128-register cap -- it takes the budget of the 1000-point kernels)
__global__ __launch_bounds__(64 * ct_Smax<T>(N), (ct_min_waves_x<T, (N == 960 ? 1000 : N), ct_Smax<T>(N)>())) void k_ct_dft(GenDft<T> a, int kind) { ct_dft_body<T, N, false, ct_Smax<T>(N)>(a, kind, blockIdx.y); }
// Two independent transform launches of the same length as one (grid.y = ny0 + the second's): the two x passes that open a delta-flow
// stage -- ifft_x(delta f) and the d/dx pass of rfft_y(f) -- have no dependence on each other.
// S: sequences (= wavefronts) per workgroup.  The x passes read and write contiguous rows, so nothing ties them to the 64-byte pieces of the
// transposed side: a launch with fewer row groups than CUs (768^2 QU: 770 rows = 97 groups of 8 on 256 CUs, two wavefronts per SIMD -- and the
// transforms are bound by VALU issue per SIMD) takes groups of 4 or 2 rows instead (Ctx::ct_rows_per_group), one wavefront per SIMD on twice
// as many CUs.
template <typename T, int N, int S = ct_S<T>()>
__global__ __launch_bounds__(64 * S, (ct_min_waves_x<T, N, S>())) void k_ct_dft2(GenDft<T> a0, int kind0, int ny0, GenDft<T> a1, int kind1) {
  if ((int)blockIdx.y < ny0) ct_dft_body<T, N, true, S>(a0, kind0, blockIdx.y);
  else ct_dft_body<T, N, true, S>(a1, kind1, blockIdx.y - (unsigned)ny0);
}
// one complex -> complex transform launch on contiguous rows (the x transforms and the one-launch d/dx pass)
template <typename T, int N, int S>
__global__ __launch_bounds__(64 * S, (ct_min_waves_x<T, N, S>())) void k_ct_dftx(GenDft<T> a, int kind /*CT_C (+ the debug builds' stamp flag)*/) { ct_dft_body<T, N, true, S>(a, kind, blockIdx.y); }

// The y passes of a forward flow stage in one launch (GenDft::yy; Ctx::gen_y_flow_stage): pair-c2r fetch, ct_flow_stage on every column,
// rfft_y(f_next) stored as a half spectrum in the layout of the inputs.  A kernel of its own: its register needs are not k_ct_dft's.
// (register budget: ONE workgroup per CU -- the operands of the pixels are held across a transform; at the sizes where a second resident
// workgroup would matter the launch has more workgroups than CUs anyway and the slice streams overlap the chains)
// (S: columns per workgroup -- the full 64-byte group, or half of it for launches that would otherwise leave most CUs idle: Ctx::ct_cols_per_group)
template <typename T, int N, int S = ct_S<T>()>
__global__ __launch_bounds__(64 * S, 1) void k_ct_flow_y(GenDft<T> a) {
  constexpr int NT = 64 * S, LD = ct_ld(N), NTW = N / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + NTW;
  const int seq0 = xcd_tile(blockIdx.x, gridDim.x) * S;
  const size_t sl = gen_slice(a);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  TwStage<T, NT, NTW> twr;
  twr.issue(a.tw);
  ct_fetch<T, N, CT_H2, true, S>(a, s, sl, seq0, true);
  twr.commit(tw);
  __syncthreads();
  if (seq0 + wave < a.nseq) ct_flow_stage<T, N>(a, s + wave * LD, tw, sl, seq0 + wave, lane);
  if (a.yy_last) return;
  __syncthreads();
  GenDft<T> b{};                                                         // the store side: rfft_y(f_next) as a half spectrum, [ky][x]
  b.out = a.yy_out; b.N = N; b.nout = a.yy_nout; b.nseq = a.nseq; b.in_real = 1; b.scale = T(1);
  b.out_seq = a.in_seq; b.out_elem = a.in_elem; b.out_slice = a.in_slice; b.out_tiled = a.in_tiled; b.tile_np = a.tile_np;
  ct_store_rows<T, N, S>(b, s, sl, seq0, true, wave, lane, false);
}

// The four y passes of a delta-flow stage in one launch (GenDft::yy = 2; Ctx::gen_y_delta_stage): the c2r of ifft_x(delta f) -> L(df), the pair
// c2r -> (d/dx f, d/dy f), the stage's pointwise work (src/lenseflow.jl:184-200: the products for the delta-phi quadrature, the f velocity
// with its RK update, the pair (p_x, p_y) L(df)), rfft_y of the next f and the pair r2c of the delta-f velocity.  Two LDS rows per column
// (row set 0: the gradient pair / next f; row set 1: L(df) / the velocity pair) and TWO WAVEFRONTS per column, one per row set: the two
// fetches, the two inverse transforms, the two forward transforms and the two stores each run side by side (the first version ran a
// column's four transforms on one wavefront: 23.4 us per launch at 768^2, profiles/r05_kernel_stats_768QU_f32_anysize_pre2w.csv); the
// pointwise part is split between them in alternating 64-pixel pieces.  Four workgroup barriers.
template <typename T, int N, int S = ct_S2<T>(N)>
__global__ __launch_bounds__(128 * S) void k_ct_delta_y(GenDft<T> a) {
  constexpr int NTH = 64 * S, NT = 2 * NTH, LD = ct_ld(N), NTW = N / 2, E = (N + 63) / 64, EH = (E + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + NTW;
  cx<T>* s2 = s + S * LD;
  const int seq0 = xcd_tile(blockIdx.x, gridDim.x) * S;
  const size_t sl = gen_slice(a);
  const int set = threadIdx.x / NTH, tid = threadIdx.x % NTH, wave = tid >> 6, lane = tid & 63, seq = seq0 + wave;
  TwStage<T, NT, NTW> twr;
  twr.issue(a.tw);
  if (set == 0) ct_fetch<T, N, CT_H2, true, S>(a, s, sl, seq0, true, tid);
  else {
    GenDft<T> a3 = a;                                                    // ifft_x(delta f): a single half plane, no multiplier
    a3.in = a.yy_in3; a3.in2 = nullptr; a3.lmul_in = nullptr;
    ct_fetch<T, N, CT_H1, false, S>(a3, s2, sl, seq0, true, tid);
  }
  twr.commit(tw);
  const bool live = seq < a.nseq;
  const GenPro<T>& e = a.pro;
  const size_t mb = sl * (size_t)e.npix, pb = (size_t)(e.ph.Bphi == 1 ? 0 : sl / e.P) * e.npix;
  const bool pc = e.ph.pcx != nullptr;
  cx<T>* r1 = s + wave * LD;
  cx<T>* r2 = s2 + wave * LD;
  // this wavefront's share of the column's pixels: pieces i = 2 j + set; their operands are requested before the transform
  T px[EH], py[EH], y0v[EH], acv[EH];
  const int seqc = live ? seq : a.nseq - 1;
#pragma unroll
  for (int j = 0; j < EH; ++j) {
    const unsigned o = (unsigned)seqc * (unsigned)N + (unsigned)min(lane + 64 * (2 * j + set), N - 1);
    if (CMBL_DY_PROBE & 4) { y0v[j] = T(o); acv[j] = T(1); px[j] = T(2); py[j] = T(3); continue; }
    y0v[j] = at32(e.y0 + mb, o); acv[j] = at32(e.acc + mb, o);
    if (pc) { px[j] = at32(e.ph.pcx + pb, o); py[j] = at32(e.ph.pcy + pb, o); }
    else {
      T m11, m12, m22;
      flow_pm(e.rk.t, at32(e.ph.gx + pb, o), at32(e.ph.gy + pb, o), at32(e.ph.hxx + pb, o), at32(e.ph.hyx + pb, o), at32(e.ph.hyy + pb, o), px[j], py[j], m11, m12, m22);
    }
  }
  __syncthreads();
  if (live && !(CMBL_DY_PROBE & 1)) ct_transform<T, N>(set ? r2 : r1, tw, lane);
  __syncthreads();
  if (live) {
#pragma unroll
    for (int j = 0; j < EH; ++j) {
      const int n = lane + 64 * (2 * j + set);
      if (2 * j + set < E && (N % 64 == 0 || n < N)) {
        const unsigned o = (unsigned)seq * (unsigned)N + (unsigned)n;
        const cx<T> z = r1[pad(n)], zl = r2[pad(n)];                     // e^{+i} transforms: the values are conj(z), conj(zl)
        const T gx = a.scale * z.x, gy = -a.scale2 * z.y, l = a.yy_scale3 * zl.x;
        const T k = px[j] * gx + py[j] * gy;
        T y = y0v[j], ac = e.rk.stage == 1 ? T(0) : acv[j];
        const T nxt = rk_update(e.rk, k, y, ac);
        if (!(CMBL_DY_PROBE & 2)) { at32(e.w1p + mb, o) = l * gx; at32(e.w2p + mb, o) = l * gy; }
        if (e.rk.stage == 4) at32(e.y0 + mb, o) = y; else at32(e.acc + mb, o) = ac;
        r1[pad(n)] = mk<T>(nxt, T(0));
        r2[pad(n)] = mk<T>(px[j] * l, py[j] * l);
      }
    }
  }
  __syncthreads();
  if (live && !(set == 0 && a.yy_last) && !(CMBL_DY_PROBE & 1)) ct_transform<T, N>(set ? r2 : r1, tw, lane);
  __syncthreads();
  GenDft<T> b{};                                                         // store side, [ky][x] like the inputs
  b.N = N; b.nout = a.yy_nout; b.nseq = a.nseq; b.in_real = 1; b.scale = T(1); b.scale2 = T(1);
  b.out_seq = a.in_seq; b.out_elem = a.in_elem; b.out_slice = a.in_slice; b.out_tiled = a.in_tiled; b.tile_np = a.tile_np;
  if (CMBL_DY_PROBE & 8) return;
  if (set == 0) {
    if (!a.yy_last) { b.out = a.yy_out; ct_store_rows<T, N, S>(b, s, sl, seq0, true, wave, lane, false, tid); }
  } else {
    b.out = a.yy_out2; b.out2 = a.yy_out3; b.in2 = a.yy_out3;            // in2 != nullptr marks the pair split (ct_put)
    ct_store_rows<T, N, S>(b, s2, sl, seq0, true, wave, lane, false, tid);
  }
}

// The y passes of an adjoint flow stage in one launch (GenDft::yy = 3; Ctx::gen_y_adj_stage): c2r of yy_in3 = ifft_x(y) -> the map y, the
// pair (p_x y, p_y y) (src/lenseflow.jl:166-170), its pair r2c split into yy_out2 / yy_out3.  One LDS row per column.
template <typename T, int N, int S = ct_S<T>()>
__global__ __launch_bounds__(64 * S, 1) void k_ct_adj_y(GenDft<T> a) {
  constexpr int NT = 64 * S, LD = ct_ld(N), NTW = N / 2, E = (N + 63) / 64;
  constexpr int PCH = E <= 12 ? E : (E + ((E + 11) / 12) - 1) / ((E + 11) / 12), NPC = (E + PCH - 1) / PCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + NTW;
  const int seq0 = xcd_tile(blockIdx.x, gridDim.x) * S;
  const size_t sl = gen_slice(a);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  TwStage<T, NT, NTW> twr;
  twr.issue(a.tw);
  ct_fetch<T, N, CT_H1, false, S>(a, s, sl, seq0, true);
  twr.commit(tw);
  __syncthreads();
  const int seq = seq0 + wave;
  if (seq < a.nseq) {
    cx<T>* row = s + wave * LD;
    const GenPro<T>& e = a.pro;
    const size_t pb = (size_t)(e.ph.Bphi == 1 ? 0 : sl / e.P) * e.npix;
    const bool pc = e.ph.pcx != nullptr;
    ct_transform<T, N>(row, tw, lane);
#pragma unroll
    for (int c = 0; c < NPC; ++c) {
      T px[PCH], py[PCH];
#pragma unroll
      for (int ii = 0; ii < PCH; ++ii) {
        const int n = min(lane + 64 * (c * PCH + ii), N - 1);
        const unsigned o = (unsigned)seq * (unsigned)N + (unsigned)n;
        if (pc) { px[ii] = at32(e.ph.pcx + pb, o); py[ii] = at32(e.ph.pcy + pb, o); }
        else {
          T m11, m12, m22;
          flow_pm(e.rk.t, at32(e.ph.gx + pb, o), at32(e.ph.gy + pb, o), at32(e.ph.hxx + pb, o), at32(e.ph.hyx + pb, o), at32(e.ph.hyy + pb, o), px[ii], py[ii], m11, m12, m22);
        }
      }
#pragma unroll
      for (int ii = 0; ii < PCH; ++ii) {
        const int n0 = lane + 64 * (c * PCH + ii), n = min(n0, N - 1);
        const T y = a.scale * row[pad(n)].x;                             // e^{+i} transform: the value is conj(z); real part
        if ((N % 64 == 0 || n0 < N) && (c * PCH + ii) < E) row[pad(n)] = mk<T>(px[ii] * y, py[ii] * y);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ct_transform<T, N>(row, tw, lane);
  }
  __syncthreads();
  GenDft<T> b{};
  b.N = N; b.nout = a.yy_nout; b.nseq = a.nseq; b.in_real = 1; b.scale = T(1); b.scale2 = T(1);
  b.out_seq = a.in_seq; b.out_elem = a.in_elem; b.out_slice = a.in_slice; b.out_tiled = a.in_tiled; b.tile_np = a.tile_np;
  b.out = a.yy_out2; b.out2 = a.yy_out3; b.in2 = a.yy_out3;              // in2 != nullptr marks the pair split (ct_put)
  ct_store_rows<T, N, S>(b, s, sl, seq0, true, wave, lane, false);
}

// The x pass that closes an adjoint-type stage in one launch (Ctx::gen_x_adj_update): fft_x of both members of the velocity pair and the RK
// update of the Fourier state, k = i lx Fx + i ly Fy (src/lenseflow.jl:163-174; k_gen_adj_rk).  A workgroup takes S/2 adjacent ky rows of
// both members (wave w: member w / (S/2), row w % (S/2)); after the transforms the two wavefronts of a row share its kx range in alternating
// 64-element pieces.  a.in / a.in2: the pair; a.yy_out / a.out2 / a.out: Y0 / acc / Ys; a.lmul_out / a.lmul_in: lx / ly.
// a.yy_out2 != nullptr (round 6): the workgroup ALSO opens the next stage -- it has the new stage input Ys of its rows in registers, so it
// transforms them back, T3 = ifft_x(Ys) (unnormalised, what the next stage's y kernel fetches), instead of a launch of its own reading
// Ys back from memory (k_ct_dft2's first half / gen_x); a.out (Ys) may then be nullptr.  Same wavefront arithmetic: bit-identical.
template <typename T, int N, int S>
__device__ __forceinline__ void ct_adj_x_body(const GenDft<T>& a, unsigned ysl) {
  constexpr int R = S / 2, LGR = ilog2c(R), NT = 64 * S, LD = ct_ld(N), NTW = N / 2, E = (N + 63) / 64, EH = (E + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + NTW;
  const size_t sl = gen_slice(a, ysl), sb = sl * a.in_slice, sbs = sl * a.out_slice;   // slice of the hand-off arrays (pair, T3) / of the Fourier state (Y0, acc, Ys)
  const int grp = a.in_tiled ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;   // (tiled: two groups share each 128-byte line of the pair and of T3)
  if (grp * R >= a.nseq) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = wave / R, r = wave % R, ky = grp * R + r;
  const bool live = ky < a.nseq;
  const int kyc = live ? ky : a.nseq - 1;
  TwStage<T, NT, NTW> twr;
  twr.issue(a.tw);
  const RKCoef<T> rk = a.pro.rk;
  // tiled arrays: thread (r, lane) of a member -> row trr of the member's R rows, x = txb + 64 i (consecutive lanes: the 4 x of a block row, then the rows)
  const int tq = r * 64 + lane, trr = (tq >> 2) & (R - 1), txb = ((tq >> (2 + LGR)) << 2) + (lane & 3);
  const unsigned tG = (unsigned)(txb >> 2) * 4u * (unsigned)a.tile_np + (unsigned)(lane & 3);
  const cx<T>* Y0 = reinterpret_cast<const cx<T>*>(a.yy_out) + sbs;
  const cx<T>* Ac = reinterpret_cast<const cx<T>*>(a.out2) + sbs;
  cx<T> v[E] = {}, y0v[EH] = {}, acv[EH] = {};
  {
    const cx<T>* src = reinterpret_cast<const cx<T>*>(m ? a.in2 : a.in) + sb;
    // ONE load statement for both layouts (the two as branches around the loop put the aggregate v[] into scratch memory in double precision):
    // tiled pair: the R wavefronts of the member walk (4 x, R rows) blocks of its R rows, element i of the thread at x = txb + 64 i;
    // [ky][x]: a wavefront walks its own row.  Offset = o0 + i * stp, only the last element can lie beyond the row (clamped)
    const bool tl = a.in_tiled != 0;
    const unsigned Fs = 4u * (unsigned)min(grp * R + trr, a.nseq - 1);
    const unsigned o0 = tl ? Fs + tG : (unsigned)kyc * (unsigned)N + (unsigned)lane, stp = tl ? 64u * (unsigned)a.tile_np : 64u;
#pragma unroll
    for (int i = 0; i < E; ++i) {
      unsigned o = o0 + (unsigned)i * stp;
      if (!(N % 64 == 0 || i < E - 1)) {
        const unsigned xc = (unsigned)min((tl ? txb : lane) + 64 * i, N - 1);
        o = tl ? Fs + (xc >> 2) * 4u * (unsigned)a.tile_np + (xc & 3u) : (unsigned)kyc * (unsigned)N + xc;
      }
      v[i] = at32(src, o);
    }
#pragma unroll
    for (int j = 0; j < EH; ++j) {                                       // the RK operands of this wavefront's share, requested up front
      const unsigned o = (unsigned)kyc * (unsigned)N + (unsigned)min(lane + 64 * (2 * j + m), N - 1);
      y0v[j] = at32(Y0, o); acv[j] = at32(Ac, o);
    }
  }
  cx<T>* row = s + wave * LD;
  {
    const bool tl = a.in_tiled != 0;
    cx<T>* const dstrow = tl ? s + (m * R + trr) * LD : row;
    const int x0 = tl ? txb : lane;
#pragma unroll
    for (int i = 0; i < E; ++i) { const int x = x0 + 64 * i; if (N % 64 == 0 || x < N) dstrow[pad(x)] = v[i]; }
  }
  twr.commit(tw);
  __syncthreads();
  if (live) ct_transform<T, N>(row, tw, lane);
  __syncthreads();
  cx<T>* const T3 = a.yy_out2 ? reinterpret_cast<cx<T>*>(a.yy_out2) + sb : nullptr;       // (uniform)
  if (live) {
    const T l_y = a.lmul_in[ky];
    cx<T>* Ys = a.out ? reinterpret_cast<cx<T>*>(a.out) + sbs : nullptr;
    cx<T>* dst = (rk.stage == 4 ? reinterpret_cast<cx<T>*>(a.yy_out) : reinterpret_cast<cx<T>*>(a.out2)) + sbs;
#pragma unroll
    for (int j = 0; j < EH; ++j) {
      const int k = lane + 64 * (2 * j + m);
      if (2 * j + m < E && (N % 64 == 0 || k < N)) {
        const cx<T> kv = mul_il(s[r * LD + pad(k)], a.lmul_out[k]) + mul_il(s[(R + r) * LD + pad(k)], l_y);
        cx<T> y = y0v[j], ac = acv[j];
        if (rk.stage == 1) { ac.x = T(0); ac.y = T(0); }                   // (member-wise: a select between aggregates goes through scratch memory in double precision)
        const cx<T> nxt = rk_update(rk, kv, y, ac);
        const unsigned o = (unsigned)ky * (unsigned)N + (unsigned)k;
        if (rk.stage == 4) at32(dst, o) = y; else at32(dst, o) = ac;
        if (Ys) at32(Ys, o) = nxt;
        if (T3) s[r * LD + pad(k)] = conj(nxt);                            // the inverse transform is conj(forward(conj .)); this lane alone read the slot
      }
    }
  }
  if (!T3) return;
  __syncthreads();
  if (live && m == 0) ct_transform<T, N>(row, tw, lane);                  // (member 0's wavefront owns row r)
  __syncthreads();
  const bool wt = wt_line<T>(N);
  if (a.out_tiled) {                                                     // tiled T3: the whole workgroup walks (4 x, R rows) blocks of its R rows
    const int q0 = (int)threadIdx.x, rr = (q0 >> 2) & (R - 1), xb = ((q0 >> (2 + LGR)) << 2) + (q0 & 3), kyr = grp * R + rr;   // x = xb + 128 j
    const unsigned o0 = ((unsigned)(xb >> 2) * (unsigned)a.tile_np + (unsigned)kyr) * 4u + (unsigned)(q0 & 3), stp = (unsigned)(NT / R) * (unsigned)a.tile_np;
#pragma unroll
    for (int j = 0; j < EH; ++j) {
      const int x = xb + j * (NT / R);
      if (x < N && kyr < a.nseq) {
        const cx<T> y = conj(s[rr * LD + pad(x)]);
        ct_store<(int)sizeof(cx<T>)>(T3, (o0 + (unsigned)j * stp) * (unsigned)sizeof(cx<T>), &y, wt);
      }
    }
  } else if (live) {
#pragma unroll
    for (int j = 0; j < EH; ++j) {
      const int k = lane + 64 * (2 * j + m);
      if (2 * j + m < E && (N % 64 == 0 || k < N)) {
        const cx<T> y = conj(s[r * LD + pad(k)]);
        ct_store<(int)sizeof(cx<T>)>(T3, ((unsigned)ky * (unsigned)N + (unsigned)k) * (unsigned)sizeof(cx<T>), &y, wt);
      }
    }
  }
}
template <typename T, int N, int S = ct_S<T>()>
__global__ __launch_bounds__(64 * S, (ct_min_waves_x<T, N, S>())) void k_ct_adj_x(GenDft<T> a) { ct_adj_x_body<T, N, S>(a, blockIdx.y); }
// ... and, in the delta flow, the d/dx pass of the next stage's f (independent of the row update: the second half of k_ct_dft2) as further
// workgroups of the same launch: grid.y = ny_adj slices of row updates + the slices of a1; grid.x covers the larger of the two parts
template <typename T, int N, int S = ct_S<T>()>
__global__ __launch_bounds__(64 * S, (ct_min_waves_x<T, N, S>())) void k_ct_adj_x_dx(GenDft<T> a, int ny_adj, GenDft<T> a1) {
  if ((int)blockIdx.y < ny_adj) ct_adj_x_body<T, N, S>(a, blockIdx.y);       // (either part leaves by itself where the grid is larger than it needs)
  else ct_dft_body<T, N, true, S>(a1, CT_C, blockIdx.y - (unsigned)ny_adj, a1.in_tiled ? (((a1.nseq + S - 1) / S + 7) & ~7) : (a1.nseq + S - 1) / S);
}

}  // namespace cmbl
