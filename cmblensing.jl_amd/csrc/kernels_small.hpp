// Small maps (32 <= Ny, Nx <= 128): a WHOLE LenseFlow -- all 4n RK stages -- as ONE launch with one workgroup per (pol, batch) slice.
//
// Reference algorithm: src/lenseflow.jl:150-174 (velocity, velocityᴴ) under src/numerical_algorithms.jl:11-24 (RK4), i.e. what
// Flow::flow_map / Flow::flow_adj_F run as 2 launches per stage (kernels_flow.hpp).  At these sizes -- the reference's own test sizes,
// test/runtests.jl:53, and the regime of its "batch of 10 for the cost of one", docs/src/06_gpu.ipynb:664 -- a launch is 4-5 us of
// latency around < 1 us of work: 56 dependent launches for 0.13 MB of state at 128^2.  Here a slice's half plane lives in LDS for the whole
// flow (2 Nyh x Nx complex values + a spare slot per column: 131 KB at 128^2 in single precision), the RK state (y0, acc) in registers, the
// stages are separated by workgroup barriers instead of launch boundaries, and p(t) comes from the per-phi cache (k_pcache) through L2.
// P * B workgroups run side by side, so batches of chains / simulations (MAP_marg: Nsims = 50) fill the chip.
//
// LDS array W[x][LDY], LDY = 2 Nyh + 1 (odd), complex.  Both transform directions map CONSECUTIVE LANES TO CONSECUTIVE SEQUENCES:
//   y transforms: sequence = column x (stride LDY, odd -> conflict free), element stride 1
//   x transforms: sequence = ky slot (stride 1),                          element stride LDY
// so no padding scheme and no transposition is needed.  Transforms: DIF forward (natural -> bit-reversed), DIT inverse (bit-reversed ->
// natural), radix up to 16 per LDS round trip with the register butterflies of fft_core.hpp -- the index algebra is fft_lds.hpp's
// dif_stage / dit_stage with strides.  Real data: a real column as an Ny/2-point transform (r2c post / c2r pre), two real columns as one
// Ny-point transform (pair); the c2r drops Im of the ky = 0 / Ny/2 entries after the x pass exactly like FFTW (DESIGN.md §3 Nyquist).
#pragma once
#include "kernels_flow.hpp"

namespace cmbl {

// The per-thread loops below are fully unrolled (their arrays live in registers); without a fence the scheduler overlaps ALL iterations --
// every operand of 16-32 pixels in flight at once -- and spills.  SM_FENCE(i) keeps groups of four iterations apart.
#define SM_FENCE(i) do { if (((i) & 3) == 3) __builtin_amdgcn_sched_barrier(0); } while (0)

template <typename T, int LGNY, int LGNX> struct SmallGeom {
  static constexpr int Ny = 1 << LGNY, Nx = 1 << LGNX, M = Ny / 2, LGM = LGNY - 1, Nyh = M + 1, LDY = 2 * Nyh + 1, NPIX = Ny * Nx, NF = Nyh * Nx;
  static constexpr int LGNTW = LGNY > LGNX ? LGNY : LGNX, NTW = 1 << LGNTW;
  // threads: a thread holds NPIX / NT pixels of RK state (y0, acc) and of p(t) in registers next to a butterfly's operands.  1024 threads
  // (128 registers each) up to 8 words of state per array, 512 threads (256 registers) beyond that and in double precision
  static constexpr int WORDS = NPIX * (int)sizeof(T) / 4;
  static constexpr int NTMAX = sizeof(T) == 8 ? 512 : 1024;
  static constexpr int NT = NPIX / 4 < NTMAX ? NPIX / 4 : NTMAX;
  static constexpr int PPT = NPIX / NT, FPT = (NF + NT - 1) / NT;          // pixels / Fourier modes per thread
  static constexpr int MAXLG = (sizeof(T) == 8 || WORDS > 8192) ? 3 : 4;   //                    // largest radix of a stage: 16 (8 in double precision: 64 registers of operands)
  // per-stage address recomputation (see the kernels) from 8 pixels per thread on; below that the registers have room for the hoisted
  // addresses and the stage saves their arithmetic: 64^2 L*f 0.146 -> 0.119 ms, L'g 0.167 -> 0.146 ms (profiles/r06_ab_small_flow.txt)
#ifndef CMBL_SMALL_LAUNDER_PPT
#define CMBL_SMALL_LAUNDER_PPT 8
#endif
  static constexpr bool LAUNDER = PPT >= CMBL_SMALL_LAUNDER_PPT;
  static constexpr bool delta_fits = NPIX * (int)sizeof(T) <= 4096 * 4;     // the one-launch delta flow: 64 x 64 in single, 32 x 64 in double precision
  static constexpr size_t lds = ((size_t)NTW + (size_t)Nx * LDY) * sizeof(cx<T>);
  // (double precision up to 64 x 64: beyond that the adjoint kernel's state does not fit the register file of a 512-thread workgroup)
  static constexpr bool fits = lds <= 160 * 1024 && LGNY >= 5 && LGNX >= 5 && LGNY <= 7 && LGNX <= 7 && (sizeof(T) == 4 || NPIX <= 4096);
};

// ---- strided in-LDS transforms (see the header comment; the stage algebra is fft_lds.hpp's) -------------------------------------
struct SmNoPre { template <typename V> __device__ __forceinline__ V operator()(V v, int, int) const { return v; } };

// one DIF stage: NSEQ sequences at in + seq * SSTR (elements ESTR apart); out may differ from in (same strides)
template <typename T, int NT, int NSEQ, int SSTR, int ESTR, int LGN, int LGNTW, int LGH, int LG>
__device__ __forceinline__ void sm_dif_stage(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, const cx<T>* __restrict__ tw, int tid) {
  using V = typename vreg<T>::type;
  constexpr int r = 1 << LG, lghmin = LGH - LG + 1, hmin = 1 << lghmin, lgnb = LGN - LG, sh = LGNTW - (LGH + 1), total = NSEQ << lgnb;
#pragma unroll 1
  for (int q = tid; q < total; q += NT) {
    const int rr = q / NSEQ, seq = q - rr * NSEQ;
    const int blk = rr >> lghmin, j = rr & (hmin - 1);
    const int o = seq * SSTR + ((blk << (LGH + 1)) + j) * ESTR;
    V v[r];
#pragma unroll
    for (int m = 0; m < r; ++m) v[m] = vload(in + o + (m << lghmin) * ESTR);
    V w[r];
    if constexpr (hmin > 1) stage_twiddles<T, r>(tw, j, sh, w);
    dft<T, LG, false>(v);
#pragma unroll
    for (int k = 0; k < r; ++k) {
      V x = v[dft_loc<LG>(k)];
      if (hmin > 1 && k > 0) x = vmul(x, w[k]);
      vstore(out + o + (brevc<LG>(k) << lghmin) * ESTR, x);
    }
  }
  __syncthreads();
}
// one DIT stage, in place; pre(value, seq, logical index) rides in the load of the FIRST stage (bit-reversed input)
template <typename T, int NT, int NSEQ, int SSTR, int ESTR, int LGN, int LGNTW, int LGH, int LG, typename PRE>
__device__ __forceinline__ void sm_dit_stage(cx<T>* __restrict__ s, const cx<T>* __restrict__ tw, int tid, PRE pre) {
  using V = typename vreg<T>::type;
  constexpr int r = 1 << LG, hmin = 1 << LGH, lgnb = LGN - LG, sh = LGNTW - (LGH + LG), total = NSEQ << lgnb;
#pragma unroll 1
  for (int q = tid; q < total; q += NT) {
    const int rr = q / NSEQ, seq = q - rr * NSEQ;
    const int blk = rr >> LGH, j = rr & (hmin - 1), b0 = (blk << (LGH + LG)) + j;
    cx<T>* p = s + seq * SSTR + b0 * ESTR;
    V v[r], w[r];
    if constexpr (hmin > 1) stage_twiddles<T, r>(tw, j, sh, w);
#pragma unroll
    for (int k = 0; k < r; ++k) {
      const int m = brevc<LG>(k);
      V x = vfrom(pre(p[(m << LGH) * ESTR], seq, b0 + (m << LGH)));
      if (hmin > 1 && k > 0) x = vmulc(x, w[k]);
      v[k] = x;
    }
    dft<T, LG, true>(v);
#pragma unroll
    for (int m = 0; m < r; ++m) vstore(p + (m << LGH) * ESTR, v[dft_loc<LG>(m)]);
  }
  __syncthreads();
}
// (MAXLG: largest radix 2^MAXLG of a stage -- 16 where the registers allow it, 8 where a thread also holds 16 pixels of RK state)
template <typename T, int NT, int NSEQ, int SSTR, int ESTR, int LGN, int LGNTW, int MAXLG, int I = 0>
__device__ __forceinline__ void sm_dif(const cx<T>* __restrict__ in, cx<T>* __restrict__ s, const cx<T>* __restrict__ tw, int tid) {
  if constexpr (I < num_stages(LGN, MAXLG)) {
    constexpr int LG = stage_lg(LGN, I, MAXLG), LGH = levels_after(LGN, I, MAXLG) + LG - 1;
    sm_dif_stage<T, NT, NSEQ, SSTR, ESTR, LGN, LGNTW, LGH, LG>(I == 0 ? in : s, s, tw, tid);
    sm_dif<T, NT, NSEQ, SSTR, ESTR, LGN, LGNTW, MAXLG, I + 1>(in, s, tw, tid);
  }
}
template <typename T, int NT, int NSEQ, int SSTR, int ESTR, int LGN, int LGNTW, int MAXLG, typename PRE = SmNoPre, int I = num_stages(LGN, MAXLG) - 1>
__device__ __forceinline__ void sm_dit(cx<T>* __restrict__ s, const cx<T>* __restrict__ tw, int tid, PRE pre = PRE()) {
  if constexpr (I >= 0) {
    constexpr int LG = stage_lg(LGN, I, MAXLG), LGH = levels_after(LGN, I, MAXLG);
    if constexpr (I == num_stages(LGN, MAXLG) - 1) sm_dit_stage<T, NT, NSEQ, SSTR, ESTR, LGN, LGNTW, LGH, LG, PRE>(s, tw, tid, pre);
    else sm_dit_stage<T, NT, NSEQ, SSTR, ESTR, LGN, LGNTW, LGH, LG, SmNoPre>(s, tw, tid, SmNoPre());
    sm_dit<T, NT, NSEQ, SSTR, ESTR, LGN, LGNTW, MAXLG, SmNoPre, I - 1>(s, tw, tid);
  }
}

// slot of half-spectrum entry k of a column (0..M): bit-reversed below M, the Nyquist entry in the spare slot M (fft_lds.hpp hslot, unpadded)
template <int LGM> __device__ __forceinline__ int sm_hslot(int k) { return k < (1 << LGM) ? brevc<LGM>(k) : (1 << LGM); }

// after the M-point DIF of z[j] = f[2j] + i f[2j+1] on every column: the half spectrum A[0..M] in place (fft_lds.hpp r2c_post)
template <typename T, typename G>
__device__ __forceinline__ void sm_r2c_post(cx<T>* __restrict__ s, const cx<T>* __restrict__ tw, int tid) {
  constexpr int np = G::M / 2 + 1, TWS = G::NTW / G::Ny;
  for (int q = tid; q < G::Nx * np; q += G::NT) {
    const int k = q / G::Nx, x = q - k * G::Nx;
    cx<T>* p = s + x * G::LDY;
    if (k == 0) {
      const cx<T> z = p[0];
      p[0] = mk<T>(z.x + z.y, 0);
      p[G::M] = mk<T>(z.x - z.y, 0);
    } else {
      const int k2 = G::M - k, i1 = brevc<G::LGM>(k), i2 = brevc<G::LGM>(k2);
      const cx<T> a = p[i1], b = p[i2];
      const cx<T> e = mk<T>(T(0.5) * (a.x + b.x), T(0.5) * (a.y - b.y));
      const cx<T> o = mk<T>(T(0.5) * (a.x - b.x), T(0.5) * (a.y + b.y));
      const cx<T> wo = mul_mi(o * tw[k * TWS]);
      p[i1] = e + wo;
      if (k2 != k) p[i2] = conj(e - wo);
    }
  }
  __syncthreads();
}
// before the M-point DIT: Z[k] from A[k], Im A[0] and Im A[M] dropped (FFTW's c2r rule; fft_lds.hpp c2r_pre)
template <typename T, typename G>
__device__ __forceinline__ void sm_c2r_pre(cx<T>* __restrict__ s, const cx<T>* __restrict__ tw, int tid) {
  constexpr int np = G::M / 2 + 1, TWS = G::NTW / G::Ny;
  for (int q = tid; q < G::Nx * np; q += G::NT) {
    const int k = q / G::Nx, x = q - k * G::Nx;
    cx<T>* p = s + x * G::LDY;
    if (k == 0) {
      const T a0 = p[0].x, am = p[G::M].x;
      p[0] = mk<T>(a0 + am, a0 - am);
    } else {
      const int k2 = G::M - k, i1 = brevc<G::LGM>(k), i2 = brevc<G::LGM>(k2);
      const cx<T> a = p[i1], b = p[i2];
      const cx<T> e = mk<T>(a.x + b.x, a.y - b.y);
      const cx<T> o = mk<T>(a.x - b.x, a.y + b.y);
      const cx<T> wo = mul_i(cmulconj(o, tw[k * TWS]));
      p[i1] = e + wo;
      if (k2 != k) p[i2] = conj(e - wo);
    }
  }
  __syncthreads();
}

template <typename T> struct SmallArgs {
  const T* in; T* out;                    // forward-type: maps [slice][x][y];  adjoint-type: F-layout half planes [slice][ky][xr] (as T pairs)
  const T* pcache;                        // [2n+1][2][Bphi][npix]  (Flow::pcache)
  const cx<T>* tw;                        // exp(-2 pi i k / NTW), full circle
  const T *lx_r, *ly;                     // l_x by x slot (bit-reversed), l_y by ky
  int n, P, Bphi, k0, dir;                // RK steps; slices per batch slot; phi slots; first stage-time index and direction on the 2n+1 grid
  T hhalf, hfull, h6;                     // h / 2, h, h / 6 (rounded on the host like Flow::coef)
};

template <typename T> __device__ __forceinline__ RKCoef<T> sm_coef(const SmallArgs<T>& a, int stage, bool last) {
  RKCoef<T> rk;
  rk.t = T(0); rk.cnext = stage <= 2 ? a.hhalf : a.hfull; rk.h6 = a.h6; rk.stage = stage; rk.last = last ? 1 : 0;
  return rk;
}

// ---- L*f / L\f (src/lenseflow.jl:150-161): df/dt = p(t) . grad f on a real map ----------------------------------------------------
template <typename T, int LGNY, int LGNX>
__global__ __launch_bounds__((SmallGeom<T, LGNY, LGNX>::NT)) void k_small_flow(SmallArgs<T> a) {
  using G = SmallGeom<T, LGNY, LGNX>;
  constexpr int NT = G::NT, Ny = G::Ny, Nx = G::Nx, M = G::M, Nyh = G::Nyh, LDY = G::LDY, PPT = G::PPT, LGM = G::LGM, LGNTW = G::LGNTW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* W = tw + G::NTW;
  T* Wf = reinterpret_cast<T*>(W);                                          // real view: pixel (x, y) of a packed column at x * 2 LDY + y
  const int tid0 = threadIdx.x;
  const size_t sl = blockIdx.x, mb = sl * (size_t)G::NPIX;
  const size_t ps = (size_t)a.Bphi * G::NPIX, pb = (size_t)(a.Bphi == 1 ? 0 : sl / a.P) * G::NPIX;
  for (int i = tid0; i < G::NTW; i += NT) tw[i] = a.tw[i];
  T y0[PPT], acc[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int p = tid0 + i * NT;
    y0[i] = a.in[mb + p]; acc[i] = T(0);
    Wf[(p >> LGNY) * (2 * LDY) + (p & (Ny - 1))] = y0[i];
    SM_FENCE(i);
  }
  __syncthreads();
  const T sx = T(1) / (T(Nx) * T(Ny)), sy = T(1) / T(Ny);
  int kt = a.k0;                                                            // stage-time index of the current stage
  for (int step = 0; step < a.n; ++step)
    for (int stage = 1; stage <= 4; ++stage) {
      // the thread index as a value the compiler cannot see through: every address below is recomputed per stage.  With the plain index all of
      // them are invariants of the 4n-stage loop, get hoisted out of it and are kept live across it (65-126 spilled registers at 128^2)
      int tid = tid0;
      if constexpr (G::LAUNDER) asm volatile("" : "+v"(tid));
      // A = rfft_y(f) by columns: M-point DIF + post -> W[x][hslot(ky)]
      sm_dif<T, NT, Nx, LDY, 1, LGM, LGNTW, G::MAXLG>(W, W, tw, tid);
      sm_r2c_post<T, G>(W, tw, tid);
      // Gx = ifft_x(i lx fft_x(A)) by ky slots -> W[x][Nyh + slot]; the multiply rides in the load of the first inverse stage
      sm_dif<T, NT, Nyh, 1, LDY, LGNX, LGNTW, G::MAXLG>(W, W + Nyh, tw, tid);
      sm_dit<T, NT, Nyh, 1, LDY, LGNX, LGNTW, G::MAXLG>(W + Nyh, tw, tid, [&](cx<T> v, int, int xs) { return mul_il(v, a.lx_r[xs]); });
      // pair: z = ext(Gx) + i ext(i ly A), Hermitian extension with FFTW's c2r rule, into the bit-reversed slots of an Ny-point column
      {
        constexpr int NIT = (Nx * Nyh + NT - 1) / NT;
        cx<T> za[NIT], zb[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int q = tid + i * NT, k = q / Nx, x = q - k * Nx;
          if (q < Nx * Nyh) {
            const int s = sm_hslot<LGM>(k);
            const cx<T> A = W[x * LDY + s], Gv = W[x * LDY + Nyh + s];
            const T l = a.ly[k];
            if (k == 0 || k == M) { za[i] = mk<T>(Gv.x, -l * A.y); zb[i] = za[i]; }
            else { za[i] = mk<T>(Gv.x - l * A.x, Gv.y - l * A.y); zb[i] = mk<T>(Gv.x + l * A.x, -Gv.y - l * A.y); }   // z[k] = G - l A, z[N - k] = conj(G) + l conj(A)
          }
          SM_FENCE(i);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int q = tid + i * NT, k = q / Nx, x = q - k * Nx;
          if (q < Nx * Nyh) {
            W[x * LDY + brevc<LGNY>(k)] = za[i];
            if (k != 0 && k != M) W[x * LDY + brevc<LGNY>(Ny - k)] = zb[i];
          }
          SM_FENCE(i);
        }
        __syncthreads();
      }
      // p(t) of this stage: requested before the last transform, consumed after it
#ifndef SM_LATE_P
      T px[PPT], py[PPT];
      {
        const T* pc = a.pcache + (size_t)(2 * kt) * ps + pb;
#pragma unroll
        for (int i = 0; i < PPT; ++i) { px[i] = pc[tid + i * NT]; py[i] = pc[ps + tid + i * NT]; SM_FENCE(i); }
      }
#endif
#ifdef SM_LATE_P
      sm_dit<T, NT, Nx, LDY, 1, LGNY, LGNTW, G::MAXLG>(W, tw, tid);
      T px[PPT], py[PPT];
      {
        const T* pc = a.pcache + (size_t)(2 * kt) * ps + pb;
#pragma unroll
        for (int i = 0; i < PPT; ++i) { px[i] = pc[tid + i * NT]; py[i] = pc[ps + tid + i * NT]; SM_FENCE(i); }
      }
#else
      sm_dit<T, NT, Nx, LDY, 1, LGNY, LGNTW, G::MAXLG>(W, tw, tid);
#endif
      // velocity and RK bookkeeping on this thread's pixels (src/numerical_algorithms.jl:15-21); the next stage input goes back as packed columns
      const bool last = step == a.n - 1 && stage == 4;
      const RKCoef<T> rk = sm_coef(a, stage, last);
      T nxt[PPT];
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int p = tid + i * NT;
        const cx<T> z = W[(p >> LGNY) * LDY + (p & (Ny - 1))];
        const T k = px[i] * (sx * z.x) + py[i] * (sy * z.y);
        nxt[i] = rk_update(rk, k, y0[i], acc[i]);
        SM_FENCE(i);
      }
      if (last) break;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < PPT; ++i) { const int p = tid + i * NT; Wf[(p >> LGNY) * (2 * LDY) + (p & (Ny - 1))] = nxt[i]; SM_FENCE(i); }
      __syncthreads();
      kt += a.dir * (stage == 1 || stage == 3 ? 1 : 0);                     // stage times t, t + h/2, t + h/2, t + h
    }
#pragma unroll
  for (int i = 0; i < PPT; ++i) a.out[mb + tid0 + i * NT] = y0[i];
}

// ---- L'g / L'\g (src/lenseflow.jl:163-174): dy/dt = i lx F(p_x F^-1 y) + i ly F(p_y F^-1 y) on a Fourier half plane (F layout) ------------
template <typename T, int LGNY, int LGNX>
__global__ __launch_bounds__((SmallGeom<T, LGNY, LGNX>::NT)) void k_small_adj(SmallArgs<T> a) {
  using G = SmallGeom<T, LGNY, LGNX>;
  constexpr int NT = G::NT, Ny = G::Ny, Nx = G::Nx, M = G::M, Nyh = G::Nyh, LDY = G::LDY, PPT = G::PPT, FPT = G::FPT, LGM = G::LGM, LGNTW = G::LGNTW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* W = tw + G::NTW;
  T* Wf = reinterpret_cast<T*>(W);
  const int tid0 = threadIdx.x;
  const size_t sl = blockIdx.x, fb = sl * (size_t)G::NF;
  const size_t ps = (size_t)a.Bphi * G::NPIX, pb = (size_t)(a.Bphi == 1 ? 0 : sl / a.P) * G::NPIX;
  const cx<T>* in = reinterpret_cast<const cx<T>*>(a.in) + fb;
  cx<T>* out = reinterpret_cast<cx<T>*>(a.out) + fb;
  for (int i = tid0; i < G::NTW; i += NT) tw[i] = a.tw[i];
  // mode e = tid + i NT of the half plane [ky][xr] -> LDS W[xr][hslot(ky)] (consecutive lanes: consecutive xr, stride LDY)
  // (the state as separate real / imaginary arrays: arrays of cx<T> under the `e < NF` guards are not split into registers by the compiler --
  //  they went to scratch memory, 420 bytes per lane)
  T Y0r[FPT], Y0i[FPT], Yar[FPT], Yai[FPT];
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    const int e = min(tid0 + i * NT, G::NF - 1);                            // (spare lanes of the last group shadow the last mode and store nothing)
    const cx<T> v = in[e];
    Y0r[i] = v.x; Y0i[i] = v.y; Yar[i] = T(0); Yai[i] = T(0);
    if (tid0 + i * NT < G::NF) W[(e & (Nx - 1)) * LDY + sm_hslot<LGM>(e >> LGNX)] = v;
  }
  __syncthreads();
  const T sc = T(1) / (T(Nx) * T(Ny));
  int kt = a.k0;
  for (int step = 0; step < a.n; ++step)
    for (int stage = 1; stage <= 4; ++stage) {
      // the thread index as a value the compiler cannot see through: every address below is recomputed per stage.  With the plain index all of
      // them are invariants of the 4n-stage loop, get hoisted out of it and are kept live across it (65-126 spilled registers at 128^2)
      int tid = tid0;
      if constexpr (G::LAUNDER) asm volatile("" : "+v"(tid));
      // y = irfft2(Y): inverse x transform by ky slots, c2r by columns (packed: pixel (x, y) at the real view)
      sm_dit<T, NT, Nyh, 1, LDY, LGNX, LGNTW, G::MAXLG>(W, tw, tid);
      sm_c2r_pre<T, G>(W, tw, tid);
#ifndef SM_LATE_P
      T px[PPT], py[PPT];
      {
        const T* pc = a.pcache + (size_t)(2 * kt) * ps + pb;
#pragma unroll
        for (int i = 0; i < PPT; ++i) { px[i] = pc[tid + i * NT]; py[i] = pc[ps + tid + i * NT]; SM_FENCE(i); }
      }
      sm_dit<T, NT, Nx, LDY, 1, LGM, LGNTW, G::MAXLG>(W, tw, tid);
#else
      sm_dit<T, NT, Nx, LDY, 1, LGM, LGNTW, G::MAXLG>(W, tw, tid);
      T px[PPT], py[PPT];
      {
        const T* pc = a.pcache + (size_t)(2 * kt) * ps + pb;
#pragma unroll
        for (int i = 0; i < PPT; ++i) { px[i] = pc[tid + i * NT]; py[i] = pc[ps + tid + i * NT]; SM_FENCE(i); }
      }
#endif
      // (p_x y, p_y y) as one complex column per x
      T yv[PPT];
#pragma unroll
      for (int i = 0; i < PPT; ++i) { const int p = tid + i * NT; yv[i] = sc * Wf[(p >> LGNY) * (2 * LDY) + (p & (Ny - 1))]; SM_FENCE(i); }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < PPT; ++i) { const int p = tid + i * NT; W[(p >> LGNY) * LDY + (p & (Ny - 1))] = mk<T>(px[i] * yv[i], py[i] * yv[i]); SM_FENCE(i); }
      __syncthreads();
      // pair r2c: Ny-point DIF by columns, split into the two half spectra -> W[x][ky], W[x][Nyh + ky] (natural ky)
      sm_dif<T, NT, Nx, LDY, 1, LGNY, LGNTW, G::MAXLG>(W, W, tw, tid);
      {
        constexpr int NIT = (Nx * Nyh + NT - 1) / NT;
        cx<T> wa[NIT], wb[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int q = tid + i * NT, k = q / Nx, x = q - k * Nx;
          if (q < Nx * Nyh) {
            const cx<T> z = W[x * LDY + brevc<LGNY>(k)], zr = conj(W[x * LDY + brevc<LGNY>((Ny - k) & (Ny - 1))]);
            wa[i] = mk<T>(T(0.5) * (z.x + zr.x), T(0.5) * (z.y + zr.y));
            const cx<T> d = mk<T>(T(0.5) * (z.x - zr.x), T(0.5) * (z.y - zr.y));
            wb[i] = mk<T>(d.y, -d.x);                                       // d / i
          }
          SM_FENCE(i);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int q = tid + i * NT, k = q / Nx, x = q - k * Nx;
          if (q < Nx * Nyh) { W[x * LDY + k] = wa[i]; W[x * LDY + Nyh + k] = wb[i]; }
          SM_FENCE(i);
        }
        __syncthreads();
      }
      // fft_x of both members by ky (2 Nyh sequences), then the velocity and the RK update of the Fourier state
      sm_dif<T, NT, 2 * Nyh, 1, LDY, LGNX, LGNTW, G::MAXLG>(W, W, tw, tid);
      const bool last = step == a.n - 1 && stage == 4;
      const RKCoef<T> rk = sm_coef(a, stage, last);
      T nr[FPT], ni[FPT];
#pragma unroll
      for (int i = 0; i < FPT; ++i) {
        const int e = min(tid + i * NT, G::NF - 1);
        const int ky = e >> LGNX, xr = e & (Nx - 1);
        const cx<T> kv = mul_il(W[xr * LDY + ky], a.lx_r[xr]) + mul_il(W[xr * LDY + Nyh + ky], a.ly[ky]);
        nr[i] = rk_update(rk, kv.x, Y0r[i], Yar[i]);
        ni[i] = rk_update(rk, kv.y, Y0i[i], Yai[i]);
        SM_FENCE(i);
      }
      if (last) break;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < FPT; ++i) {
        const int e = tid + i * NT;
        if (e < G::NF) W[(e & (Nx - 1)) * LDY + sm_hslot<LGM>(e >> LGNX)] = mk<T>(nr[i], ni[i]);
        SM_FENCE(i);
      }
      __syncthreads();
      kt += a.dir * (stage == 1 || stage == 3 ? 1 : 0);
    }
#pragma unroll
  for (int i = 0; i < FPT; ++i) { const int e = tid0 + i * NT; if (e < G::NF) out[e] = mk<T>(Y0r[i], Y0i[i]); SM_FENCE(i); }
}

// ---- (grad L)': the delta flow (src/lenseflow.jl:176-214, src/flowops.jl:40-68) on (f [map], delta f [Fourier half plane]) -- both parts of a stage
// with the half plane pair of W used twice: the f part as in k_small_flow (its gradient stays in registers), then the delta-f part as in
// k_small_adj.  The stage's products L(delta f) grad f go to the per-stage buffer of the staged path (Flow::Wst), so that delta-phi is formed
// by the same end-of-flow quadrature (Flow::dphi_finish: k_dphi_reduce, five transforms, the l-multipliers).  Up to 64 x 64 pixels.
template <typename T> struct SmallDeltaArgs {
  SmallArgs<T> a;                         // a.in / a.out: the map f (may alias); the rest as for the flows
  cx<T>* df;                              // delta f, F layout, updated in place
  T* wst;                                 // [4n][2][slices][npix] products
  long slices;
};
template <typename T, int LGNY, int LGNX>
__global__ __launch_bounds__((SmallGeom<T, LGNY, LGNX>::NT)) void k_small_delta(SmallDeltaArgs<T> d) {
  using G = SmallGeom<T, LGNY, LGNX>;
  constexpr int NT = G::NT, Ny = G::Ny, Nx = G::Nx, M = G::M, Nyh = G::Nyh, LDY = G::LDY, PPT = G::PPT, FPT = G::FPT, LGM = G::LGM, LGNTW = G::LGNTW;
  const SmallArgs<T>& a = d.a;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* W = tw + G::NTW;
  T* Wf = reinterpret_cast<T*>(W);
  const int tid0 = threadIdx.x;
  const size_t sl = blockIdx.x, mb = sl * (size_t)G::NPIX, fb = sl * (size_t)G::NF;
  const size_t ps = (size_t)a.Bphi * G::NPIX, pb = (size_t)(a.Bphi == 1 ? 0 : sl / a.P) * G::NPIX;
  cx<T>* dfp = d.df + fb;
  for (int i = tid0; i < G::NTW; i += NT) tw[i] = a.tw[i];
  T y0[PPT], acc[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int p = tid0 + i * NT;
    y0[i] = a.in[mb + p]; acc[i] = T(0);
    Wf[(p >> LGNY) * (2 * LDY) + (p & (Ny - 1))] = y0[i];
  }
  T Y0r[FPT], Y0i[FPT], Yar[FPT], Yai[FPT], nr[FPT], ni[FPT];               // (separate real / imaginary arrays: see k_small_adj)
#pragma unroll
  for (int i = 0; i < FPT; ++i) {
    const cx<T> v = dfp[min(tid0 + i * NT, G::NF - 1)];
    Y0r[i] = v.x; Y0i[i] = v.y; Yar[i] = T(0); Yai[i] = T(0); nr[i] = v.x; ni[i] = v.y;
  }
  __syncthreads();
  const T sx = T(1) / (T(Nx) * T(Ny)), sy = T(1) / T(Ny), sc = sx;
  int kt = a.k0, it = 0;
  for (int step = 0; step < a.n; ++step)
    for (int stage = 1; stage <= 4; ++stage, ++it) {
      int tid = tid0;
      if constexpr (PPT >= 4) asm volatile("" : "+v"(tid));               // (both parts of a stage hold state: the addresses are recomputed from 4 pixels per thread on)
      // ---- f part: (d/dx f, d/dy f) of the stage input at this thread's pixels
      sm_dif<T, NT, Nx, LDY, 1, LGM, LGNTW, G::MAXLG>(W, W, tw, tid);
      sm_r2c_post<T, G>(W, tw, tid);
      sm_dif<T, NT, Nyh, 1, LDY, LGNX, LGNTW, G::MAXLG>(W, W + Nyh, tw, tid);
      sm_dit<T, NT, Nyh, 1, LDY, LGNX, LGNTW, G::MAXLG>(W + Nyh, tw, tid, [&](cx<T> v, int, int xs) { return mul_il(v, a.lx_r[xs]); });
      {
        constexpr int NIT = (Nx * Nyh + NT - 1) / NT;
        cx<T> za[NIT], zb[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int q = tid + i * NT, k = q / Nx, x = q - k * Nx;
          if (q < Nx * Nyh) {
            const int s = sm_hslot<LGM>(k);
            const cx<T> A = W[x * LDY + s], Gv = W[x * LDY + Nyh + s];
            const T l = a.ly[k];
            if (k == 0 || k == M) { za[i] = mk<T>(Gv.x, -l * A.y); zb[i] = za[i]; }
            else { za[i] = mk<T>(Gv.x - l * A.x, Gv.y - l * A.y); zb[i] = mk<T>(Gv.x + l * A.x, -Gv.y - l * A.y); }
          }
          SM_FENCE(i);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int q = tid + i * NT, k = q / Nx, x = q - k * Nx;
          if (q < Nx * Nyh) {
            W[x * LDY + brevc<LGNY>(k)] = za[i];
            if (k != 0 && k != M) W[x * LDY + brevc<LGNY>(Ny - k)] = zb[i];
          }
          SM_FENCE(i);
        }
        __syncthreads();
      }
      T px[PPT], py[PPT];
      {
        const T* pc = a.pcache + (size_t)(2 * kt) * ps + pb;
#pragma unroll
        for (int i = 0; i < PPT; ++i) { px[i] = pc[tid + i * NT]; py[i] = pc[ps + tid + i * NT]; }
      }
      sm_dit<T, NT, Nx, LDY, 1, LGNY, LGNTW, G::MAXLG>(W, tw, tid);
      T gx[PPT], gy[PPT];
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int p = tid + i * NT;
        const cx<T> z = W[(p >> LGNY) * LDY + (p & (Ny - 1))];
        gx[i] = sx * z.x; gy[i] = sy * z.y;
      }
      __syncthreads();
      // ---- delta f part: L(delta f) = irfft2 of the stage input (registers nr / ni) at this thread's pixels
#pragma unroll
      for (int i = 0; i < FPT; ++i) {
        const int e = tid + i * NT;
        if (e < G::NF) W[(e & (Nx - 1)) * LDY + sm_hslot<LGM>(e >> LGNX)] = mk<T>(nr[i], ni[i]);
      }
      __syncthreads();
      sm_dit<T, NT, Nyh, 1, LDY, LGNX, LGNTW, G::MAXLG>(W, tw, tid);
      sm_c2r_pre<T, G>(W, tw, tid);
      sm_dit<T, NT, Nx, LDY, 1, LGM, LGNTW, G::MAXLG>(W, tw, tid);
      T l[PPT];
#pragma unroll
      for (int i = 0; i < PPT; ++i) { const int p = tid + i * NT; l[i] = sc * Wf[(p >> LGNY) * (2 * LDY) + (p & (Ny - 1))]; }
      __syncthreads();
      // products for the delta-phi quadrature, the f velocity with its RK update, the delta-f velocity pair as one complex column per x
      const bool last = step == a.n - 1 && stage == 4;
      const RKCoef<T> rk = sm_coef(a, stage, last);
      T* w1 = d.wst + ((size_t)(2 * it) * d.slices + sl) * G::NPIX;
      T* w2 = w1 + (size_t)d.slices * G::NPIX;
      T nf[PPT];
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int p = tid + i * NT;
        __builtin_nontemporal_store(l[i] * gx[i], w1 + p);
        __builtin_nontemporal_store(l[i] * gy[i], w2 + p);
        const T k = px[i] * gx[i] + py[i] * gy[i];
        nf[i] = rk_update(rk, k, y0[i], acc[i]);
        W[(p >> LGNY) * LDY + (p & (Ny - 1))] = mk<T>(px[i] * l[i], py[i] * l[i]);
      }
      __syncthreads();
      sm_dif<T, NT, Nx, LDY, 1, LGNY, LGNTW, G::MAXLG>(W, W, tw, tid);
      {
        constexpr int NIT = (Nx * Nyh + NT - 1) / NT;
        cx<T> wa[NIT], wb[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int q = tid + i * NT, k = q / Nx, x = q - k * Nx;
          if (q < Nx * Nyh) {
            const cx<T> z = W[x * LDY + brevc<LGNY>(k)], zr = conj(W[x * LDY + brevc<LGNY>((Ny - k) & (Ny - 1))]);
            wa[i] = mk<T>(T(0.5) * (z.x + zr.x), T(0.5) * (z.y + zr.y));
            const cx<T> dd = mk<T>(T(0.5) * (z.x - zr.x), T(0.5) * (z.y - zr.y));
            wb[i] = mk<T>(dd.y, -dd.x);
          }
          SM_FENCE(i);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int q = tid + i * NT, k = q / Nx, x = q - k * Nx;
          if (q < Nx * Nyh) { W[x * LDY + k] = wa[i]; W[x * LDY + Nyh + k] = wb[i]; }
          SM_FENCE(i);
        }
        __syncthreads();
      }
      sm_dif<T, NT, 2 * Nyh, 1, LDY, LGNX, LGNTW, G::MAXLG>(W, W, tw, tid);
#pragma unroll
      for (int i = 0; i < FPT; ++i) {
        const int e = min(tid + i * NT, G::NF - 1);
        const int ky = e >> LGNX, xr = e & (Nx - 1);
        const cx<T> kv = mul_il(W[xr * LDY + ky], a.lx_r[xr]) + mul_il(W[xr * LDY + Nyh + ky], a.ly[ky]);
        nr[i] = rk_update(rk, kv.x, Y0r[i], Yar[i]);
        ni[i] = rk_update(rk, kv.y, Y0i[i], Yai[i]);
      }
      if (last) break;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < PPT; ++i) { const int p = tid + i * NT; Wf[(p >> LGNY) * (2 * LDY) + (p & (Ny - 1))] = nf[i]; }
      __syncthreads();
      kt += a.dir * (stage == 1 || stage == 3 ? 1 : 0);
    }
#pragma unroll
  for (int i = 0; i < PPT; ++i) a.out[mb + tid0 + i * NT] = y0[i];
#pragma unroll
  for (int i = 0; i < FPT; ++i) { const int e = tid0 + i * NT; if (e < G::NF) dfp[e] = mk<T>(Y0r[i], Y0i[i]); }
}

}  // namespace cmbl
