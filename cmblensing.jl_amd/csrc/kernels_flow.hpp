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

// Register budget of the fused column kernels: aim at two resident workgroups per CU (4 waves per SIMD for 512 threads) when the
// tile is small enough (R <= 4 single precision) that the cap costs at most a handful of spilled registers.
template <typename T> constexpr int col_min_waves(int R, int NT) { return (sizeof(T) == 4 && R <= 4 && NT >= 256) ? 4 : 1; }

// shapes whose column workgroups are alone on their CU and walk several tiles with software-pipelined loads (delta_y_body_pipelined)
template <typename T> constexpr bool col_pipelined(int lgm) { return sizeof(T) == 8 && lgm >= 10; }

// pcx/pcy: p(t) of the current stage time from the per-phi cache (k_pcache), or nullptr -> formed from the five maps
template <typename T> struct PhiMaps { const T *gx, *gy, *hxx, *hyx, *hyy; int Bphi; const T *pcx, *pcy; };

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

// streaming store of one packed pair (the per-stage products are written once and read once, by k_dphi_reduce at the end of the flow:
// kept out of the caches the stage arrays live in; (∇L)† 1.195 -> 1.180 ms.  Streaming LOADS of the p(t) cache, which both pol slices
// read, measured slower: L*f 0.579 -> 0.600 ms)
typedef double nt_d2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_store(cx<float>* p, cx<float> v) { __builtin_nontemporal_store(f2{v.x, v.y}, reinterpret_cast<f2*>(p)); }
__device__ __forceinline__ void nt_store(cx<double>* p, cx<double> v) { __builtin_nontemporal_store(nt_d2s{v.x, v.y}, reinterpret_cast<nt_d2s*>(p)); }
// ---------------------------------------------------------------------------------------------
// Register-phase helpers.  A thread owns R "pairs": pair e = threadIdx.x + i*NT of the tile, column c = e >> LGM, rows
// (2jj, 2jj+1) with jj = e & (M-1) -- i.e. the cx<T> at index e of a map column tile viewed as packed pairs.
template <typename T>
__device__ __forceinline__ void load_p_pair(const PhiMaps<T>& ph, size_t pbase, unsigned e, T t, cx<T>& px, cx<T>& py, cx<T>& m11, cx<T>& m12, cx<T>& m22) {
  const cx<T> gx = at32(reinterpret_cast<const cx<T>*>(ph.gx) + pbase, e), gy = at32(reinterpret_cast<const cx<T>*>(ph.gy) + pbase, e);
  const cx<T> hxx = at32(reinterpret_cast<const cx<T>*>(ph.hxx) + pbase, e), hyx = at32(reinterpret_cast<const cx<T>*>(ph.hyx) + pbase, e);
  const cx<T> hyy = at32(reinterpret_cast<const cx<T>*>(ph.hyy) + pbase, e);
  flow_pm(t, gx.x, gy.x, hxx.x, hyx.x, hyy.x, px.x, py.x, m11.x, m12.x, m22.x);
  flow_pm(t, gx.y, gy.y, hxx.y, hyx.y, hyy.y, px.y, py.y, m11.y, m12.y, m22.y);
}
// p(t) only (forward / adjoint / delta-f parts): two cached maps instead of five -- the phi maps are the larger part of what a
// column workgroup requests before its first transform
template <typename T>
__device__ __forceinline__ void load_p_only(const PhiMaps<T>& ph, size_t pbase, unsigned e, T t, cx<T>& px, cx<T>& py) {
  if (ph.pcx) {
    px = at32(reinterpret_cast<const cx<T>*>(ph.pcx) + pbase, e); py = at32(reinterpret_cast<const cx<T>*>(ph.pcy) + pbase, e);
  } else {
    cx<T> m11, m12, m22;
    load_p_pair(ph, pbase, e, t, px, py, m11, m12, m22);
  }
}
// p(t_k), k = 0..n2 (t_k = k/n2, the 2n+1 RK stage times; src/lenseflow.jl:131-142) for every pixel: out[k][2][ntot]
template <typename T>
__global__ __launch_bounds__(NTP) void k_pcache(PhiMaps<T> ph, T* __restrict__ out, long ntot, int n2) {
  for (long i = (long)blockIdx.x * NTP + threadIdx.x; i < ntot; i += (long)gridDim.x * NTP) {
    const T gx = ph.gx[i], gy = ph.gy[i], hxx = ph.hxx[i], hyx = ph.hyx[i], hyy = ph.hyy[i];
    for (int k = 0; k <= n2; ++k) {
      T px, py, m11, m12, m22;
      flow_pm((T)k / (T)n2, gx, gy, hxx, hyx, hyy, px, py, m11, m12, m22);
      out[(size_t)(2 * k) * ntot + i] = px; out[(size_t)(2 * k + 1) * ntot + i] = py;
    }
  }
}
// tile of N-point complex columns after a pair DIT: slots pad(2jj), pad(2jj)+1 hold (x,y)[2jj], (x,y)[2jj+1]
template <typename T>
__device__ __forceinline__ void read_pair(const cx<T>* col, int jj, T scale, cx<T>& x, cx<T>& y) {
  const cx<T> z0 = col[pad(2 * jj)], z1 = col[pad(2 * jj) + 1];
  x = mk<T>(z0.x * scale, z1.x * scale); y = mk<T>(z0.y * scale, z1.y * scale);
}
template <typename T>
__device__ __forceinline__ void write_pair(cx<T>* col, int jj, cx<T> x, cx<T> y) {
  col[pad(2 * jj)] = mk<T>(x.x, y.x); col[pad(2 * jj) + 1] = mk<T>(x.y, y.y);
}
template <typename T> __device__ __forceinline__ cx<T> mul_il(cx<T> v, T l) { return mk<T>(-l * v.y, l * v.x); }   // i*l*v
template <typename T> __device__ __forceinline__ cx<T> pmul(cx<T> a, cx<T> b) { return mk<T>(a.x * b.x, a.y * b.y); }   // elementwise on a pair

// ---------------------------------------------------------------------------------------------
// N-point (pair) transforms of a column tile with the TOP radix-2 level fused into the register phase, like the row kernels:
// a thread that owns the packed pairs jj and jj + M/2 of a column holds samples a and a + M (a = 2jj, 2jj+1), so it can
//   * after the inverse sub-stages combine the two halves:  z[a] = u[a] + conj(W_N^a) v[a],  z[a + M] = u[a] - conj(W_N^a) v[a]
//   * before the forward sub-stages split them:             u[a] = z[a] + z[a + M],          v[a] = (z[a] - z[a + M]) W_N^a
// and the remaining levels run on the 2C half-columns, one per wavefront (WorkRows<NT/C, C>: 1024 points = 2 waves x 3 radix-8
// stages of one butterfly per lane; no barrier between the stages, every thread busy -- the cooperative radix-16 stages had 256
// butterflies for 512 threads).  PairMap gives the pair ownership: split tiles own (jj, jj + M/2) of R/2 columns, the others the
// interleaved pairs tid + i*NT as before.
// samples (x + i y)[a], a = 2jj, 2jj+1 and a + M from the two halves u (slots a) and v (slots a + M) of an N-point column
template <typename T>
__device__ __forceinline__ void read_pair_dit(const cx<T>* col, int jj, int M, const cx<T>* tw, T scale, cx<T>& x0, cx<T>& y0, cx<T>& x1, cx<T>& y1) {
  using V = typename vreg<T>::type;
  const cx<T>* p = col + pad(2 * jj);
  const V u0 = vload(p), u1 = vload(p + 1);
  const V t0 = vmulc(vload(p + pad(M)), vload(tw + 2 * jj)), t1 = vmulc(vload(p + pad(M) + 1), vload(tw + 2 * jj + 1));
  const cx<T> a0 = vcx(vscale(vadd(u0, t0), scale)), a1 = vcx(vscale(vadd(u1, t1), scale));
  const cx<T> b0 = vcx(vscale(vsub(u0, t0), scale)), b1 = vcx(vscale(vsub(u1, t1), scale));
  x0 = mk<T>(a0.x, a1.x); y0 = mk<T>(a0.y, a1.y); x1 = mk<T>(b0.x, b1.x); y1 = mk<T>(b0.y, b1.y);
}
template <typename T>
__device__ __forceinline__ void write_pair_dif(cx<T>* col, int jj, int M, const cx<T>* tw, cx<T> x0, cx<T> y0, cx<T> x1, cx<T> y1) {
  using V = typename vreg<T>::type;
  cx<T>* p = col + pad(2 * jj);
  const V za0 = vmake(x0.x, y0.x), za1 = vmake(x0.y, y0.y), zb0 = vmake(x1.x, y1.x), zb1 = vmake(x1.y, y1.y);
  vstore(p, vadd(za0, zb0));
  vstore(p + 1, vadd(za1, zb1));
  vstore(p + pad(M), vmul(vsub(za0, zb0), vload(tw + 2 * jj)));
  vstore(p + pad(M) + 1, vmul(vsub(za1, zb1), vload(tw + 2 * jj + 1)));
}
// inverse N-point transform of the tile (bit-reversed input already committed and synchronised) -> (x, y) pairs, scaled
template <typename T, int R, int NT, int LGM, int LD>
__device__ __forceinline__ void npt_inverse_read(cx<T>* s, const cx<T>* tw, T scale, cx<T> (&x)[R], cx<T> (&y)[R], int tid = (int)threadIdx.x) {
  using PM = PairMap<R, NT, LGM>;
  constexpr int M = 1 << LGM, LGN = LGM + 1, C = PM::C;
  if constexpr (PM::split) {
    fft_dit_w<T, LD, LGN, LGN, PM::XLG, 1>(s, WorkRows<NT / C, C>{1, C, tid}, tw);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R; i += 2) {
      const int e = PM::e(i, tid), c = e >> LGM, jj = e & (M - 1);
      read_pair_dit(s + c * LD, jj, M, tw, scale, x[i], y[i], x[i + 1], y[i + 1]);
    }
  } else {
    fft_dit<T, NT, LD, LGN, LGN, CMBL_YLGN>(s, C, tw);
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int e = PM::e(i, tid), c = e >> LGM, jj = e & (M - 1);
      read_pair(s + c * LD, jj, scale, x[i], y[i]);
    }
  }
}
// (x, y) pairs -> forward N-point transform of the tile (bit-reversed output, synchronised); the tile must be free (sync before)
// xy(i, x, y) yields pair i of the thread
template <typename T, int R, int NT, int LGM, int LD, typename XY>
__device__ __forceinline__ void npt_write_forward(cx<T>* s, const cx<T>* tw, XY&& xy, int tid = (int)threadIdx.x) {
  using PM = PairMap<R, NT, LGM>;
  constexpr int M = 1 << LGM, LGN = LGM + 1, C = PM::C;
  if constexpr (PM::split) {
#pragma unroll
    for (int i = 0; i < R; i += 2) {
      const int e = PM::e(i, tid), c = e >> LGM, jj = e & (M - 1);
      cx<T> x0, y0, x1, y1;
      xy(i, x0, y0); xy(i + 1, x1, y1);
      write_pair_dif(s + c * LD, jj, M, tw, x0, y0, x1, y1);
    }
    __syncthreads();
    fft_dif_w<T, LD, LGN, LGN, PM::XLG, 1>(s, WorkRows<NT / C, C>{1, C, tid}, tw);
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int e = PM::e(i, tid), c = e >> LGM, jj = e & (M - 1);
      cx<T> x, y;
      xy(i, x, y);
      write_pair(s + c * LD, jj, x, y);
    }
    __syncthreads();
    fft_dif<T, NT, LD, LGN, LGN, CMBL_YLGN>(s, C, tw);
  }
}

// Touch prefetch (round 5).  A column launch of several residency rounds with ONE workgroup per CU (2048 rows in double precision: 244 registers x
// 512 threads is the CU's register file) has nobody to fill the ~9 us a workgroup waits for its first tile (profiles/r05_stamps_delta_cols_2048_f64.txt).
// The workgroup that will REPLACE this one is, to dispatch-order accuracy, `dist` blocks ahead in the grid, and -- block b is observed on XCD b % 8
// (speed only) -- on this XCD when dist is a multiple of 8.  So this workgroup reads ONE dword of every 128-byte line of that block's head tiles
// into the shared L2, as soon as its own last global load has been consumed: NL lines per tile = one or two loads per thread and array.
// The loads are LDS-DMA (`global_load_lds_dword`) into a 256-byte pad behind the tile that nothing ever reads: they have NO destination register.
// (A first version loaded into registers the compiler did not know were written asynchronously; in the register-starved double-precision row kernel
// the compiler moved such a "value" to an AGPR and reused the VGPR, the late load landed on an address register, and the run died with a memory
// fault -- the column kernels happened to survive the same construction, which is not a guarantee.)  Being inline asm they are also invisible to the
// compiler's s_waitcnt bookkeeping: nothing waits for them; they return during the transforms that follow, and `drain()` -- one s_waitcnt before the
// workgroup's last stores -- makes sure that none can land in LDS that already belongs to another workgroup.
template <typename T, int NT, int LGM, int C, int NARR> struct TouchTiles {
  static constexpr int M = 1 << LGM, NyhP = mixed_rows(M + 1), RPL = 128 / (mixw<T>() * (int)sizeof(cx<T>)), NL = (M + 1 + RPL - 1) / RPL;
  static constexpr int NB = (C + mixw<T>() - 1) / mixw<T>(), K = (NL * NB + NT - 1) / NT;
  static constexpr int PAD_BYTES = 256;                                   // one wave-instruction of 64 lanes x 4 bytes; every wave uses the same pad
  // pad: LDS byte address of the pad (wave-uniform)
  __device__ __forceinline__ static void issue(const cx<T>* const (&g)[NARR > 0 ? NARR : 1] /*slice bases*/, int x0, unsigned pad) {
    const unsigned m0v = (unsigned)__builtin_amdgcn_readfirstlane((int)pad);
#pragma unroll
    for (int a = 0; a < NARR; ++a) {
      const cx<T>* tg = tile_base(g[a], x0, NyhP);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        int i = threadIdx.x + k * NT;
        if (i >= NL * NB) i = NL * NB - 1;                                  // the spare lanes of the last round re-touch the last line
        const int blk = i / NL, ln = i - blk * NL;
        const void* q = tg + ((size_t)blk * NyhP + (size_t)ln * RPL) * mixw<T>();
        unsigned keep;
        // M0 (the LDS-DMA destination base) is compiler-reserved: set and restored inside the statement (cdna_hip_programming.md, LDS-DMA recipe)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(q), "s"(m0v) : "memory");
      }
    }
  }
  __device__ __forceinline__ static void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
};

#ifndef CMBL_TOUCH_ALL
#define CMBL_TOUCH_ALL 0            // 1: compile the touch prefetch into every column kernel (experiment builds)
#endif
// compiled into the one-workgroup-per-CU shapes only (every other instantiation is byte for byte what it was)
template <typename T> constexpr bool col_touch(int lgm) { return CMBL_TOUCH_ALL || (sizeof(T) == 8 && lgm >= 10); }
// LDS byte address of the touch pad: right behind the column tile (twiddles M + C columns of LD slots); the launch asks for TOUCH_PAD bytes more
constexpr int TOUCH_PAD = 256;
template <typename T, int C, int LD> __device__ __forceinline__ unsigned touch_pad(int M) { return (unsigned)(((size_t)M + (size_t)C * LD) * sizeof(cx<T>)); }
// block -> (slice offset in the mixed layout, first column) of the workgroup `pf` blocks ahead of this one in a (tiles, slices) column grid
template <int C> __device__ __forceinline__ bool touch_target(int pf, size_t sl, int NyhP, int Nx, size_t& mo2, int& x02) {
  const unsigned t = blockIdx.y * gridDim.x + blockIdx.x + (unsigned)pf;
  if (pf <= 0 || t >= gridDim.x * gridDim.y) return false;
  mo2 = (sl - blockIdx.y + t / gridDim.x) * (size_t)NyhP * Nx;
  x02 = xcd_tile((int)(t % gridDim.x), gridDim.x) * C;
  return true;
}
// ---------------------------------------------------------------------------------------------
// Forward / inverse flow, column kernel.  grid (Nx/C, P*B).  LDS: twY[M] + C*tile_ld(2M).
//   in : A  = rfft_y(f_s)  (mixed), Gx = d/dx f_s y-transformed (mixed; from k_x_fft<MODE 2>)
//   out: y0/acc updated, Anext = rfft_y(f_{s+1}) (mixed)
// d/dx f and d/dy f come out of ONE N-point complex inverse transform of Gx + i*(i*ly*A).
template <typename T> struct FlowYArgs {
  const cx<T>* A; const cx<T>* Gx; cx<T>* Anext;
  T* y0; T* acc;
  const T* y0r;             // where the flow state is READ (the caller's input during the first RK step, then y0: no copy for out != in)
  PhiMaps<T> ph;
  const cx<T>* twY; const T* ly;
  int Nx, P;
  RKCoef<T> rk;
  int emit_last;            // the last stage also writes Anext = rfft_y(final state): the caller continues in Fourier space without a y pass
  int pf;                   // > 0: touch prefetch distance in blocks (TouchTiles)
};

template <typename T, int R, int NT, int LGM>
__global__ __launch_bounds__(NT, col_min_waves<T>(R, NT)) void k_flow_y_fwd(FlowYArgs<T> a) {
  using G = ColTile<R, NT, LGM>;
  constexpr int M = G::M, LGN = G::LGN, LD = G::LDN, Nyh = G::Nyh, C = G::C, LGC = G::LGC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;                            // half of the twiddle circle (see row_tw)
  const int Nx = a.Nx, x0 = xcd_tile(blockIdx.x, gridDim.x) * C;
  const size_t sl = blockIdx.y;
  const int bphi = a.ph.Bphi == 1 ? 0 : (int)(sl / a.P);
  const T invNy = T(1) / T(2 * M);
  constexpr int NyhP = mixed_rows(Nyh);
  const size_t moff = sl * (size_t)NyhP * Nx;
  // everything this workgroup needs from HBM is requested before anything is waited for
  TwStage<T, NT, M> twr;
  PairStage<T, NT, LGN, LGC> ps;
  twr.issue(a.twY);
  ps.issue(a.Gx + moff, a.A + moff, a.ly, Nx, x0);
  const size_t pbase = ((size_t)bphi * Nx + x0) * M, mbase = (sl * Nx + x0) * (size_t)M;
  cx<T>* y0p = reinterpret_cast<cx<T>*>(a.y0) + mbase;
  const cx<T>* y0rp = reinterpret_cast<const cx<T>*>(a.y0r) + mbase;
  cx<T>* accp = reinterpret_cast<cx<T>*>(a.acc) + mbase;
  using PM = PairMap<R, NT, LGM>;
  cx<T> px[R], py[R], y0[R], acc[R];
  // the pair tile alone gates the first transform: it is requested first and committed before p(t) and the RK state are asked for,
  // which then fly during the transform (as in delta_y_body; A/B profiles/r04_ab_load_order.txt: L*f 0.564 -> 0.552 ms, CG iteration
  // 1.341 -> 1.318 ms at 1024^2 QU.  The same order in k_adj_y, whose first transform is the short packed-real one, measured +1.5 %)
  twr.commit(tw);
  ps.template commit<LD>(s);
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const unsigned e = PM::e(i);
    load_p_only(a.ph, pbase, e, a.rk.t, px[i], py[i]);
    y0[i] = at32(y0rp, e);
    acc[i] = a.rk.stage == 1 ? mk<T>(0, 0) : at32(accp, e);
  }
  __syncthreads();
  cx<T> fn[R], dx[R], dy[R];
  npt_inverse_read<T, R, NT, LGM, LD>(s, tw, invNy, dx, dy);
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const unsigned e = PM::e(i);
    const cx<T> kv = pmul(px[i], dx[i]) + pmul(py[i], dy[i]);
    fn[i] = rk_update(a.rk, kv, y0[i], acc[i]);
    if (a.rk.stage == 4) at32(y0p, e) = y0[i]; else at32(accp, e) = acc[i];
  }
  if (a.rk.last && !a.emit_last) return;
  using Touch = TouchTiles<T, NT, LGM, C, 2>;
  size_t mo2 = 0; int x02 = 0;
  bool do_touch = false;
  if constexpr (col_touch<T>(LGM)) {
    do_touch = touch_target<C>(a.pf, sl, NyhP, Nx, mo2, x02);
    if (do_touch) { const cx<T>* const heads[2] = {a.Gx + mo2, a.A + mo2}; Touch::issue(heads, x02, touch_pad<T, C, LD>(M)); }
  }
  __syncthreads();
  mpt_write_forward<T, R, NT, LGM, LD>(s, tw, [&](int i) { return fn[i]; });
  if constexpr (col_touch<T>(LGM)) { if (do_touch) Touch::drain(); }
  half_store<T, NT, LD, LGM, LGC>(s, a.Anext + moff, tw, x0);
}

// ---------------------------------------------------------------------------------------------
// Adjoint flow, column kernel: H = ifft_x(Y) (mixed)  ->  Wx = rfft_y(px*y), Wy' = i*ly*rfft_y(py*y) (mixed)
//   (src/lenseflow.jl:163-174).  The two forward transforms are ONE N-point complex transform of px*y + i*py*y.
template <typename T> struct AdjYArgs {
  const cx<T>* H; cx<T>* Wx; cx<T>* Wy;
  PhiMaps<T> ph;
  const cx<T>* twY; const T* ly;
  int Nx, P;
  T t;
  int pf;                   // > 0: touch prefetch distance in blocks (TouchTiles)
};

template <typename T, int R, int NT, int LGM>
__global__ __launch_bounds__(NT, col_min_waves<T>(R, NT)) void k_adj_y(AdjYArgs<T> a) {
  using G = ColTile<R, NT, LGM>;
  constexpr int M = G::M, LGN = G::LGN, LD = G::LDN, Nyh = G::Nyh, C = G::C, LGC = G::LGC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;                            // half of the twiddle circle (see row_tw)
  const int Nx = a.Nx, x0 = xcd_tile(blockIdx.x, gridDim.x) * C;
  const size_t sl = blockIdx.y;
  const int bphi = a.ph.Bphi == 1 ? 0 : (int)(sl / a.P);
  const T invNy = T(1) / T(2 * M);
  constexpr int NyhP = mixed_rows(Nyh);
  const size_t moff = sl * (size_t)NyhP * Nx;
  TwStage<T, NT, M> twr;
  HalfStage<T, NT, LGM, LGC> tl;
  twr.issue(a.twY);
  tl.issue(a.H + moff, a.twY, x0);
  const size_t pbase = ((size_t)bphi * Nx + x0) * M;
  using PM = PairMap<R, NT, LGM>;
  cx<T> px[R], py[R];
#pragma unroll
  for (int i = 0; i < R; ++i) load_p_only(a.ph, pbase, (unsigned)PM::e(i), a.t, px[i], py[i]);
  T lyr[G::RZ];                                               // ly of this thread's half-spectrum entries (used after the last transform)
#pragma unroll
  for (int i = 0; i < G::RZ; ++i) { const int e = threadIdx.x + i * NT; if (e < C * (M + 1)) lyr[i] = a.ly[e >> LGC]; }
  twr.commit(tw);
  tl.template commit<LD>(s);
  __syncthreads();
  cx<T> yv[R];
  mpt_inverse_read<T, R, NT, LGM, LD>(s, tw, invNy, yv);
  using Touch = TouchTiles<T, NT, LGM, C, 1>;
  size_t mo2 = 0; int x02 = 0;
  bool do_touch = false;
  if constexpr (col_touch<T>(LGM)) {
    // every global load of this thread must have been consumed before the untracked touch loads are issued (a compiler wait for an OLDER load would
    // otherwise also wait for them): form the products now (the same arithmetic the transform's writer does)
#pragma unroll
    for (int i = 0; i < R; ++i) { px[i] = pmul(px[i], yv[i]); py[i] = pmul(py[i], yv[i]); }
    do_touch = touch_target<C>(a.pf, sl, NyhP, Nx, mo2, x02);
    if (do_touch) { const cx<T>* const heads[1] = {a.H + mo2}; Touch::issue(heads, x02, touch_pad<T, C, LD>(M)); }
  }
  __syncthreads();
  if constexpr (col_touch<T>(LGM)) npt_write_forward<T, R, NT, LGM, LD>(s, tw, [&](int i, cx<T>& x, cx<T>& y) { x = px[i]; y = py[i]; });
  else npt_write_forward<T, R, NT, LGM, LD>(s, tw, [&](int i, cx<T>& x, cx<T>& y) { x = pmul(px[i], yv[i]); y = pmul(py[i], yv[i]); });
  if constexpr (col_touch<T>(LGM)) { if (do_touch) Touch::drain(); }
  cx<T>* Wx = tile_base(a.Wx + moff, x0, NyhP); cx<T>* Wy = tile_base(a.Wy + moff, x0, NyhP);
  pair_split<T, NT, LD, LGN, LGC, G::RZ>(s, [&](int i, int k, int c, cx<T> A, cx<T> B) {
    const unsigned gi = tile_off<T, C>(k, c, x0, NyhP);
    handoff_store<T, wt_cols<T>(C, M)>(Wx, gi, A); handoff_store<T, wt_cols<T>(C, M)>(Wy, gi, mul_il(B, lyr[i]));
  });
}

// Adjoint flow, row kernel:  k = i*lx*fft_x(Wx) + fft_x(Wy')  -> RK update of the Fourier state (F layout)
//   -> Hnext = ifft_x(next stage input) (mixed).     grid = slices * ceil(Nyh / RPW) row groups.  LDS: twX + 2*RPW*row_ld(Nx)
template <typename T> struct AdjXArgs {
  const cx<T>* Wx; const cx<T>* Wy; cx<T>* Y0; cx<T>* acc; cx<T>* Hnext;
  const cx<T>* twX; const T* lx_r;
  int Nyh;
  RKCoef<T> rk;
};

template <typename T, int LGNX, int RPW>
__device__ __forceinline__ void adj_x_body(const AdjXArgs<T>& a, unsigned char* smem, unsigned blk, unsigned nblk) {
  constexpr int Nx = 1 << LGNX, NH = Nx >> 1, LD = row_ld(Nx), NT = row_nt(RPW), PF = NH >= 64 ? NH / 64 : 1;
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + row_tw<T>(Nx);
  cx<T>* s2 = s + RPW * LD;                                   // second row set: sequence RPW + row, owned by the same threads
  const RowGroup rg = row_group<RPW>(blk, a.Nyh, nblk);
  const int NyhP = mixed_rows(a.Nyh);
  const size_t mo = (size_t)rg.sl * NyhP * Nx;
  TwStage<T, NT, row_tw<T>(Nx)> twr;
  twr.issue(a.twX);
  {
    cx<T>* const sa[2] = {s, s2};
    const cx<T>* const ga[2] = {a.Wx + mo, a.Wy + mo};
    rows_load_mixed_dif<T, LGNX, RPW, 2>(sa, ga, a.twX, NyhP, rg.ky0, rg.nr);
  }
  twr.commit(tw);
  __syncthreads();
  // all forward stages but the last on both row sets; the last forward butterfly of Wx and Wy, the velocity  k = i lx Wx^ + Wy^,
  // the RK update of the Fourier state and the first inverse butterfly of the next stage input then share one register phase:
  // a butterfly owns r adjacent slots = r adjacent x of the F layout (64 bytes of Y0 / acc per lane, requested before the LDS reads)
  using V = typename vreg<T>::type;
  constexpr int XLG = row_xlg(LGNX), NS = num_stages(LGNX - 1, XLG), LG = stage_lg(LGNX - 1, NS - 1, XLG), r = 1 << LG;
  constexpr int VE = 16 / (int)sizeof(cx<T>), NV = r / VE;
  const WorkRows<ROW_RT, RPW, row_tw_quarter<T>(LGNX)> wk{1, rg.nr};
  const T inv = T(1) / T(Nx);
  const bool last = a.rk.last;
  // The Fourier state of the thread's butterflies (Y0, acc: 64 contiguous bytes each per butterfly) and lx do not depend on the
  // transforms: they are requested BEFORE the forward stages and arrive while those run, instead of inside the register phase, where
  // the wave then waited a full memory latency (A/B profiles/r04_ab_adjx_prefetch.txt: L'g 0.866 -> 0.835 ms and the Wiener-CG
  // iteration 1.726 -> 1.694 ms at 1024^2 T+QU; QU within the run-to-run spread)
  constexpr int ITEMS = 1 << (LGNX - LG), IT = (ROW_RT == 128 && ITEMS / 2 >= 64) ? ITEMS / 128 : 1;
  CxVec<T> y0p[IT][NV], acp[IT][NV];
  T lxp[IT][r];
  auto prefetch = [&]() {
    int it = 0;
    wk.template each<LGNX - LG>([&](int row, int rr) {
      const int b0 = rr << LG;
      const size_t g0 = ((size_t)rg.sl * a.Nyh + rg.ky0 + row) * Nx + b0;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        y0p[it][i] = *reinterpret_cast<const CxVec<T>*>(a.Y0 + g0 + i * VE);
        if (a.rk.stage != 1) acp[it][i] = *reinterpret_cast<const CxVec<T>*>(a.acc + g0 + i * VE);
      }
#pragma unroll
      for (int i = 0; i < r; ++i) lxp[it][i] = a.lx_r[b0 + i];
      ++it;
    });
  };
  prefetch();
  fft_dif_w<T, LD, LGNX, LGNX, XLG, 1, WorkRows<ROW_RT, RPW, row_tw_quarter<T>(LGNX)>, 0, 1>(s, WorkRows<ROW_RT, RPW, row_tw_quarter<T>(LGNX)>{2, rg.nr}, tw);
  int item = 0;
  wk.template each<LGNX - LG>([&](int row, int rr) {
    const int b0 = rr << LG;
    const size_t g0 = ((size_t)rg.sl * a.Nyh + rg.ky0 + row) * Nx + b0;
    CxVec<T> y0v[NV], acv[NV];
    T lxr[r];
#pragma unroll
    for (int i = 0; i < NV; ++i) { y0v[i] = y0p[item][i]; acv[i] = acp[item][i]; }
#pragma unroll
    for (int i = 0; i < r; ++i) lxr[i] = lxp[item][i];
    ++item;
    cx<T>* p1 = s + row * LD + pad(b0);
    const cx<T>* p2 = s2 + row * LD + pad(b0);
    V va[r], vb[r], u[r];
#pragma unroll
    for (int m = 0; m < r; ++m) { va[m] = vload(p1 + pad(m)); vb[m] = vload(p2 + pad(m)); }
    dft<T, LG, false>(va);
    dft<T, LG, false>(vb);
#pragma unroll
    for (int k = 0; k < r; ++k) {
      const int j = brevc<LG>(k);                               // X_k sits at slot b0 + j
      const cx<T> kv = mul_il(vcx(va[dft_loc<LG>(k)]), lxr[j]) + vcx(vb[dft_loc<LG>(k)]);
      cx<T> y0 = y0v[j / VE].v[j % VE], acc = a.rk.stage == 1 ? mk<T>(0, 0) : acv[j / VE].v[j % VE];
      const cx<T> fn = rk_update(a.rk, kv, y0, acc);
      y0v[j / VE].v[j % VE] = y0; acv[j / VE].v[j % VE] = acc;
      u[k] = vfrom(inv * fn);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (a.rk.stage == 4) *reinterpret_cast<CxVec<T>*>(a.Y0 + g0 + i * VE) = y0v[i];
      else *reinterpret_cast<CxVec<T>*>(a.acc + g0 + i * VE) = acv[i];
    }
    if (!last) {
      dft<T, LG, true>(u);
#pragma unroll
      for (int m = 0; m < r; ++m) vstore(p1 + pad(m), u[dft_loc<LG>(m)]);
    }
  });
  if (last) return;
  wk.sync();
  fft_dit_w<T, LD, LGNX, LGNX, XLG, 1, WorkRows<ROW_RT, RPW, row_tw_quarter<T>(LGNX)>, NoPre, NS - 2>(s, wk, tw);
  __syncthreads();
  rows_store_mixed_dit<T, LGNX, RPW>(s, a.Hnext + mo, tw, NyhP, rg.ky0, rg.nr, T(1));
}

template <typename T, int LGNX, int RPW>
__global__ __launch_bounds__(row_nt(RPW), row_min_waves<T>()) void k_adj_x(AdjXArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  adj_x_body<T, LGNX, RPW>(a, smem, blockIdx.x, gridDim.x);
}

// x-derivative row pass as a device function (same as k_x_fft<MODE 2>)
template <typename T> struct GradXArgs { const cx<T>* in; cx<T>* out; const cx<T>* twX; T dlx_over_Nx; int Nyh; };
template <typename T, int LGNX, int RPW>
__device__ __forceinline__ void grad_x_body(const GradXArgs<T>& g, unsigned char* smem, unsigned blk, unsigned nblk) {
  constexpr int Nx = 1 << LGNX, LD = row_ld(Nx), NT = row_nt(RPW);
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + row_tw<T>(Nx);
  const RowGroup rg = row_group<RPW>(blk, g.Nyh, nblk);
  const int NyhP = mixed_rows(g.Nyh);
  const size_t mo = (size_t)rg.sl * NyhP * Nx;
  TwStage<T, NT, row_tw<T>(Nx)> twr;
  twr.issue(g.twX);
  {
    cx<T>* const sa[1] = {s};
    const cx<T>* const ga[1] = {g.in + mo};
    rows_load_mixed_dif<T, LGNX, RPW, 1>(sa, ga, g.twX, NyhP, rg.ky0, rg.nr);
  }
  twr.commit(tw);
  __syncthreads();
  const WorkRows<ROW_RT, RPW, row_tw_quarter<T>(LGNX)> wk{1, rg.nr};
  // i*lx/Nx multiply between the last forward and the first inverse butterfly, in registers: slot i holds kx = bitrev(i), lx = dlx * signed(kx)
  const T dl = g.dlx_over_Nx;
  fft_dif_mid_dit_w<T, LD, LGNX, LGNX, row_xlg(LGNX), 1>(s, wk, tw, [dl](int, int b0, int j, typename vreg<T>::type v) {
    constexpr int LGL = stage_lg(LGNX - 1, num_stages(LGNX - 1, row_xlg(LGNX)) - 1, row_xlg(LGNX));
    return vfrom(mul_il_slot<T, LGNX, LGL>(vcx(v), b0, j, dl));
  });
  __syncthreads();
  rows_store_mixed_dit<T, LGNX, RPW>(s, g.out + mo, tw, NyhP, rg.ky0, rg.nr, T(1));
}

// ---------------------------------------------------------------------------------------------
// delta flow (src/lenseflow.jl:176-214), per-(pol,batch) column kernel: does the f part (== k_flow_y_fwd),
// the delta-f part (== k_adj_y) and writes the spin-adjoint partial products
//   w1p = L(df) * d/dx f,  w2p = L(df) * d/dy f     (maps, one pair per pol and stage; reduced over pols and stages by k_dphi_reduce)
template <typename T> struct DeltaYArgs {
  FlowYArgs<T> f;           // f part
  const cx<T>* H; cx<T>* Wx; cx<T>* Wy;   // delta-f part
  T* w1p; T* w2p;           // (P*B, Nx, Ny)
  int pf;                   // > 0: touch prefetch distance in blocks (TouchTiles)
};

template <typename T, int R, int NT, int LGM>
__device__ __forceinline__ void delta_y_body(const DeltaYArgs<T>& d, unsigned char* smem, size_t sl) {
  using G = ColTile<R, NT, LGM>;
  constexpr int M = G::M, LGN = G::LGN, LD = G::LDN, Nyh = G::Nyh, C = G::C, LGC = G::LGC;
  const FlowYArgs<T>& a = d.f;
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;                            // half of the twiddle circle (see row_tw)
  const int Nx = a.Nx, x0 = xcd_tile(blockIdx.x, gridDim.x) * C;
  const int bphi = a.ph.Bphi == 1 ? 0 : (int)(sl / a.P);
  const T invNy = T(1) / T(2 * M);
  constexpr int NyhP = mixed_rows(Nyh);
  const size_t moff = sl * (size_t)NyhP * Nx;
  const size_t pbase = ((size_t)bphi * Nx + x0) * M, mbase = (sl * Nx + x0) * (size_t)M;
  CMBL_WSTAMP(14);
  CMBL_STAMP(0);
  // Loads are requested in the order they are needed and as late as their consumer allows without exposing latency: every
  // workgroup of the launch starts at the same moment, so everything requested at t = 0 shares the memory system with the
  // same request of 511 other workgroups -- the pair tile, which alone gates the first transform, then arrives with the LAST
  // bytes of the burst (in-kernel stamps: first commit at 13.9k cycles of a 39k-cycle workgroup).  So: pair tile first; p(t) and
  // the delta-f tile fly during the first transform; the RK state during the second.
  TwStage<T, NT, M> twr;
  PairStage<T, NT, LGN, LGC> ps;
  HalfStage<T, NT, LGM, LGC> th;
  twr.issue(a.twY);
  ps.issue(a.Gx + moff, a.A + moff, a.ly, Nx, x0);
  twr.commit(tw);
  ps.template commit<LD>(s);
  using PM = PairMap<R, NT, LGM>;
  cx<T> px[R], py[R];
#pragma unroll
  for (int i = 0; i < R; ++i) load_p_only(a.ph, pbase, (unsigned)PM::e(i), a.rk.t, px[i], py[i]);
  th.issue(d.H + moff, a.twY, x0);
  __syncthreads();
  CMBL_STAMP(1);
  // (d/dx f, d/dy f) from one N-point inverse transform
  cx<T> dx[R], dy[R];
  npt_inverse_read<T, R, NT, LGM, LD>(s, tw, invNy, dx, dy);
  CMBL_STAMP(2);
  __syncthreads();
  CMBL_STAMP(3);
  // L(delta f) = irfft2(delta f)
  th.template commit<LD>(s);
  cx<T>* y0p = reinterpret_cast<cx<T>*>(a.y0) + mbase;
  cx<T>* accp = reinterpret_cast<cx<T>*>(a.acc) + mbase;
  cx<T> fn[R], ldf[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {                               // RK state: requested now, used after the second transform
    const unsigned e = PM::e(i);
    fn[i] = at32(reinterpret_cast<const cx<T>*>(a.y0r) + mbase, e);
    ldf[i] = a.rk.stage == 1 ? mk<T>(0, 0) : at32(accp, e);
  }
  __syncthreads();
  CMBL_STAMP(4);
  cx<T> lz[R];
  mpt_inverse_read<T, R, NT, LGM, LD>(s, tw, invNy, lz);
  CMBL_STAMP(5);
#pragma unroll
  for (int i = 0; i < R; ++i) {
    const unsigned e = PM::e(i);
    cx<T> y0 = fn[i], acc = ldf[i];
    ldf[i] = lz[i];
    nt_store(&at32(reinterpret_cast<cx<T>*>(d.w1p) + mbase, e), pmul(ldf[i], dx[i]));
    nt_store(&at32(reinterpret_cast<cx<T>*>(d.w2p) + mbase, e), pmul(ldf[i], dy[i]));
    const cx<T> kv = pmul(px[i], dx[i]) + pmul(py[i], dy[i]);
    fn[i] = rk_update(a.rk, kv, y0, acc);
    if (a.rk.stage == 4) at32(y0p, e) = y0; else at32(accp, e) = acc;
  }
  using Touch = TouchTiles<T, NT, LGM, C, 3>;
  size_t mo2 = 0; int x02 = 0;
  bool do_touch = false;
  if constexpr (col_touch<T>(LGM)) {
    do_touch = touch_target<C>(d.pf, sl, NyhP, Nx, mo2, x02);
    if (do_touch) { const cx<T>* const heads[3] = {a.Gx + mo2, a.A + mo2, d.H + mo2}; Touch::issue(heads, x02, touch_pad<T, C, LD>(M)); }
  }
  __syncthreads();
  CMBL_STAMP(6);
  // (Wx, Wy') from one N-point forward transform of px*Ldf + i*py*Ldf
  CMBL_STAMP(7);
  npt_write_forward<T, R, NT, LGM, LD>(s, tw, [&](int i, cx<T>& x, cx<T>& y) { x = pmul(px[i], ldf[i]); y = pmul(py[i], ldf[i]); });
  CMBL_STAMP(8);
  if constexpr (col_touch<T>(LGM)) { if (do_touch && a.rk.last) Touch::drain(); }       // the last launch of a flow ends after the stores below
  {
    cx<T>* Wx = tile_base(d.Wx + moff, x0, NyhP); cx<T>* Wy = tile_base(d.Wy + moff, x0, NyhP);
    pair_split<T, NT, LD, LGN, LGC, G::RZ>(s, [&](int i, int k, int c, cx<T> A, cx<T> B) {
      const unsigned gi = tile_off<T, C>(k, c, x0, NyhP);
      handoff_store<T, wt_cols<T>(C, M)>(Wx, gi, A); handoff_store<T, wt_cols<T>(C, M)>(Wy, gi, mul_il(B, ps.l[i]));    // ly[k] is still in registers from the pair load (same entry mapping)
    });
  }
  CMBL_STAMP(9);
  if (a.rk.last) { CMBL_WSTAMP(15); return; }
  __syncthreads();
  // next-stage f : rfft_y
  mpt_write_forward<T, R, NT, LGM, LD>(s, tw, [&](int i) { return fn[i]; });
  CMBL_STAMP(11);
  if constexpr (col_touch<T>(LGM)) { if (do_touch) Touch::drain(); }
  half_store<T, NT, LD, LGM, LGC>(s, a.Anext + moff, tw, x0);
  CMBL_STAMP(12);
  CMBL_WSTAMP(15);
}

#ifdef CMBL_EXPERIMENT_COL_PIPELINE
// EXPERIMENT, measured and rejected in round 5 (-DCMBL_EXPERIMENT_COL_PIPELINE; profiles/r05_ab_col_pipeline_rejected.txt).
// The same stage with the workgroup walking TPW tiles and the next tile's pair tile requested under the current tile's last phase.  For
// shapes whose register footprint leaves ONE workgroup per CU -- 2048 rows in double precision: 244 registers x 512 threads is the CU's
// whole register file -- nothing overlaps a workgroup's exposed waits: in-kernel stamps at 2048^2 fp64
// (profiles/r05_stamps_delta_cols_2048_f64.txt) show 18.9k of a workgroup's 45.8k cycles spent waiting for its pair tile.  What killed it is
// the compiler, not the idea: as soon as one thread walks two tiles -- as a loop or unrolled, WITH or WITHOUT the prefetch -- hipcc keeps
// the tile-independent address values of all load / store sites live across the tiles (loop-invariant code motion / common
// subexpressions) and spills 228-596 bytes per lane, which the launch then moves through the vector memory path: (grad L)' 14.4 -> 17.0 ms.
template <typename T, int R, int NT, int LGM, int TPW>
__device__ __forceinline__ void delta_y_body_pipelined(const DeltaYArgs<T>& d, unsigned char* smem, size_t sl) {
  constexpr int tpw = TPW;
  using G = ColTile<R, NT, LGM>;
  constexpr int M = G::M, LGN = G::LGN, LD = G::LDN, Nyh = G::Nyh, C = G::C, LGC = G::LGC;
  const FlowYArgs<T>& a = d.f;
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + M;
  const int Nx = a.Nx, ntiles = (int)gridDim.x * tpw;
  const int bphi = a.ph.Bphi == 1 ? 0 : (int)(sl / a.P);
  const T invNy = T(1) / T(2 * M);
  constexpr int NyhP = mixed_rows(Nyh);
  const size_t moff = sl * (size_t)NyhP * Nx;
  using PM = PairMap<R, NT, LGM>;
  // tile k of this workgroup: the virtual block blockIdx.x + k * gridDim.x of a one-tile-per-block launch (same XCD for every k when
  // gridDim.x is a multiple of 8; neighbouring workgroups of an XCD walk neighbouring tiles in step and share their 64-byte lines)
  auto tile_x0 = [&](int k) { return xcd_tile((int)blockIdx.x + k * (int)gridDim.x, ntiles) * C; };
  TwStage<T, NT, M> twr;
  PairStage<T, NT, LGN, LGC> ps;
  HalfStage<T, NT, LGM, LGC> th;
  cx<T> px[R], py[R];
  // CMBL_PL_HEAD = 1: the pair tile alone is prefetched (it alone gates the first transform); 2: p(t) and the delta-f tile as well (118
  // registers carried across the loop: 596 bytes of scratch per lane in double precision)
#ifndef CMBL_PL_HEAD
#define CMBL_PL_HEAD 1
#endif
  // tid: the thread index as a value the compiler cannot see through, a fresh one per tile -- every per-thread address below is then recomputed
  // per tile instead of being hoisted above the tile loop and kept live across it (round 5's version spilled 228-596 bytes per lane that way;
  // the trick is kernels_small.hpp's)
  auto issue_rest = [&](int x0, int tid) {
    const size_t pbase = ((size_t)bphi * Nx + x0) * M;
#pragma unroll
    for (int i = 0; i < R; ++i) load_p_only(a.ph, pbase, (unsigned)PM::e(i, tid), a.rk.t, px[i], py[i]);
    th.issue(d.H + moff, a.twY, x0, tid);
  };
  auto issue_head = [&](int x0, int tid) {
    ps.issue_xy(a.Gx + moff, a.A + moff, x0, tid);
    if (CMBL_PL_HEAD >= 2) issue_rest(x0, tid);
  };
  auto fresh_tid = [] { int t = (int)threadIdx.x; asm volatile("" : "+v"(t)); return t; };
  twr.issue(a.twY);
  ps.issue(a.Gx + moff, a.A + moff, a.ly, Nx, tile_x0(0), fresh_tid());
  if (CMBL_PL_HEAD >= 2) issue_rest(tile_x0(0), fresh_tid());
  twr.commit(tw);
#pragma unroll
  for (int k = 0; k < tpw; ++k) {
    const int x0 = tile_x0(k);
    const int tid = fresh_tid();
    const size_t mbase = (sl * Nx + x0) * (size_t)M;
    if (CMBL_PL_HEAD == 0 && k > 0) ps.issue_xy(a.Gx + moff, a.A + moff, x0, tid);          // timing aid: several tiles, nothing prefetched
    ps.template commit<LD>(s, tid);
    if (CMBL_PL_HEAD < 2) issue_rest(x0, tid);
    __syncthreads();
    cx<T> dx[R], dy[R];
    npt_inverse_read<T, R, NT, LGM, LD>(s, tw, invNy, dx, dy, tid);
    __syncthreads();
    th.template commit<LD>(s, tid);
    cx<T>* y0p = reinterpret_cast<cx<T>*>(a.y0) + mbase;
    cx<T>* accp = reinterpret_cast<cx<T>*>(a.acc) + mbase;
    cx<T> fn[R], ldf[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const unsigned e = PM::e(i, tid);
      fn[i] = at32(reinterpret_cast<const cx<T>*>(a.y0r) + mbase, e);
      ldf[i] = a.rk.stage == 1 ? mk<T>(0, 0) : at32(accp, e);
    }
    __syncthreads();
    cx<T> lz[R];
    mpt_inverse_read<T, R, NT, LGM, LD>(s, tw, invNy, lz, tid);
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const unsigned e = PM::e(i, tid);
      cx<T> y0 = fn[i], acc = ldf[i];
      ldf[i] = lz[i];
      nt_store(&at32(reinterpret_cast<cx<T>*>(d.w1p) + mbase, e), pmul(ldf[i], dx[i]));
      nt_store(&at32(reinterpret_cast<cx<T>*>(d.w2p) + mbase, e), pmul(ldf[i], dy[i]));
      const cx<T> kv = pmul(px[i], dx[i]) + pmul(py[i], dy[i]);
      fn[i] = rk_update(a.rk, kv, y0, acc);
      if (a.rk.stage == 4) at32(y0p, e) = y0; else at32(accp, e) = acc;
    }
    __syncthreads();
    npt_write_forward<T, R, NT, LGM, LD>(s, tw, [&](int i, cx<T>& x, cx<T>& y) { x = pmul(px[i], ldf[i]); y = pmul(py[i], ldf[i]); }, tid);
    {
      cx<T>* Wx = tile_base(d.Wx + moff, x0, NyhP); cx<T>* Wy = tile_base(d.Wy + moff, x0, NyhP);
      pair_split<T, NT, LD, LGN, LGC, G::RZ>(s, [&](int i, int kk, int c, cx<T> A, cx<T> B) {
        const unsigned gi = tile_off<T, C>(kk, c, x0, NyhP);
        handoff_store<T, wt_cols<T>(C, M)>(Wx, gi, A); handoff_store<T, wt_cols<T>(C, M)>(Wy, gi, mul_il(B, ps.l[i]));
      }, tid);
    }
    // the next tile's head: everything it waits for first, requested while this tile's last transform runs
    if (CMBL_PL_HEAD > 0 && k + 1 < tpw) issue_head(tile_x0(k + 1), fresh_tid());
    if (!a.rk.last) {
      __syncthreads();
      mpt_write_forward<T, R, NT, LGM, LD>(s, tw, [&](int i) { return fn[i]; }, tid);
      half_store<T, NT, LD, LGM, LGC>(s, a.Anext + moff, tw, x0, tid);
    }
    __syncthreads();                                          // the tile is rewritten by the next commit
  }
}

#endif
// ---------------------------------------------------------------------------------------------
// delta-phi.  Its velocity (src/lenseflow.jl:198-206),
//     d(dphi)/dt = i lx F(u1) + i ly F(u2) - lx^2 F(a) - lx ly F(b) - ly^2 F(c),   F = rfft2,
//     w = sum_pol L(df) * grad f   (spin-adjoint product, src/proj_lambert.jl:423-430),   u = M^-1(t) w   (Q1 switch),
//     a = t px u1,  b = t (py u1 + px u2),  c = t py u2,
// depends on (f, df, t) only -- never on dphi itself -- so the RK4 update of dphi is a pure quadrature: dphi(end) = sum over all
// stages of c_s * velocity_s with c_s = h/6 * (1, 2, 2, 1).  Transforms and l-multipliers are linear and the same for every stage,
// hence  dphi(end) = i lx F(sum c_s u1_s) + i ly F(sum c_s u2_s) - lx^2 F(sum c_s a_s) - ...:  the column kernel only stores the
// partial products w of each stage (two maps per slice, which it wrote before as well), ONE pointwise pass reduces them over
// stages and pols into five maps, and five real transforms per delta flow replace five per STAGE (4n times fewer).
// Identical to the reference's stage-by-stage update up to the order of floating-point summation.
// 16-byte non-temporal load (streaming data read exactly once: keeps the caches for the arrays the flow re-reads)
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef double nt_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_load16(const float* p, float (&o)[4]) { const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p)); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ void nt_load16(const double* p, double (&o)[2]) { const nt_d2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_d2*>(p)); o[0] = v.x; o[1] = v.y; }

// k_dphi_reduce: stages whose loads are in flight together per thread (SU) and the register budget (waves per SIMD).  With SU = 4 the
// kernel needed 165 registers = 3 waves per SIMD = 768 resident blocks for a grid of 1024 at 1024^2: a third of the blocks ran in a
// second round (121 us).  SU = 2 fits 128 registers, the whole grid is resident at once: 89 us (512 MB -> 5.8 TB/s).
#ifndef CMBL_DPHI_WAVES
#define CMBL_DPHI_WAVES 4
#endif
#ifndef CMBL_DPHI_SU
#define CMBL_DPHI_SU 2
#endif
// (t_s, c_s) of up to 64 stages ride in the kernel arguments (no upload per flow); longer flows pass a device table
template <typename T> struct TcTab { static constexpr int MAXST = 64; T v[2 * MAXST]; };
// V = pixels per thread: 16-byte loads when npix allows it, 1 for the any-size path (odd pixel counts)
template <typename T, int V = 16 / (int)sizeof(T)>
__global__ __launch_bounds__(NTP, CMBL_DPHI_WAVES) void k_dphi_reduce(PhiMaps<T> ph, const T* __restrict__ W /*[nst][2][slices][npix]*/,
                                                    TcTab<T> tcv, const T* __restrict__ tcd /*[nst][2] = (t_s, c_s), or nullptr: tcv*/, T* __restrict__ out /*[5][B][npix]*/,
                                                    long npix, int P, int B, int nst, int alias_quirk) {
  struct alignas(V * sizeof(T)) Vec { T v[V]; };
  const int b = blockIdx.y;
  const size_t slices = (size_t)P * B, pb = (size_t)(ph.Bphi == 1 ? 0 : b) * npix;
  const long nv = npix / V;
  for (long iv = (long)blockIdx.x * NTP + threadIdx.x; iv < nv; iv += (long)gridDim.x * NTP) {
    const long i = iv * V;
    const Vec gx = *reinterpret_cast<const Vec*>(ph.gx + pb + i), gy = *reinterpret_cast<const Vec*>(ph.gy + pb + i);
    const Vec hxx = *reinterpret_cast<const Vec*>(ph.hxx + pb + i), hyx = *reinterpret_cast<const Vec*>(ph.hyx + pb + i);
    const Vec hyy = *reinterpret_cast<const Vec*>(ph.hyy + pb + i);
    double U1[V] = {}, U2[V] = {}, A[V] = {}, Bb[V] = {}, Cc[V] = {};
    // the per-stage products are read exactly once: non-temporal loads, SU stages' worth requested before the first is used
    constexpr int SU = CMBL_DPHI_SU;
    for (int s0 = 0; s0 < nst; s0 += SU) {
      Vec w1[SU], w2[SU];
#pragma unroll
      for (int j = 0; j < SU; ++j) {
        w1[j] = Vec{}; w2[j] = Vec{};
        if (s0 + j < nst)
          for (int p = 0; p < P; ++p) {
            Vec a1, a2;
            const T* q1 = W + ((size_t)(2 * (s0 + j)) * slices + (size_t)b * P + p) * npix + i;
            const T* q2 = W + ((size_t)(2 * (s0 + j) + 1) * slices + (size_t)b * P + p) * npix + i;
            if constexpr (V * sizeof(T) == 16) { nt_load16(q1, a1.v); nt_load16(q2, a2.v); }
            else { a1 = *reinterpret_cast<const Vec*>(q1); a2 = *reinterpret_cast<const Vec*>(q2); }
#pragma unroll
            for (int k = 0; k < V; ++k) { w1[j].v[k] += a1.v[k]; w2[j].v[k] += a2.v[k]; }
          }
      }
#pragma unroll
      for (int j = 0; j < SU; ++j) {
        if (s0 + j >= nst) break;
        const T t = tcd ? tcd[2 * (s0 + j)] : tcv.v[2 * (s0 + j)], c = tcd ? tcd[2 * (s0 + j) + 1] : tcv.v[2 * (s0 + j) + 1];
#pragma unroll
        for (int k = 0; k < V; ++k) {
          T px, py, m11, m12, m22;
          flow_pm(t, gx.v[k], gy.v[k], hxx.v[k], hyx.v[k], hyy.v[k], px, py, m11, m12, m22);
          // u = M^-1 w   (src/field_vectors.jl:48-49; with the reference's aliasing, v[2] sees the updated v[1])
          const T v1 = m11 * w1[j].v[k] + m12 * w2[j].v[k];
          const T u2 = m12 * (alias_quirk ? v1 : w1[j].v[k]) + m22 * w2[j].v[k];
          U1[k] += (double)(c * v1); U2[k] += (double)(c * u2);
          A[k] += (double)(c * t * px * v1); Bb[k] += (double)(c * t * (py * v1 + px * u2)); Cc[k] += (double)(c * t * py * u2);
        }
      }
    }
    const size_t o = (size_t)b * npix + i, cs = (size_t)B * npix;
    Vec r;
#pragma unroll
    for (int k = 0; k < V; ++k) r.v[k] = (T)U1[k];
    *reinterpret_cast<Vec*>(out + o) = r;
#pragma unroll
    for (int k = 0; k < V; ++k) r.v[k] = (T)U2[k];
    *reinterpret_cast<Vec*>(out + cs + o) = r;
#pragma unroll
    for (int k = 0; k < V; ++k) r.v[k] = (T)A[k];
    *reinterpret_cast<Vec*>(out + 2 * cs + o) = r;
#pragma unroll
    for (int k = 0; k < V; ++k) r.v[k] = (T)Bb[k];
    *reinterpret_cast<Vec*>(out + 3 * cs + o) = r;
#pragma unroll
    for (int k = 0; k < V; ++k) r.v[k] = (T)Cc[k];
    *reinterpret_cast<Vec*>(out + 4 * cs + o) = r;
  }
}
// dphi = i lx F1 + i ly F2 - lx^2 FA - lx ly FB - ly^2 FC   (F layout, [5][B][plane] in, [B][plane] out)
template <typename T>
__global__ __launch_bounds__(NTP) void k_dphi_combine(const cx<T>* __restrict__ F5, cx<T>* __restrict__ out, const T* __restrict__ lx_r,
                                                     const T* __restrict__ ly, int Nx, long plane, int B) {
  const long i = (long)blockIdx.x * NTP + threadIdx.x;
  if (i >= plane) return;
  const T lx = lx_r[(unsigned)i % (unsigned)Nx], l_y = ly[(unsigned)i / (unsigned)Nx];
  const long cs = (long)B * plane;
  for (int b = 0; b < B; ++b) {
    const cx<T>* f = F5 + (long)b * plane + i;
    const cx<T> f1 = f[0], f2 = f[cs], fa = f[2 * cs], fb = f[3 * cs], fc = f[4 * cs];
    const T re = -lx * f1.y - l_y * f2.y - lx * lx * fa.x - lx * l_y * fb.x - l_y * l_y * fc.x;
    const T im = lx * f1.x + l_y * f2.x - lx * lx * fa.y - lx * l_y * fb.y - l_y * l_y * fc.y;
    out[(long)b * plane + i] = mk<T>(re, im);
  }
}

// ---------------------------------------------------------------------------------------------
// The delta flow is two launches per RK stage: the column kernel (delta_y_body per slice) and ONE row launch for the delta-f
// row pass of this stage (adj_x_body) and the d/dx pass of the NEXT stage's f (grad_x_body), which are independent.
template <typename T, int R, int NT, int LGM>
__global__ __launch_bounds__(NT, col_min_waves<T>(R, NT)) void k_delta_cols(DeltaYArgs<T> d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  delta_y_body<T, R, NT, LGM>(d, smem, blockIdx.y);
}
#ifdef CMBL_EXPERIMENT_COL_PIPELINE
template <typename T, int R, int NT, int LGM, int TPW>
__global__ __launch_bounds__(NT, col_min_waves<T>(R, NT)) void k_delta_cols_pl(DeltaYArgs<T> d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  delta_y_body_pipelined<T, R, NT, LGM, TPW>(d, smem, blockIdx.y);
}
#endif
template <typename T, int LGNX, int RPW>
__global__ __launch_bounds__(row_nt(RPW), row_min_waves<T>()) void k_delta_rows(AdjXArgs<T> a, GradXArgs<T> g, int nblk_adj) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
#ifdef CMBL_STAMPS_ROWS            // launch timeline of the last launch that has both parts (tools/gpu_stamps_rows.py)
  if (!a.rk.last && threadIdx.x == 0) g_stamps[(size_t)b * 16 + 14] = wall_clock64();
#endif
  // grid order: full row groups of the adjoint part, full groups of the d/dx part, then the short groups of both (see row_group)
  const int G = (a.Nyh + RPW - 1) / RPW, Gf = a.Nyh / RPW, nfull = (nblk_adj / G) * Gf, nshort = nblk_adj - nfull;
  const bool both = (int)gridDim.x > nblk_adj;
  bool adj = true;
  int blk = b;                                                           // row group within its part
  if (both && b >= nfull) {
    if (b < 2 * nfull) { adj = false; blk = b - nfull; }
    else if (b < 2 * nfull + nshort) blk = b - nfull;
    else { adj = false; blk = b - nfull - nshort; }
  }
  if (adj) adj_x_body<T, LGNX, RPW>(a, smem, (unsigned)blk, (unsigned)nblk_adj);
  else grad_x_body<T, LGNX, RPW>(g, smem, (unsigned)blk, (unsigned)nblk_adj);
#ifdef CMBL_STAMPS_ROWS
  if (!a.rk.last && threadIdx.x == 0) g_stamps[(size_t)b * 16 + 15] = wall_clock64();
#endif
}

// gradient / hessian multipliers for precompute (src/specialops.jl:184-188): F layout in, five F-layout outputs
//   out[0]=i lx phi, out[1]=i ly phi, out[2]=-lx^2 phi, out[3]=(i lx)(i ly) phi, out[4]=-ly^2 phi
template <typename T>
__global__ __launch_bounds__(NTP) void k_gradhess_mult(const cx<T>* __restrict__ phi, cx<T>* __restrict__ out,
                                                      const T* __restrict__ lx_r, const T* __restrict__ ly,
                                                      int Nx, int Nyh, int B) {
  const long plane = (long)Nyh * Nx;
  const long i = (long)blockIdx.x * NTP + threadIdx.x;
  if (i >= plane) return;
  const T lx = lx_r[(unsigned)i % (unsigned)Nx], l_y = ly[(unsigned)i / (unsigned)Nx];
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
