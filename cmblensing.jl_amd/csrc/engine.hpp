// Host-side engine: context (geometry, twiddles, launch geometry), layout/basis conversions, the LenseFlow
// operator and the data-model / Wiener-filter / posterior drivers.  Everything below the C ABI (api.hip).
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <map>
#include <memory>
#include <mutex>
#include <type_traits>
#include <cstdlib>
#include <hip/hip_ext.h>
#include "kernels_fft.hpp"
#include "kernels_pointwise.hpp"
#include "kernels_flow.hpp"
#include "kernels_harm.hpp"
#include "kernels_generic.hpp"
#include "kernels_ct.hpp"

namespace cmbl {

enum Basis { B_MAP = 0, B_FOURIER = 1, B_HARMONIC = 2 };
enum FlowMode { F_FWD = 0, F_INV = 1, F_ADJ = 2, F_INVADJ = 3 };

// hipFuncSetAttribute is per function, process-wide: the table is shared by all contexts, hence the lock (independent contexts
// may be driven from different host threads)
inline void raise_lds_limit(const void* fn, size_t bytes) {
  static std::map<const void*, size_t> done;
  static std::mutex mtx;
  if (bytes <= 48 * 1024) return;
  std::lock_guard<std::mutex> lock(mtx);
  auto it = done.find(fn);
  if (it != done.end() && it->second >= bytes) return;
  CMBL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  done[fn] = bytes;
}
// kernel classes for the optional per-launch event timing (cmbl_prof_*)
enum KernelId { K_LAYOUT = 0, K_Y_R2C, K_Y_C2R, K_X_FFT, K_X_GRAD, K_FLOW_Y, K_ADJ_Y, K_ADJ_X, K_DELTA_Y, K_DELTA_ROWS, K_DPHI_Y, K_DPHI_X,
                K_GRADHESS, K_HARM, K_LINCOMB, K_MASK, K_REDUCE, K_GEN_DFT, K_GEN_POINT, K_CG, K_XPW, K_PWFLAT, K_COUNT };
static const char* const kKernelNames[K_COUNT] = {"layout", "y_r2c", "y_c2r", "x_fft", "x_grad", "flow_y_fwd", "adj_y", "adj_x", "delta_cols", "delta_rows",
                                                  "dphi_reduce", "dphi_combine", "gradhess_mult", "harm_apply", "lincomb", "mask_mul", "reduce",
                                                  "generic_dft", "generic_pointwise", "cg_update", "x_harm", "harm_dot"};

// With profiling on, the launch goes through hipExtLaunchKernelGGL, whose start / stop events carry the kernel's OWN begin and end
// timestamps (what rocprofv3's kernel trace reports); an event pair recorded around a plain launch also brackets its dispatch (+2 us).
#define CMBL_LAUNCH_NT(ctxp, kid, nthreads, kernel, grid, lds, stream, ...)            \
  do {                                                                                \
    raise_lds_limit(reinterpret_cast<const void*>(kernel), (lds));                    \
    if ((ctxp)->prof_on) {                                                            \
      auto& ev_ = (ctxp)->prof_next(kid);                                             \
      hipExtLaunchKernelGGL(kernel, grid, dim3(nthreads), (lds), (stream), ev_.first, ev_.second, 0, __VA_ARGS__); \
      CMBL_HIP(hipGetLastError());                                                    \
      (ctxp)->prof_commit(kid);                                                       \
    } else { hipLaunchKernelGGL(kernel, grid, dim3(nthreads), (lds), (stream), __VA_ARGS__); CMBL_HIP(hipGetLastError()); }   \
  } while (0)
#define CMBL_LAUNCH(ctxp, kid, kernel, grid, lds, stream, ...) CMBL_LAUNCH_NT(ctxp, kid, NTP, kernel, grid, lds, stream, __VA_ARGS__)

// compiled column-tile shapes (lgM, R, NT):  C = R*NT >> lgM columns per workgroup.  Per lgM the list holds what tileY() can select in
// either precision (narrowest C >= 4 that fits LDS, 512 threads preferred; C = 2 or 1 where four double-precision columns do not fit)
// plus one wider / narrower neighbour for CMBL_TUNE_C sweeps; 1024-thread and R = 16 variants of lgM 9 / 10 spilled registers and were
// never selected.
#ifndef CMBL_COL_LIST
#define CMBL_COL_LIST(X) X(4, 1, 256) X(5, 1, 256) X(5, 2, 256) X(6, 1, 256) X(6, 2, 256) X(7, 2, 128) X(7, 2, 256) X(7, 4, 256) X(8, 2, 256) X(8, 4, 256) X(8, 8, 256) \
                         X(9, 4, 256) X(9, 8, 256) X(10, 8, 256) X(11, 4, 1024) X(11, 2, 1024) \
                         X(8, 2, 512) X(8, 4, 512) X(9, 4, 512) X(9, 8, 512) X(10, 4, 512) X(10, 8, 512)
#endif
#ifndef CMBL_ROW_LIST
#define CMBL_ROW_LIST(X) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12)
#endif
// convolution lengths 2^lg of the any-size transforms (kernels_generic.hpp)
#ifndef CMBL_GEN_LIST
#define CMBL_GEN_LIST(X) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13)
#endif

inline int env_int(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }

struct CtxBase {
  int Ny = 0, Nx = 0, Nyh = 0, M = 0, lgM = 0, lgNx = 0, dtype = 0, device = 0, num_cus = 256;
  bool generic = false;                   // any-size path (kernels_generic.hpp): Ny or Nx not a power of two (or < 32)
  // SUM_FLOAT64 by default (see kernels_pointwise.hpp); CMBL_REFERENCE_EXACT=1 starts a context with the reference's own default,
  // plain sums in the working precision (SUM_WORKING, src/util.jl:288-316)
  int sum_mode = env_int("CMBL_REFERENCE_EXACT", 0) ? 0 : 1;
  // Behaviour switches (A/B and profiling aids).  The environment is read ONCE, when the context is created; afterwards they only
  // change through cmbl_ctx_set_option -- never by a getenv in a launch path (not safe against a concurrent setenv of the host
  // language, and a switch that flips between two launches of one flow would mix code paths).
  struct Opts {
    int slice_streams = env_int("CMBL_SLICE_STREAMS", 4);                 // launch chains per flow (1 = one launch over all slices)
    int slice_streams_min_pix = env_int("CMBL_SLICE_STREAMS_MIN_PIX", 1 << 19);   //   ... whose chains carry at least this many pixels each (Flow::groups)
    int pcache = env_int("CMBL_NO_PCACHE", 0) == 0;                       // p(t_k) cache per phi (read when phi is set)
    int pcache_max_mb = env_int("CMBL_PCACHE_MAX_MB", 16384);
    int fused_harm = env_int("CMBL_NO_FUSED_HARM", 0) == 0;               // harmonic-space work on the row carrier (kernels_harm.hpp)
    int gen_separable = env_int("CMBL_GEN_SEPARABLE", 1) != 0;            // any-size path: separable stages
    int gen_prologue = env_int("CMBL_GEN_PROLOGUE", 1) != 0;              //   pointwise work in the fetch of the consuming transform
    int gen_xderiv_fused = env_int("CMBL_GEN_XDERIV_FUSED", 1) != 0;      //   d/dx pass as one launch
    int gen_slice_streams = env_int("CMBL_GEN_SLICE_STREAMS", 1) != 0;    //   one launch chain per group of slices (Flow::gen_groups)
    int gen_streams_min_pix = env_int("CMBL_GEN_STREAMS_MIN_PIX", 300000);   //   ... whose chains carry at least this many 4-byte pixels each (Flow::gen_groups)
    int gen_yy = env_int("CMBL_GEN_YY", 1) != 0;                          //   the y passes of a forward stage in one launch (GenDft::yy; needs gen_ct)
    int gen_xmerge = env_int("CMBL_GEN_XMERGE", 1) != 0;                  //   the row update of an adjoint-type stage also opens the next stage (gen_x_adj_next): one launch less per stage
    int gen_tiled = env_int("CMBL_GEN_TILED", 7);                     //   the half planes the fused any-size stages hand between their column and row launches are TILED ([x / 4][ky][x % 4], GenDft::in_tiled) instead of [ky][x]
    int gen_ct_cols = env_int("CMBL_GEN_CT_COLS", 1);                     //   half-width column groups in the fused y passes: 0 never, 1 for small launches (Ctx::ct_cols_per_group), 2 always
    int gen_ct_rows = env_int("CMBL_GEN_CT_ROWS", 1) != 0;                //   shorter row groups in x-pass launches with fewer groups than CUs (Ctx::ct_rows_per_group)
    int gen_ct = env_int("CMBL_GEN_CT", 1) != 0;                          //   compile-time plans for the lengths of CMBL_CT_LIST (kernels_ct.hpp)
    // launch geometry that fills the chip on small maps (profiles/r05_ab_occupancy_tiles.txt): narrower column tiles while a launch has
    // fewer workgroups than `fill_target` (0: the rule in Ctx::tileY), shorter row groups while it has fewer than `row_fill_target`
    // (0: half the number of CUs -- 130 -> 258 row workgroups at 512^2 QU measured slower, 34 -> 130 at 128^2 17 % faster)
    int col_prefetch = env_int("CMBL_COL_PREFETCH", -1);                  // touch prefetch of multi-round column launches: -1 = Ctx::col_prefetch's rule, 0 = off, > 0 = that distance (blocks)
    int col_pipeline = env_int("CMBL_COL_PIPELINE", 1);                   // only in -DCMBL_EXPERIMENT_COL_PIPELINE builds: two tiles per column workgroup
    int occupancy_tiles = env_int("CMBL_OCCUPANCY_TILES", 3);             // bit 0: narrower column tiles, bit 1: shorter row groups
    int fill_target = env_int("CMBL_FILL_TARGET", 0);
    int row_fill_target = env_int("CMBL_ROW_FILL_TARGET", 0);
    int small_flow = env_int("CMBL_SMALL_FLOW", 1);                       // small maps: a whole flow as ONE launch, one workgroup per slice (kernels_small.hpp): 0 off, 1 up to 64 x 64, 2 wherever compiled (128 x 128)
  } opts;
  int* opt_ptr(const std::string& k) {
    if (k == "slice_streams") return &opts.slice_streams;
    if (k == "slice_streams_min_pix") return &opts.slice_streams_min_pix;
    if (k == "pcache") return &opts.pcache;
    if (k == "pcache_max_mb") return &opts.pcache_max_mb;
    if (k == "fused_harm") return &opts.fused_harm;
    if (k == "gen_separable") return &opts.gen_separable;
    if (k == "gen_prologue") return &opts.gen_prologue;
    if (k == "gen_xderiv_fused") return &opts.gen_xderiv_fused;
    if (k == "gen_ct") return &opts.gen_ct;
    if (k == "gen_ct_rows") return &opts.gen_ct_rows;
    if (k == "gen_ct_cols") return &opts.gen_ct_cols;
    if (k == "gen_xmerge") return &opts.gen_xmerge;
    if (k == "gen_tiled") return &opts.gen_tiled;
    if (k == "gen_yy") return &opts.gen_yy;
    if (k == "gen_slice_streams") return &opts.gen_slice_streams;
    if (k == "gen_streams_min_pix") return &opts.gen_streams_min_pix;
    if (k == "col_pipeline") return &opts.col_pipeline;
    if (k == "col_prefetch") return &opts.col_prefetch;
    if (k == "occupancy_tiles") return &opts.occupancy_tiles;
    if (k == "fill_target") return &opts.fill_target;
    if (k == "row_fill_target") return &opts.row_fill_target;
    if (k == "small_flow") return &opts.small_flow;
    return nullptr;
  }
  double theta = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::vector<double> h_lx, h_ly, h_lam, h_sin2, h_cos2, h_lmag;   // reference layout ([x][ky] planes)
  // optional per-launch timing with HIP events on the context's stream (bench.py roofline leg)
  bool prof_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev[K_COUNT];
  size_t prof_used[K_COUNT] = {};
  double prof_ms[K_COUNT] = {};
  long prof_n[K_COUNT] = {};
  // the event pair of the next profiled launch of class k; it is COUNTED (prof_commit) only once the launch has been accepted, so a
  // failed launch leaves no unrecorded pair for prof_collect to trip over (which would mask the launch error)
  std::pair<hipEvent_t, hipEvent_t>& prof_next(int k) {
    if (prof_used[k] == prof_ev[k].size()) {
      hipEvent_t a, b; CMBL_HIP(hipEventCreate(&a)); CMBL_HIP(hipEventCreate(&b)); prof_ev[k].push_back({a, b});
    }
    return prof_ev[k][prof_used[k]];
  }
  void prof_commit(int k) { ++prof_used[k]; }
  void prof_collect() {                         // synchronises; folds recorded pairs into the totals
    CMBL_HIP(hipDeviceSynchronize());
    for (int k = 0; k < K_COUNT; ++k) {
      for (size_t i = 0; i < prof_used[k]; ++i) {
        float ms = 0; CMBL_HIP(hipEventElapsedTime(&ms, prof_ev[k][i].first, prof_ev[k][i].second));
        prof_ms[k] += ms; ++prof_n[k];
      }
      prof_used[k] = 0;
    }
  }
  void prof_reset() { for (int k = 0; k < K_COUNT; ++k) { prof_used[k] = 0; prof_ms[k] = 0; prof_n[k] = 0; } }
  virtual ~CtxBase() {
    for (auto& v : prof_ev) for (auto& e : v) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (own_stream && stream) (void)hipStreamDestroy(stream);
  }
  long plane() const { return (long)Nyh * Nx; }
  long mplane() const { return (long)((Nyh + 3) & ~3) * Nx; }   // one slice of a mixed-layout array (kernels_fft.hpp: mixed_rows)
  long npix() const { return (long)Ny * Nx; }
};

// The launches of the compile-time-plan kernels (kernels_ct.hpp) of ONE length N: declared here, defined in engine_ct.hpp and instantiated per
// length by tu_cty_* (column side + the plain transforms) and tu_ctx_* (row side of the fused stages), so that no other translation unit
// compiles those kernels and the build splits the list of lengths over several units.  Sx / S: wavefronts per workgroup of the launch.
template <typename T> struct Ctx;
template <typename T, int N> struct CtLaunchY {
  static void dftx(Ctx<T>* c, const GenDft<T>& a, dim3 grid, int Sx, int kind);
  static void dft(Ctx<T>* c, const GenDft<T>& a, dim3 grid, int kind);
  static void flow_y(Ctx<T>* c, const GenDft<T>& a, dim3 grid);
  static bool delta_y(Ctx<T>* c, const GenDft<T>& a, dim3 grid);       // false: two LDS row sets of this length do not fit
  static void adj_y(Ctx<T>* c, const GenDft<T>& a, dim3 grid);
};
template <typename T, int N> struct CtLaunchX {
  static void adj_x(Ctx<T>* c, const GenDft<T>& a, dim3 grid, int Sx);
  static void adj_x_dx(Ctx<T>* c, const GenDft<T>& a, dim3 grid, int Sx, int ws, const GenDft<T>& a1);
  static void dft2(Ctx<T>* c, const GenDft<T>& a0, int kind0, dim3 grid, int Sx, int ws, const GenDft<T>& a1, int kind1);
};

template <typename T>
struct Ctx : CtxBase {
  DevBuf twY, twX, lx_r, ly, lam, cos2F, sin2F, red_part, red_out;
  T dlx_over_Nx = 0;                       // dlx / Nx for the fused i*lx multiply of the d/dx row pass
  DevBuf tmpA, tmpB;                       // conversion scratch (mixed/F complex)
  DevBuf xtmp;                             // mixed-layout side of a 2-D transform (mixed_scratch)
  static constexpr int RED_BLOCKS = 256;

  Ctx(int Ny_, int Nx_, double theta_, int device_, void* stream_) {
    Ny = Ny_; Nx = Nx_; theta = theta_; device = device_; dtype = sizeof(T) == 4 ? 0 : 1;
    CMBL_REQUIRE(Ny >= 2 && Nx >= 2 && Ny <= 4096 && Nx <= 4096, ERR_SHAPE, "Ny and Nx must lie in [2, 4096]");
    CMBL_REQUIRE(theta > 0, ERR_ARG, "theta_pix must be positive");
    // powers of two >= 32 run the fused kernels; everything else (and everything, with CMBL_FORCE_GENERIC=1) the any-size path
    generic = !(ispow2(Ny) && ispow2(Nx) && Ny >= 32 && Nx >= 32) || env_int("CMBL_FORCE_GENERIC", 0) != 0;
    Nyh = Ny / 2 + 1; M = Ny / 2; lgM = ilog2(M); lgNx = ilog2(Nx);
    // the adjoint row pass keeps two row sets in LDS: where even one row of each does not fit next to the twiddle table (double
    // precision at Nx = 4096) the whole context runs the any-size path, whose transforms ping-pong in LDS one sequence at a time
    if (!generic && row_rpw<T>(lgNx, 2) == 0) generic = true;
    CMBL_HIP(hipSetDevice(device));
    { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) num_cus = n; }
    // NULL is the legacy default stream (what torch.cuda.current_stream().cuda_stream returns when the caller has not
    // switched streams): ordered against every other blocking stream, so caller-side copies and our kernels serialise.
    stream = (hipStream_t)stream_; own_stream = false;
    build_geometry();
  }

  // src/proj_lambert.jl:58-71 (computed in T like the reference, tables uploaded in the internal layouts)
  void build_geometry() {
    const T dx = (T)(theta / 60.0 * M_PI / 180.0);
    const T dlx = (T)(2.0 * M_PI / (double)((T)Nx * dx));
    const T dly = (T)(2.0 * M_PI / (double)((T)Ny * dx));
    dlx_over_Nx = dlx / (T)Nx;
    std::vector<T> lx(Nx), lyv(Nyh), lamv(Nyh, (T)2);
    for (int i = 0; i < Nx; ++i) { int k = i < (Nx + 1) / 2 ? i : i - Nx; lx[i] = (T)k * dlx; }      // ifftshift(-N÷2:(N-1)÷2)
    for (int i = 0; i < Nyh; ++i) { int k = i < (Ny + 1) / 2 ? i : i - Ny; lyv[i] = (T)k * dly; }     // last entry negative
    lamv[0] = 1; if (Ny % 2 == 0) lamv[Nyh - 1] = 1;                                                    // util_fft.jl:137-143
    std::vector<T> s2((size_t)Nx * Nyh), c2((size_t)Nx * Nyh), lm((size_t)Nx * Nyh);
    for (int x = 0; x < Nx; ++x)
      for (int k = 0; k < Nyh; ++k) {
        const T ph = std::atan2(lyv[k], lx[x]);
        s2[(size_t)x * Nyh + k] = std::sin(2 * ph);
        c2[(size_t)x * Nyh + k] = std::cos(2 * ph);
        lm[(size_t)x * Nyh + k] = std::sqrt(lx[x] * lx[x] + lyv[k] * lyv[k]);
      }
    // sin2ϕ[end, end:-1:(Nx÷2+2)] .= sin2ϕ[end, 2:Nx÷2]  (1-based; proj_lambert.jl:69-71)
    if (Ny % 2 == 0) for (int i = 0; i < Nx / 2 - 1; ++i) s2[(size_t)(Nx - 1 - i) * Nyh + (Nyh - 1)] = s2[(size_t)(1 + i) * Nyh + (Nyh - 1)];
    h_lx.assign(lx.begin(), lx.end()); h_ly.assign(lyv.begin(), lyv.end()); h_lam.assign(lamv.begin(), lamv.end());
    h_sin2.assign(s2.begin(), s2.end()); h_cos2.assign(c2.begin(), c2.end()); h_lmag.assign(lm.begin(), lm.end());
    // internal tables
    // x slot i of the F layout holds frequency index xfreq(i): bit-reversed on the fused path, natural on the any-size path
    auto xfreq = [&](int i) { if (generic) return (unsigned)i; unsigned r = 0; for (int b = 0; b < lgNx; ++b) r |= ((i >> b) & 1u) << (lgNx - 1 - b); return r; };
    std::vector<T> lxr(Nx);
    for (int i = 0; i < Nx; ++i) lxr[i] = lx[xfreq(i)];
    std::vector<T> s2F((size_t)Nx * Nyh), c2F((size_t)Nx * Nyh);
    for (int k = 0; k < Nyh; ++k)
      for (int i = 0; i < Nx; ++i) {
        const unsigned r = xfreq(i);
        s2F[(size_t)k * Nx + i] = s2[(size_t)r * Nyh + k];
        c2F[(size_t)k * Nx + i] = c2[(size_t)r * Nyh + k];
      }
    std::vector<cx<T>> ty(Ny), tx(Nx);                                         // full-circle twiddle tables exp(-2 pi i k / N)
    for (int k = 0; k < Ny; ++k) { double a = -2.0 * M_PI * k / Ny; ty[k] = mk<T>((T)std::cos(a), (T)std::sin(a)); }
    for (int k = 0; k < Nx; ++k) { double a = -2.0 * M_PI * k / Nx; tx[k] = mk<T>((T)std::cos(a), (T)std::sin(a)); }
    upload(twY, ty); upload(twX, tx); upload(lx_r, lxr); upload(ly, lyv); upload(lam, lamv); upload(cos2F, c2F); upload(sin2F, s2F);
    red_part.ensure(sizeof(double) * RED_BLOCKS * MAXBATCH * 2);
    red_out.ensure(sizeof(double) * MAXBATCH);
    if (generic) { build_axis(genY, Ny); build_axis(genX, Nx); }
  }

  // ---- any-size transforms (kernels_generic.hpp) -------------------------------------------------
  struct GenAxis { int N = 0, lgL = 0; DevBuf chirp, bhat, tw, twN; GenPlan plan{}; };
  GenAxis genY, genX;
  static void host_fft(std::vector<std::complex<double>>& v) {             // in-place radix-2, e^{-i}; table set-up only
    const size_t n = v.size();
    for (size_t i = 1, j = 0; i < n; ++i) { size_t bit = n >> 1; for (; j & bit; bit >>= 1) j ^= bit; j ^= bit; if (i < j) std::swap(v[i], v[j]); }
    for (size_t len = 2; len <= n; len <<= 1) {
      const double ang = -2.0 * M_PI / (double)len;
      for (size_t i = 0; i < n; i += len)
        for (size_t k = 0; k < len / 2; ++k) {
          const std::complex<double> w(std::cos(ang * (double)k), std::sin(ang * (double)k)), u = v[i + k], t = v[i + k + len / 2] * w;
          v[i + k] = u + t; v[i + k + len / 2] = u - t;
        }
    }
  }
  void build_axis(GenAxis& ax, int N) {
    ax.N = N; ax.lgL = std::max(3, ilog2(2 * N - 1));
    // mixed-radix plan when N has no prime factor above 13 (odd radices first, then 4s, then a 2); otherwise chirp-z
    ax.plan = GenPlan{};
    if (env_int("CMBL_GEN_BLUESTEIN", 0) == 0) {
      int rem = N, nf = 0, f[32];
      for (int p : {13, 11, 7, 5, 3}) while (rem % p == 0 && nf < 14) { f[nf++] = p; rem /= p; }
      while (rem % 4 == 0 && nf < 14) { f[nf++] = 4; rem /= 4; }
      if (rem % 2 == 0 && nf < 14) { f[nf++] = 2; rem /= 2; }
      if (rem == 1) { ax.plan.nf = nf; for (int i = 0; i < nf; ++i) ax.plan.radix[i] = f[i]; }
    }
    if (ax.plan.nf > 0) {
      std::vector<cx<T>> t(N);
      for (int k = 0; k < N; ++k) { const double a = -2.0 * M_PI * k / N; t[k] = mk<T>((T)std::cos(a), (T)std::sin(a)); }
      upload(ax.twN, t);
      return;
    }
    const int L = 1 << ax.lgL;
    std::vector<std::complex<double>> w(N), b(L, 0.0);
    for (int n = 0; n < N; ++n) {                                           // exp(-i pi n^2 / N), the phase reduced exactly
      const double a = -M_PI * (double)(((long)n * n) % (2L * N)) / (double)N;
      w[n] = {std::cos(a), std::sin(a)};
    }
    b[0] = std::conj(w[0]);
    for (int n = 1; n < N; ++n) b[n] = b[L - n] = std::conj(w[n]);
    host_fft(b);
    std::vector<cx<T>> wc(N), bh(L), tw(L);
    for (int n = 0; n < N; ++n) wc[n] = mk<T>((T)w[n].real(), (T)w[n].imag());
    for (int s = 0; s < L; ++s) {                                           // slot s of a DIF output holds frequency brev(s)
      unsigned r = 0; for (int bb = 0; bb < ax.lgL; ++bb) r |= ((s >> bb) & 1u) << (ax.lgL - 1 - bb);
      bh[s] = mk<T>((T)(b[r].real() / L), (T)(b[r].imag() / L));
    }
    for (int k = 0; k < L; ++k) { const double a = -2.0 * M_PI * k / L; tw[k] = mk<T>((T)std::cos(a), (T)std::sin(a)); }
    upload(ax.chirp, wc); upload(ax.bhat, bh); upload(ax.tw, tw);
  }
  // Slice window of the any-size launches (set by Flow::GenWindow around the launches of one group of slices, which then go to that
  // group's stream): every transform launch covers slices [gw0, gw0 + gwn) of the gwall slices of the flow -- of each part of its batch
  // when that is a multiple of gwall (pair launches).  gwn < 0: no window.
  long gw0 = 0, gwn = -1, gwall = 0;
  long gen_window(GenDft<T>& a, long slices) const {
    if (gwn < 0) return slices;
    CMBL_REQUIRE(gwall > 0 && slices % gwall == 0, ERR_STATE, "any-size launch outside its slice window");
    a.sl0 = (int)gw0; a.sln = (int)gwn; a.slstride = (int)gwall;
    return slices / gwall * gwn;
  }
  // lengths with a compile-time plan (kernels_ct.hpp): one wavefront per sequence, S = 64 bytes of sequences per workgroup
  // rows per workgroup of an x-pass launch over `rows` contiguous rows (`per`: wavefronts per row): the full group of ct_S wavefronts unless the
  // launch would then have fewer workgroups than HALF the CUs; down to a quarter (2 in single, 1 in double precision).  Measured
  // (profiles/r06_ab_anysize_row_groups.txt, first rule "fewer than the CUs"): 360^2 QU grad lnP -8 %, 768^2 QU -2.7 %, but 768^2 T+QU +1 % and 1536^2
  // +4 % (145 / 193 groups of 8: nearly one per CU already) -- hence the half
  int ct_rows_per_group(long rows, int per) const {
    int S = ct_S<T>();
    if (!opts.gen_ct_rows) return S;
    while (S > std::max(per, ct_S<T>() / 4) && (rows * per + S - 1) / S < num_cus / 2) S >>= 1;
    return S;
  }
  // columns per workgroup of a fused y-pass launch over `cols` columns of `slices` slices: the full group `S` (64-byte pieces on the transposed
  // side) unless the launch would then have fewer than 0.4 workgroups per CU -- then half (32-byte pieces, but transform phases with half the
  // wavefronts per SIMD).  Measured (profiles/r06_ab_anysize_col_groups.txt): 768^2 QU, 96 groups per slice chain: (grad L)' -6 %, grad lnP -3 %;
  // 1000^2 QU, 125 groups: +5 % -- hence the 0.4.  Option gen_ct_rows switches it off with the row groups.
  int ct_cols_per_group(long cols, long slices, int S) const {
    if (!opts.gen_ct_rows || S != ct_S<T>() || opts.gen_ct_cols == 0) return S;
    if (opts.gen_ct_cols == 2) return S / 2;
    // tiled hand-off arrays (htile), 8-byte elements: 4 columns are ONE block -- contiguous at either width -- and the half-width group won at every
    // size and batch measured (1000^2 QU grad lnP -3.7 %, 1536^2 -4.7 %, 1000^2 T+QU -9 %, 768^2 QU at B = 4 -14.5 %; never slower:
    // profiles/r06_ab_anysize_tiled_cols.txt).  16-byte elements: the full group IS one block (half of it: +1...+3 % at 768^2 / 1000^2)
    if (htile && sizeof(T) == 4) return S / 2;
    return ((cols + S - 1) / S) * slices * 5 < 2L * num_cus ? S / 2 : S;
  }
  bool gen_dft_ct(const GenAxis& ax, GenDft<T> a, long slices);   // defined in engine_gen.hpp (tu_gen_*.hip)
  void gen_dft(const GenAxis& ax, GenDft<T> a, long slices);   // defined in engine_gen.hpp (tu_gen_*.hip)
  // (the mixed scratch between the two launches of a 2-D transform is a hand-off array like those of the flows: tiled under bit 8 of gen_tiled)
  struct HandOff2D {
    Ctx<T>* c; int keep;
    explicit HandOff2D(Ctx<T>* c_) : c(c_), keep(c_->htile) { c->htile = (c->opts.gen_tiled & 8) ? c->gen_tile() : 0; }
    ~HandOff2D() { c->htile = keep; }
  };
  void gen_rfft2(const T* map, cx<T>* F, long slices) {
    cx<T>* tmp = mixed_scratch(slices);
    HandOff2D ho(this);
    GenDft<T> a{};                                                          // y: map [x][y] -> tmp [ky][x]
    a.in = map; a.out = tmp; a.in_real = 1; a.nin = Ny; a.nout = Nyh; a.nseq = Nx; a.scale = 1;
    a.in_seq = Ny; a.in_elem = 1; a.in_slice = npix(); a.out_seq = 1; a.out_elem = Nx; a.out_slice = plane();
    hand_out<1>(a);
    gen_dft(genY, a, slices);
    GenDft<T> b{};                                                          // x: tmp [ky][x] -> F [ky][kx]
    b.in = tmp; b.out = F; b.nin = Nx; b.nout = Nx; b.nseq = Nyh; b.scale = 1;
    b.in_seq = Nx; b.in_elem = 1; b.in_slice = plane(); b.out_seq = Nx; b.out_elem = 1; b.out_slice = plane();
    hand_in<2>(b);
    gen_dft(genX, b, slices);
  }
  void gen_irfft2(const cx<T>* F, T* map, long slices) {
    cx<T>* tmp = mixed_scratch(slices);
    HandOff2D ho(this);
    GenDft<T> b{};                                                          // x (inverse): F [ky][kx] -> tmp [ky][x]
    b.in = F; b.out = tmp; b.nin = Nx; b.nout = Nx; b.nseq = Nyh; b.scale = 1; b.inverse = 1;
    b.in_seq = Nx; b.in_elem = 1; b.in_slice = plane(); b.out_seq = Nx; b.out_elem = 1; b.out_slice = plane();
    hand_out<2>(b);
    gen_dft(genX, b, slices);
    GenDft<T> a{};                                                          // y (c2r): tmp [ky][x] -> map [x][y]
    a.in = tmp; a.out = map; a.herm = 1; a.out_real = 1; a.inverse = 1; a.nin = Nyh; a.nout = Ny; a.nseq = Nx;
    a.scale = (T)(1.0 / ((double)Ny * Nx));
    a.in_seq = 1; a.in_elem = Nx; a.in_slice = plane(); a.out_seq = Ny; a.out_elem = 1; a.out_slice = npix();
    hand_in<1>(a);
    gen_dft(genY, a, slices);
  }
  // Separable pieces of the any-size path (mixed layout [slice][ky][x], like the fused kernels use between their column and row passes):
  // the y transform alone (one real map, or TWO real maps per complex transform), the x transform alone with an optional i*lx multiply
  // in its store, and the pair c2r that returns (d/dx f, d/dy f) from (Gx, A) with the i*ly multiply in its fetch.
  void gen_y_r2c(const T* map, cx<T>* A, long slices, const T* map2 = nullptr, cx<T>* A2 = nullptr, const GenPro<T>* pro = nullptr) {
    GenDft<T> a{};
    if (pro) a.pro = *pro;
    a.in = map; a.out = A; a.in2 = map2; a.out2 = A2; a.in_real = 1; a.nin = Ny; a.nout = Nyh; a.nseq = Nx; a.scale = 1; a.scale2 = 1;
    a.in_seq = Ny; a.in_elem = 1; a.in_slice = npix(); a.out_seq = 1; a.out_elem = Nx; a.out_slice = plane();
    hand_out<1>(a);
    gen_dft(genY, a, slices);
  }
  void gen_y_c2r_pair(const cx<T>* G1, const cx<T>* G2, const T* lmul2, T* o1, T* o2, T s1, T s2, long slices) {
    GenDft<T> a{};
    a.in = G1; a.in2 = G2; a.lmul_in = lmul2; a.out = o1; a.out2 = o2; a.herm = 1; a.out_real = 1; a.inverse = 1; a.nin = Nyh; a.nout = Ny; a.nseq = Nx;
    a.scale = s1; a.scale2 = s2;
    a.in_seq = 1; a.in_elem = Nx; a.in_slice = plane(); a.out_seq = Ny; a.out_elem = 1; a.out_slice = npix();
    hand_in<1>(a);
    gen_dft(genY, a, slices);
  }
  // the y axis has a compile-time plan (and they are switched on): the fused y passes of a flow stage exist
  bool gen_ct_y() const {
    if (!opts.gen_ct || genY.plan.nf == 0) return false;
    switch (Ny) {
#define CMBL_X(n) case n: return true;
      CMBL_CT_LIST(CMBL_X)
#undef CMBL_X
      default: return false;
    }
  }
  // pair c2r of (G1, i ly G2) + the stage's velocity / RK update on the maps of `pro` + rfft_y of the next stage input -> Anext, one launch
  // (GenDft::yy, kernels_ct.hpp ct_flow_stage); last: the flow ends, only y0 is updated
  void gen_y_flow_stage(const cx<T>* G1, const cx<T>* G2, const T* lmul2, T s1, T s2, const GenPro<T>& pro, cx<T>* Anext, bool last, long slices);   // defined in engine_gen.hpp (tu_gen_*.hip)
  // ... and the four y passes of a delta-flow stage (two LDS rows per column: up to Ny ~ 1150)
  bool gen_ct_y2() const { return gen_ct_y() && ct_lds<T>(Ny, 2, ct_S2<T>(Ny)) <= 160 * 1024; }
  // c2r of T3 = ifft_x(delta f) -> L(df), pair c2r of (G1, i ly G2) -> grad f, the delta stage's pointwise work on the maps of `pro`
  // (w1p, w2p, y0, acc), rfft_y of the next f -> Anext, pair r2c of (p_x, p_y) L(df) -> (W2a, W2b): one launch (k_ct_delta_y)
  void gen_y_delta_stage(const cx<T>* T3, T s3, const cx<T>* G1, const cx<T>* G2, const T* lmul2, T s1, T s2, const GenPro<T>& pro, cx<T>* Anext,                          cx<T>* W2a, cx<T>* W2b, bool last, long slices);   // defined in engine_gen.hpp (tu_gen_*.hip)
  // fft_x of the pair (W2a, W2b) and the RK update of the Fourier state (Y0, acc -> Ys) with k = i lx Fx + i ly Fy in ONE launch where the x axis has
  // a compile-time plan; false: not available (the caller runs the x transform and k_gen_adj_rk)
  bool gen_x_adj_update(const cx<T>* W2a, const cx<T>* W2b, cx<T>* Y0, cx<T>* acc_, cx<T>* Ys, const RKCoef<T>& rk, long slices);   // defined in engine_gen.hpp (tu_gen_*.hip)
  // the x axis has a compile-time plan: the row update of an adjoint-type stage can open the next stage in the same launch (gen_x_adj_next)
  bool gen_ct_x() const {
    if (!opts.gen_ct || !opts.gen_yy || !opts.gen_xmerge || genX.plan.nf == 0 || !ct_rowfuse_ok<T>(Nx)) return false;
    switch (Nx) {
#define CMBL_X(n) case n: return true;
      CMBL_CT_LIST(CMBL_X)
#undef CMBL_X
      default: return false;
    }
  }
  // Tiled hand-off arrays (GenDft::in_tiled): every launch that touches them must be a compile-time-plan kernel, i.e. both axes have a plan and
  // the fused stages are on; 4 | Nx for the blocks.  Returns the rows of a block column (ky padded to a multiple of 4), 0 = [ky][x] arrays.
  int gen_tile() const {
    if (!opts.gen_tiled || !opts.gen_separable || !opts.gen_prologue || !opts.gen_xderiv_fused || (Nx & 3) || !gen_ct_x() || !gen_ct_y()) return 0;
    return (Nyh + 3) & ~3;
  }
  // set by a flow around its stages (Flow::HandOff): the layout of gA / gGx / gW2 / the mixed scratch in the launches below; slice stride hplane()
  int htile = 0;
  long hplane() const { return htile ? mplane() : plane(); }
  template <int SIDE /*1: y kernel (sequence = x), 2: x kernel (element = x)*/> void hand_in(GenDft<T>& a) const { if (htile) { a.in_tiled = SIDE; a.tile_np = htile; a.in_slice = mplane(); } }
  template <int SIDE> void hand_out(GenDft<T>& a) const { if (htile) { a.out_tiled = SIDE; a.tile_np = htile; a.out_slice = mplane(); } }
  // row groups of an x launch on tiled arrays: a multiple of 8, so that xcd_tile (kernels_fft.hpp) puts the groups that share 128-byte lines on one XCD
  static long xgroups(long groups, bool tiled) { return tiled ? (groups + 7) & ~7L : groups; }
  void gen_x_adj_next(const cx<T>* W2a, const cx<T>* W2b, cx<T>* Y0, cx<T>* acc_, const RKCoef<T>& rk, cx<T>* t3, const cx<T>* A_next, cx<T>* gx, long slices);   // defined in engine_gen.hpp (tu_gen_*.hip)
  // t3 = ifft_x(F) (unnormalised) and gx = ifft_x(i lx fft_x(A)) in ONE launch where the x axis has a compile-time plan (else two launches)
  void gen_x_inv_and_deriv(const cx<T>* F, cx<T>* t3, const cx<T>* A_, cx<T>* gx, cx<T>* tmp, const T* lx, long slices);   // defined in engine_gen.hpp (tu_gen_*.hip)
  // c2r of T3 = ifft_x(y) -> y, (p_x y, p_y y) at stage time t, its pair r2c -> (W2a, W2b): the y passes of an adjoint stage in one launch
  void gen_y_adj_stage(const cx<T>* T3, T s3, const PhiMaps<T>& phm, T t, int P, cx<T>* W2a, cx<T>* W2b, long slices);   // defined in engine_gen.hpp (tu_gen_*.hip)
  // out = ifft_x(i lx fft_x(in)) unnormalised, in ONE launch when the axis has a mixed-radix plan (else two: chirp-z transforms)
  bool gen_x_deriv(const cx<T>* in, cx<T>* out, cx<T>* tmp, const T* lx, long slices) {
    if (genX.plan.nf == 0 || !opts.gen_xderiv_fused) {
      gen_x(in, tmp, false, lx, slices);
      gen_x(tmp, out, true, nullptr, slices);
      return false;
    }
    GenDft<T> b{};
    b.in = in; b.out = out; b.nin = Nx; b.nout = Nx; b.nseq = Nyh; b.scale = 1; b.lmul_mid = lx;
    b.in_seq = Nx; b.in_elem = 1; b.in_slice = plane(); b.out_seq = Nx; b.out_elem = 1; b.out_slice = plane();
    hand_in<2>(b); hand_out<2>(b);
    gen_dft(genX, b, slices);
    return true;
  }
  void gen_x(const cx<T>* in, cx<T>* out, bool inverse, const T* lmul_out, long slices, bool hand_i = false, bool hand_o = false) {
    GenDft<T> b{};
    b.in = in; b.out = out; b.nin = Nx; b.nout = Nx; b.nseq = Nyh; b.scale = 1; b.inverse = inverse ? 1 : 0; b.lmul_out = lmul_out;
    b.in_seq = Nx; b.in_elem = 1; b.in_slice = plane(); b.out_seq = Nx; b.out_elem = 1; b.out_slice = plane();
    if (hand_i) hand_in<2>(b);
    if (hand_o) hand_out<2>(b);
    gen_dft(genX, b, slices);
  }
  template <typename V> void transpose(const V* in, V* out, int R, int C, long slices) {
    CMBL_LAUNCH(this, K_LAYOUT, (k_transpose<V>), dim3((C + 31) / 32, (R + 31) / 32, (unsigned)slices), 0, stream, in, out, R, C);
  }
  template <typename V> void upload(DevBuf& b, const std::vector<V>& v) {
    b.ensure(v.size() * sizeof(V));
    CMBL_HIP(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(V), hipMemcpyHostToDevice, stream));
    CMBL_HIP(hipStreamSynchronize(stream));
  }

  // ---- launch geometry -----------------------------------------------------------------------
  // Column kernels are compiled for the tile shapes of CMBL_COL_LIST: (lgM, R, NT) with C = R*NT/M columns per workgroup of
  // NT threads and R packed pairs per thread.
  // CMBL_TUNE_C / CMBL_TUNE_NT force a tile width / workgroup size (tuning aids); the rows per row workgroup follow row_rpw / pick_rpw.
  struct TileY { int C, NT, R; };
  mutable TileY tile_cache[2][2] = {{{0, 0, 0}, {0, 0, 0}}, {{0, 0, 0}, {0, 0, 0}}};   // [pair][narrow]: made once per kind (host launch path)
  const int tuneC = env_int("CMBL_TUNE_C", 0), tuneNT = env_int("CMBL_TUNE_NT", 0);
  // workgroups a launch needs before the tallest row group / the four-column tile is taken (see row_rpw_variants)
  long row_fill_target() const { return opts.row_fill_target > 0 ? opts.row_fill_target : num_cus / 2; }
  TileY tileY(long slices, bool pair, int preferNT = 0) const {
    // small maps: a launch of four-column tiles leaves CUs idle (512^2: 128 tiles per slice) -- take the two-column tile where compiled.
    // `slices` = all slices of the operation, not of one launch chain of it: the choice (and with it the rounding of the results) must not
    // depend on how a flow is split over streams.  Up to 512 rows only: from 1024 rows on the narrowest tile with C >= 4 measured best
    // (profiles/r02_variants.txt), and nothing narrower was measured there.
    // Default rule (fill_target = 0): narrow when four-column tiles leave CUs idle (fewer tiles than CUs) or load them unevenly (between one and
    // two tiles per CU, not a whole number: 512^2 T+QU, 384 tiles, -4 %).  Exactly one tile per CU (512^2 QU) stays on four columns: two-column
    // tiles measured within 1 % there while the L2 <-> fabric counters read 1.26 x the compulsory bytes of delta_cols instead of 1.01 (each 128-byte
    // line is then requested by two workgroups).
    const long wtiles = (long)(Nx / 4) * slices;
    const bool narrow = (opts.occupancy_tiles & 1) && lgM <= 8 &&
                        (opts.fill_target > 0 ? wtiles < opts.fill_target : (wtiles < num_cus || (wtiles < 2 * num_cus && wtiles % num_cus != 0)));
    if (preferNT == 0 && tile_cache[pair][narrow].C > 0) return tile_cache[pair][narrow];
    static const int list[][3] = {
#define CMBL_X(lgm, r, nt) {lgm, r, nt},
        CMBL_COL_LIST(CMBL_X)
#undef CMBL_X
    };
    // Measured on MI355X (1024^2, B = 1..8): the kernels are bound by resident waves per CU (register file), not by segment
    // width, so the narrowest tile with C >= 4 wins, 512 threads when compiled (more waves per tile, fewer registers per thread).
    const int forceC = tuneC, forceNT = tuneNT;
    TileY best{0, 0, 0};
    long bestScore = -1;
    for (const auto& e : list) {
      if (e[0] != lgM) continue;
      const int C = (int)(((long)e[1] * e[2]) >> lgM);
      if (C > Nx || ldsY(C, pair) > 160 * 1024) continue;
      const TileY t{C, e[2], e[1]};
      if (forceC == C && (forceNT == 0 || forceNT == e[2])) { if (preferNT == 0) tile_cache[pair][narrow] = t; return t; }
      const int wantC = narrow ? 2 : 4;
      const long score = (C >= wantC ? 1000 - C : C) * 10 + (e[2] == preferNT ? 3 : (e[2] == 512 ? 2 : (e[2] == 256 ? 1 : 0)));
      if (score > bestScore) { bestScore = score; best = t; }
    }
    CMBL_REQUIRE(best.C > 0, ERR_SHAPE, "no compiled column-tile shape fits this Ny / precision");
    if (preferNT == 0) tile_cache[pair][narrow] = best;
    return best;
  }
  // column tile: twiddles + C columns of an N-point (pair) or M-point (packed) transform, padded rows
  // The delta-flow column kernel holds the most state per thread: at 2048 rows in double precision its four-column tile (8 pairs per
  // thread) needs 256 registers + 80 spilled, the two-column tile (4 pairs) fits -- measured (grad L)' 10.95 -> 10.45 ms at 2048^2 fp64,
  // while the forward and adjoint column kernels are faster on four columns (L'g 5.13 vs 5.45 ms).
  TileY tileY_delta(long slices) const {
    if (sizeof(T) == 8 && lgM == 10 && tuneC == 0) {
      TileY t = tileY(slices, true, 0);
      static const int list[][3] = {
#define CMBL_X(lgm, r, nt) {lgm, r, nt},
          CMBL_COL_LIST(CMBL_X)
#undef CMBL_X
      };
      for (const auto& e : list)
        if (e[0] == lgM && e[2] == (opts.col_pipeline >= 2 ? 256 : 512) && (((long)e[1] * e[2]) >> lgM) == 2 && ldsY(2, true) <= 160 * 1024) return TileY{2, e[2], e[1]};   // (col_pipeline = 2, experiment builds: the 256-thread tile, whose 512 registers per thread hold a prefetched head tile)
      return t;
    }
    return tileY(slices, true);
  }
  // tiles per workgroup of the experimental software-pipelined column kernel (-DCMBL_EXPERIMENT_COL_PIPELINE; col_pipelined shapes)
  int col_tpw(int tiles, long slices) const {
    if (opts.col_pipeline <= 0) return 1;
    int tpw = 2;
    while (tpw > 1 && (tiles % (8 * tpw) != 0 || (long)(tiles / tpw) * slices < num_cus)) tpw >>= 1;
    return tpw;
  }
  // Touch-prefetch distance of a column launch (kernels_flow.hpp TouchTiles): three quarters of the workgroups of this launch that are resident at
  // once -- the CUs, shared by the K launch chains that run side by side -- as a multiple of 8 (XCDs): measured best at 80-96 for two chains and
  // 192 for one (profiles/r05_ab_touch_prefetch.txt).  Only for the shapes that run one workgroup per CU over many residency rounds (double
  // precision, 2048 rows); launches of fewer than three rounds have nobody to prefetch for.
  int col_prefetch(long tiles, long slices, int K) const {
    if (opts.col_prefetch >= 0) return opts.col_prefetch;
    if (!col_touch<T>(lgM)) return 0;
    const long resident = num_cus / std::max(K, 1);
    return tiles * slices >= 3 * resident ? (int)std::max<long>(8, (3 * resident / 4) & ~7L) : 0;
  }
  // (+ the 256-byte pad of the touch prefetch behind the tile for the shapes that prefetch: kernels_flow.hpp TouchTiles)
  size_t ldsY(int C, bool pair = true) const { return ((size_t)M + (size_t)C * tile_ld(pair ? 2 * M : M)) * sizeof(cx<T>) + (pair && col_touch<T>(lgM) ? TOUCH_PAD : 0); }
  // Row launches: one workgroup per group of RPW adjacent ky rows of a slice (RPW = row_rpw<T>(lgNx, row sets), kernels_fft.hpp)
  // Row launches of about one workgroup per CU (258 row groups at 1024^2 QU) run faster when no CU hosts two of them: two co-resident
  // row workgroups are VALU-issue bound and the launch ends with its slowest workgroup.  Asking for more than half of the CU's LDS
  // keeps them apart; the few workgroups beyond the CU count are the short ones (row_group) and run in a second, short round.
  const size_t lds_one_per_cu = env_int("CMBL_ONE_PER_CU", 1) ? (size_t)82 * 1024 : 0;
  size_t lds_apart(size_t lds, long nblk) const { return (lds_one_per_cu && nblk <= num_cus + num_cus / 8) ? std::max(lds, lds_one_per_cu) : lds; }
  size_t ldsX(int rpw, int nbuf) const { return ((size_t)row_tw<T>(Nx) + (size_t)nbuf * rpw * row_ld(Nx)) * sizeof(cx<T>); }
  long row_groups(long slices, int rpw) const { return slices * ((Nyh + rpw - 1) / rpw); }
  // rows per row workgroup of a launch over `slices` slices: the LDS-fit maximum unless the launch would then leave CUs idle
  int pick_rpw(int rpw_max, int lgnx, long slices) const {
    if (!(opts.occupancy_tiles & 2) || !row_rpw_variants(lgnx)) return rpw_max;
    int rpw = rpw_max;
    while (rpw > 1 && row_groups(slices, rpw) < row_fill_target()) rpw >>= 1;
    return rpw;
  }
  template <int RPWMAX, int LGNX, typename Fn> void dispatch_rpw(long slices, Fn&& fn) const {
    if constexpr (row_rpw_variants(LGNX)) {
      const int rpw = pick_rpw(RPWMAX, LGNX, slices);
      if constexpr (RPWMAX >= 4) { if (rpw == 4) return fn(std::integral_constant<int, 4>{}); }
      if constexpr (RPWMAX >= 2) { if (rpw == 2) return fn(std::integral_constant<int, 2>{}); }
      fn(std::integral_constant<int, 1>{});
    } else fn(std::integral_constant<int, RPWMAX>{});
  }

  template <typename Fn> void dispatch_col(const TileY& t, Fn&& fn) const {
    bool done = false;
#define CMBL_X(lgm, r, nt) if (!done && lgM == lgm && t.R == r && t.NT == nt) { fn(std::integral_constant<int, lgm>{}, std::integral_constant<int, r>{}, std::integral_constant<int, nt>{}); done = true; }
    CMBL_COL_LIST(CMBL_X)
#undef CMBL_X
    if (!done) fail(ERR_SHAPE, "unsupported column tile");
  }
  template <typename Fn> void dispatch_row(Fn&& fn) const {
    bool done = false;
#define CMBL_X(lgnx) if (!done && lgNx == lgnx) { fn(std::integral_constant<int, lgnx>{}); done = true; }
    CMBL_ROW_LIST(CMBL_X)
#undef CMBL_X
    if (!done) fail(ERR_SHAPE, "unsupported Nx");
  }

  // ---- layout / transform primitives (all on `stream`) -----------------------------------------
  void ref2F(const cx<T>* in, cx<T>* out, long slices) {
    if (generic) return transpose(in, out, Nx, Nyh, slices);
    CMBL_LAUNCH(this, K_LAYOUT, (k_ref2F<cx<T>>), dim3(Nx / 32, (Nyh + 31) / 32, (unsigned)slices), 0, stream, in, out, Nx, lgNx, Nyh);
  }
  void F2ref(const cx<T>* in, cx<T>* out, long slices) {
    if (generic) return transpose(in, out, Nyh, Nx, slices);
    CMBL_LAUNCH(this, K_LAYOUT, (k_F2ref<cx<T>>), dim3(Nx / 32, (Nyh + 31) / 32, (unsigned)slices), 0, stream, in, out, Nx, lgNx, Nyh);
  }
  void ref2F_real(const T* in, T* out, long slices) {
    if (generic) return transpose(in, out, Nx, Nyh, slices);
    CMBL_LAUNCH(this, K_LAYOUT, (k_ref2F<T>), dim3(Nx / 32, (Nyh + 31) / 32, (unsigned)slices), 0, stream, in, out, Nx, lgNx, Nyh);
  }
  void y_r2c(const T* map, cx<T>* mixed, long slices) {
    CMBL_REQUIRE(!generic, ERR_STATE, "fused column pass called on the any-size path");
    const TileY t = tileY(slices, false);
    dispatch_col(t, [&](auto lgm, auto r, auto nt) {
      constexpr int LGM = decltype(lgm)::value, R = decltype(r)::value, NT = decltype(nt)::value;
      CMBL_LAUNCH_NT(this, K_Y_R2C, NT, (k_y_r2c<T, R, NT, LGM>), dim3(Nx / t.C, (unsigned)slices), ldsY(t.C, false), stream, map, mixed, twY.as<cx<T>>(), Nx);
    });
  }
  void y_c2r(const cx<T>* mixed, T* map, long slices) {
    CMBL_REQUIRE(!generic, ERR_STATE, "fused column pass called on the any-size path");
    const TileY t = tileY(slices, false);
    dispatch_col(t, [&](auto lgm, auto r, auto nt) {
      constexpr int LGM = decltype(lgm)::value, R = decltype(r)::value, NT = decltype(nt)::value;
      CMBL_LAUNCH_NT(this, K_Y_C2R, NT, (k_y_c2r<T, R, NT, LGM>), dim3(Nx / t.C, (unsigned)slices), ldsY(t.C, false), stream, mixed, map, twY.as<cx<T>>(), Nx, (T)(1.0 / Ny));
    });
  }
  // mixed -> (map x mask) -> mixed, for the pixel-mask sandwich (in place allowed)
  void y_mask(const cx<T>* in, cx<T>* out, const T* mask, long slices) {
    CMBL_REQUIRE(!generic, ERR_STATE, "fused column pass called on the any-size path");
    const TileY t = tileY(slices, false);
    dispatch_col(t, [&](auto lgm, auto r, auto nt) {
      constexpr int LGM = decltype(lgm)::value, R = decltype(r)::value, NT = decltype(nt)::value;
      CMBL_LAUNCH_NT(this, K_MASK, NT, (k_y_mask<T, R, NT, LGM>), dim3(Nx / t.C, (unsigned)slices), ldsY(t.C, false), stream, in, out, mask, twY.as<cx<T>>(), Nx,
                     (T)(1.0 / Ny));
    });
  }
  // all_slices: the slices of the whole operation when this call is one launch chain of several (the row-group height follows it)
  template <int MODE> void x_pass(const cx<T>* in, cx<T>* out, long slices, hipStream_t st = nullptr, long all_slices = 0) {
    if (!st) st = stream;
    if (all_slices <= 0) all_slices = slices;
    CMBL_REQUIRE(in != out, ERR_ARG, "x pass cannot run in place (tiled mixed layout on one side)");
    CMBL_REQUIRE(!generic, ERR_STATE, "fused row pass called on the any-size path");
    dispatch_row([&](auto lgnx) {
      constexpr int LGNX = decltype(lgnx)::value, RPWMAX = row_rpw<T>(LGNX, 1);
      if constexpr (RPWMAX > 0) {
        dispatch_rpw<RPWMAX, LGNX>(all_slices, [&](auto rpw_) {
          constexpr int RPW = decltype(rpw_)::value;
          CMBL_LAUNCH_NT(this, (MODE == 2 ? K_X_GRAD : K_X_FFT), row_nt(RPW), (k_x_fft<T, MODE, LGNX, RPW>), dim3((unsigned)row_groups(slices, RPW)),
                         ldsX(RPW, 1), st,
                         in, out, twX.as<cx<T>>(), dlx_over_Nx, Nyh);
        });
      } else fail(ERR_SHAPE, "row tile does not fit LDS (Nx too large for this precision)");
    });
  }
  // map -> F  (m_rfft, src/util_fft.jl:20)
  // (the x pass reads the tiled mixed layout and writes F rows, or the reverse: it cannot run in place -- `xtmp` holds the mixed side)
  cx<T>* mixed_scratch(long slices) { xtmp.ensure(sizeof(cx<T>) * slices * mplane()); return xtmp.as<cx<T>>(); }
  void rfft2_F(const T* map, cx<T>* F, long slices) { if (generic) return gen_rfft2(map, F, slices); cx<T>* m = mixed_scratch(slices); y_r2c(map, m, slices); x_pass<0>(m, F, slices); }
  // F -> map without a caller-provided scratch (F is left intact)
  void F_to_map(const cx<T>* F, T* map, long slices) { if (generic) return gen_irfft2(F, map, slices); cx<T>* m = mixed_scratch(slices); x_pass<1>(F, m, slices); y_c2r(m, map, slices); }

  // harmonic-operator application (see k_harm_apply)
  void harm(const cx<T>* in, cx<T>* out, int P, int B, int kind, const T* const* d, bool transpose, bool in_qu, bool out_qu,
            const cx<T>* z = nullptr, T alpha = 0, T beta = 1) {
    HarmOpArgs<T> a{};
    a.in = in; a.out = out; a.z = z; a.cos2 = cos2F.as<T>(); a.sin2 = sin2F.as<T>();
    for (int k = 0; k < 5; ++k) a.d[k] = d ? d[k] : nullptr;
    a.kind = kind; a.in_qu = in_qu; a.out_qu = out_qu; a.transpose = transpose; a.alpha = alpha; a.beta = beta;
    a.plane = plane(); a.B = B;
    const dim3 grid((unsigned)((plane() + NTP - 1) / NTP));
    if (P == 1) CMBL_LAUNCH(this, K_HARM, (k_harm_apply<T, 1>), grid, 0, stream, a);
    else if (P == 2) CMBL_LAUNCH(this, K_HARM, (k_harm_apply<T, 2>), grid, 0, stream, a);
    else CMBL_LAUNCH(this, K_HARM, (k_harm_apply<T, 3>), grid, 0, stream, a);
  }

  // out = a*x + c*y with per-batch scalars (n = reals per batch slot)
  void lincomb(T* out, const T* x, const T* y, const double* a, const double* c, long n, int B) {
    for (int b0 = 0; b0 < B; b0 += MAXB) {
      const int nb = std::min(MAXB, B - b0);
      BScal<T> sa{}, sc{};
      for (int i = 0; i < nb; ++i) { sa.v[i] = (T)a[b0 + i]; sc.v[i] = c ? (T)c[b0 + i] : (T)0; }
      const unsigned gx = (unsigned)std::min<long>((n + NTP - 1) / NTP, 2048);
      CMBL_LAUNCH(this, K_LINCOMB, (k_lincomb<T>), dim3(gx, nb), 0, stream, out, x, y, sa, sc, n, b0);
    }
  }
  void lincomb1(T* out, const T* x, const T* y, double a, double c, long n, int B) {
    std::vector<double> va(B, a), vc(B, c);
    lincomb(out, x, y, va.data(), vc.data(), n, B);
  }
  void mask_mul(T* out, const T* in, const T* m, long slices) {
    const unsigned gx = (unsigned)std::min<long>((npix() + NTP - 1) / NTP, 2048);
    CMBL_LAUNCH(this, K_MASK, (k_mask_mul<T>), dim3(gx, (unsigned)slices), 0, stream, out, in, m, npix());
  }

  // ---- fused harmonic work (kernels_harm.hpp) -----------------------------------------------------------------------------
  // partial sums of the fused launches: region r of dot_part holds [B][nblk] doubles of one producer (PART_*), finished by its consumer
  enum PartRegion { PART_QF = 0, PART_QP, PART_QN, PART_CG_RZ, PART_CG_PAP, PART_COUNT };
  // doubles per region: B <= MAXBATCH slots x blocks per slot <= 2056 (a row carrier at Ny = 4096 with one row per workgroup has
  // Ny/2 + 1 = 2049 row groups per slot; the flat kernels at most 1024)
  static constexpr size_t PART_STRIDE = (size_t)MAXBATCH * 2056;
  DevBuf dot_part;
  ModeGeom<T> geom() const { return ModeGeom<T>{cos2F.as<T>(), sin2F.as<T>(), lam.as<T>(), plane(), Nx}; }
  double dot_scale() const { return 1.0 / ((double)Ny * Nx); }
  DotOut dot_out(int region, int nblk, int B) {
    CMBL_REQUIRE((size_t)B * nblk <= PART_STRIDE, ERR_SHAPE, "too many partial sums");
    dot_part.ensure(sizeof(double) * PART_STRIDE * PART_COUNT);
    return DotOut{dot_part.as<double>() + (size_t)region * PART_STRIDE, nblk, sum_mode};
  }
  // row carrier: P pol slices of each of B batch slots; in: mixed (IN_F = false) or F (true); out_mixed nullable.  Returns where the
  // launch left its partial sums (region `region`)
  template <int P, bool IN_F, typename PW> DotOut x_pw(const cx<T>* in, cx<T>* out_mixed, const PW& pw, int B, bool herm = false, int region = 0) {
    CMBL_REQUIRE(!generic, ERR_STATE, "fused row pass called on the any-size path");
    CMBL_REQUIRE(B <= MAXBATCH, ERR_ARG, "nbatch > 256 not supported in reductions");
    DotOut o{};
    dispatch_row([&](auto lgnx) {
      constexpr int LGNX = decltype(lgnx)::value, RPW = xpw_rpw<T>(LGNX, P);
      if constexpr (RPW > 0) {
        const int G = (Nyh + RPW - 1) / RPW;
        o = dot_out(region, G, B);
        CMBL_LAUNCH_NT(this, K_XPW, xpw_nt(P, RPW), (k_x_pw<T, LGNX, RPW, P, IN_F, PW>), dim3((unsigned)(B * G)), ldsX(RPW, P), stream, in, out_mixed,
                       twX.as<cx<T>>(), XPwIo{Nyh, herm ? 1 : 0}, pw, o, B);
      } else fail(ERR_SHAPE, "row tile does not fit LDS (Nx too large for this precision)");
    });
    return o;
  }
  int flat_blocks(int B) const { return (int)std::max<long>(1, std::min<long>((plane() + NTP - 1) / NTP, std::max(64, 1024 / B))); }
  template <typename PW> DotOut pw_flat(const PW& pw, int B, int region = 0) {
    CMBL_REQUIRE(B <= MAXBATCH, ERR_ARG, "nbatch > 256 not supported in reductions");
    const int nblk = flat_blocks(B);
    const DotOut o = dot_out(region, nblk, B);
    CMBL_LAUNCH(this, K_PWFLAT, (k_pw_flat<T, PW>), dim3((unsigned)nblk, (unsigned)B), 0, stream, pw, o, plane(), Nx, B);
    return o;
  }
  // totals of up to four producers' partials -> out[j][b] (device doubles), one launch
  void finish_parts(const DotOut* parts, double* const* outs, int nreg, int B) {
    PartRegions r{};
    for (int j = 0; j < nreg; ++j) { r.part[j] = parts[j].part; r.nblk[j] = parts[j].nblk; r.out[j] = outs[j]; }
    CMBL_LAUNCH(this, K_REDUCE, (k_finish_parts<T>), dim3((unsigned)(nreg * B)), 0, stream, r, B, dot_scale(), sum_mode);
  }
  // can the five-map delta-phi epilogue run as one row launch (five row sets in LDS)?
  bool x_dphi_fits() const {
    bool ok = false;
    if (!generic) dispatch_row([&](auto lgnx) { ok = row_rpw<T>(decltype(lgnx)::value, 5) > 0; });
    return ok;
  }
  void x_dphi(const cx<T>* in_mixed5, cx<T>* out, int B, const DphiTail<T>& tail) {
    dispatch_row([&](auto lgnx) {
      constexpr int LGNX = decltype(lgnx)::value, RPW = row_rpw<T>(LGNX, 5);
      if constexpr (RPW > 0) {
        const int G = (Nyh + RPW - 1) / RPW;
        CMBL_LAUNCH_NT(this, K_DPHI_X, row_nt(RPW), (k_x_dphi<T, LGNX, RPW>), dim3((unsigned)(B * G)), ldsX(RPW, 5), stream, in_mixed5, out, twX.as<cx<T>>(),
                       lx_r.as<T>(), ly.as<T>(), Nyh, B, tail);
      } else fail(ERR_SHAPE, "row tile does not fit LDS");
    });
  }

  // ---- per-batch reductions (src/proj_lambert.jl:318-353) ----------------------------------------------------------------
  // reduce_dev: two launches, the B results land in `out_dev` (device doubles); nothing synchronises.  `sum_mode` selects the
  // accumulation (set_sum_accuracy_mode!, src/util.jl:288-316).
  template <typename F> void reduce_dev(const F& f, long n, int B, double scale, double* out_dev) {
    CMBL_REQUIRE(B <= MAXBATCH, ERR_ARG, "nbatch > 256 not supported in reductions");
    switch (sum_mode) {
#define CMBL_X(M)                                                                                                                     \
      case M:                                                                                                                         \
        CMBL_LAUNCH(this, K_REDUCE, (k_reduce_terms<T, M, F>), dim3(RED_BLOCKS, B), 0, stream, f, red_part.as<double>(), n);           \
        CMBL_LAUNCH(this, K_REDUCE, (k_reduce_final<T, M>), dim3(B), 0, stream, red_part.as<double>(), out_dev, RED_BLOCKS, scale);    \
        break;
      CMBL_X(SUM_WORKING) CMBL_X(SUM_FLOAT64) CMBL_X(SUM_KAHAN)
#undef CMBL_X
      default: fail(ERR_ARG, "bad sum accuracy mode");
    }
  }
  void fetch(double* out_host, const double* dev, int B) {
    CMBL_HIP(hipMemcpyAsync(out_host, dev, sizeof(double) * B, hipMemcpyDeviceToHost, stream));
    CMBL_HIP(hipStreamSynchronize(stream));
  }
  void dot_F_dev(const cx<T>* a, const cx<T>* b, int P, int B, double* out_dev) {
    reduce_dev(TermDotF<T>{a, b, lam.as<T>(), (long)P * plane(), Nx, Nyh}, (long)P * plane(), B, 1.0 / ((double)Ny * Nx), out_dev);
  }
  void dot_F(const cx<T>* a, const cx<T>* b, int P, int B, double* out_host) {
    dot_F_dev(a, b, P, B, red_out.as<double>()); fetch(out_host, red_out.as<double>(), B);
  }
  void dot_map(const T* a, const T* b, int P, int B, double* out_host) {
    reduce_dev(TermDotMap<T>{a, b, (long)P * npix()}, (long)P * npix(), B, 1.0, red_out.as<double>());
    fetch(out_host, red_out.as<double>(), B);
  }
  // logdet of real operator planes in F layout (all planes of one operator: one "batch slot")
  void logdet_F(const T* d, int nplanes, double* out_host) {
    reduce_dev(TermLogdetF<T>{d, lam.as<T>(), (long)nplanes * plane(), Nx, Nyh}, (long)nplanes * plane(), 1, 1.0, red_out.as<double>());
    fetch(out_host, red_out.as<double>(), 1);
  }
  // logdet / tr of Diagonal(field): complex Fourier field in F layout, or real map
  void logdet_Fc(const cx<T>* d, int P, int B, double* out_host) {
    reduce_dev(TermLogdetFc<T>{d, lam.as<T>(), (long)P * plane(), Nx, Nyh}, (long)P * plane(), B, 1.0, red_out.as<double>());
    fetch(out_host, red_out.as<double>(), B);
  }
  void tr_Fc(const cx<T>* d, int P, int B, double* out_host) {
    reduce_dev(TermTrFc<T>{d, lam.as<T>(), (long)P * plane(), Nx, Nyh}, (long)P * plane(), B, 1.0, red_out.as<double>());
    fetch(out_host, red_out.as<double>(), B);
  }
  void tr_map(const T* d, int P, int B, double* out_host) {
    reduce_dev(TermTrMap<T>{d, (long)P * npix()}, (long)P * npix(), B, 1.0, red_out.as<double>());
    fetch(out_host, red_out.as<double>(), B);
  }
  void logdet_map(const T* d, int P, int B, double* out_host) {
    const long n = (long)P * npix();
    reduce_dev(TermLogAbsMap<T>{d, n}, n, B, 1.0, red_out.as<double>());
    CMBL_LAUNCH(this, K_REDUCE, (k_sign_map<T>), dim3(RED_BLOCKS, B), 0, stream, d, red_part.as<unsigned long long>(), n);
    CMBL_LAUNCH(this, K_REDUCE, k_sign_final, dim3(B), 0, stream, red_part.as<unsigned long long>(), red_out.as<double>(), RED_BLOCKS);
    fetch(out_host, red_out.as<double>(), B);
  }

  // ---- quadratic-estimate / line-search helpers --------------------------------------------------
  // QE_leg (src/quadratic_estimate.jl:89-91): Fourier S0 field (reference layout) -> map
  void qe_leg(const cx<T>* in_ref, T* out_map, int n, int p1, int p2, int B) {
    tmpA.ensure(sizeof(cx<T>) * B * plane());
    cx<T>* F = tmpA.as<cx<T>>();
    ref2F(in_ref, F, B);
    CMBL_LAUNCH(this, K_HARM, (k_qe_leg<T>), dim3((unsigned)((plane() + NTP - 1) / NTP)), 0, stream, F, F, lx_r.as<T>(), ly.as<T>(), Nx, plane(), B, n, p1, p2, 0);
    F_to_map(F, out_map, B);
  }
  // (i lx)^p1 (i ly)^p2 * rfft2(map)  (or its modulus, stored in the real part) -> Fourier reference layout
  void fourier_lmul(const T* in_map, cx<T>* out_ref, int p1, int p2, bool take_abs, int B) {
    tmpA.ensure(sizeof(cx<T>) * B * plane());
    cx<T>* F = tmpA.as<cx<T>>();
    rfft2_F(in_map, F, B);
    CMBL_LAUNCH(this, K_HARM, (k_qe_leg<T>), dim3((unsigned)((plane() + NTP - 1) / NTP)), 0, stream, F, F, lx_r.as<T>(), ly.as<T>(), Nx, plane(), B, 0, p1, p2, take_abs ? 1 : 0);
    F2ref(F, out_ref, B);
  }
  void map_fma(T* out, const T* a, const T* b, double scale, bool accumulate, long n) {
    const unsigned gx = (unsigned)std::min<long>((n + NTP - 1) / NTP, 4096);
    CMBL_LAUNCH(this, K_LINCOMB, (k_map_fma<T>), dim3(gx), 0, stream, out, a, b, (T)scale, accumulate ? 1 : 0, n);
  }

  // white noise maps: slot b of `out` (n_per_slot reals) is stream `stream` of generator seeds[b]
  void randn(T* out, const uint64_t* seeds, int nslots, uint64_t strm, long n_per_slot) {
    const unsigned gx = (unsigned)std::min<long>(((n_per_slot + 3) / 4 + NTP - 1) / NTP, 8192);
    for (int b = 0; b < nslots; ++b)
      CMBL_LAUNCH(this, K_LINCOMB, (k_randn<T>), dim3(gx), 0, stream, out + (long)b * n_per_slot, n_per_slot, seeds[b], strm);
  }

  // ---- basis conversion between reference-layout arrays and the internal F layout ----------------
  // to F: out_F is (P*B*plane) complex in `want` (B_FOURIER = QU Fourier, B_HARMONIC = EB Fourier)
  void to_F(int basis_in, const void* in, cx<T>* out_F, int want, int P, int B) {
    const long slices = (long)P * B;
    if (basis_in == B_MAP) rfft2_F((const T*)in, out_F, slices);
    else ref2F((const cx<T>*)in, out_F, slices);
    const bool have_h = (basis_in == B_HARMONIC), want_h = (want == B_HARMONIC);
    if (P >= 2 && have_h != want_h) harm(out_F, out_F, P, B, 0, nullptr, false, !have_h, !want_h);
  }
  // from F (in `have` basis) to a reference-layout array; clobbers in_F
  void from_F(cx<T>* in_F, int have, int basis_out, void* out, int P, int B) {
    const long slices = (long)P * B;
    const bool have_h = (have == B_HARMONIC), want_h = (basis_out == B_HARMONIC);
    if (P >= 2 && have_h != want_h) harm(in_F, in_F, P, B, 0, nullptr, false, !have_h, !want_h);
    if (basis_out == B_MAP) F_to_map(in_F, (T*)out, slices);
    else F2ref(in_F, (cx<T>*)out, slices);
  }
};

// =================================================================================================
// LenseFlow operator  (src/lenseflow.jl, src/flowops.jl)
template <typename T>
struct Flow {
  Ctx<T>* c;
  int n;                                  // RK4 steps (src/lenseflow.jl:29 default 7)
  int Bphi = 0;
  DevBuf phimaps;                         // [5][Bphi][Nx][Ny] : gx, gy, Hxx, Hyx, Hyy
  DevBuf pcache;                          // [2n+1][2][Bphi][Nx][Ny] : p(t_k)
  bool use_pcache = false;
  DevBuf phiF, gh;                        // scratch for precompute
  DevBuf A, A2, A3, Gx, y0, acc;          // forward flow state          (slices); A3: second primary y-transform buffer (see flow_map)
  DevBuf H, Wx, Wy, Y0, Yacc;             // adjoint flow state          (slices)
  DevBuf P0;                              // delta flow: dphi result (F layout)
  DevBuf cvt;                             // boundary conversion scratch
  DevBuf mls_p, mls_e, mls_f;             // max_lensing_step scratch

  DevBuf Wst, U5, F5, tcbuf;              // per-stage partial products, the five reduced maps and their transforms, (t_s, c_s)
  std::vector<T> tc_host;

  // Slice groups.  At B = 1 a launch over the P pol slices is a single residency wave of workgroups, so it lasts as long as one
  // workgroup's dependent chain.  The slices' chains are independent until the flow ends, so each slice runs as its own launch
  // chain on its own stream (slice 0 on the context's stream): the chains drift apart and one's HBM burst overlaps the other's
  // transforms (measured: L*f 0.82 -> 0.72 ms at 1024^2 QU).  One fork and one join event per flow, none per stage.
  static constexpr int MAXG = 4;
  hipStream_t sub[MAXG - 1] = {nullptr, nullptr, nullptr};
  hipEvent_t evFork = nullptr, evJoin[MAXG - 1] = {nullptr, nullptr, nullptr};
  int max_groups = 1;

  Flow(Ctx<T>* ctx, int nsteps) : c(ctx), n(nsteps) {
    CMBL_REQUIRE(nsteps >= 1 && nsteps <= 512, ERR_ARG, "nsteps out of range");
    max_groups = MAXG;
    if (max_groups > 1) {
      CMBL_HIP(hipEventCreateWithFlags(&evFork, hipEventDisableTiming));
      for (int i = 0; i < max_groups - 1; ++i) {
        CMBL_HIP(hipStreamCreateWithFlags(&sub[i], hipStreamNonBlocking));
        CMBL_HIP(hipEventCreateWithFlags(&evJoin[i], hipEventDisableTiming));
      }
    }
  }
  ~Flow() {
    for (int i = 0; i < MAXG - 1; ++i) {
      if (sub[i]) { (void)hipStreamSynchronize(sub[i]); (void)hipStreamDestroy(sub[i]); }
      if (evJoin[i]) (void)hipEventDestroy(evJoin[i]);
    }
    if (evFork) (void)hipEventDestroy(evFork);
  }
  // one slice per group; the `slice_streams` option (cmbl_ctx_set_option) lets a profiler switch the splitting off (bench.py roofline leg)
  // Only where a launch is long enough (>= ~10 us: 2^20 pixels) for the host to keep several chains fed -- a launch costs the host
  // ~4 us, and at 512^2 the kernels last 7 us, so splitting there makes the flow host-bound (measured: 512^2 L*f 0.41 -> 0.52 ms).
  // B = 1: one pol slice per group.  B > 1: groups of whole batch slots (K = the largest divisor of B that fits), so that a group's
  // phi slots are contiguous (phi_off) -- fewer, larger launches per chain, still several chains in flight.
  // A chain must carry 2 x slice_streams_min_pix pixels (2^20: 1024^2 QU / T+QU as before), the two chains of a delta flow half of that -- round 6:
  // counted per CHAIN, not per slice, so that
  // batches of small maps and wide patches get their chains too.  grad lnP with / without (profiles/r06_ab_pow2_streams_retuned.txt): B = 1 512 x 1024
  // QU (524 k per chain) -7 %, 512^2 QU (262 k) +6 %, 512^2 T+QU (three chains of 262 k) +26 %; two chains of batch slots: 256^2 QU B = 8 (524 k)
  // -12 %, 512^2 QU B = 4 -13 %, B = 8 -8 %, 512^2 T+QU B = 2 -9 %, 512^2 QU B = 2 (524 k) -0.5 %, 128^2 QU B = 16 (262 k) +5 %.
  // The launches of a delta flow last ~1.6 x those of a map or adjoint flow, and the host has to keep every chain fed (~4 us per launch): the map
  // and adjoint flows need twice the pixels (512^2 QU B = 2, 524 k per chain: (grad L)' -6 %, L*f +13 %; 512 x 1024: -6 %, L*f -4...+13 % by box).
  int groups(int P, int B, bool delta = false) const {
    const int cap = std::min(max_groups, c->opts.slice_streams);
    auto pays = [&](long slices_per_chain, int k) { return c->npix() * slices_per_chain >= (long)c->opts.slice_streams_min_pix * ((delta && k < 3) ? 1 : 2); };
    if (B == 1) { if (Bphi == 1) for (int k = std::min(P, cap); k > 1; --k) if (P % k == 0 && pays(P / k, k)) return k; return 1; }
    // measured at 1024^2 QU: B = 2 -> 2 chains +6 %; B = 4: 2 chains 262 evaluations/s, 4 chains 240, 1 chain 247; B = 8: 266 vs 258
    for (int k = std::min(cap, B >= 4 ? 2 : B); k > 1; --k) if (B % k == 0 && pays((long)P * (B / k), k)) return k;
    return 1;
  }
  // batch-slot offset of group g's phi maps (0 when one phi is shared by all slots)
  long phi_off(int g, int K, int B) const { return (B > 1 && Bphi > 1) ? (long)g * (B / K) : 0; }
  hipStream_t gstream(int g) const { return g == 0 ? c->stream : sub[g - 1]; }
  void fork(int K) {
    if (K <= 1) return;
    CMBL_HIP(hipEventRecord(evFork, c->stream));
    for (int g = 1; g < K; ++g) CMBL_HIP(hipStreamWaitEvent(sub[g - 1], evFork, 0));
  }
  void join(int K) {
    for (int g = 1; g < K; ++g) {
      CMBL_HIP(hipEventRecord(evJoin[g - 1], sub[g - 1]));
      CMBL_HIP(hipStreamWaitEvent(c->stream, evJoin[g - 1], 0));
    }
  }
  Flow(const Flow&) = delete;
  Flow& operator=(const Flow&) = delete;

  // t < 0: no stage time (the five maps only); otherwise p(t) is also taken from the cache when it exists
  PhiMaps<T> ph(double t = -1, long boff = 0) const {
    const size_t s = (size_t)Bphi * c->npix(), o = (size_t)boff * c->npix();
    const T* b = phimaps.as<T>() + o;
    PhiMaps<T> r{b, b + s, b + 2 * s, b + 3 * s, b + 4 * s, Bphi, nullptr, nullptr};
    if (t >= 0 && use_pcache) {
      const int k = (int)std::lround(t * 2 * n);
      r.pcx = pcache.as<T>() + (size_t)(2 * k) * s + o; r.pcy = r.pcx + s;
    }
    return r;
  }

  // precompute! (src/lenseflow.jl:131-142): gradhess(phi) -> five maps; p(t), M^-1(t) are formed on the fly
  void set_phi_F(const cx<T>* phi_F, int nb) {
    Bphi = nb;
    const long pl = c->plane();
    gh.ensure(sizeof(cx<T>) * 5 * nb * pl);
    phimaps.ensure(sizeof(T) * 5 * nb * c->npix());
    // multipliers: out[comp][b][plane]
    CMBL_LAUNCH(c, K_GRADHESS, (k_gradhess_mult<T>), dim3((unsigned)((pl + NTP - 1) / NTP)), 0, c->stream, phi_F, gh.as<cx<T>>(), c->lx_r.template as<T>(),
                c->ly.template as<T>(), c->Nx, c->Nyh, nb);
    c->F_to_map(gh.as<cx<T>>(), phimaps.as<T>(), 5L * nb);
    // p(t) at the 2n+1 stage times (the reference caches p and M^-1, src/lenseflow.jl:45-46,88-90): 2(2n+1) maps per phi slot,
    // 120 MB at 1024^2 fp32 n = 7.  M^-1(t), needed only by the delta-phi kernel, is still formed on the fly.
    const size_t ntot = (size_t)nb * c->npix(), bytes = sizeof(T) * 2 * (2 * n + 1) * ntot;
    use_pcache = c->opts.pcache && bytes <= ((size_t)c->opts.pcache_max_mb << 20);
    if (use_pcache) {
      pcache.ensure(bytes);
      const unsigned gx = (unsigned)std::min<size_t>((ntot + NTP - 1) / NTP, 16384);
      CMBL_LAUNCH(c, K_GRADHESS, (k_pcache<T>), dim3(gx), 0, c->stream, ph(), pcache.as<T>(), (long)ntot, 2 * n);
    }
  }
  void set_phi(int basis, const void* phi, int nb) {
    CMBL_REQUIRE(basis == B_MAP || basis == B_FOURIER || basis == B_HARMONIC, ERR_ARG, "bad basis");
    phiF.ensure(sizeof(cx<T>) * nb * c->plane());
    c->to_F(basis, phi, phiF.as<cx<T>>(), B_FOURIER, 1, nb);
    set_phi_F(phiF.as<cx<T>>(), nb);
  }

  // gradhess maps [5][nb][npix] of an S0 field given in F layout (scratch: gh)
  void gradhess_maps(const cx<T>* phi_F, T* maps, int nb) {
    const long pl = c->plane();
    gh.ensure(sizeof(cx<T>) * 5 * nb * pl);
    CMBL_LAUNCH(c, K_GRADHESS, (k_gradhess_mult<T>), dim3((unsigned)((pl + NTP - 1) / NTP)), 0, c->stream, phi_F, gh.as<cx<T>>(), c->lx_r.template as<T>(),
                c->ly.template as<T>(), c->Nx, c->Nyh, nb);
    c->F_to_map(gh.as<cx<T>>(), maps, 5L * nb);
  }
  // get_max_lensing_step (src/lenseflow.jl:242-256); does not touch the flow's own phi cache
  void max_lensing_step(int basis, const void* phi, const void* eta, int nb, double* out_host) {
    CMBL_REQUIRE(nb <= MAXBATCH, ERR_ARG, "nbatch > 256 not supported in reductions");
    const long np = c->npix();
    DevBuf &mp = mls_p, &me = mls_e, &pf = mls_f;                           // pooled: called once per line-search evaluation
    mp.ensure(sizeof(T) * 5 * nb * np); me.ensure(sizeof(T) * 5 * nb * np); pf.ensure(sizeof(cx<T>) * nb * c->plane());
    c->to_F(basis, phi, pf.as<cx<T>>(), B_FOURIER, 1, nb); gradhess_maps(pf.as<cx<T>>(), mp.as<T>(), nb);
    c->to_F(basis, eta, pf.as<cx<T>>(), B_FOURIER, 1, nb); gradhess_maps(pf.as<cx<T>>(), me.as<T>(), nb);
    CMBL_LAUNCH(c, K_REDUCE, (k_max_step<T>), dim3(Ctx<T>::RED_BLOCKS, nb), 0, c->stream, mp.as<T>(), me.as<T>(), c->red_part.template as<double>(), np, (long)nb * np);
    CMBL_LAUNCH(c, K_REDUCE, k_min_final, dim3(nb), 0, c->stream, c->red_part.template as<double>(), c->red_out.template as<double>(), Ctx<T>::RED_BLOCKS);
    CMBL_HIP(hipMemcpyAsync(out_host, c->red_out.p, sizeof(double) * nb, hipMemcpyDeviceToHost, c->stream));
    CMBL_HIP(hipStreamSynchronize(c->stream));
  }

  RKCoef<T> coef(int step, int stage, double t0, double h, bool last) const {
    // stage times t, t+h/2, t+h/2, t+h on the 2n+1 grid (src/numerical_algorithms.jl:15-19, src/lenseflow.jl:78)
    const double ts = t0 + step * h + (stage == 1 ? 0 : (stage == 4 ? h : h / 2));
    const int kidx = (int)std::lround(ts * 2 * n);
    RKCoef<T> r;
    r.t = (T)kidx / (T)(2 * n);
    r.cnext = (T)(stage <= 2 ? h / 2 : h);
    r.h6 = (T)(h / 6);
    r.stage = stage; r.last = last ? 1 : 0;
    return r;
  }

  // Small maps: the whole flow in one launch, one workgroup per slice with the half plane resident in LDS (kernels_small.hpp; defined in
  // engine_small.hpp, compiled by tu_small_*.hip).  small_ok: the shape has such a kernel in this precision and p(t) is cached.
  bool small_ok() const;
  void small_flow_map(const T* in, T* out, int P, int B, bool inverse);
  void small_flow_adj(const cx<T>* in, cx<T>* out, int P, int B, bool inverse);
  bool small_delta_ok() const;
  void small_flow_delta(T* f, cx<T>* df, cx<T>* dphi, int P, int B, bool forward_primal, bool alias_quirk, const DphiTail<T>* tail);

  void check_ready(int B) const {
    CMBL_REQUIRE(Bphi >= 1, ERR_STATE, "cmbl_lenseflow_set_phi has not been called");
    CMBL_REQUIRE(Bphi == 1 || Bphi == B, ERR_SHAPE, "nbatch of phi must be 1 or equal to nbatch of f");
  }


  // ---- any-size path: the reference's pass structure on k_gen_dft + pointwise kernels (kernels_generic.hpp) ----------------------
  // (x, y) pairs live in the two halves of ONE buffer so that both members go through a transform in one launch
  DevBuf gF, gFxy, gmxy, gms, gYs, gLdf, gWxy;
  // (pointwise launches of a stage: over the slice window when one is set)
  long wsl(long slices) const { return c->gwn >= 0 ? c->gwn : slices; }
  int wsl0() const { return c->gwn >= 0 ? (int)c->gw0 : 0; }
  dim3 pgrid(long n, long slices) const { return dim3((unsigned)std::min<long>((n + NTP - 1) / NTP, 4096), (unsigned)wsl(slices)); }
  dim3 fgrid(long slices) const { return dim3((unsigned)((c->plane() + NTP - 1) / NTP), (unsigned)wsl(slices)); }
  // One launch chain per group of slices for the flows of the any-size path, like the fused flows (groups()): a launch there is one
  // residency round of workgroups that all fetch, then all transform, then all store (k_ct_dft stamps: each phase alone runs at
  // 3 - 5 TB/s and the memory system idles during the transforms), so two chains side by side fill each other's gaps.
  // Measured (profiles/r05_anysize_times.txt): 1536^2 QU -17 %, 768^2 and 1000^2 T+QU -5 .. -7 %, 1000^2 QU +2 %, 768^2 QU +5 %, 360^2 +4 %,
  // 96 x 160 +20 % -- below ~2^21 pixels per launch the kernels last 6 - 9 us and the host (~5 us per launch) cannot feed two chains, so the
  // chains are split from 2^21 pixels on, three or more slices from 2^19.
  // Launch chains of an any-size flow: the largest divisor K of the slice count (up to max_groups / option slice_streams) whose chains each carry at
  // least gen_streams_min_pix pixels (in units of 4-byte pixels: a double-precision map counts twice), a third more for three or more chains.
  // Measured with the tiled kernels (profiles/r06_ab_anysize_streams_retuned.txt; grad lnP with / without chains): QU 480^2 (230 k pixels per chain)
  // +1.5 %, 480 x 640 (307 k) -3...-8 %, 640^2 -12 %, 720^2 -12 %; T+QU (three chains) 360^2 +7 %, 384^2 +9 %, 480 x 640 (307 k) +9.5 %, 640^2
  // (410 k) -7 %; QU at B = 8 (four chains of four slices) 192^2 (147 k per chain) +68 %, 360^2 (518 k) -15 %, 480^2 -9 %; double precision 480^2
  // -1.5 %, 640^2 -21 %.  (Round 5's rule -- pixels of ONE slice, x 4 from three slices on -- switched the chains on for T+QU from 131 k pixels.)
  // (One rule for all flows: chains in the delta flows alone -- their launches are longer -- make (grad L)' faster from 200 k pixels per chain, 480^2 QU
  //  -2.4 %, 480 x 640 T+QU -9 %, but the grad lnP around it SLOWER, +3.4 % / +7 %: profiles/r06_ab_anysize_streams_retuned.txt, last block.)
  int gen_groups(long slices) const {
    if (!c->opts.gen_slice_streams || !gen_sep()) return 1;
    for (int k = (int)std::min<long>(std::min(max_groups, c->opts.slice_streams), slices); k > 1; --k) {
      if (slices % k) continue;
      const long per_chain = c->npix() * (slices / k) * (long)(sizeof(T) / 4), need = (long)c->opts.gen_streams_min_pix * (k >= 3 ? 4 : 3) / 3;
      if (per_chain >= need) return k;
    }
    return 1;
  }
  struct GenWindow {                                                    // RAII: the launches of group g of K go to its stream
    Ctx<T>* c; hipStream_t main;
    GenWindow(Flow* f, int g, int K, long slices) : c(f->c), main(f->c->stream) {
      if (K <= 1) return;
      c->gwall = slices; c->gwn = slices / K; c->gw0 = g * (slices / K);
      c->stream = g == 0 ? main : f->sub[g - 1];
    }
    ~GenWindow() { c->stream = main; c->gwn = -1; c->gw0 = 0; c->gwall = 0; }
  };
  // (gmx, gmy) = grad of the stage input f_s from its y transform A = rfft_y(f_s) (mixed layout), using the separability the fused
  // kernels use: d/dy needs the y transform alone (the i*ly multiply commutes with the x transforms, which then cancel), d/dx one
  // forward / i*lx / inverse x pass; ONE complex inverse y transform returns both real maps (pair c2r).  Equal to the reference's
  // rfft2 -> (i lx, i ly) -> 2 x irfft2 (src/lenseflow.jl:155-157) incl. FFTW's c2r rule at ky = 0 / Nyquist, in 3 launches and
  // 3 slice-passes instead of 5 launches and 7.5.  CMBL_GEN_SEPARABLE=0: the reference's own pass structure (kept for A/B).
  DevBuf gA, gGx, gT, gW2;
  // layout of the hand-off arrays for the launches of one flow (Ctx::htile): tiled when the flow runs its fused stages and Ctx::gen_tile() allows
  struct HandOff {
    Ctx<T>* c;
    HandOff(Ctx<T>* c_, bool fused, int kind /*option gen_tiled: 1 map flows, 2 adjoint flows, 4 delta flows*/) : c(c_) { c->htile = fused && (c->opts.gen_tiled & kind) ? c->gen_tile() : 0; }
    ~HandOff() { c->htile = 0; }
  };
  bool gen_sep() const { return c->opts.gen_separable != 0; }
  void gen_grad_sep(const cx<T>* A_, long slices) {
    const long pl = c->plane(), np = c->npix();
    gT.ensure(sizeof(cx<T>) * slices * pl); gGx.ensure(sizeof(cx<T>) * slices * pl); gmxy.ensure(sizeof(T) * 2 * slices * np);
    c->gen_x_deriv(A_, gGx.as<cx<T>>(), gT.as<cx<T>>(), c->lx_r.template as<T>(), slices);     // Nx * d/dx in mixed space: ifft_x(i lx fft_x(A))
    c->gen_y_c2r_pair(gGx.as<cx<T>>(), A_, c->ly.template as<T>(), gmxy.as<T>(), gmxy.as<T>() + slices * np,
                      (T)(1.0 / ((double)c->Ny * c->Nx)), (T)(1.0 / (double)c->Ny), slices);
  }
  // (gmx, gmy) = grad of the map `ys`  (rfft2, i l multiplies, irfft2 of both components)
  void gen_grad(const T* ys, long slices) {
    const long pl = c->plane(), np = c->npix();
    gF.ensure(sizeof(cx<T>) * slices * pl); gFxy.ensure(sizeof(cx<T>) * 2 * slices * pl); gmxy.ensure(sizeof(T) * 2 * slices * np);
    c->rfft2_F(ys, gF.as<cx<T>>(), slices);
    CMBL_LAUNCH(c, K_GEN_POINT, (k_gen_lmul2<T>), fgrid(slices), 0, c->stream, gF.as<cx<T>>(), gFxy.as<cx<T>>(), gFxy.as<cx<T>>() + slices * pl,
                c->lx_r.template as<T>(), c->ly.template as<T>(), c->Nx, pl);
    c->F_to_map(gFxy.as<cx<T>>(), gmxy.as<T>(), 2 * slices);
  }
  // d(Fourier state)/dt from the maps (Wx, Wy) = the halves of Wxy: rfft2 of both + the RK update with k = i lx Fx + i ly Fy
  // (separable form: the two real maps go through ONE complex y transform, then one x launch over both)
  bool gen_pro() const { return c->opts.gen_prologue != 0; }
  void gen_adj_update(const T* Wxy, cx<T>* Y0, cx<T>* Yacc_, cx<T>* Ys, const RKCoef<T>& rk, long slices, const GenPro<T>* pro = nullptr) {
    const long pl = c->plane(), np = c->npix();
    gFxy.ensure(sizeof(cx<T>) * 2 * slices * pl);
    if (gen_sep()) {
      gW2.ensure(sizeof(cx<T>) * 2 * slices * pl);
      // pro (mode 3): the pair is formed in the fetch from L(df) and p; `in` / `in2` only mark the launch as a real pair
      if (pro) c->gen_y_r2c(pro->Ldf, gW2.as<cx<T>>(), slices, pro->Ldf, gW2.as<cx<T>>() + slices * pl, pro);
      else c->gen_y_r2c(Wxy, gW2.as<cx<T>>(), slices, Wxy + slices * np, gW2.as<cx<T>>() + slices * pl);
      c->gen_x(gW2.as<cx<T>>(), gFxy.as<cx<T>>(), false, nullptr, 2 * slices, true, false);
    } else c->rfft2_F(Wxy, gFxy.as<cx<T>>(), 2 * slices);
    CMBL_LAUNCH(c, K_GEN_POINT, (k_gen_adj_rk<T>), fgrid(slices), 0, c->stream, gFxy.as<cx<T>>(), gFxy.as<cx<T>>() + slices * pl, c->lx_r.template as<T>(),
                c->ly.template as<T>(), c->Nx, Y0, Yacc_, Ys, rk, pl, wsl0());
  }
  void gen_flow_map(const T* in, T* out, int P, int B, bool inverse) {
    const long slices = (long)P * B, np = c->npix();
    acc.ensure(sizeof(T) * slices * np); gms.ensure(sizeof(T) * slices * np);
    if (in != out) CMBL_HIP(hipMemcpyAsync(out, in, sizeof(T) * slices * np, hipMemcpyDeviceToDevice, c->stream));
    CMBL_HIP(hipMemcpyAsync(gms.p, out, sizeof(T) * slices * np, hipMemcpyDeviceToDevice, c->stream));
    const double t0 = inverse ? 1.0 : 0.0, h = (inverse ? -1.0 : 1.0) / n;
    const bool sep = gen_sep();
    const bool yy = sep && gen_pro() && c->opts.gen_yy && c->opts.gen_xderiv_fused && c->gen_ct_y();
    HandOff ho(c, yy, 1);
    const long pl = c->hplane();
    if (sep) { gA.ensure(sizeof(cx<T>) * slices * pl); c->gen_y_r2c(gms.as<T>(), gA.as<cx<T>>(), slices); }
    const int K = gen_groups(slices);
    // every scratch buffer a stage can grow is sized BEFORE the chains fork: a growth (hipFree / hipMalloc) inside the multi-stream region
    // would pull a buffer from under a launch of another chain (the separate-launch path used to ensure inside gen_grad_sep / gen_grad)
    if (sep) { gT.ensure(sizeof(cx<T>) * slices * pl); gGx.ensure(sizeof(cx<T>) * slices * pl); if (!yy) gmxy.ensure(sizeof(T) * 2 * slices * np); }
    else { gF.ensure(sizeof(cx<T>) * slices * pl); gFxy.ensure(sizeof(cx<T>) * 2 * slices * pl); gmxy.ensure(sizeof(T) * 2 * slices * np); c->mixed_scratch(2 * slices); }
    fork(K);
    for (int step = 0; step < n; ++step)
      for (int stage = 1; stage <= 4; ++stage) {
        const bool last = step == n - 1 && stage == 4;
        const RKCoef<T> rk = coef(step, stage, t0, h, last);
        for (int g = 0; g < K; ++g) {
          GenWindow w(this, g, K, slices);
          if (yy) {                                                          // d/dx pass, then every y pass of the stage in one launch
            GenPro<T> e{};
            e.mode = 1; e.ph = ph(rk.t); e.rk = rk; e.y0 = out; e.acc = acc.as<T>(); e.npix = np; e.P = P;
            c->gen_x_deriv(gA.as<cx<T>>(), gGx.as<cx<T>>(), gT.as<cx<T>>(), c->lx_r.template as<T>(), slices);
            c->gen_y_flow_stage(gGx.as<cx<T>>(), gA.as<cx<T>>(), c->ly.template as<T>(), (T)(1.0 / ((double)c->Ny * c->Nx)), (T)(1.0 / (double)c->Ny), e,
                                gA.as<cx<T>>(), last, slices);
            continue;
          }
          if (sep) gen_grad_sep(gA.as<cx<T>>(), slices); else gen_grad(gms.as<T>(), slices);
          if (sep && !last && gen_pro()) {                                   // velocity + RK bookkeeping in the fetch of the next stage's y transform
            GenPro<T> e{};
            e.mode = 1; e.ph = ph(rk.t); e.rk = rk; e.gx = gmxy.as<T>(); e.gy = gmxy.as<T>() + slices * np; e.y0 = out; e.acc = acc.as<T>(); e.npix = np; e.P = P;
            c->gen_y_r2c(gms.as<T>(), gA.as<cx<T>>(), slices, nullptr, nullptr, &e);
            continue;
          }
          CMBL_LAUNCH(c, K_GEN_POINT, (k_gen_vel_rk<T>), pgrid(np, slices), 0, c->stream, gmxy.as<T>(), gmxy.as<T>() + slices * np, ph(rk.t), out, acc.as<T>(),
                      gms.as<T>(), rk, np, P, wsl0());
          if (sep && !last) c->gen_y_r2c(gms.as<T>(), gA.as<cx<T>>(), slices);
        }
      }
    join(K);
  }
  void gen_flow_adj_F(const cx<T>* in, cx<T>* out, int P, int B, bool inverse) {
    const long slices = (long)P * B, pl = c->plane(), np = c->npix();
    Yacc.ensure(sizeof(cx<T>) * slices * pl); gYs.ensure(sizeof(cx<T>) * slices * pl);
    gms.ensure(sizeof(T) * slices * np); gmxy.ensure(sizeof(T) * 2 * slices * np);
    if (in != out) CMBL_HIP(hipMemcpyAsync(out, in, sizeof(cx<T>) * slices * pl, hipMemcpyDeviceToDevice, c->stream));
    CMBL_HIP(hipMemcpyAsync(gYs.p, out, sizeof(cx<T>) * slices * pl, hipMemcpyDeviceToDevice, c->stream));
    const double t0 = inverse ? 0.0 : 1.0, h = (inverse ? 1.0 : -1.0) / n;
    const bool yy = gen_sep() && c->opts.gen_yy && c->gen_ct_y();
    HandOff ho(c, yy, 2);
    const long hpl = c->hplane();                                          // (the pair W2 and the mixed scratch t3 are hand-off arrays; the Fourier state is not)
    gFxy.ensure(sizeof(cx<T>) * 2 * slices * pl); gW2.ensure(sizeof(cx<T>) * 2 * slices * hpl); (void)c->mixed_scratch(slices);   // before the chains fork
    const int K = gen_groups(slices);
    fork(K);
    for (int step = 0; step < n; ++step)
      for (int stage = 1; stage <= 4; ++stage) {
        const RKCoef<T> rk = coef(step, stage, t0, h, step == n - 1 && stage == 4);
        for (int g = 0; g < K; ++g) {
          GenWindow w(this, g, K, slices);
          if (yy) {                                                          // ifft_x, every y pass of the stage in one launch, fft_x of the pair, RK update
            cx<T>* t3 = c->mixed_scratch(slices);
            const bool mrg = c->gen_ct_x();                                  // the row update of stage s also writes t3 = ifft_x(Ys) of stage s + 1
            if (!mrg || (step == 0 && stage == 1)) c->gen_x(gYs.as<cx<T>>(), t3, true, nullptr, slices, false, true);
            c->gen_y_adj_stage(t3, (T)(1.0 / ((double)c->Ny * c->Nx)), ph(rk.t), rk.t, P, gW2.as<cx<T>>(), gW2.as<cx<T>>() + slices * hpl, slices);
            if (mrg && !rk.last) { c->gen_x_adj_next(gW2.as<cx<T>>(), gW2.as<cx<T>>() + slices * hpl, out, Yacc.as<cx<T>>(), rk, t3, nullptr, nullptr, slices); continue; }
            if (c->gen_x_adj_update(gW2.as<cx<T>>(), gW2.as<cx<T>>() + slices * hpl, out, Yacc.as<cx<T>>(), gYs.as<cx<T>>(), rk, slices)) continue;
            c->gen_x(gW2.as<cx<T>>(), gFxy.as<cx<T>>(), false, nullptr, 2 * slices, true, false);
            CMBL_LAUNCH(c, K_GEN_POINT, (k_gen_adj_rk<T>), fgrid(slices), 0, c->stream, gFxy.as<cx<T>>(), gFxy.as<cx<T>>() + slices * pl, c->lx_r.template as<T>(),
                        c->ly.template as<T>(), c->Nx, out, Yacc.as<cx<T>>(), gYs.as<cx<T>>(), rk, pl, wsl0());
            continue;
          }
          c->F_to_map(gYs.as<cx<T>>(), gms.as<T>(), slices);
          CMBL_LAUNCH(c, K_GEN_POINT, (k_gen_pmul<T>), pgrid(np, slices), 0, c->stream, gms.as<T>(), ph(rk.t), rk.t, gmxy.as<T>(), gmxy.as<T>() + slices * np, np, P, wsl0());
          gen_adj_update(gmxy.as<T>(), out, Yacc.as<cx<T>>(), gYs.as<cx<T>>(), rk, slices);
        }
      }
    join(K);
  }
  void gen_flow_delta(T* f, cx<T>* df, cx<T>* dphi, int P, int B, bool forward_primal, bool alias_quirk) {
    const long slices = (long)P * B, pl = c->plane(), np = c->npix();
    const int nst = 4 * n;
    acc.ensure(sizeof(T) * slices * np); gms.ensure(sizeof(T) * slices * np); gLdf.ensure(sizeof(T) * slices * np);
    gWxy.ensure(sizeof(T) * 2 * slices * np);
    Yacc.ensure(sizeof(cx<T>) * slices * pl); gYs.ensure(sizeof(cx<T>) * slices * pl);
    Wst.ensure(sizeof(T) * (size_t)nst * 2 * slices * np);
    U5.ensure(sizeof(T) * 5 * B * np); F5.ensure(sizeof(cx<T>) * 5 * B * pl); tcbuf.ensure(sizeof(T) * 2 * nst);
    CMBL_HIP(hipMemcpyAsync(gms.p, f, sizeof(T) * slices * np, hipMemcpyDeviceToDevice, c->stream));
    CMBL_HIP(hipMemcpyAsync(gYs.p, df, sizeof(cx<T>) * slices * pl, hipMemcpyDeviceToDevice, c->stream));
    const double t0 = forward_primal ? 1.0 : 0.0, h = (forward_primal ? -1.0 : 1.0) / n;
    const bool sep = gen_sep();
    const bool yy = sep && gen_pro() && c->opts.gen_yy && c->opts.gen_xderiv_fused && c->gen_ct_y2() && c->genX.plan.nf > 0;
    HandOff ho(c, yy, 4);
    const long hpl = c->hplane();                                          // (A, Gx, the pair W2 and the mixed scratch t3 are hand-off arrays; the Fourier state is not)
    if (sep) { gA.ensure(sizeof(cx<T>) * slices * hpl); c->gen_y_r2c(gms.as<T>(), gA.as<cx<T>>(), slices); }
    tc_host.resize(2 * (size_t)nst);
    gT.ensure(sizeof(cx<T>) * slices * hpl); gGx.ensure(sizeof(cx<T>) * slices * hpl); gmxy.ensure(sizeof(T) * 2 * slices * np);   // before the chains fork
    gFxy.ensure(sizeof(cx<T>) * 2 * slices * pl); gW2.ensure(sizeof(cx<T>) * 2 * slices * hpl); (void)c->mixed_scratch(slices);
    const int K = gen_groups(slices);
    fork(K);
    int it = 0;
    for (int step = 0; step < n; ++step)
      for (int stage = 1; stage <= 4; ++stage, ++it) {
        const RKCoef<T> rk = coef(step, stage, t0, h, step == n - 1 && stage == 4);
        tc_host[2 * it] = rk.t;
        tc_host[2 * it + 1] = (T)((stage == 1 || stage == 4 ? 1.0 : 2.0) * h / 6);
        T* w1p = Wst.as<T>() + (size_t)(2 * it) * slices * np;
        for (int g = 0; g < K; ++g) {
          GenWindow w(this, g, K, slices);
          if (yy) {
            // x passes (ifft_x of delta f, d/dx of rfft_y(f)), then every y pass of the stage in one launch, then the delta-f velocity's x pass + RK update
            cx<T>* t3 = c->mixed_scratch(slices);
            const bool mrg = c->gen_ct_x();                                  // the row update of stage s also runs both x passes that open stage s + 1
            if (!mrg || it == 0) c->gen_x_inv_and_deriv(gYs.as<cx<T>>(), t3, gA.as<cx<T>>(), gGx.as<cx<T>>(), gT.as<cx<T>>(), c->lx_r.template as<T>(), slices);
            GenPro<T> e{};
            e.mode = 2; e.ph = ph(rk.t); e.rk = rk; e.y0 = f; e.acc = acc.as<T>(); e.w1p = w1p; e.w2p = w1p + (size_t)slices * np; e.npix = np; e.P = P;
            c->gen_y_delta_stage(t3, (T)(1.0 / ((double)c->Ny * c->Nx)), gGx.as<cx<T>>(), gA.as<cx<T>>(), c->ly.template as<T>(),
                                 (T)(1.0 / ((double)c->Ny * c->Nx)), (T)(1.0 / (double)c->Ny), e, gA.as<cx<T>>(), gW2.as<cx<T>>(), gW2.as<cx<T>>() + slices * hpl,
                                 rk.last != 0, slices);
            if (mrg && !rk.last) {
              c->gen_x_adj_next(gW2.as<cx<T>>(), gW2.as<cx<T>>() + slices * hpl, df, Yacc.as<cx<T>>(), rk, t3, gA.as<cx<T>>(), gGx.as<cx<T>>(), slices);
              continue;
            }
            if (c->gen_x_adj_update(gW2.as<cx<T>>(), gW2.as<cx<T>>() + slices * hpl, df, Yacc.as<cx<T>>(), gYs.as<cx<T>>(), rk, slices)) continue;
            c->gen_x(gW2.as<cx<T>>(), gFxy.as<cx<T>>(), false, nullptr, 2 * slices, true, false);
            CMBL_LAUNCH(c, K_GEN_POINT, (k_gen_adj_rk<T>), fgrid(slices), 0, c->stream, gFxy.as<cx<T>>(), gFxy.as<cx<T>>() + slices * pl, c->lx_r.template as<T>(),
                        c->ly.template as<T>(), c->Nx, df, Yacc.as<cx<T>>(), gYs.as<cx<T>>(), rk, pl, wsl0());
            continue;
          }
          c->F_to_map(gYs.as<cx<T>>(), gLdf.as<T>(), slices);                 // L(df)
          if (sep) gen_grad_sep(gA.as<cx<T>>(), slices); else gen_grad(gms.as<T>(), slices);   // grad f -> gmxy
          if (sep && !rk.last && gen_pro()) {
            // the stage's pointwise work rides in the fetches of the two y transforms that consume it: f part + products -> rfft_y(f_{s+1});
            // (p_x L(df), p_y L(df)) -> the pair r2c of the delta-f velocity
            GenPro<T> e{};
            e.ph = ph(rk.t); e.rk = rk; e.gx = gmxy.as<T>(); e.gy = gmxy.as<T>() + slices * np; e.Ldf = gLdf.as<T>(); e.y0 = f; e.acc = acc.as<T>();
            e.w1p = w1p; e.w2p = w1p + (size_t)slices * np; e.npix = np; e.P = P;
            e.mode = 2;
            c->gen_y_r2c(gms.as<T>(), gA.as<cx<T>>(), slices, nullptr, nullptr, &e);
            e.mode = 3;
            gen_adj_update(nullptr, df, Yacc.as<cx<T>>(), gYs.as<cx<T>>(), rk, slices, &e);
            continue;
          }
          CMBL_LAUNCH(c, K_GEN_POINT, (k_gen_delta<T>), pgrid(np, slices), 0, c->stream, gLdf.as<T>(), gmxy.as<T>(), gmxy.as<T>() + slices * np, ph(rk.t),
                      gWxy.as<T>(), gWxy.as<T>() + slices * np, w1p, w1p + (size_t)slices * np, f, acc.as<T>(), gms.as<T>(), rk, np, P, wsl0());
          if (sep && !rk.last) c->gen_y_r2c(gms.as<T>(), gA.as<cx<T>>(), slices);
          gen_adj_update(gWxy.as<T>(), df, Yacc.as<cx<T>>(), gYs.as<cx<T>>(), rk, slices);
        }
      }
    join(K);
    dphi_finish(dphi, P, B, nst, alias_quirk);
  }
  // delta-phi: quadrature over the stored stages, five real transforms, the l-multipliers (shared by both paths)
  void dphi_finish(cx<T>* dphi, int P, int B, int nst, bool alias_quirk, const DphiTail<T>* tail = nullptr) {
    const long pl = c->plane(), np = c->npix();
    TcTab<T> tcv{};
    const T* tcd = nullptr;
    if (nst <= TcTab<T>::MAXST) std::copy(tc_host.begin(), tc_host.begin() + 2 * nst, tcv.v);
    else { CMBL_HIP(hipMemcpyAsync(tcbuf.p, tc_host.data(), sizeof(T) * 2 * nst, hipMemcpyHostToDevice, c->stream)); tcd = tcbuf.as<T>(); }
    constexpr int V = 16 / (int)sizeof(T);
    if (np % V == 0)
      CMBL_LAUNCH(c, K_DPHI_Y, (k_dphi_reduce<T, V>), dim3((unsigned)std::min<long>((np / V + NTP - 1) / NTP, 8192), (unsigned)B), 0, c->stream, ph(), Wst.as<T>(),
                  tcv, tcd, U5.as<T>(), np, P, B, nst, alias_quirk ? 1 : 0);
    else
      CMBL_LAUNCH(c, K_DPHI_Y, (k_dphi_reduce<T, 1>), dim3((unsigned)std::min<long>((np + NTP - 1) / NTP, 8192), (unsigned)B), 0, c->stream, ph(), Wst.as<T>(),
                  tcv, tcd, U5.as<T>(), np, P, B, nst, alias_quirk ? 1 : 0);
    if (c->x_dphi_fits() && c->opts.fused_harm) {
      // the five x transforms and the l-multipliers in one row launch (five row sets in LDS), the caller's tail on top
      cx<T>* m5 = c->mixed_scratch(5L * B);
      c->y_r2c(U5.as<T>(), m5, 5L * B);
      c->x_dphi(m5, dphi, B, tail ? *tail : DphiTail<T>{nullptr, nullptr, nullptr});
      return;
    }
    c->rfft2_F(U5.as<T>(), F5.as<cx<T>>(), 5L * B);
    CMBL_LAUNCH(c, K_DPHI_X, (k_dphi_combine<T>), dim3((unsigned)((pl + NTP - 1) / NTP)), 0, c->stream, F5.as<cx<T>>(), dphi, c->lx_r.template as<T>(),
                c->ly.template as<T>(), c->Nx, pl, B);
    if (tail && tail->Ginv) {                                               // unfused tail: dphi = Ginv * (dphi + add1 - sub1)
      PwPhiGrad<T> g{pl, tail->Ginv, dphi, tail->add1, tail->sub1, dphi};
      c->pw_flat(g, B);
    }
  }

  // L*f (inverse=false) or L\f (inverse=true) on maps; out may alias in.
  // abuf: the primary y-transform buffer of this flow (default A; the ping-pong partner is always A2).  a_ready: abuf already holds
  // rfft_y(in) (written by a fused row pass), so the initial y pass is skipped.  emit_last: the last stage also leaves rfft_y(out) --
  // in abuf again, the number of stages being even -- for a caller that continues in Fourier space.
  cx<T>* abuf_ensure(DevBuf& b, long slices) { b.ensure(sizeof(cx<T>) * slices * c->mplane()); return b.template as<cx<T>>(); }
  void flow_map(const T* in, T* out, int P, int B, bool inverse, bool a_ready = false, bool emit_last = false, DevBuf* abuf = nullptr) {
    check_ready(B);
    if (c->generic) return gen_flow_map(in, out, P, B, inverse);
    if (small_ok()) {                                                      // (a_ready: `in` itself is still the state; emit_last: one y pass more)
      small_flow_map(in, out, P, B, inverse);
      if (emit_last) c->y_r2c(out, abuf_ensure(abuf ? *abuf : A, (long)P * B), (long)P * B);
      return;
    }
    const long slices = (long)P * B, pl = c->mplane(), np = c->npix();          // A, Gx: mixed layout
    DevBuf& Ab = abuf ? *abuf : A;
    Ab.ensure(sizeof(cx<T>) * slices * pl); A2.ensure(sizeof(cx<T>) * slices * pl); Gx.ensure(sizeof(cx<T>) * slices * pl);
    acc.ensure(sizeof(T) * slices * np);
    T* y = out;
    cx<T>* a_cur = Ab.template as<cx<T>>(); cx<T>* a_nxt = A2.as<cx<T>>();
    if (!a_ready) c->y_r2c(in, a_cur, slices);                             // the first RK step reads the state from `in` (y0r), the rest from `out`
    const int K = groups(P, B);
    const long gs = slices / K;                                            // slices per group
    const auto tile = c->tileY(slices, true);
    const double t0 = inverse ? 1.0 : 0.0, h = (inverse ? -1.0 : 1.0) / n;
    fork(K);
    for (int step = 0; step < n; ++step)
      for (int stage = 1; stage <= 4; ++stage) {
        for (int g = 0; g < K; ++g) {
          hipStream_t st = gstream(g);
          const long so = g * gs;
          c->template x_pass<2>(a_cur + so * pl, Gx.as<cx<T>>() + so * pl, gs, st, slices);
          FlowYArgs<T> a{};
          a.A = a_cur + so * pl; a.Gx = Gx.as<cx<T>>() + so * pl; a.Anext = a_nxt + so * pl; a.y0 = y + so * np; a.acc = acc.as<T>() + so * np;
          a.y0r = (step == 0 ? in : y) + so * np;
          a.twY = c->twY.template as<cx<T>>(); a.ly = c->ly.template as<T>();
          a.Nx = c->Nx; a.P = P; a.emit_last = emit_last ? 1 : 0;
          a.rk = coef(step, stage, t0, h, step == n - 1 && stage == 4);
          a.ph = ph(a.rk.t, phi_off(g, K, B));
          a.pf = c->col_prefetch(c->Nx / tile.C, gs, K);
          c->dispatch_col(tile, [&](auto lgm, auto r, auto nt) {
            constexpr int LGM = decltype(lgm)::value, R = decltype(r)::value, NT = decltype(nt)::value;
            CMBL_LAUNCH_NT(c, K_FLOW_Y, NT, (k_flow_y_fwd<T, R, NT, LGM>), dim3(c->Nx / tile.C, (unsigned)gs), c->ldsY(tile.C), st, a);
          });
        }
        std::swap(a_cur, a_nxt);
      }
    join(K);
  }

  // L'*g (inverse=false, t 1->0) or L'\g (inverse=true, t 0->1); F layout, QU-Fourier basis; out may alias in
  // h_ready: H already holds ifft_x(in) in the mixed layout (written by a fused row pass next to `in` itself)
  cx<T>* hbuf_ensure(long slices) { H.ensure(sizeof(cx<T>) * slices * c->mplane()); return H.as<cx<T>>(); }
  void flow_adj_F(const cx<T>* in, cx<T>* out, int P, int B, bool inverse, bool h_ready = false) {
    check_ready(B);
    if (c->generic) return gen_flow_adj_F(in, out, P, B, inverse);
    if (small_ok()) return small_flow_adj(in, out, P, B, inverse);       // (h_ready: `in` itself is still the state)
    const long slices = (long)P * B, pl = c->plane(), mpl = c->mplane();
    H.ensure(sizeof(cx<T>) * slices * mpl); Wx.ensure(sizeof(cx<T>) * slices * mpl); Wy.ensure(sizeof(cx<T>) * slices * mpl);
    Yacc.ensure(sizeof(cx<T>) * slices * pl);
    if (in != out) CMBL_HIP(hipMemcpyAsync(out, in, sizeof(cx<T>) * slices * pl, hipMemcpyDeviceToDevice, c->stream));
    if (!h_ready) c->template x_pass<1>(out, H.as<cx<T>>(), slices);
    const int K = groups(P, B);
    const long gs = slices / K;
    const auto tile = c->tileY(slices, true);
    const double t0 = inverse ? 0.0 : 1.0, h = (inverse ? 1.0 : -1.0) / n;
    fork(K);
    for (int step = 0; step < n; ++step)
      for (int stage = 1; stage <= 4; ++stage) {
        const RKCoef<T> rk = coef(step, stage, t0, h, step == n - 1 && stage == 4);
        for (int g = 0; g < K; ++g) {
          hipStream_t st = gstream(g);
          const long so = g * gs * pl, som = g * gs * mpl;
          AdjYArgs<T> a{};
          a.H = H.as<cx<T>>() + som; a.Wx = Wx.as<cx<T>>() + som; a.Wy = Wy.as<cx<T>>() + som; a.ph = ph(rk.t, phi_off(g, K, B));
          a.twY = c->twY.template as<cx<T>>(); a.ly = c->ly.template as<T>();
          a.Nx = c->Nx; a.P = P; a.t = rk.t;
          a.pf = c->col_prefetch(c->Nx / tile.C, gs, K);
          c->dispatch_col(tile, [&](auto lgm, auto r, auto nt) {
            constexpr int LGM = decltype(lgm)::value, R = decltype(r)::value, NT = decltype(nt)::value;
            CMBL_LAUNCH_NT(c, K_ADJ_Y, NT, (k_adj_y<T, R, NT, LGM>), dim3(c->Nx / tile.C, (unsigned)gs), c->ldsY(tile.C), st, a);
          });
          AdjXArgs<T> x{};
          x.Wx = a.Wx; x.Wy = a.Wy; x.Y0 = out + so; x.acc = Yacc.as<cx<T>>() + so; x.Hnext = H.as<cx<T>>() + som;
          x.twX = c->twX.template as<cx<T>>(); x.lx_r = c->lx_r.template as<T>(); x.Nyh = c->Nyh; x.rk = rk;
          c->dispatch_row([&](auto lgnx) {
            constexpr int LGNX = decltype(lgnx)::value, RPWMAX = row_rpw<T>(LGNX, 2);
            if constexpr (RPWMAX > 0) {
              c->template dispatch_rpw<RPWMAX, LGNX>(slices, [&](auto rpw_) {
                constexpr int RPW = decltype(rpw_)::value;
                // (lds_apart keeps two FULL-height groups off one CU; shorter groups are meant to share)
                const size_t lds = RPW == RPWMAX ? c->lds_apart(c->ldsX(RPW, 2), c->row_groups(gs, RPW)) : c->ldsX(RPW, 2);
                CMBL_LAUNCH_NT(c, K_ADJ_X, row_nt(RPW), (k_adj_x<T, LGNX, RPW>), dim3((unsigned)c->row_groups(gs, RPW)), lds, st, x);
              });
            } else fail(ERR_SHAPE, "row tile does not fit LDS (Nx too large for this precision)");
          });
        }
      }
    join(K);
  }

  // delta flow (src/flowops.jl:48,63): state (f [map], df [F, QU-Fourier], dphi [F, S0]); f and df updated in place, dphi written.
  // forward_primal=true : pullback of L*f  -> integrate t 1->0;  false: pullback of L\f -> t 0->1.
  // Two launches per RK stage (k_delta_cols, k_delta_rows) for the (f, delta f) chain; the stage's partial products go to a
  // per-stage buffer and delta-phi -- a pure quadrature over the stages -- is formed once at the end (k_dphi_reduce, 5 rffts, combine).
  // a_ready / abuf: as in flow_map (abuf holds rfft_y(f)); h_ready: H holds ifft_x(df); tail: the last line of the posterior gradient
  // rides on the delta-phi epilogue (dphi_finish)
  void flow_delta(T* f, cx<T>* df, cx<T>* dphi, int P, int B, bool forward_primal, bool alias_quirk, bool a_ready = false, DevBuf* abuf = nullptr,
                  bool h_ready = false, const DphiTail<T>* tail = nullptr) {
    check_ready(B);
    if (c->generic) { CMBL_REQUIRE(!tail, ERR_STATE, "fused tail on the any-size path"); return gen_flow_delta(f, df, dphi, P, B, forward_primal, alias_quirk); }
    if (small_delta_ok()) return small_flow_delta(f, df, dphi, P, B, forward_primal, alias_quirk, tail);   // (a_ready / h_ready: f and df themselves are still the state)
    const long slices = (long)P * B, pl = c->plane(), mpl = c->mplane(), np = c->npix();
    const int nst = 4 * n;
    DevBuf& Ab = abuf ? *abuf : A;
    Ab.ensure(sizeof(cx<T>) * slices * mpl); A2.ensure(sizeof(cx<T>) * slices * mpl); Gx.ensure(sizeof(cx<T>) * slices * mpl);
    acc.ensure(sizeof(T) * slices * np);
    H.ensure(sizeof(cx<T>) * slices * mpl); Wx.ensure(sizeof(cx<T>) * slices * mpl); Wy.ensure(sizeof(cx<T>) * slices * mpl);
    Yacc.ensure(sizeof(cx<T>) * slices * pl);
    Wst.ensure(sizeof(T) * (size_t)nst * 2 * slices * np);              // 4n x 2 maps per slice: 448 MB at 1024^2 QU fp32, n = 7
    U5.ensure(sizeof(T) * 5 * B * np); F5.ensure(sizeof(cx<T>) * 5 * B * pl); tcbuf.ensure(sizeof(T) * 2 * nst);
    cx<T>* a_cur = Ab.template as<cx<T>>(); cx<T>* a_nxt = A2.as<cx<T>>();
    if (!a_ready) c->y_r2c(f, a_cur, slices);
    if (!h_ready) c->template x_pass<1>(df, H.as<cx<T>>(), slices);
    const int K = groups(P, B, true);
    const long gs = slices / K;
    const auto tile = c->tileY_delta(slices);
    const double t0 = forward_primal ? 1.0 : 0.0, h = (forward_primal ? -1.0 : 1.0) / n;
    c->template x_pass<2>(a_cur, Gx.as<cx<T>>(), slices);                // later d/dx passes ride along with the previous stage's row launch
    tc_host.resize(2 * (size_t)nst);
    fork(K);
    int it = 0;
    for (int step = 0; step < n; ++step)
      for (int stage = 1; stage <= 4; ++stage, ++it) {
        const bool last = step == n - 1 && stage == 4;
        const RKCoef<T> rk = coef(step, stage, t0, h, last);
        tc_host[2 * it] = rk.t;
        tc_host[2 * it + 1] = (T)((stage == 1 || stage == 4 ? 1.0 : 2.0) * h / 6);   // RK4 weights (src/numerical_algorithms.jl:20)
        for (int g = 0; g < K; ++g) {
          hipStream_t st = gstream(g);
          const long so = g * gs, sp = so * pl, spm = so * mpl, sm = so * np;
          DeltaYArgs<T> d{};
          FlowYArgs<T>& a = d.f;
          a.A = a_cur + spm; a.Gx = Gx.as<cx<T>>() + spm; a.Anext = a_nxt + spm; a.y0 = f + sm; a.y0r = f + sm; a.acc = acc.as<T>() + sm; a.ph = ph(rk.t, phi_off(g, K, B));
          a.twY = c->twY.template as<cx<T>>(); a.ly = c->ly.template as<T>();
          a.Nx = c->Nx; a.P = P; a.rk = rk;
          d.H = H.as<cx<T>>() + spm; d.Wx = Wx.as<cx<T>>() + spm; d.Wy = Wy.as<cx<T>>() + spm;
          d.w1p = Wst.as<T>() + ((size_t)(2 * it) * slices + so) * np; d.w2p = d.w1p + (size_t)slices * np;
          d.pf = c->col_prefetch(c->Nx / tile.C, gs, K);
          c->dispatch_col(tile, [&](auto lgm, auto r, auto nt) {
            constexpr int LGM = decltype(lgm)::value, R = decltype(r)::value, NT = decltype(nt)::value;
            // one-workgroup-per-CU shapes (2048 rows in double precision) walk several tiles per workgroup with the next tile's loads
            // under the current tile's last phase (delta_y_body_pipelined)
#ifdef CMBL_EXPERIMENT_COL_PIPELINE      // measured and rejected (profiles/r05_ab_col_pipeline_rejected.txt); the kernel is kept for the record
            if constexpr (col_pipelined<T>(LGM)) {
              const int tiles = c->Nx / tile.C, tpw = c->col_tpw(tiles, gs);
              if (tpw == 2) {
                CMBL_LAUNCH_NT(c, K_DELTA_Y, NT, (k_delta_cols_pl<T, R, NT, LGM, 2>), dim3(tiles / tpw, (unsigned)gs), c->ldsY(tile.C), st, d);
                return;
              }
            }
#endif
            CMBL_LAUNCH_NT(c, K_DELTA_Y, NT, (k_delta_cols<T, R, NT, LGM>), dim3(c->Nx / tile.C, (unsigned)gs), c->ldsY(tile.C), st, d);
          });
          // delta-f row pass (RK update of df + next H) + d/dx of the next stage's f (a_nxt holds A_{s+1} after this column launch)
          AdjXArgs<T> x{};
          x.Wx = d.Wx; x.Wy = d.Wy; x.Y0 = df + sp; x.acc = Yacc.as<cx<T>>() + sp; x.Hnext = H.as<cx<T>>() + spm;
          x.twX = c->twX.template as<cx<T>>(); x.lx_r = c->lx_r.template as<T>(); x.Nyh = c->Nyh; x.rk = rk;
          GradXArgs<T> gx{a_nxt + spm, Gx.as<cx<T>>() + spm, x.twX, c->dlx_over_Nx, c->Nyh};
          c->dispatch_row([&](auto lgnx) {
            constexpr int LGNX = decltype(lgnx)::value, RPWMAX = row_rpw<T>(LGNX, 2);
            if constexpr (RPWMAX > 0) {
              c->template dispatch_rpw<RPWMAX, LGNX>(2 * slices, [&](auto rpw_) {          // adjoint part + d/dx part: twice the row groups
                constexpr int RPW = decltype(rpw_)::value;
                const int nb_adj = (int)c->row_groups(gs, RPW);
                CMBL_LAUNCH_NT(c, K_DELTA_ROWS, row_nt(RPW), (k_delta_rows<T, LGNX, RPW>), dim3((unsigned)(nb_adj + (last ? 0 : nb_adj))), c->ldsX(RPW, 2), st, x, gx, nb_adj);
              });
            } else fail(ERR_SHAPE, "row tile does not fit LDS (Nx too large for this precision)");
          });
        }
        std::swap(a_cur, a_nxt);
      }
    join(K);
    dphi_finish(dphi, P, B, nst, alias_quirk, tail);
  }

  // ---- boundary-level entry points ---------------------------------------------------------------
  void apply(int mode, int basis_in, const void* in, int basis_out, void* out, int P, int B) {
    const long slices = (long)P * B, pl = c->plane(), np = c->npix();
    if (mode == F_FWD || mode == F_INV) {
      y0.ensure(sizeof(T) * slices * np);
      T* y = (basis_out == B_MAP) ? (T*)out : y0.as<T>();
      if (basis_in == B_MAP) flow_map((const T*)in, y, P, B, mode == F_INV);
      else {
        cvt.ensure(sizeof(cx<T>) * slices * pl);
        c->ref2F((const cx<T>*)in, cvt.as<cx<T>>(), slices);
        c->from_F(cvt.as<cx<T>>(), basis_in, B_MAP, y, P, B);
        flow_map(y, y, P, B, mode == F_INV);
      }
      if (basis_out != B_MAP) {
        cvt.ensure(sizeof(cx<T>) * slices * pl);
        c->to_F(B_MAP, y, cvt.as<cx<T>>(), basis_out, P, B);
        c->F2ref(cvt.as<cx<T>>(), (cx<T>*)out, slices);
      }
    } else {
      Y0.ensure(sizeof(cx<T>) * slices * pl);
      c->to_F(basis_in, in, Y0.as<cx<T>>(), B_FOURIER, P, B);
      flow_adj_F(Y0.as<cx<T>>(), Y0.as<cx<T>>(), P, B, mode == F_INVADJ);
      c->from_F(Y0.as<cx<T>>(), B_FOURIER, basis_out, out, P, B);
    }
  }

  void grad(int mode, const void* f_end, int basis_delta, const void* delta, void* dphi_out, int basis_df, void* df_out,
            void* f_start_out, int P, int B, bool quirk) {
    const long slices = (long)P * B, pl = c->plane(), np = c->npix();
    y0.ensure(sizeof(T) * slices * np); Y0.ensure(sizeof(cx<T>) * slices * pl); P0.ensure(sizeof(cx<T>) * B * pl);
    T* f = f_start_out ? (T*)f_start_out : y0.as<T>();
    CMBL_HIP(hipMemcpyAsync(f, f_end, sizeof(T) * slices * np, hipMemcpyDeviceToDevice, c->stream));
    c->to_F(basis_delta, delta, Y0.as<cx<T>>(), B_FOURIER, P, B);
    flow_delta(f, Y0.as<cx<T>>(), P0.as<cx<T>>(), P, B, mode == F_FWD, quirk);
    c->from_F(Y0.as<cx<T>>(), B_FOURIER, basis_df, df_out, P, B);
    c->F2ref(P0.as<cx<T>>(), (cx<T>*)dphi_out, B);
  }
};

// =================================================================================================
// Data model, Wiener filter, posterior   (src/dataset.jl, src/maximization.jl, src/numerical_algorithms.jl)
enum OpId { OP_CF_INV = 0, OP_CN_INV, OP_B, OP_MF, OP_D, OP_D_INV, OP_PRECOND_INV, OP_CPHI_INV, OP_G_INV, OP_MPIX, OP_COUNT };

template <typename T>
struct Dataset {
  Ctx<T>* c;
  int P;
  struct Op { DevBuf buf; int nplanes = 0; const T* d[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; int kind = 0; };
  Op ops[OP_COUNT];
  DevBuf d_h; int Bd = 0;                  // data, F layout harmonic
  double logdet_sum = 0;
  // scratch
  DevBuf t1, t2, t3, mp, mp2, xs, rs, zs, ps, aps, best, bb, phiF, gphi, dphi1, dphi2, fh, fhat, ftil, cvt;

  Dataset(Ctx<T>* ctx, int npol) : c(ctx), P(npol) { CMBL_REQUIRE(npol >= 1 && npol <= 3, ERR_ARG, "npol must be 1, 2 or 3"); }

  void set_op(int which, const void* planes, int nplanes) {
    CMBL_REQUIRE(which >= 0 && which < OP_COUNT, ERR_ARG, "bad operator id");
    Op& o = ops[which];
    if (which == OP_MPIX) {
      CMBL_REQUIRE(nplanes == 1, ERR_ARG, "pixel mask is one map");
      o.buf.ensure(sizeof(T) * c->npix());
      CMBL_HIP(hipMemcpyAsync(o.buf.p, planes, sizeof(T) * c->npix(), hipMemcpyDeviceToDevice, c->stream));
      o.nplanes = 1; o.d[0] = o.buf.template as<T>(); o.kind = 1;
      return;
    }
    const bool s0 = (which == OP_CPHI_INV || which == OP_G_INV);
    CMBL_REQUIRE(s0 ? nplanes == 1 : (nplanes == P || (P == 3 && nplanes == 5)), ERR_SHAPE, "wrong number of planes for this operator");
    o.buf.ensure(sizeof(T) * nplanes * c->plane());
    c->ref2F_real((const T*)planes, o.buf.template as<T>(), nplanes);
    o.nplanes = nplanes; o.kind = (nplanes == 5) ? 2 : 1;
    for (int k = 0; k < 5; ++k) o.d[k] = k < nplanes ? o.buf.template as<T>() + (size_t)k * c->plane() : nullptr;
  }
  const Op& op(int which) const {
    CMBL_REQUIRE(ops[which].nplanes > 0, ERR_STATE, "a required dataset operator has not been set");
    return ops[which];
  }
  bool has(int which) const { return ops[which].nplanes > 0; }
  void apply(int which, const cx<T>* in, cx<T>* out, int B, bool transpose = false, bool in_qu = false, bool out_qu = false,
             const cx<T>* z = nullptr, T alpha = 0, T beta = 1, int Pp = -1) {
    const Op& o = op(which);
    c->harm(in, out, Pp < 0 ? P : Pp, B, o.kind, o.d, transpose, in_qu, out_qu, z, alpha, beta);
  }
  void set_data(const void* d_ref, int B) {
    d_h.ensure(sizeof(cx<T>) * (long)P * B * c->plane());
    c->ref2F((const cx<T>*)d_ref, d_h.template as<cx<T>>(), (long)P * B);
    Bd = B;
  }
  long fsize(int B) const { return (long)P * B * c->plane(); }

  // x (QU Fourier, F layout) <- rfft2(mask .* irfft2(x)): x pass, column kernel (c2r, mask, r2c in LDS), x pass
  void pixel_mask(cx<T>* x, long sl) {
    if (c->generic) {
      mp.ensure(sizeof(T) * sl * c->npix());
      c->F_to_map(x, mp.template as<T>(), sl); c->mask_mul(mp.template as<T>(), mp.template as<T>(), ops[OP_MPIX].d[0], sl); c->rfft2_F(mp.template as<T>(), x, sl);
      return;
    }
    cx<T>* m = c->mixed_scratch(sl);
    c->template x_pass<1>(x, m, sl); c->y_mask(m, m, ops[OP_MPIX].d[0], sl); c->template x_pass<0>(m, x, sl);
  }
  // x (harmonic F) -> M x = Mf * (Mpix * x) ; transpose: Mpix' * (Mf' * x)   (src/dataset.jl:279-285)
  // qu: the pixel side of M is exchanged in QU Fourier (x comes in QU Fourier for M, goes out in QU Fourier for M'), so that the beam
  // next to it does the basis change in its own launch instead of a launch of its own
  void apply_M(cx<T>* x, int B, bool transpose, bool qu = false) {
    const long sl = (long)P * B;
    if (!has(OP_MPIX)) { apply(OP_MF, x, x, B, transpose, qu && !transpose, qu && transpose); return; }
    if (!transpose) {
      if (!qu) c->harm(x, x, P, B, 0, nullptr, false, false, true);        // -> QU Fourier
      pixel_mask(x, sl);
      apply(OP_MF, x, x, B, false, true, false);
    } else {
      apply(OP_MF, x, x, B, true, false, true);
      pixel_mask(x, sl);
      if (!qu) c->harm(x, x, P, B, 0, nullptr, false, true, false);
    }
  }

  // ---- fused path (kernels_harm.hpp; power-of-two maps) ----------------------------------------------------------------------------
  bool fused() const { return !c->generic && c->opts.fused_harm; }
  OpRef<T> opref(int which, bool transpose = false) const {
    const Op& o = op(which);
    OpRef<T> r{};
    for (int k = 0; k < 5; ++k) r.d[k] = o.d[k];
    r.kind = o.kind; r.transpose = transpose ? 1 : 0;
    return r;
  }
  template <typename Fn> void by_pol(Fn&& fn) const {
    if (P == 1) fn(std::integral_constant<int, 1>{}); else if (P == 2) fn(std::integral_constant<int, 2>{}); else fn(std::integral_constant<int, 3>{});
  }
  DevBuf m1;                                 // mixed-layout scratch of the data-space part
  // From rfft_y(L f) in `a_ftil` (mixed, left intact) to  w = scale * B'M' Cn^-1 (M B L f - d):  w in QU Fourier to `w_F` (F layout) and
  // ifft_x(w) to L.H (mixed), ready for an adjoint / delta flow with h_ready.  Returns where the
  // partial sums of (MBLf - d)' Cn^-1 (MBLf - d) were left.  value_only: stop after the quadratic form.  dd == nullptr: d = 0.   (src/dataset.jl:59-66,76-80)
  DotOut data_space(Flow<T>& L, const cx<T>* a_ftil, const cx<T>* dd, int dB, T scale, cx<T>* w_F, bool value_only, int B) {
    const long sl = (long)P * B;
    DotOut qn{};
    cx<T>* H = value_only ? nullptr : L.hbuf_ensure(sl);
    by_pol([&](auto pp) {
      constexpr int PP = decltype(pp)::value;
      const ModeGeom<T> g = c->geom();
      if (has(OP_MPIX)) {
        m1.ensure(sizeof(cx<T>) * sl * c->mplane());
        cx<T>* m = m1.template as<cx<T>>();
        PwChain<T, PP> b1{g, {}, 1, 1, nullptr, T(0), nullptr, T(1)};
        b1.ch.push(opref(OP_B));
        c->template x_pw<PP, false>(a_ftil, m, b1, B);                                  // beam
        c->y_mask(m, m, ops[OP_MPIX].d[0], sl);                                         // pixel mask
        PwResid<T, PP> rs{g, {}, {}, opref(OP_CN_INV), dd, dB, T(1), nullptr};
        rs.pre.push(opref(OP_MF)); rs.post.push(opref(OP_MF, true));
        qn = c->template x_pw<PP, false>(m, value_only ? nullptr : m, rs, B, false, Ctx<T>::PART_QN);   // Fourier mask, residual, Cn^-1, quadratic form, Fourier mask'
        if (value_only) return;
        c->y_mask(m, m, ops[OP_MPIX].d[0], sl);                                         // pixel mask'
        PwChain<T, PP> b2{g, {}, 1, 1, nullptr, T(0), w_F, scale};
        b2.ch.push(opref(OP_B, true));
        c->template x_pw<PP, false>(m, H, b2, B);                                       // beam', w stored, ifft_x(w) -> H
      } else {
        PwResid<T, PP> rs{g, {}, {}, opref(OP_CN_INV), dd, dB, scale, value_only ? nullptr : w_F};
        rs.pre.push(opref(OP_B)); rs.pre.push(opref(OP_MF));
        rs.post.push(opref(OP_MF, true)); rs.post.push(opref(OP_B, true));
        qn = c->template x_pw<PP, false>(a_ftil, H, rs, B, false, Ctx<T>::PART_QN);
      }
    });
    return qn;
  }
  // f (harmonic F) -> L f: leaves the lensed maps in `ftil` and rfft_y(L f) in L.A
  void lens_from_harmonic(Flow<T>& L, const cx<T>* f_h, T* ftil, int B) {
    const long sl = (long)P * B;
    mp2.ensure(sizeof(T) * sl * c->npix());
    cx<T>* a = L.abuf_ensure(L.A, sl);
    by_pol([&](auto pp) {
      constexpr int PP = decltype(pp)::value;
      PwChain<T, PP> r{c->geom(), {}, 0, 1, nullptr, T(0), nullptr, T(1)};
      c->template x_pw<PP, true>(f_h, a, r, B, true);                                   // EB -> QU, ifft_x, Hermitian rows projected
    });
    c->y_c2r(a, mp2.template as<T>(), sl);
    L.flow_map(mp2.template as<T>(), ftil, P, B, false, true, true, &L.A);
  }
  // Y = L'B'M'Cn^-1 (d - M B L f)   (QU Fourier, F layout; the data part of gradientf_logpdf)
  void model_adjoint(Flow<T>& L, const cx<T>* f_h, const cx<T>* dd, int dB, cx<T>* Y, int B) {
    const long sl = (long)P * B;
    ftil.ensure(sizeof(T) * sl * c->npix());
    lens_from_harmonic(L, f_h, ftil.template as<T>(), B);
    data_space(L, L.A.template as<cx<T>>(), dd, dB, T(-1), Y, false, B);
    L.flow_adj_F(Y, Y, P, B, false, true);
  }

  // mu = M B L f : harmonic F in (f may be nullptr == 0) -> harmonic F out (t2); uses mp2 for maps
  // `ftil_out`: where the lensed maps f~ = L f are left (default: the scratch mp2)
  void mean(Flow<T>& L, const cx<T>* f_h, cx<T>* out, int B, T* ftil_out = nullptr) {
    const long sl = (long)P * B;
    mp2.ensure(sizeof(T) * sl * c->npix());
    T* m = ftil_out ? ftil_out : mp2.template as<T>();
    c->harm(f_h, out, P, B, 0, nullptr, false, false, true);
    c->F_to_map(out, mp2.template as<T>(), sl);
    L.flow_map(mp2.template as<T>(), m, P, B, false);
    c->rfft2_F(m, out, sl);
    apply(OP_B, out, out, B, false, true, true);                          // QU Fourier -> (EB) x beam -> QU Fourier
    apply_M(out, B, false, true);
  }

  // gradientf_logpdf (src/dataset.jl:76-80): L'B'M'Cn^-1 (d - M B L f) - Cf^-1 f.  f_h == nullptr means f = 0;
  // d_h == nullptr means d = 0.  All harmonic F layout.  out must not alias f_h.
  void gradientf(Flow<T>& L, const cx<T>* f_h, const cx<T>* dd, cx<T>* out, int B) {
    const long n = fsize(B);
    t2.ensure(sizeof(cx<T>) * n);
    cx<T>* r = t2.template as<cx<T>>();
    if (f_h && fused()) {
      model_adjoint(L, f_h, dd, B, r, B);
      c->harm(r, r, P, B, 0, nullptr, false, true, false);                 // -> harmonic
      apply(OP_CF_INV, f_h, out, B, false, false, false, r, (T)1, (T)-1);  // out = r - Cf^-1 f
      return;
    }
    if (f_h) {
      mean(L, f_h, r, B);
      if (dd) c->lincomb1((T*)r, (const T*)dd, (const T*)r, 1.0, -1.0, 2 * n / B, B);
    } else {
      CMBL_REQUIRE(dd != nullptr, ERR_ARG, "gradientf with f = 0 and d = 0 is identically zero");
      CMBL_HIP(hipMemcpyAsync(r, dd, sizeof(cx<T>) * n, hipMemcpyDeviceToDevice, c->stream));
    }
    if (f_h && !dd) apply(OP_CN_INV, r, r, B, false, false, false, nullptr, 0, (T)-1);       // d = 0: the sign of -(M B L f) rides along
    else apply(OP_CN_INV, r, r, B);
    apply_M(r, B, true, true);
    apply(OP_B, r, r, B, true, true, true);                                // QU Fourier -> (EB) x beam' -> QU Fourier
    L.flow_adj_F(r, r, P, B, false);
    if (f_h) {
      c->harm(r, r, P, B, 0, nullptr, false, true, false);                 // -> harmonic
      apply(OP_CF_INV, f_h, out, B, false, false, false, r, (T)1, (T)-1);  // out = r - Cf^-1 f
    } else {
      c->harm(r, out, P, B, 0, nullptr, false, true, false);
    }
  }

  // conjugate_gradient (src/numerical_algorithms.jl:73-134) driving argmaxf_logpdf (src/maximization.jl:17-42).
  // a0 = gradientf(f=0,d=0) is identically zero for this linear model (all operators finite), so it is not evaluated.
  // The scalars (res, alpha, beta, best residual, history, stop flag) live on the device (CgState): an iteration is enqueued
  // without waiting for its reductions; the host reads the stop flag of iteration i-1 while iteration i runs.
  DevBuf cg_scal, cg_hist, qdev, cg_rzfin;
  int* cg_flag_host = nullptr;                 // pinned: [slot][done, nan]
  hipEvent_t cg_ev[2] = {nullptr, nullptr};
  ~Dataset() {
    if (cg_flag_host) (void)hipHostFree(cg_flag_host);
    for (auto& e : cg_ev) if (e) (void)hipEventDestroy(e);
  }
  // Fused form of the same iteration (power-of-two maps): per iteration the two flows, 7 launches between them (basis change + ifft_x,
  // c2r, beam, mask, Fourier mask / Cn^-1 / Fourier mask', mask', beam' + ifft_x) and 3 after them (A p and p'Ap -> alpha; x, r, r'z ->
  // beta and the stop test; p and the best iterate).  Every scalar is produced by the launch that produced its sum.
  int wiener_cg_fused(Flow<T>& L, const cx<T>* dd, const cx<T>* fstart, double tol, int maxit, cx<T>* f_out, double* hist, int B) {
    const long n = fsize(B), nr = 2 * n / B;
    xs.ensure(sizeof(cx<T>) * n); rs.ensure(sizeof(cx<T>) * n); ps.ensure(sizeof(cx<T>) * n);
    aps.ensure(sizeof(cx<T>) * n); best.ensure(sizeof(cx<T>) * n); bb.ensure(sizeof(cx<T>) * n); zs.ensure(sizeof(cx<T>) * n);
    cx<T>*x = xs.template as<cx<T>>(), *r = rs.template as<cx<T>>(), *p = ps.template as<cx<T>>(), *Y = zs.template as<cx<T>>();
    cx<T>*Ap = aps.template as<cx<T>>(), *bx = best.template as<cx<T>>(), *b = bb.template as<cx<T>>();
    CMBL_REQUIRE(B <= MAXBATCH, ERR_ARG, "nbatch > 256 not supported in reductions");
    cg_scal.ensure(sizeof(double) * 7 * MAXBATCH + sizeof(int) * 8);
    cg_rzfin.ensure(sizeof(double) * MAXBATCH);
    cg_hist.ensure(sizeof(double) * (size_t)maxit * B);
    if (!cg_flag_host) {
      CMBL_HIP(hipHostMalloc((void**)&cg_flag_host, sizeof(int) * 16, hipHostMallocDefault));
      for (auto& e : cg_ev) CMBL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    CgScal st;
    double* sd = cg_scal.template as<double>();
    st.res = sd; st.best = sd + 2 * MAXBATCH;
    st.hist = cg_hist.template as<double>();
    int* si = reinterpret_cast<int*>(sd + 7 * MAXBATCH);
    st.done = si; st.better = si + 2; st.nan = si + 4; st.nh = si + 5;
    hipStream_t sm = c->stream;
    std::vector<double> one(B, 1.0), mone(B, -1.0);
    // b = -gradientf(f=0, d) = -L'B'M'Cn^-1 d
    gradientf(L, nullptr, dd, b, B);
    c->lincomb((T*)b, (T*)b, nullptr, mone.data(), nullptr, nr, B);
    if (fstart) {
      CMBL_HIP(hipMemcpyAsync(x, fstart, sizeof(cx<T>) * n, hipMemcpyDeviceToDevice, sm));
      gradientf(L, x, nullptr, Ap, B);                                      // A x
      c->lincomb((T*)r, (T*)b, (T*)Ap, one.data(), mone.data(), nr, B);     // r = b - A x
    } else {
      CMBL_HIP(hipMemsetAsync(x, 0, sizeof(cx<T>) * n, sm));
      CMBL_HIP(hipMemcpyAsync(r, b, sizeof(cx<T>) * n, hipMemcpyDeviceToDevice, sm));
    }
    const ModeGeom<T> g = c->geom();
    const double sc = c->dot_scale();
    by_pol([&](auto pp) {
      constexpr int PP = decltype(pp)::value;
      const DotOut o = c->pw_flat(PwCgStart<T, PP>{g, opref(OP_PRECOND_INV), x, r, p, bx}, B, Ctx<T>::PART_CG_RZ);
      CMBL_LAUNCH(c, K_CG, (k_cg_init<T>), dim3(1), 0, sm, st, o, sc, B);
    });
    auto post_flags = [&](int slot) {
      CMBL_HIP(hipMemcpyAsync(cg_flag_host + 8 * slot, st.done, sizeof(int) * 6, hipMemcpyDeviceToHost, sm));
      CMBL_HIP(hipEventRecord(cg_ev[slot], sm));
    };
    auto wait_flags = [&](int slot) { CMBL_HIP(hipEventSynchronize(cg_ev[slot])); return cg_flag_host + 8 * slot; };
    post_flags(0);
    int par = 0;
    if (wait_flags(0)[4] == 0) {                                            // a NaN start residual is reported below
      for (int it = 2; it <= maxit; ++it) {
        model_adjoint(L, p, nullptr, B, Y, B);                              // Y = -L'B'M'Cn^-1 M B L p
        by_pol([&](auto pp) {
          constexpr int PP = decltype(pp)::value;
          const DotOut oA = c->pw_flat(PwCgAp<T, PP>{g, opref(OP_CF_INV), Y, p, Ap}, B, Ctx<T>::PART_CG_PAP);                // A p = Y - Cf^-1 p
          const DotOut oR = c->pw_flat(PwCgXr<T, PP>{g, opref(OP_PRECOND_INV), x, r, p, Ap, st, par, oA, sc}, B, Ctx<T>::PART_CG_RZ);   // alpha ; x, r
          // beta, stop test ; p, best iterate.  Every block needs r'z of ALL slots (`all(res < bestres)`, :111): up to 16 slots it adds
          // their partials itself in its prologue; beyond that one small launch finishes them first (O(B) instead of O(B^2) per block)
          const double* rzf = nullptr;
          if (B > 16) { double* const o1[1] = {cg_rzfin.as<double>()}; c->finish_parts(&oR, o1, 1, B); rzf = o1[0]; }
          c->pw_flat(PwCgP<T, PP>{g, opref(OP_PRECOND_INV), x, r, p, bx, st, par, tol, oR, sc, rzf}, B);
        });
        par ^= 1;
        post_flags(it & 1);
        if (it > 2) { const int* fl = wait_flags((it - 1) & 1); if (fl[it & 1] != 0) break; }   // previous iteration's flags: it left `done` at parity (it - 2) & 1
      }
    }
    CMBL_HIP(hipMemcpyAsync(f_out, bx, sizeof(cx<T>) * n, hipMemcpyDeviceToDevice, sm));
    int fl[6];
    CMBL_HIP(hipMemcpyAsync(fl, st.done, sizeof(int) * 6, hipMemcpyDeviceToHost, sm));
    CMBL_HIP(hipStreamSynchronize(sm));
    const int nh = fl[5];
    CMBL_HIP(hipMemcpy(hist, st.hist, sizeof(double) * (size_t)nh * B, hipMemcpyDeviceToHost));
    CMBL_REQUIRE(fl[4] == 0, ERR_NAN, "NaN residual in conjugate gradient");
    return nh;
  }
  int wiener_cg(Flow<T>& L, const cx<T>* dd, const cx<T>* fstart, double tol, int maxit, cx<T>* f_out, double* hist, int B) {
    if (fused()) return wiener_cg_fused(L, dd, fstart, tol, maxit, f_out, hist, B);
    const long n = fsize(B), nr = 2 * n / B;
    xs.ensure(sizeof(cx<T>) * n); rs.ensure(sizeof(cx<T>) * n); zs.ensure(sizeof(cx<T>) * n); ps.ensure(sizeof(cx<T>) * n);
    aps.ensure(sizeof(cx<T>) * n); best.ensure(sizeof(cx<T>) * n); bb.ensure(sizeof(cx<T>) * n);
    cx<T>*x = xs.template as<cx<T>>(), *r = rs.template as<cx<T>>(), *z = zs.template as<cx<T>>(), *p = ps.template as<cx<T>>();
    cx<T>*Ap = aps.template as<cx<T>>(), *bx = best.template as<cx<T>>(), *b = bb.template as<cx<T>>();
    CMBL_REQUIRE(B <= MAXBATCH, ERR_ARG, "nbatch > 256 not supported in reductions");
    cg_scal.ensure(sizeof(double) * 6 * MAXBATCH + sizeof(int) * 8);
    cg_hist.ensure(sizeof(double) * (size_t)maxit * B);
    if (!cg_flag_host) {
      CMBL_HIP(hipHostMalloc((void**)&cg_flag_host, sizeof(int) * 16, hipHostMallocDefault));
      for (auto& e : cg_ev) CMBL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    CgState st;
    double* sd = cg_scal.template as<double>();
    st.res = sd; st.pAp = sd + MAXBATCH; st.res2 = sd + 2 * MAXBATCH; st.alpha = sd + 3 * MAXBATCH; st.beta = sd + 4 * MAXBATCH; st.best = sd + 5 * MAXBATCH;
    st.hist = cg_hist.template as<double>();
    int* si = reinterpret_cast<int*>(sd + 6 * MAXBATCH);
    st.done = si; st.nan = si + 1; st.nh = si + 2; st.better = si + 3;
    hipStream_t sm = c->stream;
    const unsigned gx = (unsigned)std::min<long>((nr + NTP - 1) / NTP, 2048);
    std::vector<double> one(B, 1.0), mone(B, -1.0);
    // b = -gradientf(f=0, d) = -L'B'M'Cn^-1 d
    gradientf(L, nullptr, dd, b, B);
    c->lincomb((T*)b, (T*)b, nullptr, mone.data(), nullptr, nr, B);
    if (fstart) {
      CMBL_HIP(hipMemcpyAsync(x, fstart, sizeof(cx<T>) * n, hipMemcpyDeviceToDevice, sm));
      gradientf(L, x, nullptr, Ap, B);                                      // A x
      c->lincomb((T*)r, (T*)b, (T*)Ap, one.data(), mone.data(), nr, B);     // r = b - A x
    } else {
      CMBL_HIP(hipMemsetAsync(x, 0, sizeof(cx<T>) * n, sm));
      CMBL_HIP(hipMemcpyAsync(r, b, sizeof(cx<T>) * n, hipMemcpyDeviceToDevice, sm));
    }
    apply(OP_PRECOND_INV, r, z, B);
    CMBL_HIP(hipMemcpyAsync(p, z, sizeof(cx<T>) * n, hipMemcpyDeviceToDevice, sm));
    c->dot_F_dev(r, z, P, B, st.res);
    CMBL_LAUNCH_NT(c, K_CG, 64, k_cg_start, dim3(1), 0, sm, st, B);
    CMBL_HIP(hipMemcpyAsync(bx, x, sizeof(cx<T>) * n, hipMemcpyDeviceToDevice, sm));
    auto post_flags = [&](int slot) {
      CMBL_HIP(hipMemcpyAsync(cg_flag_host + 2 * slot, st.done, sizeof(int) * 2, hipMemcpyDeviceToHost, sm));
      CMBL_HIP(hipEventRecord(cg_ev[slot], sm));
    };
    auto stopped = [&](int slot) {                                          // waits for the flags posted in `slot`
      CMBL_HIP(hipEventSynchronize(cg_ev[slot]));
      return cg_flag_host[2 * slot] != 0;
    };
    post_flags(0);
    (void)stopped(0);
    if (cg_flag_host[1] == 0) {                                             // a NaN start residual is reported below
      for (int it = 2; it <= maxit; ++it) {
        gradientf(L, p, nullptr, Ap, B);
        c->dot_F_dev(p, Ap, P, B, st.pAp);
        CMBL_LAUNCH_NT(c, K_CG, 64, k_cg_alpha, dim3(1), 0, sm, st, B);
        CMBL_LAUNCH(c, K_CG, (k_cg_xr<T>), dim3(gx, B), 0, sm, (T*)x, (T*)r, (const T*)p, (const T*)Ap, st, nr);
        apply(OP_PRECOND_INV, r, z, B);
        c->dot_F_dev(r, z, P, B, st.res2);
        CMBL_LAUNCH_NT(c, K_CG, 64, k_cg_beta, dim3(1), 0, sm, st, B, tol);
        CMBL_LAUNCH(c, K_CG, (k_cg_p<T>), dim3(gx, B), 0, sm, (T*)p, (const T*)z, st, nr);
        CMBL_LAUNCH(c, K_CG, (k_cg_keep_best<T>), dim3(gx), 0, sm, (T*)bx, (const T*)x, st, 2 * n);
        post_flags(it & 1);
        if (it > 2 && stopped((it - 1) & 1)) break;                         // flags of the previous iteration
      }
    }
    CMBL_HIP(hipMemcpyAsync(f_out, bx, sizeof(cx<T>) * n, hipMemcpyDeviceToDevice, sm));
    int fl[4];
    CMBL_HIP(hipMemcpyAsync(fl, st.done, sizeof(int) * 4, hipMemcpyDeviceToHost, sm));
    CMBL_HIP(hipStreamSynchronize(sm));
    const int nh = fl[2];
    CMBL_HIP(hipMemcpy(hist, st.hist, sizeof(double) * (size_t)nh * B, hipMemcpyDeviceToHost));
    CMBL_REQUIRE(fl[1] == 0, ERR_NAN, "NaN residual in conjugate gradient");
    return nh;
  }

  // logpdf(Mixed(ds); f°, phi°) and optionally its gradient.  fo: map (reference layout == internal), phio: F layout S0.
  // gfo (map) / gphio (F) may be nullptr for value only.
  // Fused form (power-of-two maps).  Launches outside the four flows: phi prior (1), precompute (4), the y pass of f° (1), unmix + Cf^-1
  // + its quadratic form + ifft_x (1), c2r (1), the data-space part (5 with a pixel mask, 1 without), D'^-1 between the delta flows (1),
  // per delta flow the first d/dx pass and the delta-phi epilogue (1 + 3), the final irfft2 (2): 25 with a pixel mask.
  void logpdf_mixed_fused(Flow<T>& L, const T* fo, const cx<T>* phio_F, double* lp, T* gfo, cx<T>* gphio_F, int B, bool quirk) {
    const long sl = (long)P * B, n = fsize(B), np = c->npix(), pl = c->plane();
    phiF.ensure(sizeof(cx<T>) * B * pl); fhat.ensure(sizeof(T) * sl * np); ftil.ensure(sizeof(T) * sl * np);
    fh.ensure(sizeof(cx<T>) * n); t2.ensure(sizeof(cx<T>) * n); t3.ensure(sizeof(cx<T>) * n);
    gphi.ensure(sizeof(cx<T>) * B * pl); dphi1.ensure(sizeof(cx<T>) * B * pl);
    mp2.ensure(sizeof(T) * sl * np);
    cx<T>*phi = phiF.template as<cx<T>>(), *f_h = fh.template as<cx<T>>(), *w = t3.template as<cx<T>>(), *cfif = t2.template as<cx<T>>();
    cx<T>* cpip = gphi.template as<cx<T>>();
    CMBL_REQUIRE(B <= MAXBATCH, ERR_ARG, "nbatch > 256 not supported in reductions");
    qdev.ensure(sizeof(double) * 3 * MAXBATCH);
    double* qd = qdev.template as<double>();
    // phi = G \ phi° ; Cphi^-1 phi ; phi' Cphi^-1 phi
    DotOut parts[3];
    parts[1] = c->pw_flat(PwPhiPrior<T>{c->lam.template as<T>(), pl, op(OP_G_INV).d[0], op(OP_CPHI_INV).d[0], phio_F, phi, cpip}, B, Ctx<T>::PART_QP);
    L.set_phi_F(phi, B);
    // fhat = L \ f° (its y transform stays in A3) ; f = D \ fhat ; Cf^-1 f ; f' Cf^-1 f ; the y-transformed f goes straight into the forward flow
    L.flow_map(fo, fhat.template as<T>(), P, B, true, false, true, &L.A3);
    cx<T>* a = L.abuf_ensure(L.A, sl);
    by_pol([&](auto pp) {
      constexpr int PP = decltype(pp)::value;
      parts[0] = c->template x_pw<PP, false>(L.A3.template as<cx<T>>(), a, PwUnmixPrior<T, PP>{c->geom(), opref(OP_D_INV), opref(OP_CF_INV), f_h, cfif}, B, true, Ctx<T>::PART_QF);
    });
    c->y_c2r(a, mp2.template as<T>(), sl);
    L.flow_map(mp2.template as<T>(), ftil.template as<T>(), P, B, false, true, true, &L.A);       // f~ = L f in ftil, rfft_y(f~) in A
    // z = M B L f - d ; z' Cn^-1 z ; w = -B'M'Cn^-1 z  (QU Fourier) with ifft_x(w) ready in L.H
    parts[2] = data_space(L, L.A.template as<cx<T>>(), d_h.template as<cx<T>>(), Bd, T(-1), w, gfo == nullptr, B);
    double* const outs[3] = {qd, qd + MAXBATCH, qd + 2 * MAXBATCH};
    c->finish_parts(parts, outs, 3, B);
    auto finish_lp = [&]() {
      double q[3 * MAXBATCH];
      CMBL_HIP(hipMemcpyAsync(q, qd, sizeof(double) * 3 * MAXBATCH, hipMemcpyDeviceToHost, c->stream));
      CMBL_HIP(hipStreamSynchronize(c->stream));
      for (int i = 0; i < B; ++i) lp[i] = -0.5 * (q[i] + q[MAXBATCH + i] + q[2 * MAXBATCH + i] + logdet_sum);
    };
    if (!gfo) { finish_lp(); return; }
    // pullback through f~ = L f : delta flow t 1->0 from (f~, w, 0)
    L.flow_delta(ftil.template as<T>(), w, dphi1.template as<cx<T>>(), P, B, true, quirk, true, &L.A, true);
    // g_f = df1 - Cf^-1 f (harmonic) ; d/dfhat = D' \ g_f -> QU Fourier, with its ifft_x in L.H
    by_pol([&](auto pp) {
      constexpr int PP = decltype(pp)::value;
      PwChain<T, PP> ch{c->geom(), {}, 1, 1, cfif, T(-1), w, T(1)};
      ch.ch.push(opref(OP_D_INV, true));
      c->template x_pw<PP, true>(w, L.hbuf_ensure(sl), ch, B);
    });
    // pullback through fhat = L \ f° : delta flow t 0->1 from (fhat, w, 0); the epilogue forms d/dphi° = G' \ (dphi1 + dphi2 - Cphi^-1 phi)
    const DphiTail<T> tail{dphi1.template as<cx<T>>(), cpip, op(OP_G_INV).d[0]};
    L.flow_delta(fhat.template as<T>(), w, gphio_F, P, B, false, quirk, true, &L.A3, true, &tail);
    c->F_to_map(w, gfo, sl);                                                // d/df° in f°'s basis (QU map)
    finish_lp();
  }
  void logpdf_mixed(Flow<T>& L, const T* fo, const cx<T>* phio_F, double* lp, T* gfo, cx<T>* gphio_F, int B, bool quirk) {
    CMBL_REQUIRE(Bd == B, ERR_SHAPE, "dataset data batch size differs from nbatch");
    if (fused()) return logpdf_mixed_fused(L, fo, phio_F, lp, gfo, gphio_F, B, quirk);
    const long sl = (long)P * B, n = fsize(B), np = c->npix(), pl = c->plane();
    phiF.ensure(sizeof(cx<T>) * B * pl); fhat.ensure(sizeof(T) * sl * np); ftil.ensure(sizeof(T) * sl * np);
    fh.ensure(sizeof(cx<T>) * n); t1.ensure(sizeof(cx<T>) * n); t3.ensure(sizeof(cx<T>) * n);
    gphi.ensure(sizeof(cx<T>) * B * pl); dphi1.ensure(sizeof(cx<T>) * B * pl); dphi2.ensure(sizeof(cx<T>) * B * pl);
    cx<T>*phi = phiF.template as<cx<T>>(), *f_h = fh.template as<cx<T>>(), *z = t1.template as<cx<T>>(), *w = t3.template as<cx<T>>();
    // phi = G \ phi°
    apply(OP_G_INV, phio_F, phi, B, false, false, false, nullptr, 0, 1, 1);
    L.set_phi_F(phi, B);
    // fhat = L \ f° ; f = D \ fhat
    L.flow_map(fo, fhat.template as<T>(), P, B, true);
    c->rfft2_F(fhat.template as<T>(), f_h, sl);
    apply(OP_D_INV, f_h, f_h, B, false, true, false);
    // z = M B L f - d
    mean(L, f_h, z, B, ftil.template as<T>());                            // leaves f~ = L f in ftil
    c->lincomb1((T*)z, (const T*)z, (const T*)d_h.p, 1.0, -1.0, 2 * n / B, B);
    // quadratic forms: the three sets of B sums stay on the device and are read back ONCE, after everything else of this call has been
    // enqueued -- a read-back per term drained the stream three times in the middle of a gradient evaluation
    CMBL_REQUIRE(B <= MAXBATCH, ERR_ARG, "nbatch > 256 not supported in reductions");
    qdev.ensure(sizeof(double) * 3 * MAXBATCH);
    double* qd = qdev.template as<double>();
    apply(OP_CN_INV, z, w, B);                c->dot_F_dev(z, w, P, B, qd + 2 * MAXBATCH);       // w = Cn^-1 z  (kept)
    cx<T>* cfif = t2.template as<cx<T>>(); t2.ensure(sizeof(cx<T>) * n); cfif = t2.template as<cx<T>>();
    apply(OP_CF_INV, f_h, cfif, B);           c->dot_F_dev(f_h, cfif, P, B, qd);
    cx<T>* cpip = gphi.template as<cx<T>>();
    apply(OP_CPHI_INV, phi, cpip, B, false, false, false, nullptr, 0, 1, 1);
    c->dot_F_dev(phi, cpip, 1, B, qd + MAXBATCH);
    // A NaN logpdf is a VALUE, not an error: MAP_joint's line search penalises it (src/maximization.jl:194-199) and hmc_step
    // rejects the proposal (log(rand()) < NaN is false, src/sampling.jl:414), so it must reach the caller.
    auto finish_lp = [&]() {
      double q[3 * MAXBATCH];
      CMBL_HIP(hipMemcpyAsync(q, qd, sizeof(double) * 3 * MAXBATCH, hipMemcpyDeviceToHost, c->stream));
      CMBL_HIP(hipStreamSynchronize(c->stream));
      for (int i = 0; i < B; ++i) lp[i] = -0.5 * (q[i] + q[MAXBATCH + i] + q[2 * MAXBATCH + i] + logdet_sum);
    };
    if (!gfo) { finish_lp(); return; }
    // d/df~ = -B'M'Cn^-1 z  -> QU Fourier
    apply_M(w, B, true, true);
    apply(OP_B, w, w, B, true, true, true, nullptr, 0, (T)-1);
    // pullback through f~ = L f : delta flow t 1->0 from (f~, w, 0)
    L.flow_delta(ftil.template as<T>(), w, dphi1.template as<cx<T>>(), P, B, true, quirk);
    // g_f = df1 - Cf^-1 f   (harmonic) ; then d/dfhat = D' \ g_f  -> QU Fourier
    c->harm(w, w, P, B, 0, nullptr, false, true, false);
    c->lincomb1((T*)w, (const T*)w, (const T*)cfif, 1.0, -1.0, 2 * n / B, B);
    apply(OP_D_INV, w, w, B, true, false, true);
    // pullback through fhat = L \ f° : delta flow t 0->1 from (fhat, w, 0)
    L.flow_delta(fhat.template as<T>(), w, dphi2.template as<cx<T>>(), P, B, false, quirk);
    // d/df° in f°'s basis (QU map)
    c->F_to_map(w, gfo, sl);
    // g_phi = dphi1 + dphi2 - Cphi^-1 phi ; d/dphi° = G' \ g_phi
    cx<T>* g = dphi1.template as<cx<T>>();
    c->lincomb1((T*)g, (const T*)g, (const T*)dphi2.p, 1.0, 1.0, 2 * pl, B);
    c->lincomb1((T*)g, (const T*)g, (const T*)cpip, 1.0, -1.0, 2 * pl, B);
    apply(OP_G_INV, g, gphio_F, B, true, false, false, nullptr, 0, 1, 1);
    finish_lp();
  }
};

}  // namespace cmbl
