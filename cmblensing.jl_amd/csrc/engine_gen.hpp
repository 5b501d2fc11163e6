// The any-size transform LAUNCHES of Ctx<T> (declared in engine.hpp): the members that set up a GenDft argument block and launch k_gen_dft* or, by
// length, the compile-time-plan kernels through CtLaunchY / CtLaunchX (engine_ct.hpp).  Defined out of class (hence not inline) and instantiated
// explicitly by tu_gen_{f32,f64}.hip, so that no other translation unit compiles the run-time-plan kernels (api_decl.hpp has the map of the build).
#pragma once
#include "engine.hpp"

namespace cmbl {

template <typename T>
bool Ctx<T>::gen_dft_ct(const GenAxis& ax, GenDft<T> a, long slices) {
  a.N = ax.N; a.tw = ax.twN.template as<cx<T>>(); a.S = ct_Smax<T>(ax.N);   // (the groups of k_ct_dft: as many sequences as fit the LDS)
  const dim3 grid((unsigned)((a.nseq + a.S - 1) / a.S), (unsigned)slices);
  int kind = ct_kind(a);
#ifdef CMBL_STAMPS_CT
  static const int stamp_kind = env_int("CMBL_CT_STAMP_KIND", -1);
  if (kind + (a.lmul_mid ? 8 : 0) == stamp_kind) kind |= 256;
#endif
  if ((kind & 255) == CT_C && a.in_elem == 1 && a.out_elem == 1) {                 // contiguous rows: groups of 8 / 4 / 2 (ct_rows_per_group)
    const int Sx = std::min(ct_rows_per_group((long)a.nseq * slices, 1), ct_Smax<T>(ax.N));
    a.S = Sx;
    const dim3 gx((unsigned)xgroups((a.nseq + Sx - 1) / Sx, a.in_tiled || a.out_tiled), (unsigned)slices);
    switch (ax.N) {
#define CMBL_X(n) case n: CtLaunchY<T, n>::dftx(this, a, gx, Sx, kind); return true;
      CMBL_CT_LIST(CMBL_X)
#undef CMBL_X
      default: return false;
    }
  }
  switch (ax.N) {
#define CMBL_X(n) case n: CtLaunchY<T, n>::dft(this, a, grid, kind); return true;
    CMBL_CT_LIST(CMBL_X)
#undef CMBL_X
    default: return false;
  }
}

template <typename T>
void Ctx<T>::gen_dft(const GenAxis& ax, GenDft<T> a, long slices) {
  slices = gen_window(a, slices);
  if (ax.plan.nf > 0 && opts.gen_ct && gen_dft_ct(ax, a, slices)) return;
  if (ax.plan.nf > 0) {
    a.N = ax.N; a.tw = ax.twN.template as<cx<T>>();
    // sequences per workgroup: enough of them for coalesced strided access, but not so many that the launch has fewer than a few
    // workgroups per CU (small maps); the two LDS buffers + the twiddle table within 64 KB
    auto lds_of = [&](int S, bool tw) { return ((size_t)2 * S * ax.N + (tw ? ax.N : 0)) * sizeof(cx<T>); };
    int minR = 13; bool big = false;
    for (int i = 0; i < ax.plan.nf; ++i) { minR = std::min(minR, ax.plan.radix[i]); big = big || ax.plan.radix[i] > 5; }
    const long total = (long)a.nseq * slices;
    const bool strided = a.in_elem != 1 || a.out_elem != 1;
    a.S = (int)std::max<long>(1, std::min<long>(std::min(16, 2048 / ax.N), total / (4L * num_cus)));
    while (a.S > 1 && lds_of(a.S, true) > 64 * 1024) --a.S;
    // a strided side is read / written in pieces of S elements: at least 64 bytes of them, in one large workgroup per CU
    const int Smin = 64 / (int)sizeof(cx<T>);
    if (strided && !big && a.S < Smin && total >= (long)Smin * num_cus / 2 && lds_of(Smin, false) <= 150 * 1024) a.S = Smin;
    const bool tw_lds = lds_of(a.S, true) <= 158 * 1024;
    const int nthr = std::max(64, std::min(big ? NTP : 1024, ((a.S * ax.N / minR + 63) / 64) * 64));
    const dim3 grid((unsigned)((a.nseq + a.S - 1) / a.S), (unsigned)slices);
    if (big) CMBL_LAUNCH_NT(this, K_GEN_DFT, nthr, (k_gen_dft_mr<T, true>), grid, lds_of(a.S, tw_lds), stream, a, ax.plan, tw_lds ? 1 : 0);
    else CMBL_LAUNCH_NT(this, K_GEN_DFT, nthr, (k_gen_dft_mr<T, false>), grid, lds_of(a.S, tw_lds), stream, a, ax.plan, tw_lds ? 1 : 0);
    return;
  }
  const int L = 1 << ax.lgL;
  a.chirp = ax.chirp.template as<cx<T>>(); a.bhat = ax.bhat.template as<cx<T>>(); a.tw = ax.tw.template as<cx<T>>(); a.N = ax.N;
  a.S = std::max(1, std::min(8, 2048 / L));
  const size_t lds = (size_t)a.S * tile_ld(L) * sizeof(cx<T>);
  const dim3 grid((unsigned)((a.nseq + a.S - 1) / a.S), (unsigned)slices);
  bool done = false;
#define CMBL_X(lg) if (!done && ax.lgL == lg) { CMBL_LAUNCH(this, K_GEN_DFT, (k_gen_dft<T, lg>), grid, lds, stream, a); done = true; }
  CMBL_GEN_LIST(CMBL_X)
#undef CMBL_X
  if (!done) fail(ERR_SHAPE, "unsupported transform length");
}

template <typename T>
void Ctx<T>::gen_y_flow_stage(const cx<T>* G1, const cx<T>* G2, const T* lmul2, T s1, T s2, const GenPro<T>& pro, cx<T>* Anext, bool last, long slices) {
  GenDft<T> a{};
  a.pro = pro;
  a.in = G1; a.in2 = G2; a.lmul_in = lmul2; a.herm = 1; a.out_real = 1; a.inverse = 1; a.nin = Nyh; a.nout = Ny; a.nseq = Nx;
  a.scale = s1; a.scale2 = s2;
  a.in_seq = 1; a.in_elem = Nx; a.in_slice = plane(); a.out_seq = Ny; a.out_elem = 1; a.out_slice = npix();
  a.yy = 1; a.yy_last = last ? 1 : 0; a.yy_nout = Nyh; a.yy_out = Anext;
  hand_in<1>(a);
  slices = gen_window(a, slices);
  a.N = Ny; a.tw = genY.twN.template as<cx<T>>(); a.S = std::min(ct_cols_per_group((long)a.nseq, slices, ct_S<T>()), ct_Smax<T>(Ny));
  const dim3 grid((unsigned)((a.nseq + a.S - 1) / a.S), (unsigned)slices);
  switch (Ny) {
#define CMBL_X(n) case n: CtLaunchY<T, n>::flow_y(this, a, grid); return;
    CMBL_CT_LIST(CMBL_X)
#undef CMBL_X
    default: fail(ERR_STATE, "fused y passes need a compile-time plan for Ny");
  }
}

template <typename T>
void Ctx<T>::gen_y_delta_stage(const cx<T>* T3, T s3, const cx<T>* G1, const cx<T>* G2, const T* lmul2, T s1, T s2, const GenPro<T>& pro, cx<T>* Anext, cx<T>* W2a, cx<T>* W2b, bool last, long slices) {
  GenDft<T> a{};
  a.pro = pro;
  a.in = G1; a.in2 = G2; a.lmul_in = lmul2; a.herm = 1; a.out_real = 1; a.inverse = 1; a.nin = Nyh; a.nout = Ny; a.nseq = Nx;
  a.scale = s1; a.scale2 = s2;
  a.in_seq = 1; a.in_elem = Nx; a.in_slice = plane(); a.out_seq = Ny; a.out_elem = 1; a.out_slice = npix();
  a.yy = 2; a.yy_last = last ? 1 : 0; a.yy_nout = Nyh; a.yy_out = Anext; a.yy_in3 = T3; a.yy_scale3 = s3; a.yy_out2 = W2a; a.yy_out3 = W2b;
  hand_in<1>(a);
  slices = gen_window(a, slices);
  a.N = Ny; a.tw = genY.twN.template as<cx<T>>(); a.S = ct_cols_per_group((long)a.nseq, slices, ct_S2<T>(Ny));
  const dim3 grid((unsigned)((a.nseq + a.S - 1) / a.S), (unsigned)slices);
  switch (Ny) {
#define CMBL_X(n) case n: if (CtLaunchY<T, n>::delta_y(this, a, grid)) return; break;
    CMBL_CT_LIST(CMBL_X)
#undef CMBL_X
    default: break;
  }
  fail(ERR_STATE, "fused delta-stage y passes need a compile-time plan for Ny that fits LDS twice");
}

template <typename T>
bool Ctx<T>::gen_x_adj_update(const cx<T>* W2a, const cx<T>* W2b, cx<T>* Y0, cx<T>* acc_, cx<T>* Ys, const RKCoef<T>& rk, long slices) {
  if (!opts.gen_ct || !opts.gen_yy || genX.plan.nf == 0 || !ct_rowfuse_ok<T>(Nx)) return false;
  GenDft<T> a{};
  a.in = W2a; a.in2 = W2b; a.nin = Nx; a.nout = Nx; a.nseq = Nyh; a.scale = 1;
  a.in_seq = Nx; a.in_elem = 1; a.in_slice = plane(); a.out_seq = Nx; a.out_elem = 1; a.out_slice = plane();
  a.pro.rk = rk; a.yy_out = Y0; a.out2 = acc_; a.out = Ys; a.lmul_out = lx_r.template as<T>(); a.lmul_in = ly.template as<T>();
  hand_in<2>(a);
  slices = gen_window(a, slices);
  a.N = Nx; a.tw = genX.twN.template as<cx<T>>();
  const int Sx = std::min(std::max(ct_S<T>() / 2, ct_rows_per_group((long)a.nseq * slices, 2)), ct_Smax<T>(Nx));   // S wavefronts = S / 2 rows x the two members of the pair
  a.S = Sx;
  const int R = Sx / 2;
  const dim3 grid((unsigned)xgroups((a.nseq + R - 1) / R, a.in_tiled != 0), (unsigned)slices);
  switch (Nx) {
#define CMBL_X(n) case n: CtLaunchX<T, n>::adj_x(this, a, grid, Sx); return true;
    CMBL_CT_LIST(CMBL_X)
#undef CMBL_X
    default: return false;
  }
}

// gen_x_adj_update that ALSO opens the next stage (round 6): the row workgroups that hold the new stage input Ys transform it back, t3 = ifft_x(Ys)
// (no Ys round trip through memory, one launch less per stage), and -- delta flow: A_next given -- the d/dx pass gx = ifft_x(i lx fft_x(A_next)) of
// the next stage's f rides in the same launch as further workgroups (k_ct_adj_x_dx).  Replaces gen_x_adj_update(s) + gen_x_inv_and_deriv(s + 1)
// / gen_x(s + 1); needs a compile-time plan for Nx (gen_ct_x()).  Results bit-identical to the separate launches.
template <typename T>
void Ctx<T>::gen_x_adj_next(const cx<T>* W2a, const cx<T>* W2b, cx<T>* Y0, cx<T>* acc_, const RKCoef<T>& rk, cx<T>* t3, const cx<T>* A_next, cx<T>* gx, long slices) {
  GenDft<T> a{};
  a.in = W2a; a.in2 = W2b; a.nin = Nx; a.nout = Nx; a.nseq = Nyh; a.scale = 1;
  a.in_seq = Nx; a.in_elem = 1; a.in_slice = plane(); a.out_seq = Nx; a.out_elem = 1; a.out_slice = plane();
  a.pro.rk = rk; a.yy_out = Y0; a.out2 = acc_; a.out = nullptr; a.yy_out2 = t3; a.lmul_out = lx_r.template as<T>(); a.lmul_in = ly.template as<T>();
  GenDft<T> a1{};
  if (A_next) {
    a1.in = A_next; a1.out = gx; a1.nin = Nx; a1.nout = Nx; a1.nseq = Nyh; a1.scale = 1; a1.lmul_mid = lx_r.template as<T>();
    a1.in_seq = Nx; a1.in_elem = 1; a1.in_slice = plane(); a1.out_seq = Nx; a1.out_elem = 1; a1.out_slice = plane();
  }
  hand_in<2>(a); a.out_tiled = a.in_tiled;                              // the pair in, t3 out (slice stride in_slice); Y0 / acc: the Fourier state [ky][kx] (out_slice)
  if (A_next) { hand_in<2>(a1); hand_out<2>(a1); }
  const long ws = gen_window(a, slices);
  if (A_next) (void)gen_window(a1, slices);
  a.N = a1.N = Nx; a.tw = a1.tw = genX.twN.template as<cx<T>>();
  // group height: the row-update part has Nyh / (S / 2) workgroups per slice, the d/dx part Nyh / S
  // (half-height groups while the launch has fewer than 1.5 full-height groups per CU: measured at 768^2 / 1000^2 QU -3.5 %, 768^2 T+QU +1.4 %,
  //  profiles/r06_ab_anysize_xmerge.txt)
  const int Sx = std::min((opts.gen_ct_rows && ((A_next ? 3L : 2L) * a.nseq * ws + ct_S<T>() - 1) / ct_S<T>() < 3L * num_cus / 2) ? ct_S<T>() / 2 : ct_S<T>(), ct_Smax<T>(Nx));
  a.S = a1.S = Sx;
  const int R = Sx / 2;
  const dim3 grid((unsigned)xgroups((a.nseq + R - 1) / R, a.in_tiled != 0), (unsigned)((A_next ? 2 : 1) * ws));
  switch (Nx) {
#define CMBL_X(n) case n: if (A_next) CtLaunchX<T, n>::adj_x_dx(this, a, grid, Sx, (int)ws, a1); else CtLaunchX<T, n>::adj_x(this, a, grid, Sx); return;
    CMBL_CT_LIST(CMBL_X)
#undef CMBL_X
    default: fail(ERR_STATE, "merged x passes need a compile-time plan for Nx");
  }
}

template <typename T>
void Ctx<T>::gen_x_inv_and_deriv(const cx<T>* F, cx<T>* t3, const cx<T>* A_, cx<T>* gx, cx<T>* tmp, const T* lx, long slices) {
  bool ct = opts.gen_ct && opts.gen_xderiv_fused && genX.plan.nf > 0;
  if (ct) {
    GenDft<T> a0{}, a1{};
    a0.in = F; a0.out = t3; a0.nin = Nx; a0.nout = Nx; a0.nseq = Nyh; a0.scale = 1; a0.inverse = 1;
    a0.in_seq = Nx; a0.in_elem = 1; a0.in_slice = plane(); a0.out_seq = Nx; a0.out_elem = 1; a0.out_slice = plane();
    a1 = a0; a1.in = A_; a1.out = gx; a1.inverse = 0; a1.lmul_mid = lx;
    hand_out<2>(a0); hand_in<2>(a1); hand_out<2>(a1);
    const long ws = gen_window(a0, slices);
    (void)gen_window(a1, slices);
    a0.N = a1.N = Nx; a0.tw = a1.tw = genX.twN.template as<cx<T>>();
    const int Sx = std::min(std::max(ct_S<T>() / 2, ct_rows_per_group(2L * a0.nseq * ws, 1)), ct_Smax<T>(Nx));
    a0.S = a1.S = Sx;
    const dim3 grid((unsigned)xgroups((a0.nseq + a0.S - 1) / a0.S, a0.out_tiled != 0), (unsigned)(2 * ws));
    switch (Nx) {
#define CMBL_X(n) case n: CtLaunchX<T, n>::dft2(this, a0, ct_kind(a0), grid, Sx, (int)ws, a1, ct_kind(a1)); return;
      CMBL_CT_LIST(CMBL_X)
#undef CMBL_X
      default: break;
    }
  }
  gen_x(F, t3, true, nullptr, slices, false, true);
  gen_x_deriv(A_, gx, tmp, lx, slices);
}

template <typename T>
void Ctx<T>::gen_y_adj_stage(const cx<T>* T3, T s3, const PhiMaps<T>& phm, T t, int P, cx<T>* W2a, cx<T>* W2b, long slices) {
  GenDft<T> a{};
  a.pro.ph = phm; a.pro.rk.t = t; a.pro.npix = npix(); a.pro.P = P;
  a.in = T3; a.herm = 1; a.out_real = 1; a.inverse = 1; a.nin = Nyh; a.nout = Ny; a.nseq = Nx; a.scale = s3;
  a.in_seq = 1; a.in_elem = Nx; a.in_slice = plane(); a.out_seq = Ny; a.out_elem = 1; a.out_slice = npix();
  a.yy = 3; a.yy_nout = Nyh; a.yy_out2 = W2a; a.yy_out3 = W2b;
  hand_in<1>(a);
  slices = gen_window(a, slices);
  a.N = Ny; a.tw = genY.twN.template as<cx<T>>(); a.S = std::min(ct_cols_per_group((long)a.nseq, slices, ct_S<T>()), ct_Smax<T>(Ny));
  const dim3 grid((unsigned)((a.nseq + a.S - 1) / a.S), (unsigned)slices);
  switch (Ny) {
#define CMBL_X(n) case n: CtLaunchY<T, n>::adj_y(this, a, grid); return;
    CMBL_CT_LIST(CMBL_X)
#undef CMBL_X
    default: fail(ERR_STATE, "fused y passes need a compile-time plan for Ny");
  }
}

#define CMBL_INSTANTIATE_GEN(T) \
  template bool Ctx<T>::gen_dft_ct(const GenAxis& ax, GenDft<T> a, long slices); \
  template void Ctx<T>::gen_dft(const GenAxis& ax, GenDft<T> a, long slices); \
  template void Ctx<T>::gen_y_flow_stage(const cx<T>* G1, const cx<T>* G2, const T* lmul2, T s1, T s2, const GenPro<T>& pro, cx<T>* Anext, bool last, long slices); \
  template void Ctx<T>::gen_y_delta_stage(const cx<T>* T3, T s3, const cx<T>* G1, const cx<T>* G2, const T* lmul2, T s1, T s2, const GenPro<T>& pro, cx<T>* Anext, cx<T>* W2a, cx<T>* W2b, bool last, long slices); \
  template void Ctx<T>::gen_y_adj_stage(const cx<T>* T3, T s3, const PhiMaps<T>& phm, T t, int P, cx<T>* W2a, cx<T>* W2b, long slices);
// (the x-side launches: instantiated by the same units since the kernels moved behind CtLaunchX, engine_ct.hpp)
#define CMBL_INSTANTIATE_GENX(T) \
  template bool Ctx<T>::gen_x_adj_update(const cx<T>* W2a, const cx<T>* W2b, cx<T>* Y0, cx<T>* acc_, cx<T>* Ys, const RKCoef<T>& rk, long slices); \
  template void Ctx<T>::gen_x_adj_next(const cx<T>* W2a, const cx<T>* W2b, cx<T>* Y0, cx<T>* acc_, const RKCoef<T>& rk, cx<T>* t3, const cx<T>* A_next, cx<T>* gx, long slices); \
  template void Ctx<T>::gen_x_inv_and_deriv(const cx<T>* F, cx<T>* t3, const cx<T>* A_, cx<T>* gx, cx<T>* tmp, const T* lx, long slices);

}  // namespace cmbl
