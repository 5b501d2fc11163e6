// The launches of the compile-time-plan kernels of one length (CtLaunchY / CtLaunchX, declared in engine.hpp): the only code that references
// k_ct_* -- two thirds of the library's device code.  Instantiated per length by tu_cty_{f32,f64}_{a,b}.hip and tu_ctx_{f32,f64}_{a,b}.hip over the
// two halves of CMBL_CT_LIST (api_decl.hpp has the map of the build).
#pragma once
#include "engine.hpp"

namespace cmbl {

template <typename T, int N>
void CtLaunchY<T, N>::dftx(Ctx<T>* c, const GenDft<T>& a, dim3 grid, int Sx, int kind) {
  constexpr bool FULL = ct_Smax<T>(N) == ct_S<T>();   // the full group of ct_S sequences fits the LDS (the host never asks for more than ct_Smax)
  if constexpr (FULL) { if (Sx == ct_S<T>()) { CMBL_LAUNCH_NT(c, K_GEN_DFT, 64 * ct_S<T>(), (k_ct_dftx<T, N, ct_S<T>()>), grid, ct_lds<T>(N), c->stream, a, kind); return; } }
  if (Sx == ct_S<T>() / 2) CMBL_LAUNCH_NT(c, K_GEN_DFT, 32 * ct_S<T>(), (k_ct_dftx<T, N, ct_S<T>() / 2>), grid, (ct_lds<T>(N, 1, ct_S<T>() / 2)), c->stream, a, kind);
  else CMBL_LAUNCH_NT(c, K_GEN_DFT, 16 * ct_S<T>(), (k_ct_dftx<T, N, ct_S<T>() / 4>), grid, (ct_lds<T>(N, 1, ct_S<T>() / 4)), c->stream, a, kind);
}
template <typename T, int N>
void CtLaunchY<T, N>::dft(Ctx<T>* c, const GenDft<T>& a, dim3 grid, int kind) {
  CMBL_LAUNCH_NT(c, K_GEN_DFT, 64 * ct_Smax<T>(N), (k_ct_dft<T, N>), grid, (ct_lds<T>(N, 1, ct_Smax<T>(N))), c->stream, a, kind);
}
template <typename T, int N>
void CtLaunchY<T, N>::flow_y(Ctx<T>* c, const GenDft<T>& a, dim3 grid) {
  constexpr bool FULL = ct_Smax<T>(N) == ct_S<T>();
  if constexpr (FULL) { if (a.S == ct_S<T>()) { CMBL_LAUNCH_NT(c, K_GEN_DFT, 64 * ct_S<T>(), (k_ct_flow_y<T, N, ct_S<T>()>), grid, ct_lds<T>(N), c->stream, a); return; } }
  CMBL_LAUNCH_NT(c, K_GEN_DFT, 32 * ct_S<T>(), (k_ct_flow_y<T, N, ct_S<T>() / 2>), grid, (ct_lds<T>(N, 1, ct_S<T>() / 2)), c->stream, a);
}
template <typename T, int N>
bool CtLaunchY<T, N>::delta_y(Ctx<T>* c, const GenDft<T>& a, dim3 grid) {
  if constexpr (ct_lds<T>(N, 2, ct_S2<T>(N)) <= 160 * 1024) {
    if (a.S == ct_S2<T>(N)) CMBL_LAUNCH_NT(c, K_GEN_DFT, 128 * ct_S2<T>(N), (k_ct_delta_y<T, N, ct_S2<T>(N)>), grid, ct_lds<T>(N, 2, ct_S2<T>(N)), c->stream, a);
    else if constexpr (ct_S2<T>(N) == ct_S<T>()) CMBL_LAUNCH_NT(c, K_GEN_DFT, 64 * ct_S<T>(), (k_ct_delta_y<T, N, ct_S<T>() / 2>), grid, (ct_lds<T>(N, 2, ct_S<T>() / 2)), c->stream, a);
    return true;
  } else { (void)c; (void)a; (void)grid; return false; }
}
template <typename T, int N>
void CtLaunchY<T, N>::adj_y(Ctx<T>* c, const GenDft<T>& a, dim3 grid) {
  constexpr bool FULL = ct_Smax<T>(N) == ct_S<T>();
  if constexpr (FULL) { if (a.S == ct_S<T>()) { CMBL_LAUNCH_NT(c, K_GEN_DFT, 64 * ct_S<T>(), (k_ct_adj_y<T, N, ct_S<T>()>), grid, ct_lds<T>(N), c->stream, a); return; } }
  CMBL_LAUNCH_NT(c, K_GEN_DFT, 32 * ct_S<T>(), (k_ct_adj_y<T, N, ct_S<T>() / 2>), grid, (ct_lds<T>(N, 1, ct_S<T>() / 2)), c->stream, a);
}

template <typename T, int N>
void CtLaunchX<T, N>::adj_x(Ctx<T>* c, const GenDft<T>& a, dim3 grid, int Sx) {
  constexpr bool FULL = ct_Smax<T>(N) == ct_S<T>();
  if constexpr (!ct_rowfuse_ok<T>(N)) { (void)a; (void)grid; (void)Sx; (void)c; fail(ERR_STATE, "no fused row update at this length and precision"); } else {
  if constexpr (FULL) { if (Sx == ct_S<T>()) { CMBL_LAUNCH_NT(c, K_GEN_DFT, 64 * ct_S<T>(), (k_ct_adj_x<T, N, ct_S<T>()>), grid, ct_lds<T>(N), c->stream, a); return; } }
  CMBL_LAUNCH_NT(c, K_GEN_DFT, 32 * ct_S<T>(), (k_ct_adj_x<T, N, ct_S<T>() / 2>), grid, (ct_lds<T>(N, 1, ct_S<T>() / 2)), c->stream, a);
  }
}
template <typename T, int N>
void CtLaunchX<T, N>::adj_x_dx(Ctx<T>* c, const GenDft<T>& a, dim3 grid, int Sx, int ws, const GenDft<T>& a1) {
  constexpr bool FULL = ct_Smax<T>(N) == ct_S<T>();
  if constexpr (!ct_rowfuse_ok<T>(N)) { (void)a; (void)grid; (void)Sx; (void)ws; (void)a1; (void)c; fail(ERR_STATE, "no fused row update at this length and precision"); } else {
  if constexpr (FULL) { if (Sx == ct_S<T>()) { CMBL_LAUNCH_NT(c, K_GEN_DFT, 64 * ct_S<T>(), (k_ct_adj_x_dx<T, N, ct_S<T>()>), grid, ct_lds<T>(N), c->stream, a, ws, a1); return; } }
  CMBL_LAUNCH_NT(c, K_GEN_DFT, 32 * ct_S<T>(), (k_ct_adj_x_dx<T, N, ct_S<T>() / 2>), grid, (ct_lds<T>(N, 1, ct_S<T>() / 2)), c->stream, a, ws, a1);
  }
}
template <typename T, int N>
void CtLaunchX<T, N>::dft2(Ctx<T>* c, const GenDft<T>& a0, int kind0, dim3 grid, int Sx, int ws, const GenDft<T>& a1, int kind1) {
  constexpr bool FULL = ct_Smax<T>(N) == ct_S<T>();
  if constexpr (FULL) { if (Sx == ct_S<T>()) { CMBL_LAUNCH_NT(c, K_GEN_DFT, 64 * ct_S<T>(), (k_ct_dft2<T, N, ct_S<T>()>), grid, ct_lds<T>(N), c->stream, a0, kind0, ws, a1, kind1); return; } }
  CMBL_LAUNCH_NT(c, K_GEN_DFT, 32 * ct_S<T>(), (k_ct_dft2<T, N, ct_S<T>() / 2>), grid, (ct_lds<T>(N, 1, ct_S<T>() / 2)), c->stream, a0, kind0, ws, a1, kind1);
}

}  // namespace cmbl
