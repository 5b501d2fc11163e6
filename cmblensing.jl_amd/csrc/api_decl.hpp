// What api.hip (the entry points) and the per-precision translation units share: the handle structs and the DECLARATIONS of the typed
// bodies behind the entry points.  The build is split by explicit instantiation (lib.py builds the objects in parallel):
//   api.hip                 extern "C" entry points, argument checks, error plumbing -- instantiates NO kernel
//   tu_main_{f32,f64}.hip   every do_*<T> below (api_body.hpp) and with them Ctx<T>, Flow<T>, Dataset<T>, Drivers<T> and their kernels
//   tu_gen_{f32,f64}.hip    the host side of the any-size transform launches (engine_gen.hpp) and the run-time-plan kernels k_gen_dft*
//   tu_cty_{f32,f64}_{a,b}.hip   the compile-time-plan kernels of the column side and the plain transforms (engine_ct.hpp CtLaunchY: k_ct_dft,
//                           k_ct_dftx, k_ct_flow_y, k_ct_delta_y, k_ct_adj_y), lengths of CMBL_CT_LIST_A / _B (kernels_ct.hpp)
//   tu_ctx_{f32,f64}_{a,b}.hip   ... and of the row side of the fused stages (CtLaunchX: k_ct_adj_x, k_ct_adj_x_dx, k_ct_dft2)
//   tu_small_{f32,f64}.hip  the one-launch flows of small maps (engine_small.hpp: k_small_flow, k_small_adj)
// Rule that keeps api.hip free of kernels: it must not ODR-use a member function that launches (members defined in class are inline, and
// an explicit instantiation DECLARATION does not stop inline functions from being instantiated -- [temp.explicit]/10); it calls do_*<T>
// only, which are declared here and defined in api_body.hpp.
#pragma once
#include "engine.hpp"
#include "drivers.hpp"
#include "../../include/cmblens.h"

struct cmbl_ctx { std::unique_ptr<cmbl::CtxBase> p; };
struct cmbl_flow { cmbl_ctx* ctx; std::unique_ptr<cmbl::Flow<float>> f32; std::unique_ptr<cmbl::Flow<double>> f64; };
struct cmbl_dataset {
  cmbl_ctx* ctx; std::unique_ptr<cmbl::Dataset<float>> f32; std::unique_ptr<cmbl::Dataset<double>> f64;
  std::map<const void*, std::unique_ptr<cmbl::Drivers<float>>> drv32;        // driver scratch per (dataset, flow) pair
  std::map<const void*, std::unique_ptr<cmbl::Drivers<double>>> drv64;
  std::vector<std::unique_ptr<cmbl::DevBuf>> qe_pool;                        // legs and products of cmbl_quadratic_estimate, reused between calls
};

namespace cmbl {
template <typename T> void do_convert(cmbl_ctx* ctx, int bi, const void* in, int bo, void* out, int P, int B);
template <typename T> void do_diag(cmbl_ctx* ctx, int kind, int bd, const void* diag, int nplanes, bool transpose, int bi, const void* in, int bo, void* out, int P, int B);
template <typename T> void do_dot(cmbl_ctx* ctx, int basis, const void* a, const void* b, int P, int B, double* out);
template <typename T> void do_diag_reduce(cmbl_ctx* ctx, int which, int basis, const void* d, int P, int B, double* out);
template <typename T> void do_logdet(cmbl_ctx* ctx, const void* d, int nplanes, double* out);
template <typename T> void do_gradf(cmbl_dataset* dsh, cmbl_flow* Lh, const void* f, const void* d, int zero_d, void* out, int B);
template <typename T> void do_cg(cmbl_dataset* dsh, cmbl_flow* Lh, const void* d, const void* fstart, double tol, int maxit, void* f_out, double* hist, int* nit, int B);
template <typename T> void do_lpm(cmbl_dataset* dsh, cmbl_flow* Lh, const void* fo, const void* phio, double* lp, void* gfo, void* gphio, int B, int quirk);
template <typename T> void do_hmc(cmbl_dataset* dsh, cmbl_flow* Lh, const void* fo, const void* phio, const void* mass, const void* white_p, const double* log_u, const uint64_t* seeds, uint64_t step, int nleap, double eps, int always, int quirk, int B, void* phio_out, double* dH, int* accept);
template <typename T> void do_map_step(cmbl_dataset* dsh, cmbl_flow* Lh, const void* phi, const void* fstart, const void* hinv, double amax, double atol, double cg_tol, int cg_maxit, int quirk, int B, void* f_out, void* phi_out, double* logpdf, double* alpha, int* ncg, int* nls);
template <typename T> void do_qe(cmbl_dataset* dsh, int which, const double* Cf, const double* Cft, const double* Cn, const double* TF, const double* Cphi, int wiener, const double* AL_in, void* phiqe_out, double* AL_out, int B);
template <typename T> CtxBase* do_ctx_create(int Ny, int Nx, double theta, int device, void* stream);
template <typename T> void do_axpby(cmbl_ctx* ctx, const double* a, const void* x, const double* b, const void* y, void* out, long n, int B);
template <typename T> void do_qe_leg(cmbl_ctx* ctx, const void* in_fourier, int n, int p1, int p2, void* out_map, int B);
template <typename T> void do_fourier_lmul(cmbl_ctx* ctx, const void* in_map, int p1, int p2, int take_abs, void* out_fourier, int B);
template <typename T> void do_map_fma(cmbl_ctx* ctx, const void* a, const void* b, double scale, void* out, int accumulate, long n);
template <typename T> void do_randn(cmbl_ctx* ctx, const uint64_t* seeds, int nslots, uint64_t stream, void* out, long n_per_slot);
template <typename T> void do_flow_create(cmbl_flow* h, int nsteps);
template <typename T> void do_flow_set_phi(cmbl_flow* L, int basis, const void* phi, int nb);
template <typename T> void do_flow_apply(cmbl_flow* L, int mode, int bi, const void* in, int bo, void* out, int P, int B);
template <typename T> void do_flow_grad(cmbl_flow* L, int mode, const void* f_end, int bdel, const void* delta, void* dphi, int bdf, void* df, void* f_start, int P, int B, int quirk);
template <typename T> void do_max_lensing_step(cmbl_flow* L, int basis, const void* phi, const void* eta, int nb, double* out);
template <typename T> void do_dataset_create(cmbl_dataset* h, int npol);
template <typename T> void do_dataset_set_op(cmbl_dataset* ds, int which, const void* planes, int nplanes);
template <typename T> void do_dataset_set_data(cmbl_dataset* ds, const void* d, int B);

#define CMBL_INSTANTIATE_API(T) \
  template void do_convert<T>(cmbl_ctx* ctx, int bi, const void* in, int bo, void* out, int P, int B); \
  template void do_diag<T>(cmbl_ctx* ctx, int kind, int bd, const void* diag, int nplanes, bool transpose, int bi, const void* in, int bo, void* out, int P, int B); \
  template void do_dot<T>(cmbl_ctx* ctx, int basis, const void* a, const void* b, int P, int B, double* out); \
  template void do_diag_reduce<T>(cmbl_ctx* ctx, int which, int basis, const void* d, int P, int B, double* out); \
  template void do_logdet<T>(cmbl_ctx* ctx, const void* d, int nplanes, double* out); \
  template void do_gradf<T>(cmbl_dataset* dsh, cmbl_flow* Lh, const void* f, const void* d, int zero_d, void* out, int B); \
  template void do_cg<T>(cmbl_dataset* dsh, cmbl_flow* Lh, const void* d, const void* fstart, double tol, int maxit, void* f_out, double* hist, int* nit, int B); \
  template void do_lpm<T>(cmbl_dataset* dsh, cmbl_flow* Lh, const void* fo, const void* phio, double* lp, void* gfo, void* gphio, int B, int quirk); \
  template void do_hmc<T>(cmbl_dataset* dsh, cmbl_flow* Lh, const void* fo, const void* phio, const void* mass, const void* white_p, const double* log_u, const uint64_t* seeds, uint64_t step, int nleap, double eps, int always, int quirk, int B, void* phio_out, double* dH, int* accept); \
  template void do_map_step<T>(cmbl_dataset* dsh, cmbl_flow* Lh, const void* phi, const void* fstart, const void* hinv, double amax, double atol, double cg_tol, int cg_maxit, int quirk, int B, void* f_out, void* phi_out, double* logpdf, double* alpha, int* ncg, int* nls); \
  template void do_qe<T>(cmbl_dataset* dsh, int which, const double* Cf, const double* Cft, const double* Cn, const double* TF, const double* Cphi, int wiener, const double* AL_in, void* phiqe_out, double* AL_out, int B); \
  template CtxBase* do_ctx_create<T>(int Ny, int Nx, double theta, int device, void* stream); \
  template void do_axpby<T>(cmbl_ctx* ctx, const double* a, const void* x, const double* b, const void* y, void* out, long n, int B); \
  template void do_qe_leg<T>(cmbl_ctx* ctx, const void* in_fourier, int n, int p1, int p2, void* out_map, int B); \
  template void do_fourier_lmul<T>(cmbl_ctx* ctx, const void* in_map, int p1, int p2, int take_abs, void* out_fourier, int B); \
  template void do_map_fma<T>(cmbl_ctx* ctx, const void* a, const void* b, double scale, void* out, int accumulate, long n); \
  template void do_randn<T>(cmbl_ctx* ctx, const uint64_t* seeds, int nslots, uint64_t stream, void* out, long n_per_slot); \
  template void do_flow_create<T>(cmbl_flow* h, int nsteps); \
  template void do_flow_set_phi<T>(cmbl_flow* L, int basis, const void* phi, int nb); \
  template void do_flow_apply<T>(cmbl_flow* L, int mode, int bi, const void* in, int bo, void* out, int P, int B); \
  template void do_flow_grad<T>(cmbl_flow* L, int mode, const void* f_end, int bdel, const void* delta, void* dphi, int bdf, void* df, void* f_start, int P, int B, int quirk); \
  template void do_max_lensing_step<T>(cmbl_flow* L, int basis, const void* phi, const void* eta, int nb, double* out); \
  template void do_dataset_create<T>(cmbl_dataset* h, int npol); \
  template void do_dataset_set_op<T>(cmbl_dataset* ds, int which, const void* planes, int nplanes); \
  template void do_dataset_set_data<T>(cmbl_dataset* ds, const void* d, int B);
}  // namespace cmbl
