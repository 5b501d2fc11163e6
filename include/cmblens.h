/* libcmblens_hip.so -- C ABI of the MI355X-native flat-sky lensing field engine.
 *
 * Drop-in boundary for the LenseFlow / Wiener-filter hot path of marius311/CMBLensing.jl
 * (reference @ v0.10.1; citations are `file:line` under /root/reference).  The reference has no
 * FFI for this path -- its "plugin" surface is the array storage type `A` of `BaseField{B,M,T,A}`
 * (src/base_fields.jl:14) plus the operator slot `ds.L` (src/dataset.jl:55).  Each entry point
 * below replaces the reference method(s) it cites; INTEGRATION.md shows the Julia `ccall` glue.
 *
 * Conventions
 *   - every function returns 0 (CMBL_OK) or a CMBL_ERR_* code; cmbl_last_error() gives the text
 *     (thread-local).  Nothing throws across the ABI.
 *   - all field pointers are DEVICE pointers (HIP) unless the name ends in `_host`.
 *   - array layouts are the reference's (src/proj_cartesian.jl:13-36, column-major, Ny fastest):
 *       map     : real    (Ny, Nx, npol, nbatch)
 *       fourier : complex (Ny/2+1, Nx, npol, nbatch)   interleaved (re,im)
 *     npol = 1 (I), 2 (QU), 3 (IQU).  Operators diagonal in l are real (Ny/2+1, Nx) planes.
 *   - dtype: CMBL_F32 or CMBL_F64 (the whole context works in one precision).
 *   - Ny, Nx: any integers in [2, 4096], like the reference's FFTW plans (src/util_fft.jl:32-35).  Powers of two >= 32 on both
 *     sides run the fused kernels; other sizes the any-size path (mixed-radix / chirp-z transforms, csrc/kernels_generic.hpp),
 *     same results, ~5x slower per pixel.  Sides above 4096 return CMBL_ERR_SHAPE.
 *   - nbatch (chains / simulations as batch slots of one call): up to 256 per call for the entry points that return per-slot scalars
 *     (reductions, cmbl_wiener_cg, cmbl_logpdf_mixed, cmbl_grad_logpdf_mixed); more return CMBL_ERR_ARG.  The flows have no such limit.
 *   - a handle is used by one host thread at a time; different contexts are independent.
 *   - calls are asynchronous on the context's stream unless they return host values (`*_host` outputs), i.e. every
 *     field-to-field entry point is already the `_async` form; cmbl_ctx_synchronize() waits for the stream.
 */
#ifndef CMBLENS_H
#define CMBLENS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cmbl_ctx cmbl_ctx;
typedef struct cmbl_flow cmbl_flow;
typedef struct cmbl_dataset cmbl_dataset;

enum { CMBL_OK = 0, CMBL_ERR_ARG = 1, CMBL_ERR_SHAPE = 2, CMBL_ERR_HIP = 3, CMBL_ERR_NAN = 4,
       CMBL_ERR_STATE = 5, CMBL_ERR_ALLOC = 6 };
enum { CMBL_F32 = 0, CMBL_F64 = 1 };

/* bases (src/generic.jl:42-98): MAP = Map/QUMap/IQUMap (LenseBasis), FOURIER = Fourier/QUFourier/
 * IQUFourier (DerivBasis), HARMONIC = Fourier/EBFourier/IEBFourier (basis covariances are diagonal in) */
enum { CMBL_MAP = 0, CMBL_FOURIER = 1, CMBL_HARMONIC = 2 };

/* LenseFlow operator modes (src/flowops.jl:11-14) */
enum { CMBL_FLOW_FWD = 0,     /* L * f   : velocity,  t 0->1 */
       CMBL_FLOW_INV = 1,     /* L \ f   : velocity,  t 1->0 */
       CMBL_FLOW_ADJ = 2,     /* L' * f  : velocityH, t 1->0 */
       CMBL_FLOW_INVADJ = 3   /* L' \ f  : velocityH, t 0->1 */ };

/* diagonal operator application kinds (src/specialops.jl:9-10) */
enum { CMBL_DIAG_MUL = 1, CMBL_DIAG_DIV_NAN2ZERO = 3 };

/* ABI revision of this header.  Bumped whenever an existing entry point changes its signature or meaning (additions do not bump
 * it): 2 = cmbl_device_malloc / cmbl_device_free take the context (round 3); 3 = behaviour switches are read from the environment
 * once per context and changed through cmbl_ctx_set_option (round 4).  A caller compiled against this header checks
 * cmbl_abi_version() == CMBL_ABI_VERSION before anything else (julia/CMBLensingHIPExt.jl __init__, tests/c_abi/ *.c). */
#define CMBL_ABI_VERSION 3

const char* cmbl_last_error(void);
int cmbl_version(void);          /* library release, 100 * major + minor */
int cmbl_abi_version(void);      /* the CMBL_ABI_VERSION the library was built with */

/* ---- context: replaces the memoized ProjLambert + FFT plans
 *      (src/proj_lambert.jl:48-75, src/util_fft.jl:32-39).  `stream` is a hipStream_t the caller owns
 *      (e.g. torch's current stream); NULL is the legacy default stream. */
int cmbl_ctx_create(int Ny, int Nx, double theta_pix_arcmin, int dtype, int device, void* stream, cmbl_ctx** out);
int cmbl_ctx_destroy(cmbl_ctx* ctx);
int cmbl_ctx_synchronize(cmbl_ctx* ctx);
/* geometry queries, host output, double: which = 0 lx[Nx], 1 ly[Ny/2+1], 2 lambda_rfft[Ny/2+1],
 * 3 sin2phi[(Ny/2+1)*Nx], 4 cos2phi[...], 5 lmag[...]  (planes in the reference layout) */
int cmbl_ctx_geometry_host(cmbl_ctx* ctx, int which, double* out_host, size_t n);

/* ---- behaviour switches of a context (A/B and profiling aids; none changes results beyond rounding).  Each starts from the
 *      environment variable named below, read ONCE when the context is created, and afterwards changes only through this call:
 *        "slice_streams"         CMBL_SLICE_STREAMS (4)            launch chains per flow: pol slices / batch-slot groups on their own streams; 1 = one launch over all slices
 *        "slice_streams_min_pix" CMBL_SLICE_STREAMS_MIN_PIX (2^19) pixels a launch chain of a delta flow must carry (all its slices together); map and adjoint flows, and three or more chains: twice that
 *        "pcache"                !CMBL_NO_PCACHE (1)               cache p(t_k) at the 2n+1 stage times per phi (src/lenseflow.jl:45-46,131-142); read by cmbl_lenseflow_set_phi
 *        "pcache_max_mb"         CMBL_PCACHE_MAX_MB (16384)
 *        "fused_harm"            !CMBL_NO_FUSED_HARM (1)           harmonic-space operator chains inside one row pass
 *        "gen_separable", "gen_prologue", "gen_xderiv_fused"       any-size path stage fusions (CMBL_GEN_SEPARABLE / _PROLOGUE / _XDERIV_FUSED, all 1)
 *        "gen_ct"                                                   any-size path: compile-time-plan transforms for the lengths 2^a 3^b 5^c of
 *                                                                   CMBL_CT_LIST (CMBL_GEN_CT, 1; 0 = the run-time-planned kernel for every length)
 *        "gen_ct_rows"                                              any-size path: x-pass launches with fewer row groups than CUs take groups of 4 / 2 rows instead
 *                                                                   of 8 (CMBL_GEN_CT_ROWS, 1; results bit-identical either way)
 *        "gen_ct_cols"                                              any-size flows: half-width column groups (4 instead of 8 columns in single precision) in the fused y
 *                                                                   launches: 0 never, 1 (default) for launches below 0.4 workgroups per CU, 2 always (CMBL_GEN_CT_COLS;
 *                                                                   results bit-identical)
 *        "gen_xmerge"                                               any-size flows: the row update that closes an adjoint-type stage also runs the x passes that open the
 *                                                                   next stage (CMBL_GEN_XMERGE, 1: 2 instead of 3 launches per stage; results bit-identical either way)
 *        "gen_tiled"                                                any-size flows (both axes with a compile-time plan): the half planes the fused stages hand between their
 *                                                                   column and row launches are stored as [x/4][ky][x%4] blocks instead of [ky][x]; bit mask 1 = map flows,
 *                                                                   2 = adjoint flows, 4 = delta flows, 8 = the scratch of the 2-D basis transforms (CMBL_GEN_TILED, 7: bit 8 measured neutral; results bit-identical to 0)
 *        "gen_yy"                                                   any-size flows: the passes of a stage that can share a launch do, where the axes have
 *                                                                   compile-time plans (CMBL_GEN_YY, 1; results bit-identical either way)
 *        "gen_slice_streams", "gen_streams_min_pix"                 any-size flows: launch chains over groups of slices when every chain carries at least this many
 *                                                                   4-byte pixels, a third more from three chains on (CMBL_GEN_SLICE_STREAMS 1, CMBL_GEN_STREAMS_MIN_PIX 300000)
 *        "occupancy_tiles"       CMBL_OCCUPANCY_TILES (3)          small maps: bit 0 = two-column tiles when four-column tiles leave CUs idle or unevenly loaded, bit 1 = shorter row groups
 *        "fill_target"           CMBL_FILL_TARGET (0)              > 0: narrow the column tiles below that many tiles per launch instead of the built-in rule
 *        "row_fill_target"       CMBL_ROW_FILL_TARGET (0 = CUs/2)  shorten the row groups below that many groups per launch
 *        "col_prefetch"          CMBL_COL_PREFETCH (-1)            touch prefetch of the double-precision >= 2048-row column kernels: -1 = built-in distance, 0 = off, > 0 = blocks ahead
 *        "col_pipeline"          CMBL_COL_PIPELINE (1)             only in -DCMBL_EXPERIMENT_COL_PIPELINE builds (the measured-and-rejected two-tile column workgroup,
 *                                                                   profiles/r0{5,6}_ab_col_pipeline_rejected.txt): 0 off, 1 on, 2 on the 256-thread tile; no effect in the shipped library
 *        "small_flow"            CMBL_SMALL_FLOW (1)               maps of 32..128 pixels per side: L*f, L\f, L'g, L'\g as ONE launch, one workgroup per (pol, batch) slice with
 *                                                                   the half plane resident in LDS (csrc/kernels_small.hpp): 0 = off, 1 = up to 64 x 64 pixels (faster at every
 *                                                                   batch size), 2 = wherever compiled (up to 128 x 128 in single, 64 x 64 in double precision).  Results agree
 *                                                                   with the staged path to rounding, not bit for bit (tests/test_gpu_small.py)
 *      (the launch-geometry and prefetch switches change no result at all: tests/test_gpu_boundary.py, tests/test_gpu_fullsize.py)
 *      Unknown names return CMBL_ERR_ARG.  The reference has no counterpart (its switches are Julia keyword arguments). */
int cmbl_ctx_set_option(cmbl_ctx* ctx, const char* name, int value);
int cmbl_ctx_get_option(cmbl_ctx* ctx, const char* name, int* value_host);

/* ---- optional per-launch timing of the library's kernel classes with HIP events on the context's stream
 *      (the reference wraps the same call sites in TimerOutputs `@⌛`, src/util.jl:351-390).  Disabled by default. */
int cmbl_prof_enable(cmbl_ctx* ctx, int on);
int cmbl_prof_reset(cmbl_ctx* ctx);
int cmbl_prof_count(void);
const char* cmbl_prof_name(int kernel_class);
int cmbl_prof_get(cmbl_ctx* ctx, int kernel_class, double* total_ms_host, long* launches_host);

/* report of the accumulated timings as text ("name launches total_ms mean_us" lines), like TimerOutputs' table; returns the
 * number of bytes the full report needs (excluding the terminator) -- call with buf = NULL to size the buffer. */
int cmbl_timer_report(cmbl_ctx* ctx, char* buf, size_t buflen);

/* ---- device memory helpers for callers that do not link HIP themselves (a Julia process without AMDGPU.jl, the plain-C
 *      test): every field pointer of this API is a device pointer; these give a host program the means to own some.
 *      Buffers live on the context's device (the call selects it: the caller cannot, it does not link HIP).
 *      Copies are ordered on the context's stream and complete on return. */
int cmbl_device_malloc(cmbl_ctx* ctx, size_t bytes, void** out);
int cmbl_device_free(cmbl_ctx* ctx, void* p);
int cmbl_copy_to_device(cmbl_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);
int cmbl_copy_to_host(cmbl_ctx* ctx, void* dst_host, const void* src_device, size_t bytes);

/* ---- basis transforms: m_rfft / m_irfft and the Basis conversion lattice
 *      (src/util_fft.jl:20-31, src/proj_lambert.jl:245-300) */
int cmbl_rfft(cmbl_ctx* ctx, const void* map, void* fourier, int npol, int nbatch);
int cmbl_irfft(cmbl_ctx* ctx, const void* fourier, void* map, int npol, int nbatch);
int cmbl_convert(cmbl_ctx* ctx, int basis_in, const void* in, int basis_out, void* out, int npol, int nbatch);

/* ---- diagonal operators: DiagOp `*` and `\` with automatic basis conversion, BlockDiagIEB
 *      (src/specialops.jl:9-10, 61-118; src/field_vectors.jl:64-66).
 *      diag: real (Ny/2+1, Nx, npol) planes, diagonal in `basis_diag` (FOURIER or HARMONIC). */
int cmbl_diag_apply(cmbl_ctx* ctx, int kind, int basis_diag, const void* diag,
                    int basis_in, const void* in, int basis_out, void* out, int npol, int nbatch);
/* te_bb: 5 real planes (TT, TE, ET, EE, BB): (i,e) = [TT TE; ET EE](I,E), b = BB*B */
int cmbl_blockdiag_ieb_apply(cmbl_ctx* ctx, const void* te_bb, int transpose,
                             int basis_in, const void* in, int basis_out, void* out, int nbatch);

/* ---- per-batch reductions: dot, logdet (src/proj_lambert.jl:318-342) */
int cmbl_dot(cmbl_ctx* ctx, int basis, const void* a, const void* b, int npol, int nbatch, double* out_host);
int cmbl_logdet(cmbl_ctx* ctx, const void* diag_fourier, int nplanes, double* out_host);
/* norm(f) = sqrt(dot(f,f)) (src/generic.jl:373), one value per batch slot */
int cmbl_norm(cmbl_ctx* ctx, int basis, const void* a, int npol, int nbatch, double* out_host);
/* logdet / tr of Diagonal(field) (src/proj_lambert.jl:331-353), per batch slot.  basis MAP: `diag` is a real map,
 * logdet = sum log|d| + log(prod sign d) -- NaN for an odd number of negative entries (Julia's log(-1.0) throws), -Inf with a
 * zero entry; FOURIER / HARMONIC: `diag` is a complex half-plane field, logdet = sum lambda_rfft * log|d| (non-finite terms
 * dropped), tr = sum lambda_rfft * Re d. */
int cmbl_logdet_diag(cmbl_ctx* ctx, int basis, const void* diag, int npol, int nbatch, double* out_host);
int cmbl_tr_diag(cmbl_ctx* ctx, int basis, const void* diag, int npol, int nbatch, double* out_host);
/* set_sum_accuracy_mode! (src/util.jl:288-316) for every reduction of this context (dot, norm, logdet, tr, the quadratic forms
 * of logpdf, the conjugate-gradient residuals).  Every term is formed in the working precision like the reference's broadcast;
 * the mode selects how the terms are added: WORKING = plain sum in the working precision (the reference's default `nothing`),
 * FLOAT64 = sum(Float64.(A)), KAHAN = compensated (sum_kbn).  The engine's default is FLOAT64: the reductions are HBM-bound
 * and double accumulation is free, whereas fp32 accumulation of ~1e6 terms loses the O(1) differences HMC accepts on. */
enum { CMBL_SUM_WORKING = 0, CMBL_SUM_FLOAT64 = 1, CMBL_SUM_KAHAN = 2 };
int cmbl_set_sum_accuracy_mode(cmbl_ctx* ctx, int mode);

/* ---- LenseFlow: LenseFlow / CachedLenseFlow, precompute!!, the four flow operators and the two
 *      Zygote pullbacks (src/lenseflow.jl:19-214, src/flowops.jl:11-14, 40-68) */
int cmbl_lenseflow_create(cmbl_ctx* ctx, int nsteps, cmbl_flow** out);
int cmbl_lenseflow_destroy(cmbl_flow* L);
/* precompute!(L): phi in `basis` (MAP or FOURIER), (.., 1, nbatch_phi) */
int cmbl_lenseflow_set_phi(cmbl_flow* L, int basis, const void* phi, int nbatch_phi);
int cmbl_lenseflow_apply(cmbl_flow* L, int mode, int basis_in, const void* in, int basis_out, void* out,
                         int npol, int nbatch);
/* pullback of  mode=FWD: ftilde = L*f   (delta flow t 1->0 from (ftilde, delta, 0))
 *              mode=INV: f = L\ftilde   (delta flow t 0->1 from (f, delta, 0)).
 * f_end: the OUTPUT of the primal op (MAP basis); delta: cotangent in basis_delta.
 * outputs: dphi (FOURIER, (Ny/2+1,Nx,1,nbatch)), df (basis_df), f_start (MAP; may be NULL).
 * alias_quirk != 0 reproduces the reference's in-place aliasing (src/lenseflow.jl:198-200 with
 * src/field_vectors.jl:48-49); 0 gives the mathematically consistent gradient.
 * dphi is formed as the RK4 quadrature sum over all stages (its velocity never depends on dphi itself), which equals the reference's
 * stage-by-stage update up to the order of floating-point summation.  The handle keeps 4*nsteps*2 maps of scratch per (pol,batch)
 * slice for it (448 MB at 1024^2 QU fp32, nsteps = 7).  Streams: the handle owns a few internal streams that it forks from / joins
 * into the context's stream inside a call (independent pol slices / batch groups run as concurrent launch chains); on return all
 * work is ordered on the context's stream as for every other entry point. */
int cmbl_lenseflow_grad(cmbl_flow* L, int mode, const void* f_end, int basis_delta, const void* delta,
                        void* dphi_out, int basis_df, void* df_out, void* f_start_out,
                        int npol, int nbatch, int alias_quirk);

/* get_max_lensing_step(phi, eta) (src/lenseflow.jl:242-256): largest alpha keeping I + grad grad(phi + alpha eta)
 * non-singular, one value per batch slot. */
int cmbl_max_lensing_step(cmbl_flow* L, int basis, const void* phi, const void* eta, int nbatch, double* out_host);

/* ---- small helpers used by the drivers above the hot kernels
 * axpby: out = a[b]*x + b[b]*y per batch slot (y may be NULL) -- the FieldTuple / Field broadcasts of the CG, line-search
 *        and leapfrog updates (src/numerical_algorithms.jl:102-107, src/sampling.jl:29-31).
 * qe_leg: Map(nan2zero(in * (i lx)^p1 (i ly)^p2 / |l|^n)) (src/quadratic_estimate.jl:89-91), in: Fourier S0.
 * fourier_lmul: (i lx)^p1 (i ly)^p2 * rfft(map) or (take_abs) its modulus in the real part (src/quadratic_estimate.jl:97,118).
 * map_fma: out = [out +] scale * a * b on maps (products of legs). */
int cmbl_axpby(cmbl_ctx* ctx, int basis, const double* a_host, const void* x, const double* b_host, const void* y, void* out,
               int npol, int nbatch);
int cmbl_qe_leg(cmbl_ctx* ctx, const void* in_fourier, int n, int p1, int p2, void* out_map, int nbatch);
int cmbl_fourier_lmul(cmbl_ctx* ctx, const void* in_map, int p1, int p2, int take_abs, void* out_fourier, int nbatch);
int cmbl_map_fma(cmbl_ctx* ctx, const void* a, const void* b, double scale, void* out, int accumulate, int nslices);

/* ---- white noise for `simulate` / `randn!` (src/specialops.jl:6,93: sqrt(D) * randn!(rng, similar(diag(D)));
 * src/base_fields.jl:169-170: randn! fills the Map array).  Replaces the reference's host RNG + upload
 * (ext/CMBLensingCUDAExt.jl:72-73) / CURAND stream.  Slot b of `out` (n_per_slot reals of the context's dtype, e.g. one
 * chain's (Ny,Nx,P) map block) is filled with N(0,1) draws of the counter-based generator Philox4x32-10 keyed by
 * seeds_host[b]; `stream` selects an independent sequence of the same key (e.g. a running draw counter).  Element j of a slot
 * depends only on (seed, stream, j): counter (j/4, stream), Box-Muller in fp64 on u = (w + 0.5)/2^32 of word pairs. */
int cmbl_randn(cmbl_ctx* ctx, const uint64_t* seeds_host, int nslots, uint64_t stream, void* out, long n_per_slot);

/* ---- data model, Wiener filter and posterior (src/dataset.jl:37-137, src/maximization.jl:17-42,
 *      src/numerical_algorithms.jl:73-134).  Operators are set as real planes in the reference
 *      layout; *_INV operators are the caller's pinv() of the reference operators. */
enum { CMBL_OP_CF_INV = 0,     /* pinv(Cf)                                   harmonic, npol planes (5 for IQU) */
       CMBL_OP_CN_INV = 1,     /* pinv(Cn)                                                                      */
       CMBL_OP_B = 2,          /* beam / transfer function                                                      */
       CMBL_OP_MF = 3,         /* Fourier-space mask                                                            */
       CMBL_OP_D = 4,          /* mixing matrix D                                                               */
       CMBL_OP_D_INV = 5,      /* operator applied for `D \ f` (pinv(D), or D itself with DIV semantics)        */
       CMBL_OP_PRECOND_INV = 6,/* pinv(Cf^-1 + B'M'Cn^-1 M B)  (src/dataset.jl:129-132)                          */
       CMBL_OP_CPHI_INV = 7,   /* pinv(Cphi), 1 plane                                                           */
       CMBL_OP_G_INV = 8,      /* pinv(G), 1 plane                                                              */
       CMBL_OP_MPIX = 9,       /* pixel mask, real (Ny,Nx) map; optional                                        */
       CMBL_OP_COUNT = 10 };
int cmbl_dataset_create(cmbl_ctx* ctx, int npol, cmbl_dataset** out);
int cmbl_dataset_destroy(cmbl_dataset* ds);
int cmbl_dataset_set_op(cmbl_dataset* ds, int which, const void* planes, int nplanes);
int cmbl_dataset_set_data(cmbl_dataset* ds, const void* d_harmonic, int nbatch);
/* sum of the three logdet terms of logpdf (logdet Cf + logdet Cphi + logdet Cn), per batch identical */
int cmbl_dataset_set_logdet(cmbl_dataset* ds, double logdet_sum);

/* gradientf_logpdf (src/dataset.jl:76-80): f, out HARMONIC; d = NULL uses ds.d; use_zero_d != 0 uses d = 0 */
int cmbl_gradientf_logpdf(cmbl_dataset* ds, cmbl_flow* L, const void* f, const void* d, int use_zero_d,
                          void* out, int nbatch);
/* argmaxf_logpdf (src/maximization.jl:17-42) by conjugate_gradient (src/numerical_algorithms.jl:73-134):
 * fstart may be NULL; res_hist_host has room for maxit*nbatch doubles; *nit_host = history length. */
int cmbl_wiener_cg(cmbl_dataset* ds, cmbl_flow* L, const void* d, const void* fstart, double tol, int maxit,
                   void* f_out, double* res_hist_host, int* nit_host, int nbatch);
/* logpdf(Mixed(ds); f°, phi°) and its gradient (src/dataset.jl:84-117, src/maximization.jl:178):
 * fo MAP, phio FOURIER; gfo MAP, gphio FOURIER.  L's phi is overwritten with G \ phi°. */
int cmbl_logpdf_mixed(cmbl_dataset* ds, cmbl_flow* L, const void* fo, const void* phio, double* lp_host, int nbatch);
int cmbl_grad_logpdf_mixed(cmbl_dataset* ds, cmbl_flow* L, const void* fo, const void* phio, double* lp_host,
                           void* gfo, void* gphio, int nbatch, int alias_quirk);

/* ---- loop bodies of the reference's drivers, for hosts that are neither Julia (which keeps src/maximization.jl:116-233 and
 *      src/sampling.jl:388-464 itself on top of the entry points above) nor Python (cmblensing.jl_amd/drivers.py).  Control flow on the
 *      host inside the library, every field operation one of the launches above; both return when their host outputs are final.
 *
 * hmc_step (src/sampling.jl:405-418; leapfrog = symplectic_integrate, :14-46): one HMC update of phi° at fixed f° with
 *   U = logpdf(Mixed(ds)).  mass = Lambda, the real (Ny/2+1, Nx) plane of mass_matrix_phi (:422-425).  The momentum is
 *   p0 = sqrt(Lambda) .* rfft(white_p), white_p a unit white-noise MAP (Ny, Nx, 1, nbatch); white_p == NULL draws it on the device and
 *   log_u_host == NULL draws log(rand()) from the engine's counter-based generator with the stream convention of the Python drivers:
 *   batch slot b uses key seeds_host[b], momentum stream 2 + 16 step, uniform stream 3 + 16 step (a chain is reproducible whatever
 *   GPU or launch geometry runs it).  accept = always_accept || log_u < dH; a NaN dH (diverged trajectory) rejects.
 *   phio_out (FOURIER, may alias phio) = accepted ? proposal : phio, per batch slot. */
int cmbl_hmc_step(cmbl_dataset* ds, cmbl_flow* L, const void* fo, const void* phio, const void* mass, const void* white_p,
                  const double* log_u_host, const uint64_t* seeds_host, uint64_t step, int nleap, double eps, int always_accept,
                  int alias_quirk, int nbatch, void* phio_out, double* dH_host, int* accept_host);
/* MAP_joint loop body (src/maximization.jl:160-206) with G = I as the reference sets it (:146; the dataset's own G is put back on
 *   return): f = argmaxf_logpdf(phi; fstart, cg_tol, cg_maxit) [HARMONIC]; (f°, phi°) = mix(f, phi); g = d logpdf(Mixed) / d phi°;
 *   step direction hinv .* g, hinv the real plane pinv(Cphi^-1 + Nphi^-1) (src/dataset.jl:134-137); alpha = argmin over [0, alpha_max]
 *   of -sum_b logpdf(Mixed; f°, phi° + alpha * step) by Brent's method (abs_tol = alpha_tol, rel_tol = sqrt(eps(T)), a NaN logpdf is
 *   penalised as (alpha / alpha_max) * floatmax(T), :194-199); phi_out = unmix(phi° + alpha * step).  One alpha for all batch slots.
 *   Outputs: f_out HARMONIC (the Wiener-filtered f), phi_out FOURIER, logpdf_host[nbatch] at the new point, *alpha_host,
 *   *ncg_host = CG iterations, *nls_host = logpdf evaluations of the line search. */
int cmbl_map_joint_step(cmbl_dataset* ds, cmbl_flow* L, const void* phi, const void* fstart, const void* hinv, double alpha_max,
                        double alpha_tol, double cg_tol, int cg_maxit, int alias_quirk, int nbatch, void* f_out, void* phi_out,
                        double* logpdf_host, double* alpha_host, int* ncg_host, int* nls_host);
/* quadratic_estimate(ds, which) (src/quadratic_estimate.jl:29-200) on the dataset's data: which = 0 TT, 1 EE, 2 EB (the pairs the
 *   reference implements, :41).  The *_host arguments are real (Ny/2+1, Nx) planes in double precision, one per component the
 *   estimator uses (TT: T; EE: E; EB: E then B): Cf (unlensed), Cftilde (lensed), Cn, and TF = Mf .* B, the Fourier-diagonal
 *   approximations of mask x beam and noise the reference uses (`ds.M̂`, `ds.B̂`, `ds.Cn̂`), plus Cphi.  AL_in_host != NULL skips the
 *   normalisation sums and uses that plane (:38).  Outputs: phiqe_out FOURIER (Ny/2+1, Nx, 1, nbatch) = (wiener_filtered ?
 *   Cphi/(Cphi+AL) : 1) .* AL .* unnormalised estimate; AL_out_host (may be NULL) the normalisation = N0 bias plane.
 *   Every plane argument (inputs and AL_out_host) may be a HOST or a DEVICE pointer (detected with hipPointerGetAttributes): a caller that
 *   keeps the planes of a dataset on the device -- they change only with theta -- pays no transfer per call.  All plane algebra runs on
 *   the device in double precision, rounded once to the working precision, exactly as the host algebra of the reference does. */
int cmbl_quadratic_estimate(cmbl_dataset* ds, int which, const double* Cf_host, const double* Cftilde_host, const double* Cn_host,
                            const double* TF_host, const double* Cphi_host, int wiener_filtered, const double* AL_in_host,
                            void* phiqe_out, double* AL_out_host, int nbatch);

#ifdef __cplusplus
}
#endif
#endif
