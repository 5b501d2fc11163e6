/* Plain-C caller of the posterior entry points of libcmblens_hip.so -- the call bench.py times, from a host with no HIP:
 *   ctx_create -> dataset_create -> dataset_set_op x10 / set_data / set_logdet -> lenseflow_create ->
 *   cmbl_logpdf_mixed, cmbl_grad_logpdf_mixed (src/dataset.jl:84-87 and its Zygote gradient, src/maximization.jl:178) ->
 *   compare with the float64 oracle vectors of tests/golden/cabi_posterior.bin (tools/make_cabi_posterior_golden.py).
 * This is what julia/CMBLensingHIPExt.jl's `logpdf(::Mixed{<:HIPDataSet})` and its adjoint do through `ccall`.
 * Build: gcc -std=c99 -O1 -I include tests/c_abi/posterior.c -ldl -lm -o posterior_c     (no HIP headers, no HIP link)
 * Run:   ./posterior_c cmblensing.jl_amd/libcmblens_hip.so tests/golden/cabi_posterior.bin
 */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "cmblens.h"

#define SYM(name) name##_t p_##name = (name##_t)dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }
typedef const char* (*cmbl_last_error_t)(void);
typedef int (*cmbl_abi_version_t)(void);
typedef int (*cmbl_ctx_create_t)(int, int, double, int, int, void*, cmbl_ctx**);
typedef int (*cmbl_ctx_destroy_t)(cmbl_ctx*);
typedef int (*cmbl_device_malloc_t)(cmbl_ctx*, size_t, void**);
typedef int (*cmbl_device_free_t)(cmbl_ctx*, void*);
typedef int (*cmbl_copy_to_device_t)(cmbl_ctx*, void*, const void*, size_t);
typedef int (*cmbl_copy_to_host_t)(cmbl_ctx*, void*, const void*, size_t);
typedef int (*cmbl_lenseflow_create_t)(cmbl_ctx*, int, cmbl_flow**);
typedef int (*cmbl_lenseflow_destroy_t)(cmbl_flow*);
typedef int (*cmbl_dataset_create_t)(cmbl_ctx*, int, cmbl_dataset**);
typedef int (*cmbl_dataset_destroy_t)(cmbl_dataset*);
typedef int (*cmbl_dataset_set_op_t)(cmbl_dataset*, int, const void*, int);
typedef int (*cmbl_dataset_set_data_t)(cmbl_dataset*, const void*, int);
typedef int (*cmbl_dataset_set_logdet_t)(cmbl_dataset*, double);
typedef int (*cmbl_logpdf_mixed_t)(cmbl_dataset*, cmbl_flow*, const void*, const void*, double*, int);
typedef int (*cmbl_grad_logpdf_mixed_t)(cmbl_dataset*, cmbl_flow*, const void*, const void*, double*, void*, void*, int, int);

static double rel_l2(const double* a, const double* b, size_t n) {
  double num = 0, den = 0;
  for (size_t i = 0; i < n; ++i) { num += (a[i] - b[i]) * (a[i] - b[i]); den += b[i] * b[i]; }
  return sqrt(num / den);
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s libcmblens_hip.so cabi_posterior.bin\n", argv[0]); return 2; }
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  SYM(cmbl_last_error) SYM(cmbl_ctx_create) SYM(cmbl_ctx_destroy) SYM(cmbl_device_malloc) SYM(cmbl_device_free)
  SYM(cmbl_copy_to_device) SYM(cmbl_copy_to_host) SYM(cmbl_lenseflow_create) SYM(cmbl_lenseflow_destroy)
  SYM(cmbl_dataset_create) SYM(cmbl_dataset_destroy) SYM(cmbl_dataset_set_op) SYM(cmbl_dataset_set_data) SYM(cmbl_dataset_set_logdet)
  SYM(cmbl_logpdf_mixed) SYM(cmbl_grad_logpdf_mixed)
SYM(cmbl_abi_version)
  if (p_cmbl_abi_version() != CMBL_ABI_VERSION) { fprintf(stderr, "ABI version %d, header %d\n", p_cmbl_abi_version(), CMBL_ABI_VERSION); return 2; }
#define CHK(call) do { int rc_ = (call); if (rc_ != CMBL_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, p_cmbl_last_error()); return 1; } } while (0)

  FILE* fh = fopen(argv[2], "rb");
  if (!fh) { perror(argv[2]); return 2; }
  int32_t hdr[4]; double theta, logdet_sum;
  if (fread(hdr, 4, 4, fh) != 4 || fread(&theta, 8, 1, fh) != 1 || fread(&logdet_sum, 8, 1, fh) != 1) return 2;
  const int Ny = hdr[0], Nx = hdr[1], P = hdr[2], nsteps = hdr[3], Nyh = Ny / 2 + 1;
  const size_t nmap = (size_t)Ny * Nx, npl = (size_t)Nyh * Nx, nfou = npl * 2;       /* doubles per map / real plane / complex plane */
  /* 10 operators in CMBL_OP_* order, then d, fo, phio, expected lp, gfo, gphio */
  enum { NARR = 16 };
  const int op_planes[10] = {P, P, P, P, P, P, P, 1, 1, 1};
  size_t sz[NARR];
  for (int i = 0; i < 9; ++i) sz[i] = (size_t)op_planes[i] * npl;
  sz[9] = nmap; sz[10] = P * nfou; sz[11] = P * nmap; sz[12] = nfou; sz[13] = 1; sz[14] = P * nmap; sz[15] = nfou;
  double* h[NARR];
  for (int i = 0; i < NARR; ++i) {
    h[i] = (double*)malloc(sz[i] * sizeof(double));
    if (fread(h[i], sizeof(double), sz[i], fh) != sz[i]) { fprintf(stderr, "short read (array %d)\n", i); return 2; }
  }
  fclose(fh);

  cmbl_ctx* ctx = NULL; cmbl_flow* L = NULL; cmbl_dataset* ds = NULL;
  CHK(p_cmbl_ctx_create(Ny, Nx, theta, CMBL_F64, 0, NULL, &ctx));
  CHK(p_cmbl_dataset_create(ctx, P, &ds));
  CHK(p_cmbl_lenseflow_create(ctx, nsteps, &L));
  void* dev[NARR];
  for (int i = 0; i < NARR; ++i) CHK(p_cmbl_device_malloc(ctx, sz[i] * 8, &dev[i]));
  for (int i = 0; i < 13; ++i) CHK(p_cmbl_copy_to_device(ctx, dev[i], h[i], sz[i] * 8));
  for (int i = 0; i < 10; ++i) CHK(p_cmbl_dataset_set_op(ds, i, dev[i], op_planes[i]));
  CHK(p_cmbl_dataset_set_data(ds, dev[10], 1));
  CHK(p_cmbl_dataset_set_logdet(ds, logdet_sum));

  int bad = 0;
  double lp = 0, lp2 = 0;
  CHK(p_cmbl_logpdf_mixed(ds, L, dev[11], dev[12], &lp, 1));
  CHK(p_cmbl_grad_logpdf_mixed(ds, L, dev[11], dev[12], &lp2, dev[14], dev[15], 1, 0));
  printf("logpdf(Mixed) %.10f  (gradient call: %.10f)  oracle %.10f\n", lp, lp2, h[13][0]);
  if (!(fabs(lp - h[13][0]) < 1e-9 * fabs(h[13][0])) || !(fabs(lp2 - h[13][0]) < 1e-9 * fabs(h[13][0]))) bad = 1;
  double* got = (double*)malloc(sz[14] * sizeof(double));
  CHK(p_cmbl_copy_to_host(ctx, got, dev[14], sz[14] * 8));
  { const double e = rel_l2(got, h[14], sz[14]); printf("grad f°   rel L2 error vs float64 oracle: %.3e\n", e); if (!(e < 1e-8)) bad = 1; }
  CHK(p_cmbl_copy_to_host(ctx, got, dev[15], sz[15] * 8));
  { const double e = rel_l2(got, h[15], sz[15]); printf("grad phi° rel L2 error vs float64 oracle: %.3e\n", e); if (!(e < 1e-8)) bad = 1; }

  CHK(p_cmbl_lenseflow_destroy(L));
  CHK(p_cmbl_dataset_destroy(ds));
  for (int i = 0; i < NARR; ++i) p_cmbl_device_free(ctx, dev[i]);
  CHK(p_cmbl_ctx_destroy(ctx));
  puts(bad ? "C_ABI_FAIL" : "C_ABI_PASS");
  return bad;
}
