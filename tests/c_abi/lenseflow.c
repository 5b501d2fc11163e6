/* Plain-C caller of libcmblens_hip.so: what a non-Python host (Julia's ccall, C, Fortran) does at the boundary.
 *   ctx_create -> device buffers -> lenseflow_create -> set_phi -> apply (L*f) -> grad (pullback of L*f) -> compare with the
 *   float64 oracle vectors of tests/golden/cabi_lenseflow.bin (tools/make_cabi_golden.py).
 * Build: gcc -std=c99 -O1 -I include tests/c_abi/lenseflow.c -ldl -lm -o lenseflow_c     (no HIP headers, no HIP link)
 * Run:   ./lenseflow_c cmblensing.jl_amd/libcmblens_hip.so tests/golden/cabi_lenseflow.bin
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
typedef int (*cmbl_lenseflow_set_phi_t)(cmbl_flow*, int, const void*, int);
typedef int (*cmbl_lenseflow_apply_t)(cmbl_flow*, int, int, const void*, int, void*, int, int);
typedef int (*cmbl_lenseflow_grad_t)(cmbl_flow*, int, const void*, int, const void*, void*, int, void*, void*, int, int, int);
typedef int (*cmbl_dot_t)(cmbl_ctx*, int, const void*, const void*, int, int, double*);

static double rel_l2(const double* a, const double* b, size_t n) {
  double num = 0, den = 0;
  for (size_t i = 0; i < n; ++i) { num += (a[i] - b[i]) * (a[i] - b[i]); den += b[i] * b[i]; }
  return sqrt(num / den);
}

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s libcmblens_hip.so cabi_lenseflow.bin\n", argv[0]); return 2; }
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  SYM(cmbl_last_error) SYM(cmbl_ctx_create) SYM(cmbl_ctx_destroy) SYM(cmbl_device_malloc) SYM(cmbl_device_free)
  SYM(cmbl_copy_to_device) SYM(cmbl_copy_to_host) SYM(cmbl_lenseflow_create) SYM(cmbl_lenseflow_destroy)
  SYM(cmbl_lenseflow_set_phi) SYM(cmbl_lenseflow_apply) SYM(cmbl_lenseflow_grad) SYM(cmbl_dot)
SYM(cmbl_abi_version)
  if (p_cmbl_abi_version() != CMBL_ABI_VERSION) { fprintf(stderr, "ABI version %d, header %d\n", p_cmbl_abi_version(), CMBL_ABI_VERSION); return 2; }
#define CHK(call) do { int rc_ = (call); if (rc_ != CMBL_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, p_cmbl_last_error()); return 1; } } while (0)

  FILE* fh = fopen(argv[2], "rb");
  if (!fh) { perror(argv[2]); return 2; }
  int32_t hdr[4]; double theta;
  if (fread(hdr, 4, 4, fh) != 4 || fread(&theta, 8, 1, fh) != 1) return 2;
  const int Ny = hdr[0], Nx = hdr[1], P = hdr[2], nsteps = hdr[3], Nyh = Ny / 2 + 1;
  const size_t nmap = (size_t)Ny * Nx, nfou = (size_t)Nyh * Nx * 2;            /* doubles per plane */
  const size_t sz[6] = {nmap, P * nmap, P * nfou, P * nmap, nfou, P * nfou};   /* phi, f, delta, Lf, dphi, df */
  double* h[6];
  for (int i = 0; i < 6; ++i) {
    h[i] = (double*)malloc(sz[i] * sizeof(double));
    if (fread(h[i], sizeof(double), sz[i], fh) != sz[i]) { fprintf(stderr, "short read\n"); return 2; }
  }
  fclose(fh);

  cmbl_ctx* ctx = NULL; cmbl_flow* L = NULL;
  CHK(p_cmbl_ctx_create(Ny, Nx, theta, CMBL_F64, 0, NULL, &ctx));
  void *d_phi, *d_f, *d_delta, *d_Lf, *d_dphi, *d_df, *d_f0;
  CHK(p_cmbl_device_malloc(ctx, sz[0] * 8, &d_phi)); CHK(p_cmbl_device_malloc(ctx, sz[1] * 8, &d_f)); CHK(p_cmbl_device_malloc(ctx, sz[2] * 8, &d_delta));
  CHK(p_cmbl_device_malloc(ctx, sz[3] * 8, &d_Lf)); CHK(p_cmbl_device_malloc(ctx, sz[4] * 8, &d_dphi)); CHK(p_cmbl_device_malloc(ctx, sz[5] * 8, &d_df));
  CHK(p_cmbl_device_malloc(ctx, sz[1] * 8, &d_f0));
  CHK(p_cmbl_copy_to_device(ctx, d_phi, h[0], sz[0] * 8)); CHK(p_cmbl_copy_to_device(ctx, d_f, h[1], sz[1] * 8));
  CHK(p_cmbl_copy_to_device(ctx, d_delta, h[2], sz[2] * 8));

  CHK(p_cmbl_lenseflow_create(ctx, nsteps, &L));
  /* a flow used before its phi is set is a status code, not a crash */
  if (p_cmbl_lenseflow_apply(L, CMBL_FLOW_FWD, CMBL_MAP, d_f, CMBL_MAP, d_Lf, P, 1) != CMBL_ERR_STATE) { fprintf(stderr, "expected CMBL_ERR_STATE\n"); return 1; }
  CHK(p_cmbl_lenseflow_set_phi(L, CMBL_MAP, d_phi, 1));
  CHK(p_cmbl_lenseflow_apply(L, CMBL_FLOW_FWD, CMBL_MAP, d_f, CMBL_MAP, d_Lf, P, 1));
  CHK(p_cmbl_lenseflow_grad(L, CMBL_FLOW_FWD, d_Lf, CMBL_FOURIER, d_delta, d_dphi, CMBL_FOURIER, d_df, d_f0, P, 1, 0));

  double* got = (double*)malloc(sz[2] * sizeof(double));
  int bad = 0;
  const char* names[3] = {"L*f", "dphi", "df"};
  void* dev[3] = {d_Lf, d_dphi, d_df};
  for (int i = 0; i < 3; ++i) {
    CHK(p_cmbl_copy_to_host(ctx, got, dev[i], sz[3 + i] * 8));
    const double e = rel_l2(got, h[3 + i], sz[3 + i]);
    printf("%-5s rel L2 error vs float64 oracle: %.3e\n", names[i], e);
    if (!(e < 1e-9)) bad = 1;
  }
  /* the delta flow carries f~ back to f */
  CHK(p_cmbl_copy_to_host(ctx, got, d_f0, sz[1] * 8));
  { const double e = rel_l2(got, h[1], sz[1]); printf("f0    rel L2 error vs f: %.3e\n", e); if (!(e < 1e-4)) bad = 1; }
  double dot = 0;
  CHK(p_cmbl_dot(ctx, CMBL_MAP, d_f, d_f, P, 1, &dot));
  { double ref = 0; for (size_t i = 0; i < sz[1]; ++i) ref += h[1][i] * h[1][i]; if (fabs(dot - ref) > 1e-10 * ref) { fprintf(stderr, "dot mismatch\n"); bad = 1; } }

  CHK(p_cmbl_lenseflow_destroy(L));
  p_cmbl_device_free(ctx, d_phi); p_cmbl_device_free(ctx, d_f); p_cmbl_device_free(ctx, d_delta); p_cmbl_device_free(ctx, d_Lf);
  p_cmbl_device_free(ctx, d_dphi); p_cmbl_device_free(ctx, d_df); p_cmbl_device_free(ctx, d_f0);
  CHK(p_cmbl_ctx_destroy(ctx));
  puts(bad ? "C_ABI_FAIL" : "C_ABI_PASS");
  return bad;
}
