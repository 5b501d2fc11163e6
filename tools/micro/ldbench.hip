// Micro-benchmark behind the tiled "mixed" layout: how fast does a launch of column-tile workgroups pull its tile from
// memory when the tile is (A) C = 4 complex values wide out of rows of Nx (32-byte segments, 8 KB apart -- the [ky][x] layout) or
// (B) one contiguous block per tile (the [x/4][ky][4] layout)?  Same bytes, same grid, same instruction count.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/ldbench.hip -o gpurun_out/ldbench && gpurun_out/ldbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct alignas(8) c2 { float x, y; };
template <int MODE>
__global__ __launch_bounds__(512) void k(const c2* __restrict__ a, const c2* __restrict__ b, float* __restrict__ out, int Nx, int Nyh) {
  const int x0 = blockIdx.x * 4;
  const size_t sl = blockIdx.y, moff = sl * (size_t)Nyh * Nx;
  c2 X[5], Y[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int e = threadIdx.x + i * 512;
    if (e < 4 * Nyh) {
      const size_t gi = MODE == 0 ? (size_t)(e >> 2) * Nx + x0 + (e & 3) : (size_t)blockIdx.x * Nyh * 4 + e;
      X[i] = a[moff + gi]; Y[i] = b[moff + gi];
    } else { X[i] = c2{0, 0}; Y[i] = c2{0, 0}; }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) s += X[i].x * Y[i].y + X[i].y * Y[i].x;
  if (s == 12345.678f) out[blockIdx.x] = s;        // never true: keeps the loads alive
}
// row-kernel side: a workgroup of 256 threads takes 4 adjacent ky rows (all x).  MODE 0: rows contiguous ([ky][x]);
// MODE 1: tiled layout, 128-byte chunks (4 rows x 4 values) 16 KB apart; 16-byte loads, 8 per thread
struct alignas(16) c4 { float x, y, z, w; };
template <int MODE>
__global__ __launch_bounds__(256) void krow(const c2* __restrict__ a, float* __restrict__ out, int Nx, int Nyh) {
  const int G = (Nyh + 3) / 4, g = blockIdx.x % G, sl = blockIdx.x / G, ky0 = 4 * g;
  const c2* base = a + (size_t)sl * Nyh * Nx;
  float s = 0;
  c4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int e = threadIdx.x + i * 256;                  // 16-byte unit index: 2048 units = 4 rows x 1024 / 2
    size_t gi;
    if (MODE == 0) { const int r = e >> 9, xh = e & 511; gi = (size_t)(ky0 + r) * Nx + 2 * xh; }
    else { const int xt = e >> 3, r = (e >> 1) & 3, ch = e & 1; gi = ((size_t)xt * Nyh + ky0 + r) * 4 + 2 * ch; }
    const bool ok = ky0 + ((MODE == 0) ? (e >> 9) : ((e >> 1) & 3)) < Nyh;
    v[i] = ok ? *reinterpret_cast<const c4*>(base + gi) : c4{0, 0, 0, 0};
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i].x * v[i].w + v[i].y * v[i].z;
  if (s == 12345.678f) out[blockIdx.x] = s;
}
int main() {
  const int Nx = 1024, Nyh = 513, S = 2, NBUF = 24;            // 24 x 2 x 8.4 MB = 403 MB > the 256 MB Infinity Cache
  const size_t n = (size_t)S * Nyh * Nx;
  std::vector<c2*> A(NBUF), B(NBUF);
  for (int i = 0; i < NBUF; ++i) { hipMalloc(&A[i], n * sizeof(c2)); hipMalloc(&B[i], n * sizeof(c2)); hipMemset(A[i], 0, n * sizeof(c2)); hipMemset(B[i], 0, n * sizeof(c2)); }
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rot = 0; rot < 2; ++rot)                           // rot = 0: the same two arrays every launch (cache-resident); 1: rotating (HBM)
    for (int mode = 0; mode < 2; ++mode) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<0>, dim3(Nx / 4, S), dim3(512), 0, 0, A[0], B[0], out, Nx, Nyh);
      hipDeviceSynchronize();
      const int reps = 200;
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) {
        const int i = rot ? r % NBUF : 0;
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(Nx / 4, S), dim3(512), 0, 0, A[i], B[i], out, Nx, Nyh);
        else hipLaunchKernelGGL(k<1>, dim3(Nx / 4, S), dim3(512), 0, 0, A[i], B[i], out, Nx, Nyh);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps, gb = 2.0 * n * sizeof(c2) / 1e9;
      printf("%s  %s : %.2f us per launch, %.0f GB/s\n", rot ? "rotating buffers (HBM)   " : "same buffers (cache)     ",
             mode == 0 ? "strided 32-B segments [ky][x]" : "contiguous tile [x/4][ky][4]  ", us, gb / (us * 1e-6));
    }
  for (int rot = 0; rot < 2; ++rot)
    for (int mode = 0; mode < 2; ++mode) {
      const int G = (Nyh + 3) / 4;
      hipDeviceSynchronize();
      const int reps = 200;
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) {
        const int i = rot ? r % NBUF : 0;
        if (mode == 0) hipLaunchKernelGGL(krow<0>, dim3(G * S), dim3(256), 0, 0, A[i], out, Nx, Nyh);
        else hipLaunchKernelGGL(krow<1>, dim3(G * S), dim3(256), 0, 0, A[i], out, Nx, Nyh);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps, gb = 1.0 * n * sizeof(c2) / 1e9;
      printf("rows: %s  %s : %.2f us per launch, %.0f GB/s\n", rot ? "rotating buffers (HBM)" : "same buffers (cache)  ",
             mode == 0 ? "contiguous rows [ky][x]            " : "tiled, 128-B chunks [x/4][ky][4]   ", us, gb / (us * 1e-6));
    }
  return 0;
}
