// Kernel boundary vs in-kernel grid barrier on MI355X: what a dependent launch costs and what a persistent kernel would pay
// instead.  The flow kernels are chains of ~250 dependent launches per step, each ONE residency wave, so the
// fixed cost per link of the chain matters as much as bandwidth at B = 1.
//
// Workload per link ("transpose-like", so that every workgroup consumes what ALL other workgroups produced in the previous
// link -- the dependency pattern of the column <-> row kernels): G workgroups of 512 threads, array of G x G x CH floats;
// workgroup g reads chunk [h][g][:] for every h and writes chunk [g][h][:] (+1).  After K links the array holds a known value.
//   variant L: K launches on one stream (ping-pong buffers)
//   variant P: ONE persistent launch of G workgroups (all resident: G <= 2 per CU), K links separated by a grid barrier
//              (agent-scope release / acquire around a global counter)
//   variant 0: the same two with CH = 0 (no work): the pure cost of a link
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/gridbar.hip -o /tmp/gridbar && /tmp/gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
constexpr int NT = 512;

__device__ __forceinline__ void link_body(const float* __restrict__ in, float* __restrict__ out, int G, int CH, int g) {
  // chunk = CH floats (multiple of 4); thread t handles float4 t, t + NT, ... of every chunk
  const int nv = CH / 4;
  for (int h = 0; h < G; ++h) {
    const float4* src = reinterpret_cast<const float4*>(in + ((size_t)h * G + g) * CH);
    float4* dst = reinterpret_cast<float4*>(out + ((size_t)g * G + h) * CH);
    for (int i = threadIdx.x; i < nv; i += NT) { float4 v = src[i]; v.x += 1; v.y += 1; v.z += 1; v.w += 1; dst[i] = v; }
  }
}

__global__ __launch_bounds__(NT) void k_link(const float* __restrict__ in, float* __restrict__ out, int G, int CH) {
  extern __shared__ float lds[];
  link_body(in, out, G, CH, blockIdx.x);
}

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();                                             // every wave's stores are issued and (workgroup-scope release) complete
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

template <int FENCE_ALL>
__global__ __launch_bounds__(NT) void k_persistent(float* a, float* b, int G, int CH, int K, unsigned* ctr) {
  extern __shared__ float lds[];
  float* in = a; float* out = b;
  for (int k = 0; k < K; ++k) {
    link_body(in, out, G, CH, blockIdx.x);
    if (FENCE_ALL) __threadfence();                            // agent-scope fence by every wave (belt and braces variant)
    grid_barrier(ctr, (unsigned)(k + 1) * gridDim.x);
    if (FENCE_ALL) __threadfence();
    float* t = in; in = out; out = t;
  }
}

static double check(const std::vector<float>& h, float want) {
  double bad = 0;
  for (float v : h) if (v != want) bad += 1;
  return bad;
}

int main(int argc, char** argv) {
  int dev = 0; CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  const int K = argc > 1 ? std::atoi(argv[1]) : 200;
  hipStream_t st; CHECK(hipStreamCreate(&st));
  unsigned* ctr; CHECK(hipMalloc(&ctr, 4));
  std::printf("%s: %d CUs, K = %d links\n", prop.name, cus, K);
  for (int wpc = 1; wpc <= 2; ++wpc) {
    const int G = cus * wpc;
    const size_t lds = wpc == 1 ? 96 * 1024 : 64 * 1024;       // pins the residency: at most `wpc` workgroups per CU
    for (int CH0 : {0, 16, 64}) {                              // floats per chunk: total bytes per link = G*G*CH*4 read + the same written
      const int CH = CH0 * (wpc == 1 ? 4 : 1);                 // the same 17 / 67 MB per link for both grid sizes
      const size_t n = (size_t)G * G * (CH ? CH : 4);
      float *a, *b; CHECK(hipMalloc(&a, n * 4)); CHECK(hipMalloc(&b, n * 4));
      std::vector<float> h(n);
      const double mb = (double)G * G * CH * 4 / 1e6;
      // variant L
      CHECK(hipMemsetAsync(a, 0, n * 4, st)); CHECK(hipMemsetAsync(b, 0, n * 4, st));
      CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_link), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipMemsetAsync(a, 0, n * 4, st));
        CHECK(hipEventRecord(e0, st));
        float *in = a, *out = b;
        for (int k = 0; k < K; ++k) { hipLaunchKernelGGL(k_link, dim3(G), dim3(NT), lds, st, in, out, G, CH); float* t = in; in = out; out = t; }
        CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 1) {
          CHECK(hipMemcpy(h.data(), in, n * 4, hipMemcpyDeviceToHost));
          std::printf("wpc %d  chunk %4d floats (%7.2f MB r + w per link)  launches : %7.3f us/link  wrong %g\n", wpc, CH, mb, 1e3 * ms / K, CH ? check(h, (float)K) : 0.0);
        }
      }
      // variant P (thread 0 fences) and P' (every wave fences)
      for (int fa = 0; fa < 2; ++fa) {
        auto kern = fa ? k_persistent<1> : k_persistent<0>;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int occ = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, NT, lds));
        if (occ * cus < G) { std::printf("persistent variant skipped: occupancy %d x %d CUs < %d workgroups\n", occ, cus, G); continue; }
        for (int rep = 0; rep < 2; ++rep) {
          hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
          CHECK(hipMemsetAsync(a, 0, n * 4, st)); CHECK(hipMemsetAsync(ctr, 0, 4, st));
          CHECK(hipEventRecord(e0, st));
          hipLaunchKernelGGL(kern, dim3(G), dim3(NT), lds, st, a, b, G, CH, K, ctr);
          CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
          float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
          if (rep == 1) {
            CHECK(hipMemcpy(h.data(), (K & 1) ? b : a, n * 4, hipMemcpyDeviceToHost));
            std::printf("wpc %d  chunk %4d floats (%7.2f MB r + w per link)  %s : %7.3f us/link  wrong %g\n", wpc, CH, mb,
                        fa ? "barrier+f" : "barrier  ", 1e3 * ms / K, CH ? check(h, (float)K) : 0.0);
          }
        }
      }
      CHECK(hipFree(a)); CHECK(hipFree(b));
    }
  }
  return 0;
}
