// Counter calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section: "calibrate on a known
// byte count in your own access pattern before trusting an absolute").  Streaming copies of known size in the access widths the
// library's kernels use: 4, 8 and 16 bytes per lane, plain and non-temporal stores, contiguous per wavefront; read-only (sum) and
// write-only (fill) variants separate the two counters.  Sizes: 64 MiB (fits the 256 MiB Infinity Cache: re-reads of a previous
// kernel's output are cache hits, which the fabric-side counters still count) and 1 GiB (cannot fit).
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/calib_copy.hip -o /tmp/calib_copy
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d out/fetch -o c -- /tmp/calib_copy
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d out/write -o c -- /tmp/calib_copy
//   python tools/make_calib_json.py out/fetch/..counter_collection.csv out/write/..counter_collection.csv profiles/r03_counter_calibration.json
// Kernel names carry the width and the variant; the byte count of every launch is n * sizeof(V) (printed as a table on stdout).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef float f1;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

// W = bytes per lane, BIG = 0: 64 MiB arrays, 1: 1 GiB arrays (both only tag the kernel name for the post-processing)
template <int W> struct VecOf;
template <> struct VecOf<4> { using type = f1; };
template <> struct VecOf<8> { using type = f2; };
template <> struct VecOf<16> { using type = f4; };
#define TV template <int W, int BIG, typename V = typename VecOf<W>::type>
TV __global__ __launch_bounds__(256) void copy_plain(const V* __restrict__ in, V* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
TV __global__ __launch_bounds__(256) void copy_ntstore(const V* __restrict__ in, V* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(in[i], out + i);
}
TV __global__ __launch_bounds__(256) void copy_ntload(const V* __restrict__ in, V* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = __builtin_nontemporal_load(in + i);
}
__device__ inline float lane_sum(f1 v) { return v; }
__device__ inline float lane_sum(f2 v) { return v.x + v.y; }
__device__ inline float lane_sum(f4 v) { return v.x + v.y + v.z + v.w; }
TV __global__ __launch_bounds__(256) void read_only(const V* __restrict__ in, float* __restrict__ out, size_t n) {
  float s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += lane_sum(in[i]);
  if (s == 12345.678f) out[0] = s;                    // never true for the fill pattern; keeps the loads alive
}
TV __global__ __launch_bounds__(256) void write_only(V* __restrict__ out, size_t n) {
  V v; for (unsigned k = 0; k < sizeof(V) / 4; ++k) reinterpret_cast<float*>(&v)[k] = 1.0f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}

template <int W, int BIG> void run(const char* name, void* a, void* b, size_t bytes) {
  using V = typename VecOf<W>::type;
  const size_t n = bytes / sizeof(V);
  const unsigned grid = (unsigned)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((copy_plain<W, BIG>), dim3(grid), dim3(256), 0, 0, (const V*)a, (V*)b, n);
    hipLaunchKernelGGL((copy_ntstore<W, BIG>), dim3(grid), dim3(256), 0, 0, (const V*)a, (V*)b, n);
    hipLaunchKernelGGL((copy_ntload<W, BIG>), dim3(grid), dim3(256), 0, 0, (const V*)a, (V*)b, n);
    hipLaunchKernelGGL((read_only<W, BIG>), dim3(grid), dim3(256), 0, 0, (const V*)a, (float*)b, n);
    hipLaunchKernelGGL((write_only<W, BIG>), dim3(grid), dim3(256), 0, 0, (V*)b, n);
  }
  CHECK(hipDeviceSynchronize());
  std::printf("%s bytes_per_array %zu  (copy: read + write that many; read_only: read; write_only: write)\n", name, bytes);
}

int main(int argc, char** argv) {
  const size_t big = (size_t)1 << 30, small = (size_t)64 << 20;
  void *a, *b;
  CHECK(hipMalloc(&a, big)); CHECK(hipMalloc(&b, big));
  CHECK(hipMemset(a, 0, big)); CHECK(hipMemset(b, 0, big));
  run<4, 0>("width4", a, b, small); run<8, 0>("width8", a, b, small); run<16, 0>("width16", a, b, small);
  run<4, 1>("width4", a, b, big); run<8, 1>("width8", a, b, big); run<16, 1>("width16", a, b, big);
  CHECK(hipFree(a)); CHECK(hipFree(b));
  (void)argc; (void)argv;
  return 0;
}
