// Micro-benchmark with in-kernel phase timestamps (s_memtime) for the row kernel k_x_fft<float,2,256,10>.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/xbench.hip -o build_variants/xbench && ./build_variants/xbench
#include "../../cmblensing.jl_amd/csrc/kernels_fft.hpp"
#include <algorithm>
namespace cmbl { thread_local std::string g_last_error; }
using namespace cmbl;

template <typename T, int NT, int LGNX, int MAXLG>
__global__ __launch_bounds__(NT) void k_probe(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, const cx<T>* __restrict__ twX,
                                              const T* __restrict__ lx_r, long rows, int RX, long long* __restrict__ ts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int Nx = 1 << LGNX, LD = tile_ld(Nx);
  cx<T>* tw = reinterpret_cast<cx<T>*>(smem);
  cx<T>* s = tw + (Nx >> 1);
  long long t[7];
  t[0] = __builtin_readcyclecounter();
  const long r0 = (long)blockIdx.x * RX;
  const int nr = (int)min((long)RX, rows - r0);
  load_twiddles<T, NT>(tw, twX, Nx >> 1);
  const cx<T>* src = in + r0 * Nx;
  for (int e = threadIdx.x; e < nr * Nx; e += NT) s[(e >> LGNX) * LD + pad(e & (Nx - 1))] = src[e];
  __syncthreads();
  t[1] = __builtin_readcyclecounter();
  const T inv = T(1) / T(Nx);
  fft_dif<T, NT, LD, LGNX, LGNX, MAXLG>(s, nr, tw);
  t[2] = __builtin_readcyclecounter();
  for (int e = threadIdx.x; e < nr * Nx; e += NT) {
    const int i = e & (Nx - 1), si = (e >> LGNX) * LD + pad(i);
    const T l = lx_r[i] * inv;
    cx<T> v = s[si];
    s[si] = mk<T>(-l * v.y, l * v.x);
  }
  __syncthreads();
  t[3] = __builtin_readcyclecounter();
  fft_dit<T, NT, LD, LGNX, LGNX, MAXLG>(s, nr, tw);
  t[4] = __builtin_readcyclecounter();
  cx<T>* dst = out + r0 * Nx;
  for (int e = threadIdx.x; e < nr * Nx; e += NT) dst[e] = s[(e >> LGNX) * LD + pad(e & (Nx - 1))];
  t[5] = __builtin_readcyclecounter();
  if (threadIdx.x == 0) for (int i = 0; i < 6; ++i) ts[blockIdx.x * 6 + i] = t[i];
}

template <int NT, int MAXLG> void run(int RX, const char* label) {
  constexpr int LGNX = 10, Nx = 1024;
  const long rows = 1026;
  cx<float>*in, *out, *tw; float* lx; long long* ts;
  hipMalloc(&in, rows * Nx * 8); hipMalloc(&out, rows * Nx * 8); hipMalloc(&tw, Nx * 4); hipMalloc(&lx, Nx * 4);
  const int nblk = (int)((rows + RX - 1) / RX);
  hipMalloc(&ts, nblk * 6 * 8);
  hipMemset(in, 0, rows * Nx * 8); hipMemset(tw, 0, Nx * 4); hipMemset(lx, 0, Nx * 4);
  const size_t lds = ((size_t)Nx / 2 + (size_t)RX * tile_ld(Nx)) * 8;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_probe<float, NT, LGNX, MAXLG>), dim3(nblk), dim3(NT), lds, 0, in, out, tw, lx, rows, RX, ts);
  hipEventRecord(a);
  const int reps = 20;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_probe<float, NT, LGNX, MAXLG>), dim3(nblk), dim3(NT), lds, 0, in, out, tw, lx, rows, RX, ts);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<long long> h(nblk * 6); hipMemcpy(h.data(), ts, nblk * 48, hipMemcpyDeviceToHost);
  double ph[5] = {0, 0, 0, 0, 0}; long long tmin = h[0], tmax = 0; std::vector<long long> life;
  for (int i = 0; i < nblk; ++i) { for (int p = 0; p < 5; ++p) ph[p] += h[i * 6 + p + 1] - h[i * 6 + p]; tmin = std::min(tmin, h[i * 6]); tmax = std::max(tmax, h[i * 6 + 5]); life.push_back(h[i*6+5]-h[i*6]); }
  std::sort(life.begin(), life.end());
  // start-time distribution: how many blocks start late (second generation)
  int late = 0; for (int i = 0; i < nblk; ++i) if (h[i * 6] - tmin > life[nblk / 2] / 2) ++late;
  printf("%-28s %7.2f us/launch | blocks %4d late-start %4d | span %6lld ticks | per-block ticks: load %6.0f dif %6.0f mul %5.0f dit %6.0f store %5.0f | life med %lld max %lld\n", label, ms / reps * 1e3, nblk, late,
         tmax - tmin, ph[0] / nblk, ph[1] / nblk, ph[2] / nblk, ph[3] / nblk, ph[4] / nblk, life[nblk/2], life.back());
  hipFree(in); hipFree(out); hipFree(tw); hipFree(lx); hipFree(ts);
}

int main() {
  run<256, 4>(1, "NT256 radix16 RX1"); run<256, 4>(2, "NT256 radix16 RX2"); run<256, 4>(4, "NT256 radix16 RX4");
  run<256, 3>(1, "NT256 radix8  RX1"); run<256, 2>(1, "NT256 radix4  RX1"); run<256, 2>(2, "NT256 radix4  RX2");
  run<128, 3>(1, "NT128 radix8  RX1"); run<64, 4>(1, "NT64  radix16 RX1"); run<512, 2>(2, "NT512 radix4  RX2");
  run<1024, 2>(4, "NT1024 radix4 RX4");
  return 0;
}
