// Shared host/device helpers for libcmblens_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <stdexcept>

namespace cmbl {

constexpr int NTP = 256;           // threads per workgroup of the pointwise / reduction / layout kernels
// FFT-carrying kernels take their workgroup size as a template parameter NT (256, 512 or 1024)

template <typename T> struct alignas(2 * sizeof(T)) cx { T x, y; };   // 8/16-byte aligned: one ds_*_b64/b128, global_*_dwordx2/x4 per element

template <typename T> __host__ __device__ __forceinline__ cx<T> mk(T a, T b) { cx<T> r; r.x = a; r.y = b; return r; }
template <typename T> __host__ __device__ __forceinline__ cx<T> operator+(cx<T> a, cx<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T> __host__ __device__ __forceinline__ cx<T> operator-(cx<T> a, cx<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <typename T> __host__ __device__ __forceinline__ cx<T> operator*(cx<T> a, cx<T> b) { return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
template <typename T> __host__ __device__ __forceinline__ cx<T> operator*(T s, cx<T> a) { return mk<T>(s * a.x, s * a.y); }
template <typename T> __host__ __device__ __forceinline__ cx<T> conj(cx<T> a) { return mk<T>(a.x, -a.y); }
template <typename T> __host__ __device__ __forceinline__ cx<T> mul_i(cx<T> a) { return mk<T>(-a.y, a.x); }      // i*a
template <typename T> __host__ __device__ __forceinline__ cx<T> mul_mi(cx<T> a) { return mk<T>(a.y, -a.x); }     // -i*a
template <typename T> __host__ __device__ __forceinline__ cx<T> cmulconj(cx<T> a, cx<T> w) {                     // a*conj(w)
  return mk<T>(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y);
}

__device__ __forceinline__ int brev(int i, int bits) { return bits == 0 ? 0 : (int)(__brev((unsigned)i) >> (32 - bits)); }

// ---- error plumbing: never throw across the C ABI -------------------------------------------
extern thread_local std::string g_last_error;
enum Status { OK = 0, ERR_ARG = 1, ERR_SHAPE = 2, ERR_HIP = 3, ERR_NAN = 4, ERR_STATE = 5, ERR_ALLOC = 6 };

struct Error { int code; std::string msg; };
[[noreturn]] inline void fail(int code, const std::string& m) { throw Error{code, m}; }

#define CMBL_HIP(expr)                                                                         \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      ::cmbl::fail(::cmbl::ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));         \
  } while (0)

#define CMBL_REQUIRE(cond, code, msg)                                                          \
  do { if (!(cond)) ::cmbl::fail(code, std::string(msg) + " [" #cond "]"); } while (0)

inline int ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }
inline bool ispow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

// RAII device buffer
struct DevBuf {
  void* p = nullptr; size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
  void ensure(size_t n) {
    if (n <= bytes) return;
    release();
    hipError_t e = hipMalloc(&p, n);
    // (a failed hipMalloc also leaves its code in the runtime's last-error slot: cleared here, or the hipGetLastError behind the NEXT
    //  kernel launch would report an out-of-memory error for a launch that succeeded)
    if (e != hipSuccess) { p = nullptr; bytes = 0; (void)hipGetLastError(); fail(ERR_ALLOC, std::string("hipMalloc(") + std::to_string(n) + "): " + hipGetErrorString(e)); }
    bytes = n;
  }
  template <typename U> U* as() const { return reinterpret_cast<U*>(p); }
};

}  // namespace cmbl
