// Common definitions for the gfx950 LineTR library (host + device).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/linetr_hip.h"

namespace lt {

constexpr int D = 256;      // descriptor_dim
constexpr int HEADS = 4;
constexpr int DH = 64;      // D / HEADS
constexpr int POOLW = 544;  // pooled row per head: [dbar(256) | abar(256) | p0 | 31 zeros]
constexpr int WAVE = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

inline thread_local std::string g_err;

inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define LT_HIP(expr)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return lt::fail(LINETR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),  \
                      __FILE__, __LINE__);                                                  \
  } while (0)

#define LT_LAUNCH_CHECK()                                                                   \
  do {                                                                                      \
    hipError_t e_ = hipGetLastError();                                                      \
    if (e_ != hipSuccess)                                                                   \
      return lt::fail(LINETR_E_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), \
                      __FILE__, __LINE__);                                                  \
  } while (0)

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device helpers -------------------------------------------------------------------------

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace lt
