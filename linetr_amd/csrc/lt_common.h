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

// The shipped library (liblinetr_hip.so) is built WITHOUT LINETR_EXPERIMENTS: no tuning switches, none of the kernels that
// were measured and lost (stream-K tail, fused signature MLP, split-tile / LDS-DMA path, GEMM chains, side stream, two-pass
// pooling, four-wave tiles).  `python -m linetr_amd.build --experiments` builds liblinetr_hip_experiments.so from the same
// sources with everything in, for tools/ and tests/test_gpu_experiments.py.  LT_XENV is getenv there and a constant null
// pointer in the product, so every switch folds away.
#ifdef LINETR_EXPERIMENTS
#define LT_XENV(name) getenv(name)
#else
#define LT_XENV(name) (static_cast<const char*>(nullptr))
#endif

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
// bit of the calling thread's current HIP device (function attributes such as the dynamic-LDS opt-in are per device)
inline unsigned long long current_device_bit() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return 1ull << (dev & 63);
}
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- device helpers -------------------------------------------------------------------------

// constants of the CLS-row attention pooling (lt_model.h: the cls_pool kernels; prepared in float64 by linetr_create)
struct ClsPoolConst {
  const float* U;     // [4][256]  u_h
  const float* U2;    // [4][256]  W5^T u_h
  float c_tok[4];     // u_h.b5 + c_h   (additive constant of token rows)
  float s_cls[4];     // u_h.cls + c_h  (score of the CLS key, row 0)
};

constexpr float LOG2E = 1.44269504088896340736f;

// Wave-wide reductions on the VALU's DPP paths (no LDS crossbar): xor-1 / xor-2 inside quads, half-row and row
// mirrors give every lane its 16-lane row total, row_bcast15 / row_bcast31 fold the four rows into row 3, and lane 63 is
// broadcast through an SGPR.  7 dependent VALU steps instead of 6 dependent ds_bpermute round trips (__shfl_xor).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v, float masked) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, masked), __builtin_bit_cast(int, v),
                                                               CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_get<0xB1, 0xf>(v, 0.f);    // quad_perm [1,0,3,2]
  v += dpp_get<0x4E, 0xf>(v, 0.f);    // quad_perm [2,3,0,1]
  v += dpp_get<0x141, 0xf>(v, 0.f);   // row_half_mirror
  v += dpp_get<0x140, 0xf>(v, 0.f);   // row_mirror
  v += dpp_get<0x142, 0xa>(v, 0.f);   // row_bcast15 into rows 1, 3
  v += dpp_get<0x143, 0xc>(v, 0.f);   // row_bcast31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// Exchange between the two 32-lane halves of a wave on gfx950's v_permlane32_swap (no LDS round trip): after the swap one
// register holds the lower half's values in both halves and the other the upper half's, so max / sum of the pair is
// the xor-32 butterfly step, identical in both halves.
// (Inline asm on purpose: with this toolchain __builtin_amdgcn_permlane32_swap hands back the same register for both
// results -- tools/ubench/permlane_test.hip.  The s_nops cover the VALU-write -> permlane-read wait states the
// compiler would otherwise insert itself.)
__device__ __forceinline__ void halves_of(float x, float& lo, float& hi) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
  lo = __builtin_bit_cast(float, a);
  hi = __builtin_bit_cast(float, b);
}
// a's upper half <-> b's lower half (raw v_permlane32_swap on two different registers)
__device__ __forceinline__ void halves_swap(float& a, float& b) {
  unsigned x = __builtin_bit_cast(unsigned, a), y = __builtin_bit_cast(unsigned, b);
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  a = __builtin_bit_cast(float, x);
  b = __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float xor32_max(float x) { float a, b; halves_of(x, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor32_sum(float x) { float a, b; halves_of(x, a, b); return a + b; }

// N independent sums at once: the N chains interleave, so the 2 wait states a DPP read needs after the VALU write of
// its source are filled with useful work instead of s_nops.
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_get<0xB1, 0xf>(v[i], 0.f);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_get<0x4E, 0xf>(v[i], 0.f);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_get<0x141, 0xf>(v[i], 0.f);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_get<0x140, 0xf>(v[i], 0.f);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_get<0x142, 0xa>(v[i], 0.f);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] += dpp_get<0x143, 0xc>(v[i], 0.f);
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[i]), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_get<0xB1, 0xf>(v, v));
  v = fmaxf(v, dpp_get<0x4E, 0xf>(v, v));
  v = fmaxf(v, dpp_get<0x141, 0xf>(v, v));
  v = fmaxf(v, dpp_get<0x140, 0xf>(v, v));
  v = fmaxf(v, dpp_get<0x142, 0xa>(v, v));
  v = fmaxf(v, dpp_get<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

}  // namespace lt
