// Split-tile ("ST") operand image for gfx950: a matrix pre-split into three bf16 planes and laid out as the LDS image of MFMA
// fragments, so that a K step of it travels HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR staging, no split VALU, no
// ds_write) or sits in registers / LDS for a whole launch.  Product users: the q/k/v weights of the fused projection + attention kernel
// (lt_attn_fused.h), the positional encoders' weights and inter-layer activations (lt_tokmlp.h), the weight-stationary K = 128 GEMM
// (lt_gemm_ws.h).  The full ST GEMM (activations kept in this format in HBM) was measured and not shipped: experiments/csrc/lt_gemm_st.h.
//
// ST image of a matrix X[R][K] (fp32 values, K % 16 == 0), x = p0 + p1 + p2 exactly (bf16 planes, lt_gemm_split.h):
//   chunk(kt, rb, p) = 512 B at byte ((kt * RB + rb) * 3 + p) * 512     kt = k / 16, rb = row / 16, p = plane,
//                      RB = row blocks of the image (rows padded to a multiple of 128: st_row_blocks)
//   inside a chunk   = [q = 0..1][r = 0..15] x 16 B; piece (q, r) holds plane p of X[16 rb + r][16 kt + 8 q .. +8]
// K-step-major on purpose: for one 16-wide K step the chunks of consecutive row blocks are contiguous, so the 128-row
// (or 256-row) operand panel a block needs per K step is ONE contiguous span that LDS-DMA copies linearly (destination =
// wave-uniform base + lane * 16; it cannot pad or scatter).
// Why this chunk shape: an MFMA 32x32x16 fragment (lane -> row lane & 31, k = 8 (lane >> 5) .. +8) is one ds_read_b128
// whose 16-lane groups ({0-3,12-15,20-27} ...) hit 16 distinct 16-byte slots of a 256-byte bank row -- rows 0-15 of a
// (chunk, q) are 256 contiguous bytes and the second row block of a 32-row fragment sits 1536 B = 6 x 256 B away --
// conflict-free without padding.
//
#pragma once
#include <type_traits>

#include "lt_gemm_split.h"

namespace lt {

constexpr int ST_CHUNK = 512;
constexpr int ST_RB = 3 * ST_CHUNK;          // the three planes of one (K step, row block)
inline __host__ __device__ int64_t st_row_blocks(int64_t rows) { return (rows + 127) / 128 * 8; }
inline int64_t st_bytes(int64_t rows, int K) { return st_row_blocks(rows) * (int64_t)(K / 16) * ST_RB; }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// fp32 [rows][ld] -> ST image (pad rows are zero).  One thread per piece; a 32-lane group writes one chunk per plane.
__global__ __launch_bounds__(256) void to_st_kernel(const float* __restrict__ X, int ld, int rows, int nkt,
                                                    unsigned char* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int l = (int)(gid & 31);
  const int64_t c = gid >> 5;                       // (kt, rb)
  const int64_t RB = st_row_blocks(rows);
  if (c >= RB * nkt) return;
  const int kt = (int)(c / RB);
  const int64_t rb = c % RB;
  const int q = l >> 4, r = l & 15;
  const int64_t row = rb * 16 + r;
  f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
  if (row < rows) {
    const float* p = X + row * ld + kt * 16 + q * 8;
    x0 = *reinterpret_cast<const f32x4*>(p);
    x1 = *reinterpret_cast<const f32x4*>(p + 4);
  }
  unsigned a[3], b[3], cc[3], d[3];
  split_pair<3>(x0[0], x0[1], a); split_pair<3>(x0[2], x0[3], b);
  split_pair<3>(x1[0], x1[1], cc); split_pair<3>(x1[2], x1[3], d);
  unsigned char* dst = out + c * ST_RB + l * 16;
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + p * ST_CHUNK) = u32x4{a[p], b[p], cc[p], d[p]};
}

// two bf16 packed in a dword -> fp32 pair
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// 8 values of an ST piece back to fp32: (p2 + p1) + p0 is exact (p1 + p2 is the fp32 remainder x - p0)
__device__ __forceinline__ void st_piece_to_f32(const u32x4& p0, const u32x4& p1, const u32x4& p2, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[2 * e] = (bf_lo(p2[e]) + bf_lo(p1[e])) + bf_lo(p0[e]);
    v[2 * e + 1] = (bf_hi(p2[e]) + bf_hi(p1[e])) + bf_hi(p0[e]);
  }
}

// ST image -> fp32 [rows][ld]
__global__ __launch_bounds__(256) void from_st_kernel(const unsigned char* __restrict__ in, int rows, int nkt,
                                                      float* __restrict__ X, int ld) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int l = (int)(gid & 31);
  const int64_t c = gid >> 5;
  const int64_t RB = st_row_blocks(rows);
  if (c >= RB * nkt) return;
  const int kt = (int)(c / RB);
  const int64_t rb = c % RB;
  const int q = l >> 4, r = l & 15;
  const int64_t row = rb * 16 + r;
  if (row >= rows) return;
  const unsigned char* src = in + c * ST_RB + l * 16;
  float v[8];
  st_piece_to_f32(*reinterpret_cast<const u32x4*>(src), *reinterpret_cast<const u32x4*>(src + ST_CHUNK),
                  *reinterpret_cast<const u32x4*>(src + 2 * ST_CHUNK), v);
  float* dst = X + row * ld + kt * 16 + q * 8;
  *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

#define LT_GLDS(gp, lp, off)                                                                                      \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp),                           \
                                   (__attribute__((address_space(3))) void*)(lp), 16, (off), 0)

// swap the upper half of a with the lower half of b (see halves_of in lt_common.h for why this is inline asm)
__device__ __forceinline__ void swap32(float& a, float& b) {
  unsigned x = __builtin_bit_cast(unsigned, a), y = __builtin_bit_cast(unsigned, b);
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  a = __builtin_bit_cast(float, x);
  b = __builtin_bit_cast(float, y);
}

}  // namespace lt
