// "Row-owner" split-bf16 GEMM (bf16x6):  Y[M,N] = epi([A1 | A2] W^T + bias), fp32 activations in HBM, pre-split weights in the
// split-tile image of lt_gemm_st.h.  The main loop of the fused projection + attention kernel (lt_attn_fused.h) as a GEMM:
//   * a block of FOUR waves owns a 128 x 256 tile; wave w owns rows 32 w .. 32 w + 31 and ALL 256 columns (eight 32 x 32
//     accumulator tiles of the transposed product, weights = MFMA A operand);
//   * a K step's operands travel by LDS-DMA into a two-slot ring: the weight panel (256 rows x 16 k x 3 planes = 24 KiB, one
//     linear span of the K-step-major ST image) and the activations as fp32 (128 rows x 64 B, gathered row by row); a lane
//     reads back the 8 fp32 of ITS OWN row and splits them into the three planes -- no staging registers, no ds_write, and no
//     redundant split work (every other tiling has several waves split the same activation rows);
//   * 64 KiB of LDS and ~200 VGPRs: TWO blocks per CU, so one block's prologue, DMA stalls and epilogue overlap the other
//     block's MFMAs (what the 128 x 128 s tile of lt_gemm_split.h does for short K, here at the full 128 x 256 intensity);
//   * epilogue without LDS: bias rides in the accumulators, one half-wave swap per register pair gives a lane 8 consecutive
//     columns of its row (dwordx4 residual loads and stores); a wave owns COMPLETE rows of an N = 256 problem, so LayerNorm /
//     L2 normalisation are in-lane sums plus one half-wave exchange.
#pragma once
#include "lt_gemm_st.h"

namespace lt {

struct RoGemmArgs {
  const float* A1 = nullptr; int lda1 = 0; int nk1 = 0;     // activations [M][16 nk1 ..]
  const float* A2 = nullptr; int lda2 = 0; int nk2 = 0;     // optional second source, concatenated along K
  const unsigned char* Wst = nullptr;                       // ST image of W [N][16 (nk1 + nk2)]
  const float* bias = nullptr;                              // [N], never null
  const float* R = nullptr; int ldr = 0;                    // residual (added after the activation) or null
  float* Y = nullptr; int ldy = 0;
  int M = 0, N = 0, act = 0;
  int norm = 0;                                             // 1 LayerNorm, 2 L2 (N == 256 only); then + add2
  const float* gamma = nullptr; const float* beta = nullptr; const float* add2 = nullptr; int ldadd2 = 0; float eps = 0.f;
};

constexpr int RO_BM = 128, RO_BN = 256;
constexpr int RO_W_BYTES = RO_BN / 16 * ST_RB;              // 24 576
constexpr int RO_SLOT = RO_W_BYTES + RO_BM * 64;            // 32 768
constexpr int RO_LDS = 2 * RO_SLOT;                         // 65 536

__global__ __launch_bounds__(256, 2) void gemm_ro_kernel(RoGemmArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char ro_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h2 = lane >> 5, lq = lane & 31;
  const int gx = a.N / RO_BN, gy = (a.M + RO_BM - 1) / RO_BM;
  int tile;
  {
    const int ntile = gx * gy, b = blockIdx.x, q = ntile / 8, r = ntile % 8, xcd = b % 8, k = b / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m0 = (tile / gx) * RO_BM, n0 = (tile % gx) * RO_BN;
  const int nk = a.nk1 + a.nk2;
  const int64_t RBW = a.N / 16;
  const int row = m0 + wave * 32 + lq;                     // the row this lane owns

  // ---- DMA plan of a K step: 32 instructions of 1 KiB, 8 per wave: x = wave + 4 d; x < 24 weights (linear), else the
  // activation rows 16 (x - 24) .. + 15 (a lane fetches 16 bytes of row lane >> 2; rows past M repeat the last row)
  const unsigned lane16 = lane * 16;
  auto issue_one = [&](int d, int k, int slot) {
    const int x = wave + 4 * d;
    unsigned char* dst = ro_smem + slot * RO_SLOT + x * 1024;
    if (x < 24) {
      LT_GLDS(a.Wst + ((int64_t)k * RBW + n0 / 16) * ST_RB + x * 1024 + lane16, dst, 0);
    } else {
      int r = m0 + (x - 24) * 16 + (lane >> 2);
      r = r < a.M ? r : a.M - 1;
      const float* g = k < a.nk1 ? a.A1 + (int64_t)r * a.lda1 + k * 16 + (lane & 3) * 4
                                 : a.A2 + (int64_t)r * a.lda2 + (k - a.nk1) * 16 + (lane & 3) * 4;
      LT_GLDS(g, dst, 0);
    }
  };

  // accumulators start at the bias: register 4 b + c of tile i is column n0 + 32 i + 8 b + 4 h2 + c
  f32x16 acc[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) issue_one(d, 0, 0);
  if (nk > 1) {
#pragma unroll
    for (int d = 0; d < 8; ++d) issue_one(d, 1, 1);
  }
  {
    const float* bp = a.bias + n0 + 4 * h2;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bp + i * 32 + 8 * bq);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][4 * bq + c] = v[c];
      }
  }
  if (nk > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int lfrag = ((lane >> 4) & 1) * ST_RB + (lane >> 5) * 256 + (lane & 15) * 16;
  const int zoff = RO_W_BYTES + (wave * 32 + lq) * 64 + h2 * 32;
  constexpr int TW[6] = {2, 1, 0, 1, 0, 0}, TA[6] = {0, 1, 2, 0, 1, 0};     // smallest cross terms first
  auto read_w = [&](int slot, int i, int p, bf16x8 (&wf)[3]) {
    wf[p] = *reinterpret_cast<const bf16x8*>(ro_smem + slot * RO_SLOT + lfrag + i * 2 * ST_RB + p * ST_CHUNK);
  };
  auto split_z = [&](const f32x4& x0, const f32x4& x1, bf16x8 (&zf)[3]) {
    unsigned p0[3], p1[3], p2[3], p3[3];
    split_pair<3>(x0[0], x0[1], p0); split_pair<3>(x0[2], x0[3], p1);
    split_pair<3>(x1[0], x1[1], p2); split_pair<3>(x1[2], x1[3], p3);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      union { bf16x8 v; unsigned w[4]; } u;
      u.w[0] = p0[p]; u.w[1] = p1[p]; u.w[2] = p2[p]; u.w[3] = p3[p];
      zf[p] = u.v;
    }
  };
  bf16x8 zf[3], wfA[3], wfB[3];
  // One K step, fixed issue order: 48 MFMAs; the weight fragments of n-tile i+1 arrive under the six products of n-tile i.
  // Ring of two: the barrier at the end of a step publishes step s+1 (every wave waited for its own DMA first) and frees
  // the slot of step s, which takes step s+2 -- issued from inside the next step, one DMA instruction per MFMA slot.
#pragma unroll 1
  for (int s = 0; s < nk; ++s) {
    const int slot = s & 1;
    {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(ro_smem + slot * RO_SLOT + zoff);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(ro_smem + slot * RO_SLOT + zoff + 16);
#pragma unroll
      for (int p = 0; p < 3; ++p) read_w(slot, 0, p, wfA);
      split_z(x0, x1, zf);
    }
    const bool refill = s >= 1 && s + 1 < nk;              // step s+1 goes into the slot step s-1 left
#pragma unroll
    for (int m = 0; m < 48; ++m) {
      const int i = m / 6, t = m % 6;
      bf16x8 (&wc)[3] = (i & 1) ? wfB : wfA;
      bf16x8 (&wn)[3] = (i & 1) ? wfA : wfB;
      acc[i] = mfma_split<0>(wc[TW[t]], zf[TA[t]], acc[i]);
      if (t < 3 && i < 7) read_w(slot, i + 1, t, wn);
      if (m < 8 && refill) issue_one(m, s + 1, slot ^ 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue, in place in the accumulator registers: activation, half-wave swap -> 8 consecutive columns per lane,
  // residual, (row norm), dwordx4 stores
  const bool live = row < a.M;
  const int rr = live ? row : a.M - 1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (a.act == ACT_RELU) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = fmaxf(acc[i][r], 0.f);
    } else if (a.act == ACT_GELU) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.5f * acc[i][r] * (1.f + erff(acc[i][r] * 0.70710678118654752440f));
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float x0 = acc[i][c], x1 = acc[i][4 + c], x2 = acc[i][8 + c], x3 = acc[i][12 + c];
      halves_swap(x0, x1);
      halves_swap(x2, x3);
      acc[i][c] = x0; acc[i][4 + c] = x1; acc[i][8 + c] = x2; acc[i][12 + c] = x3;
    }
    // now acc[i][0..7] = columns n0 + 32 i + 8 h2 .. + 8 and acc[i][8..15] = columns n0 + 32 i + 16 + 8 h2 .. + 8 of row `row`
    if (a.R) {
      const float* rp = a.R + (int64_t)rr * a.ldr + n0 + i * 32 + 8 * h2;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp + 16 * g), r1 = *reinterpret_cast<const f32x4*>(rp + 16 * g + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc[i][8 * g + c] += r0[c]; acc[i][8 * g + 4 + c] += r1[c]; }
      }
    }
    if (!a.norm && live) {
      float* yp = a.Y + (int64_t)row * a.ldy + n0 + i * 32 + 8 * h2;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        sk_store16(yp + 16 * g, f32x4{acc[i][8 * g], acc[i][8 * g + 1], acc[i][8 * g + 2], acc[i][8 * g + 3]});
        sk_store16(yp + 16 * g + 4, f32x4{acc[i][8 * g + 4], acc[i][8 * g + 5], acc[i][8 * g + 6], acc[i][8 * g + 7]});
      }
    }
  }
  if (!a.norm) return;
  // a wave owns complete rows (N == 256): the lane pair (lq, lq + 32) holds the whole row
  float mean = 0.f, scale;
  if (a.norm == 1) {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[i][r];
    mean = xor32_sum(sum) * (1.f / 256);
    float qq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { const float d = acc[i][r] - mean; qq += d * d; }
    scale = 1.f / sqrtf(xor32_sum(qq) * (1.f / 256) + a.eps);
  } else {
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[i][r] * acc[i][r];
    scale = 1.f / fmaxf(sqrtf(xor32_sum(sum)), 1e-12f);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int col = i * 32 + 16 * g + 8 * h2;
      f32x4 o0, o1;
#pragma unroll
      for (int c = 0; c < 4; ++c) { o0[c] = (acc[i][8 * g + c] - mean) * scale; o1[c] = (acc[i][8 * g + 4 + c] - mean) * scale; }
      if (a.norm == 1) {
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.gamma + col), g1 = *reinterpret_cast<const f32x4*>(a.gamma + col + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.beta + col), b1 = *reinterpret_cast<const f32x4*>(a.beta + col + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { o0[c] = o0[c] * g0[c] + b0[c]; o1[c] = o1[c] * g1[c] + b1[c]; }
      }
      if (a.add2) {
        const float* ap = a.add2 + (int64_t)rr * a.ldadd2 + col;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(ap), r1 = *reinterpret_cast<const f32x4*>(ap + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { o0[c] += r0[c]; o1[c] += r1[c]; }
      }
      if (live) {
        float* yp = a.Y + (int64_t)row * a.ldy + col;
        sk_store16(yp, o0);
        sk_store16(yp + 4, o1);
      }
    }
}

inline int gemm_ro_launch(const RoGemmArgs& a, hipStream_t st) {
  if (a.M <= 0) return 0;
  if (a.N % RO_BN != 0 || a.nk1 < 1 || (a.A2 && a.nk2 < 1) || (!a.A2 && a.nk2 != 0) || !a.bias || !a.Wst || a.ldy % 4 || a.lda1 % 4 ||
      (a.A2 && a.lda2 % 4) || (a.R && a.ldr % 4) || (a.norm && (a.N != 256 || a.ldadd2 % 4)))
    return fail(LINETR_E_ARG, "gemm_ro: unsupported shape M=%d N=%d nk=%d+%d", a.M, a.N, a.nk1, a.nk2);
  static unsigned long long attr_done = 0;
  const unsigned long long dev_bit = current_device_bit();
  if (!(attr_done & dev_bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ro_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RO_LDS);
    attr_done |= dev_bit;
  }
  hipLaunchKernelGGL(gemm_ro_kernel, dim3((a.N / RO_BN) * cdiv(a.M, RO_BM)), dim3(256), RO_LDS, st, a);
  LT_LAUNCH_CHECK();
  return 0;
}

}  // namespace lt
