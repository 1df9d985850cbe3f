// Fused SuperPoint head post-processing: turns the two raw head outputs into the dense maps
// the line tokeniser consumes, in the layout it wants.
//   score head  models/superpoint.py:161-167   softmax over 65 channels, drop the dustbin, depth-to-space (8x8 cells)
//   desc head   models/superpoint.py:190-193   F.normalize(p=2, dim=1) -- here fused with the NCHW -> NHWC transposition
//                                              that linetr_describe otherwise does in a separate pass
// Both kernels are HBM-bound by design: every input byte is read once, every output byte written once.
#pragma once
#include "lt_common.h"

namespace lt {

// One block = CELLS consecutive cells (h*Wc + w) of one image x all 256 channels.  LDS tile [256][CELLS + 1]: the odd
// stride makes the NCHW-side accesses (lanes along cells) and the NHWC-side accesses (lanes along channels) both
// bank-conflict-free.  CELLS = 32 (33 KiB of LDS, 4 blocks per CU) keeps more loads in flight than 64 (2 blocks per
// CU): 5.4 vs 4.3 TB/s on a cfg3-sized batch (tools/producer_bench.py).
template <int CELLS>
__global__ __launch_bounds__(256) void sp_desc_head_kernel(const float* __restrict__ raw, float* __restrict__ nhwc,
                                                           float* __restrict__ nchw, int HW) {
  constexpr int LS = CELLS + 1;
  constexpr int LPC = CELLS / 4;          // lanes per channel row (one float4 each)
  constexpr int CPP = 256 / LPC;          // channels per pass
  constexpr int NG = 256 / CELLS;         // partial sums per cell
  __shared__ float tile[D * LS];
  __shared__ float part[256];
  const int t = threadIdx.x, b = blockIdx.y, p0 = blockIdx.x * CELLS;
  const float* src = raw + (int64_t)b * D * HW;
  const bool vec = (HW % 4) == 0;
  const int cl = t / LPC, p4 = (t % LPC) * 4;
  // phase 1: CPP channels per pass, LPC lanes x 4 cells per channel
#pragma unroll 4
  for (int pass = 0; pass < D / CPP; ++pass) {
    const int c = pass * CPP + cl;
    const float* s = src + (int64_t)c * HW + p0 + p4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (vec && p0 + p4 + 3 < HW) v = *reinterpret_cast<const f32x4*>(s);
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (p0 + p4 + j < HW) v[j] = s[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[c * LS + p4 + j] = v[j];
  }
  __syncthreads();
  // phase 2: squared norm of every cell over the 256 channels (NG partial sums of D / NG channels)
  {
    const int p = t % CELLS, g = t / CELLS;
    float ss = 0.f;
#pragma unroll 8
    for (int c = g * (D / NG); c < (g + 1) * (D / NG); ++c) { const float v = tile[c * LS + p]; ss += v * v; }
    part[g * CELLS + p] = ss;
  }
  __syncthreads();
  if (t < CELLS) {
    float ss = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) ss += part[g * CELLS + t];
    part[t] = 1.f / fmaxf(sqrtf(ss), 1e-12f);   // F.normalize: x / max(||x||, eps)   (part[t] is only read by t above)
  }
  __syncthreads();
  // phase 3a: NHWC, one cell = 1 KiB contiguous; a wave writes 256 B per instruction
  if (nhwc) {
    const int lane = t & 63, w = t >> 6;
    float* dst = nhwc + ((int64_t)b * HW + p0) * D;
    for (int p = w; p < CELLS; p += 4) {
      if (p0 + p >= HW) break;
      const float inv = part[p];
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[(int64_t)p * D + lane + 64 * i] = tile[(lane + 64 * i) * LS + p] * inv;
    }
  }
  // phase 3b: normalised NCHW (the reference's 'dense_descriptor' key), same access pattern as phase 1
  if (nchw) {
    float* dstb = nchw + (int64_t)b * D * HW;
#pragma unroll 4
    for (int pass = 0; pass < D / CPP; ++pass) {
      const int c = pass * CPP + cl;
      float* d = dstb + (int64_t)c * HW + p0 + p4;
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = tile[c * LS + p4 + j] * part[p4 + j];
      if (vec && p0 + p4 + 3 < HW) *reinterpret_cast<f32x4*>(d) = v;
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (p0 + p4 + j < HW) d[j] = v[j];
      }
    }
  }
}

// One block = 64 consecutive cells of one image x 65 logits.  Thread (cell p, quarter g) owns the 16 channels of
// sub-pixel rows dy = 2g, 2g+1 of its cell's 8x8 patch (quarter 3 also the dustbin): one expf per output, the
// max and the sum are combined across the four quarters through LDS.
__global__ __launch_bounds__(256) void sp_score_head_kernel(const float* __restrict__ logits, float* __restrict__ score,
                                                            int Hc, int Wc) {
  constexpr int LS = 65;
  __shared__ float tile[65 * LS];
  __shared__ float red[2][4 * 64];
  const int t = threadIdx.x, b = blockIdx.y, p0 = blockIdx.x * 64, HW = Hc * Wc;
  const float* src = logits + (int64_t)b * 65 * HW;
  const bool vec = (HW % 4) == 0;
  {  // 16 lanes x 4 cells per channel row, 16 channels per pass (+ the dustbin row)
    const int cl = t >> 4, p4 = (t & 15) * 4;
#pragma unroll
    for (int pass = 0; pass < 5; ++pass) {
      const int c = pass * 16 + cl;
      if (c < 65) {
        const float* sp = src + (int64_t)c * HW + p0 + p4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (vec && p0 + p4 + 3 < HW) v = *reinterpret_cast<const f32x4*>(sp);
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (p0 + p4 + j < HW) v[j] = sp[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[c * LS + p4 + j] = v[j];
      }
    }
  }
  __syncthreads();
  const int p = t & 63, g = t >> 6;
  float x[17];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = tile[(g * 16 + i) * LS + p];
  x[16] = g == 3 ? tile[64 * LS + p] : -INFINITY;
  float m = x[0];
#pragma unroll
  for (int i = 1; i < 17; ++i) m = fmaxf(m, x[i]);
  red[0][g * 64 + p] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0][p], red[0][64 + p]), fmaxf(red[0][128 + p], red[0][192 + p]));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 17; ++i) { x[i] = expf(x[i] - m); sum += x[i]; }   // expf(-inf) = 0 for the missing dustbin slots
  red[1][g * 64 + p] = sum;
  __syncthreads();
  if (p0 + p >= HW) return;
  const float inv = 1.f / ((red[1][p] + red[1][64 + p]) + (red[1][128 + p] + red[1][192 + p]));
  const int cell = p0 + p, hh = cell / Wc, ww = cell % Wc;
  const int W8 = Wc * 8;
  float* dst = score + (int64_t)b * (Hc * 8) * W8 + (int64_t)(hh * 8) * W8 + ww * 8;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    float* d = dst + (int64_t)(2 * g + r) * W8;     // 8 * ww floats into a row of 8 * Wc: always 16-byte aligned
    *reinterpret_cast<f32x4*>(d) = f32x4{x[r * 8 + 0] * inv, x[r * 8 + 1] * inv, x[r * 8 + 2] * inv, x[r * 8 + 3] * inv};
    *reinterpret_cast<f32x4*>(d + 4) = f32x4{x[r * 8 + 4] * inv, x[r * 8 + 5] * inv, x[r * 8 + 6] * inv, x[r * 8 + 7] * inv};
  }
}

}  // namespace lt
