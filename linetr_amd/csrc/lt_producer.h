// Fused SuperPoint head post-processing: turns the two raw head outputs into the dense maps
// the line tokeniser consumes, in the layout it wants.
//   score head  models/superpoint.py:161-167   softmax over 65 channels, drop the dustbin, depth-to-space (8x8 cells)
//   desc head   models/superpoint.py:190-193   F.normalize(p=2, dim=1) -- here fused with the NCHW -> NHWC transposition
//                                              that linetr_describe otherwise does in a separate pass
// Both kernels are HBM-bound by design: every input byte is read once, every output byte written once.
#pragma once
#include "lt_common.h"

namespace lt {

// One block = 64 consecutive cells (h*Wc + w) of one image x all 256 channels.  LDS tile [256][65]: the stride of 65
// floats makes the NCHW-side accesses (lanes along cells) and the NHWC-side accesses (lanes along channels) both
// bank-conflict-free.
__global__ __launch_bounds__(256) void sp_desc_head_kernel(const float* __restrict__ raw, float* __restrict__ nhwc,
                                                           float* __restrict__ nchw, int HW) {
  constexpr int LS = 65;
  __shared__ float tile[D * LS];
  __shared__ float part[4 * 64];
  const int t = threadIdx.x, b = blockIdx.y, p0 = blockIdx.x * 64;
  const float* src = raw + (int64_t)b * D * HW;
  const bool vec = (HW % 4) == 0;
  // phase 1: 16 channels per pass, 16 lanes x 4 cells per channel
  {
    const int cl = t >> 4, p4 = (t & 15) * 4;
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
      const int c = pass * 16 + cl;
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
  }
  __syncthreads();
  // phase 2: squared norm of every cell over the 256 channels (4 partial sums of 64 channels)
  {
    const int p = t & 63, g = t >> 6;
    float ss = 0.f;
#pragma unroll 8
    for (int c = g * 64; c < g * 64 + 64; ++c) { const float v = tile[c * LS + p]; ss += v * v; }
    part[g * 64 + p] = ss;
  }
  __syncthreads();
  if (t < 64) {
    const float n = sqrtf(part[t] + part[64 + t] + part[128 + t] + part[192 + t]);
    part[t] = 1.f / fmaxf(n, 1e-12f);   // F.normalize: x / max(||x||, eps)
  }
  __syncthreads();
  // phase 3a: NHWC, one cell = 1 KiB contiguous; a wave writes 256 B per instruction
  if (nhwc) {
    const int lane = t & 63, w = t >> 6;
    float* dst = nhwc + ((int64_t)b * HW + p0) * D;
    for (int p = w; p < 64; p += 4) {
      if (p0 + p >= HW) break;
      const float inv = part[p];
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[(int64_t)p * D + lane + 64 * i] = tile[(lane + 64 * i) * LS + p] * inv;
    }
  }
  // phase 3b: normalised NCHW (the reference's 'dense_descriptor' key), same access pattern as phase 1
  if (nchw) {
    const int cl = t >> 4, p4 = (t & 15) * 4;
    float* dstb = nchw + (int64_t)b * D * HW;
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
      const int c = pass * 16 + cl;
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

// One block = 64 consecutive cells of one image x 65 logits.  Thread (cell p, quarter g) computes the cell's softmax
// statistics (redundantly per quarter: 65 LDS reads) and writes sub-pixel rows dy = 2g, 2g+1 of its 8x8 patch.
__global__ __launch_bounds__(256) void sp_score_head_kernel(const float* __restrict__ logits, float* __restrict__ score,
                                                            int Hc, int Wc) {
  constexpr int LS = 65;
  __shared__ float tile[65 * LS];
  const int t = threadIdx.x, b = blockIdx.y, p0 = blockIdx.x * 64, HW = Hc * Wc;
  const float* src = logits + (int64_t)b * 65 * HW;
  for (int idx = t; idx < 65 * 64; idx += 256) {
    const int c = idx >> 6, p = idx & 63;
    tile[c * LS + p] = p0 + p < HW ? src[(int64_t)c * HW + p0 + p] : 0.f;
  }
  __syncthreads();
  const int p = t & 63, g = t >> 6;
  if (p0 + p >= HW) return;
  float m = -INFINITY;
  for (int c = 0; c < 65; ++c) m = fmaxf(m, tile[c * LS + p]);
  float sum = 0.f;
  for (int c = 0; c < 65; ++c) sum += expf(tile[c * LS + p] - m);
  const float inv = 1.f / sum;
  const int cell = p0 + p, hh = cell / Wc, ww = cell % Wc;
  const int W8 = Wc * 8;
  float* dst = score + (int64_t)b * (Hc * 8) * W8 + (int64_t)(hh * 8) * W8 + ww * 8;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int dy = 2 * g + r;
    f32x4 lo, hi;
#pragma unroll
    for (int dx = 0; dx < 4; ++dx) {
      lo[dx] = expf(tile[(dy * 8 + dx) * LS + p] - m) * inv;
      hi[dx] = expf(tile[(dy * 8 + 4 + dx) * LS + p] - m) * inv;
    }
    float* d = dst + (int64_t)dy * W8;
    *reinterpret_cast<f32x4*>(d) = lo;       // 8 * ww floats into a row of 8 * Wc: always 16-byte aligned
    *reinterpret_cast<f32x4*>(d + 4) = hi;
  }
}

}  // namespace lt
