// Small-M split-bf16 GEMM (single image pair: M = a few hundred rows).  Same contract and arithmetic as
// gemm_split_kernel (lt_gemm_split.h): Y = epi(A W^T + bias), fp32 in/out, PL bf16/fp16 planes per operand.
//
// Why a second kernel: with M ~ 400 the tiled kernel has 28-112 blocks that each walk the whole K range through
// LDS with a block barrier per 32-wide K tile -- ~0.9 us per K tile of pure latency, 14 us per GEMM, 28 GEMMs per
// forward.  Here nothing is staged and nothing is synchronised inside the main loop:
//   * one block = one 32 x 32 output tile, so a 400 x 512 GEMM still launches 208 blocks;
//   * the block's 4 waves split K four ways; every lane fetches its MFMA fragments straight from global memory
//     (activations: 2 x dwordx4 of fp32 per 16-wide K step, split into planes in registers; weights: one 16-byte
//     piece per plane of the pre-split image [N][K/32][PL][32]) with up to 4 K tiles (40 loads) in flight per wave;
//   * the four partial 32 x 32 accumulators are summed through LDS in a FIXED order (deterministic) and the epilogue
//     (bias / activation / residual) writes float4 rows.
// A tiles are re-read N/32 times and W tiles M/32 times, all from L2 -- fine for the small problems this kernel is
// dispatched for (run_gemm: fewer than 256 64 x 64 tiles).
#pragma once
#include "lt_gemm_split.h"

namespace lt {

template <int PL, int FMT>
__global__ __launch_bounds__(256) void gemm_split_small_kernel(SplitGemmArgs sa) {
  const GemmArgs& g = sa.g;
  __shared__ float red[4][32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gx = g.N / 32;
  const int m0 = (blockIdx.x / gx) * 32, n0 = (blockIdx.x % gx) * 32;
  const int grp = blockIdx.y;
  const float* A = g.A + grp * g.gA;
  const float* A2 = g.A2 ? g.A2 + grp * g.gA : nullptr;
  const unsigned char* Wsp = sa.Wsp + grp * sa.gWsp;
  const int K1 = g.A2 ? g.K1 : g.K;
  const int nk = g.K / 32;
  const int kt0 = nk * wave / 4, kt1 = nk * (wave + 1) / 4;   // this wave's K tiles

  const int frow = lane & 31, half = lane >> 5;
  int arow = m0 + frow;
  arow = arow < g.M ? arow : g.M - 1;
  const float* a_row = A + (int64_t)arow * g.lda;
  const float* a2_row = A2 ? A2 + (int64_t)arow * g.lda2 : nullptr;
  const unsigned char* w_row = Wsp + (int64_t)(n0 + frow) * nk * (PL * 64) + half * 16;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  constexpr int GRP = 4;                       // K tiles in flight per wave
  for (int kt = kt0; kt < kt1; kt += GRP) {
    f32x4 ra[GRP][2][2];                       // [tile][step][low/high 4 floats of this lane's 8]
    f32x4 rw[GRP][2][PL];
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      if (kt + u < kt1) {                      // wave-uniform
        const int k0 = (kt + u) * 32;
        const float* src = k0 < K1 ? a_row + k0 : a2_row + (k0 - K1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          ra[u][s][0] = *reinterpret_cast<const f32x4*>(src + s * 16 + half * 8);
          ra[u][s][1] = *reinterpret_cast<const f32x4*>(src + s * 16 + half * 8 + 4);
#pragma unroll
          for (int p = 0; p < PL; ++p)
            rw[u][s][p] = *reinterpret_cast<const f32x4*>(w_row + (int64_t)(kt + u) * (PL * 64) + p * 64 + s * 32);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      if (kt + u < kt1) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          unsigned a[PL], b[PL], c[PL], d[PL];
          split_pair<PL, FMT>(ra[u][s][0][0], ra[u][s][0][1], a);
          split_pair<PL, FMT>(ra[u][s][0][2], ra[u][s][0][3], b);
          split_pair<PL, FMT>(ra[u][s][1][0], ra[u][s][1][1], c);
          split_pair<PL, FMT>(ra[u][s][1][2], ra[u][s][1][3], d);
          bf16x8 af[PL], bf[PL];
#pragma unroll
          for (int p = 0; p < PL; ++p) {
            union { bf16x8 v; unsigned w[4]; } x;
            x.w[0] = a[p]; x.w[1] = b[p]; x.w[2] = c[p]; x.w[3] = d[p];
            af[p] = x.v;
            bf[p] = __builtin_bit_cast(bf16x8, rw[u][s][p]);
          }
          // cross terms with pa + pb <= PL-1, smallest first (same order as gemm_split_kernel)
#pragma unroll
          for (int ord = PL - 1; ord >= 0; --ord)
#pragma unroll
            for (int pa = PL - 1; pa >= 0; --pa) {
              const int pb = ord - pa;
              if (pb < 0 || pb >= PL) continue;
              acc = mfma_split<FMT>(af[pa], bf[pb], acc);
            }
        }
      }
    }
  }

  // partial sums -> LDS (row stride 33: the 16 ds_write_b32 of a wave hit distinct banks)
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + frow] = acc[r];
  __syncthreads();
  const int row = tid >> 3, c4 = (tid & 7) * 4;
  if (m0 + row < g.M) {
    const float* bias = g.bias ? g.bias + grp * g.gBias : nullptr;
    float* Y = g.Y + grp * g.gY;
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int o = row * 33 + c4 + c;
      v[c] = ((red[0][o] + red[1][o]) + red[2][o]) + red[3][o];   // fixed order
      if (bias) v[c] += bias[n0 + c4 + c];
      if (g.act == ACT_RELU) v[c] = fmaxf(v[c], 0.f);
      else if (g.act == ACT_GELU) v[c] = 0.5f * v[c] * (1.f + erff(v[c] * 0.70710678118654752440f));
      else if (g.act == ACT_DIST) v[c] = fmaxf(2.f - 2.f * v[c], 0.f);
    }
    if (g.R) {
      const f32x4 rr = *reinterpret_cast<const f32x4*>(g.R + (int64_t)(m0 + row) * g.ldr + n0 + c4);
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] += rr[c];
    }
    *reinterpret_cast<f32x4*>(Y + (int64_t)(m0 + row) * g.ldy + n0 + c4) = v;
  }
}

// dispatched when even 64 x 64 tiles would leave most CUs without a block
inline bool small_gemm_wins(const GemmArgs& g, int groups) {
  static const bool off = getenv("LINETR_NO_SMALL_GEMM") != nullptr;   // tuning aid
  if (off || g.N % 32 != 0 || g.K % 32 != 0 || g.K < 128) return false;
  if (g.lda % 4 != 0 || g.ldy % 4 != 0 || (g.R && g.ldr % 4 != 0) || (g.A2 && (g.lda2 % 4 != 0 || g.K1 % 32 != 0))) return false;
  return (int64_t)cdiv(g.M, 64) * (g.N / 64) * groups < 256;
}

template <int PL, int FMT>
inline void gemm_split_small_launch(const SplitGemmArgs& sa, int groups, hipStream_t st) {
  dim3 grid((unsigned)((sa.g.N / 32) * cdiv(sa.g.M, 32)), (unsigned)groups);
  hipLaunchKernelGGL((gemm_split_small_kernel<PL, FMT>), grid, dim3(256), 0, st, sa);
}

}  // namespace lt
