// Small-M split-bf16 GEMM (single image pair: M = a few hundred rows).  Same contract and arithmetic as
// gemm_split_kernel (lt_gemm_split.h): Y = epi(A W^T + bias), fp32 in/out, PL bf16/fp16 planes per operand.
//
// Why a second kernel: with M ~ 400 the tiled kernel has 28-112 blocks that each walk the whole K range through
// LDS with a block barrier per 32-wide K tile -- ~0.9 us per K tile of pure latency, 14 us per GEMM, 28 GEMMs per
// forward.  Here nothing is staged and nothing is synchronised inside the main loop:
//   * one block = one 32 x 32 output tile, so a 400 x 512 GEMM still launches 208 blocks;
//   * the block's 4 waves split K four ways; every wave stages its own K tiles through a WAVE-PRIVATE LDS region:
//     coalesced global loads (a K tile of an activation row is one 128-byte line, of a weight row PL*64 contiguous
//     bytes; fetching MFMA fragments straight from global memory made every lane its own cache line and the address
//     unit the bottleneck), up to 4 K tiles (40 loads) in flight per wave in registers, then ds_write -> ds_read of the
//     fragments -> split -> MFMAs with no block barrier, because the LDS executes one wave's instructions in order;
//   * the four partial 32 x 32 accumulators are summed through LDS in a FIXED order (deterministic) and the epilogue
//     (bias / activation / residual) writes float4 rows.
// A tiles are re-read N/32 times and W tiles M/32 times, all from L2 -- fine for the small problems this kernel is
// dispatched for (run_gemm: fewer than 256 64 x 64 tiles).
#pragma once
#include "lt_gemm_split.h"

namespace lt {

template <int PL, int FMT, int NWV = 4, int NBUF = (NWV == 4 ? 2 : 1)>
__global__ __launch_bounds__(NWV * 64) void gemm_split_small_kernel(SplitGemmArgs sa) {
  const GemmArgs& g = sa.g;
  constexpr int RS = PL * 64 + 16;                 // W row stride in LDS (bytes), as in gemm_split_kernel
  constexpr int AS = 36;                           // A row stride in LDS (floats): 4 * odd -> conflict-free ds_read_b128
  constexpr int A_BYTES = 32 * AS * 4, W_BYTES = 32 * RS;
  constexpr int W_PCS = PL * 4;                    // 16-byte pieces per W row and K tile
  constexpr int W_LD = (32 * W_PCS + 63) / 64;     // W load instructions per K tile
  // NBUF staging buffers per wave.  8 waves: one each (LDS), every wave has half the K tiles; 4 waves with ONE buffer is
  // the 62 KB variant of which two blocks share a CU (grids of more than one block per CU, see the launcher)
  __shared__ __attribute__((aligned(16))) unsigned char stage[NWV][NBUF][A_BYTES + W_BYTES];   // [wave][buffer]
  __shared__ float red[NWV][32 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gx = g.N / 32;
  const int m0 = (blockIdx.x / gx) * 32, n0 = (blockIdx.x % gx) * 32;
  const int grp = blockIdx.y;
  const float* A = g.A + grp * g.gA;
  const float* A2 = g.A2 ? g.A2 + grp * g.gA : nullptr;
  const unsigned char* Wsp = sa.Wsp + grp * sa.gWsp;
  const int K1 = g.A2 ? g.K1 : g.K;
  const int nk = g.K / 32;
  const int kt0 = nk * wave / NWV, kt1 = nk * (wave + 1) / NWV;   // this wave's K tiles

  // global loads, coalesced: A rows as 8 lanes x 16 B (a K tile of a row is one 128-byte line), W rows as W_PCS lanes x 16 B
  const int a_r = lane >> 3, a_c = (lane & 7) * 4;
  int64_t a_off[4], a2_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = m0 + a_r + 8 * i;
    r = r < g.M ? r : g.M - 1;
    a_off[i] = (int64_t)r * g.lda + a_c;
    a2_off[i] = (int64_t)r * g.lda2 + a_c;
  }
  int64_t w_off[W_LD];
  int w_lds[W_LD];
#pragma unroll
  for (int i = 0; i < W_LD; ++i) {
    int q = lane + 64 * i;
    q = q < 32 * W_PCS ? q : 32 * W_PCS - 1;       // PL = 3: the last instruction is half empty, lanes repeat the last piece
    const int r = q / W_PCS, pc = q % W_PCS;
    w_off[i] = (int64_t)(n0 + r) * nk * (PL * 64) + pc * 16;
    w_lds[i] = r * RS + pc * 16;
  }
  // fragment reads: lane = (row lane & 31, K half lane >> 5)
  const int frow = lane & 31, half = lane >> 5;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // the epilogue's bias and residual values are fetched NOW: a cold load behind the reduction sits on the critical path of a
  // kernel that lives for ~5 us (measured inside the persistent-network experiment, DESIGN.md 12: ~1 us per GEMM)
  const int e_row = tid >> 3, e_c4 = (tid & 7) * 4;
  const bool e_on = tid < 256 && m0 + e_row < g.M;
  f32x4 e_bias = f32x4{0.f, 0.f, 0.f, 0.f}, e_res = f32x4{0.f, 0.f, 0.f, 0.f};
  if (e_on && g.bias) e_bias = *reinterpret_cast<const f32x4*>(g.bias + grp * g.gBias + n0 + e_c4);
  if (e_on && g.R) e_res = *reinterpret_cast<const f32x4*>(g.R + (int64_t)(m0 + e_row) * g.ldr + n0 + e_c4);

  constexpr int GRP = 4;                           // K tiles in flight per wave (registers)
  for (int kt = kt0; kt < kt1; kt += GRP) {
    f32x4 ra[GRP][4], rw[GRP][W_LD];
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      if (kt + u < kt1) {                          // wave-uniform
        const int k0 = (kt + u) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          ra[u][i] = k0 < K1 ? *reinterpret_cast<const f32x4*>(A + a_off[i] + k0)
                             : *reinterpret_cast<const f32x4*>(A2 + a2_off[i] + (k0 - K1));
#pragma unroll
        for (int i = 0; i < W_LD; ++i)
          rw[u][i] = *reinterpret_cast<const f32x4*>(Wsp + w_off[i] + (int64_t)(kt + u) * (PL * 64));
      }
    }
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      if (kt + u < kt1) {
        // wave-private staging (no block barrier: the LDS executes one wave's instructions in order); two buffers so the
        // stores of tile u+1 do not wait behind the fragment reads of tile u
        unsigned char* As = stage[wave][u % NBUF];
        unsigned char* Ws = As + A_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(As + ((a_r + 8 * i) * AS + a_c) * 4) = ra[u][i];
#pragma unroll
        for (int i = 0; i < W_LD; ++i) *reinterpret_cast<f32x4*>(Ws + w_lds[i]) = rw[u][i];
        __builtin_amdgcn_wave_barrier();           // compiler-only: the fragment reads below see other lanes' stores
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(As + (frow * AS + s * 16 + half * 8) * 4);
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(As + (frow * AS + s * 16 + half * 8 + 4) * 4);
          unsigned a[PL], b[PL], c[PL], d[PL];
          split_pair<PL, FMT>(x0[0], x0[1], a);
          split_pair<PL, FMT>(x0[2], x0[3], b);
          split_pair<PL, FMT>(x1[0], x1[1], c);
          split_pair<PL, FMT>(x1[2], x1[3], d);
          bf16x8 af[PL], bf[PL];
#pragma unroll
          for (int p = 0; p < PL; ++p) {
            union { bf16x8 v; unsigned w[4]; } x;
            x.w[0] = a[p]; x.w[1] = b[p]; x.w[2] = c[p]; x.w[3] = d[p];
            af[p] = x.v;
            bf[p] = *reinterpret_cast<const bf16x8*>(Ws + frow * RS + p * 64 + s * 32 + half * 16);
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
        // compiler-only: with one staging buffer the stores of tile u+1 reuse this region and must stay behind the
        // fragment reads of tile u (other lanes' data)
        __builtin_amdgcn_wave_barrier();
      }
    }
  }

  // partial sums -> LDS (row stride 33: the 16 ds_write_b32 of a wave hit distinct banks)
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + frow] = acc[r];
  __syncthreads();
  if (e_on) {
    float* Y = g.Y + grp * g.gY;
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int o = e_row * 33 + e_c4 + c;
      float acc4 = red[0][o];
#pragma unroll
      for (int w = 1; w < NWV; ++w) acc4 += red[w][o];            // fixed order
      v[c] = acc4 + e_bias[c];
      if (g.act == ACT_RELU) v[c] = fmaxf(v[c], 0.f);
      else if (g.act == ACT_GELU) v[c] = 0.5f * v[c] * (1.f + erff(v[c] * 0.70710678118654752440f));
      else if (g.act == ACT_DIST) v[c] = fmaxf(2.f - 2.f * v[c], 0.f);
      v[c] += e_res[c];
    }
    *reinterpret_cast<f32x4*>(Y + (int64_t)(m0 + e_row) * g.ldy + n0 + e_c4) = v;
  }
}

// dispatched when even 64 x 64 tiles would leave most CUs without a block
inline bool small_gemm_wins(const GemmArgs& g, int groups) {
  static const bool off = LT_XENV("LINETR_NO_SMALL_GEMM") != nullptr;   // tuning aid
  if (off || g.N % 32 != 0 || g.K % 32 != 0 || g.K < 128) return false;
  if (g.lda % 4 != 0 || g.ldy % 4 != 0 || (g.R && g.ldr % 4 != 0) || (g.A2 && (g.lda2 % 4 != 0 || g.K1 % 32 != 0))) return false;
  return (int64_t)cdiv(g.M, 64) * (g.N / 64) * groups < 256;
}

template <int PL, int FMT>
inline void gemm_split_small_launch(const SplitGemmArgs& sa, int groups, hipStream_t st) {
  dim3 grid((unsigned)((sa.g.N / 32) * cdiv(sa.g.M, 32)), (unsigned)groups);
  // K >= 256: eight waves split K (half the loads, splits and MFMAs on every wave's critical path)
  static const bool w4 = LT_XENV("LINETR_SMALL_GEMM_4WAVE") != nullptr;   // tuning aid
  static const bool no2 = LT_XENV("LINETR_SMALL_GEMM_NO_2PERCU") != nullptr;   // tuning aid
  // the 8-wave block claims 121 KB of LDS = one block per CU: a grid of 257..512 blocks (q/k/v of a single pair: 312) would
  // run two rounds; four waves with one staging buffer (62 KB) put two blocks on a CU and finish it in one
  const int64_t blocks = (int64_t)grid.x * grid.y;
  if (!no2 && !w4 && sa.g.K >= 256 && blocks > 256 && blocks <= 512)
    hipLaunchKernelGGL((gemm_split_small_kernel<PL, FMT, 4, 1>), grid, dim3(256), 0, st, sa);
  else if (!w4 && sa.g.K >= 256) hipLaunchKernelGGL((gemm_split_small_kernel<PL, FMT, 8>), grid, dim3(512), 0, st, sa);
  else hipLaunchKernelGGL((gemm_split_small_kernel<PL, FMT, 4>), grid, dim3(256), 0, st, sa);
}

}  // namespace lt
