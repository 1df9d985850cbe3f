// Split-plane MFMA GEMM with direct-to-LDS tile loads:  Y[M,N] = epi(A[M,K] * W[N,K]^T + bias).
//
// Same arithmetic as lt_gemm_split.h (bf16x6 / bf16x3 / f16x3 cross products, smallest terms first, fp32
// accumulation), different data path: BOTH operands arrive already split ([rows][K/32][PL][32] 16-bit planes) and
// the K tiles are copied global -> LDS by `global_load_lds_dwordx4` (the data never passes through VGPRs, no
// per-tile split VALU, no ds_write).  The LDS image keeps the padded row stride of lt_gemm_split.h (PL*64 + 16 B,
// conflict-free ds_read_b128): a DMA instruction writes 64 consecutive 16-byte pieces (lane order), so the lane
// that lands on a row's pad piece simply re-reads the row's last real piece.
#pragma once
#include "lt_gemm_split.h"

namespace lt {

struct DmaGemmArgs {
  GemmArgs g;                 // g.A / g.A2 / g.W unused
  const unsigned char* Asp;   // split activations, row stride (K1/32)*PL*64 bytes
  const unsigned char* Asp2;  // second K range (concat), row stride ((K-K1)/32)*PL*64 bytes, or null
  const unsigned char* Wsp;   // split weights
};

template <int BM, int BN, int WM, int WN, int PL, int FMT = 0>
__global__ __launch_bounds__(WM * WN * 64) void gemm_dma_kernel(DmaGemmArgs da) {
  const GemmArgs& g = da.g;
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int RS = PL * 64 + 16;                 // LDS row stride in bytes
  constexpr int PR = RS / 16;                      // 16-byte pieces per LDS row (last one is the pad)
  constexpr int BUF = (BM + BN) * RS;              // bytes per stage
  constexpr int NINS = (BM + BN) * PR / 64;        // DMA instructions per stage
  constexpr int A_INS = BM * PR / 64;              // ... of which the first A_INS carry A rows
  constexpr int SLOTS = (NINS + NW - 1) / NW;
  static_assert((BM * PR) % 64 == 0 && (BN * PR) % 64 == 0, "tile rows must fill whole DMA instructions");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_d[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int gx = g.N / BN, gy = (g.M + BM - 1) / BM;
  const int ntile = gx * gy;
  int tile;
  {
    const int b = blockIdx.x, q = ntile / 8, r = ntile % 8, xcd = b % 8, k = b / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m0 = (tile / gx) * BM, n0 = (tile % gx) * BN;
  const int nk = g.K / 32;
  const int nk1 = da.Asp2 ? g.K1 / 32 : nk;

  // per-slot source row / piece (constant over the K loop)
  int s_row[SLOTS], s_pc[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int d = wave + s * NW;
    const int q = d * 64 + lane;
    int row = q / PR, pc = q % PR;
    pc = pc < PR - 1 ? pc : PR - 2;
    if (d < A_INS) { row = m0 + row; row = row < g.M ? row : g.M - 1; }
    else row = n0 + (row - BM);
    s_row[s] = row; s_pc[s] = pc * 16;
  }
  auto dma = [&](int kt, int buf) {
    const unsigned char* abase = da.Asp; int anks = nk1, akt = kt;
    if (kt >= nk1) { abase = da.Asp2; anks = nk - nk1; akt = kt - nk1; }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int d = wave + s * NW;
      if (d < NINS) {          // wave-uniform
        const unsigned char* src = d < A_INS
            ? abase + ((int64_t)s_row[s] * anks + akt) * (PL * 64) + s_pc[s]
            : da.Wsp + ((int64_t)s_row[s] * nk + kt) * (PL * 64) + s_pc[s];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(smem_d + buf * BUF + d * 1024), 16, 0, 0);
      }
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fk = (lane >> 5) * 16;
  auto compute = [&](int buf) {
    const unsigned char* Ab = smem_d + buf * BUF + (wm * TM + frow) * RS + fk;
    const unsigned char* Bb = smem_d + buf * BUF + (BM + wn * TN + frow) * RS + fk;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[MI][PL], bf[NI][PL];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int p = 0; p < PL; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * RS + p * 64 + s * 32);
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int p = 0; p < PL; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * RS + p * 64 + s * 32);
#pragma unroll
      for (int ord = 2 * (PL - 1); ord >= 0; --ord) {
        if (ord > PL - 1) continue;
#pragma unroll
        for (int pa = PL - 1; pa >= 0; --pa) {
          const int pb = ord - pa;
          if (pb < 0 || pb >= PL) continue;
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = mfma_split<FMT>(af[i][pa], bf[j][pb], acc[i][j]);
        }
      }
    }
  };

  dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) dma(kt + 1, buf ^ 1);   // flies during this tile's MFMAs; buf^1 was released by the last barrier
    compute(buf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  const float* bias = g.bias;
  float* Y = g.Y;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int col = n0 + wn * TN + j * 32 + (lane & 31);
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < g.M) {
          float v = acc[i][j][r] + bv;
          if (g.act == ACT_RELU) v = fmaxf(v, 0.f);
          else if (g.act == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
          else if (g.act == ACT_DIST) v = fmaxf(2.f - 2.f * v, 0.f);
          if (g.R) v += g.R[(int64_t)row * g.ldr + col];
          Y[(int64_t)row * g.ldy + col] = v;
        }
      }
    }
  }
}

template <int BM, int BN, int WM, int WN, int PL, int FMT = 0>
inline void gemm_dma_launch_t(const DmaGemmArgs& da, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * (BM + BN) * (PL * 64 + 16);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma_kernel<BM, BN, WM, WN, PL, FMT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  dim3 grid((da.g.N / BN) * cdiv(da.g.M, BM));
  hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, WM, WN, PL, FMT>), grid, dim3(WM * WN * 64), lds, st, da);
}

template <int PL, int FMT = 0>
inline int gemm_dma_launch(const DmaGemmArgs& da, hipStream_t st) {
  const GemmArgs& g = da.g;
  if (g.M <= 0) return 0;
  if (g.N % 128 != 0 || g.K % 32 != 0) return fail(LINETR_E_ARG, "gemm_dma: unsupported shape M=%d N=%d K=%d", g.M, g.N, g.K);
  static const char* tile_env = getenv("LINETR_GEMM_TILE");
  const char* tile = tile_env ? tile_env : split_tile_name(g, 1);
  if (!strcmp(tile, "256x128")) gemm_dma_launch_t<256, 128, 4, 2, PL, FMT>(da, st);
  else gemm_dma_launch_t<128, 128, 2, 2, PL, FMT>(da, st);
  LT_LAUNCH_CHECK();
  return 0;
}

}  // namespace lt
