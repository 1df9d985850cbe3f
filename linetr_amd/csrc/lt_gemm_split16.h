// Split-plane MFMA GEMM on 16-row tiles:  Y[M,N] = epi(A[M,K] * W[N,K]^T + bias), same arithmetic and the same
// operands as lt_gemm_split.h (fp32 activations split in the loader, pre-split weights, cross products smallest first,
// fp32 accumulation) -- but the block tile is 112 x 256.
//
// Why 112 rows: the descriptor network's GEMMs have M = 25 472 rows (cfg3) = 199 row tiles of 128, so every column slab
// fills 199 of the 256 CUs per round (78 %).  25 472 / 256 = 99.5 rows per CU; the smallest tile height above that in
// MFMA granularity is 112 = 7 x 16 (228 row tiles <= 256, each 0.875 of a 128-row tile).  That needs
// v_mfma_f32_16x16x32_bf16 and a wave layout without an M split: each of the 8 waves owns ALL 112 rows x 32 columns
// (7 x 2 accumulator tiles of 16 x 16), keeps its W fragments for the whole K chunk and streams the A fragments.
//
// Pipeline (one barrier per 32-wide K chunk, fixed issue order like the 128-row kernel): a chunk is 7 row tiles x
// (2 column tiles x NTERM products) MFMA slots.  Slots of row tiles 0-4 carry the A-fragment reads of row tiles 2-6 and
// the split + LDS stores of tile kt+1; then the barrier; the slots of row tiles 5-6 carry the fragment reads of the NEXT
// chunk (W fragments, A row tiles 0-1).  Every staging register is refilled with tile kt+2 one slot after its store.
#pragma once
#include "lt_gemm_split.h"

namespace lt {

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int FMT>
__device__ __forceinline__ f32x4v mfma16_split(const bf16x8& a, const bf16x8& b, const f32x4v& c) {
  if constexpr (FMT == 0) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

template <int PL, int FMT = 0>
__global__ __launch_bounds__(512) void gemm_split16_kernel(SplitGemmArgs sa) {
  const GemmArgs& g = sa.g;
  constexpr int BM = 112, BN = 256, NT = 512;
  constexpr int MI = 7, NI = 2;                        // 16-row / 16-column accumulator tiles per wave
  constexpr int RS = PL * 64 + 16;                     // LDS row stride in bytes (as in lt_gemm_split.h)
  constexpr int A_F4 = 2;                              // float4 per thread per A tile; the 2nd pass covers rows 64..111
  constexpr int B_PCS = BN * PL * 4 / NT;              // 16-byte pieces per thread per W tile (6 / 4)
  constexpr int NTERM = PL * (PL + 1) / 2;
  constexpr int SPR = NI * NTERM;                      // MFMA slots per row tile
  constexpr int N_SLOT = MI * SPR;
  constexpr int N_ST = A_F4 + B_PCS;                   // store / refill units per tile

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_x[];
  unsigned char* As = smem_x;                          // [2][BM][RS]
  unsigned char* Bs = smem_x + 2 * BM * RS;            // [2][BN][RS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gx = g.N / BN, gy = (g.M + BM - 1) / BM;
  const int ntile = gx * gy;
  int tile;
  {  // XCD-aware order, as in gemm_split_kernel
    const int b = blockIdx.x, q = ntile / 8, r = ntile % 8, xcd = b % 8, k = b / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m0 = (tile / gx) * BM, n0 = (tile % gx) * BN;
  const float* A = g.A;
  const float* A2 = g.A2;
  const unsigned char* Wsp = sa.Wsp;
  const int K1 = g.A2 ? g.K1 : g.K;
  const int nk = g.K / 32;

  const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
  f32x4 ra[A_F4], rb[B_PCS];
  auto unit_gload = [&](int u, int kt) {
    if (u < A_F4) {
      const int row_l = lrow + u * 64;
      if (u == 0 || row_l < BM) {                      // pass 1: threads 0..383 only
        const int k0 = kt * 32;
        const float* src = A; int ld = g.lda; int kk = k0;
        if (k0 >= K1) { src = A2; ld = g.lda2; kk = k0 - K1; }
        int r = m0 + row_l;
        r = r < g.M ? r : g.M - 1;
        ra[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)r * ld + kk + lc4);
      }
    } else {
      const int pq = tid + (u - A_F4) * NT;
      const int r = pq / (PL * 4), pc = pq % (PL * 4);
      rb[u - A_F4] = *reinterpret_cast<const f32x4*>(Wsp + ((int64_t)(n0 + r) * nk + kt) * (PL * 64) + pc * 16);
    }
  };
  auto unit_store = [&](int u, int buf) {
    if (u < A_F4) {
      const int row_l = lrow + u * 64;
      if (u == 0 || row_l < BM) {
        unsigned a[PL], b[PL];
        split_pair<PL, FMT>(ra[u][0], ra[u][1], a);
        split_pair<PL, FMT>(ra[u][2], ra[u][3], b);
        unsigned char* dst = As + (buf * BM + row_l) * RS + lc4 * 2;
#pragma unroll
        for (int p = 0; p < PL; ++p) *reinterpret_cast<u32x2*>(dst + p * 64) = u32x2{a[p], b[p]};
      }
    } else {
      const int pq = tid + (u - A_F4) * NT;
      const int r = pq / (PL * 4), pc = pq % (PL * 4);
      *reinterpret_cast<f32x4*>(Bs + (buf * BN + r) * RS + pc * 16) = rb[u - A_F4];
    }
  };

  f32x4v acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // fragment addressing: lane -> (row | column) lane & 15, K group lane >> 4 (8 consecutive k = 16 bytes of a plane)
  const int frow = lane & 15, fk = (lane >> 4) * 16;
  auto read_a = [&](int buf, int i, int p) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(As + (buf * BM + 16 * i + frow) * RS + p * 64 + fk);
  };
  auto read_b = [&](int buf, int j, int p) -> bf16x8 {
    return *reinterpret_cast<const bf16x8*>(Bs + (buf * BN + wave * 32 + 16 * j + frow) * RS + p * 64 + fk);
  };
  // cross terms, smallest first: (pa, pb) with pa + pb descending
  constexpr int TPA[6] = {PL == 3 ? 2 : 1, PL == 3 ? 1 : 0, 0, 1, 0, 0};
  constexpr int TPB[6] = {0, 1, PL == 3 ? 2 : 0, 0, 1, 0};

  // prologue: tile 0 -> LDS, tile 1 -> registers, first fragments
#pragma unroll
  for (int u = 0; u < N_ST; ++u) unit_gload(u, 0);
#pragma unroll
  for (int u = 0; u < N_ST; ++u) unit_store(u, 0);
#pragma unroll
  for (int u = 0; u < N_ST; ++u) unit_gload(u, nk > 1 ? 1 : 0);
  __syncthreads();
  bf16x8 bcur[NI][PL], a0[PL], a1[PL];                 // loop-carried: fragments of the chunk about to be computed
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int p = 0; p < PL; ++p) bcur[j][p] = read_b(0, j, p);
#pragma unroll
  for (int p = 0; p < PL; ++p) { a0[p] = read_a(0, 0, p); a1[p] = read_a(0, 1, p); }

  constexpr int BAR_SLOT = 5 * SPR;                    // the barrier sits in front of this slot
  constexpr int TAIL = N_SLOT - BAR_SLOT;              // slots after the barrier (row tiles 5, 6)
  constexpr int N_PRE = (NI + 2) * PL;                 // next-chunk fragment reads after the barrier
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const int ktn = kt + 2 < nk ? kt + 2 : nk - 1;     // branch-free tail: reloads the last tile (never consumed)
    bf16x8 af[MI][PL], bnxt[NI][PL], a0n[PL], a1n[PL];
#pragma unroll
    for (int p = 0; p < PL; ++p) { af[0][p] = a0[p]; af[1][p] = a1[p]; }
#pragma clang loop unroll(full)
    for (int m = 0; m < N_SLOT; ++m) {
      if (m == BAR_SLOT) {
        __syncthreads();               // tile kt+1 visible; every A fragment of this chunk has been read
        __builtin_amdgcn_sched_barrier(0);
      }
      // row tiles are walked in pairs (0,1) (2,3) (4,5) 6 with the two tiles alternating, so consecutive MFMAs on one
      // accumulator are 4 slots apart instead of 2
      int i, t, j;
      if (m < 6 * SPR) { const int w = m % (2 * SPR), rest = w / 2; i = 2 * (m / (2 * SPR)) + (w & 1); t = rest / NI; j = rest % NI; }
      else { const int w = m - 6 * SPR; i = 6; t = w / NI; j = w % NI; }
      acc[i][j] = mfma16_split<FMT>(af[i][TPA[t]], bcur[j][TPB[t]], acc[i][j]);
      if (m < BAR_SLOT) {
        // A fragments of row tile r (2..6): plane p at slot SPR (r - 2) + p * (SPR / PL)
#pragma unroll
        for (int r = 2; r < MI; ++r)
#pragma unroll
          for (int p = 0; p < PL; ++p)
            if (SPR * (r - 2) + p * (SPR / PL) == m) af[r][p] = read_a(buf, r, p);
        // split + LDS store of tile kt+1 spread over the slots in front of the barrier; each staging register is
        // refilled (tile kt+2) one slot after it has been stored, i.e. a whole chunk before it is needed again
#pragma unroll
        for (int u = 0; u < N_ST; ++u) {
          if ((2 * u + 1) * BAR_SLOT / (2 * N_ST) == m) unit_store(u, buf ^ 1);
          if ((2 * u + 1) * BAR_SLOT / (2 * N_ST) + 1 == m) unit_gload(u, ktn);
        }
      } else {
        const int s = m - BAR_SLOT;
        // next chunk: W fragments first, then A row tiles 0 and 1, evenly over the tail
#pragma unroll
        for (int q = 0; q < N_PRE; ++q)
          if (q * TAIL / N_PRE == s) {
            if (q < NI * PL) bnxt[q / PL][q % PL] = read_b(buf ^ 1, q / PL, q % PL);
            else if (q < (NI + 1) * PL) a0n[q - NI * PL] = read_a(buf ^ 1, 0, q - NI * PL);
            else a1n[q - (NI + 1) * PL] = read_a(buf ^ 1, 1, q - (NI + 1) * PL);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int p = 0; p < PL; ++p) {
      a0[p] = a0n[p]; a1[p] = a1n[p];
#pragma unroll
      for (int j = 0; j < NI; ++j) bcur[j][p] = bnxt[j][p];
    }
  }

  // epilogue: C tile (i, j): column lane & 15, rows 4 (lane >> 4) + r
  const float* bias = g.bias;
  float* Y = g.Y;
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int col = n0 + wave * 32 + 16 * j + (lane & 15);
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * i + 4 * (lane >> 4) + r;
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

template <int PL, int FMT = 0>
inline void gemm_split16_launch(const SplitGemmArgs& sa, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * (112 + 256) * (PL * 64 + 16);
  static unsigned long long attr_done = 0;   // one bit per device: the opt-in is a per-device function attribute
  const unsigned long long dev_bit = current_device_bit();
  if (!(attr_done & dev_bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split16_kernel<PL, FMT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done |= dev_bit;
  }
  dim3 grid((sa.g.N / 256) * cdiv(sa.g.M, 112));
  hipLaunchKernelGGL((gemm_split16_kernel<PL, FMT>), grid, dim3(512), lds, st, sa);
}

// 112-row tiles pay when they save a round of blocks: rounds x tile cost against the 128-row kernel.  Measured on the
// pipeline's shapes (M = 25 472): +6-8 % in the 3-product modes; +-0 in bf16x6, where the chip sits at its power cap and
// the CUs a shorter round would have left idle were lending their power budget to the busy ones anyway (and this
// kernel reads 29 % more LDS bytes per MFMA: 198 vs 214 TF on a GEMM without any quantisation effect).  So the
// dispatcher (split_tile_name) only picks it for PL == 2.
inline bool split16_wins(const GemmArgs& g, int groups) {
  if (groups != 1 || g.N % 256 != 0) return false;
  const int64_t t128 = (int64_t)cdiv(g.M, 128) * (g.N / 256), t112 = (int64_t)cdiv(g.M, 112) * (g.N / 256);
  if (t128 < 192) return false;
  const double c128 = (double)cdiv((int)t128, 256) * 1.0, c112 = (double)cdiv((int)t112, 256) * 0.875;
  return c112 < 0.97 * c128;
}

}  // namespace lt
