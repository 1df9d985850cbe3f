// fp32 MFMA GEMM for gfx950:  Y[M,N] = epi(A[M,K] * W[N,K]^T + bias)   (W in PyTorch [out,in] layout)
//
// Every dense contraction of the LineTR forward goes through this kernel (the reference runs them as
// nn.Linear / Conv1d(k=1): models/line_transformer.py:9-20,:139-166, models/line_attention.py:55-94).
//
// CDNA4 mapping
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = the chip's 157 TF fp32 peak).
//   * 256 threads = 4 wave64; block tile BM x BN, BK = 32; each wave owns (BM/WM) x (BN/WN).
//   * A and W tiles are staged global -> VGPR (dwordx4) -> LDS as row-major [rows][32+4]; the +4 pad
//     makes the row stride 36 dwords = 4*odd, so the 16-lane groups of ds_read_b128 hit 64 distinct
//     banks (conflict-free, MI355X_MICROARCH LDS table).
//   * K-permutation trick: one ds_read_b128 gives a lane 4 consecutive k of its row.  MFMA step s of
//     chunk kk consumes element s from both halves of the wave, i.e. k = 8kk + 4*(lane>>5) + s.  A and
//     W use the same mapping, so the sum over k is complete and no repacking is needed.
//   * double-buffered LDS; global loads for tile t+1 are issued before the MFMAs of tile t.
#pragma once
#include "lt_common.h"

namespace lt {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_DIST = 3 /* max(2-2x,0): get_dist_matrix */ };

struct GemmArgs {
  const float* A;   int lda;             // logical A columns [0,K1)
  const float* A2;  int lda2;  int K1;   // optional second source for columns [K1,K) (concat); A2==nullptr -> K1=K
  const float* W;   int ldw;             // [N][K]
  const float* bias;                     // [N] or nullptr
  const float* R;   int ldr;             // residual [M,N] or nullptr
  float* Y;         int ldy;
  int M, N, K;
  int act;
  // grouped launch (blockIdx.z = g): element strides added per group
  int64_t gA, gW, gBias, gY;
  // optional row normalisation fused into the epilogue of tiles that own complete 256-wide rows (split-bf16 128x256
  // tile only; run_gemm_norm falls back to row_norm_kernel otherwise).  Applied AFTER bias / activation / residual:
  //   norm 1: LayerNorm(x) * gamma + beta (eps inside the sqrt), norm 2: x / max(||x||_2, 1e-12); then + add2.
  int norm = 0;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  const float* add2 = nullptr;
  int ldadd2 = 0;
  float eps = 0.f;
};

constexpr int GEMM_BK = 32;
constexpr int GEMM_LDS_STRIDE = GEMM_BK + 4;

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  static_assert(WM * WN == 4 && MI >= 1 && NI >= 1, "bad tiling");
  constexpr int LS = GEMM_LDS_STRIDE;
  constexpr int A_F4 = BM * 8 / 256;  // float4 per thread per A tile
  constexpr int B_F4 = BN * 8 / 256;
  static_assert(A_F4 >= 1 && B_F4 >= 1, "tile too small");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][BM][LS]
  float* Bs = smem + 2 * BM * LS;      // [2][BN][LS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int grp = blockIdx.z;
  const float* A = g.A + grp * g.gA;
  const float* A2 = g.A2 ? g.A2 + grp * g.gA : nullptr;
  const float* W = g.W + grp * g.gW;
  const int K1 = g.A2 ? g.K1 : g.K;

  const int lrow = tid >> 3, lc4 = (tid & 7) * 4;  // 32 rows x 8 float4 per pass

  f32x4 ra[A_F4], rb[B_F4];
  auto gload = [&](int k0) {
    const float* src = A; int ld = g.lda; int kk = k0;
    if (k0 >= K1) { src = A2; ld = g.lda2; kk = k0 - K1; }
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      int r = m0 + lrow + i * 32;
      r = r < g.M ? r : g.M - 1;
      ra[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)r * ld + kk + lc4);
    }
#pragma unroll
    for (int i = 0; i < B_F4; ++i) {
      int r = n0 + lrow + i * 32;
      rb[i] = *reinterpret_cast<const f32x4*>(W + (int64_t)r * g.ldw + k0 + lc4);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_F4; ++i)
      *reinterpret_cast<f32x4*>(&As[(buf * BM + lrow + i * 32) * LS + lc4]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_F4; ++i)
      *reinterpret_cast<f32x4*>(&Bs[(buf * BN + lrow + i * 32) * LS + lc4]) = rb[i];
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / GEMM_BK;
  gload(0);
  lstore(0);
  __syncthreads();
  const int frow = lane & 31, fk = (lane >> 5) * 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * GEMM_BK);
    const float* Ab = &As[(buf * BM + wm * TM + frow) * LS + fk];
    const float* Bb = &Bs[(buf * BN + wn * TN + frow) * LS + fk];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LS + kk * 8);
#pragma unroll
      for (int j = 0; j < NI; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LS + kk * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const float* bias = g.bias ? g.bias + grp * g.gBias : nullptr;
  float* Y = g.Y + grp * g.gY;
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

template <int BM, int BN, int WM, int WN>
inline int gemm_launch_t(const GemmArgs& g, int groups, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * (BM + BN) * GEMM_LDS_STRIDE * sizeof(float);
  static unsigned long long attr_done = 0;   // one bit per device: the opt-in is a per-device function attribute
  const unsigned long long dev_bit = current_device_bit();
  if (!(attr_done & dev_bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<BM, BN, WM, WN>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done |= dev_bit;
  }
  dim3 grid(g.N / BN, cdiv(g.M, BM), groups);
  hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN>), grid, dim3(256), lds, st, g);
  return 0;
}

// Picks a tile for the problem shape.  N must be a multiple of 64, K of 32.
inline int gemm_launch(const GemmArgs& g, int groups, hipStream_t st) {
  if (g.M <= 0) return 0;
  if (g.N % 64 != 0 || g.K % GEMM_BK != 0 || (g.A2 && g.K1 % GEMM_BK != 0))
    return fail(LINETR_E_ARG, "gemm: unsupported shape M=%d N=%d K=%d", g.M, g.N, g.K);
  if (g.N % 128 != 0) {
    gemm_launch_t<128, 64, 4, 1>(g, groups, st);
  } else {
    // enough 128x128 tiles to fill 256 CUs twice? otherwise use 64-row tiles for more blocks
    int64_t big_tiles = (int64_t)cdiv(g.M, 128) * (g.N / 128) * groups;
    if (big_tiles >= 384) gemm_launch_t<128, 128, 2, 2>(g, groups, st);
    else gemm_launch_t<64, 128, 2, 2>(g, groups, st);
  }
  LT_LAUNCH_CHECK();
  return 0;
}

}  // namespace lt
