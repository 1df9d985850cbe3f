// Fused MLP of a line-signature layer (models/line_transformer.py:157-183 with the merge conv folded into W1):
//     z' = z + W2 relu(W1 [z ; msg] + b1) + b2                     rows x (512 -> 512 -> 256), fp32 in / fp32 out
// in ONE kernel, hidden activations never leaving the registers (as two launches W1 writes and W2 re-reads a
// [rows,512] fp32 tensor, 2 x 52 MB per layer at cfg3, and each launch pays ~19 us that do not depend on K).
//
// Same split-bf16 arithmetic as gemm_split_kernel (PL planes per operand, cross terms with pa + pb <= PL-1, smallest
// first, fp32 accumulation), on the TRANSPOSED products -- the mlp123 trick:
//     H^T [512 x rows] = W1 X^T ,   Y^T [256 x rows] = W2 H^T
// A = weights (rows of the MFMA), B = activations^T (a lane owns ONE batch row = MFMA column).  The C/D layout of H^T --
// lane = batch row, registers = 16 hidden units (r&3) + 8(r>>2) + 4(lane>>5) of a 32-unit tile -- is exactly a B operand
// of the second product once W2's K index is permuted inside every 16-group to [0 1 2 3 8 9 10 11 | 4 5 6 7 12 13 14 15]
// (done once at load time: `W2 perm` image), so bias + ReLU + the bf16 split happen in registers.
//
// Block = 4 waves = 128 batch rows (wave w: rows 32w..32w+31), ONE wave per SIMD (496 VGPRs).  The hidden layer is walked
// in 4 chunks of 128 units; per chunk 16 K tiles of W1 (128 units x 32 k x PL planes, 26 KB) feed the chunk's accumulators,
// then 4 K tiles x 2 output halves of W2 (same tile shape) fold the chunk into the 8 output accumulator tiles: 96 uniform
// tiles of 48 MFMAs per wave.  Weight tiles stream L2 -> registers -> LDS into 4 slots, committed two iterations before
// they are read, one barrier per tile; the A fragments are double-buffered in registers across tiles; X^T fragments come
// straight from global memory (a lane reads 8 consecutive floats of its own row), fetched two tiles and split one tile ahead.
//
// STATUS: opt-in experiment (LINETR_FUSED_SIG_MLP=1), parity-tested, SLOWER than the two tiled GEMMs at cfg3:
//   212 us per layer against 147 us (92 + 55).  Ablation on the device: MFMAs + fragment reads alone
//   run at 76 us (ideal 70 us on the 199 busy CUs), i.e. the register pipeline works; the weight / X loads add 115 us because
//   a tile's loads are issued and committed inside ONE iteration (no registers left for a second staging set), so every
//   iteration waits a full L2 round trip with a single wave per SIMD and nothing to switch to; prologue, 96 barriers, the
//   ReLU/split passes and the epilogue are another 67 us that nothing overlaps at one block per CU.
#pragma once
#include "lt_gemm_split.h"

namespace lt {

struct SigMlpArgs {
  const float* z;   int ldz;            // [M, 256] layer input (also the residual)
  const float* msg; int ldm;            // [M, 256] attention message (head-major)
  const unsigned char* W1sp;            // split planes [512][16][PL][32]   (K = [z ; msg])
  const float* b1;                      // [512]
  const unsigned char* W2sp;            // split planes [256][16][PL][32], K permuted inside 16-groups (see above)
  const float* b2;                      // [256]
  float* out;       int ldo;            // [M, 256]
  int M;
};

template <int PL, int FMT>
__global__ __launch_bounds__(256, 1) void sig_mlp_fused_kernel(SigMlpArgs a) {
  constexpr int RS = PL * 64 + 16;                 // LDS row stride of a weight tile row (bytes)
  constexpr int PCS = PL * 4;                      // 16-byte pieces per row and K tile
  constexpr int SLOT = 128 * RS;                   // every tile: 128 weight rows x 32 k x PL planes
  constexpr int NSLOT = 4;                         // tiles are committed two iterations before they are read
  constexpr int LD = 128 * PCS / 256;              // pieces per thread and tile
  constexpr int TPC = 24;                          // tiles per hidden chunk: 16 of W1, then 4 K tiles x 2 output halves of W2
  constexpr int NTILE = 4 * TPC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
  float* bias1 = reinterpret_cast<float*>(smem_f + NSLOT * SLOT);   // [512]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 128;
  int row = m0 + wave * 32 + n;
  const bool row_ok = row < a.M;
  row = row_ok ? row : a.M - 1;
  const float* zrow = a.z + (int64_t)row * a.ldz;
  const float* mrow = a.msg + (int64_t)row * a.ldm;
  bias1[tid] = a.b1[tid];
  bias1[tid + 256] = a.b1[tid + 256];

  // ---- weight tile stream.  tile id t = chunk * 24 + i;  i < 16: W1 rows 128 chunk.., K tile i;
  //      i >= 16: W2 rows 128 ((i - 16) & 1).., K tile 4 chunk + ((i - 16) >> 1)
  const int st_r = tid / PCS, st_pc = tid % PCS;   // piece u of this thread: row st_r + u * (256 / PCS) when 256 % PCS == 0
  auto tile_src = [&](int t) -> const unsigned char* {
    const int c = t / TPC, i = t % TPC;
    if (i < 16) return a.W1sp + ((int64_t)(c * 128) * 16 + i) * (PL * 64);
    return a.W2sp + ((int64_t)(128 * ((i - 16) & 1)) * 16 + 4 * c + ((i - 16) >> 1)) * (PL * 64);
  };
  f32x4 stg[LD];
  auto issue = [&](int t) {
    const unsigned char* src = tile_src(t);
#pragma unroll
    for (int u = 0; u < LD; ++u) {
      const int q = tid + 256 * u, r = q / PCS, pc = q % PCS;
      stg[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)r * 16 * (PL * 64) + pc * 16);
    }
  };
  auto commit = [&](int t) {
    unsigned char* dst = smem_f + (t % NSLOT) * SLOT;
#pragma unroll
    for (int u = 0; u < LD; ++u) {
      const int q = tid + 256 * u, r = q / PCS, pc = q % PCS;
      *reinterpret_cast<f32x4*>(dst + r * RS + pc * 16) = stg[u];
    }
  };
  (void)st_r; (void)st_pc;
  // A fragments (weights) of group g = (K step s = g >> 2, 32-row tile j = g & 3) of the tile in slot `slot`
  auto frag = [&](int t, int g, bf16x8 (&af)[PL]) {
    const unsigned char* wt = smem_f + (t % NSLOT) * SLOT + ((g & 3) * 32 + n) * RS + (g >> 2) * 32 + h * 16;
#pragma unroll
    for (int p = 0; p < PL; ++p) af[p] = *reinterpret_cast<const bf16x8*>(wt + p * 64);
  };
  // X^T fragments of first-product K tile kt: 2 K steps x 8 consecutive floats of this lane's row
  auto xload = [&](int kt, f32x4 (&x)[4]) {
    const float* src = (kt < 8 ? zrow : mrow) + (kt & 7) * 32 + h * 8;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      x[2 * s] = *reinterpret_cast<const f32x4*>(src + s * 16);
      x[2 * s + 1] = *reinterpret_cast<const f32x4*>(src + s * 16 + 4);
    }
  };
  auto xsplit = [&](const f32x4 (&x)[4], bf16x8 (&xb)[2][PL]) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      unsigned p0[PL], p1[PL], p2[PL], p3[PL];
      split_pair<PL, FMT>(x[2 * s][0], x[2 * s][1], p0);
      split_pair<PL, FMT>(x[2 * s][2], x[2 * s][3], p1);
      split_pair<PL, FMT>(x[2 * s + 1][0], x[2 * s + 1][1], p2);
      split_pair<PL, FMT>(x[2 * s + 1][2], x[2 * s + 1][3], p3);
#pragma unroll
      for (int p = 0; p < PL; ++p) {
        union { bf16x8 v; unsigned w[4]; } y;
        y.w[0] = p0[p]; y.w[1] = p1[p]; y.w[2] = p2[p]; y.w[3] = p3[p];
        xb[s][p] = y.v;
      }
    }
  };
  auto mma = [&](const bf16x8 (&af)[PL], const bf16x8 (&b)[PL], f32x16& acc) {
#pragma unroll
    for (int ord = PL - 1; ord >= 0; --ord)
#pragma unroll
      for (int pa = PL - 1; pa >= 0; --pa) {
        const int pb = ord - pa;
        if (pb < 0 || pb >= PL) continue;
        acc = mfma_split<FMT>(af[pa], b[pb], acc);
      }
  };

  f32x16 yacc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o)
#pragma unroll
    for (int r = 0; r < 16; ++r) yacc[o][r] = 0.f;

  // prologue: tiles 0 and 1 in LDS, X of K tile 0 split, X of K tile 1 in flight
  issue(0); commit(0);
  issue(1); commit(1);
  f32x4 xl[4];
  bf16x8 xb[2][PL];
  xload(0, xl);
  xsplit(xl, xb);
  xload(1, xl);
  __syncthreads();
  bf16x8 af0[PL];                                  // group 0 of the tile about to be consumed
  frag(0, 0, af0);

  int t = 0;
  for (int c = 0; c < 4; ++c) {
    f32x16 hacc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) hacc[j][r] = 0.f;
    // ---- first product: 16 K tiles of X^T against this chunk's 128 hidden units
#pragma unroll 2
    for (int kt = 0; kt < 16; ++kt, ++t) {
      issue(t + 2);                                // t + 2 <= 95 here
      bf16x8 xbn[2][PL];
      // the next K tile's X (loaded one iteration ago) is split under this tile's MFMAs; the one after is fetched now
      const bool nx = kt + 1 < 16 || c + 1 < 4;
      if (kt + 1 < 16 || c + 1 < 4) xsplit(xl, xbn);
      {
        const int k2 = (kt + 2) & 15;
        xload(k2, xl);
      }
      bf16x8 afa[PL], afb[PL];
#pragma unroll
      for (int p = 0; p < PL; ++p) afa[p] = af0[p];
#pragma unroll
      for (int g = 0; g < 8; g += 2) {
        frag(t, g + 1, afb);
        mma(afa, xb[g >> 2], hacc[g & 3]);
        if (g + 2 < 8) frag(t, g + 2, afa); else frag(t + 1, 0, afa);   // tile t + 1 has been visible since the last barrier
        mma(afb, xb[(g + 1) >> 2], hacc[(g + 1) & 3]);
      }
#pragma unroll
      for (int p = 0; p < PL; ++p) af0[p] = afa[p];
      if (nx) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int p = 0; p < PL; ++p) xb[s][p] = xbn[s][p];
      }
      commit(t + 2);
      __syncthreads();
    }
    // ---- bias + ReLU + split: the chunk's H^T as B-operand planes of 8 K steps (16 hidden units each)
    bf16x8 hb[8][PL];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int half = 0; half < 2; ++half) {       // registers 8 half .. 8 half + 7 of tile j = K step 2j + half
        unsigned w[4][PL];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r0 = 8 * half + 2 * e, r1 = r0 + 1;
          const int u0 = c * 128 + j * 32 + (r0 & 3) + 8 * (r0 >> 2) + 4 * h;
          const int u1 = c * 128 + j * 32 + (r1 & 3) + 8 * (r1 >> 2) + 4 * h;
          const float v0 = fmaxf(hacc[j][r0] + bias1[u0], 0.f);
          const float v1 = fmaxf(hacc[j][r1] + bias1[u1], 0.f);
          split_pair<PL, FMT>(v0, v1, w[e]);
        }
#pragma unroll
        for (int p = 0; p < PL; ++p) {
          union { bf16x8 v; unsigned u[4]; } x;
          x.u[0] = w[0][p]; x.u[1] = w[1][p]; x.u[2] = w[2][p]; x.u[3] = w[3][p];
          hb[2 * j + half][p] = x.v;
        }
      }
    // ---- second product: 4 K tiles (32 hidden units) x 2 output halves of the K-permuted W2 against the chunk
#pragma unroll
    for (int i = 0; i < 8; ++i, ++t) {
      const int q = i >> 1, oh = i & 1;
      const bool more2 = t + 2 < NTILE, more1 = t + 1 < NTILE;
      if (more2) issue(t + 2);
      bf16x8 afa[PL], afb[PL];
#pragma unroll
      for (int p = 0; p < PL; ++p) afa[p] = af0[p];
#pragma unroll
      for (int g = 0; g < 8; g += 2) {
        frag(t, g + 1, afb);
        mma(afa, hb[2 * q + (g >> 2)], yacc[4 * oh + (g & 3)]);
        if (g + 2 < 8) frag(t, g + 2, afa); else if (more1) frag(t + 1, 0, afa);
        mma(afb, hb[2 * q + ((g + 1) >> 2)], yacc[4 * oh + ((g + 1) & 3)]);
      }
#pragma unroll
      for (int p = 0; p < PL; ++p) af0[p] = afa[p];
      if (more2) commit(t + 2);
      __syncthreads();
    }
  }
  // ---- epilogue: + b2 + residual, store (a lane owns one batch row: 4 consecutive outputs per register quad)
  if (row_ok) {
    float* orow = a.out + (int64_t)row * a.ldo;
#pragma unroll
    for (int o = 0; o < 8; ++o)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int u = o * 32 + 8 * g4 + 4 * h;     // outputs u .. u+3 = registers 4 g4 .. 4 g4 + 3
        const f32x4 bb = *reinterpret_cast<const f32x4*>(a.b2 + u);
        const f32x4 zz = *reinterpret_cast<const f32x4*>(zrow + u);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = yacc[o][4 * g4 + e] + bb[e] + zz[e];
        *reinterpret_cast<f32x4*>(orow + u) = v;
      }
  }
}

// K permutation of the second weight matrix inside every group of 16 hidden units (host side, at load time)
inline int sig_mlp_kperm(int k) {
  static const int perm[16] = {0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15};
  return (k & ~15) | perm[k & 15];
}

template <int PL, int FMT>
inline void sig_mlp_fused_launch(const SigMlpArgs& a, hipStream_t st) {
  constexpr size_t lds = 4 * 128 * (PL * 64 + 16) + 512 * sizeof(float);
  static unsigned long long attr_done = 0;
  const unsigned long long dev_bit = current_device_bit();
  if (!(attr_done & dev_bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sig_mlp_fused_kernel<PL, FMT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done |= dev_bit;
  }
  hipLaunchKernelGGL((sig_mlp_fused_kernel<PL, FMT>), dim3((unsigned)cdiv(a.M, 128)), dim3(256), lds, st, a);
}

}  // namespace lt
