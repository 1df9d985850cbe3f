// Weight-stationary split-bf16 GEMM (bf16x6) for the one long-and-thin shape of the path:  Y[M,256] = act(A[M,128] W^T + bias),
// M = every real token of the batch (cfg3: 291 208 rows) -- layer 4 of the WordPositionalEncoder MLP (models/line_transformer.py:9-20,
// 52-73: Conv1d(128 -> 256, k = 1) + folded BatchNorm + ReLU).  With K = 128 a tiled GEMM is all prologue and epilogue (four K tiles
// per 64 KiB of output); here the WEIGHTS stay put and the token rows stream through:
//   * a block of 8 waves lives for the whole launch (one per CU); wave w keeps the three bf16 planes of output channels
//     32 w .. 32 w + 31 in 96 VGPRs, as the A operand of the transposed product D^T = W . X^T (8 K steps x 3 planes), read once
//     from the split-tile image of lt_st_image.h;
//   * token rows arrive 64 at a time: a thread loads two 32-byte runs of one row, splits them into planes ONCE (a tiled kernel
//     splits every activation element once per column tile) and stores them as 16-byte pieces of a split-tile image in LDS
//     (48 KiB, double-buffered), from which every wave reads its B fragments with conflict-free ds_read_b128 -- one read per two
//     MFMAs, two K steps ahead -- the loads of tile t+2 in flight under the 96 MFMAs of tile t, one barrier per tile;
//   * the accumulators start at the bias (kept in LDS); a half-wave swap per register pair leaves a lane with two runs of 8
//     consecutive channels of its token: ReLU and four dwordx4 stores, no LDS in the epilogue, and the epilogue of one 32-token
//     half runs in the MFMA slots of the other.  A wave writes complete 128-byte lines.
// cfg3 (291 208 rows, tools/ubench/ws_gemm_bench.hip): 110-114 us = 170 TF-eq, 4.0 TB/s of its 447 MB; the tiled kernel it replaces:
// 187 us.  Ablations: no MFMAs 96 us (the memory side alone: the kernel sits 15 % above it), no loads and no stores 92 us (MFMAs + LDS
// fragment reads: eight waves each read the whole tile, 50 % of the LDS read bandwidth), neither 31 us.  Plain stores: write-through
// (sc1) or nontemporal ones double the time (a wave writes 32-byte runs; the L2 has to merge them into lines).
// Since lt_tokmlp.h runs all four MLP layers of both positional encoders in one kernel (its layer-4 stage IS this kernel's slot loop),
// this GEMM is the stand-alone form: taken when the one-kernel MLP does not apply (non-reference channel widths) and by the unit tests.
#pragma once
#include "lt_st_image.h"

namespace lt {

struct WsGemmArgs {
  const float* A = nullptr; int lda = 0;         // [M][128 ..]
  const unsigned char* Wst = nullptr;            // split-tile image of W [256][128]
  const float* bias = nullptr;                   // [256], never null
  float* Y = nullptr; int ldy = 0;               // [M][256 ..]
  int M = 0, act = 0;                            // ACT_NONE / ACT_RELU
};

constexpr int WS_TM = 64, WS_NK = 8, WS_N = 256, WS_K = 128;
constexpr int WS_BUF = WS_NK * (WS_TM / 16) * ST_RB;      // 49 152: [K step][16-token block][plane][k half][token] x 16 B
constexpr int WS_LDS = 2 * WS_BUF + WS_N * 4;             // + the bias vector

__global__ __launch_bounds__(512) void gemm_ws_kernel(WsGemmArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char ws_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h2 = lane >> 5, lq = lane & 31;
  const int ntiles = (a.M + WS_TM - 1) / WS_TM;
  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  // fragment address inside a K step of a split-tile image: 16-row block (lane >> 4) & 1, k half lane >> 5, row lane & 15
  const int lfrag = ((lane >> 4) & 1) * ST_RB + (lane >> 5) * 256 + (lane & 15) * 16;

  // ---- loader: thread = (token tid >> 3 of the tile, 8-wide k pieces pc and pc + 8)
  const int tt = tid >> 3, pc = tid & 7;
  // Inside a 256-byte (k half) row of a chunk the 16 token slots are ROTATED by 2 (piece & 7): the 16 lanes of a store group are
  // 2 tokens x 8 pieces, which would otherwise all hit the same two 16-byte bank groups (8-way conflicts: measured 59 % of the
  // LDS-active cycles); a fragment read sees one piece and 16 tokens, a rotation of a conflict-free row.
  const int wr_off = ((pc >> 1) * (WS_TM / 16) + (tt >> 4)) * ST_RB + (pc & 1) * 256 + (((tt & 15) + 2 * pc) & 15) * 16;
  constexpr int WR_HALF = 4 * (WS_TM / 16) * ST_RB;        // piece pc + 8 sits four K steps further
  f32x4 raw[4];
  auto load_tile = [&](int t) {
    int r = t * WS_TM + tt;
    r = r < a.M ? r : a.M - 1;
    const float* g = a.A + (int64_t)r * a.lda + 8 * pc;
    raw[0] = *reinterpret_cast<const f32x4*>(g);
    raw[1] = *reinterpret_cast<const f32x4*>(g + 4);
    raw[2] = *reinterpret_cast<const f32x4*>(g + 64);
    raw[3] = *reinterpret_cast<const f32x4*>(g + 68);
  };
  auto write_piece = [&](const f32x4& x0, const f32x4& x1, unsigned char* dst) {
    unsigned p0[3], p1[3], p2[3], p3[3];
    split_pair<3>(x0[0], x0[1], p0); split_pair<3>(x0[2], x0[3], p1);
    split_pair<3>(x1[0], x1[1], p2); split_pair<3>(x1[2], x1[3], p3);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + p * ST_CHUNK) = u32x4{p0[p], p1[p], p2[p], p3[p]};
  };
  auto write_tile = [&](int buf) {
    write_piece(raw[0], raw[1], ws_smem + buf * WS_BUF + wr_off);
    write_piece(raw[2], raw[3], ws_smem + buf * WS_BUF + wr_off + WR_HALF);
  };
  load_tile(tile);

  // ---- this wave's weights (A operand: 32 channels x 16 k per fragment) and bias, resident for the whole launch
  bf16x8 wreg[WS_NK][3];
  {
    const unsigned char* wp = a.Wst + (int64_t)(2 * wave) * ST_RB + lfrag;
#pragma unroll
    for (int kt = 0; kt < WS_NK; ++kt)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        wreg[kt][p] = *reinterpret_cast<const bf16x8*>(wp + (int64_t)kt * (WS_N / 16) * ST_RB + p * ST_CHUNK);
  }
  // the bias sits behind the two tile buffers; register 4 b + c of an accumulator tile is channel 32 wave + 8 b + 4 h2 + c
  if (tid < WS_N / 4) *reinterpret_cast<f32x4*>(ws_smem + 2 * WS_BUF + tid * 16) = *reinterpret_cast<const f32x4*>(a.bias + tid * 4);
  const unsigned char* bias_lds = ws_smem + 2 * WS_BUF + (32 * wave + 4 * h2) * 4;
  write_tile(0);
  load_tile(tile + gridDim.x);
  __syncthreads();

  int zfrag[4];                                            // lfrag with the token slot rotated by 2 (2 kt + k half): four variants
#pragma unroll
  for (int c = 0; c < 4; ++c) zfrag[c] = ((lane >> 4) & 1) * ST_RB + (lane >> 5) * 256 + (((lane & 15) + 2 * (lane >> 5) + 4 * c) & 15) * 16;
  constexpr int TW[6] = {2, 1, 0, 1, 0, 0}, TA[6] = {0, 1, 2, 0, 1, 0};     // smallest cross terms first
  const int step = gridDim.x;
  // One tile = 96 MFMA slots in a fixed order (sched_barrier after each): the 48 of tokens 0..31 (acc0), then the 48 of tokens
  // 32..63 (acc1).  Everything else rides in the slots: the B fragments of the next K step (3 reads at the head of each K step),
  // the epilogue of acc0 under the MFMAs of acc1 and the epilogue of acc1 under the NEXT tile's acc0 MFMAs (no accumulator copies:
  // each set is idle for half a tile), the split + LDS stores of the next tile's rows (slots 30..41) and the global loads of the
  // tile after that (slot 44).  Both waves of a SIMD keep feeding the matrix pipe instead of meeting in an epilogue phase
  // between two barriers.
  // No block-uniform branches inside the slot loop (they would split it into basic blocks and turn every conditional touch of
  // an accumulator into copies): the steps run unconditionally -- acc1's epilogue before the first tile works on zeros and stores
  // nothing, the split of a tile past the end refills a buffer nobody reads, loads past the end clamp to the last row.
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
  int prev_tile = -1;
  auto epi_step = [&](f32x16& acc, int t, int j, int e) {  // e = 0..10: ReLU, 8 half-wave swap pairs, 2 x 32-byte-run stores
    if (e == 0) {
      if (a.act == ACT_RELU) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
      }
    } else if (e <= 8) {
      const int k = e - 1, lo = k < 4 ? k : 8 + (k - 4);
      float x = acc[lo], y = acc[lo + 4];
      halves_swap(x, y);
      acc[lo] = x; acc[lo + 4] = y;
    } else {
      // acc[0..7] = channels 32 wave + 8 h2 .. + 8, acc[8..15] = channels 32 wave + 16 + 8 h2 .. + 8 of token lq of half j
      const int part = e - 9, o = 8 * part;
      const int row = t * WS_TM + 32 * j + lq;
      bool ok = (unsigned)row < (unsigned)a.M;             // also false for the pass before the first tile (t = -1)
      if (ok) {
        float* yp = a.Y + (int64_t)row * a.ldy + 32 * wave + 8 * h2 + 16 * part;
        // (plain stores: write-through / nontemporal stores DOUBLE this kernel -- a wave writes 32-byte runs the L2 has to merge)
        *reinterpret_cast<f32x4*>(yp) = f32x4{acc[o], acc[o + 1], acc[o + 2], acc[o + 3]};
        *reinterpret_cast<f32x4*>(yp + 4) = f32x4{acc[o + 4], acc[o + 5], acc[o + 6], acc[o + 7]};
      }
    }
  };
  unsigned sp[4][3];                                       // split planes of a piece on their way to LDS
  auto split_step = [&](int e, int buf) {                  // e = 0..11: per piece two slots of pair splits, three LDS stores
    const int pcs = e / 6, k = e % 6;
    const f32x4& x0 = raw[2 * pcs];
    const f32x4& x1 = raw[2 * pcs + 1];
    if (k == 0) { split_pair<3>(x0[0], x0[1], sp[0]); split_pair<3>(x0[2], x0[3], sp[1]); }
    else if (k == 1) { split_pair<3>(x1[0], x1[1], sp[2]); split_pair<3>(x1[2], x1[3], sp[3]); }
    else if (k < 5) {
      const int p = k - 2;
      *reinterpret_cast<u32x4*>(ws_smem + buf * WS_BUF + wr_off + pcs * WR_HALF + p * ST_CHUNK) = u32x4{sp[0][p], sp[1][p], sp[2][p], sp[3][p]};
    }
  };
  auto init_acc = [&](f32x16& acc) {
#pragma unroll
    for (int bq = 0; bq < 4; ++bq) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(bias_lds + 32 * bq);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[4 * bq + c] = v[c];
    }
  };
#pragma unroll 1
  for (int it = 0; tile < ntiles; ++it) {
    const unsigned char* src = ws_smem + (it & 1) * WS_BUF;
    bf16x8 z[3][3];                                        // B fragments: the K step in flight and the next TWO (LDS latency under
                                                           // eight waves' reads is longer than one K step of MFMAs)
    auto read_z1 = [&](int g, int p) {                     // g = 8 j + kt
      z[g % 3][p] = *reinterpret_cast<const bf16x8*>(src + zfrag[g & 3] + ((g & 7) * (WS_TM / 16) + 2 * (g >> 3)) * ST_RB + p * ST_CHUNK);
    };
#pragma unroll
    for (int p = 0; p < 3; ++p) { read_z1(0, p); read_z1(1, p); }
    init_acc(acc0);
#pragma clang loop unroll(full)
    for (int m = 0; m < 96; ++m) {
      const int g = m / 6, t = m % 6, kt = g & 7;
      if (m == 48) init_acc(acc1);
      if (g < 8) acc0 = mfma_split<0>(wreg[kt][TW[t]], z[g % 3][TA[t]], acc0);
      else acc1 = mfma_split<0>(wreg[kt][TW[t]], z[g % 3][TA[t]], acc1);
      if (t < 3 && g + 2 < 16) read_z1(g + 2, t);   // into the set group g - 1 has just left
      if (m >= 2 && m < 13) epi_step(acc1, prev_tile, 1, m - 2);
      if (m >= 30 && m < 42) split_step(m - 30, (it & 1) ^ 1);
      if (m == 44) load_tile(tile + 2 * step);
      if (m >= 50 && m < 61) epi_step(acc0, tile, 0, m - 50);
      __builtin_amdgcn_sched_barrier(0);
    }
    prev_tile = tile;
    tile += step;
    __syncthreads();
  }
#pragma unroll
  for (int e = 0; e < 11; ++e) epi_step(acc1, prev_tile, 1, e);
}

inline bool gemm_ws_fits(int M, int N, int K, int lda, int ldy, int act) {
  return N == WS_N && K == WS_K && lda % 4 == 0 && ldy % 4 == 0 && (act == ACT_NONE || act == ACT_RELU) && M >= 16384;
}

inline int gemm_ws_launch(const WsGemmArgs& a, hipStream_t st) {
  if (a.M <= 0) return 0;
  if (!a.A || !a.Wst || !a.bias || !a.Y || a.lda % 4 || a.ldy % 4)
    return fail(LINETR_E_ARG, "gemm_ws: unsupported operands M=%d", a.M);
  static unsigned long long attr_done = 0;
  const unsigned long long dev_bit = current_device_bit();
  if (!(attr_done & dev_bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
    attr_done |= dev_bit;
  }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const int ntiles = cdiv(a.M, WS_TM);
  // two blocks per CU in sequence: measured 4-6 % faster than one (the second block's weight prologue hides under the first one's tail)
  hipLaunchKernelGGL(gemm_ws_kernel, dim3(std::min(ntiles, 2 * n_cu)), dim3(512), WS_LDS, st, a);
  LT_LAUNCH_CHECK();
  return 0;
}

}  // namespace lt
