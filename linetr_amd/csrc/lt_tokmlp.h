// The WordPositionalEncoder / LinePositionalEncoder MLP up to its last ReLU in ONE kernel (bf16x6):  a4 = relu(W4 relu(W3 relu(W2 relu(W1 f + b1) + b2) + b3) + b4)
// for every real token row of the batch, f = (x, y, score), or for every sub-line, f = (mid x, mid y, response, cos 2 theta, sin 2 theta)
// (models/line_transformer.py:9-20, 40-73; BatchNorm folded).  The fifth, linear layer is applied after the pooling (word encoder,
// DESIGN.md section 3) or by the next GEMM (line encoder).  Replaces mlp123_kernel + the weight-stationary K = 128 GEMM for
// the word encoder: the 128-channel activations (512 B per token written and read back) never reach HBM, the kernel reads 12 bytes
// and writes 1 KiB per token.
//
// A block of 8 waves per CU lives for the whole launch and walks tiles of 64 token rows.  Layer 4's weights stay in registers (96
// VGPRs per wave, as in lt_gemm_ws.h: wave w owns output channels 32 w .. + 31), those of layers 2 and 3 in LDS as split-tile images
// (lt_st_image.h), and every layer's activations go to the next one through LDS as split-tile images of bf16 planes:
//   A  layer 1 on the VALU, rounded like mlp123_kernel (wave & 3 = 8 channels, lane = token), in the MFMA slots
//      of the PREVIOUS tile's D                                                                                 -> image A1 [64 x 32]
//   B  layer 2, 4 tiles of 32 channels x 32 tokens, K = 32 (waves 0-3)                                          -> image A2 [64 x 64]
//   C  layer 3, 8 tiles, K = 64 (one per wave)                                                                  -> image A3 [64 x 128]
//   D  layer 4 exactly as gemm_ws_kernel: 96 MFMA slots per wave in a fixed order, epilogue of one 32-token half in the slots of
//      the other, the second half's epilogue at the head of the next tile
// with a block barrier before B, before C and before D.  All products are the transposed
// ones (weights = MFMA A operand), so an accumulator tile is [32 channels][32 tokens], a lane owns one token, and one half-wave swap
// per register pair turns it into two 8-channel pieces of that token's row: exactly the 16-byte pieces of the next image.
// Measured at cfg3 (291 208 token rows): 140-150 us against 78 (mlp123) + 108 (K = 128 GEMM) for the pair it replaces; the line encoder
// (25 472 rows): 29 against 16.5 + 23 us; a single pair (4 378 / 398 rows): 15.7 us each, mostly the weight prologue, against 21 us each.  A software-pipelined
// variant (tiles of 32 rows, every image double-buffered, layer 4 of tile s - 3 / layer 3 of s - 2 on waves 4-7 / layer 2 of s - 1 on
// waves 0-1 / layer 1 of s in ONE barrier interval) was built and measured on the same box: 165 us -- the matrix pipe is not idle
// for lack of work here, the MFMAs alone take 80 us of layer 4's 110 at the clock the power cap allows (tools/ubench/ws_gemm_bench.hip:
// removing every B-fragment read changes nothing), so filling the A / B / C gaps buys nothing and the extra barriers cost.
#pragma once
#include "lt_gemm_ws.h"

namespace lt {

struct TokMlpArgs {
  // WORD: p0 = token coordinates [rows][2] (pixels), p1 = scores [rows]; LINE: p0 = sub-line end points [rows][4], p1 = responses
  // [rows], p2 = (cos 2 theta, sin 2 theta) [rows][2]  (word_feat / line_feat of lt_model.h)
  const float* p0 = nullptr; const float* p1 = nullptr; const float* p2 = nullptr;
  int64_t rows = 0;
  float cx = 0.f, cy = 0.f, scale = 1.f;         // normalize_keylines (line_transformer.py:22-38)
  const float* W1 = nullptr; const float* b1 = nullptr;            // [32][3 or 5], [32]
  const unsigned char* W2st = nullptr; const float* b2 = nullptr;  // split-tile image of [64][32]
  const unsigned char* W3st = nullptr; const float* b3 = nullptr;  // of [128][64]
  const unsigned char* W4st = nullptr; const float* b4 = nullptr;  // of [256][128]
  float* Y = nullptr; int ldy = 0;               // [rows][256 ..]
};

constexpr int TK_TM = 64, TK_TB = TK_TM / 16;
constexpr int TK_A1 = 0;                                   // 2 K steps x 4 token blocks x 1536
constexpr int TK_A2 = TK_A1 + 2 * TK_TB * ST_RB;           // 12 288: 4 K steps
constexpr int TK_A3 = TK_A2 + 4 * TK_TB * ST_RB;           // 36 864: 8 K steps
constexpr int TK_W2 = TK_A3 + 8 * TK_TB * ST_RB;           // 86 016: [2 K steps][4 row blocks] (the image's pad row blocks dropped)
constexpr int TK_W3 = TK_W2 + 2 * 4 * ST_RB;               // 98 304: [4 K steps][8 row blocks]
constexpr int TK_P = TK_W3 + 4 * 8 * ST_RB;                // 147 456: W1 as [32][8], b2, b3, b4
constexpr int TK_P_FLOATS = 32 * 8 + 64 + 128 + 256;        // W1 rows as [w0 .. w4, -, -, bias]
constexpr int TK_LDS = TK_P + TK_P_FLOATS * 4;             // 150 272

template <bool WORD>
__device__ __forceinline__ void tok_mlp_body(const TokMlpArgs& a, const int block, const int nblocks) {
  constexpr int IN = WORD ? 3 : 5;
  extern __shared__ __attribute__((aligned(1024))) unsigned char tk_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h2 = lane >> 5, lq = lane & 31;
  const int ntiles = (int)((a.rows + TK_TM - 1) / TK_TM);
  int tile = block;
  if (tile >= ntiles) return;
  const int step = nblocks;
  const int lfrag = ((lane >> 4) & 1) * ST_RB + (lane >> 5) * 256 + (lane & 15) * 16;
  float* prm = reinterpret_cast<float*>(tk_smem + TK_P);

  // ---- once per block: weights of layers 2 / 3 and the small vectors into LDS, this wave's slice of layer 4 into registers
  for (int i = tid; i < 2 * 4 * ST_RB / 16; i += 512) {     // W2: K step kt, row blocks 0..3 of the 8 the padded image has
    const int kt = i / (4 * ST_RB / 16), r = i % (4 * ST_RB / 16);
    *reinterpret_cast<u32x4*>(tk_smem + TK_W2 + i * 16) = *reinterpret_cast<const u32x4*>(a.W2st + (int64_t)kt * 8 * ST_RB + r * 16);
  }
  for (int i = tid; i < 4 * 8 * ST_RB / 16; i += 512)
    *reinterpret_cast<u32x4*>(tk_smem + TK_W3 + i * 16) = *reinterpret_cast<const u32x4*>(a.W3st + (int64_t)i * 16);
  if (tid < 256) {
    const int k = tid >> 3, c = tid & 7;
    prm[tid] = c < IN ? a.W1[k * IN + c] : (c == 7 ? a.b1[k] : 0.f);
  } else if (tid < 320) prm[tid] = a.b2[tid - 256];
  else if (tid < 448) prm[tid] = a.b3[tid - 320];
  for (int i = tid; i < 256; i += 512) prm[448 + i] = a.b4[i];
  bf16x8 wreg[8][3];
  {
    const unsigned char* wp = a.W4st + (int64_t)(2 * wave) * ST_RB + lfrag;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int p = 0; p < 3; ++p) wreg[kt][p] = *reinterpret_cast<const bf16x8*>(wp + (int64_t)kt * 16 * ST_RB + p * ST_CHUNK);
  }
  // token features of the tile: waves 0-3, lane = token (each of the four waves needs all 64 tokens)
  float feat[IN];
  auto load_feat = [&](int t) {
    int64_t row = (int64_t)t * TK_TM + lane;
    row = row < a.rows ? row : a.rows - 1;
    if constexpr (WORD) word_feat(a.p0, a.p1, row, a.cx, a.cy, a.scale, feat);
    else line_feat(a.p0, a.p1, a.p2, row, a.cx, a.cy, a.scale, feat);
  };
  load_feat(tile);
  __syncthreads();

  constexpr int TW[6] = {2, 1, 0, 1, 0, 0}, TA[6] = {0, 1, 2, 0, 1, 0};     // smallest cross terms first
  auto init_acc = [&](f32x16& acc, const float* bias32) {  // register 4 b + c of a tile is channel 8 b + 4 h2 + c of its 32
#pragma unroll
    for (int bq = 0; bq < 4; ++bq) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(bias32 + 8 * bq + 4 * h2);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[4 * bq + c] = v[c];
    }
  };
  auto write_piece = [&](const float* x, unsigned char* dst) {   // 8 consecutive channels of one token -> three 16-byte plane pieces
    unsigned p0[3], p1[3], p2[3], p3[3];
    split_pair<3>(x[0], x[1], p0); split_pair<3>(x[2], x[3], p1);
    split_pair<3>(x[4], x[5], p2); split_pair<3>(x[6], x[7], p3);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + p * ST_CHUNK) = u32x4{p0[p], p1[p], p2[p], p3[p]};
  };
  // ReLU, then the two 8-channel pieces (channels 8 h2 .. and 16 + 8 h2 .. of the tile's 32) of token `tok` into image `img`,
  // whose K index is the channel: piece P = channel / 8 sits in K step P / 2, k half P & 1
  auto tile_to_image = [&](f32x16& acc, int img, int ch0, int tok) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[r], 0.f);
#pragma unroll
    for (int c = 0; c < 4; ++c) { halves_swap(v[c], v[4 + c]); halves_swap(v[8 + c], v[12 + c]); }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int P = ch0 / 8 + 2 * g + h2;
      write_piece(v + 8 * g, tk_smem + img + ((P >> 1) * TK_TB + (tok >> 4)) * ST_RB + (P & 1) * 256 + (tok & 15) * 16);
    }
  };
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
  int prev_tile = -1;
  auto epi_step = [&](f32x16& acc, int t, int j, int e) {  // e = 0..10: ReLU, 8 half-wave swap pairs, 2 x 32-byte-run stores
    if (e == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.f);
    } else if (e <= 8) {
      const int k = e - 1, lo = k < 4 ? k : 8 + (k - 4);
      float x = acc[lo], y = acc[lo + 4];
      halves_swap(x, y);
      acc[lo] = x; acc[lo + 4] = y;
    } else {
      const int part = e - 9, o = 8 * part;
      const int64_t row = (int64_t)t * TK_TM + 32 * j + lq;
      if (t >= 0 && row < a.rows) {
        float* yp = a.Y + row * a.ldy + 32 * wave + 8 * h2 + 16 * part;
        *reinterpret_cast<f32x4*>(yp) = f32x4{acc[o], acc[o + 1], acc[o + 2], acc[o + 3]};
        *reinterpret_cast<f32x4*>(yp + 4) = f32x4{acc[o + 4], acc[o + 5], acc[o + 6], acc[o + 7]};
      }
    }
  };

  // ---- A: layer 1 on the VALU (one multiply and one add per term, bias first, like mlp123_kernel / word_mlp1_kernel), cut into
  // steps that ride in the MFMA slots of D: piece wave & 3 (8 channels) of token `lane`.  Waves 4-7 repeat the work of waves 0-3 and
  // store the same bytes: no wave-dependent branch inside the slot loop.
  float av[8], af[IN];
  auto a_step = [&](int e) {
#pragma clang fp contract(off)
    if (e == 0) {
#pragma unroll
      for (int i = 0; i < IN; ++i) af[i] = feat[i];        // (free the registers for the next tile's loads)
    } else if (e <= 8) {
      const int c = e - 1;
      const float* wr = prm + (8 * (wave & 3) + c) * 8;
      const f32x4 wa = *reinterpret_cast<const f32x4*>(wr), wb = *reinterpret_cast<const f32x4*>(wr + 4);
      const float wv[8] = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3]};
      float t = wv[7];                                     // bias
#pragma unroll
      for (int i = 0; i < IN; ++i) t += wv[i] * af[i];
      av[c] = fmaxf(t, 0.f);
    } else {
      write_piece(av, tk_smem + TK_A1 + (((wave & 3) >> 1) * TK_TB + (lane >> 4)) * ST_RB + (wave & 1) * 256 + (lane & 15) * 16);
    }
  };
#pragma unroll
  for (int e = 0; e < 10; ++e) a_step(e);                  // the first tile's layer 1
#pragma unroll 1
  for (; tile < ntiles; tile += step) {
    // the second half of the previous tile leaves while this tile's first layers run
#pragma unroll
    for (int e = 0; e < 11; ++e) epi_step(acc1, prev_tile, 1, e);
    load_feat(tile + step);                                // consumed in the slots of D (past the end: clamps to the last row)
    __syncthreads();                                       // A1 of this tile (written under the previous tile's D) is complete
    // ---- B: layer 2, K = 32: wave (cg = wave & 1: 32 channels, th = wave >> 1: 32 tokens), waves 0-3
    if (wave < 4) {
      const int cg = wave & 1, th = wave >> 1;
      init_acc(acc0, prm + 256 + 32 * cg);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        bf16x8 wf[3], zf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          wf[p] = *reinterpret_cast<const bf16x8*>(tk_smem + TK_W2 + (kt * 4 + 2 * cg) * ST_RB + lfrag + p * ST_CHUNK);
          zf[p] = *reinterpret_cast<const bf16x8*>(tk_smem + TK_A1 + (kt * TK_TB + 2 * th) * ST_RB + lfrag + p * ST_CHUNK);
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) acc0 = mfma_split<0>(wf[TW[t]], zf[TA[t]], acc0);
      }
      tile_to_image(acc0, TK_A2, 32 * cg, 32 * th + lq);
    }
    __syncthreads();
    // ---- C: layer 3, K = 64: wave (cg = wave & 3, th = wave >> 2)
    {
      const int cg = wave & 3, th = wave >> 2;
      init_acc(acc0, prm + 320 + 32 * cg);
      bf16x8 wf[2][3], zf[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        wf[0][p] = *reinterpret_cast<const bf16x8*>(tk_smem + TK_W3 + (2 * cg) * ST_RB + lfrag + p * ST_CHUNK);
        zf[0][p] = *reinterpret_cast<const bf16x8*>(tk_smem + TK_A2 + (2 * th) * ST_RB + lfrag + p * ST_CHUNK);
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        if (kt + 1 < 4) {
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            wf[(kt + 1) & 1][p] = *reinterpret_cast<const bf16x8*>(tk_smem + TK_W3 + ((kt + 1) * 8 + 2 * cg) * ST_RB + lfrag + p * ST_CHUNK);
            zf[(kt + 1) & 1][p] = *reinterpret_cast<const bf16x8*>(tk_smem + TK_A2 + ((kt + 1) * TK_TB + 2 * th) * ST_RB + lfrag + p * ST_CHUNK);
          }
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) acc0 = mfma_split<0>(wf[kt & 1][TW[t]], zf[kt & 1][TA[t]], acc0);
      }
      tile_to_image(acc0, TK_A3, 32 * cg, 32 * th + lq);
    }
    __syncthreads();
    // ---- D: layer 4, the slot loop of gemm_ws_kernel on image A3
    {
      const unsigned char* src = tk_smem + TK_A3 + lfrag;
      bf16x8 z[3][3];
      auto read_z1 = [&](int g, int p) {                   // g = 8 j + kt
        z[g % 3][p] = *reinterpret_cast<const bf16x8*>(src + ((g & 7) * TK_TB + 2 * (g >> 3)) * ST_RB + p * ST_CHUNK);
      };
#pragma unroll
      for (int p = 0; p < 3; ++p) { read_z1(0, p); read_z1(1, p); }
      init_acc(acc0, prm + 448 + 32 * wave);
#pragma clang loop unroll(full)
      for (int m = 0; m < 96; ++m) {
        const int g = m / 6, t = m % 6, kt = g & 7;
        if (m == 48) init_acc(acc1, prm + 448 + 32 * wave);
        if (g < 8) acc0 = mfma_split<0>(wreg[kt][TW[t]], z[g % 3][TA[t]], acc0);
        else acc1 = mfma_split<0>(wreg[kt][TW[t]], z[g % 3][TA[t]], acc1);
        if (t < 3 && g + 2 < 16) read_z1(g + 2, t);
        if (m >= 2 && m < 12) a_step(m - 2);               // layer 1 of the NEXT tile into A1 (B of this tile has left it)
        if (m >= 50 && m < 61) epi_step(acc0, tile, 0, m - 50);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    prev_tile = tile;
  }
#pragma unroll
  for (int e = 0; e < 11; ++e) epi_step(acc1, prev_tile, 1, e);
}

template <bool WORD>
__global__ __launch_bounds__(512) void tok_mlp_kernel(TokMlpArgs a) { tok_mlp_body<WORD>(a, blockIdx.x, gridDim.x); }

// both encoders in one launch (small batches: the two sets of blocks fit the chip side by side, one launch and one weight prologue
// less on the critical path of a single pair): blocks [0, word_blocks) run the word encoder, the rest the line encoder
__global__ __launch_bounds__(512) void tok_mlp_dual_kernel(TokMlpArgs aw, TokMlpArgs al, int word_blocks) {
  if ((int)blockIdx.x < word_blocks) tok_mlp_body<true>(aw, blockIdx.x, word_blocks);
  else tok_mlp_body<false>(al, blockIdx.x - word_blocks, gridDim.x - word_blocks);
}

inline int tok_mlp_launch(const TokMlpArgs& a, bool word, hipStream_t st) {
  if (a.rows <= 0) return 0;
  if (!a.p0 || !a.p1 || (!word && !a.p2) || !a.W1 || !a.W2st || !a.W3st || !a.W4st || !a.b1 || !a.b2 || !a.b3 || !a.b4 || !a.Y || a.ldy % 4)
    return fail(LINETR_E_ARG, "tok_mlp: missing operand");
  static unsigned long long attr_done = 0;
  const unsigned long long dev_bit = current_device_bit();
  if (!(attr_done & dev_bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tok_mlp_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TK_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tok_mlp_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TK_LDS);
    attr_done |= dev_bit;
  }
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const int64_t ntiles = (a.rows + TK_TM - 1) / TK_TM;
  const dim3 grid((unsigned)std::min<int64_t>(ntiles, n_cu));
  if (word) hipLaunchKernelGGL(tok_mlp_kernel<true>, grid, dim3(512), TK_LDS, st, a);
  else hipLaunchKernelGGL(tok_mlp_kernel<false>, grid, dim3(512), TK_LDS, st, a);
  LT_LAUNCH_CHECK();
  return 0;
}

// large batches: every persistent block walks its share of the word encoder's tiles, then its share of the line encoder's (one launch
// instead of two; the second weight prologue costs a block ~4 us, a separate launch of the line encoder 29 us at cfg3)
__global__ __launch_bounds__(512) void tok_mlp_seq_kernel(TokMlpArgs aw, TokMlpArgs al) {
  tok_mlp_body<true>(aw, blockIdx.x, gridDim.x);
  __syncthreads();                                         // the line encoder's weights replace the word encoder's in LDS
  tok_mlp_body<false>(al, blockIdx.x, gridDim.x);
}

// both encoders side by side: only when their blocks (one per 64-row tile) fit the chip together
inline bool tok_mlp_dual_fits(int64_t rows_word, int64_t rows_line) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return rows_word > 0 && rows_line > 0 && (rows_word + TK_TM - 1) / TK_TM + (rows_line + TK_TM - 1) / TK_TM <= n_cu;
}
// one launch for both encoders: side by side when the blocks fit the chip, otherwise one after the other inside every block
inline int tok_mlp_launch_dual(const TokMlpArgs& aw, const TokMlpArgs& al, hipStream_t st) {
  const int64_t tw = (aw.rows + TK_TM - 1) / TK_TM, tl = (al.rows + TK_TM - 1) / TK_TM;
  if (aw.rows <= 0 || al.rows <= 0) return fail(LINETR_E_ARG, "tok_mlp: empty encoder input");
  if (!aw.p0 || !aw.p1 || !al.p0 || !al.p1 || !al.p2 || !aw.Y || !al.Y || aw.ldy % 4 || al.ldy % 4) return fail(LINETR_E_ARG, "tok_mlp: missing operand");
  static unsigned long long attr_done = 0;
  const unsigned long long dev_bit = current_device_bit();
  if (!(attr_done & dev_bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tok_mlp_dual_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TK_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(tok_mlp_seq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TK_LDS);
    attr_done |= dev_bit;
  }
  if (tok_mlp_dual_fits(aw.rows, al.rows)) {
    hipLaunchKernelGGL(tok_mlp_dual_kernel, dim3((unsigned)(tw + tl)), dim3(512), TK_LDS, st, aw, al, (int)tw);
  } else {
    int dev = 0;
    hipDeviceProp_t prop;
    static int n_cu = 0;
    if (!n_cu) n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    hipLaunchKernelGGL(tok_mlp_seq_kernel, dim3((unsigned)std::min<int64_t>(std::max(tw, tl), n_cu)), dim3(512), TK_LDS, st, aw, al);
  }
  LT_LAUNCH_CHECK();
  return 0;
}

}  // namespace lt
