// Non-GEMM device stages of LineTransformer.forward (models/line_transformer.py:225-249).
#pragma once
#include "lt_common.h"
#include "lt_token.h"
#include "lt_gemm_split.h"

namespace lt {

// ---------------------------------------------------------------------------------------------
// First MLP layer of the two positional encoders on the VALU (3 -> 32 and 5 -> 32, BN folded, ReLU),
// fused with normalize_keylines (models/line_transformer.py:22-38, :40-73).
// 8 threads per row, 4 output channels each.
// ---------------------------------------------------------------------------------------------
template <bool RELU = true>      // RELU = false: the pre-activations, for BatchNorm in training mode (lt_bntrain.h)
__global__ void word_mlp1_kernel(const float* __restrict__ pnt, const float* __restrict__ score, int64_t rows,
                                 float cx, float cy, float scale, const float* __restrict__ W /*[32][3]*/,
                                 const float* __restrict__ b, float* __restrict__ out /*[rows][32]*/) {
#pragma clang fp contract(off)
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = gid >> 3;
  if (row >= rows) return;
  const int c0 = (int)(gid & 7) * 4;
  const float x = (pnt[row * 2 + 0] - cx) / scale;
  const float y = (pnt[row * 2 + 1] - cy) / scale;
  const float s = score[row];
  f32x4 o;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float* w = W + (c0 + c) * 3;
    float v = b[c0 + c] + w[0] * x + w[1] * y + w[2] * s;
    o[c] = RELU ? fmaxf(v, 0.f) : v;
  }
  *reinterpret_cast<f32x4*>(out + row * 32 + c0) = o;
}

template <bool RELU = true>
__global__ void line_mlp1_kernel(const float* __restrict__ sublines /*[N][2][2]*/, const float* __restrict__ resp,
                                 const float* __restrict__ angle, int N, float cx, float cy, float scale,
                                 const float* __restrict__ W /*[32][5]*/, const float* __restrict__ b,
                                 float* __restrict__ out /*[N][32]*/) {
#pragma clang fp contract(off)
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = gid >> 3;
  if (row >= N) return;
  const int c0 = (gid & 7) * 4;
  const float* sl = sublines + (int64_t)row * 4;
  const float sx = (sl[0] - cx) / scale, sy = (sl[1] - cy) / scale;
  const float ex = (sl[2] - cx) / scale, ey = (sl[3] - cy) / scale;
  const float in[5] = {(sx + ex) / 2.f, (sy + ey) / 2.f, resp[row], angle[row * 2], angle[row * 2 + 1]};
  f32x4 o;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float* w = W + (c0 + c) * 5;
    float v = b[c0 + c];
#pragma unroll
    for (int i = 0; i < 5; ++i) v += w[i] * in[i];
    o[c] = RELU ? fmaxf(v, 0.f) : v;
  }
  *reinterpret_cast<f32x4*>(out + (int64_t)row * 32 + c0) = o;
}

// ---- layers 1-3 of a positional-encoder MLP in ONE pass (in -> 32 -> 64 -> 128, BN folded, ReLU) ---------------
// Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) on the TRANSPOSED product: neurons are the MFMA rows (A = weights,
// resident in VGPRs for the whole kernel), the 32 rows (tokens / sub-lines) of a step are the MFMA columns
// (B = activations).  The C/D layout -- lane = column, registers = rows (r&3) + 8(r>>2) + 4(lane>>5) -- is then
// exactly a B-operand layout of the next layer with the K index permuted, so bias + ReLU happen in registers and the
// activations never leave them: no LDS, no broadcasts (a v_readlane version spent 2/3 of its time in readlanes, a
// lane-per-row version with scalar-loaded weights thrashed the 16 KiB scalar cache).  Replaces mlp_first + the
// K = 32 and K = 64 GEMM launches (all prologue/epilogue at 9-30 TF) and keeps a1 / a2 out of HBM.
__device__ __forceinline__ void word_feat(const float* __restrict__ pnt, const float* __restrict__ score, int64_t row,
                                          float cx, float cy, float scale, float (&in)[3]) {
#pragma clang fp contract(off)
  in[0] = (pnt[row * 2 + 0] - cx) / scale;
  in[1] = (pnt[row * 2 + 1] - cy) / scale;
  in[2] = score[row];
}
__device__ __forceinline__ void line_feat(const float* __restrict__ sublines, const float* __restrict__ resp,
                                          const float* __restrict__ angle, int64_t row, float cx, float cy, float scale,
                                          float (&in)[5]) {
#pragma clang fp contract(off)
  const float* sl = sublines + row * 4;
  const float sx = (sl[0] - cx) / scale, sy = (sl[1] - cy) / scale;
  const float ex = (sl[2] - cx) / scale, ey = (sl[3] - cy) / scale;
  in[0] = (sx + ex) / 2.f; in[1] = (sy + ey) / 2.f; in[2] = resp[row]; in[3] = angle[row * 2]; in[4] = angle[row * 2 + 1];
}
// K index served by register r of an accumulator tile in lane half h (see above)
__device__ __forceinline__ constexpr int cd_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <bool WORD>
__global__ __launch_bounds__(256) void mlp123_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                     const float* __restrict__ p2, int64_t rows, int rows_per_wave,
                                                     float cx, float cy, float scale,
                                                     const float* __restrict__ W1 /*[32][IN]*/, const float* __restrict__ b1,
                                                     const float* __restrict__ W2 /*[64][32]*/, const float* __restrict__ b2,
                                                     const float* __restrict__ W3 /*[128][64]*/, const float* __restrict__ b3,
                                                     float* __restrict__ out /*[rows][128]*/) {
  constexpr int IN = WORD ? 3 : 5;
  const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t r_begin = wave * rows_per_wave;
  const int64_t r_end = r_begin + rows_per_wave < rows ? r_begin + rows_per_wave : rows;
  // Stage W2 / W3 / biases in LDS once per block (coalesced float4 loads; row strides 33 / 65 floats so that the
  // per-lane gathers below are bank-conflict-free) -- per-lane gathers straight from global touch 32 lines per load.
  // LDS: the weight images are dead once every wave has gathered its registers; the per-wave output staging tiles
  // (32 rows x 68 floats each) reuse that space, which keeps the block at 43 KiB so that it can share a CU with the
  // blocks of a concurrent kernel (the NHWC transposition on the side stream) instead of waiting for them to drain.
  __shared__ __attribute__((aligned(16))) float sAll[64 * 33 + 128 * 65];
  __shared__ __attribute__((aligned(16))) float sW1[32 * 8];
  __shared__ __attribute__((aligned(16))) float sB[64 + 128];
  float* sW2 = sAll;
  float* sW3 = sAll + 64 * 33;
  float* stg = sAll + (threadIdx.x >> 6) * (32 * 68);
  static_assert(4 * 32 * 68 <= 64 * 33 + 128 * 65, "staging tiles must fit into the dead weight images");
  {  // all 10 loads of a thread are issued before the first LDS write (a load -> write loop pays the L2 latency 10x)
    f32x4 v2[2], v3[8];
#pragma unroll
    for (int u = 0; u < 2; ++u) v2[u] = *reinterpret_cast<const f32x4*>(W2 + (threadIdx.x + 256 * u) * 4);
#pragma unroll
    for (int u = 0; u < 8; ++u) v3[u] = *reinterpret_cast<const f32x4*>(W3 + (threadIdx.x + 256 * u) * 4);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int idx = threadIdx.x + 256 * u, r = idx / 8, c = (idx % 8) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) sW2[r * 33 + c + q] = v2[u][q];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = threadIdx.x + 256 * u, r = idx / 16, c = (idx % 16) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) sW3[r * 65 + c + q] = v3[u][q];
    }
  }
  {  // layer-1 rows as [w0 .. w4, -, -, bias]
    const int k = threadIdx.x >> 3, c = threadIdx.x & 7;
    sW1[threadIdx.x] = c < IN ? W1[k * IN + c] : (c == 7 ? b1[k] : 0.f);
  }
  if (threadIdx.x < 64) sB[threadIdx.x] = b2[threadIdx.x];
  else if (threadIdx.x < 192) sB[threadIdx.x] = b3[threadIdx.x - 64];
  __syncthreads();
  // resident A operands.  Layer 2: K step s uses k = 2s + h (layer 1 is computed straight into that order).
  // Layer 3: K step (i, r) uses k = 32 i + cd_row(r, h), the order layer 2's accumulators come out in.
  float w2[2][16], w3[4][32];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) w2[i][s2] = sW2[(32 * i + col) * 33 + 2 * s2 + h];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) w3[j][i * 16 + r] = sW3[(32 * j + col) * 65 + 32 * i + cd_row(r, h)];
  __syncthreads();                       // all waves hold their weights: sAll becomes the staging area
  if (r_begin >= r_end) return;
  float feat[IN], feat_next[IN];
  {
    int64_t row = r_begin + col;
    row = row < r_end ? row : r_end - 1;            // tail columns recompute the last row; their stores are masked
    if constexpr (WORD) word_feat(p0, p1, row, cx, cy, scale, feat);
    else line_feat(p0, p1, p2, row, cx, cy, scale, feat);
  }
  for (int64_t base = r_begin; base < r_end; base += 32) {
    {  // next step's inputs are requested now and consumed after this step's 160 MFMAs
      int64_t row = base + 32 + col;
      row = row < r_end ? row : r_end - 1;
      if constexpr (WORD) word_feat(p0, p1, row, cx, cy, scale, feat_next);
      else line_feat(p0, p1, p2, row, cx, cy, scale, feat_next);
    }
    // layer 1 on the VALU, rounded like word_mlp1_kernel / line_mlp1_kernel; this lane's neurons are k = 2s + h, their
    // weights come from LDS (kept as wave-uniform scalars they overflowed the SGPR file and were spilled lane by lane)
    float a1[16];
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int s1 = 0; s1 < 16; ++s1) {
        const float* wr = sW1 + (2 * s1 + h) * 8;
        const f32x4 wa = *reinterpret_cast<const f32x4*>(wr), wb = *reinterpret_cast<const f32x4*>(wr + 4);
        const float wv[8] = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3]};
        float t = wv[7];                       // bias
#pragma unroll
        for (int i = 0; i < IN; ++i) t += wv[i] * feat[i];
        a1[s1] = fmaxf(t, 0.f);
      }
    }
    // layer 2: a2^T[64 neurons][32 rows] = W2 a1^T + b2.  The two 32-neuron tiles are interleaved: a chain of
    // dependent MFMAs on ONE accumulator runs at a fraction of the pipe rate (measured 4x slower per step).
    f32x16 acc2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; r += 4) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(sB + 32 * i + cd_row(r, h));
        acc2[i][r] = bv[0]; acc2[i][r + 1] = bv[1]; acc2[i][r + 2] = bv[2]; acc2[i][r + 3] = bv[3];
      }
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2)
#pragma unroll
      for (int i = 0; i < 2; ++i) acc2[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2[i][s2], a1[s2], acc2[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][r] = fmaxf(acc2[i][r], 0.f);
    // layer 3, two 32-neuron tiles at a time (two independent accumulator chains); the 16 registers of a tile are
    // 4 runs of 4 consecutive neurons
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      f32x16 acc3[2];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; r += 4) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(sB + 64 + 32 * (2 * jp + q) + cd_row(r, h));
          acc3[q][r] = bv[0]; acc3[q][r + 1] = bv[1]; acc3[q][r + 2] = bv[2]; acc3[q][r + 3] = bv[3];
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            acc3[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(w3[2 * jp + q][i * 16 + r], acc2[i][r], acc3[q], 0, 0, 0);
      // A lane owns one ROW in the accumulators, so storing from them would write 32-byte pieces 512 bytes apart.
      // Through this wave's staging tile ([32 rows][64 neurons of this pass], stride 68 floats) every store instruction
      // writes four complete 256-byte half rows.  LDS operations of one wave execute in order: no barrier needed.
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; r += 4)
          *reinterpret_cast<f32x4*>(stg + col * 68 + 32 * q + cd_row(r, h)) =
              f32x4{fmaxf(acc3[q][r], 0.f), fmaxf(acc3[q][r + 1], 0.f), fmaxf(acc3[q][r + 2], 0.f), fmaxf(acc3[q][r + 3], 0.f)};
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int t = 4 * i + (lane >> 4), pc = (lane & 15) * 4;
        if (base + t < r_end)
          *reinterpret_cast<f32x4*>(out + (base + t) * 128 + 64 * jp + pc) = *reinterpret_cast<const f32x4*>(stg + t * 68 + pc);
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int i = 0; i < IN; ++i) feat[i] = feat_next[i];
  }
}

// ---------------------------------------------------------------------------------------------
// CLS-row attention pooling of the line-descriptive layer (models/line_attention.py:6-75 restricted
// to query row 0, the only row that reaches the output: models/line_transformer.py:128).
//
// For head h the CLS query q_h is a model constant, so
//     score_hj = q_h . (Wk_h x_j + bk_h) = u_h . x_j + c_h ,   x_j = desc_j + W5 a4_j + b5
//              = u_h . desc_j + (W5^T u_h) . a4_j + const_h
// and the attention output only needs the attention-weighted means of desc_j and a4_j (the value and
// last-MLP projections are linear, they are applied after pooling by a [N x 544] x [544 x 64] GEMM).
// One 256-thread block per sub-line.
//   pooled[n][h] = [ sum_j p_hj desc_j (256) | sum_j p_hj a4_j (256) | p_h0 | 0 x31 ]
// ---------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void cls_pool_kernel(const float* __restrict__ desc /*[N][T][256]*/,
                                                       const float* __restrict__ a4 /*[N*T][256]*/, int T,
                                                       ClsPoolConst cc, float* __restrict__ pooled /*[N][4][544]*/) {
  extern __shared__ float sm[];  // [4][T+1]
  const int S = T + 1;
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* dn = desc + (int64_t)n * T * D;
  const float* an = a4 + (int64_t)n * T * D;
  f32x4 u[4], u2[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    u[h] = *reinterpret_cast<const f32x4*>(cc.U + h * D + lane * 4);
    u2[h] = *reinterpret_cast<const f32x4*>(cc.U2 + h * D + lane * 4);
  }
  for (int j = wave; j < T; j += 4) {
    const f32x4 dv = *reinterpret_cast<const f32x4*>(dn + j * D + lane * 4);
    const f32x4 av = *reinterpret_cast<const f32x4*>(an + j * D + lane * 4);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      float p = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) p += dv[c] * u[h][c] + av[c] * u2[h][c];
      p = wave_sum(p);
      if (lane == 0) sm[h * S + 1 + j] = p + cc.c_tok[h];
    }
  }
  if (tid < 4) sm[tid * S] = cc.s_cls[tid];
  __syncthreads();
  if (tid < 4) {  // softmax over the S keys of head `tid` (F.softmax, line_attention.py:18)
    float* s = sm + tid * S;
    float mx = s[0];
    for (int j = 1; j < S; ++j) mx = fmaxf(mx, s[j]);
    float sum = 0.f;
    for (int j = 0; j < S; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
    for (int j = 0; j < S; ++j) s[j] = s[j] / sum;
  }
  __syncthreads();
  float db[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < T; ++j) {
    const float dv = dn[j * D + tid], av = an[j * D + tid];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float p = sm[h * S + 1 + j];
      db[h] += p * dv;
      ab[h] += p * av;
    }
  }
  float* out = pooled + (int64_t)n * 4 * POOLW;
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    out[h * POOLW + tid] = db[h];
    out[h * POOLW + 256 + tid] = ab[h];
  }
  if (tid < 4 * 32) {
    const int h = tid >> 5, i = tid & 31;
    out[h * POOLW + 512 + i] = i == 0 ? sm[h * S] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// Fused variant for the batched fast path (linetr_describe): same mathematics as cls_pool_kernel, but
//   * token descriptors are sampled from the NHWC map on the fly (sample_one) and kept in LDS -- the
//     [N,T,256] tensor of the reference is never materialised;
//   * only REAL tokens are visited; every zero-padding slot of an image holds the same coordinate (0,0)
//     (models/line_process.py:133-139), i.e. identical descriptor / a4 row / score, so the padding slots
//     enter the softmax as ONE key with multiplicity n_pad.
// a4 / cpnt are indexed by the compact token list (rec.first_tok), the image's padding token sits at
// first_pad + image.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cls_pool_fused_kernel(
    const LinetrLineRec* __restrict__ recs, const int* __restrict__ sub2line_g, const float* __restrict__ cpnt,
    const float* __restrict__ a4, int64_t first_pad, int T, const float* __restrict__ nhwc, int Hc, int Wc,
    int align_corners, ClsPoolConst cc, float* __restrict__ pooled /*[N][4][544]*/) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* descs = sm;                       // [(T+1)][256]
  float* sc = sm + (T + 1) * D;            // [4][T+2]   scores -> weights
  const int SS = T + 2;
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const LinetrLineRec r = recs[sub2line_g[n]];
  const int j = n - r.first_sub;
  const int n_valid = min(T, r.n_tok - j * T);
  const int n_pad = T - n_valid;
  const int ntk = n_valid + (n_pad > 0 ? 1 : 0);
  const int64_t tok0 = (int64_t)r.first_tok + (int64_t)j * T;
  const int64_t padrow = first_pad + r.image;
  const float* nhwc_img = nhwc + (int64_t)r.image * Hc * Wc * D;
  f32x4 u[4], u2[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    u[h] = *reinterpret_cast<const f32x4*>(cc.U + h * D + lane * 4);
    u2[h] = *reinterpret_cast<const f32x4*>(cc.U2 + h * D + lane * 4);
  }
  for (int jj = wave; jj < ntk; jj += 4) {
    const int64_t row = jj < n_valid ? tok0 + jj : padrow;
    const f32x4 dv = sample_one(cpnt[row * 2], cpnt[row * 2 + 1], nhwc_img, Hc, Wc, align_corners, lane);
    *reinterpret_cast<f32x4*>(descs + jj * D + lane * 4) = dv;
    const f32x4 av = *reinterpret_cast<const f32x4*>(a4 + row * D + lane * 4);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      float p = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) p += dv[c] * u[h][c] + av[c] * u2[h][c];
      p = wave_sum(p);
      if (lane == 0) sc[h * SS + 1 + jj] = p + cc.c_tok[h];
    }
  }
  if (tid < 4) sc[tid * SS] = cc.s_cls[tid];
  __syncthreads();
  if (tid < 4) {  // softmax over CLS + n_valid real keys + (padding key x n_pad)
    float* s = sc + tid * SS;
    float mx = s[0];
    for (int i = 1; i <= ntk; ++i) mx = fmaxf(mx, s[i]);
    float sum = 0.f;
    for (int i = 0; i <= ntk; ++i) {
      s[i] = expf(s[i] - mx);
      sum += (i == n_valid + 1) ? s[i] * (float)n_pad : s[i];
    }
    for (int i = 0; i <= ntk; ++i) {
      const float p = s[i] / sum;
      s[i] = (i == n_valid + 1) ? p * (float)n_pad : p;
    }
  }
  __syncthreads();
  float db[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  for (int jj = 0; jj < ntk; ++jj) {
    const int64_t row = jj < n_valid ? tok0 + jj : padrow;
    const float dv = descs[jj * D + tid], av = a4[row * D + tid];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float p = sc[h * SS + 1 + jj];
      db[h] += p * dv;
      ab[h] += p * av;
    }
  }
  float* out = pooled + (int64_t)n * 4 * POOLW;
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    out[h * POOLW + tid] = db[h];
    out[h * POOLW + 256 + tid] = ab[h];
  }
  if (tid < 4 * 32) {
    const int h = tid >> 5, i = tid & 31;
    out[h * POOLW + 512 + i] = i == 0 ? sc[h * SS] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// One-pass form of cls_pool_fused_kernel: ONE WAVE PER SUB-LINE (4 per block), online softmax (running max / sum
// per head, rescaled accumulators) so every token is visited once, nothing is staged in LDS and there is no
// barrier; the bilinear taps and the a4 row of token j+1 are in flight while token j is reduced.
// Same mathematics as cls_pool_kernel up to fp32 rounding (the softmax normalisation is applied at the end).
// ---------------------------------------------------------------------------------------------
#ifndef LT_POOL_WAVES_PER_EU
#define LT_POOL_WAVES_PER_EU 3   // second __launch_bounds__ argument = waves per SIMD the register allocation must allow
#endif
// SPLIT = 4 (a few hundred sub-lines: a single pair): the block's four waves share ONE sub-line, wave w taking tokens w, w + 4, ..;
// the four partial (max, sum, pooled sums) states are merged through LDS, wave h finishing head h.  The dependent chain per wave is a
// quarter as long (21 tokens -> 6): 24 -> 10 us at cfg2.
template <int SPLIT>
__global__ __launch_bounds__(256, LT_POOL_WAVES_PER_EU) void cls_pool_online_kernel(
    const LinetrLineRec* __restrict__ recs, const int* __restrict__ sub2line_g, const float* __restrict__ cpnt,
    const float* __restrict__ a4, int64_t first_pad, int N, int T, const float* __restrict__ nhwc, int Hc, int Wc,
    int align_corners, ClsPoolConst cc, float* __restrict__ pooled /*[N][4][544]*/, int reverse) {
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  // reverse: the LAST sub-lines first -- their a4 rows are the ones the token MLP wrote most recently and are still in the
  // Infinity Cache (a4 of a cfg3 batch is 298 MB, the cache 256 MB: walking forward finds none of it)
  const int n_fwd = SPLIT == 1 ? blockIdx.x * 4 + wv : blockIdx.x;
  if (n_fwd >= N) return;
  const int n = reverse ? N - 1 - n_fwd : n_fwd;
  const LinetrLineRec r = recs[sub2line_g[n]];
  const int j = n - r.first_sub;
  const int n_valid = min(T, r.n_tok - j * T);
  const int n_pad = T - n_valid;
  const int ntk = n_valid + (n_pad > 0 ? 1 : 0);
  const int64_t tok0 = (int64_t)r.first_tok + (int64_t)j * T;
  const int64_t padrow = first_pad + r.image;
  const float* nhwc_img = nhwc + (int64_t)r.image * Hc * Wc * D;
  f32x4 u[4], u2[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    u[h] = *reinterpret_cast<const f32x4*>(cc.U + h * D + lane * 4);
    u2[h] = *reinterpret_cast<const f32x4*>(cc.U2 + h * D + lane * 4);
  }
  // running state per head: max m, sum l, weight of the CLS key w0, pooled sums (this lane's 4 channels)
  float m[4], l[4], w0[4];
  f32x4 db[4], ab[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    m[h] = cc.s_cls[h] * LOG2E; l[h] = 1.f; w0[h] = 1.f;
    if (SPLIT > 1 && wv != 0) { m[h] = -INFINITY; l[h] = 0.f; w0[h] = 0.f; }   // the CLS key belongs to wave 0's share
    db[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    ab[h] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  auto row_of = [&](int jj) -> int64_t { return jj < n_valid ? tok0 + jj : padrow; };
  // The tap geometry of a token is wave-uniform and costs ~150 VALU instructions (two divisions, floor, clamps):
  // lane i computes it once for token chunk + i, and the loop below fetches a token's 4 offsets + 4 weights with
  // 8 v_readlane instead of recomputing them in all 64 lanes.
  int t_off[4];
  float t_w[4];
  auto fill_taps = [&](int chunk) {
    const int jj = chunk + lane;
    const int64_t row = row_of(jj < ntk ? jj : ntk - 1);
    tap_coords(cpnt[row * 2], cpnt[row * 2 + 1], Hc, Wc, align_corners, t_off, t_w);
  };
  auto issue = [&](int jj, TapSet& t, f32x4& a) {
    int off[4];
    float wt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      off[k] = __builtin_amdgcn_readlane(t_off[k], jj & 63);
      wt[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_w[k]), jj & 63));
    }
    taps_load(nhwc_img, off, wt, lane, t);
    a = *reinterpret_cast<const f32x4*>(a4 + row_of(jj) * D + lane * 4);
  };
  TapSet cur, nxt;
  f32x4 a_cur, a_nxt;
  const int jj0 = SPLIT == 1 ? 0 : wv;
  fill_taps(0);
  if (jj0 < ntk) issue(jj0, cur, a_cur);
  for (int jj = jj0; jj < ntk; jj += SPLIT) {
    if (jj + SPLIT < ntk) {  // wave-uniform
      if (((jj + SPLIT) >> 6) != (jj >> 6)) fill_taps((jj + SPLIT) & ~63);
      issue(jj + SPLIT, nxt, a_nxt);
    }
    const f32x4 dv = taps_finish<true>(cur);
    const float mult = jj < n_valid ? 1.f : (float)n_pad;
    float sc[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      float p = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) p += dv[c] * u[h][c] + a_cur[c] * u2[h][c];
      sc[h] = p;
    }
    wave_sum_n<4>(sc);                      // the four head scores reduce together
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      // running softmax in the log2 domain: one v_exp_f32 per exponential (m / l / w0 are only ever used as ratios)
      const float sh = (sc[h] + cc.c_tok[h]) * LOG2E;
      if (__any(sh > m[h])) {   // wave-uniform (sh and m are): the running maximum moves on few tokens only
        const float alpha = __builtin_amdgcn_exp2f(m[h] - sh);
        m[h] = sh;
        l[h] *= alpha;
        w0[h] *= alpha;
#pragma unroll
        for (int c = 0; c < 4; ++c) { db[h][c] *= alpha; ab[h][c] *= alpha; }
      }
      const float e = __builtin_amdgcn_exp2f(sh - m[h]) * mult;
      l[h] += e;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        db[h][c] += e * dv[c];
        ab[h][c] += e * a_cur[c];
      }
    }
    cur = nxt;
    a_cur = a_nxt;
  }
  float* out = pooled + (int64_t)n * 4 * POOLW;
  if constexpr (SPLIT > 1) {
    // merge the four waves' states: wave h finishes head h
    __shared__ float st_ml[4][4][4];                        // [wave][head][m, l, w0, -]
    __shared__ f32x4 st_v[4][4][2][64];                     // [wave][head][d / a][lane]
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if (lane == 0) { st_ml[wv][h][0] = m[h]; st_ml[wv][h][1] = l[h]; st_ml[wv][h][2] = w0[h]; }
      st_v[wv][h][0][lane] = db[h];
      st_v[wv][h][1][lane] = ab[h];
    }
    __syncthreads();
    const int h = wv;
    float M = st_ml[0][h][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) M = fmaxf(M, st_ml[w][h][0]);
    float L = 0.f, W0 = 0.f;
    f32x4 d = {0.f, 0.f, 0.f, 0.f}, a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float sc = __builtin_amdgcn_exp2f(st_ml[w][h][0] - M);     // a wave without tokens has m = -inf: weight 0
      L += st_ml[w][h][1] * sc;
      W0 += st_ml[w][h][2] * sc;
      const f32x4 dv = st_v[w][h][0][lane], av = st_v[w][h][1][lane];
#pragma unroll
      for (int c = 0; c < 4; ++c) { d[c] += dv[c] * sc; a[c] += av[c] * sc; }
    }
    const float inv = 1.f / L;
#pragma unroll
    for (int c = 0; c < 4; ++c) { d[c] *= inv; a[c] *= inv; }
    *reinterpret_cast<f32x4*>(out + h * POOLW + lane * 4) = d;
    *reinterpret_cast<f32x4*>(out + h * POOLW + 256 + lane * 4) = a;
    if (lane < 8) {  // [p_h0, 0 x 31]
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      if (lane == 0) z[0] = W0 * inv;
      *reinterpret_cast<f32x4*>(out + h * POOLW + 512 + lane * 4) = z;
    }
    return;
  }
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const float inv = 1.f / l[h];
    f32x4 d = db[h], a = ab[h];
#pragma unroll
    for (int c = 0; c < 4; ++c) { d[c] *= inv; a[c] *= inv; }
    *reinterpret_cast<f32x4*>(out + h * POOLW + lane * 4) = d;
    *reinterpret_cast<f32x4*>(out + h * POOLW + 256 + lane * 4) = a;
    if (lane < 8) {  // [p_h0, 0 x 31]
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      if (lane == 0) z[0] = w0[h] * inv;
      *reinterpret_cast<f32x4*>(out + h * POOLW + 512 + lane * 4) = z;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Row kernels over [rows][256]; one wave64 per row, lane = 4 channels.
//   mode 0: y = LayerNorm(x) * gamma + beta (+ add)      eps = 1e-6 (models/line_attention.py:40,83)
//   mode 1: y = x / max(||x||_2, 1e-12)                  F.normalize (models/line_transformer.py:246)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_norm_kernel(const float* __restrict__ x, int rows, int mode,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ add, float eps, float* __restrict__ y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  f32x4 v = *reinterpret_cast<const f32x4*>(x + (int64_t)row * D + lane * 4);
  f32x4 o;
  if (mode == 0) {
    float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { float d = v[c] - mean; q += d * d; }
    const float rstd = 1.f / sqrtf(wave_sum(q) * (1.f / D) + eps);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + lane * 4);
    const f32x4 b = *reinterpret_cast<const f32x4*>(beta + lane * 4);
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = (v[c] - mean) * rstd * g[c] + b[c];
    if (add) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(add + (int64_t)row * D + lane * 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] += a[c];
    }
  } else {
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) q += v[c] * v[c];
    const float nrm = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = v[c] / nrm;
  }
  *reinterpret_cast<f32x4*>(y + (int64_t)row * D + lane * 4) = o;
}

// ---------------------------------------------------------------------------------------------
// Neighbour-line ("signature") multi-head attention, one image = one softmax domain
// (models/line_transformer.py:132-154).  Flash-style, fp32 MFMA, no N x N matrix in memory.
//
// grid (image, head, q-tile of 128), block 256 = 4 wave64, wave w owns 32 query rows.
// qkv [N,768] = [q | k | v], each head-major (c = h*64+d; the reference's interleaved c = d*4+h is
// undone by permuting weight rows at load time) and q pre-scaled by 1/8 (exact, power of two).
//
// Per 32-row kv chunk and wave:
//   S^T[kv][q] = K_h[kv,:] . Q_h[q,:]      (A = K tile from LDS, B = Q fragment held in VGPRs)
//   in the 32x32 C/D layout a lane owns ONE query column (q = lane&31) and 16 kv rows, so the online
//   softmax max/sum are in-lane + one exchange with lane^32;
//   the exponentiated S^T registers ARE the B operand of  O^T[d][q] += V^T[d][kv] . P^T[kv][q]
//   under the k-permutation k = 8kk + 4*(lane>>5) + s  <->  register 4kk+s, so P never leaves VGPRs.
// ---------------------------------------------------------------------------------------------
constexpr int ATT_QT = 128;   // query rows per block
constexpr int ATT_KT = 64;    // kv rows staged per iteration
constexpr int ATT_KS = DH + 4;

__global__ __launch_bounds__(256) void sig_attn_kernel(const float* __restrict__ qkv, const int* __restrict__ cu_sub,
                                                       float* __restrict__ out /*[N][256] head-major*/) {
  __shared__ __attribute__((aligned(16))) float Ks[ATT_KT * ATT_KS];
  __shared__ __attribute__((aligned(16))) float Vs[ATT_KT * DH];
  const int img = blockIdx.x, head = blockIdx.y;
  const int n0 = cu_sub[img], Ni = cu_sub[img + 1] - n0;
  const int q0 = blockIdx.z * ATT_QT;
  if (q0 >= Ni) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h2 = lane >> 5, lq = lane & 31;
  const int q = q0 + wave * 32 + lq;
  const bool wave_active = q0 + wave * 32 < Ni;  // wave-uniform
  const float* base = qkv + (int64_t)n0 * 768;

  f32x4 qf[8];
  {
    const int qr = q < Ni ? q : Ni - 1;
    const float* qp = base + (int64_t)qr * 768 + head * DH + h2 * 4;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = *reinterpret_cast<const f32x4*>(qp + kk * 8);
  }
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m = -INFINITY, l = 0.f;

  const int srow = tid >> 4, sc4 = (tid & 15) * 4;  // staging: 16 rows x 16 float4 per pass
  for (int t0 = 0; t0 < Ni; t0 += ATT_KT) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = srow + i * 16;
      const int kv = t0 + r;
      f32x4 kx = {0.f, 0.f, 0.f, 0.f}, vx = {0.f, 0.f, 0.f, 0.f};
      if (kv < Ni) {
        const float* p = base + (int64_t)kv * 768 + head * DH + sc4;
        kx = *reinterpret_cast<const f32x4*>(p + 256);
        vx = *reinterpret_cast<const f32x4*>(p + 512);
      }
      *reinterpret_cast<f32x4*>(&Ks[r * ATT_KS + sc4]) = kx;
      *reinterpret_cast<f32x4*>(&Vs[r * DH + sc4]) = vx;
    }
    __syncthreads();
    if (!wave_active) continue;
#pragma unroll
    for (int c = 0; c < ATT_KT / 32; ++c) {
      const int kv0 = t0 + c * 32;
      if (kv0 >= Ni) break;
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      const float* kp = &Ks[(c * 32 + lq) * ATT_KS + h2 * 4];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(kp + kk * 8);
#pragma unroll
        for (int s = 0; s < 4; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], qf[kk][s], st, 0, 0, 0);
      }
      // online softmax over this lane's 16 kv rows (+ the other half-wave's 16)
      if (kv0 + 32 > Ni) {   // wave-uniform: only the image's last chunk has keys past the end
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kv0 + (r & 3) + 8 * (r >> 2) + 4 * h2 >= Ni) st[r] = -INFINITY;
      }
      float mx = st[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
      mx = xor32_max(mx);
      const float m_new = fmaxf(m, mx);
      const float alpha = expf(m - m_new);  // m = -inf on the first chunk -> 0
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = expf(st[r] - m_new); ps += st[r]; }
      ps = xor32_sum(ps);
      l = l * alpha + ps;
      m = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      // O^T += V^T . P^T
      const float* vp = &Vs[(c * 32 + h2 * 4) * DH + lq];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float v0 = vp[(kk * 8 + s) * DH];
          const float v1 = vp[(kk * 8 + s) * DH + 32];
          o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, st[kk * 4 + s], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, st[kk * 4 + s], o1, 0, 0, 0);
        }
      }
    }
  }
  if (wave_active && q < Ni) {
    const float inv = 1.f / l;
    float* op = out + (int64_t)(n0 + q) * D + head * DH;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = (r & 3) + 8 * (r >> 2) + 4 * h2;
      op[d] = o0[r] * inv;
      op[d + 32] = o1[r] * inv;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Same attention with the two contractions on split-bf16 MFMA (3 planes / 6 products = fp32-faithful, see
// lt_gemm_split.h): v_mfma_f32_32x32x16_bf16 is 16x the rate of the fp32 MFMA, so QK^T + PV cost 48 x 32 cycles
// per 32-row kv chunk instead of 64 x 64.
//   * Q fragments: split once into VGPRs (12 x bf16x8).
//   * K tile in LDS as planes [kv][3][64 d] (row stride 400 B = 4*25 dwords -> conflict-free ds_read_b128).
//   * V tile in LDS row-major as planes [kv][3][64 d] like K (row stride 576 B); the PV B-operand P^T[kv][q] comes
//     straight from the S^T accumulator registers (k-slot e of step t is register 8t+e, i.e.
//     kv = 16t + 4*(lane>>5) + e for e<4 and +8 for e>=4), and the matching A operand V^T[d][kv] -- two runs of four kv
//     for one d -- is fetched with gfx950's transposing LDS read (v_frags_tr), so staging V costs three 8-byte stores per
//     thread instead of twelve 2-byte scatter stores into a transposed image.
//   * softmax stays fp32 and in-lane as in sig_attn_kernel.
// ---------------------------------------------------------------------------------------------
#ifndef LT_ATTN_OCC2
// 1: the 8-wave kernel is capped at 128 VGPRs so that TWO blocks share a CU (4 waves per SIMD) and fetches its K / V rows
// right before staging them -- the second block covers the latency -- instead of carrying a register prefetch.
// Measured at cfg3 (7 launches per step): 0.366 ms with one block per CU and the prefetch, 0.337 ms this way.
#define LT_ATTN_OCC2 1
#endif
constexpr int ATS_RK = 3 * 128 + 16;   // K plane row stride (bytes)
constexpr int ATS_RV = 576;            // V plane row stride (bytes): 144 dwords = 16 (mod 64), see below

// Six transposing LDS reads (gfx950 ds_read_b64_tr_b16) = the V^T fragments of one 16-wide kv step and one 32-wide d
// block: V stays ROW-major
// in LDS ([kv][3 planes][64 d], staged with three 8-byte stores per thread like K) and the hardware hands lane
// (d = lane & 31, half h) the four values V[kv0 + 4h + j][d], j = 0..3 -- exactly the k-slots the P^T accumulator
// registers occupy (tools/ubench/tr_read_probe.hip prints the mapping).  Within a 16-lane group lane i addresses row
// (i >> 2), 4-column chunk (i & 3); lanes 16-31 take the next 16 columns, the upper half-wave starts 4 rows down.
// Bank check: a half-wave reads 4 rows x 64 bytes; with a row stride of 16 dwords (mod 64) the four rows land on four
// disjoint quarters of the 64 banks.  `base` is the lane's byte address for kv0 = 0, plane 0, d block 0.
// out[p][0/1] for d block DT: plane p, kv run KV0 + {0..3} / KV0 + 8 + {0..3}.
template <int KV0, int DT>
__device__ __forceinline__ void v_frags_tr(unsigned base, u32x2 (&o)[3][2]) {
  constexpr int R0 = KV0 * ATS_RV + DT * 64, R1 = (KV0 + 8) * ATS_RV + DT * 64;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %6 offset:%7\n\t"
      "ds_read_b64_tr_b16 %1, %6 offset:%8\n\t"
      "ds_read_b64_tr_b16 %2, %6 offset:%9\n\t"
      "ds_read_b64_tr_b16 %3, %6 offset:%10\n\t"
      "ds_read_b64_tr_b16 %4, %6 offset:%11\n\t"
      "ds_read_b64_tr_b16 %5, %6 offset:%12\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0][0]), "=&v"(o[0][1]), "=&v"(o[1][0]), "=&v"(o[1][1]), "=&v"(o[2][0]), "=&v"(o[2][1])
      : "v"(base), "n"(R0), "n"(R1), "n"(R0 + 128), "n"(R1 + 128), "n"(R0 + 256), "n"(R1 + 256)
      : "memory");
}

// NW waves per block = 32 NW query rows per block.  With 8 waves the K / V tiles of an (image, head) are split and
// staged once for up to 256 queries instead of once per 128.
template <int NW>
__global__ __launch_bounds__(NW * 64, (NW == 8 && LT_ATTN_OCC2) ? 4 : 1) void sig_attn_split_kernel(const float* __restrict__ qkv,
                                                                 const int* __restrict__ cu_sub,
                                                                 float* __restrict__ out /*[N][256] head-major*/) {
  __shared__ __attribute__((aligned(16))) unsigned char Ks[ATT_KT * ATS_RK];
  __shared__ __attribute__((aligned(16))) unsigned char Vs[ATT_KT * ATS_RV];
  const int img = blockIdx.x, head = blockIdx.y;
  const int n0 = cu_sub[img], Ni = cu_sub[img + 1] - n0;
  // the image's queries are dealt out EVENLY over the gridDim.z blocks of its (image, head), in whole waves: 599 queries on
  // three 256-query blocks are 7 + 7 + 5 busy waves instead of 8 + 8 + 3 (every block stages all K / V tiles either way)
  const int per = ((Ni + (int)gridDim.z - 1) / (int)gridDim.z + 31) / 32 * 32;     // <= NW * 32
  const int q0 = blockIdx.z * per;
  if (q0 >= Ni) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h2 = lane >> 5, lq = lane & 31;
  const int q = q0 + wave * 32 + lq;
  const bool wave_active = wave * 32 < per && q0 + wave * 32 < Ni;  // wave-uniform
  const float* base = qkv + (int64_t)n0 * 768;

  bf16x8 qf[4][3];
  {
    const int qr = q < Ni ? q : Ni - 1;
    const float* qp = base + (int64_t)qr * 768 + head * DH + h2 * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 x0 = *reinterpret_cast<const f32x4*>(qp + s * 16);
      f32x4 x1 = *reinterpret_cast<const f32x4*>(qp + s * 16 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { x0[e] *= LOG2E; x1[e] *= LOG2E; }   // scores in log2 units: exp -> v_exp_f32
      unsigned a[3], b[3], c[3], d[3];
      split_pair<3>(x0[0], x0[1], a); split_pair<3>(x0[2], x0[3], b);
      split_pair<3>(x1[0], x1[1], c); split_pair<3>(x1[2], x1[3], d);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        union { bf16x8 v; unsigned w[4]; } u;
        u.w[0] = a[p]; u.w[1] = b[p]; u.w[2] = c[p]; u.w[3] = d[p];
        qf[s][p] = u.v;
      }
    }
  }
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m = -INFINITY, l = 0.f;

  // lane address of the transposing V reads for kv0 = 0 (see v_frags_tr)
  const unsigned v_base = (unsigned)(size_t)(Vs + (((lane & 15) >> 2) + 4 * h2) * ATS_RV + (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);
  const int srow = tid >> 4, sc4 = (tid & 15) * 4;  // staging: 4 NW rows x 16 float4 per pass
  constexpr int NPASS = ATT_KT / (4 * NW);
  // K / V rows of the NEXT tile travel in registers while the current tile is being consumed: the global-load latency
  // (1-2 us per tile, 4 tiles per block at N = 199) is off the critical path
  f32x4 kreg[NPASS], vreg[NPASS];
  auto fetch = [&](int t0) {
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int kv = t0 + srow + i * (4 * NW);
      kreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      vreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kv < Ni) {
        const float* p = base + (int64_t)kv * 768 + head * DH + sc4;
        kreg[i] = *reinterpret_cast<const f32x4*>(p + 256);
        vreg[i] = *reinterpret_cast<const f32x4*>(p + 512);
      }
    }
  };
  if (!LT_ATTN_OCC2) fetch(0);
  for (int t0 = 0; t0 < Ni; t0 += ATT_KT) {
    if (LT_ATTN_OCC2) fetch(t0);     // no register prefetch: the second resident block covers the latency instead
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int r = srow + i * (4 * NW);
      const f32x4 kx = kreg[i], vx = vreg[i];
      unsigned a[3], b[3];
      split_pair<3>(kx[0], kx[1], a);
      split_pair<3>(kx[2], kx[3], b);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(Ks + r * ATS_RK + p * 128 + sc4 * 2) = u32x2{a[p], b[p]};
      split_pair<3>(vx[0], vx[1], a);
      split_pair<3>(vx[2], vx[3], b);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(Vs + r * ATS_RV + p * 128 + sc4 * 2) = u32x2{a[p], b[p]};
    }
    if (!LT_ATTN_OCC2 && t0 + ATT_KT < Ni) fetch(t0 + ATT_KT);   // block-uniform
    __syncthreads();
    if (!wave_active) continue;
#pragma unroll
    for (int c = 0; c < ATT_KT / 32; ++c) {
      const int kv0 = t0 + c * 32;
      if (kv0 >= Ni) break;
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
      const unsigned char* kp = Ks + (c * 32 + lq) * ATS_RK + h2 * 16;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 ka[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) ka[p] = *reinterpret_cast<const bf16x8*>(kp + p * 128 + s * 32);
        // six products, smallest first: (2,0) (1,1) (0,2) (1,0) (0,1) (0,0)
        st = mfma_split<0>(ka[2], qf[s][0], st);
        st = mfma_split<0>(ka[1], qf[s][1], st);
        st = mfma_split<0>(ka[0], qf[s][2], st);
        st = mfma_split<0>(ka[1], qf[s][0], st);
        st = mfma_split<0>(ka[0], qf[s][1], st);
        st = mfma_split<0>(ka[0], qf[s][0], st);
      }
      if (kv0 + 32 > Ni) {   // wave-uniform: only the image's last chunk has keys past the end
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kv0 + (r & 3) + 8 * (r >> 2) + 4 * h2 >= Ni) st[r] = -INFINITY;
      }
      float mx = st[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
      mx = xor32_max(mx);
      const float m_new = fmaxf(m, mx);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r] - m_new); ps += st[r]; }
      ps = xor32_sum(ps);
      if (__any(m_new != m)) {      // wave-uniform: once the running max has settled the accumulators need no rescale
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);   // m = -inf on the first chunk -> 0
        l *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      }
      l += ps;
      m = m_new;
      // P^T planes straight from the accumulator registers
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8 pp[3];
        {
          unsigned w[4][3];
#pragma unroll
          for (int e = 0; e < 4; ++e) split_pair<3>(st[8 * t + 2 * e], st[8 * t + 2 * e + 1], w[e]);
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            union { bf16x8 v; unsigned u[4]; } x;
            x.u[0] = w[0][p]; x.u[1] = w[1][p]; x.u[2] = w[2][p]; x.u[3] = w[3][p];
            pp[p] = x.v;
          }
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          u32x2 vr[3][2];
          if (dt == 0) {
            if (c == 0) { if (t == 0) v_frags_tr<0, 0>(v_base, vr); else v_frags_tr<16, 0>(v_base, vr); }
            else        { if (t == 0) v_frags_tr<32, 0>(v_base, vr); else v_frags_tr<48, 0>(v_base, vr); }
          } else {
            if (c == 0) { if (t == 0) v_frags_tr<0, 1>(v_base, vr); else v_frags_tr<16, 1>(v_base, vr); }
            else        { if (t == 0) v_frags_tr<32, 1>(v_base, vr); else v_frags_tr<48, 1>(v_base, vr); }
          }
          bf16x8 va[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            union { bf16x8 v; unsigned u[4]; } x;
            x.u[0] = vr[p][0][0]; x.u[1] = vr[p][0][1]; x.u[2] = vr[p][1][0]; x.u[3] = vr[p][1][1];
            va[p] = x.v;
          }
          f32x16& o = dt == 0 ? o0 : o1;
          o = mfma_split<0>(va[2], pp[0], o);
          o = mfma_split<0>(va[1], pp[1], o);
          o = mfma_split<0>(va[0], pp[2], o);
          o = mfma_split<0>(va[1], pp[0], o);
          o = mfma_split<0>(va[0], pp[1], o);
          o = mfma_split<0>(va[0], pp[0], o);
        }
      }
    }
  }
  if (wave_active) {
    // O^T: a lane owns ONE query and 4-runs of d (d = 8 (r >> 2) + 4 h2 + (r & 3)).  One v_permlane32_swap per register
    // pair hands each half-wave the other half's 4-run, so a lane ends up with 8 consecutive d and stores them as two
    // dwordx4 (r02 stored 32 single dwords per lane: 32 rows x 4 B per instruction, a store-issue-bound tail).
    const float inv = 1.f / l;
    float* op = out + (int64_t)(n0 + (q < Ni ? q : Ni - 1)) * D + head * DH + 8 * h2;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = (dt == 0 ? o0[r] : o1[r]) * inv;
#pragma unroll
      for (int c = 0; c < 4; ++c) {        // (v[c], v[4 + c]) and (v[8 + c], v[12 + c]): upper half of the first <-> lower half of the second
        float a0 = v[c], b0 = v[4 + c], a1 = v[8 + c], b1 = v[12 + c];
        halves_swap(a0, b0);
        halves_swap(a1, b1);
        v[c] = a0; v[4 + c] = b0; v[8 + c] = a1; v[12 + c] = b1;
      }
      if (q < Ni) {
        // after the swaps v[0..7] = d 32 dt + 8 h2 .. + 8 and v[8..15] = d 32 dt + 16 + 8 h2 .. + 8
        *reinterpret_cast<f32x4*>(op + dt * 32) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(op + dt * 32 + 4) = f32x4{v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(op + dt * 32 + 16) = f32x4{v[8], v[9], v[10], v[11]};
        *reinterpret_cast<f32x4*>(op + dt * 32 + 20) = f32x4{v[12], v[13], v[14], v[15]};
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Latency form of sig_attn_split_kernel for a few images (single pair): the KV range is split across the block's 4 waves
// as well.  grid (image, head, 32-query tile), 256 threads; wave w takes the 32-row KV chunks w, w+4, ... of its
// (image, head), stages them in a WAVE-PRIVATE LDS region (no block barrier in the loop), runs the same split-bf16
// QK^T / online softmax / PV as above, and the four partial (m, l, O) are merged through LDS at the end (exact: the
// softmax is rescaled to the common maximum).  A single 2 x 199-line pair launches 56 blocks of 4 waves whose critical
// path is 2 chunks, instead of 8 (or 32) blocks walking 7 chunks each.
// ---------------------------------------------------------------------------------------------
constexpr int ATL_RK = 3 * 128 + 16;   // K plane row stride (bytes): [kv][3][64 d]
constexpr int ATL_RV = 3 * 64 + 8;     // V^T plane row stride (bytes): [d][3][32 kv]
constexpr int ATL_WAVE_BYTES = 32 * ATL_RK + DH * ATL_RV;   // 12 800 + 12 800

__global__ __launch_bounds__(256) void sig_attn_small_kernel(const float* __restrict__ qkv, const int* __restrict__ cu_sub,
                                                             float* __restrict__ out /*[N][256] head-major*/,
                                                             int ldq /*row stride of qkv in floats: 768, or 1024 when q/k/v sit behind x_out*/) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * ATL_WAVE_BYTES];
  const int img = blockIdx.x, head = blockIdx.y;
  const int n0 = cu_sub[img], Ni = cu_sub[img + 1] - n0;
  const int q0 = blockIdx.z * 32;
  if (q0 >= Ni) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h2 = lane >> 5, lq = lane & 31;
  const int q = q0 + lq;
  const float* base = qkv + (int64_t)n0 * ldq;
  unsigned char* Ks = lds + wave * ATL_WAVE_BYTES;
  unsigned char* Vt = Ks + 32 * ATL_RK;

  bf16x8 qf[4][3];
  {
    const int qr = q < Ni ? q : Ni - 1;
    const float* qp = base + (int64_t)qr * ldq + head * DH + h2 * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 x0 = *reinterpret_cast<const f32x4*>(qp + s * 16);
      f32x4 x1 = *reinterpret_cast<const f32x4*>(qp + s * 16 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { x0[e] *= LOG2E; x1[e] *= LOG2E; }
      unsigned a[3], b[3], c[3], d[3];
      split_pair<3>(x0[0], x0[1], a); split_pair<3>(x0[2], x0[3], b);
      split_pair<3>(x1[0], x1[1], c); split_pair<3>(x1[2], x1[3], d);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        union { bf16x8 v; unsigned w[4]; } u;
        u.w[0] = a[p]; u.w[1] = b[p]; u.w[2] = c[p]; u.w[3] = d[p];
        qf[s][p] = u.v;
      }
    }
  }
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m = -INFINITY, l = 0.f;

  // staging of one 32-row chunk by one wave: 16 lanes x float4 per row, 4 rows per pass, 8 passes
  const int srow = lane >> 4, sc4 = (lane & 15) * 4;
  f32x4 kreg[8], vreg[8];
  auto fetch = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kv = kv0 + srow + 4 * i;
      kreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      vreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kv < Ni) {
        const float* p = base + (int64_t)kv * ldq + head * DH + sc4;
        kreg[i] = *reinterpret_cast<const f32x4*>(p + 256);
        vreg[i] = *reinterpret_cast<const f32x4*>(p + 512);
      }
    }
  };
  if (wave * 32 < Ni) fetch(wave * 32);
  for (int kv0 = wave * 32; kv0 < Ni; kv0 += 128) {       // wave-uniform trip count
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = srow + 4 * i;
      unsigned a[3], b[3];
      split_pair<3>(kreg[i][0], kreg[i][1], a);
      split_pair<3>(kreg[i][2], kreg[i][3], b);
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(Ks + r * ATL_RK + p * 128 + sc4 * 2) = u32x2{a[p], b[p]};
      split_pair<3>(vreg[i][0], vreg[i][1], a);
      split_pair<3>(vreg[i][2], vreg[i][3], b);
#pragma unroll
      for (int p = 0; p < 3; ++p) {  // transposed: element (kv=r, d=sc4+j) -> Vt[d][p][r]
        unsigned short* col = reinterpret_cast<unsigned short*>(Vt + p * 64 + r * 2);
        col[(sc4 + 0) * (ATL_RV / 2)] = (unsigned short)(a[p] & 0xffffu);
        col[(sc4 + 1) * (ATL_RV / 2)] = (unsigned short)(a[p] >> 16);
        col[(sc4 + 2) * (ATL_RV / 2)] = (unsigned short)(b[p] & 0xffffu);
        col[(sc4 + 3) * (ATL_RV / 2)] = (unsigned short)(b[p] >> 16);
      }
    }
    if (kv0 + 128 < Ni) fetch(kv0 + 128);
    __builtin_amdgcn_wave_barrier();     // compiler-only: wave-private LDS, executed in order
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    const unsigned char* kp = Ks + lq * ATL_RK + h2 * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 ka[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) ka[p] = *reinterpret_cast<const bf16x8*>(kp + p * 128 + s * 32);
      st = mfma_split<0>(ka[2], qf[s][0], st);
      st = mfma_split<0>(ka[1], qf[s][1], st);
      st = mfma_split<0>(ka[0], qf[s][2], st);
      st = mfma_split<0>(ka[1], qf[s][0], st);
      st = mfma_split<0>(ka[0], qf[s][1], st);
      st = mfma_split<0>(ka[0], qf[s][0], st);
    }
    if (kv0 + 32 > Ni) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kv0 + (r & 3) + 8 * (r >> 2) + 4 * h2 >= Ni) st[r] = -INFINITY;
    }
    float mx = st[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
    mx = xor32_max(mx);
    const float m_new = fmaxf(m, mx);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r] - m_new); ps += st[r]; }
    ps = xor32_sum(ps);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);   // m = -inf on the first chunk -> 0
    l = l * alpha + ps;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    m = m_new;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16x8 pp[3];
      {
        unsigned w[4][3];
#pragma unroll
        for (int e = 0; e < 4; ++e) split_pair<3>(st[8 * t + 2 * e], st[8 * t + 2 * e + 1], w[e]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          union { bf16x8 v; unsigned u[4]; } x;
          x.u[0] = w[0][p]; x.u[1] = w[1][p]; x.u[2] = w[2][p]; x.u[3] = w[3][p];
          pp[p] = x.v;
        }
      }
      const unsigned char* vp = Vt + lq * ATL_RV + (16 * t + 4 * h2) * 2;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        bf16x8 va[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const u32x2 lo = *reinterpret_cast<const u32x2*>(vp + dt * 32 * ATL_RV + p * 64);
          const u32x2 hi = *reinterpret_cast<const u32x2*>(vp + dt * 32 * ATL_RV + p * 64 + 16);
          union { bf16x8 v; unsigned u[4]; } x;
          x.u[0] = lo[0]; x.u[1] = lo[1]; x.u[2] = hi[0]; x.u[3] = hi[1];
          va[p] = x.v;
        }
        f32x16& o = dt == 0 ? o0 : o1;
        o = mfma_split<0>(va[2], pp[0], o);
        o = mfma_split<0>(va[1], pp[1], o);
        o = mfma_split<0>(va[0], pp[2], o);
        o = mfma_split<0>(va[1], pp[0], o);
        o = mfma_split<0>(va[0], pp[1], o);
        o = mfma_split<0>(va[0], pp[0], o);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // ---- merge the four partial results: O = sum_w 2^(m_w - M) O_w / sum_w 2^(m_w - M) l_w
  __syncthreads();                                   // all staging regions are dead
  float* Op = reinterpret_cast<float*>(lds);         // [4][64 d][33]  (q padded to 33)
  float* ML = Op + 4 * 64 * 33;                      // [4][2][32]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = (r & 3) + 8 * (r >> 2) + 4 * h2;
    Op[(wave * 64 + d) * 33 + lq] = o0[r];
    Op[(wave * 64 + d + 32) * 33 + lq] = o1[r];
  }
  if (h2 == 0) { ML[(wave * 2 + 0) * 32 + lq] = m; ML[(wave * 2 + 1) * 32 + lq] = l; }
  __syncthreads();
  const int oq = tid >> 3, od = (tid & 7) * 8;       // thread -> (query, 8 consecutive d)
  if (q0 + oq < Ni) {
    float mw[4], M = -INFINITY, L = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { mw[w] = ML[(w * 2) * 32 + oq]; M = fmaxf(M, mw[w]); }
    float sc[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      sc[w] = mw[w] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw[w] - M);   // a wave without any chunk contributes nothing
      L += sc[w] * ML[(w * 2 + 1) * 32 + oq];
    }
    const float inv = 1.f / L;
    float res[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) a += sc[w] * Op[(w * 64 + od + e) * 33 + oq];
      res[e] = a * inv;
    }
    float* op = out + (int64_t)(n0 + q0 + oq) * D + head * DH + od;
    *reinterpret_cast<f32x4*>(op) = f32x4{res[0], res[1], res[2], res[3]};
    *reinterpret_cast<f32x4*>(op + 4) = f32x4{res[4], res[5], res[6], res[7]};
  }
}

}  // namespace lt
