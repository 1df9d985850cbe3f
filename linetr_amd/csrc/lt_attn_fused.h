// q/k/v projection + signature attention of one (image, head) in ONE kernel (models/line_transformer.py:132-154: the three
// Conv1d projections, the per-head softmax attention; the merge conv is folded into the next GEMM).
//
// Why: as separate launches the projection is a K = 256 GEMM that pays the per-tile prologue / epilogue of a 128 x 256 or
// 128 x 128 tile for 16 K steps of work (69 us for 27 us of MFMA time at cfg3) and writes 78 MB of q/k/v that the attention
// reads straight back.  Here a block of eight waves owns one (image, head):
//   1. projection, computed TRANSPOSED (weights = MFMA A operand): wave w owns rows 32 w .. 32 w + 31 of the image and all
//      192 output channels of the head (q | k | v: six 32 x 32 accumulator tiles).  A K step's operands -- the weight panel
//      (192 rows x 16 k x 3 planes = 18 KiB, pre-split in the split-tile image of lt_st_image.h) and the image's activations
//      (256 rows x 16 fp32 = 16 KiB, gathered row by row) -- travel by LDS-DMA into a ring of four slots with counted vmcnt
//      waits, exactly like the ST GEMM; a lane reads the 8 fp32 of ITS OWN row back and splits them into the three bf16
//      planes on the fly (no ordinary global load in the loop, no staging registers).
//   2. in the C/D layout of the transposed product a lane owns ONE row and 4-runs of channels; one half-wave swap per
//      register pair gives 8 consecutive channels, which IS the B-operand layout of S^T = K Q^T for Q (stays in VGPRs)
//      and a 16-byte row piece of the K / V images of sig_attn_split_kernel for K and V (written to LDS in two halves of
//      128 keys; the weight ring is dead by then).
//   3. attention exactly as sig_attn_split_kernel (lt_model.h): S^T by split-bf16 MFMA, in-lane softmax, P^T from the
//      accumulator registers, V^T fragments by transposing LDS reads; the message leaves as fp32 rows.
// q, k, v never reach HBM; one launch per layer instead of two.  Images of up to 256 sub-lines (eight waves).
#pragma once
#include "lt_st_image.h"
#include "lt_model.h"

namespace lt {

constexpr int FQA_W_BYTES = 3 * 4 * ST_RB;              // 18 432 B: (q | k | v) x 4 row blocks x 3 planes x 512
constexpr int FQA_SLOT = FQA_W_BYTES + 256 * 64;        // + 256 rows x 16 fp32 of activations = 34 816 B
constexpr int FQA_KEYS = 128;                           // keys per LDS half
constexpr int FQA_LDS = 4 * FQA_SLOT;                   // 139 264 B; the K / V half images (124 928 B) alias the dead ring

// The six transposing reads of v_frags_tr (lt_model.h) WITHOUT the wait: issued a (t, d block) ahead of their MFMAs.  hipcc does
// not count an asm load, so the consumer side is fr_wait: one lgkmcnt(0) that names every destination (cdna_hip_programming.md 5.7).
template <int KV0, int DT>
__device__ __forceinline__ void fr_issue(unsigned base, u32x2 (&o)[3][2]) {
  constexpr int R0 = KV0 * ATS_RV + DT * 64, R1 = (KV0 + 8) * ATS_RV + DT * 64;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %6 offset:%7\n\t"
      "ds_read_b64_tr_b16 %1, %6 offset:%8\n\t"
      "ds_read_b64_tr_b16 %2, %6 offset:%9\n\t"
      "ds_read_b64_tr_b16 %3, %6 offset:%10\n\t"
      "ds_read_b64_tr_b16 %4, %6 offset:%11\n\t"
      "ds_read_b64_tr_b16 %5, %6 offset:%12"
      : "=&v"(o[0][0]), "=&v"(o[0][1]), "=&v"(o[1][0]), "=&v"(o[1][1]), "=&v"(o[2][0]), "=&v"(o[2][1])
      : "v"(base), "n"(R0), "n"(R1), "n"(R0 + 128), "n"(R1 + 128), "n"(R0 + 256), "n"(R1 + 256)
      : "memory");
}
__device__ __forceinline__ void fr_wait(u32x2 (&o)[3][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[0][0]), "+v"(o[0][1]), "+v"(o[1][0]), "+v"(o[1][1]), "+v"(o[2][0]), "+v"(o[2][1]) : : "memory");
}

__global__ __launch_bounds__(512) void sig_qkv_attn_kernel(const float* __restrict__ z /*[N][256]*/,
                                                           const unsigned char* __restrict__ Wst /*ST image of Wqkv [768][256]*/,
                                                           const float* __restrict__ bqkv /*[768]*/,
                                                           const int* __restrict__ cu_sub, float* __restrict__ out /*[N][256]*/) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char fq_smem[];
  const int img = blockIdx.x, head = blockIdx.y;     // (an XCD-grouped 1-D order -- the four heads of an image on one L2 -- measured level)
  const int n0 = cu_sub[img], Ni = cu_sub[img + 1] - n0;
  if (Ni <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h2 = lane >> 5, lq = lane & 31;
  const int q = wave * 32 + lq;                          // row of the image this lane owns (query AND key)
  const bool wave_active = wave * 32 < Ni;               // wave-uniform
  constexpr int RBW = 3 * D / 16, NK = D / 16;           // 48 row blocks of the weight image, 16 K steps

  // ---- 1. projection --------------------------------------------------------------------------------------------
  // Ring slot of a K step (34 KiB): [weights: (q | k | v) x 4 row blocks x 3 planes x 512 B = 18 KiB][activations: 256 rows x 16
  // fp32 = 16 KiB, row-major].  DMA plan: 34 instructions of 1 KiB; x < 18: weight span x / 6, linear; x >= 18: activation
  // rows 16 (x - 18) .. + 15, a lane fetching 16 bytes of row lane >> 2 (gather: the source address is per lane, rows past the
  // end of the image repeat its last row).  Wave w issues x = w, w + 8, w + 16, w + 24 and, for w < 2, w + 32.
  // (The activations go through the ring, as fp32, rather than through ordinary loads: a lane's half row would be 128 VGPRs
  // next to 96 accumulators, and ordinary loads in flight beside LDS-DMA make hipcc drain the whole queue at their use.)
  const int n_dma = wave < 2 ? 5 : 4;
  const unsigned lane16 = lane * 16;
  auto issue_one = [&](int d, int k, int slot) {
    const int x = wave + 8 * d;
    unsigned char* dst = fq_smem + slot * FQA_SLOT + x * 1024;
    if (x < 18) {
      const unsigned char* g = Wst + ((int64_t)k * RBW + (x / 6) * 16 + head * 4) * ST_RB + (x % 6) * 1024;
      LT_GLDS(g + lane16, dst, 0);
    } else {
      int row = (x - 18) * 16 + (lane >> 2);
      row = row < Ni ? row : Ni - 1;
      // LDS-DMA places lane l's 16 bytes at dst + 16 l: slot (lane & 3) of row (lane >> 2).  The four 16-byte quarters of a row are
      // ROTATED by (row >> 2) & 3 inside its 64 bytes (this lane fetches quarter (slot - rotation) & 3): the 16 lanes of a ds_read_b128
      // group (lanes {0-3, 12-15, 20-27} ...: four runs of 4 rows whose row >> 2 differ mod 4), all reading the same quarter, then hit
      // 16 distinct 16-byte slots of the 64-bank row ((row & 3) x 16 + 4 x slot) instead of four (r06: these reads were 55 % of the
      // kernel's bank-conflict cycles; the rest are the V staging stores).  Four adjacent lanes still fetch 64 contiguous bytes of one row.
      const float* g = z + (int64_t)(n0 + row) * D + k * 16 + (((lane & 3) - ((lane >> 4) & 3)) & 3) * 4;
      LT_GLDS(g, dst, 0);
    }
  };
  auto issue_step = [&](int k, int slot) {
#pragma unroll
    for (int d = 0; d < 5; ++d)
      if (d < n_dma) issue_one(d, k, slot);
  };
  auto wait_dma = [&](int steps) {                         // at most `steps` K steps of this wave's DMA in flight
    if (steps <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (wave < 2) {
      if (steps == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else {
      if (steps == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
  };
#pragma unroll
  for (int k = 0; k < 4; ++k) issue_step(k, k);
  // accumulators start at the bias: in the transposed C/D layout register 4 b + c of tile i is channel 32 (i & 1) + 8 b + 4 h2 + c
  // of q / k / v, so a group of four registers is one dwordx4 of the bias vector.  (Ordinary loads beside the DMA: hipcc
  // drains the queue at their use, which here only means waiting for the prologue's four steps a little early.)
  f32x16 acc[6];
  {
    const float* bp = bqkv + head * DH + 4 * h2;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int bq = 0; bq < 4; ++bq) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bp + (i >> 1) * D + (i & 1) * 32 + 8 * bq);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[i][4 * bq + c] = v[c];
      }
  }
  wait_dma(2);                                             // steps 0 and 1 landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int lfrag = ((lane >> 4) & 1) * ST_RB + (lane >> 5) * 256 + (lane & 15) * 16;
  // this lane's two quarters (8 fp32 = k 8 h2 .. + 8 of its row) at their rotated slots (issue_one)
  const int zrot = (lq >> 2) & 3;
  const int zoff0 = FQA_W_BYTES + (wave * 32 + lq) * 64 + ((2 * h2 + zrot) & 3) * 16;
  const int zoff1 = FQA_W_BYTES + (wave * 32 + lq) * 64 + ((2 * h2 + 1 + zrot) & 3) * 16;
  constexpr int TW[6] = {2, 1, 0, 1, 0, 0}, TA[6] = {0, 1, 2, 0, 1, 0};     // smallest cross terms first
  // Fixed issue order (sched_barrier after every MFMA slot, as in the GEMMs): while n-tile i of step s multiplies, the weight
  // fragments of n-tile i+1 are fetched; the activations of step s+1 are fetched at the start of step s and split into their
  // planes under its MFMAs; during the last n-tile the first weight fragments of step s+1 arrive.  Steps s and s+1 are both
  // visible while step s runs (the barrier at the end of step s-1 published step s+1), the DMA of step s+3 is issued from
  // inside step s into the slot step s-1 left.
  auto read_w = [&](int k, int i, int p, bf16x8 (&wf)[3]) {
    wf[p] = *reinterpret_cast<const bf16x8*>(fq_smem + (k & 3) * FQA_SLOT + lfrag + ((i >> 1) * 4 + (i & 1) * 2) * ST_RB + p * ST_CHUNK);
  };
  auto split_z = [&](const f32x4& x0, const f32x4& x1, bf16x8 (&zf)[3]) {
    unsigned a[3], b[3], c[3], d[3];
    split_pair<3>(x0[0], x0[1], a); split_pair<3>(x0[2], x0[3], b);
    split_pair<3>(x1[0], x1[1], c); split_pair<3>(x1[2], x1[3], d);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      union { bf16x8 v; unsigned w[4]; } u;
      u.w[0] = a[p]; u.w[1] = b[p]; u.w[2] = c[p]; u.w[3] = d[p];
      zf[p] = u.v;
    }
  };
  bf16x8 zfA[3], zfB[3], wfA[3], wfB[3];
  {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(fq_smem + zoff0), x1 = *reinterpret_cast<const f32x4*>(fq_smem + zoff1);
    split_z(x0, x1, zfA);
#pragma unroll
    for (int p = 0; p < 3; ++p) read_w(0, 0, p, wfA);
  }
  auto step = [&](int s, bf16x8 (&zc)[3], bf16x8 (&zn)[3]) {
    f32x4 zr0, zr1;
    const bool more = s + 1 < NK;
#pragma unroll
    for (int m = 0; m < 36; ++m) {
      const int i = m / 6, t = m % 6;
      bf16x8 (&wc)[3] = (i & 1) ? wfB : wfA;
      bf16x8 (&wn)[3] = (i & 1) ? wfA : wfB;
      acc[i] = mfma_split<0>(wc[TW[t]], zc[TA[t]], acc[i]);
      if (t < 3) {                                         // next weight fragments: n-tile i+1 of this step, or n-tile 0 of the next
        if (i < 5) read_w(s, i + 1, t, wn);
        else read_w(s + 1, 0, t, wn);                      // (past the last step: reads a slot that is never used; harmless)
      }
      if (m == 3) zr0 = *reinterpret_cast<const f32x4*>(fq_smem + ((s + 1) & 3) * FQA_SLOT + zoff0);
      if (m == 4) zr1 = *reinterpret_cast<const f32x4*>(fq_smem + ((s + 1) & 3) * FQA_SLOT + zoff1);
      if (m == 15) split_z(zr0, zr1, zn);
      if (m >= 18 && m < 23 && s >= 1 && s + 3 < NK) { if (m - 18 < n_dma) issue_one(m - 18, s + 3, (s + 3) & 3); }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (more) {
      wait_dma(s + 3 < NK ? 1 : 0);                        // step s+2 has landed; step s+3 may stay in flight
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll 1
  for (int s = 0; s < NK; s += 2) {
    step(s, zfA, zfB);         // six n-tiles: the weight fragments alternate A, B, A, B, A, B and hand A to the next step
    step(s + 1, zfB, zfA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- 2. Q to B fragments (registers), K / V pieces (registers until their half is staged) ----------------------------
  // piece g of accumulator tile i = channels 32 (i & 1) + 16 g + 8 h2 .. + 8 of this lane's row, after the half-wave swap
  auto pieces = [&](const f32x16& a, float scale, float (&v)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = a[r] * scale;
#pragma unroll
    for (int c = 0; c < 4; ++c) { halves_swap(v[c], v[4 + c]); halves_swap(v[8 + c], v[12 + c]); }
  };
  auto split8 = [&](const float* x, bf16x8 (&o)[3]) {
    unsigned a[3], b[3], c[3], d[3];
    split_pair<3>(x[0], x[1], a); split_pair<3>(x[2], x[3], b);
    split_pair<3>(x[4], x[5], c); split_pair<3>(x[6], x[7], d);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      union { bf16x8 v; unsigned w[4]; } u;
      u.w[0] = a[p]; u.w[1] = b[p]; u.w[2] = c[p]; u.w[3] = d[p];
      o[p] = u.v;
    }
  };
  bf16x8 qf[4][3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float v[16];
    pieces(acc[i], LOG2E, v);                              // scores in log2 units: exp -> v_exp_f32 (q is pre-scaled by 1/8)
    split8(v, qf[2 * i]);
    split8(v + 8, qf[2 * i + 1]);
  }
  unsigned char* Ks = fq_smem;                             // [128][ATS_RK]
  unsigned char* Vs = fq_smem + FQA_KEYS * ATS_RK;         // [128][ATS_RV]
  auto stage_kv = [&]() {                                  // this wave's 32 keys into the current half
    const int r = (wave & 3) * 32 + lq;
#pragma unroll
    for (int i = 2; i < 6; ++i) {
      float v[16];
      pieces(acc[i], 1.f, v);
      unsigned char* dst = (i < 4 ? Ks + r * ATS_RK : Vs + r * ATS_RV) + ((i & 1) * 32 + 8 * h2) * 2;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        bf16x8 pl[3];
        split8(v + 8 * g, pl);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(dst + p * 128 + g * 32) = pl[p];
      }
    }
  };
  // ---- 3. attention over two halves of 128 keys ----------------------------------------------------------------------
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const unsigned v_base = (unsigned)(size_t)(Vs + (((lane & 15) >> 2) + 4 * h2) * ATS_RV + (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2);
  auto chunk = [&](int kv0, auto kv_local) {               // 32 keys kv0 .. kv0 + 31 of the image; kv_local = their row in the half
    constexpr int KL = decltype(kv_local)::value;
    constexpr int KB = KL >= 64 ? 64 : 0;                  // the immediate offset of a ds_read is 16 bits: keys 64.. via the base
    const unsigned vb = v_base + KB * ATS_RV;
    // V^T fragments of (t = 0, d block 0) are on their way before the scores are even computed
    u32x2 vrA[3][2], vrB[3][2];
    fr_issue<KL - KB, 0>(vb, vrA);
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    const unsigned char* kp = Ks + (KL + lq) * ATS_RK + h2 * 16;
    bf16x8 kaA[3], kaB[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) kaA[p] = *reinterpret_cast<const bf16x8*>(kp + p * 128);
    // S^T = K Q^T, fixed issue order: the K fragments of channel step s+1 arrive under the six products of step s
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      const int sq = m / 6, t = m % 6;
      bf16x8 (&kc)[3] = (sq & 1) ? kaB : kaA;
      bf16x8 (&kn)[3] = (sq & 1) ? kaA : kaB;
      constexpr int PK[6] = {2, 1, 0, 1, 0, 0}, PQ[6] = {0, 1, 2, 0, 1, 0};
      st = mfma_split<0>(kc[PK[t]], qf[sq][PQ[t]], st);
      if (t < 3 && sq < 3) kn[t] = *reinterpret_cast<const bf16x8*>(kp + t * 128 + (sq + 1) * 32);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kv0 + 32 > Ni) {                                   // wave-uniform: only the image's last chunk has keys past the end
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
    if (__any(m_new != m)) {
      const float alpha = __builtin_amdgcn_exp2f(m - m_new);
      l *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }
    l += ps;
    m = m_new;
    // O^T += V^T P^T: four (t, d block) groups of six products; the reads of a group are issued one group ahead
    bf16x8 pp0[3], pp1[3];
    {
      float sv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) sv[e] = st[e];
      split8(sv, pp0);
    }
    auto pv = [&](u32x2 (&vr)[3][2], const bf16x8 (&pp)[3], f32x16& o) {
      bf16x8 va[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        union { bf16x8 v; unsigned u[4]; } x;
        x.u[0] = vr[p][0][0]; x.u[1] = vr[p][0][1]; x.u[2] = vr[p][1][0]; x.u[3] = vr[p][1][1];
        va[p] = x.v;
      }
      o = mfma_split<0>(va[2], pp[0], o);
      o = mfma_split<0>(va[1], pp[1], o);
      o = mfma_split<0>(va[0], pp[2], o);
      o = mfma_split<0>(va[1], pp[0], o);
      o = mfma_split<0>(va[0], pp[1], o);
      o = mfma_split<0>(va[0], pp[0], o);
    };
    fr_wait(vrA);
    fr_issue<KL - KB, 1>(vb, vrB);
    pv(vrA, pp0, o0);
    __builtin_amdgcn_sched_barrier(0);
    {
      float sv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) sv[e] = st[8 + e];
      split8(sv, pp1);
    }
    fr_wait(vrB);
    fr_issue<KL - KB + 16, 0>(vb, vrA);
    pv(vrB, pp0, o1);
    __builtin_amdgcn_sched_barrier(0);
    fr_wait(vrA);
    fr_issue<KL - KB + 16, 1>(vb, vrB);
    pv(vrA, pp1, o0);
    __builtin_amdgcn_sched_barrier(0);
    fr_wait(vrB);
    pv(vrB, pp1, o1);
  };
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half * FQA_KEYS >= Ni) break;                      // block-uniform
    __syncthreads();                                       // the ring / the previous half is dead
    if ((wave >> 2) == half && wave_active) stage_kv();
    __syncthreads();
    if (!wave_active) continue;
    const int k0 = half * FQA_KEYS;
    if (k0 < Ni) chunk(k0, std::integral_constant<int, 0>{});
    if (k0 + 32 < Ni) chunk(k0 + 32, std::integral_constant<int, 32>{});
    if (k0 + 64 < Ni) chunk(k0 + 64, std::integral_constant<int, 64>{});
    if (k0 + 96 < Ni) chunk(k0 + 96, std::integral_constant<int, 96>{});
  }

  // ---- 4. epilogue: O^T / l -> 8 consecutive d per lane -> dwordx4 stores ------------------------------------------------
  if (wave_active) {
    const float inv = 1.f / l;
    float* op = out + (int64_t)(n0 + (q < Ni ? q : Ni - 1)) * D + head * DH + 8 * h2;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      float v[16];
      pieces(dt == 0 ? o0 : o1, inv, v);
      if (q < Ni) {
        *reinterpret_cast<f32x4*>(op + dt * 32) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(op + dt * 32 + 4) = f32x4{v[4], v[5], v[6], v[7]};
        *reinterpret_cast<f32x4*>(op + dt * 32 + 16) = f32x4{v[8], v[9], v[10], v[11]};
        *reinterpret_cast<f32x4*>(op + dt * 32 + 20) = f32x4{v[12], v[13], v[14], v[15]};
      }
    }
  }
}

}  // namespace lt
