// Split-tile ("ST") GEMM for gfx950 (EXPERIMENTS build only; measured and not shipped, HISTORY.md r03): both operands live in HBM
// ALREADY split into three bf16 planes and already in the image the LDS wants (lt_st_image.h), so a K step travels HBM/L2 -> LDS by
// LDS-DMA (global_load_lds_dwordx4) with no VGPR staging, no split VALU and no ds_write in the main loop, and the epilogue writes its
// result in the same format for the next GEMM.
//
// The product is computed TRANSPOSED, D[n][m] = sum_k W[n][k] X[m][k] (weights = MFMA A operand, activations = B
// operand): in the 32x32 C/D layout a lane then owns ONE activation row m and 4-runs of consecutive output features n,
// so one v_permlane32_swap per register pair turns the accumulators into 8-wide pieces of the output's ST image (or
// 32-byte runs of an fp32 row): the epilogue needs no LDS, no barrier, and stores 256-byte runs.
#pragma once
#include "lt_st_image.h"

namespace lt {

struct StGemmArgs {
  const unsigned char* A1 = nullptr; int nk1 = 0;    // ST activations [M][16 nk1]
  const unsigned char* A2 = nullptr; int nk2 = 0;    // optional second source, concatenated along K
  const unsigned char* W = nullptr;                  // ST weights [N][16 (nk1 + nk2)]
  const float* bias = nullptr;                       // [N], never null (the handle keeps a zero vector)
  const unsigned char* R = nullptr;                  // ST residual [M][N] or null (added after the activation)
  unsigned char* Yst = nullptr;                      // ST output [M][N] ...
  float* Y = nullptr; int ldy = 0;                   // ... or fp32 rows
  int M = 0, N = 0, act = 0;
};

// Block tile 256 n x 128 m, 8 waves = 4 (n) x 2 (m), wave tile 64 n x 64 m = 2 x 2 MFMA tiles, 6 products each.
// LDS: a ring of FOUR 36 KiB slots, one 16-wide K step each (24 KiB weight panel + 12 KiB activation panel), step k in
// slot k & 3.  Phase p multiplies the fragments of K step p (in registers) while it fetches the fragments of step p+1
// from the ring and issues the DMA of step p+4 into step p's slot (free: every wave fetched step p's fragments before
// the previous barrier).  The barrier at the end of the phase publishes step p+2: every wave first waits for its OWN
// DMA instructions of that step with a COUNTED vmcnt, so steps p+3 and p+4 stay in flight across the barrier.
// A DMA therefore has three phases (~2 us) to land and ~100 KiB per CU are in flight all the time: measured, the
// two-stage / one-tile-ahead form of this loop was bound by DMA latency (tools/ubench/st_gemm_bench.hip: DMA alone took
// as long as the MFMAs alone).
constexpr int STG_BN = 256, STG_BM = 128;
constexpr int STG_SLOT = (STG_BN + STG_BM) / 16 * ST_RB;     // 36 864
constexpr int STG_W_BYTES = STG_BN / 16 * ST_RB;             // 24 576

// DBG (tools/ubench/st_gemm_bench.hip only): 1 no MFMAs, 2 no DMA inside the loop, 4 no fragment reads inside the loop,
// 8 no epilogue
template <int DBG = 0>
__global__ __launch_bounds__(512) void gemm_st_kernel(StGemmArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char st_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wm = wave >> 2;
  const int gx = a.N / STG_BN, gy = (a.M + STG_BM - 1) / STG_BM;
  int tile;
  {
    const int ntile = gx * gy, b = blockIdx.x, q = ntile / 8, r = ntile % 8, xcd = b % 8, k = b / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m0 = (tile / gx) * STG_BM, n0 = (tile % gx) * STG_BN;
  const int64_t RBa = st_row_blocks(a.M), RBw = a.N / 16;
  const int nk = a.nk1 + a.nk2;

  // ---- LDS-DMA.  The panel of K step k is two contiguous spans (24 KiB of weights, 12 KiB of activations) = 36
  // instructions of 1 KiB; wave w issues instructions w, w+8, w+16, w+24 and, for w < 4, 32+w: x < 24 weights, else
  // activations.  LDS offset inside a slot = x * 1024 either way.  All addresses are wave-uniform + lane * 16.
  constexpr int NI_MAX = 5;
  const int n_dma = wave < 4 ? 5 : 4;
  const unsigned lane16 = lane * 16;
  const unsigned char* g1[NI_MAX];   // instruction d of K step k reads g1[d] + k * gstep[d] for k < nk1 ...
  const unsigned char* g2[NI_MAX];   // ... and g2[d] + k * gstep[d] from there on (second source; weights: the same pointer)
  int64_t gstep[NI_MAX];
#pragma unroll
  for (int d = 0; d < NI_MAX; ++d) {
    const int x = wave + 8 * d;
    if (x < 24) {
      gstep[d] = RBw * ST_RB;
      g1[d] = a.W + (int64_t)(n0 / 16) * ST_RB + x * 1024;
      g2[d] = g1[d];
    } else {
      gstep[d] = RBa * ST_RB;
      g1[d] = a.A1 + (int64_t)(m0 / 16) * ST_RB + (x - 24) * 1024;
      g2[d] = a.A2 ? a.A2 + (int64_t)(m0 / 16) * ST_RB + (x - 24) * 1024 - a.nk1 * gstep[d] : g1[d];
    }
  }
  auto issue_one = [&](int d, int k, int slot) {               // d-th instruction of this wave for K step k
    const unsigned char* g = (k < a.nk1 ? g1[d] : g2[d]) + k * gstep[d];
    LT_GLDS(g + lane16, st_smem + slot * STG_SLOT + (wave + 8 * d) * 1024, 0);
  };
  auto issue_step = [&](int k, int slot) {
#pragma unroll
    for (int d = 0; d < NI_MAX; ++d)
      if (d < n_dma) issue_one(d, k, slot);
  };
  // counted waits: at most `steps` K steps of this wave's DMA still in flight
  auto wait_dma = [&](int steps) {
    if (steps <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (wave < 4) {
      if (steps == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else if (steps == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    } else {
      if (steps == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (steps == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
  };

  // ---- fragments: one ds_read_b128 each; f = 0..11 in the order the MFMAs first use them
  const int lfrag = ((lane >> 4) & 1) * ST_RB + (lane >> 5) * 256 + (lane & 15) * 16;
  const unsigned char* wbase = st_smem + wn * 4 * ST_RB + lfrag;
  const unsigned char* abase = st_smem + STG_W_BYTES + wm * 4 * ST_RB + lfrag;
  // term t multiplies weight plane TW[t] with activation plane TA[t]; smallest cross terms first
  constexpr int TW[6] = {2, 1, 0, 1, 0, 0}, TA[6] = {0, 1, 2, 0, 1, 0};
  auto read_one = [&](int f, int slotoff, bf16x8 (&wf)[2][3], bf16x8 (&af)[2][3]) {
    const int grp = f / 4, w = f % 4;          // group 0: planes (W 2, A 0), group 1: (1, 1), group 2: (0, 2)
    const int i = w >> 1;
    if ((w & 1) == 0) {
      const int p = 2 - grp;
      wf[i][p] = *reinterpret_cast<const bf16x8*>(wbase + slotoff + i * 2 * ST_RB + p * ST_CHUNK);
    } else {
      const int p = grp;
      af[i][p] = *reinterpret_cast<const bf16x8*>(abase + slotoff + i * 2 * ST_RB + p * ST_CHUNK);
    }
  };
  f32x16 acc[2][2];
  auto mma_one = [&](int m, const bf16x8 (&wf)[2][3], const bf16x8 (&af)[2][3]) {
    const int t = m / 4, i = (m >> 1) & 1, j = m & 1;
    if (DBG & 1) return;
    acc[i][j] = mfma_split<0>(wf[i][TW[t]], af[j][TA[t]], acc[i][j]);
  };

  // ---- prologue: K steps 0..3 into slots 0..3 (a problem with fewer steps fetches its last one again)
#pragma unroll
  for (int k = 0; k < 4; ++k) issue_step(k < nk ? k : nk - 1, k);
  // accumulators start at the bias (scalar loads: n depends on the lane only through its half)
  {
    const int h = lane >> 5;
    const float* bp = a.bias + n0 + wn * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float lo = bp[i * 32 + 8 * (r >> 2) + (r & 3)], hi = bp[i * 32 + 8 * (r >> 2) + 4 + (r & 3)];
        const float v = h ? hi : lo;
        acc[i][0][r] = v;
        acc[i][1][r] = v;
      }
  }
  wait_dma(3);                                   // step 0 landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  bf16x8 wfA[2][3], afA[2][3], wfB[2][3], afB[2][3];
#pragma unroll
  for (int f = 0; f < 12; ++f) read_one(f, 0, wfA, afA);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wait_dma(2);                                   // step 1 landed
  __builtin_amdgcn_s_barrier();                  // ... for everyone, and slot 0 is free
  asm volatile("" ::: "memory");

  // Phase p, fixed issue order (sched_barrier after every MFMA slot; a slot carries at most one ds_read_b128 or one DMA
  // instruction behind its MFMA).  `later` = this wave's K steps allowed in flight across the barrier (2 in steady state).
  auto phase = [&](int p, auto dma_flag, int later, bf16x8 (&wc)[2][3], bf16x8 (&ac)[2][3], bf16x8 (&wx)[2][3],
                   bf16x8 (&ax)[2][3]) {
    constexpr bool DMA = decltype(dma_flag)::value;
    const int rd_off = ((p + 1) & 3) * STG_SLOT;        // slot of step p+1
    const int wr_slot = p & 3;                          // slot of step p <- step p+4
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      mma_one(m, wc, ac);
      if (m < 12) { if (!(DBG & 4)) read_one(m, rd_off, wx, ax); }
      else if (DMA && m - 12 < NI_MAX && !(DBG & 2)) { if (m - 12 < n_dma) issue_one(m - 12, p + 4, wr_slot); }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_dma(later);                                    // step p+2 has landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  int p = 0;
  for (; p + 5 < nk; p += 2) {
    phase(p, std::true_type{}, 2, wfA, afA, wfB, afB);
    phase(p + 1, std::true_type{}, 2, wfB, afB, wfA, afA);
  }
  for (; p < nk; p += 2) {
    // steps after p+2 that exist (and were issued): they may stay in flight
    const int l0 = nk - 3 - p < 0 ? 0 : nk - 3 - p > 2 ? 2 : nk - 3 - p;
    if (p + 4 < nk) phase(p, std::true_type{}, l0, wfA, afA, wfB, afB);
    else phase(p, std::false_type{}, l0, wfA, afA, wfB, afB);
    const int l1 = nk - 4 - p < 0 ? 0 : nk - 4 - p > 2 ? 2 : nk - 4 - p;
    if (p + 5 < nk) phase(p + 1, std::true_type{}, l1, wfB, afB, wfA, afA);
    else phase(p + 1, std::false_type{}, l1, wfB, afB, wfA, afA);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- epilogue: activation, half-swap into 8-wide pieces, residual, split, store
  if (DBG & 8) {
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
    if (sacc == 123.456f) a.Y[tid] = sacc;
    return;
  }
  const int h = lane >> 5, ml = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wm * 64 + j * 32 + ml;
    const int rb = m >> 4, r16 = m & 15;
    const bool live = rb < RBa;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = a.act == ACT_RELU ? fmaxf(acc[i][j][r], 0.f) : acc[i][j][r];
#pragma unroll
      for (int c = 0; c < 4; ++c) { swap32(v[c], v[4 + c]); swap32(v[8 + c], v[12 + c]); }
      const int kto = (n0 + wn * 64 + i * 32) >> 4;       // first of the two 16-wide output K steps of this 32-wide n tile
#pragma unroll
      for (int g = 0; g < 2; ++g) {       // piece (K step kto + g, q = h): values v[8 g .. 8 g + 7] = n 16 g + 8 h .. + 7
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = v[8 * g + e];
        const int64_t off = ((int64_t)(kto + g) * RBa + rb) * ST_RB + h * 256 + r16 * 16;
        if (a.R && live) {
          float rr[8];
          st_piece_to_f32(*reinterpret_cast<const u32x4*>(a.R + off), *reinterpret_cast<const u32x4*>(a.R + off + ST_CHUNK),
                          *reinterpret_cast<const u32x4*>(a.R + off + 2 * ST_CHUNK), rr);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += rr[e];
        }
        if (a.Yst) {
          unsigned s0[3], s1[3], s2[3], s3[3];
          split_pair<3>(x[0], x[1], s0); split_pair<3>(x[2], x[3], s1);
          split_pair<3>(x[4], x[5], s2); split_pair<3>(x[6], x[7], s3);
          if (live) {
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
              *reinterpret_cast<u32x4*>(a.Yst + off + pp * ST_CHUNK) = u32x4{s0[pp], s1[pp], s2[pp], s3[pp]};
          }
        } else if (m < a.M) {
          float* yp = a.Y + (int64_t)m * a.ldy + n0 + wn * 64 + i * 32 + (2 * g + h) * 8;
          *reinterpret_cast<f32x4*>(yp) = f32x4{x[0], x[1], x[2], x[3]};
          *reinterpret_cast<f32x4*>(yp + 4) = f32x4{x[4], x[5], x[6], x[7]};
        }
      }
    }
  }
}

inline int gemm_st_launch(const StGemmArgs& a, hipStream_t st) {
  if (a.M <= 0) return 0;
  if (a.N % STG_BN != 0 || a.nk1 < 1 || (a.A2 && a.nk2 < 1) || (!a.A2 && a.nk2 != 0) || (a.nk1 + a.nk2) % 2 || !a.bias ||
      (!a.Yst && (!a.Y || a.ldy % 4)))
    return fail(LINETR_E_ARG, "gemm_st: unsupported shape M=%d N=%d nk=%d+%d", a.M, a.N, a.nk1, a.nk2);
  constexpr int lds = 4 * STG_SLOT;
  static unsigned long long attr_done = 0;
  const unsigned long long dev_bit = current_device_bit();
  if (!(attr_done & dev_bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_st_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done |= dev_bit;
  }
  const dim3 grid((a.N / STG_BN) * cdiv(a.M, STG_BM));
  hipLaunchKernelGGL(gemm_st_kernel<0>, grid, dim3(512), lds, st, a);
  LT_LAUNCH_CHECK();
  return 0;
}

}  // namespace lt
