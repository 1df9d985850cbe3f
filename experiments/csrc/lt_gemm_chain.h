// Row-tile-local GEMM chains in ONE launch (bf16x6, the pipelined 128 x 256 tile of lt_gemm_split.h).
//
// Most of the descriptor network's Linear layers only mix channels: row tile i of layer n+1 needs row tile i of layer n
// and nothing else (models/line_transformer.py:157-183: W1 -> ReLU -> W2 + residual; the next layer's q/k/v projection;
// models/line_attention.py:36-40, 79-83: fc + LayerNorm -> w_1 -> GELU -> w_2 + LayerNorm).  Launched one by one, every
// such GEMM pays its own launch, first-tile fetch, output burst and partly empty last round -- ~19 us per launch that do
// not depend on K, of a 35-90 us kernel.  Here a block owns ONE 128-row tile and walks the whole chain
// for it: stage after stage, column tile after column tile, through the same software-pipelined main loop and the same
// LDS epilogue (bias / ReLU / GELU / residual / two-source A / fused LayerNorm or L2 norm).  A stage's output goes to
// global memory with plain stores (it stays in the XCD's L2) and is read back by the same block in the next stage after a
// block barrier: no inter-block synchronisation exists or is needed.
#pragma once
#include "lt_gemm_split.h"

namespace lt {

constexpr int CHAIN_MAX = 4;
struct ChainArgs {
  SplitGemmArgs st[CHAIN_MAX];
  int n = 0;
};

__global__ __launch_bounds__(512) void gemm_chain_kernel(ChainArgs c) {
  constexpr int BM = 128, BN = 256, WN = 4, PL = 3, NT = 512;
  constexpr int TM = 64, TN = 64, MI = 2, NI = 2;
  constexpr int RS = PL * 64 + 16;
  constexpr int A_F4 = BM * 8 / NT;                // 2
  constexpr int B_PCS = BN * PL * 4 / NT;          // 6
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_c[];
  unsigned char* As = smem_c;                      // [2][BM][RS]
  unsigned char* Bs = smem_c + 2 * BM * RS;        // [2][BN][RS]
  const int M = c.st[0].g.M;
  const int gy = (M + BM - 1) / BM;
  int mt;
  {  // XCD-aware order of the row tiles (neighbouring row tiles share nothing, but it keeps the launch shape familiar)
    const int b = blockIdx.x, q = gy / 8, r = gy % 8, xcd = b % 8, k = b / 8;
    mt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const int m0 = mt * BM;
  constexpr int N_TERM = 6, N_MMA = MI * NI * N_TERM, N_FRAG = (MI + NI) * PL, E1 = 2 * A_F4 + B_PCS;
  constexpr int TPA[6] = {2, 1, 0, 1, 0, 0};
  constexpr int TPB[6] = {0, 1, 2, 0, 1, 0};

#pragma unroll 1
  for (int s = 0; s < c.n; ++s) {
    const SplitGemmArgs& sa = c.st[s];
    const GemmArgs& g = sa.g;
    const int K1 = g.A2 ? g.K1 : g.K;
    const int nkw = g.K / 32, nk = nkw;
    // the operands the main loop touches, pinned in SGPRs: read through `g` the compiler re-loads them from the kernel
    // argument segment inside the K loop (s_load + lgkmcnt waits in the middle of the LDS pipeline)
    typedef const __attribute__((address_space(1))) char* gptr_t;       // explicitly global: a pointer that went through an asm
    uint64_t uA = (uint64_t)g.A, uA2 = (uint64_t)(g.A2 ? g.A2 : g.A), uW = (uint64_t)sa.Wsp;   // statement would be FLAT otherwise
    int s_lda = g.lda, s_lda2 = g.A2 ? g.lda2 : g.lda;
    asm volatile("" : "+s"(uA), "+s"(uA2), "+s"(uW), "+s"(s_lda), "+s"(s_lda2));
    const gptr_t sA = (gptr_t)uA, sA2 = (gptr_t)uA2, sW = (gptr_t)uW;
#pragma unroll 1
    for (int n0 = 0; n0 < g.N; n0 += BN) {
      // Everything derived from the thread index is re-derived from an OPAQUE copy inside the tile loop: otherwise LICM
      // hoists dozens of per-thread addresses out of the two loops and keeps them live across the whole main loop
      // (234 spilled VGPRs next to the 256-register pipeline; 9 without the loops).
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      const int wm = wave / WN, wn = wave % WN;
      const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
      const int frow = lane & 31, fk = (lane >> 5) * 16;
      f32x4 ra_[2][A_F4];
      f32x4 rb_[2][B_PCS];
      f32x16 acc[MI][NI];
      // 32-bit byte offsets against wave-uniform bases (saddr + voffset loads): half the address registers of 64-bit
      // per-lane pointers.  The launcher checks that every operand is below 4 GiB.
      unsigned a_row4[A_F4], w_off[B_PCS];
#pragma unroll
      for (int u = 0; u < A_F4; ++u) {
        int r = m0 + lrow + u * (NT / 8);
        r = r < g.M ? r : g.M - 1;
        a_row4[u] = (unsigned)r * 4u;
      }
#pragma unroll
      for (int u = 0; u < B_PCS; ++u) {
        const int pq = tid + u * NT;
        const int r = pq / (PL * 4), pc = pq % (PL * 4);
        w_off[u] = (unsigned)(n0 + r) * (unsigned)nkw * (PL * 64) + pc * 16;
      }
      auto step_gload = [&](int u, int kt, f32x4 (&ra)[A_F4], f32x4 (&rb)[B_PCS]) {
        if (u < A_F4) {
          const int k0 = kt * 32;
          const bool first = k0 < K1;                                  // wave-uniform: selects, no branch around the load
          const gptr_t base = first ? sA + k0 * 4 : sA2 + (k0 - K1) * 4;
          const unsigned ld = first ? (unsigned)s_lda : (unsigned)s_lda2;
          ra[u] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(base + (a_row4[u] * ld + lc4 * 4u));
        } else {
          rb[u - A_F4] = *reinterpret_cast<const __attribute__((address_space(1))) f32x4*>(sW + (size_t)kt * (PL * 64) + w_off[u - A_F4]);
        }
      };
      auto step_store = [&](int u, int buf, const f32x4 (&ra)[A_F4], const f32x4 (&rb)[B_PCS]) {
        if (u < A_F4) {
          unsigned a[PL], b[PL];
          split_pair<PL, 0>(ra[u][0], ra[u][1], a);
          split_pair<PL, 0>(ra[u][2], ra[u][3], b);
          unsigned char* dst = As + (buf * BM + lrow + u * (NT / 8)) * RS + lc4 * 2;
#pragma unroll
          for (int pp = 0; pp < PL; ++pp) *reinterpret_cast<u32x2*>(dst + pp * 64) = u32x2{a[pp], b[pp]};
        } else {
          const int pq = tid + (u - A_F4) * NT;
          const int r = pq / (PL * 4), pc = pq % (PL * 4);
          *reinterpret_cast<f32x4*>(Bs + (buf * BN + r) * RS + pc * 16) = rb[u - A_F4];
        }
      };
      auto step_read = [&](int k, int buf, int st, bf16x8 (&af)[MI][PL], bf16x8 (&bf)[NI][PL]) {
        const unsigned char* Ab = As + (buf * BM + wm * TM + frow) * RS + fk + st * 32;
        const unsigned char* Bb = Bs + (buf * BN + wn * TN + frow) * RS + fk + st * 32;
        if (k < MI * PL) { const int i = k / PL, pp = k % PL; af[i][pp] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * RS + pp * 64); }
        else { const int q = k - MI * PL, j = q / PL, pp = q % PL; bf[j][pp] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * RS + pp * 64); }
      };
      bf16x8 af0[MI][PL], bf0[NI][PL], af1[MI][PL], bf1[NI][PL];
      auto step_mma = [&](int m, const bf16x8 (&af)[MI][PL], const bf16x8 (&bf)[NI][PL]) {
        const int t = m / (MI * NI), ij = m % (MI * NI), i = ij / NI, j = ij % NI;
        acc[i][j] = mfma_split<0>(af[i][TPA[t]], bf[j][TPB[t]], acc[i][j]);
      };
      // fragment read order = order of first use by the MFMAs (lt_gemm_split.h)
#define LT_CH_FRAG(k) (((k) / (MI + NI)) == 0 ? (((k) % (MI + NI)) < MI ? ((k) % (MI + NI)) * PL + (PL - 1) : MI * PL + (((k) % (MI + NI)) - MI) * PL) \
                     : ((k) / (MI + NI)) == PL - 1 ? (((k) % (MI + NI)) < MI ? ((k) % (MI + NI)) * PL : MI * PL + (((k) % (MI + NI)) - MI) * PL + (PL - 1)) \
                     : (((k) % (MI + NI)) < MI ? ((k) % (MI + NI)) * PL + 1 : MI * PL + (((k) % (MI + NI)) - MI) * PL + 1))
      auto first_half = [&](int kt, f32x4 (&ra)[A_F4], f32x4 (&rb)[B_PCS]) {
        const int buf = kt & 1;
        const int ktn = kt + 3 < nk ? kt + 3 : nk - 1;
#pragma unroll
        for (int m = 0; m < N_MMA; ++m) {
          step_mma(m, af0, bf0);
#pragma unroll
          for (int k = 0; k < N_FRAG; ++k)
            if (k * N_MMA / N_FRAG == m) step_read(LT_CH_FRAG(k), buf, 1, af1, bf1);
#pragma unroll
          for (int e = 0; e < E1; ++e)
            if ((2 * e + 1) * N_MMA / (2 * E1) == m) {
              if (e < 2 * A_F4) { if (e % 2 == 0) step_store(e / 2, buf ^ 1, ra, rb); else step_gload(e / 2, ktn, ra, rb); }
              else step_store(A_F4 + (e - 2 * A_F4), buf ^ 1, ra, rb);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto second_half = [&](int kt, f32x4 (&ra)[A_F4], f32x4 (&rb)[B_PCS]) {
        const int buf = kt & 1;
        const int ktn = kt + 3 < nk ? kt + 3 : nk - 1;
#pragma unroll
        for (int m = 0; m < N_MMA; ++m) {
          step_mma(m, af1, bf1);
#pragma unroll
          for (int k = 0; k < N_FRAG; ++k)
            if (k * N_MMA / N_FRAG == m) step_read(LT_CH_FRAG(k), buf ^ 1, 0, af0, bf0);
#pragma unroll
          for (int e = 0; e < B_PCS; ++e)
            if ((2 * e + 1) * N_MMA / (2 * B_PCS) == m) step_gload(A_F4 + e, ktn, ra, rb);
          __builtin_amdgcn_sched_barrier(0);
        }
      };

      // ---- main loop (two register sets of prefetch, barrier in the middle of a K tile; see lt_gemm_split.h)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
      for (int u = 0; u < A_F4 + B_PCS; ++u) step_gload(u, 0, ra_[0], rb_[0]);
#pragma unroll
      for (int u = 0; u < A_F4 + B_PCS; ++u) step_store(u, 0, ra_[0], rb_[0]);
      if (nk > 1) {
#pragma unroll
        for (int u = 0; u < A_F4 + B_PCS; ++u) step_gload(u, 1, ra_[1], rb_[1]);
      }
#pragma unroll
      for (int u = 0; u < A_F4 + B_PCS; ++u) step_gload(u, nk > 2 ? 2 : nk - 1, ra_[0], rb_[0]);
      __syncthreads();
#pragma unroll
      for (int k = 0; k < N_FRAG; ++k) step_read(LT_CH_FRAG(k), 0, 0, af0, bf0);
      for (int kt = 0; kt < nk; kt += 2) {
        first_half(kt, ra_[1], rb_[1]);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        second_half(kt, ra_[1], rb_[1]);
        if (kt + 1 < nk) {
          first_half(kt + 1, ra_[0], rb_[0]);
          __syncthreads();
          __builtin_amdgcn_sched_barrier(0);
          second_half(kt + 1, ra_[0], rb_[0]);
        }
      }
#undef LT_CH_FRAG

      // ---- epilogue through LDS (lt_gemm_split.h: every wave parks its 64 x 64 accumulator tile in its own region)
      constexpr int LPR = TN / 4, RPI = 64 / LPR, NIT = TM / RPI;
      __syncthreads();
      float* ep = reinterpret_cast<float*>(smem_c) + wave * (TM * TN);
      const int er = lane / LPR, ec = (lane % LPR) * 4;
      const int grow0 = m0 + wm * TM + er, gcol = n0 + wn * TN + ec;
      const bool row_norm = g.norm != 0;           // dispatcher guarantees N == 256 for such a stage
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const float bv = g.bias ? g.bias[n0 + wn * TN + j * 32 + (lane & 31)] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[i][j][r] + bv;
            if (g.act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (g.act == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
            ep[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TN + j * 32 + (lane & 31)] = v;
          }
        }
      }
      if (row_norm) {
        // LayerNorm / L2 normalisation of whole rows, same arithmetic and order as row_norm_kernel
        __syncthreads();
        constexpr int RPW = BM / 8;
        const float* epb = reinterpret_cast<const float*>(smem_c);
        const int c0 = lane * 4, wn_c = c0 / TN, cc = c0 % TN;
        for (int it = 0; it < RPW; ++it) {
          const int rl = wave * RPW + it;
          const int row = m0 + rl;
          if (row >= g.M) break;                        // wave-uniform
          f32x4 v = *reinterpret_cast<const f32x4*>(epb + ((rl / TM) * WN + wn_c) * (TM * TN) + (rl % TM) * TN + cc);
          if (g.R) {
            const f32x4 rr = *reinterpret_cast<const f32x4*>(g.R + (int64_t)row * g.ldr + c0);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += rr[q];
          }
          f32x4 o;
          if (g.norm == 1) {
            const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256);
            float qq = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float d = v[q] - mean; qq += d * d; }
            const float rstd = 1.f / sqrtf(wave_sum(qq) * (1.f / 256) + g.eps);
            const f32x4 ga = *reinterpret_cast<const f32x4*>(g.gamma + c0);
            const f32x4 be = *reinterpret_cast<const f32x4*>(g.beta + c0);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (v[q] - mean) * rstd * ga[q] + be[q];
          } else {
            float qq = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) qq += v[q] * v[q];
            const float nrm = fmaxf(sqrtf(wave_sum(qq)), 1e-12f);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = v[q] / nrm;
          }
          if (g.add2) {
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(g.add2 + (int64_t)row * g.ldadd2 + c0);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] += a2[q];
          }
          *reinterpret_cast<f32x4*>(g.Y + (int64_t)row * g.ldy + c0) = o;
        }
      } else {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int row = grow0 + it * RPI;
          f32x4 v = *reinterpret_cast<const f32x4*>(ep + (er + it * RPI) * TN + ec);
          if (row < g.M) {
            if (g.R) {
              const f32x4 rr = *reinterpret_cast<const f32x4*>(g.R + (int64_t)row * g.ldr + gcol);
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] += rr[q];
            }
            *reinterpret_cast<f32x4*>(g.Y + (int64_t)row * g.ldy + gcol) = v;
          }
        }
      }
      // the epilogue regions alias the tile buffers of the next main loop, and the next stage reads what was just stored
      __syncthreads();
    }
  }
}

// every stage: bf16x6 weights (sa.Wsp set), N % 256 == 0, K % 32 == 0, the same M, 16-byte aligned rows; a stage with a
// fused row normalisation has N == 256
inline int gemm_chain_launch(const ChainArgs& c, hipStream_t st) {
  if (c.n < 1 || c.n > CHAIN_MAX) return fail(LINETR_E_ARG, "gemm_chain: bad stage count %d", c.n);
  const int M = c.st[0].g.M;
  if (M <= 0) return 0;
  for (int s = 0; s < c.n; ++s) {
    const GemmArgs& g = c.st[s].g;
    if (g.M != M || g.N % 256 != 0 || g.K % 32 != 0 || g.K < 64 || (g.A2 && g.K1 % 32 != 0) || g.lda % 4 != 0 || g.ldy % 4 != 0 ||
        (g.A2 && g.lda2 % 4 != 0) || (g.R && g.ldr % 4 != 0) || (g.norm && g.N != 256) || !c.st[s].Wsp ||
        (int64_t)M * std::max(g.lda, g.lda2) * 4 >= ((int64_t)1 << 32) || (int64_t)g.N * g.K * 6 >= ((int64_t)1 << 32))
      return fail(LINETR_E_ARG, "gemm_chain: unsupported stage %d (M=%d N=%d K=%d)", s, g.M, g.N, g.K);
  }
  constexpr int lds = 2 * (128 + 256) * (3 * 64 + 16);    // two tile buffers (156 KB); the epilogue image (128 KB) aliases them
  static unsigned long long attr_done = 0;
  const unsigned long long dev_bit = current_device_bit();
  if (!(attr_done & dev_bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_done |= dev_bit;
  }
  hipLaunchKernelGGL(gemm_chain_kernel, dim3(cdiv(M, 128)), dim3(512), lds, st, c);
  LT_LAUNCH_CHECK();
  return 0;
}

}  // namespace lt
