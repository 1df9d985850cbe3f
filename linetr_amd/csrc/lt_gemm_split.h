// Split-bf16 MFMA GEMM for gfx950:  Y[M,N] = epi(A[M,K] * W[N,K]^T + bias), fp32 in / fp32 out.
//
// fp32-input MFMA runs at the fp32 VECTOR rate on CDNA4 (157 TF, 1/16 of the bf16 MFMA rate).  This
// kernel instead decomposes every fp32 operand into PL bf16 planes (x = p0 + p1 (+ p2), each plane
// the round-to-nearest bf16 of the running remainder) and accumulates the significant cross products
// with v_mfma_f32_32x32x16_bf16 in fp32:
//     PL = 2  "bf16x3":  a0b0 + a0b1 + a1b0                     rel. error/product ~1e-5  (5.3x the fp32-MFMA ceiling)
//     PL = 3  "bf16x6":  a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0  ~2^-23, i.e. fp32-faithful (2.7x the ceiling)
// bf16 x bf16 products are exact in fp32 and the MFMA accumulates in fp32, so the only error is the
// dropped low-order cross terms (smallest terms are accumulated first).
//
// Layout: weights are pre-split once ([N][K/32][PL][32] bf16, same bytes/element as fp32 for PL=2);
// activations stay fp32 in HBM and are split by the loader on their way into LDS (v_cvt_pk_bf16_f32,
// 3 VALU ops / element, hidden under the MFMAs).  LDS rows hold the PL planes of a 32-wide K chunk
// back to back + 16 B pad: row stride 36 (PL=2) / 52 (PL=3) dwords = 4*odd, so ds_read_b128 of the
// 8-element MFMA fragments is bank-conflict-free (same argument as lt_gemm.h).
// The STORES into that image are conflict-free by lane assignment (r06; the r03-r05 counter passes showed 18 % of the LDS-active
// cycles in bank conflicts, all of it from the two store streams -- `stage_row` / `stage_piece` below).
#pragma once
#include "lt_gemm.h"

namespace lt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// FMT 0: bf16 planes (fp32 exponent range, 8-bit significand each)
// FMT 1: fp16 planes (11-bit significand each: 2 planes carry 22 bits -> 3 products give ~2^-22 per product, but
//        operands must stay below 65504 in magnitude; small remainders fall into fp16 subnormals, which only costs
//        absolute accuracy below 3e-8)
template <int PL, int FMT = 0>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned (&out)[PL]) {
  float r0 = x0, r1 = x1;
#pragma unroll
  for (int p = 0; p < PL; ++p) {
    if constexpr (FMT == 0) {
      const bf16x2 h = __builtin_convertvector(f32x2{r0, r1}, bf16x2);  // v_cvt_pk_bf16_f32 (RNE)
      const unsigned u = __builtin_bit_cast(unsigned, h);
      out[p] = u;
      if (p + 1 < PL) {
        r0 -= __builtin_bit_cast(float, u << 16);
        r1 -= __builtin_bit_cast(float, u & 0xffff0000u);
      }
    } else {
      const f16x2 h = __builtin_convertvector(f32x2{r0, r1}, f16x2);    // v_cvt_pk_f16_f32 (RNE)
      out[p] = __builtin_bit_cast(unsigned, h);
      if (p + 1 < PL) {
        const f32x2 hf = __builtin_convertvector(h, f32x2);
        r0 -= hf[0];
        r1 -= hf[1];
      }
    }
  }
}

template <int FMT>
__device__ __forceinline__ f32x16 mfma_split(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  if constexpr (FMT == 0) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// fp32 [rows][K] -> split planes [rows][K/32][PL][32] bf16.  One thread per 4 consecutive k.
template <int PL, int FMT = 0>
__global__ void split_rows_kernel(const float* __restrict__ W, unsigned char* __restrict__ out, int64_t rows, int K) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int kq = K / 4;
  if (gid >= rows * kq) return;
  const int64_t r = gid / kq;
  const int k = (int)(gid % kq) * 4;
  const f32x4 v = *reinterpret_cast<const f32x4*>(W + r * K + k);
  unsigned a[PL], b[PL];
  split_pair<PL, FMT>(v[0], v[1], a);
  split_pair<PL, FMT>(v[2], v[3], b);
  unsigned char* base = out + (r * (K / 32) + k / 32) * (PL * 64) + (k % 32) * 2;
#pragma unroll
  for (int p = 0; p < PL; ++p) *reinterpret_cast<u32x2*>(base + p * 64) = u32x2{a[p], b[p]};
}

struct SplitGemmArgs {
  GemmArgs g;                 // g.W unused
  const unsigned char* Wsp;   // split weights
  int64_t gWsp;               // bytes between groups
  int wide_epi = 0;           // epilogue through LDS with dwordx4 row stores (set by gemm_split_launch_t)
  // stream-K tail (128x256 pipelined kernel only; see gemm_split_kernel): tiles >= sk_first are shared by sk_blocks
  // blocks that each take an equal run of (tile, K-tile) iterations
  int sk_first = 0, sk_blocks = 0;
  float* sk_ws = nullptr;           // [sk_blocks][BM * BN] partial accumulator tiles
  unsigned* sk_flags = nullptr;     // [sk_blocks] = epoch once the block's partial tile is published; [sk_blocks] = error flag
  unsigned sk_epoch = 0;
};

// ---- stream-K hand-off (MI355X_MICROARCH.md, "publish-large" / "Valid forms"): the partial tile is written with
// write-through (sc1) 16-byte stores and read back with sc1 loads, the flag is a relaxed agent-scope atomic behind a
// drained vmcnt.  No release / acquire fences: an agent release writes back the XCD's whole L2 -- which is full of the
// output rows the data-parallel blocks of the same launch have just stored -- once per publishing block.
__device__ __forceinline__ void sk_store16(float* p, const f32x4& v) {
  // the trailing s_nop: hipcc does not pad an asm store, and its next instruction may otherwise overwrite the data
  // registers before the store has read them (cdna_hip_programming.md 5.7)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 sk_load16(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void sk_publish(unsigned* flag, unsigned epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's write-through stores have reached memory
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sk_wait(const unsigned* flag, unsigned epoch, unsigned* err) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > 400000) {      // ~0.1 s: never hang the device; the error flag marks the launch as failed
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
}

template <int BM, int BN, int WM, int WN, int PL, bool DB = true, int FMT = 0, int PFD = 1, bool PIPE = false, bool SKT = false>
__global__ __launch_bounds__(WM * WN * 64) void gemm_split_kernel(SplitGemmArgs sa) {
  const GemmArgs& g = sa.g;
  constexpr int NT = WM * WN * 64;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 32, NI = TN / 32;
  constexpr int RS = PL * 64 + 16;                 // LDS row stride in bytes
  constexpr int A_F4 = BM * 8 / NT;                // fp32 float4 per thread per A tile
  constexpr int B_PCS = BN * PL * 4 / NT;          // 16-byte pieces per thread per W tile
  static_assert(A_F4 >= 1 && B_PCS >= 1 && (BM * 8) % NT == 0 && (BN * PL * 4) % NT == 0, "bad tiling");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
  constexpr int NBUF = DB ? 2 : 1;                 // single-buffered variant: less LDS -> more blocks per CU
  unsigned char* As = smem_s;                      // [NBUF][BM][RS]
  unsigned char* Bs = smem_s + NBUF * BM * RS;     // [NBUF][BN][RS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order: hardware places block b on XCD b % 8 (speed assumption only).  Give every XCD a
  // contiguous run of tiles, column tiles of one row tile adjacent, so the A row tile is fetched into ONE L2.
  constexpr bool SK = SKT;   // the instantiation that carries a stream-K tail (its own kernel: the extra state costs registers)
  static_assert(!SKT || (PIPE && BM == 128 && BN == 256), "stream-K is built for the pipelined 128x256 tile only");
  const int gx = g.N / BN, gy = (g.M + BM - 1) / BM;
  const int ntile_all = gx * gy;
  const int ntile = (SK && sa.sk_blocks > 0) ? sa.sk_first : ntile_all;   // tiles of the data-parallel part
  int tile;
  {
    const int b = blockIdx.x, q = ntile / 8, r = ntile % 8, xcd = b % 8, k = b / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  int m0 = (tile / gx) * BM, n0 = (tile % gx) * BN;   // re-assigned per segment by stream-K blocks
  const int grp = blockIdx.y;
  const float* A = g.A + grp * g.gA;
  const float* A2 = g.A2 ? g.A2 + grp * g.gA : nullptr;
  const unsigned char* Wsp = sa.Wsp + grp * sa.gWsp;
  const int K1 = g.A2 ? g.K1 : g.K;
  const int nkw = g.K / 32;          // K tiles of the whole problem (row stride of the split weights)
  int nk = nkw, kbase = 0;           // K tiles of the current segment and its first K tile (stream-K: a part of a tile's K range)

  // A staging: 8 lanes per row (one 8-byte ds_write per plane each).  A 16-lane store group covers two rows; rows 1 apart sit
  // 36 / 52 dwords apart and share four banks, rows 4 apart sit 16 banks apart: lanes 8-15 of a group take row + 4.
  const int lrow = (tid >> 6) * 8 + (lane >> 4) + 4 * ((lane >> 3) & 1), lc4 = (tid & 7) * 4;
  // W staging (16-byte pieces, PL * 4 per row): an 8-lane store group writing 8 consecutive pieces of the global image crosses a
  // row's 16-byte pad when PL = 3 (12 pieces per row) and hits four banks twice.  Instead a thread's first BN * 8 / NT pieces are
  // pieces 0-7 of a row (an 8-lane group = 128 contiguous bytes of ONE row), its last BN * 4 / NT pair the remaining pieces 8-11 of rows
  // r and r + 4 (the pad moves those 16 banks apart).  Piece i of a thread sits a fixed number of rows behind piece i - 1 of the same
  // kind, so the compiler keeps two base offsets and immediates (an irregular mapping spilled 16 VGPRs of the 128x256 kernel).
  constexpr int B_ROWPCS = PL == 3 ? BN * 8 / NT : B_PCS;
  static_assert(PL != 3 || ((BN * 8) % NT == 0 && (BN * 4) % NT == 0 && BN % 8 == 0), "W staging: whole 8-lane groups per thread slot");
  auto stage_piece = [&](int i, int& r, int& pc) {
    if constexpr (PL == 3) {
      if (i < B_ROWPCS) { const int a = tid + i * NT; r = a >> 3; pc = a & 7; }
      else { const int q = tid + (i - B_ROWPCS) * NT, gq = q >> 3, l = q & 7; r = (gq >> 2) * 8 + (gq & 3) + 4 * (l >> 2); pc = 8 + (l & 3); }
    } else { const int q = tid + i * NT; r = q / (PL * 4); pc = q % (PL * 4); }
  };
  static_assert(PFD >= 1 && (DB || PFD == 1) && (!PIPE || PFD <= 2), "deep prefetch needs the double-buffered LDS");
  f32x4 ra_[PFD][A_F4];   // PFD tiles in flight in registers (PFD > 1: small-M launches, where one tile's MFMAs are
  f32x4 rb_[PFD][B_PCS];  // far shorter than the L2/HBM latency and one-deep prefetch leaves the CU waiting)
  auto gload_set = [&](int kt, f32x4 (&ra)[A_F4], f32x4 (&rb)[B_PCS]) {
    const int k0 = (kbase + kt) * 32;
    const float* src = A; int ld = g.lda; int kk = k0;
    if (k0 >= K1) { src = A2; ld = g.lda2; kk = k0 - K1; }
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      int r = m0 + lrow + i * (NT / 8);
      r = r < g.M ? r : g.M - 1;
      ra[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)r * ld + kk + lc4);
    }
#pragma unroll
    for (int i = 0; i < B_PCS; ++i) {
      int r, pc;
      stage_piece(i, r, pc);
      rb[i] = *reinterpret_cast<const f32x4*>(Wsp + ((int64_t)(n0 + r) * nkw + kbase + kt) * (PL * 64) + pc * 16);
    }
  };
  auto lstore_set = [&](int buf, const f32x4 (&ra)[A_F4], const f32x4 (&rb)[B_PCS]) {
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      unsigned a[PL], b[PL];
      split_pair<PL, FMT>(ra[i][0], ra[i][1], a);
      split_pair<PL, FMT>(ra[i][2], ra[i][3], b);
      unsigned char* dst = As + (buf * BM + lrow + i * (NT / 8)) * RS + lc4 * 2;
#pragma unroll
      for (int p = 0; p < PL; ++p) *reinterpret_cast<u32x2*>(dst + p * 64) = u32x2{a[p], b[p]};
    }
#pragma unroll
    for (int i = 0; i < B_PCS; ++i) {
      int r, pc;
      stage_piece(i, r, pc);
      *reinterpret_cast<f32x4*>(Bs + (buf * BN + r) * RS + pc * 16) = rb[i];
    }
  };
  auto gload = [&](int kt) { gload_set(kt, ra_[0], rb_[0]); };
  auto lstore = [&](int buf) { lstore_set(buf, ra_[0], rb_[0]); };

  f32x16 acc[MI][NI];

  const int frow = lane & 31, fk = (lane >> 5) * 16;  // byte offset of this lane's 8 bf16 inside a 16-wide K step
  // Fragments of one 16-wide K step: MI + NI rows x PL planes, one ds_read_b128 each.
  auto read_frags = [&](int buf, int s, bf16x8 (&af)[MI][PL], bf16x8 (&bf)[NI][PL]) {
    const unsigned char* Ab = As + (buf * BM + wm * TM + frow) * RS + fk + s * 32;
    const unsigned char* Bb = Bs + (buf * BN + wn * TN + frow) * RS + fk + s * 32;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int p = 0; p < PL; ++p) af[i][p] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * RS + p * 64);
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int p = 0; p < PL; ++p) bf[j][p] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * RS + p * 64);
  };
  auto mma = [&](const bf16x8 (&af)[MI][PL], const bf16x8 (&bf)[NI][PL]) {
    // smallest cross terms first
#pragma unroll
    for (int ord = 2 * (PL - 1); ord >= 0; --ord) {
      if (ord > PL - 1) continue;  // keep only terms with pa + pb <= PL-1
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
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[MI][PL], bf[NI][PL];
      read_frags(buf, s, af, bf);
      mma(af, bf);
    }
  };

  // one pass over K tiles [kbase, kbase + nk) of tile (m0, n0): zeroes the accumulators, stages, multiplies
  auto mainloop = [&]() {
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  gload(0);
  lstore(0);
  if constexpr (PFD == 1 || PIPE) {
    if (nk > 1) gload_set(1, ra_[PFD - 1], rb_[PFD - 1]);
    if constexpr (PIPE && PFD == 2) gload_set(nk > 2 ? 2 : nk - 1, ra_[0], rb_[0]);
    __syncthreads();
    // Registers hold the NEXT tile: it is split and written into the other LDS buffer at the top of the iteration
    // (branch-free) and the registers are immediately refilled with tile kt+2, whose L2/HBM latency is then covered
    // by this iteration's MFMAs.
    // Measured on MI355X (tools/gemm_clock_probe.py): the 8-wave kernels run at the 1400 W package power cap
    // (shader clock throttled to ~1.7 GHz), so what counts is energy per MFMA; see DESIGN.md 4.1.
    if constexpr (DB && PIPE) {
      // Software pipeline inside the wave, in a FIXED issue order.
      // Why: the per-tile barrier keeps all waves of the block in the same phase, so "all fragment reads, wait, all
      // MFMAs, all stores, all loads" makes 8 waves hit the LDS, then the matrix pipe, then the address unit together
      // and every s_waitcnt in front of a cluster stalls in-order issue (no MFMAs either).  Here every MFMA is
      // followed by at most one memory mini-step, pinned with sched_barrier(0) (+8-11 % in same-box A/B runs):
      //   first half  (MFMAs of K step 0): ds_reads of step 1, then split + ds_write of tile kt+1   -> barrier
      //   second half (MFMAs of K step 1): global loads of tile kt+2, then ds_reads of step 0 of tile kt+1
      bf16x8 af0[MI][PL], bf0[NI][PL], af1[MI][PL], bf1[NI][PL];
      constexpr int N_TERM = PL * (PL + 1) / 2;
      constexpr int N_MMA = MI * NI * N_TERM;                // MFMAs per 16-wide K step
      constexpr int N_FRAG = (MI + NI) * PL;                 // ds_read_b128 per K step
      constexpr int E1 = 2 * A_F4 + B_PCS;                   // first-half store / refill steps
      // fragment read order = order of first use by the MFMAs (plane PL-1 of A and plane 0 of B first)
#define FRAG_ORDER(k) (((k) / (MI + NI)) == 0 ? (((k) % (MI + NI)) < MI ? ((k) % (MI + NI)) * PL + (PL - 1) : MI * PL + (((k) % (MI + NI)) - MI) * PL) \
                     : ((k) / (MI + NI)) == PL - 1 ? (((k) % (MI + NI)) < MI ? ((k) % (MI + NI)) * PL : MI * PL + (((k) % (MI + NI)) - MI) * PL + (PL - 1)) \
                     : (((k) % (MI + NI)) < MI ? ((k) % (MI + NI)) * PL + 1 : MI * PL + (((k) % (MI + NI)) - MI) * PL + 1))
      // cross terms, smallest first: (pa, pb) with pa + pb descending
      constexpr int TPA[6] = {PL == 3 ? 2 : 1, PL == 3 ? 1 : 0, 0, 1, 0, 0};
      constexpr int TPB[6] = {0, 1, PL == 3 ? 2 : 0, 0, 1, 0};
      auto step_mma = [&](int m, const bf16x8 (&af)[MI][PL], const bf16x8 (&bf)[NI][PL]) {
        const int t = m / (MI * NI), ij = m % (MI * NI), i = ij / NI, j = ij % NI;
        acc[i][j] = mfma_split<FMT>(af[i][TPA[t]], bf[j][TPB[t]], acc[i][j]);
      };
      // Fragment k of a K step, in the order the MFMAs consume them (lowest planes last).
      auto step_read = [&](int k, int buf, int s, bf16x8 (&af)[MI][PL], bf16x8 (&bf)[NI][PL]) {
        const unsigned char* Ab = As + (buf * BM + wm * TM + frow) * RS + fk + s * 32;
        const unsigned char* Bb = Bs + (buf * BN + wn * TN + frow) * RS + fk + s * 32;
        if (k < MI * PL) { const int i = k / PL, pp = k % PL; af[i][pp] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * RS + pp * 64); }
        else { const int q = k - MI * PL, j = q / PL, pp = q % PL; bf[j][pp] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * RS + pp * 64); }
      };
      auto step_gload = [&](int u, int kt, f32x4 (&ra)[A_F4], f32x4 (&rb)[B_PCS]) {
        if (u < A_F4) {
          const int k0 = (kbase + kt) * 32;
          const float* src = A; int ld = g.lda; int kk = k0;
          if (k0 >= K1) { src = A2; ld = g.lda2; kk = k0 - K1; }
          int r = m0 + lrow + u * (NT / 8);
          r = r < g.M ? r : g.M - 1;
          ra[u] = *reinterpret_cast<const f32x4*>(src + (int64_t)r * ld + kk + lc4);
        } else {
          int r, pc;
          stage_piece(u - A_F4, r, pc);
          rb[u - A_F4] = *reinterpret_cast<const f32x4*>(Wsp + ((int64_t)(n0 + r) * nkw + kbase + kt) * (PL * 64) + pc * 16);
        }
      };
      auto step_store = [&](int u, int buf, const f32x4 (&ra)[A_F4], const f32x4 (&rb)[B_PCS]) {
        if (u < A_F4) {
          unsigned a[PL], b[PL];
          split_pair<PL, FMT>(ra[u][0], ra[u][1], a);
          split_pair<PL, FMT>(ra[u][2], ra[u][3], b);
          unsigned char* dst = As + (buf * BM + lrow + u * (NT / 8)) * RS + lc4 * 2;
#pragma unroll
          for (int pp = 0; pp < PL; ++pp) *reinterpret_cast<u32x2*>(dst + pp * 64) = u32x2{a[pp], b[pp]};
        } else {
          int r, pc;
          stage_piece(u - A_F4, r, pc);
          *reinterpret_cast<f32x4*>(Bs + (buf * BN + r) * RS + pc * 16) = rb[u - A_F4];
        }
      };
      // Slot assignment.  An r02 sweep of ablation builds (components compiled out one at a time, bf16x6 8192x4096x4096; HISTORY.md):
      // MFMAs + barrier alone run at 2.07 PF; the fragment reads add 39 % to that, the global loads + LDS stores 31 %
      // (almost all of it the vmcnt waits in front of the stores), the split VALU 4 %, the barrier 1 %.  The spans
      // below say over which part of a half the LDS instructions are spread; PFD is how many tiles ahead the global
      // loads run (register sets).
      // (measured: spreading the LDS instructions over the whole half beats front-loading them -- 50 / 75 % spans -- by 1-2 %)
      constexpr int R_SPAN = N_MMA > N_FRAG ? N_MMA : N_FRAG;
      constexpr int S_SPAN = N_MMA;
      // first half of tile kt: MFMAs of K step 0.  Memory stream 1: fragment reads of step 1.  Stream 2: for every
      // staged register, split + ds_write (tile kt+1) followed one step later by its refill (tile kt+1+PFD).
      auto first_half = [&](int kt, f32x4 (&ra)[A_F4], f32x4 (&rb)[B_PCS]) {
        const int buf = kt & 1;
        const int ktn = kt + 1 + PFD < nk ? kt + 1 + PFD : nk - 1;   // branch-free tail: reloads the last tile (never consumed)
#pragma unroll
        for (int m = 0; m < N_MMA; ++m) {
          step_mma(m, af0, bf0);
#pragma unroll
          for (int k = 0; k < N_FRAG; ++k)
            if (k * R_SPAN / N_FRAG == m) step_read(FRAG_ORDER(k), buf, 1, af1, bf1);
#pragma unroll
          for (int e = 0; e < E1; ++e)
            if ((2 * e + 1) * S_SPAN / (2 * E1) == m) {
              if (e < 2 * A_F4) { if (e % 2 == 0) step_store(e / 2, buf ^ 1, ra, rb); else step_gload(e / 2, ktn, ra, rb); }
              else step_store(A_F4 + (e - 2 * A_F4), buf ^ 1, ra, rb);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // second half: MFMAs of K step 1; fragment reads of step 0 of tile kt+1 and the refill of the W registers
      auto second_half = [&](int kt, f32x4 (&ra)[A_F4], f32x4 (&rb)[B_PCS]) {
        const int buf = kt & 1;
        const int ktn = kt + 1 + PFD < nk ? kt + 1 + PFD : nk - 1;
#pragma unroll
        for (int m = 0; m < N_MMA; ++m) {
          step_mma(m, af1, bf1);
#pragma unroll
          for (int k = 0; k < N_FRAG; ++k)
            if (k * R_SPAN / N_FRAG == m) step_read(FRAG_ORDER(k), buf ^ 1, 0, af0, bf0);
#pragma unroll
          for (int e = 0; e < B_PCS; ++e)
            if ((2 * e + 1) * N_MMA / (2 * B_PCS) == m) step_gload(A_F4 + e, ktn, ra, rb);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      // (A rotated body [barrier | second half | first half of the next tile] makes the compiler's lgkmcnt waits exact
      // at the loop header but costs a peeled prologue/epilogue: same speed at K = 4096, 12 % slower at K = 128.)
#pragma unroll
      for (int k = 0; k < N_FRAG; ++k) step_read(FRAG_ORDER(k), 0, 0, af0, bf0);
      if constexpr (PFD == 1) {
        for (int kt = 0; kt < nk; ++kt) {
          first_half(kt, ra_[0], rb_[0]);
          __syncthreads();   // tile kt+1 visible; every wave holds its step-1 fragments of tile kt
          __builtin_amdgcn_sched_barrier(0);
          second_half(kt, ra_[0], rb_[0]);
        }
      } else {
        // two register sets: set 1 holds odd tiles, set 0 even tiles; tile kt+1 is stored while tile kt+3 is requested
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
      }
    } else
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = DB ? (kt & 1) : 0;
      if (DB) {
        lstore(buf ^ 1);               // branch-free: on the last iteration this rewrites the idle buffer (never read)
        if (kt + 2 < nk) gload(kt + 2);  // registers are free again: next-next tile flies during this tile's MFMAs
      }
      compute(buf);
      if (!DB) {
        __syncthreads();               // everyone done reading the single buffer
        lstore(0);
        if (kt + 2 < nk) gload(kt + 2);
      }
      __syncthreads();
    }
  } else {
    // deep prefetch: register set u holds tile kt+1 for kt = u (mod PFD); it is refilled with tile kt+1+PFD right
    // after being written to LDS, so PFD tiles are always in flight.
#pragma unroll
    for (int u = 0; u < PFD; ++u)
      if (1 + u < nk) gload_set(1 + u, ra_[u], rb_[u]);
    __syncthreads();
    for (int kt0 = 0; kt0 < nk; kt0 += PFD) {
#pragma unroll
      for (int u = 0; u < PFD; ++u) {
        const int kt = kt0 + u;
        if (kt < nk) {                 // block-uniform
          const int buf = kt & 1;
          lstore_set(buf ^ 1, ra_[u], rb_[u]);
          if (kt + 1 + PFD < nk) gload_set(kt + 1 + PFD, ra_[u], rb_[u]);
          compute(buf);
          __syncthreads();
        }
      }
    }
  }

  };   // mainloop

  const float* bias = g.bias ? g.bias + grp * g.gBias : nullptr;
  float* Y = g.Y + grp * g.gY;
  auto epilogue = [&]() {
  if (sa.wide_epi) {
    // Epilogue through LDS: the 32x32 C/D layout gives a lane ONE column and 16 rows, so direct stores are 64 dword
    // stores per lane (512 wave-instructions of 256 B per 128x256 block) and the block ends on a store-ISSUE-bound tail
    // about as long as the main loop of a K = 256 GEMM.  Instead every wave parks its TM x TN accumulator tile in its own
    // LDS region (row stride TN floats: the 32-lane halves of ds_write_b32 and the 16-lane groups of ds_read_b128 both
    // hit distinct banks) and writes it out as dwordx4 rows: 16 store instructions per lane, 1 KiB each; the residual
    // comes in the same way.  Wave-private regions: one block barrier (the tile buffers are dead), none inside.
    constexpr int LPR = TN / 4;           // lanes per output row
    constexpr int RPI = 64 / LPR;         // rows per wave-instruction
    constexpr int NIT = TM / RPI;
    __syncthreads();
    float* ep = reinterpret_cast<float*>(smem_s) + wave * (TM * TN);
    const int er = lane / LPR, ec = (lane % LPR) * 4;
    const int grow0 = m0 + wm * TM + er, gcol = n0 + wn * TN + ec;
    const bool row_norm = BN == 256 && g.norm != 0;      // this tile owns complete rows (dispatcher guarantees N == 256)
    f32x4 rres[NIT];
    if (g.R && !row_norm) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int row = grow0 + it * RPI;
        rres[it] = row < g.M ? *reinterpret_cast<const f32x4*>(g.R + (int64_t)row * g.ldr + gcol) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const float bv = bias ? bias[n0 + wn * TN + j * 32 + (lane & 31)] : 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + bv;
          if (g.act == ACT_RELU) v = fmaxf(v, 0.f);
          else if (g.act == ACT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
          else if (g.act == ACT_DIST) v = fmaxf(2.f - 2.f * v, 0.f);
          ep[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TN + j * 32 + (lane & 31)] = v;
        }
      }
    }
    if constexpr (BN == 256) {
      if (row_norm) {
        // LayerNorm / L2 normalisation of whole rows (same arithmetic, in the same order, as row_norm_kernel): every wave
        // takes BM / waves rows, a lane 4 consecutive columns; the row is gathered from the WN accumulator regions.
        __syncthreads();
        constexpr int RPW = BM / (WM * WN);
        const float* epb = reinterpret_cast<const float*>(smem_s);
        const int c0 = lane * 4, wn_c = c0 / TN, cc = c0 % TN;
        for (int it = 0; it < RPW; ++it) {
          const int rl = wave * RPW + it;
          const int row = m0 + rl;
          if (row >= g.M) break;                        // wave-uniform
          f32x4 v = *reinterpret_cast<const f32x4*>(epb + ((rl / TM) * WN + wn_c) * (TM * TN) + (rl % TM) * TN + cc);
          if (g.R) {
            const f32x4 rr = *reinterpret_cast<const f32x4*>(g.R + (int64_t)row * g.ldr + c0);
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] += rr[c];
          }
          f32x4 o;
          if (g.norm == 1) {
            const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.f / 256);
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float d = v[c] - mean; q += d * d; }
            const float rstd = 1.f / sqrtf(wave_sum(q) * (1.f / 256) + g.eps);
            const f32x4 ga = *reinterpret_cast<const f32x4*>(g.gamma + c0);
            const f32x4 be = *reinterpret_cast<const f32x4*>(g.beta + c0);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (v[c] - mean) * rstd * ga[c] + be[c];
          } else {
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) q += v[c] * v[c];
            const float nrm = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = v[c] / nrm;
          }
          if (g.add2) {
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(g.add2 + (int64_t)row * g.ldadd2 + c0);
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] += a2[c];
          }
          *reinterpret_cast<f32x4*>(Y + (int64_t)row * g.ldy + c0) = o;
        }
        return;
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int row = grow0 + it * RPI;
      f32x4 v = *reinterpret_cast<const f32x4*>(ep + (er + it * RPI) * TN + ec);
      if (g.R) {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += rres[it][c];
      }
      // write-through (sc1) stores: the fresh activations leave the XCD's L2 while the launch is still running instead of
      // being written back behind it (same-box A/B at cfg3: 9.72 -> 9.82 M descriptors/s, twice)
      if (row < g.M) sk_store16(Y + (int64_t)row * g.ldy + gcol, v);
    }
    return;
  }
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
  };   // epilogue

  // ---- stream-K tail.  The tiles the data-parallel part leaves over (ntile_all - sk_first of them: fewer than one per
  // CU, so a plain launch would end on a partly empty round) are cut into (tile, K tile) iterations and every one of the
  // sk_blocks blocks takes an equal run of them.  A tile whose K range is shared is finished by the block that holds its
  // FIRST K tiles: the others publish their partial accumulator tile through the workspace.
  // Runs are handed out in REVERSE block order: block s takes run (sk_blocks - 1 - s).  A block starts with the tail of
  // a shared tile (publishes immediately) and ends with the head of another one, whose remaining parts belong to
  // blocks s-1, s-2 .. -- LOWER indices, dispatched earlier (so the wait cannot starve an un-dispatched block) and
  // published at the very beginning of those blocks' lives (so the wait is normally over already).  Fixed partition
  // and fixed summation order: deterministic.  Data-parallel blocks run the loop below exactly once.
  bool sk = false;
  int s_ = 0, S = 1;
  // 32-bit on purpose (at most 256 tiles x 64 K tiles x 256 runs): 64-bit divisions cost dozens of registers here
  unsigned it_i = 0, it_hi = 1, I = 0;
  if constexpr (SK) {
    if (sa.sk_blocks > 0 && (int)blockIdx.x >= sa.sk_first) {
      sk = true;
      S = sa.sk_blocks; s_ = (int)blockIdx.x - sa.sk_first;
      I = (unsigned)(ntile_all - sa.sk_first) * (unsigned)nkw;
      it_i = I * (unsigned)(S - 1 - s_) / (unsigned)S;
      it_hi = I * (unsigned)(S - s_) / (unsigned)S;
      if (it_i >= it_hi) return;
    }
  }
  constexpr int SLOT = BM * BN;
  for (;;) {
    int kb = 0, ke = nkw;
    if (SK && sk) {
      const unsigned tq = it_i / (unsigned)nkw;
      const int t = sa.sk_first + (int)tq;
      kb = (int)(it_i - tq * (unsigned)nkw);
      ke = kb + (int)(it_hi - it_i) < nkw ? kb + (int)(it_hi - it_i) : nkw;
      m0 = (t / gx) * BM; n0 = (t % gx) * BN;
      __syncthreads();                         // LDS of the previous segment (tiles or epilogue regions) is dead
    }
    kbase = kb; nk = ke - kb;
    mainloop();
    if (SK && sk && kb > 0) {                  // tail / middle part of a shared tile: publish
      float* slot = sa.sk_ws + (int64_t)s_ * SLOT;
#pragma unroll
      for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            sk_store16(slot + ((((a * NI + b) * 4 + q) * NT) + tid) * 4,
                       f32x4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]});
      sk_publish(sa.sk_flags + s_, sa.sk_epoch);
    } else {
      if (SK && sk && ke < nkw) {              // head of a shared tile (last segment of this block): collect the rest
        int rem = nkw - ke;
        for (int j = s_ - 1; rem > 0 && j >= 0; --j) {
          const int len = (int)(I * (unsigned)(S - j) / (unsigned)S - I * (unsigned)(S - 1 - j) / (unsigned)S);
          if (len == 0) continue;
          sk_wait(sa.sk_flags + j, sa.sk_epoch, sa.sk_flags + S);
          const float* slot = sa.sk_ws + (int64_t)j * SLOT;
#pragma unroll
          for (int a = 0; a < MI; ++a) {       // one accumulator row block (8 x 16 bytes per lane) in flight at a time
            f32x4 pv[NI][4];
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
              for (int q = 0; q < 4; ++q) pv[b][q] = sk_load16(slot + ((((a * NI + b) * 4 + q) * NT) + tid) * 4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int b = 0; b < NI; ++b)
#pragma unroll
              for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[a][b][4 * q + c] += pv[b][q][c];
          }
          rem -= len < rem ? len : rem;
        }
      }
      epilogue();
    }
    if (!(SK && sk)) break;
    it_i += (unsigned)(ke - kb);
    if (it_i >= it_hi) break;
  }
}

template <int BM, int BN, int WM, int WN, int PL, bool DB = true, int FMT = 0, int PFD = 1, bool PIPE = false>
inline void gemm_split_launch_t(const SplitGemmArgs& sa, int groups, hipStream_t st) {
  constexpr size_t lds_main = (size_t)(DB ? 2 : 1) * (BM + BN) * (PL * 64 + 16);
  constexpr size_t lds_epi = (size_t)BM * BN * sizeof(float);    // every wave's TM x TN accumulator tile
  constexpr bool epi_fits = lds_epi <= 160 * 1024;
  constexpr size_t lds = (epi_fits && lds_epi > lds_main) ? lds_epi : lds_main;
  static const bool narrow = LT_XENV("LINETR_GEMM_NARROW_EPI") != nullptr;   // tuning aid: direct dword stores
  SplitGemmArgs sa2 = sa;
  // the LDS epilogue needs 16-byte aligned rows (ldy / ldr multiples of 4 floats; the entry points guarantee it for
  // their own buffers, debug_gemm checks it)
  sa2.wide_epi = !narrow && sa.g.ldy % 4 == 0 && (!sa.g.R || sa.g.ldr % 4 == 0) && epi_fits;
  static unsigned long long attr_done = 0;   // one bit per device: the opt-in is a per-device function attribute
  const unsigned long long dev_bit = current_device_bit();
  if (!(attr_done & dev_bit)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<BM, BN, WM, WN, PL, DB, FMT, PFD, PIPE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done |= dev_bit;
  }
  dim3 grid((sa.g.N / BN) * cdiv(sa.g.M, BM), groups);
  sa2.sk_first = sa2.sk_blocks = 0;
#ifdef LINETR_EXPERIMENTS
  if constexpr (PIPE && BM == 128 && BN == 256) {
    // stream-K tail: when the last round of tiles would leave a good part of the chip idle, those tiles are shared by one
    // block per CU instead (see the kernel).  Measured quantisation: 25472 x 512 x 512 (398 tiles) took as long as
    // 32768 x 512 x 512 (512 tiles), 104 us.
    // OFF by default (LINETR_STREAMK=1 turns it on, read per launch so that the tests can exercise it): the stream-K
    // kernel is correct and deterministic (tests/test_gpu_gemm.py) but as built it LOSES -- 25472x512x512 110 us vs 92 us,
    // 25472x768x256 114 us vs 82 us -- because the segment loop's extra state spills 60 VGPRs and 70 SGPRs next to the
    // 256-register pipelined main loop, which slows the data-parallel tiles of the same launch as well.  Kept as the
    // starting point for a leaner version.
    const bool no_sk = LT_XENV("LINETR_STREAMK") == nullptr;
    static int n_cu = 0;
    if (!n_cu) {
      int dev = 0;
      hipDeviceProp_t prop;
      n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const int T = (int)grid.x, nkw = sa.g.K / 32;
    const int full = T / n_cu * n_cu, rem = T - full;
    const int idle_ok = full > 0 ? n_cu * 13 / 16 : n_cu * 11 / 16;    // only when >= 3/16 (5/16 for a single round) of the CUs would idle
    if (!no_sk && sa.sk_ws && groups == 1 && sa2.wide_epi && rem > 0 && rem <= idle_ok && (int64_t)rem * nkw >= 2 * n_cu &&
        n_cu <= 256 && nkw <= 4096) {
      sa2.sk_first = full;
      sa2.sk_blocks = n_cu;
      grid.x = full + n_cu;
      // the stream-K kernel is its own instantiation with one register set of prefetch (PFD = 1): with two, the segment
      // loop's extra state spilled 116 VGPRs
      static unsigned long long attr_sk = 0;
      if (!(attr_sk & dev_bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_kernel<BM, BN, WM, WN, PL, DB, FMT, 1, PIPE, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_sk |= dev_bit;
      }
      hipLaunchKernelGGL((gemm_split_kernel<BM, BN, WM, WN, PL, DB, FMT, 1, PIPE, true>), grid, dim3(WM * WN * 64), lds, st, sa2);
      return;
    }
  }
#endif
  hipLaunchKernelGGL((gemm_split_kernel<BM, BN, WM, WN, PL, DB, FMT, PFD, PIPE>), grid, dim3(WM * WN * 64), lds, st, sa2);
}

template <int PL, int FMT>
inline void gemm_split16_launch(const SplitGemmArgs& sa, hipStream_t st);   // lt_gemm_split16.h (112-row tiles)
inline bool split16_wins(const GemmArgs& g, int groups);
template <int PL, int FMT>
inline void gemm_split_small_launch(const SplitGemmArgs& sa, int groups, hipStream_t st);   // lt_gemm_small.h (single-pair sizes)
inline bool small_gemm_wins(const GemmArgs& g, int groups);

// Tile choice, from the measured table tools/gemm_tiles_probe.py prints (bf16x6, MI355X, profiles/r02_tiles_probe.txt):
//   * 8-wave 128x256 blocks (fixed-order pipeline, one block per CU) as soon as there are ~140 of them: 9584x512x512 runs
//     43 us on 150 such blocks against 64 us on 300 128x128 blocks;
//   * below that the 4-wave 64x64 tile (53 KB of LDS and 156 VGPRs: three blocks per CU) while its grid fits the
//     768 resident slots, then 64x128 (two per CU, 512 slots), then 64x256;
//   * single-pair sizes go to the barrier-free K-split kernel (lt_gemm_small.h).
inline const char* split_tile_name(const GemmArgs& g, int groups, int pl = 3) {
  if (small_gemm_wins(g, groups)) return "32x32k4";   // latency-bound sizes: barrier-free K-split kernel
  if (g.N % 128 != 0) return "128x64";
  static const bool no112 = LT_XENV("LINETR_NO_TILE112") != nullptr;   // tuning aid
  if (!no112 && pl == 2 && split16_wins(g, groups)) return "112x256";   // saves a round of blocks (lt_gemm_split16.h)
  const int64_t r128 = cdiv(g.M, 128), r64 = cdiv(g.M, 64);
  // short K, wide N, many tiles: the single-buffered 128x128 tile (64 KB of LDS, eight waves at 102 VGPRs: two blocks =
  // 4 waves per SIMD whose prologues / epilogues overlap each other's main loops) beats the one-block-per-CU pipeline:
  // 25472x768x256 73 vs 80 us, 291208x256x128 (the word-MLP layer) 192 vs 225 us; in the cfg3 step its nine launches take
  // 0.75 ms (0.85 ms with the earlier four-wave layout at 212 VGPRs = 2 waves per SIMD).  For the other shapes the two
  // tiles are level inside the step.
  static const bool no128s = LT_XENV("LINETR_NO_TILE128S") != nullptr;   // tuning aid
  // r04 (profiles/r04_tiles_probe.txt): with K <= 256 and N >= 768 it already wins from ~400 tiles (9584 rows, cfg5 at 8 pairs:
  // 9584x1024x256 45.6 vs 55.4 us, 9584x768x256 30.0 vs 31.8 us)
  const int64_t t128 = r128 * (g.N / 128) * groups;
  if (!no128s && pl == 3 && ((g.K <= 256 && g.N >= 768 && t128 >= 400) || (g.K <= 128 && t128 >= 1024))) return "128x128s";
  if (g.N % 256 == 0 && r128 * (g.N / 256) * groups >= 140) return "128x256";
  // r04 probe, 9584 rows x N = 256 (75 row tiles: too few 128 x 256 blocks): two 128 x 128 s blocks per CU beat the 64 x 64 tile for
  // K <= 512 (27.7 vs 29.5 us at K = 512, 17.0 vs 18.0 at K = 256); at K = 1024 the tiles are level
  if (!no128s && pl == 3 && g.K <= 512 && t128 >= 140) return "128x128s";
  if (g.N % 256 != 0 && (int64_t)cdiv(g.M, 256) * (g.N / 128) * groups >= 192) return "256x128";
  if (r64 * (g.N / 64) * groups <= 768) return "64x64";
  if (r64 * (g.N / 128) * groups <= 512) return "64x128";
  if (g.N % 256 == 0) return r64 * (g.N / 256) * groups >= 140 ? "64x256" : "64x128";
  return "128x128";
}

template <int PL, int FMT = 0>
inline int gemm_split_launch(const SplitGemmArgs& sa, int groups, hipStream_t st) {
  const GemmArgs& g = sa.g;
  if (g.M <= 0) return 0;
  if (g.N % 64 != 0 || g.K % 32 != 0 || (g.A2 && g.K1 % 32 != 0))
    return fail(LINETR_E_ARG, "gemm_split: unsupported shape M=%d N=%d K=%d", g.M, g.N, g.K);
  static const char* tile_env = LT_XENV("LINETR_GEMM_TILE");  // tuning aid: force a tile
  const char* tile = tile_env ? tile_env : split_tile_name(g, groups, PL);
  // the launcher is authoritative about the fused row normalisation: only the 128x256 tile with the LDS epilogue owns
  // complete rows of an N = 256 problem; anything else would silently skip the normalisation
  if (g.norm != 0) {
    static const bool narrow_env = LT_XENV("LINETR_GEMM_NARROW_EPI") != nullptr;
    if (strcmp(tile, "128x256") != 0 || g.N != 256 || narrow_env || g.ldy % 4 != 0 || (g.R && g.ldr % 4 != 0))
      return fail(LINETR_E_ARG, "gemm_split: fused row normalisation asked of tile %s (N=%d): dispatcher bug", tile, g.N);
  }
  if (!strcmp(tile, "32x32k4")) gemm_split_small_launch<PL, FMT>(sa, groups, st);
  else if (!strcmp(tile, "112x256")) gemm_split16_launch<PL, FMT>(sa, st);
  else if (g.N % 128 != 0 || !strcmp(tile, "128x64")) gemm_split_launch_t<128, 64, 4, 1, PL, true, FMT>(sa, groups, st);
  else if (!strcmp(tile, "256x128")) {
#ifdef LINETR_EXPERIMENTS
    static const bool nopipe = LT_XENV("LINETR_GEMM_NOPIPE") != nullptr;   // tuning aid: the pre-pipelining main loop
    if (nopipe) gemm_split_launch_t<256, 128, 4, 2, PL, true, FMT>(sa, groups, st);
    else
#endif
    gemm_split_launch_t<256, 128, 4, 2, PL, true, FMT, 1, true>(sa, groups, st);
  }
  // (16 waves of 32 x 64 or 64 x 32 on this tile, without the software pipeline, run the cfg3 step within noise of this
  // one: at one block per CU the extra waves meet at the same barriers)
  else if (!strcmp(tile, "128x256") && g.N % 256 == 0) gemm_split_launch_t<128, 256, 2, 4, PL, true, FMT, PL == 3 ? 2 : 1, true>(sa, groups, st);
  else if (!strcmp(tile, "64x256") && g.N % 256 == 0) gemm_split_launch_t<64, 256, 1, 4, PL, true, FMT, PL == 3 ? 2 : 1, true>(sa, groups, st);
  else if (!strcmp(tile, "128x128")) gemm_split_launch_t<128, 128, 2, 2, PL, true, FMT>(sa, groups, st);
  // single LDS buffer, EIGHT waves of 64 x 32 (102 VGPRs): two blocks = 4 waves per SIMD.  With four waves of 64 x 64
  // (212 VGPRs, 2 waves per SIMD) 25472x768x256 took 76.4 us (now 73.2), 291208x256x128 206 us (now 192)
  else if (!strcmp(tile, "128x128s")) {
#ifdef LINETR_EXPERIMENTS
    static const bool w4 = LT_XENV("LINETR_TILE128S_4WAVE") != nullptr;   // tuning aid: the four-wave layout
    if (w4) gemm_split_launch_t<128, 128, 2, 2, PL, false, FMT>(sa, groups, st);
    else
#endif
    gemm_split_launch_t<128, 128, 2, 4, PL, false, FMT>(sa, groups, st);
  }
#ifdef LINETR_EXPERIMENTS
  else if (PL == 2 && !strcmp(tile, "256x256") && g.N % 256 == 0) gemm_split_launch_t<256, 256, 4, 2, 2, true, FMT>(sa, groups, st);
#endif
  // 64x64: three tiles of register prefetch = 156 VGPRs = THREE blocks per CU (53 KB of LDS each); with four it was 172
  // VGPRs = two blocks: 9584 x 256 x {256, 512, 1024} 20.9 / 33.2 / 58.8 us -> 18.6 / 29.8 / 52.8 us
  else if (!strcmp(tile, "64x64")) gemm_split_launch_t<64, 64, 2, 2, PL, true, FMT, 3>(sa, groups, st);
  else gemm_split_launch_t<64, 128, 2, 2, PL, true, FMT, 3>(sa, groups, st);
  LT_LAUNCH_CHECK();
  return 0;
}

}  // namespace lt
