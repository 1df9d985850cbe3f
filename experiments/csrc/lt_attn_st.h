// Signature attention (models/line_transformer.py:132-154) on split-tile operands (lt_gemm_st.h): q / k / v arrive as the
// ST image the projection GEMM wrote ([N][768], three bf16 planes), the message leaves as an ST image ([N][256]).
// Same mathematics and the same six-product MFMA scheme as sig_attn_split_kernel (lt_model.h); what changes is the
// data path:
//   * Q fragments are 12 plain 16-byte loads per lane (the planes exist already: no split VALU);
//   * K and V tiles travel HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4): K as linear copies of ST chunks (the chunk
//     image IS the conflict-free A-fragment image), V as a GATHER (the DMA's source address is per lane, only the
//     destination is linear) into [plane][kv][8 x 16 B] rows with the d-piece index XOR-swizzled by bit 1 of kv, which is
//     what makes gfx950's transposing read (ds_read_b64_tr_b16) of V^T conflict-free without padding;
//   * the O^T accumulators (lane = query, 4-runs of d) become 8-wide ST pieces with one v_permlane32_swap per register
//     pair and are stored as 256-byte runs: no LDS, no per-dword row scatter in the epilogue.
// Key tiles are aligned to the image's first 16-row block, not to the image: rows of neighbouring images that share a
// block are masked like the rows past the end.
#pragma once
#include "lt_gemm_st.h"
#include "lt_model.h"

namespace lt {

constexpr int ATQ_KT = 64;                       // keys per tile
constexpr int ATQ_K_BYTES = 4 * 4 * ST_RB;       // [4 K steps of the head's 64 channels][4 row blocks][3 planes][512]
constexpr int ATQ_V_BYTES = 3 * ATQ_KT * 128;    // [3 planes][64 kv][8 d-pieces x 16 B]

// Six transposing reads = V^T fragments of one 16-wide kv step (KV0) and one 32-wide d block; `base` already carries the
// lane's row / piece / swizzle and the d block.  out[p][0/1]: plane p, kv run KV0 + {0..3} / KV0 + 8 + {0..3} (+ 4 h2).
template <int KV0>
__device__ __forceinline__ void v_frags_st(unsigned base, u32x2 (&o)[3][2]) {
  constexpr int R0 = KV0 * 128, R1 = (KV0 + 8) * 128, PB = ATQ_KT * 128;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %6 offset:%7\n\t"
      "ds_read_b64_tr_b16 %1, %6 offset:%8\n\t"
      "ds_read_b64_tr_b16 %2, %6 offset:%9\n\t"
      "ds_read_b64_tr_b16 %3, %6 offset:%10\n\t"
      "ds_read_b64_tr_b16 %4, %6 offset:%11\n\t"
      "ds_read_b64_tr_b16 %5, %6 offset:%12\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(o[0][0]), "=&v"(o[0][1]), "=&v"(o[1][0]), "=&v"(o[1][1]), "=&v"(o[2][0]), "=&v"(o[2][1])
      : "v"(base), "n"(R0), "n"(R1), "n"(R0 + PB), "n"(R1 + PB), "n"(R0 + 2 * PB), "n"(R1 + 2 * PB)
      : "memory");
}

// grid (image, head, 256-query tile), 512 threads: wave w owns 32 queries.  OCC = blocks per CU the register budget is cut
// for (2: <= 128 VGPRs, 1: <= 256).  All byte offsets are 32-bit: the launcher checks that the q/k/v image is < 4 GiB.
template <int OCC>
__global__ __launch_bounds__(512, OCC == 2 ? 4 : 2) void sig_attn_st_kernel(const unsigned char* __restrict__ qkv /*ST [N][768]*/,
                                                             const int* __restrict__ cu_sub, int n_images, int N,
                                                             unsigned char* __restrict__ out /*ST [N][256]*/) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[ATQ_K_BYTES + ATQ_V_BYTES];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + ATQ_K_BYTES;
  const int img = blockIdx.x, head = blockIdx.y;
  const int n0 = cu_sub[img], Ni = cu_sub[img + 1] - n0;
  const int q0 = blockIdx.z * 256;
  if (q0 >= Ni) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h2 = lane >> 5, lq = lane & 31;
  const int q = q0 + wave * 32 + lq;
  const bool wave_active = q0 + wave * 32 < Ni;  // wave-uniform
  const unsigned RB = (unsigned)st_row_blocks(N);

  // ---- Q fragments: B operand of S^T = K Q^T; lane -> query lq, channels 16 s + 8 h2 .. + 8 of the head
  bf16x8 qf[4][3];
  {
    const int qr = n0 + (q < Ni ? q : Ni - 1);
    const unsigned qo = ((unsigned)(head * 4) * RB + (unsigned)(qr >> 4)) * ST_RB + h2 * 256 + (qr & 15) * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        qf[s][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(qkv + (qo + (unsigned)s * RB * ST_RB + p * ST_CHUNK)));
  }
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m = -INFINITY, l = 0.f;      // running max in log2 units, running sum

  // ---- DMA plan: waves 0-3 copy K (24 instructions of 1 KiB: K step s = instruction / 6, a 6 KiB span of 4 row blocks),
  // waves 4-7 gather V (24 instructions: plane = instruction / 8, 8 kv rows x 8 d-pieces each)
  const int rb_img = n0 >> 4;                      // first row block of the image; key tile t starts at row block rb_img + 4 t
  const int krow0 = rb_img * 16;                   // global row of key 0 of tile 0
  // lane address of the transposing V reads for kv0 = 0 and d block 0 / 1 (see the header comment)
  const unsigned v_lane = (unsigned)((((lane & 15) >> 2) + 4 * h2) * 128 +
                                     (((((lane >> 4) & 1) * 2 + ((lane & 3) >> 1)) ^ (((lane >> 3) & 1) << 2)) * 16) + (lane & 1) * 8);
  const unsigned v_base0 = (unsigned)(size_t)Vs + v_lane, v_base1 = v_base0 ^ 64u;
  const unsigned char* kfrag = Ks + ((lq >> 4) * 3) * ST_CHUNK + h2 * 256 + (lq & 15) * 16;

  const int n_tiles = (n0 + Ni - krow0 + ATQ_KT - 1) / ATQ_KT;
  for (int t = 0; t < n_tiles; ++t) {
    __syncthreads();                               // everyone is done with the previous tile
    const int rb_t = rb_img + 4 * t;
    if (wave < 4) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int x = wave * 6 + i;                // 0..23
        const int s = x / 6, byte = (x % 6) * 1024 + lane * 16;      // byte inside the 6 KiB span of K step s
        unsigned rb = (unsigned)rb_t + byte / ST_RB;
        rb = rb < RB ? rb : RB - 1;                // past the image: any finite row will do (masked below)
        const unsigned char* g = qkv + (((unsigned)(16 + head * 4 + s) * RB + rb) * ST_RB + byte % ST_RB);
        LT_GLDS(g, Ks + x * 1024, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int x = (wave - 4) * 6 + i;          // 0..23
        const int p = x >> 3, kv = (x & 7) * 8 + (lane >> 3);
        const int dp = (lane & 7) ^ (((kv >> 1) & 1) << 2);
        const unsigned row = (unsigned)rb_t * 16 + kv;
        unsigned rb = row >> 4;
        rb = rb < RB ? rb : RB - 1;
        const unsigned char* g = qkv + (((unsigned)(32 + head * 4 + (dp >> 1)) * RB + rb) * ST_RB + p * ST_CHUNK + (dp & 1) * 256 + (row & 15) * 16);
        LT_GLDS(g, Vs + x * 1024, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!wave_active) continue;
#pragma unroll
    for (int c = 0; c < ATQ_KT / 32; ++c) {
      const int kvg = rb_t * 16 + c * 32;          // global row of key 0 of this chunk
      if (kvg >= n0 + Ni) break;
      f32x16 st;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 ka[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) ka[p] = *reinterpret_cast<const bf16x8*>(kfrag + ((s * 4 + 2 * c) * 3 + p) * ST_CHUNK);
        // six products, smallest first: (2,0) (1,1) (0,2) (1,0) (0,1) (0,0)
        st = mfma_split<0>(ka[2], qf[s][0], st);
        st = mfma_split<0>(ka[1], qf[s][1], st);
        st = mfma_split<0>(ka[0], qf[s][2], st);
        st = mfma_split<0>(ka[1], qf[s][0], st);
        st = mfma_split<0>(ka[0], qf[s][1], st);
        st = mfma_split<0>(ka[0], qf[s][0], st);
      }
      if (kvg < n0 || kvg + 32 > n0 + Ni) {   // wave-uniform: rows of a neighbouring image / past the end
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = kvg + (r & 3) + 8 * (r >> 2) + 4 * h2;
          if (row < n0 || row >= n0 + Ni) st[r] = -INFINITY;
        }
      }
      float mx = st[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
      mx = xor32_max(mx) * LOG2E;
      const float m_new = fmaxf(m, mx);
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { st[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], LOG2E, -m_new)); ps += st[r]; }
      ps = xor32_sum(ps);
      if (__any(m_new != m)) {      // wave-uniform: once the running max has settled the accumulators need no rescale
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);   // m = -inf on the first chunk -> 0
        l *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      }
      l += ps;
      m = m_new;
      // P^T planes straight from the accumulator registers; V^T fragments by transposing reads
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        bf16x8 pp[3];
        {
          unsigned w[4][3];
#pragma unroll
          for (int e = 0; e < 4; ++e) split_pair<3>(st[8 * t2 + 2 * e], st[8 * t2 + 2 * e + 1], w[e]);
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
          const unsigned vb = dt == 0 ? v_base0 : v_base1;
          if (c == 0) { if (t2 == 0) v_frags_st<0>(vb, vr); else v_frags_st<16>(vb, vr); }
          else        { if (t2 == 0) v_frags_st<32>(vb, vr); else v_frags_st<48>(vb, vr); }
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

  // ---- epilogue: O^T / l -> 8-wide pieces (half-swap) -> three planes -> ST image of the message
  if (wave_active) {
    const float inv = 1.f / l;
    const int row = n0 + q;
    const unsigned rb = (unsigned)row >> 4;
    const int r16 = row & 15;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = (dt == 0 ? o0[r] : o1[r]) * inv;
#pragma unroll
      for (int c = 0; c < 4; ++c) { swap32(v[c], v[4 + c]); swap32(v[8 + c], v[12 + c]); }
#pragma unroll
      for (int g = 0; g < 2; ++g) {        // output K step head * 4 + 2 dt + g, piece h2: d = 32 dt + 16 g + 8 h2 .. + 8
        unsigned s0[3], s1[3], s2[3], s3[3];
        split_pair<3>(v[8 * g + 0], v[8 * g + 1], s0); split_pair<3>(v[8 * g + 2], v[8 * g + 3], s1);
        split_pair<3>(v[8 * g + 4], v[8 * g + 5], s2); split_pair<3>(v[8 * g + 6], v[8 * g + 7], s3);
        if (q < Ni) {
          unsigned char* dst = out + (((unsigned)(head * 4 + 2 * dt + g) * RB + rb) * ST_RB + h2 * 256 + r16 * 16);
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(dst + p * ST_CHUNK) = u32x4{s0[p], s1[p], s2[p], s3[p]};
        }
      }
    }
  }
  // the rows between N and the end of the ST image are read (never used) by the next GEMM's activation panels and by the
  // next layer's key tiles, where a NaN bit pattern would survive the masking as 0 x NaN: keep them zero
  if (img == n_images - 1 && blockIdx.z == 0) {
    const int pad = (int)(RB * 16) - N;                         // < 128
    for (int i = tid; i < pad * 4 * 2 * 3; i += 512) {          // (row, K step of the head, q, plane)
      const int prow = N + i / 24, rest = i % 24, ks = rest / 6, qq = (rest % 6) / 3, p = rest % 3;
      *reinterpret_cast<u32x4*>(out + (((unsigned)(head * 4 + ks) * RB + (unsigned)(prow >> 4)) * ST_RB + p * ST_CHUNK + qq * 256 + (prow & 15) * 16)) =
          u32x4{0u, 0u, 0u, 0u};
    }
  }
}

}  // namespace lt
