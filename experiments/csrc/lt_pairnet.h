// The line-signature network of a SINGLE image pair (or a few images) as ONE persistent launch.
//
// Replaces, for small batches, the chain of ~30 dependent launches that models/line_transformer.py:132-183 + :245-246
// (seven AttentionalPropagation layers, final_proj, F.normalize) turn into at M = a few hundred rows: every one of those
// launches does ~2 us of work behind ~5 us of dispatch, first-touch weight fetch and drain.  Here one block
// per CU lives for the whole network and walks a fixed list of stages
//     QKV(0) | ATTN(l) W1(l) W2(l) QKV(l+1) ... | ATTN(L-1) W1(L-1) FINAL NORM
// whose units (a 32 x 32 output tile of a GEMM, a (32 queries, head) slice of the attention, a 32-row slice of the
// normalisation) are dealt round-robin over the blocks.  There is no grid barrier: a unit waits only for the producer
// units it reads -- one arrival counter per (stage, 32-row tile) -- so the tiles of a row band chase each other through
// the layers, and a unit's weight tiles (which depend on nothing) are already in flight while it waits.
//
// Hand-off (MI355X_MICROARCH.md "Valid forms", cdna_hip_programming.md G16 R1): producers store their output WRITE-THROUGH
// (16-byte sc1 stores), every storing thread drains vmcnt, one lane adds to the counter (agent scope); consumers poll the
// counter relaxed from one lane and read the payload with sc1 loads (L1 bypassed, so no acquire fence is needed -- every
// buffer is written once per launch).  All spins are bounded: a block that waits longer than PN_TIMEOUT_TICKS raises the
// abort word, every other block sees it at its next poll and the launch ends (the host reads the word from mapped memory
// before the next call and retires this path for the handle).
//
// Arithmetic = the per-launch kernels': gemm_split_small_kernel<3, 0, 8> (lt_gemm_small.h: 32 x 32 tile, the 8 waves split K,
// wave-private staging, partial sums added in wave order) and sig_attn_small_kernel (lt_model.h: 32 queries, 4 waves split the
// keys, merged through LDS).  Row tiles are per IMAGE (an image's last tile is partial), so a tile never straddles two
// softmax domains.
#pragma once
#include "lt_gemm_split.h"
#include "lt_gemm_split16.h"
#include "lt_gemm_small.h"

namespace lt {

constexpr int PN_MAX_IMAGES = 8;
constexpr int PN_MAX_LAYERS = 15;
constexpr int PN_MAX_RT = 64;                 // 32-row tiles over all images
constexpr int PN_THREADS = 512;
constexpr int PN_CNT_STRIDE = 32;             // ints between two arrival counters: one 128-byte line each (256 pollers on one line
                                              // serialise at ~12 ns per access and hold up the producers' arrivals behind them)
constexpr long long PN_TIMEOUT_TICKS = 200000000;   // wall_clock64 runs at 100 MHz: 2 s
// Two variations that were measured on MI355X and lost (tools/pairnet_timeline.py, cfg2: 279 us without, 307 us with both):
constexpr bool PN_WIDE_QKV = false;      // 32 x 64 q/k/v units (168 instead of 336: one round on 256 CUs) -- 196 KB of weights per unit
constexpr bool PN_TOUCH_NEXT = false;    // touch the next unit's weight lines into L2 while the current unit runs -- more per-CU traffic

struct PnLayer {
  const unsigned char *Wqkv, *W1, *W2;        // split-bf16x3 planes [N][K/32][3][32] (lt_gemm_split.h)
  const float *bqkv, *b1, *b2;
};

struct PairNetArgs {
  int n_images, n_layers, n_rt, N;
  int img_row0[PN_MAX_IMAGES + 1];            // first row of every image (= cu_sub)
  int img_rt0[PN_MAX_IMAGES + 1];             // first row tile of every image
  PnLayer layer[PN_MAX_LAYERS];
  const unsigned char* Wfin;                  // [256][768]: final projection with the last layer's second MLP GEMM folded in
  const float* bfin;
  const float* z0;                            // [N][256] input of the signature network
  float* out;                                 // [N][256] line_desc
  float* ws;                                  // activations, pn_ws_floats(N, L) floats
  int* cnt;                                   // [n_stages][PN_MAX_RT][PN_CNT_STRIDE] arrival counters, zeroed before the launch
  unsigned* abort_word;                       // host-mapped; != 0 after a timed-out launch
  unsigned long long* stamps;                 // diagnostics (NULL normally): [block][stage][8] wall-clock ticks (lt_pairnet.h PN_STAMP_*)
};

// workspace: per layer  z (unused for layer 0) | qkv | msg | hid, then the un-normalised output of the final projection
__host__ __device__ inline int64_t pn_layer_floats(int N) { return (int64_t)N * (256 + 768 + 256 + 512); }
__host__ __device__ inline int64_t pn_ws_floats(int N, int L) { return pn_layer_floats(N) * L + (int64_t)N * 256; }
__host__ __device__ inline int pn_stages(int L) { return 4 * L + 1; }
__host__ __device__ inline int64_t pn_cnt_bytes(int L) { return (int64_t)pn_stages(L) * PN_MAX_RT * PN_CNT_STRIDE * 4; }

enum { PN_GEMM = 0, PN_ATTN = 1, PN_NORM = 2 };

struct PnStage {
  int type;
  // GEMM
  const float *A, *A2, *R, *bias;
  const unsigned char* W;
  float* Y;
  int lda, lda2, ldy, K, K1, N, act;
  int wide;                 // a unit covers 64 output columns: waves 0-3 / 4-7 take one 32-column tile each and split K four ways
  // dependency: counters of stage `dep` (-1: none), `target` arrivals per row tile (a GEMM / norm unit waits for its own row tile,
  // an attention wave for the row tile that holds its 32 keys)
  int dep, target;
  // attention / norm
  const float* qkv;
  float* msg;
};

__device__ __forceinline__ PnStage pn_decode(const PairNetArgs& a, int s) {
  PnStage st{};
  const int L = a.n_layers, N = a.N;
  const int64_t LS = pn_layer_floats(N);
  auto z_of = [&](int l) -> const float* { return l == 0 ? a.z0 : a.ws + LS * l; };
  auto qkv_of = [&](int l) -> float* { return a.ws + LS * l + (int64_t)N * 256; };
  auto msg_of = [&](int l) -> float* { return a.ws + LS * l + (int64_t)N * 1024; };
  auto hid_of = [&](int l) -> float* { return a.ws + LS * l + (int64_t)N * 1280; };
  float* fin = a.ws + LS * L;
  st.dep = -1;
  auto qkv_stage = [&](int l) {
    st.type = PN_GEMM; st.A = z_of(l); st.lda = 256; st.K = st.K1 = 256; st.N = 768; st.W = a.layer[l].Wqkv; st.bias = a.layer[l].bqkv;
    st.Y = qkv_of(l); st.ldy = 768; st.act = ACT_NONE;
    st.wide = PN_WIDE_QKV ? 1 : 0;
    if (l > 0) { st.dep = s - 1; st.target = 8; }
  };
  if (s == 0) { qkv_stage(0); return st; }
  const int t = s - 1, l = t >> 2, k = t & 3;
  if (k == 0) {            // attention of layer l: needs q/k/v of the whole image
    st.type = PN_ATTN; st.qkv = qkv_of(l); st.msg = msg_of(l); st.dep = s - 1; st.target = PN_WIDE_QKV ? 12 : 24;
  } else if (k == 1) {     // hid = relu(W1 [z ; msg] + b1)      (merge conv folded into W1)
    st.type = PN_GEMM; st.A = z_of(l); st.lda = 256; st.K1 = 256; st.A2 = msg_of(l); st.lda2 = 256; st.K = 512; st.N = 512;
    st.W = a.layer[l].W1; st.bias = a.layer[l].b1; st.Y = hid_of(l); st.ldy = 512; st.act = ACT_RELU; st.dep = s - 1; st.target = 4;
  } else if (k == 2 && l < L - 1) {   // z' = z + W2 hid + b2
    st.type = PN_GEMM; st.A = hid_of(l); st.lda = 512; st.K = st.K1 = 512; st.N = 256; st.W = a.layer[l].W2; st.bias = a.layer[l].b2;
    st.R = z_of(l); st.Y = const_cast<float*>(z_of(l + 1)); st.ldy = 256; st.act = ACT_NONE; st.dep = s - 1; st.target = 16;
  } else if (k == 2) {     // final_proj(z + W2 hid + b2) = [Wfin | Wfin W2] [z ; hid] + b
    st.type = PN_GEMM; st.A = z_of(l); st.lda = 256; st.K1 = 256; st.A2 = hid_of(l); st.lda2 = 512; st.K = 768; st.N = 256;
    st.W = a.Wfin; st.bias = a.bfin; st.Y = fin; st.ldy = 256; st.act = ACT_NONE; st.dep = s - 1; st.target = 16;
  } else if (l < L - 1) {
    qkv_stage(l + 1);
  } else {                 // F.normalize
    st.type = PN_NORM; st.qkv = fin; st.msg = a.out; st.dep = s - 1; st.target = 8;
  }
  return st;
}

// ---- write-through / L1-bypassing 16-byte accesses (compiler-visible: it places the waits) ------------------------
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pn_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7ffffff0, 0x00020000);
}
__device__ __forceinline__ f32x4 pn_load16(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, /*sc1*/ 16));
}
__device__ __forceinline__ void pn_store16(__amdgpu_buffer_rsrc_t r, int byte_off, const f32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), r, byte_off, 0, /*sc1*/ 16);
}

// (The diagnostics stamps inside the units add `(value == 12345.f)` of a just-computed register to the clock: always 0, but it ties
// the clock read behind the arithmetic it is meant to time.)
// thread 0 of the block: wait until *c >= target; false on abort / timeout.  The poll is a relaxed agent-scope load (L2-served,
// ~0.3 us apart); the clock and the host-mapped abort word (a PCIe round trip) are only looked at every few hundred polls.
__device__ __forceinline__ bool pn_poll(const int* c, int target, unsigned* abort_word) {
  if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
  long long t0 = 0;
  for (int spin = 1;; ++spin) {
    __builtin_amdgcn_s_sleep(8);
    if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    if ((spin & 511) == 0) {
      const long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return false;
      if (now - t0 > PN_TIMEOUT_TICKS) {
        __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
      }
    }
  }
}

// LDS of the block: the GEMM's wave-private staging + reduction image, or the attention's four staging regions
constexpr int PN_RS = 3 * 64 + 16;                               // W row stride in a staged K tile (bytes)
constexpr int PN_AS = 36;                                        // A row stride (floats)
constexpr int PN_A_BYTES = 32 * PN_AS * 4, PN_W_BYTES = 32 * PN_RS;
constexpr int PN_STAGE_BYTES = PN_A_BYTES + PN_W_BYTES;          // 11 264 per wave
constexpr int PN_RED_OFF = 8 * PN_STAGE_BYTES;                   // 90 112
constexpr int PN_ATL_RK = 3 * 128 + 16, PN_ATL_RV = 3 * 64 + 8;
constexpr int PN_LDS_BYTES = PN_RED_OFF + 8 * 32 * 33 * 4;       // 123 904 (attention: 8 x 12 800 V staging + 8 704 query tile; 69 632 merge image)

// ---------------------------------------------------------------------------------------------
// one 32 x 32 output tile of Y = act([A | A2] W^T + bias) (+ R); rows [row0, row0 + nrows) of the image, clamped reads
// ---------------------------------------------------------------------------------------------
// The weights of the block's NEXT unit are touched (one dword per 128-byte line, into the XCD's L2) while this unit runs, so that the
// next unit's weight loads do not start cold behind a unit that has just finished.  Compiler-visible loads whose values are
// "used" at the very end of the unit (a never-true test), i.e. after everything on the critical path.
struct PnTouch { const unsigned char* p; int lines; };
__device__ __forceinline__ void pn_touch_issue(const PnTouch& t, unsigned (&v)[2]) {
  v[0] = v[1] = 0u;
  const int i0 = threadIdx.x, i1 = threadIdx.x + PN_THREADS;
  if (t.p && i0 < t.lines) v[0] = *reinterpret_cast<const unsigned*>(t.p + (int64_t)i0 * 128);
  if (t.p && i1 < t.lines) v[1] = *reinterpret_cast<const unsigned*>(t.p + (int64_t)i1 * 128);
}
__device__ __forceinline__ void pn_touch_retire(const unsigned (&v)[2], unsigned* abort_word) {
  if ((v[0] ^ v[1]) == 0x9e3779b9u && threadIdx.x == 4095) *abort_word = 2u;    // never true; keeps the loads
}

__device__ __noinline__ bool pn_gemm_unit(const PnStage& st, int row0, int nrows, int n0_unit, const int* dep_cnt, int dep_target,
                                             unsigned* abort_word, unsigned char* lds, int* s_ok, unsigned long long* stamp, PnTouch touch) {
  constexpr int PL = 3, W_PCS = PL * 4, W_LD = (32 * W_PCS + 63) / 64, GRP = 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nk = st.K / 32;
  // narrow unit: one 32-column tile, the eight waves split K; wide unit: two tiles, four waves each
  const int grp = st.wide ? wave >> 2 : 0, nsplit = st.wide ? 4 : 8, kw = st.wide ? wave & 3 : wave;
  const int n0 = n0_unit + 32 * grp;
  const int kt0 = nk * kw / nsplit, kt1 = nk * (kw + 1) / nsplit; // this wave's K tiles (1, 2 or 3)
  unsigned char* As = lds + wave * PN_STAGE_BYTES;
  unsigned char* Ws = As + PN_A_BYTES;
  float* red = reinterpret_cast<float*>(lds + PN_RED_OFF);

  // ---- weights first: they depend on nothing, so their latency hides behind the wait for the producers
  f32x4 rw[GRP][W_LD];
  int w_lds[W_LD];
  {
    const unsigned char* Wb = st.W + (int64_t)n0 * nk * (PL * 64);
#pragma unroll
    for (int i = 0; i < W_LD; ++i) {
      int q = lane + 64 * i;
      q = q < 32 * W_PCS ? q : 32 * W_PCS - 1;
      const int r = q / W_PCS, pc = q % W_PCS;
      w_lds[i] = r * PN_RS + pc * 16;
      const unsigned char* p = Wb + (int64_t)r * nk * (PL * 64) + pc * 16;
#pragma unroll
      for (int u = 0; u < GRP; ++u)
        if (kt0 + u < kt1) rw[u][i] = *reinterpret_cast<const f32x4*>(p + (int64_t)(kt0 + u) * (PL * 64));
    }
  }
  // the epilogue's bias values, too (a cold scalar load behind the reduction would sit on the critical path)
  const int e_tile = tid >> 8, e_row = (tid & 255) >> 3, e_c4 = (tid & 7) * 4;      // epilogue: thread -> (tile, row, 4 columns)
  const bool e_on = e_tile == 0 || st.wide;
  const int e_n0 = n0_unit + 32 * e_tile;
  f32x4 rbias = f32x4{0.f, 0.f, 0.f, 0.f};
  if (st.bias && e_on) rbias = *reinterpret_cast<const f32x4*>(st.bias + e_n0 + e_c4);
  // ---- wait for the rows this tile reads
  if (tid == 0) {
    *s_ok = (dep_cnt == nullptr) || pn_poll(dep_cnt, dep_target, abort_word);
    if (stamp) stamp[1] = wall_clock64();
  }
  __syncthreads();
  if (!*s_ok) return false;
  // ---- activations: 8 lanes x 16 B per row and K tile (one 128-byte line), L1 bypassed
  const int a_r = lane >> 3, a_c = (lane & 7) * 4;
  const __amdgpu_buffer_rsrc_t rA = pn_rsrc(st.A), rA2 = pn_rsrc(st.A2 ? st.A2 : st.A);
  f32x4 ra[GRP][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = a_r + 8 * i;
    r = row0 + (r < nrows ? r : nrows - 1);
#pragma unroll
    for (int u = 0; u < GRP; ++u)
      if (kt0 + u < kt1) {
        const int k0 = (kt0 + u) * 32;
        ra[u][i] = k0 < st.K1 ? pn_load16(rA, (r * st.lda + k0 + a_c) * 4) : pn_load16(rA2, (r * st.lda2 + (k0 - st.K1) + a_c) * 4);
      }
  }
  // the residual rows of the epilogue travel with the activations (one round trip, not two)
  f32x4 rres = f32x4{0.f, 0.f, 0.f, 0.f};
  if (st.R && e_on && e_row < nrows) rres = pn_load16(pn_rsrc(st.R), ((row0 + e_row) * 256 + e_n0 + e_c4) * 4);
  unsigned tv[2];
  pn_touch_issue(touch, tv);
  const int frow = lane & 31, half = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int u = 0; u < GRP; ++u) {
    if (kt0 + u < kt1) {                                          // wave-uniform
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(As + ((a_r + 8 * i) * PN_AS + a_c) * 4) = ra[u][i];
#pragma unroll
      for (int i = 0; i < W_LD; ++i) *reinterpret_cast<f32x4*>(Ws + w_lds[i]) = rw[u][i];
      __builtin_amdgcn_wave_barrier();                            // wave-private staging: the LDS runs one wave's instructions in order
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const f32x4 x0 = *reinterpret_cast<const f32x4*>(As + (frow * PN_AS + s * 16 + half * 8) * 4);
        const f32x4 x1 = *reinterpret_cast<const f32x4*>(As + (frow * PN_AS + s * 16 + half * 8 + 4) * 4);
        unsigned a[PL], b[PL], c[PL], d[PL];
        split_pair<PL, 0>(x0[0], x0[1], a);
        split_pair<PL, 0>(x0[2], x0[3], b);
        split_pair<PL, 0>(x1[0], x1[1], c);
        split_pair<PL, 0>(x1[2], x1[3], d);
        bf16x8 af[PL], bf[PL];
#pragma unroll
        for (int p = 0; p < PL; ++p) {
          union { bf16x8 v; unsigned w[4]; } x;
          x.w[0] = a[p]; x.w[1] = b[p]; x.w[2] = c[p]; x.w[3] = d[p];
          af[p] = x.v;
          bf[p] = *reinterpret_cast<const bf16x8*>(Ws + frow * PN_RS + p * 64 + s * 32 + half * 16);
        }
#pragma unroll
        for (int ord = PL - 1; ord >= 0; --ord)
#pragma unroll
          for (int pa = PL - 1; pa >= 0; --pa) {
            const int pb = ord - pa;
            if (pb < 0 || pb >= PL) continue;
            acc = mfma_split<0>(af[pa], bf[pb], acc);
          }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (stamp && tid == 0) stamp[4] = wall_clock64() + (long long)(acc[0] == 12345.f);    // wave 0: MFMAs done
  // ---- the eight partial tiles, added in wave order
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave * (32 * 33) + ((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + frow] = acc[r];
  __syncthreads();
  if (stamp && tid == 0) stamp[5] = wall_clock64();               // all partial tiles in LDS
  if (e_on && e_row < nrows) {
    const float* rp = red + (st.wide ? e_tile * 4 : 0) * (32 * 33);     // this tile's partials: waves 4 e_tile .. (wide) or 0 .. 7
    const int np = st.wide ? 4 : 8;
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int o = e_row * 33 + e_c4 + c;
      float s = rp[o];
#pragma unroll
      for (int w = 1; w < 8; ++w)
        if (w < np) s += rp[w * (32 * 33) + o];
      v[c] = s + rbias[c];
      if (st.act == ACT_RELU) v[c] = fmaxf(v[c], 0.f);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] += rres[c];
    pn_store16(pn_rsrc(st.Y), ((row0 + e_row) * st.ldy + e_n0 + e_c4) * 4, v);
  }
  pn_touch_retire(tv, abort_word);
  return true;
}

// ---------------------------------------------------------------------------------------------
// attention of 32 queries [q0, q0 + 32) of one (image, head).  The block's EIGHT waves split the keys (wave w takes the 32-key
// chunks w, w + 8, ..: one chunk each up to 256 keys), so the critical path of a unit is one chunk + the merge.  All loads of a
// wave -- its K rows as MFMA fragments straight from global memory (lane = (key, 8-channel run): no LDS staging for K), its V
// rows, the Q fragment -- are issued together, one exposed round trip.  V goes through a wave-private transposed LDS image as
// in sig_attn_small_kernel; the eight partial (m, l, O) are merged through LDS in wave order.
// ---------------------------------------------------------------------------------------------
constexpr int PN_VT_BYTES = DH * PN_ATL_RV;                       // 12 800 per wave
__device__ __noinline__ bool pn_attn_unit(const PnStage& st, int n0, int Ni, int q0, int head, const int* dep_cnt,
                                             unsigned* abort_word, unsigned char* lds, int* s_ok, unsigned long long* stamp, PnTouch touch) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned tv[2];
  pn_touch_issue(touch, tv);
  if (tid == 0) *s_ok = 1;
  __syncthreads();
  const int h2 = lane >> 5, lq = lane & 31;
  const __amdgpu_buffer_rsrc_t rQ = pn_rsrc(st.qkv + (int64_t)n0 * 768);
  unsigned char* Vt = lds + wave * PN_VT_BYTES;
  float* Qs = reinterpret_cast<float*>(lds + 8 * PN_VT_BYTES);     // [32 q][68]: the unit's query tile, loaded once for the eight waves
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const int srow = lane >> 4, sc4 = (lane & 15) * 4;
  f32x4 kraw[8], vreg[8];
  // a wave needs the producers of ITS 32 keys only (key chunk w of an image = the image's row tile w): it starts loading as soon as
  // that tile has arrived, while other tiles of the image may still be on their way
  auto wait_tile = [&](int tile) -> bool {
    int ok = 1;
    if (lane == 0) ok = pn_poll(dep_cnt + tile * PN_CNT_STRIDE, st.target, abort_word) ? 1 : 0;
    return __builtin_amdgcn_readfirstlane(ok) != 0;
  };
  auto fetch = [&](int kv0) {
    // K: row kv0 + lq, channels s * 16 + h2 * 8 .. + 8 for the four 16-wide K steps (rows past the image repeat its last row and are
    // masked below); V: 4 rows per pass, 16 lanes x 16 B per row
    const int kr = min(kv0 + lq, Ni - 1);
    const int ko = (kr * 768 + 256 + head * DH + h2 * 8) * 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      kraw[2 * s] = pn_load16(rQ, ko + s * 64);
      kraw[2 * s + 1] = pn_load16(rQ, ko + s * 64 + 16);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kv = min(kv0 + srow + 4 * i, Ni - 1);
      vreg[i] = pn_load16(rQ, (kv * 768 + 512 + head * DH + sc4) * 4);
    }
  };
  auto stage_v = [&]() {   // V rows -> transposed planes: element (kv = r, d = sc4 + j) -> Vt[d][p][r]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = srow + 4 * i;
      unsigned a[3], b[3];
      split_pair<3>(vreg[i][0], vreg[i][1], a);
      split_pair<3>(vreg[i][2], vreg[i][3], b);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        unsigned short* col = reinterpret_cast<unsigned short*>(Vt + p * 64 + r * 2);
        col[(sc4 + 0) * (PN_ATL_RV / 2)] = (unsigned short)(a[p] & 0xffffu);
        col[(sc4 + 1) * (PN_ATL_RV / 2)] = (unsigned short)(a[p] >> 16);
        col[(sc4 + 2) * (PN_ATL_RV / 2)] = (unsigned short)(b[p] & 0xffffu);
        col[(sc4 + 3) * (PN_ATL_RV / 2)] = (unsigned short)(b[p] >> 16);
      }
    }
  };
  const bool active = wave * 32 < Ni;                             // wave-uniform
  bool okw = true;
  if (active) {
    okw = wait_tile(wave);
    if (okw) fetch(wave * 32);
  }
  // the query tile: 32 rows x 64 channels, one 16-byte piece per thread
  if (okw) okw = wait_tile(q0 >> 5);
  if (!okw && lane == 0) *s_ok = 0;
  {
    const int qr = min(q0 + (tid >> 4), Ni - 1);
    const f32x4 x = okw ? pn_load16(rQ, (qr * 768 + head * DH + (tid & 15) * 4) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(Qs + (tid >> 4) * 68 + (tid & 15) * 4) = x;
  }
  if (active && okw) stage_v();                                   // the first chunk's V (frees its registers before the query fragments)
  if (tid == 0 && stamp) stamp[1] = wall_clock64();
  __syncthreads();
  if (!*s_ok) return false;
  bf16x8 qf[4][3];
  if (active) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 x0 = *reinterpret_cast<const f32x4*>(Qs + lq * 68 + s * 16 + h2 * 8);
      f32x4 x1 = *reinterpret_cast<const f32x4*>(Qs + lq * 68 + s * 16 + h2 * 8 + 4);
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
  for (int kv0 = wave * 32; kv0 < Ni; kv0 += 256) {               // wave-uniform trip count
    if (kv0 != wave * 32) stage_v();                              // (the first chunk was staged ahead of the query tile)
    if (stamp && tid == 0) stamp[4] = wall_clock64();             // V staged (its loads have landed)
    f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      unsigned a[3], b[3], c[3], d[3];
      split_pair<3>(kraw[2 * s][0], kraw[2 * s][1], a); split_pair<3>(kraw[2 * s][2], kraw[2 * s][3], b);
      split_pair<3>(kraw[2 * s + 1][0], kraw[2 * s + 1][1], c); split_pair<3>(kraw[2 * s + 1][2], kraw[2 * s + 1][3], d);
      bf16x8 ka[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        union { bf16x8 v; unsigned w[4]; } u;
        u.w[0] = a[p]; u.w[1] = b[p]; u.w[2] = c[p]; u.w[3] = d[p];
        ka[p] = u.v;
      }
      sc = mfma_split<0>(ka[2], qf[s][0], sc);
      sc = mfma_split<0>(ka[1], qf[s][1], sc);
      sc = mfma_split<0>(ka[0], qf[s][2], sc);
      sc = mfma_split<0>(ka[1], qf[s][0], sc);
      sc = mfma_split<0>(ka[0], qf[s][1], sc);
      sc = mfma_split<0>(ka[0], qf[s][0], sc);
    }
    if (stamp && tid == 0) stamp[5] = wall_clock64() + (long long)(sc[0] == 12345.f);   // scores done
    const int kv_here = kv0;
    __builtin_amdgcn_wave_barrier();                              // wave-private LDS image: executed in order
    if (kv_here + 32 > Ni) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (kv_here + (r & 3) + 8 * (r >> 2) + 4 * h2 >= Ni) sc[r] = -INFINITY;
    }
    float mx = sc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[r]);
    mx = xor32_max(mx);
    const float m_new = fmaxf(m, mx);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sc[r] = __builtin_amdgcn_exp2f(sc[r] - m_new); ps += sc[r]; }
    ps = xor32_sum(ps);
    const float alpha = __builtin_amdgcn_exp2f(m - m_new);        // m = -inf on the first chunk -> 0
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
        for (int e = 0; e < 4; ++e) split_pair<3>(sc[8 * t + 2 * e], sc[8 * t + 2 * e + 1], w[e]);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          union { bf16x8 v; unsigned u[4]; } x;
          x.u[0] = w[0][p]; x.u[1] = w[1][p]; x.u[2] = w[2][p]; x.u[3] = w[3][p];
          pp[p] = x.v;
        }
      }
      const unsigned char* vp = Vt + lq * PN_ATL_RV + (16 * t + 4 * h2) * 2;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        bf16x8 va[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const u32x2 lo = *reinterpret_cast<const u32x2*>(vp + dt * 32 * PN_ATL_RV + p * 64);
          const u32x2 hi = *reinterpret_cast<const u32x2*>(vp + dt * 32 * PN_ATL_RV + p * 64 + 16);
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
    if (kv0 + 256 < Ni) {                                         // images of more than 256 keys: this wave's next chunk
      if (!wait_tile((kv0 + 256) >> 5)) { if (lane == 0) *s_ok = 0; break; }
      fetch(kv0 + 256);
    }
  }
  if (stamp && tid == 0) stamp[6] = wall_clock64() + (long long)(o0[0] == 12345.f);     // P V done
  // ---- merge the eight partial results: O = sum_w 2^(m_w - M) O_w / sum_w 2^(m_w - M) l_w   (wave order: deterministic)
  __syncthreads();                                                // all staging regions are dead
  if (!*s_ok) return false;
  if (stamp && tid == 0) stamp[7] = wall_clock64();               // every wave has arrived
  float* Op = reinterpret_cast<float*>(lds);                      // [8][64 d][33]
  float* ML = Op + 8 * 64 * 33;                                   // [8][2][32]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = (r & 3) + 8 * (r >> 2) + 4 * h2;
    Op[(wave * 64 + d) * 33 + lq] = o0[r];
    Op[(wave * 64 + d + 32) * 33 + lq] = o1[r];
  }
  if (h2 == 0) { ML[(wave * 2 + 0) * 32 + lq] = m; ML[(wave * 2 + 1) * 32 + lq] = l; }
  __syncthreads();
  const int oq = tid >> 4, od = (tid & 15) * 4;                   // thread -> (query, 4 consecutive d); all 512 threads
  if (q0 + oq < Ni) {
    float mw[8], M = -INFINITY, Lsum = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { mw[w] = ML[(w * 2) * 32 + oq]; M = fmaxf(M, mw[w]); }
    float wsc[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      wsc[w] = mw[w] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw[w] - M);   // a wave without any chunk contributes nothing
      Lsum += wsc[w] * ML[(w * 2 + 1) * 32 + oq];
    }
    const float inv = 1.f / Lsum;
    f32x4 res;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float acc = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) acc += wsc[w] * Op[(w * 64 + od + e) * 33 + oq];
      res[e] = acc * inv;
    }
    pn_store16(pn_rsrc(st.msg), ((n0 + q0 + oq) * D + head * DH + od) * 4, res);
  }
  pn_touch_retire(tv, abort_word);
  return true;
}

// F.normalize of 32 rows of the final projection: one wave per row, lane = 4 channels
__device__ __noinline__ bool pn_norm_unit(const PnStage& st, int row0, int nrows, const int* dep_cnt, unsigned* abort_word, int* s_ok,
                                          unsigned long long* stamp) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    *s_ok = pn_poll(dep_cnt, st.target, abort_word);
    if (stamp) stamp[1] = wall_clock64();
  }
  __syncthreads();
  if (!*s_ok) return false;
  const __amdgpu_buffer_rsrc_t rX = pn_rsrc(st.qkv);
  for (int r = wave; r < nrows; r += 8) {
    const f32x4 v = pn_load16(rX, ((row0 + r) * D + lane * 4) * 4);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) q += v[c] * v[c];
    const float nrm = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = v[c] / nrm;
    *reinterpret_cast<f32x4*>(st.msg + (int64_t)(row0 + r) * D + lane * 4) = o;
  }
  return true;
}

__device__ __forceinline__ int pn_units_per_rt(const PnStage& st) {
  return st.type == PN_GEMM ? st.N / (st.wide ? 64 : 32) : st.type == PN_ATTN ? HEADS : 1;
}

__global__ __launch_bounds__(PN_THREADS) void pair_net_kernel(const PairNetArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[PN_LDS_BYTES];
  __shared__ int s_ok;
  const int G = gridDim.x;
  const int n_stages = pn_stages(a.n_layers);
  // this block's units, stage by stage: the units of a stage are dealt round-robin, starting where the previous stage stopped
  struct Cursor { int s, u, base; };
  auto seek = [&](Cursor c) -> Cursor {
    while (c.s < n_stages) {
      const int nu = a.n_rt * pn_units_per_rt(pn_decode(a, c.s));
      if (c.u < nu) return c;
      c.base += nu;
      ++c.s;
      c.u = ((int)blockIdx.x - c.base % G + G) % G;
    }
    return c;
  };
  Cursor cur = seek(Cursor{0, (int)blockIdx.x, 0});
  while (cur.s < n_stages) {
    const int s = cur.s, u = cur.u;
    const PnStage st = pn_decode(a, s);
    // row-tile-major would put the tiles of one row band on consecutive blocks; column-major spreads a band's arrivals
    const int rt = u % a.n_rt, sub = u / a.n_rt;
    int img = 0;
    while (img + 1 < a.n_images && rt >= a.img_rt0[img + 1]) ++img;
    const int n0 = a.img_row0[img], Ni = a.img_row0[img + 1] - n0;
    const int lrow = (rt - a.img_rt0[img]) * 32;                   // first row of the tile inside its image
    const int nrows = min(32, Ni - lrow);
    const int* dep = st.dep >= 0 ? a.cnt + (int64_t)st.dep * PN_MAX_RT * PN_CNT_STRIDE : nullptr;
    // the unit after this one: its weight tile is touched while this one runs
    const Cursor nxt = seek(Cursor{s, u + G, cur.base});
    PnTouch touch{nullptr, 0};
    if (PN_TOUCH_NEXT && nxt.s < n_stages) {
      const PnStage sn = pn_decode(a, nxt.s);
      if (sn.type == PN_GEMM) {
        const int cols = sn.wide ? 64 : 32;
        touch.p = sn.W + (int64_t)(nxt.u / a.n_rt) * cols * (sn.K / 32) * 192;
        touch.lines = cols * (sn.K / 32) * 192 / 128;
      }
    }
    bool ok;
    __syncthreads();                                              // the previous unit's LDS reads are done
    unsigned long long* stamp = a.stamps ? a.stamps + ((int64_t)blockIdx.x * n_stages + s) * 8 : nullptr;
    if (stamp && threadIdx.x == 0 && stamp[0] == 0) stamp[0] = wall_clock64();            // first unit of the stage picked up
    if (st.type == PN_GEMM) ok = pn_gemm_unit(st, n0 + lrow, nrows, sub * (st.wide ? 64 : 32), dep ? dep + rt * PN_CNT_STRIDE : nullptr, st.target, a.abort_word, lds, &s_ok, stamp, touch);
    else if (st.type == PN_ATTN) ok = pn_attn_unit(st, n0, Ni, lrow, sub, dep + a.img_rt0[img] * PN_CNT_STRIDE, a.abort_word, lds, &s_ok, stamp, touch);
    else ok = pn_norm_unit(st, n0 + lrow, nrows, dep + rt * PN_CNT_STRIDE, a.abort_word, &s_ok, stamp);
    if (!ok) return;
    if (stamp && threadIdx.x == 0) stamp[2] = wall_clock64();                              // body done ([1]: dependency seen)
    if (s + 1 < n_stages) {
      // publish: every storing thread's write-through stores have left, then ONE arrival on the row tile's counter
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(a.cnt + ((int64_t)s * PN_MAX_RT + rt) * PN_CNT_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (stamp && threadIdx.x == 0) stamp[3] = wall_clock64();                              // published (last unit of the stage)
    cur = nxt;
  }
}

}  // namespace lt
