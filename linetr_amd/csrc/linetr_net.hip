// liblinetr_hip.so, translation unit 2 of 4: tokenise / forward / describe and the diagnostics entry points of the C ABI, the
// GEMM dispatcher, and every kernel of the descriptor network (csrc/lt_*.h).
#include <algorithm>
#include <numeric>

#include "lt_handle.h"
#include "lt_gemm.h"
#include "lt_gemm_split.h"
#include "lt_gemm_split16.h"
#include "lt_st_image.h"
#ifdef LINETR_EXPERIMENTS
#include "lt_gemm_st.h"
#include "lt_gemm_ro.h"
#include "lt_gemm_chain.h"
#endif
#include "lt_gemm_small.h"
#ifdef LINETR_EXPERIMENTS
#include "lt_mlp_fused.h"
#endif
#include "lt_model.h"
#include "lt_attn_fused.h"
#include "lt_gemm_ws.h"
#include "lt_tokmlp.h"
#ifdef LINETR_EXPERIMENTS
#include "lt_attn_st.h"
#endif
#include "lt_token.h"
#include "lt_bntrain.h"

using namespace lt;

namespace {

const char* gemm_class_name(const GemmArgs& g, int groups, const char* kind) {
  const char* tile;
  if (strcmp(kind, "gemm_f32") != 0) {
    static const char* tile_env = LT_XENV("LINETR_GEMM_TILE");
    tile = tile_env ? tile_env : split_tile_name(g, groups, strcmp(kind, "gemm_bf16x6") == 0 ? 3 : 2);
  } else if (g.N % 128 != 0) tile = "128x64";
  else {
    int64_t big = (int64_t)cdiv(g.M, 128) * (g.N / 128) * groups;
    tile = big >= 384 ? "128x128" : "64x128";
  }
  // LINETR_PROFILE_SHAPES=1: one profile class per GEMM shape (tuning aid)
  static const bool by_shape = LT_XENV("LINETR_PROFILE_SHAPES") != nullptr;
  static std::map<std::string, std::string> names;
  char buf[160];
  if (by_shape) snprintf(buf, sizeof buf, "%s_%s[M=%d,N=%d,K=%d,g=%d]", kind, tile, g.M, g.N, g.K, groups);
  else snprintf(buf, sizeof buf, "%s_%s", kind, tile);
  auto it = names.emplace(buf, buf).first;
  return it->second.c_str();
}

struct NormSpec {          // row normalisation that follows a [M,256] GEMM (see GemmArgs::norm)
  int mode = 0;            // 1 LayerNorm, 2 L2
  const float* gamma = nullptr;
  const float* beta = nullptr;
  const float* add2 = nullptr;   // added after the normalisation, row stride 256
  float eps = 0.f;
};

int run_gemm(LinetrHandle* h, hipStream_t st, const float* A, int lda, const float* A2, int lda2, int K1,
             const float* W, const float* bias, const float* R, int ldr, float* Y, int ldy, int M, int N,
             int K, int act, int groups = 1, int64_t gA = 0, int64_t gW = 0, int64_t gBias = 0, int64_t gY = 0,
             const NormSpec* fused_norm = nullptr) {
  GemmArgs g;
  g.A = A; g.lda = lda; g.A2 = A2; g.lda2 = lda2; g.K1 = K1;
  g.W = W; g.ldw = K; g.bias = bias; g.R = R; g.ldr = ldr; g.Y = Y; g.ldy = ldy;
  g.M = M; g.N = N; g.K = K; g.act = act;
  g.gA = gA; g.gW = gW; g.gBias = gBias; g.gY = gY;
  if (fused_norm) {
    g.norm = fused_norm->mode; g.gamma = fused_norm->gamma; g.beta = fused_norm->beta;
    g.add2 = fused_norm->add2; g.ldadd2 = D; g.eps = fused_norm->eps;
  }
  const double fl = 2.0 * M * (double)N * K * groups;
  const double by = 4.0 * groups * ((double)M * K + (double)N * K + (double)M * N);
  if (h->precision == LINETR_PREC_F32) {
    ProfScope ps(h, st, gemm_class_name(g, groups, "gemm_f32"), fl, by);
    return gemm_launch(g, groups, st);
  }
  auto it = h->split.find(W);
  if (it == h->split.end()) return fail(LINETR_E_ARG, "gemm: weight has no split-bf16 copy");
  SplitGemmArgs sa;
  sa.g = g;
#ifdef LINETR_EXPERIMENTS
  if (LT_XENV("LINETR_STREAMK")) {   // opt-in experiment (lt_gemm_split.h): the 32 MB workspace is only allocated when asked for
    if (!h->sk_ws) {
      constexpr size_t slots = 256, slot_bytes = 128 * 256 * sizeof(float);
      LT_HIP(hipMalloc((void**)&h->sk_ws, slots * slot_bytes));
      LT_HIP(hipMalloc((void**)&h->sk_flags, (slots + 1) * sizeof(unsigned)));
      LT_HIP(hipMemset(h->sk_flags, 0, (slots + 1) * sizeof(unsigned)));
      LT_HIP(hipDeviceSynchronize());
    }
    sa.sk_ws = h->sk_ws; sa.sk_flags = h->sk_flags; sa.sk_epoch = ++h->sk_epoch;
  }
#endif
  if (h->precision == LINETR_PREC_BF16X3) {
    sa.Wsp = h->split_arena + it->second.off2;
    sa.gWsp = gW * 4;
    ProfScope ps(h, st, gemm_class_name(g, groups, "gemm_bf16x3"), fl, by);
    return gemm_split_launch<2>(sa, groups, st);
  }
  if (h->precision == LINETR_PREC_F16X3) {
    sa.Wsp = h->split_arena + it->second.offh;
    sa.gWsp = gW * 4;
    ProfScope ps(h, st, gemm_class_name(g, groups, "gemm_f16x3"), fl, by);
    return gemm_split_launch<2, 1>(sa, groups, st);
  }
#ifdef LINETR_EXPERIMENTS
  // row-owner kernel (lt_gemm_ro.h): 4-wave blocks, two per CU, operands by LDS-DMA.  Measured at cfg3: 117 TF-eq against 137 for
  // the register-staged tiles (a two-slot ring leaves a DMA one K step to land, and the barrier comes every 48 MFMAs), so opt-in.
  if (LT_XENV("LINETR_GEMM_RO") != nullptr && groups == 1 && N % 256 == 0 && K % 32 == 0 && it->second.offst && lda % 4 == 0 && ldy % 4 == 0 &&
      (!A2 || (lda2 % 4 == 0 && K1 % 16 == 0)) && (!R || ldr % 4 == 0) && act != ACT_DIST && cdiv(M, 128) * (N / 256) >= 140) {
    RoGemmArgs a;
    a.A1 = A; a.lda1 = lda; a.nk1 = (A2 ? K1 : K) / 16; a.A2 = A2; a.lda2 = lda2; a.nk2 = A2 ? (K - K1) / 16 : 0;
    a.Wst = h->split_arena + it->second.offst; a.bias = bias ? bias : h->zeros; a.R = R; a.ldr = ldr; a.Y = Y; a.ldy = ldy;
    a.M = M; a.N = N; a.act = act;
    if (fused_norm) { a.norm = fused_norm->mode; a.gamma = fused_norm->gamma; a.beta = fused_norm->beta; a.add2 = fused_norm->add2; a.ldadd2 = D; a.eps = fused_norm->eps; }
    ProfScope ps(h, st, "gemm_bf16x6_ro128x256", fl, by);
    return gemm_ro_launch(a, st);
  }
#endif
  // every token row of the batch through a K = 128 layer: weights stay in registers, rows stream (lt_gemm_ws.h)
  if (groups == 1 && !A2 && !R && !fused_norm && it->second.offst && gemm_ws_fits(M, N, K, lda, ldy, act) && !LT_XENV("LINETR_NO_GEMM_WS")) {
    WsGemmArgs a;
    a.A = A; a.lda = lda; a.Wst = h->split_arena + it->second.offst; a.bias = bias ? bias : h->zeros; a.Y = Y; a.ldy = ldy; a.M = M; a.act = act;
    ProfScope ps(h, st, "gemm_bf16x6_ws64x256", fl, by);
    return gemm_ws_launch(a, st);
  }
  sa.Wsp = h->split_arena + it->second.off3;
  sa.gWsp = gW * 6;
  ProfScope ps(h, st, gemm_class_name(g, groups, "gemm_bf16x6"), fl, by);
  return gemm_split_launch<3>(sa, groups, st);
}

#ifdef LINETR_EXPERIMENTS
// ---- row-tile-local GEMM chains (lt_gemm_chain.h) ----------------------------------------------------------------
struct ChainBuilder {
  LinetrHandle* h;
  ChainArgs c;
  double flops = 0, bytes = 0;
  int err = 0;
  explicit ChainBuilder(LinetrHandle* h_) : h(h_) {}
  // Y[M,N] = norm(act(A (| A2) W^T + bias) (+ R)) (+ add2)
  void add(const float* A, int lda, const float* A2, int lda2, int K1, const float* W, const float* bias, const float* R,
           float* Y, int M, int N, int K, int act, const NormSpec* ns = nullptr) {
    if (err) return;
    if (c.n >= CHAIN_MAX) { err = fail(LINETR_E_ARG, "gemm chain: too many stages"); return; }
    auto it = h->split.find(W);
    if (it == h->split.end()) { err = fail(LINETR_E_ARG, "gemm chain: weight has no split-bf16 copy"); return; }
    SplitGemmArgs& sa = c.st[c.n++];
    sa = SplitGemmArgs{};
    GemmArgs& g = sa.g;
    g = GemmArgs{};
    g.A = A; g.lda = lda; g.A2 = A2; g.lda2 = lda2; g.K1 = K1; g.W = W; g.ldw = K; g.bias = bias; g.R = R; g.ldr = N; g.Y = Y; g.ldy = N;
    g.M = M; g.N = N; g.K = K; g.act = act;
    if (ns) { g.norm = ns->mode; g.gamma = ns->gamma; g.beta = ns->beta; g.add2 = ns->add2; g.ldadd2 = D; g.eps = ns->eps; }
    sa.Wsp = h->split_arena + it->second.off3;
    sa.wide_epi = 1;
    flops += 2.0 * M * (double)N * K;
    bytes += 4.0 * ((double)M * K + (double)N * K + (double)M * N);
  }
  int run(hipStream_t st, const char* name) {
    if (err) return err;
    ProfScope ps(h, st, name, flops, bytes);
    return gemm_chain_launch(c, st);
  }
};

// A chain is one block per 128-row tile for the WHOLE chain: worth it when the row tiles fill the chip in one round (the
// partly empty second round of a plain launch would be a whole chain long) or there are many rounds.
bool chain_wins(const LinetrHandle* h, int rows) {
  // opt-in (read per call): measured at cfg3, 2.70 vs 2.60 ms per step -- a chain keeps 199 of the 256 CUs busy for all of its
  // stages and the per-tile prologue / epilogue cost, not the launch, is what a GEMM of this size pays
  if (LT_XENV("LINETR_GEMM_CHAIN") == nullptr || h->precision != LINETR_PREC_BF16X6) return false;
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const int gy = cdiv(rows, 128);
  return (gy >= 140 && gy <= n_cu) || gy >= 4 * n_cu;
}

#endif  // LINETR_EXPERIMENTS
// Y[M,256] = norm(epi(A W^T + bias) (+ R)) (+ add2).  The split-bf16 128x256 tile owns complete rows and normalises them
// in its epilogue (one launch and one [M,256] round trip less); every other case runs the GEMM into `tmp` and then
// row_norm_kernel.  Same arithmetic either way.
int run_gemm_norm(LinetrHandle* h, hipStream_t st, const float* A, int lda, const float* A2, int lda2, int K1,
                  const float* W, const float* bias, const float* R, float* tmp, float* Y, int M, int K,
                  const NormSpec& ns) {
  const bool no_fuse = LT_XENV("LINETR_NO_FUSED_NORM") != nullptr;   // tuning / test aid (read per call)
  bool fuse = !no_fuse && h->precision != LINETR_PREC_F32 && !LT_XENV("LINETR_GEMM_TILE") && !LT_XENV("LINETR_GEMM_NARROW_EPI");
  if (fuse) {
    GemmArgs g{};
    g.M = M; g.N = D; g.K = K; g.lda = lda; g.ldy = D; g.ldr = D; g.R = R; g.A2 = A2; g.lda2 = lda2; g.K1 = K1;
    fuse = strcmp(split_tile_name(g, 1, h->precision == LINETR_PREC_BF16X6 ? 3 : 2), "128x256") == 0;
  }
  if (fuse) return run_gemm(h, st, A, lda, A2, lda2, K1, W, bias, R, D, Y, D, M, D, K, ACT_NONE, 1, 0, 0, 0, 0, &ns);
  if (int e = run_gemm(h, st, A, lda, A2, lda2, K1, W, bias, R, D, tmp, D, M, D, K, ACT_NONE)) return e;
  ProfScope ps(h, st, "row_norm", 0, (double)M * D * (ns.add2 ? 12 : 8));
  hipLaunchKernelGGL(row_norm_kernel, dim3(cdiv(M, 4)), dim3(256), 0, st, tmp, M, ns.mode == 2 ? 1 : 0, ns.gamma, ns.beta,
                     ns.add2, ns.eps, Y);
  LT_LAUNCH_CHECK();
  return 0;
}

#ifdef LINETR_EXPERIMENTS
// z' = z + W2 relu(W1 [z ; msg] + b1) + b2 in one launch (split-bf16 modes only)
int run_sig_mlp(LinetrHandle* h, hipStream_t st, const float* z, const float* msg, const SigLayer& S, float* out, int M) {
  auto i1 = h->split.find(S.W1), i2 = h->split.find(S.W2p);
  if (i1 == h->split.end() || i2 == h->split.end()) return fail(LINETR_E_ARG, "sig_mlp: weight has no split copy");
  SigMlpArgs a;
  a.z = z; a.ldz = D; a.msg = msg; a.ldm = D; a.b1 = S.b1; a.b2 = S.b2; a.out = out; a.ldo = D; a.M = M;
  const double fl = 2.0 * M * (2.0 * D * 2 * D + 2.0 * D * D), by = 4.0 * M * 3.0 * D;
  if (h->precision == LINETR_PREC_BF16X3) {
    a.W1sp = h->split_arena + i1->second.off2; a.W2sp = h->split_arena + i2->second.off2;
    ProfScope ps(h, st, "sig_mlp_bf16x3", fl, by);
    sig_mlp_fused_launch<2, 0>(a, st);
  } else if (h->precision == LINETR_PREC_F16X3) {
    a.W1sp = h->split_arena + i1->second.offh; a.W2sp = h->split_arena + i2->second.offh;
    ProfScope ps(h, st, "sig_mlp_f16x3", fl, by);
    sig_mlp_fused_launch<2, 1>(a, st);
  } else {
    a.W1sp = h->split_arena + i1->second.off3; a.W2sp = h->split_arena + i2->second.off3;
    ProfScope ps(h, st, "sig_mlp_bf16x6", fl, by);
    sig_mlp_fused_launch<3, 0>(a, st);
  }
  LT_LAUNCH_CHECK();
  return 0;
}

#endif  // LINETR_EXPERIMENTS
}  // namespace

// split-bf16 / fp16 / split-tile copies of the prepared GEMM weights (called once by linetr_create)
int lt::make_split_copies(LinetrHandle* H, const std::vector<GemmWSpec>& weights) {
  size_t total = 0;
  for (auto& w : weights) {
    LinetrHandle::SplitW sw;
    sw.rows = w.rows; sw.K = w.K;
    sw.off2 = total; total += align_up(w.rows * w.K * 4, 256);
    sw.off3 = total; total += align_up(w.rows * w.K * 6, 256);
    sw.offh = total; total += align_up(w.rows * w.K * 4, 256);
#ifdef LINETR_EXPERIMENTS
    const bool want_st = true;
#else
    const bool want_st = w.st;                          // the weights that travel by LDS-DMA or stay in registers / LDS
#endif
    if (want_st && w.rows % 16 == 0 && w.K % 32 == 0) { total = align_up(total, 1024); sw.offst = total; total += st_bytes(w.rows, w.K); }   // ST image (rows padded to 128)
    H->split[w.W] = sw;
  }
  LT_HIP(hipMalloc((void**)&H->split_arena, total));
  LT_HIP(hipMalloc((void**)&H->zeros, 4096 * sizeof(float)));
  LT_HIP(hipMemset(H->zeros, 0, 4096 * sizeof(float)));
  for (auto& kv : H->split) {
    const int64_t n4 = kv.second.rows * kv.second.K / 4;
    hipLaunchKernelGGL(split_rows_kernel<2>, dim3((unsigned)cdiv((int)n4, 256)), dim3(256), 0, 0, kv.first,
                       H->split_arena + kv.second.off2, kv.second.rows, kv.second.K);
    hipLaunchKernelGGL(split_rows_kernel<3>, dim3((unsigned)cdiv((int)n4, 256)), dim3(256), 0, 0, kv.first,
                       H->split_arena + kv.second.off3, kv.second.rows, kv.second.K);
    hipLaunchKernelGGL((split_rows_kernel<2, 1>), dim3((unsigned)cdiv((int)n4, 256)), dim3(256), 0, 0, kv.first,
                       H->split_arena + kv.second.offh, kv.second.rows, kv.second.K);
    if (kv.second.offst) {
      const int64_t thr = st_row_blocks(kv.second.rows) * (kv.second.K / 16) * 32;
      hipLaunchKernelGGL(to_st_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, 0, kv.first, kv.second.K,
                         (int)kv.second.rows, kv.second.K / 16, H->split_arena + kv.second.offst);
    }
  }
  LT_LAUNCH_CHECK();
  LT_HIP(hipDeviceSynchronize());
  return LINETR_OK;
}

// =============================================================================================
// tokenise
// =============================================================================================

namespace {
// where each image's mat_klines2sublines block goes (LinetrTokens.mat); cu_sub may be NULL for a single image
int k2s_table(const LinetrTokens& out, int n_images, int K, int N, const int32_t* h_cu_sub, K2sTable& tab, const char* who) {
  tab = K2sTable{};
  if (!out.mat) return 0;
  if (n_images > K2S_MAX_IMAGES) return fail(LINETR_E_ARG, "%s: mat_klines2sublines is written for calls of up to %d images", who, K2S_MAX_IMAGES);
  if (n_images == 1) {   // whatever image index the caller's records carry, it is this one image
    for (int i = 0; i < K2S_MAX_IMAGES; ++i) tab.img[i] = K2sImage{0, N, 0};
    return 0;
  }
  if (!out.h_cu_klines || !h_cu_sub) return fail(LINETR_E_ARG, "%s: mat_klines2sublines of several images needs the host prefix sums of key-lines and sub-lines", who);
  if (out.h_cu_klines[n_images] != K) return fail(LINETR_E_ARG, "%s: h_cu_klines does not end at K", who);
  int64_t off = 0;
  for (int i = 0; i < n_images; ++i) {
    const int ki = out.h_cu_klines[i + 1] - out.h_cu_klines[i], ni = h_cu_sub[i + 1] - h_cu_sub[i];
    if (ki < 0 || ni < 0) return fail(LINETR_E_ARG, "%s: prefix sums not monotone", who);
    tab.img[i] = K2sImage{off, ni, h_cu_sub[i]};
    off += (int64_t)ki * ni;
  }
  return 0;
}
}  // namespace

extern "C" int64_t linetr_tokenize_workspace_bytes(int32_t n_images, int32_t height, int32_t width, int32_t N) {
  const int64_t P = (int64_t)(height / 8) * (width / 8);
  return align_up(n_images * P * D * 4, 256) + align_up((int64_t)std::max(N, 1) * 4, 256) + 256;
}

extern "C" int linetr_tokenize(LinetrHandle* h, const LinetrLineRec* d_recs, int32_t K, int32_t N, double td,
                               int32_t T, const float* d_dense_desc, const float* d_dense_score, int32_t n_images,
                               int32_t height, int32_t width, int32_t clip_height, int32_t clip_width, int32_t align_corners,
                               int32_t dense_is_nhwc, LinetrTokens out, int32_t* d_sub2line, void* d_ws, int64_t ws_bytes,
                               void* stream) {
  if (K <= 0 || N <= 0) return LINETR_OK;
  if (clip_height <= 0) clip_height = height;
  if (clip_width <= 0) clip_width = width;
  const double clip_x = (double)clip_width - 0.6, clip_y = (double)clip_height - 0.6;
  if (!d_recs || !d_dense_score || !out.sublines || !out.pnt || !out.mask || !out.resp || !out.angle_sub ||
      !out.score || (out.desc && !d_dense_desc))
    return fail(LINETR_E_ARG, "tokenize: null pointer");
  if (T < 1 || T > 4096 || height % 8 || width % 8) return fail(LINETR_E_ARG, "tokenize: bad max_tokens / image size");
  if (out.mat && n_images != 1) return fail(LINETR_E_ARG, "tokenize: mat_klines2sublines of several images: use linetr_describe (it has the sub-line prefix sums)");
  K2sTable k2s;
  if (int e = k2s_table(out, n_images, K, N, nullptr, k2s, "tokenize")) return e;
  if (ws_bytes < linetr_tokenize_workspace_bytes(n_images, height, width, N))
    return fail(LINETR_E_WORKSPACE, "tokenize: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (h) LT_HIP(hipSetDevice(h->device));   // no weights involved: without a handle the current device is used
  const int Hc = height / 8, Wc = width / 8, P = Hc * Wc;
  float* nhwc = (float*)d_ws;
  int* s2l_g = (int*)((char*)d_ws + align_up((int64_t)n_images * P * D * 4, 256));
  {
    ProfScope ps(h, st, "line_fill", 0, (double)K * 80 + (double)N * 8);
    hipLaunchKernelGGL(line_fill_kernel, dim3(cdiv(K, 256)), dim3(256), 0, st, d_recs, K, clip_x, clip_y, out.klines, out.length,
                       out.angles, s2l_g, d_sub2line);
    LT_LAUNCH_CHECK();
  }
  {
    ProfScope ps(h, st, "tokenize", 0, (double)N * T * 16);
    // (with out.mat: K more blocks write the rows of mat_klines2sublines in the same launch)
    hipLaunchKernelGGL(tokenize_kernel, dim3(N + (out.mat ? K : 0)), dim3(64), 0, st, d_recs, s2l_g, N, td, T, height, width,
                       clip_x, clip_y, d_dense_score, out.sublines, out.pnt, out.mask, out.resp, out.angle_sub, out.score, (float*)nullptr,
                       (float*)nullptr, 0, (int64_t)0, out.mat, k2s);
    LT_LAUNCH_CHECK();
  }
  if (out.desc) {
    if (!dense_is_nhwc) {   // a channel-last map (the repo's producer) is sampled in place
      ProfScope ps(h, st, "nchw_to_nhwc", 0, 2.0 * n_images * P * D * 4);
      hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(P, 64), D / 64, n_images), dim3(256), 0, st, d_dense_desc,
                         nhwc, D, P);
      LT_LAUNCH_CHECK();
    }
    const int64_t ntok = (int64_t)N * T;
    ProfScope ps(h, st, "sample_desc", 0, (double)ntok * D * 4 * 2);
    hipLaunchKernelGGL(sample_desc_kernel, dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, st, out.pnt, s2l_g, d_recs,
                       ntok, T, dense_is_nhwc ? d_dense_desc : nhwc, Hc, Wc, align_corners, out.desc);
    LT_LAUNCH_CHECK();
  }
  return LINETR_OK;
}

// sample_descriptors (models/line_process.py:86-98) on its own: n points of ONE image
extern "C" int64_t linetr_sample_descriptors_workspace_bytes(int32_t Hc, int32_t Wc, int32_t dense_is_nhwc) {
  return dense_is_nhwc ? 0 : align_up((int64_t)Hc * Wc * D * 4, 256);
}

extern "C" int linetr_sample_descriptors(LinetrHandle* h, const float* d_points, int64_t n, const float* d_dense_desc, int32_t Hc,
                                         int32_t Wc, int32_t align_corners, int32_t dense_is_nhwc, float* d_out, void* d_ws,
                                         int64_t ws_bytes, void* stream) {
  if (n < 0 || Hc <= 0 || Wc <= 0) return fail(LINETR_E_ARG, "sample_descriptors: bad shape");
  if (n == 0) return LINETR_OK;
  if (!d_points || !d_dense_desc || !d_out) return fail(LINETR_E_ARG, "sample_descriptors: null pointer");
  if (ws_bytes < linetr_sample_descriptors_workspace_bytes(Hc, Wc, dense_is_nhwc) || (!dense_is_nhwc && !d_ws))
    return fail(LINETR_E_WORKSPACE, "sample_descriptors: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (h) LT_HIP(hipSetDevice(h->device));
  const int P = Hc * Wc;
  const float* nhwc = d_dense_desc;
  if (!dense_is_nhwc) {
    ProfScope ps(h, st, "nchw_to_nhwc", 0, 2.0 * P * D * 4);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(P, 64), D / 64, 1), dim3(256), 0, st, d_dense_desc, (float*)d_ws, D, P);
    LT_LAUNCH_CHECK();
    nhwc = (const float*)d_ws;
  }
  ProfScope ps(h, st, "sample_desc", 0, (double)n * D * 4 * 2);
  // T = 1 and no records: every point is its own "sub-line" of image 0
  hipLaunchKernelGGL(sample_desc_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, d_points, (const int*)nullptr,
                     (const LinetrLineRec*)nullptr, n, 1, nhwc, Hc, Wc, align_corners, d_out);
  LT_LAUNCH_CHECK();
  return LINETR_OK;
}

// =============================================================================================
// forward
// =============================================================================================

namespace {
// W2 + residual + next q/k/v projection as one contraction (SigLayer::Wnext) is taken up to this many sub-lines: the sizes at which
// its N = 1024, K = 768 product still goes to the latency kernel (small_gemm_wins) -- above, 2.4x the flops cost more than a launch
constexpr int SIG_FOLD_MAX_ROWS = 960;
struct FwdWs {
  float *a1, *a2, *a3, *a4, *pooled, *att, *fc, *o, *f1, *f2, *l1, *l2, *l3, *l4, *lpos, *zA, *zB, *qkv, *msgp, *msg, *hid;
  float *zqA, *zqB;                                // [N][4D] = [x_out | q/k/v of the next layer] (single-pair sizes: SigLayer::Wnext)
  unsigned char *zsA, *zsB, *qkvs, *msgs, *hids;   // split-tile images of the signature network's activations (experiments: lt_gemm_st.h)
  int* cu;
  char* pn;                                        // activations + arrival counters of the single-pair persistent network (lt_pairnet.h)
  int64_t total;
};
FwdWs fwd_layout(const LinetrHandle* h, int N, int64_t rows, int n_images, char* base) {
  const LinetrModelConfig& c = h->cfg;
  FwdWs w;
  int64_t off = 0;
  auto take = [&](int64_t floats) { float* p = (float*)(base + off); off += align_up(floats * 4, 256); return p; };
  w.a1 = take(rows * c.enc_channels[0]); w.a2 = take(rows * c.enc_channels[1]);
  w.a3 = take(rows * c.enc_channels[2]); w.a4 = take(rows * c.enc_channels[3]);
  w.pooled = take((int64_t)N * HEADS * POOLW);
  w.att = take((int64_t)N * D); w.fc = take((int64_t)N * D); w.o = take((int64_t)N * D);
  w.f1 = take((int64_t)N * c.d_inner); w.f2 = take((int64_t)N * D);
  w.l1 = take((int64_t)N * c.enc_channels[0]); w.l2 = take((int64_t)N * c.enc_channels[1]);
  w.l3 = take((int64_t)N * c.enc_channels[2]); w.l4 = take((int64_t)N * c.enc_channels[3]);
  w.lpos = take((int64_t)N * D);
  w.zA = take((int64_t)N * D); w.zB = take((int64_t)N * D);
  w.qkv = take((int64_t)N * 3 * D); w.msgp = take((int64_t)N * D); w.msg = take((int64_t)N * D);
  w.hid = take((int64_t)N * 2 * D);
  const int64_t nq = N <= SIG_FOLD_MAX_ROWS ? (int64_t)N * 4 * D : 0;
  w.zqA = take(nq); w.zqB = take(nq);
#ifdef LINETR_EXPERIMENTS
  auto take_st = [&](int cols) { unsigned char* p = (unsigned char*)(base + off); off += align_up(st_bytes(N, cols), 1024); return p; };
  off = align_up(off, 1024);
  w.zsA = take_st(D); w.zsB = take_st(D); w.qkvs = take_st(3 * D); w.msgs = take_st(D); w.hids = take_st(2 * D);
#else
  w.zsA = w.zsB = w.qkvs = w.msgs = w.hids = nullptr;
#endif
  w.cu = (int*)take(n_images + 1);
  w.pn = base + off;
#ifdef LINETR_EXPERIMENTS
  off += pairnet_ws_bytes(h, N);                   // 0 for batches the path does not take (too many rows)
#endif
  w.total = off;
  return w;
}
}  // namespace

extern "C" int64_t linetr_forward_workspace_bytes(const LinetrHandle* h, int32_t N, int32_t T) {
  if (!h) return -1;
  // the image count only sizes a tiny prefix-sum array; reserve for the worst case (every sub-line its own image)
  return fwd_layout(h, std::max(N, 1), (int64_t)std::max(N, 1) * T, std::max(N, 1), nullptr).total;
}

namespace {
// Cut points of a pipelined batch (linetr_describe_submit), in launch order.  Stage k = the launches between cut k - 1 and cut k, on
// stream k of the handle's pipeline; the next stage's stream waits for an event recorded at the cut.
enum {
  CUT_TOKENS = 0,     // behind the tokeniser and the layout pass
  CUT_MLP = 1,        // behind the positional encoders' MLPs
  CUT_POOL = 2,       // behind the CLS pooling + value projection
  CUT_SENTENCE = 3,   // behind the descriptive layer's tail: in front of the line-signature network
  CUT_SIG0 = 4        // CUT_SIG0 + l: behind signature layer l
};
struct PipeStages {
  hipStream_t stream[LinetrHandle::PIPE_STREAMS];
  hipEvent_t ev[LinetrHandle::PIPE_STREAMS - 1];
  int cut[LinetrHandle::PIPE_STREAMS - 1];
  int n_cuts = 0, next = 0;
};
// at position `pos` of the launch sequence: if the plan cuts here, everything that follows is queued on the next stage's stream
int pipe_boundary(PipeStages* p, int pos, hipStream_t& st) {
  if (!p || p->next >= p->n_cuts || p->cut[p->next] != pos) return LINETR_OK;
  LT_HIP(hipEventRecord(p->ev[p->next], st));
  LT_HIP(hipStreamWaitEvent(p->stream[p->next + 1], p->ev[p->next], 0));
  st = p->stream[++p->next];
  return LINETR_OK;
}

struct TokenStage {            // how the token stage (word MLP + CLS pooling) is fed
  // dense path (linetr_forward): [N,T] tensors of the reference
  const float *pnt = nullptr, *score = nullptr, *desc = nullptr;
  // fused path (linetr_describe): compact real-token list + NHWC map
  const float *cpnt = nullptr, *cscore = nullptr, *nhwc = nullptr;
  const LinetrLineRec* recs = nullptr;
  const int* sub2line_g = nullptr;
  int64_t rows = 0;            // rows of the word-MLP GEMMs (N*T dense, n_real + n_images fused)
  int64_t first_pad = 0;
  int Hc = 0, Wc = 0, align_corners = 0;
  const BnTrain* bn = nullptr; // training-time forward (linetr_forward_train): BatchNorm on batch statistics, convolutions unfolded
  PipeStages* pipe = nullptr;  // pipelined call (linetr_describe_submit): where the launch sequence moves on to the next stream
};

bool fused_mlp_enabled(const LinetrModelConfig& c) {
  static const bool off = LT_XENV("LINETR_NO_FUSED_MLP") != nullptr;   // tuning aid: the three-launch chain
  return !off && c.enc_channels[1] == 64 && c.enc_channels[2] == 128;
}

// rows handled by one wave of mlp123_kernel (multiples of the 32-row MFMA step).  The kernel holds 160 weights per lane,
// so one wave fits a SIMD (1024 on the chip) and workgroup dispatch is slow for such fat blocks (~10 blocks/us
// measured): give every wave one long run of rows -- a single round of <= 256 blocks -- rather than many short ones.
int mlp123_rows_per_wave(int64_t rows) {
  static const char* env = LT_XENV("LINETR_MLP_RPW");   // tuning aid
  if (env) return atoi(env);
  const int64_t waves = 256 * 4;
  return (int)std::max<int64_t>(cdiv((int)cdiv((int)rows, (int)waves), 32) * 32, 32);
}

#ifdef LINETR_EXPERIMENTS
// Signature network on split-tile operands: z -> [q|k|v] -> attention -> W1 [z ; message] -> W2 + z, seven times, then the
// final projection (with the last W2 folded in) and the L2 normalisation.  models/line_transformer.py:132-183, 245-246.
int sig_network_st(LinetrHandle* h, hipStream_t st, FwdWs& w, const int32_t* h_cu, const int* cu_dev, int n_images, int N,
                   int max_n, float* d_line_desc) {
  auto wst = [&](const float* W) -> const unsigned char* {
    auto it = h->split.find(W);
    return (it == h->split.end() || !it->second.offst) ? nullptr : h->split_arena + it->second.offst;
  };
  auto gemm = [&](const char* role, const unsigned char* A1, int K1, const unsigned char* A2, int K2, const float* W, const float* bias,
                  const unsigned char* R, unsigned char* Yst, float* Y, int Nout, int act) -> int {
    StGemmArgs a;
    a.A1 = A1; a.nk1 = K1 / 16; a.A2 = A2; a.nk2 = A2 ? K2 / 16 : 0;
    a.W = wst(W); a.bias = bias ? bias : h->zeros; a.R = R; a.Yst = Yst; a.Y = Y; a.ldy = D; a.M = N; a.N = Nout; a.act = act;
    if (!a.W) return fail(LINETR_E_ARG, "sig_network_st: weight has no split-tile image");
    const double K = K1 + (A2 ? K2 : 0);
    ProfScope ps(h, st, role, 2.0 * N * Nout * K, 6.0 * ((double)N * K + (double)Nout * K + (double)N * Nout));
    return gemm_st_launch(a, st);
  };
  if (st_bytes(N, 3 * D) >= (int64_t)1 << 32) return fail(LINETR_E_ARG, "sig_network_st: batch too large (q/k/v image >= 4 GiB)");
  int e;
  {
    ProfScope ps(h, st, "to_st", 0, (double)N * D * 10);
    const int64_t thr = st_row_blocks(N) * (D / 16) * 32;
    hipLaunchKernelGGL(to_st_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, st, w.zA, D, N, D / 16, w.zsA);
    LT_LAUNCH_CHECK();
  }
  unsigned char *z = w.zsA, *zn = w.zsB;
  double attn_fl = 0;
  for (int i = 0; i < n_images; ++i) { const double n = h_cu[i + 1] - h_cu[i]; attn_fl += 2.0 * 2.0 * n * n * D; }
  for (size_t l = 0; l < h->sig.size(); ++l) {
    const SigLayer& S = h->sig[l];
    if ((e = gemm("gemm_st_bf16x6_qkv", z, D, nullptr, 0, S.Wqkv, S.bqkv, nullptr, w.qkvs, nullptr, 3 * D, ACT_NONE))) return e;
    {
      ProfScope ps(h, st, "sig_attn_st", attn_fl, (double)N * D * 24);
      const bool occ1 = LT_XENV("LINETR_ATTN_ST_OCC1") != nullptr;   // tuning aid: one block per CU, 256 VGPRs
      if (occ1) hipLaunchKernelGGL(sig_attn_st_kernel<1>, dim3(n_images, HEADS, cdiv(max_n, 256)), dim3(512), 0, st, w.qkvs, cu_dev,
                                   n_images, N, w.msgs);
      else hipLaunchKernelGGL(sig_attn_st_kernel<2>, dim3(n_images, HEADS, cdiv(max_n, 256)), dim3(512), 0, st, w.qkvs, cu_dev,
                              n_images, N, w.msgs);
      LT_LAUNCH_CHECK();
    }
    if ((e = gemm("gemm_st_bf16x6_w1", z, D, w.msgs, D, S.W1, S.b1, nullptr, w.hids, nullptr, 2 * D, ACT_RELU))) return e;
    if (l + 1 == h->sig.size()) break;   // the last layer's second MLP GEMM is folded into the final projection
    if ((e = gemm("gemm_st_bf16x6_w2", w.hids, 2 * D, nullptr, 0, S.W2, S.b2, z, zn, nullptr, D, ACT_NONE))) return e;
    std::swap(z, zn);
  }
  // final_proj(z + W2 hid + b2) = [Wfin | Wfin W2] [z ; hid] + (Wfin b2 + bfin), then F.normalize
  if ((e = gemm("gemm_st_bf16x6_final", z, D, w.hids, 2 * D, h->Wfin2, h->bfin2, nullptr, nullptr, w.zB, D, ACT_NONE))) return e;
  ProfScope ps(h, st, "row_norm", 0, (double)N * D * 8);
  hipLaunchKernelGGL(row_norm_kernel, dim3(cdiv(N, 4)), dim3(256), 0, st, w.zB, N, 1, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, 0.f, d_line_desc);
  LT_LAUNCH_CHECK();
  return LINETR_OK;
}

#endif  // LINETR_EXPERIMENTS
int forward_core(LinetrHandle* h, hipStream_t st, const TokenStage& ts, const float* sublines, const float* resp,
                 const float* angle_sub, const int32_t* h_cu, const int* cu_dev, int n_images, int N, int T,
                 float* d_line_desc, FwdWs& w) {
  const LinetrModelConfig& c = h->cfg;
  int max_n = 0;
  for (int i = 0; i < n_images; ++i) max_n = std::max(max_n, h_cu[i + 1] - h_cu[i]);
  const int64_t rows = ts.rows;
  const int e0 = c.enc_channels[0], e1 = c.enc_channels[1], e2 = c.enc_channels[2], e3 = c.enc_channels[3];
  const float cx = c.norm_width / 2.f, cy = c.norm_height / 2.f;           // line_transformer.py:30-32
  const float scale = (float)std::max(c.norm_width, c.norm_height) * 0.7f;
  int e;
  // experiment (LINETR_PAIRNET=1; measured and not shipped): the whole signature network of a single pair as ONE
  // persistent launch (lt_pairnet.h); its arrival counters are zeroed here, far ahead of it on the stream
#ifdef LINETR_EXPERIMENTS
  const bool pairnet = pairnet_fits(h, n_images, N, h_cu);
  if (pairnet && (e = pairnet_prepare(h, st, N, w.pn))) return e;
#endif
  // ---- word positional encoder up to the last ReLU (a4); its final linear layer is applied after pooling
  const bool fused_mlp = fused_mlp_enabled(c);
  // layers 1-4 in one kernel (lt_tokmlp.h): large batches in the default precision, the reference's channel widths
  auto st_of = [&](const float* W) -> const unsigned char* {
    auto it = h->split.find(W);
    return it != h->split.end() && it->second.offst ? h->split_arena + it->second.offst : nullptr;
  };
  int64_t tok_mlp_min_rows = 0;      // any size: a single pair (4 k token rows, 400 sub-lines) gains too: 42 -> 31 us for the two encoders
  if (const char* v = LT_XENV("LINETR_TOKMLP_MIN_ROWS")) tok_mlp_min_rows = atoll(v);   // tuning aid (experiments build)
  const bool tok_mlp_ok = fused_mlp && h->precision == LINETR_PREC_BF16X6 && e0 == 32 && e1 == 64 && e2 == 128 && e3 == 256 &&
                          !LT_XENV("LINETR_NO_TOKMLP");
  const bool tok_mlp = tok_mlp_ok && rows >= tok_mlp_min_rows && st_of(h->wW2) && st_of(h->wW3) && st_of(h->wW4);
  const bool line_mlp = tok_mlp_ok && N >= tok_mlp_min_rows && st_of(h->lW2) && st_of(h->lW3) && st_of(h->lW4);
  TokMlpArgs amw, aml;
  if (tok_mlp) {
    amw.p0 = ts.cpnt ? ts.cpnt : ts.pnt; amw.p1 = ts.cpnt ? ts.cscore : ts.score; amw.rows = rows; amw.cx = cx; amw.cy = cy; amw.scale = scale;
    amw.W1 = h->wW1; amw.b1 = h->wb1; amw.W2st = st_of(h->wW2); amw.b2 = h->wb2; amw.W3st = st_of(h->wW3); amw.b3 = h->wb3;
    amw.W4st = st_of(h->wW4); amw.b4 = h->wb4; amw.Y = w.a4; amw.ldy = e3;
  }
  if (line_mlp) {
    aml.p0 = sublines; aml.p1 = resp; aml.p2 = angle_sub; aml.rows = N; aml.cx = cx; aml.cy = cy; aml.scale = scale;
    aml.W1 = h->lW1; aml.b1 = h->lb1; aml.W2st = st_of(h->lW2); aml.b2 = h->lb2; aml.W3st = st_of(h->lW3); aml.b3 = h->lb3;
    aml.W4st = st_of(h->lW4); aml.b4 = h->lb4; aml.Y = w.l4; aml.ldy = e3;
  }
  bool line_done = false;
  if (ts.bn) {
    // training mode (train.py:127): conv -> BatchNorm(batch statistics) -> ReLU, layer by layer, on the unfolded convolutions of a
    // bn_batch_stats handle.  Statistics run over ALL rows of the batch: B*N*T token positions (padding tokens included, as the
    // reference's [B*N, 3, T] input has them) for the word encoder, B*N sub-lines for the line encoder.
    const BnTrain& bt = *ts.bn;
    int64_t off = 0;
    auto bn = [&](int layer, float* zbuf, int64_t r, int C) {
      const int64_t o = off; off += 2 * C;
      return bn_train_layer(st, bt, zbuf, r, C, C, h->bn_g[layer], h->bn_b[layer], o);
    };
#define LT_BN(layer, zbuf, r, C) do { if ((e = bn(layer, zbuf, r, C))) return e; } while (0)
    hipLaunchKernelGGL(word_mlp1_kernel<false>, dim3((unsigned)cdiv((int)(rows * 8), 256)), dim3(256), 0, st, ts.pnt, ts.score, rows,
                       cx, cy, scale, h->wW1, h->wb1, w.a1);
    LT_LAUNCH_CHECK();
    LT_BN(0, w.a1, rows, e0);
    if ((e = run_gemm(h, st, w.a1, e0, nullptr, 0, 0, h->wW2, h->wb2, nullptr, 0, w.a2, e1, (int)rows, e1, e0, ACT_NONE))) return e;
    LT_BN(1, w.a2, rows, e1);
    if ((e = run_gemm(h, st, w.a2, e1, nullptr, 0, 0, h->wW3, h->wb3, nullptr, 0, w.a3, e2, (int)rows, e2, e1, ACT_NONE))) return e;
    LT_BN(2, w.a3, rows, e2);
    if ((e = run_gemm(h, st, w.a3, e2, nullptr, 0, 0, h->wW4, h->wb4, nullptr, 0, w.a4, e3, (int)rows, e3, e2, ACT_NONE))) return e;
    LT_BN(3, w.a4, rows, e3);
    hipLaunchKernelGGL(line_mlp1_kernel<false>, dim3(cdiv(N * 8, 256)), dim3(256), 0, st, sublines, resp, angle_sub, N, cx, cy, scale,
                       h->lW1, h->lb1, w.l1);
    LT_LAUNCH_CHECK();
    LT_BN(4, w.l1, N, e0);
    if ((e = run_gemm(h, st, w.l1, e0, nullptr, 0, 0, h->lW2, h->lb2, nullptr, 0, w.l2, e1, N, e1, e0, ACT_NONE))) return e;
    LT_BN(5, w.l2, N, e1);
    if ((e = run_gemm(h, st, w.l2, e1, nullptr, 0, 0, h->lW3, h->lb3, nullptr, 0, w.l3, e2, N, e2, e1, ACT_NONE))) return e;
    LT_BN(6, w.l3, N, e2);
    if ((e = run_gemm(h, st, w.l3, e2, nullptr, 0, 0, h->lW4, h->lb4, nullptr, 0, w.l4, e3, N, e3, e2, ACT_NONE))) return e;
    LT_BN(7, w.l4, N, e3);
#undef LT_BN
    line_done = true;
  } else
  // both encoders in ONE launch: side by side for a small batch, one after the other inside every persistent block for a large one
  if (tok_mlp && line_mlp && rows > 0 && N > 0 && !LT_XENV("LINETR_NO_DUAL_MLP")) {
    ProfScope ps(h, st, "pos_mlp_dual_bf16x6", 2.0 * rows * (3 * e0 + e0 * e1 + e1 * e2 + e2 * e3) + 2.0 * N * (5 * e0 + e0 * e1 + e1 * e2 + e2 * e3),
                 (double)rows * (12 + 4 * e3) + (double)N * (28 + 4 * e3));
    if ((e = tok_mlp_launch_dual(amw, aml, st))) return e;
    line_done = true;
  }
  if (line_done) {
  } else if (tok_mlp) {
    ProfScope ps(h, st, "tok_mlp_bf16x6", 2.0 * rows * (3 * e0 + e0 * e1 + e1 * e2 + e2 * e3), (double)rows * (12 + 4 * e3));
    if ((e = tok_mlp_launch(amw, true, st))) return e;
  } else {
  if (fused_mlp) {   // layers 1-3 in one exact-fp32 MFMA kernel (lt_model.h)
    ProfScope ps(h, st, "mlp123", 2.0 * rows * (3 * e0 + e0 * e1 + e1 * e2), (double)rows * (12 + 4 * e2));
    const int rpw = mlp123_rows_per_wave(rows);
    hipLaunchKernelGGL(mlp123_kernel<true>, dim3((unsigned)cdiv((int)cdiv((int)rows, rpw), 4)), dim3(256), 0, st,
                       ts.cpnt ? ts.cpnt : ts.pnt, ts.cpnt ? ts.cscore : ts.score, (const float*)nullptr, rows, rpw, cx, cy,
                       scale, h->wW1, h->wb1, h->wW2, h->wb2, h->wW3, h->wb3, w.a3);
    LT_LAUNCH_CHECK();
  } else {
    {
      ProfScope ps(h, st, "mlp_first", 2.0 * rows * 3 * e0, (double)rows * (12 + 4 * e0));
      hipLaunchKernelGGL(word_mlp1_kernel<true>, dim3((unsigned)cdiv((int)(rows * 8), 256)), dim3(256), 0, st,
                         ts.cpnt ? ts.cpnt : ts.pnt, ts.cpnt ? ts.cscore : ts.score, rows, cx, cy, scale, h->wW1, h->wb1, w.a1);
      LT_LAUNCH_CHECK();
    }
    if ((e = run_gemm(h, st, w.a1, e0, nullptr, 0, 0, h->wW2, h->wb2, nullptr, 0, w.a2, e1, (int)rows, e1, e0, ACT_RELU))) return e;
    if ((e = run_gemm(h, st, w.a2, e1, nullptr, 0, 0, h->wW3, h->wb3, nullptr, 0, w.a3, e2, (int)rows, e2, e1, ACT_RELU))) return e;
  }
  if ((e = run_gemm(h, st, w.a3, e2, nullptr, 0, 0, h->wW4, h->wb4, nullptr, 0, w.a4, e3, (int)rows, e3, e2, ACT_RELU))) return e;
  }
  // ---- line positional encoder
  hipStream_t ls = st;
  if (line_done) {
  } else if (line_mlp) {   // layers 1-4 in one kernel, as for the word encoder
    ProfScope ps(h, ls, "line_mlp_bf16x6", 2.0 * N * (5 * e0 + e0 * e1 + e1 * e2 + e2 * e3), (double)N * (28 + 4 * e3));
    if ((e = tok_mlp_launch(aml, false, ls))) return e;
  } else {
  if (fused_mlp) {
    ProfScope ps(h, ls, "mlp123_line", 2.0 * N * (5 * e0 + e0 * e1 + e1 * e2), (double)N * (28 + 4 * e2));
    const int rpw = mlp123_rows_per_wave(N);
    hipLaunchKernelGGL(mlp123_kernel<false>, dim3((unsigned)cdiv(cdiv(N, rpw), 4)), dim3(256), 0, ls, sublines, resp, angle_sub,
                       (int64_t)N, rpw, cx, cy, scale, h->lW1, h->lb1, h->lW2, h->lb2, h->lW3, h->lb3, w.l3);
    LT_LAUNCH_CHECK();
  } else {
    {
      ProfScope ps(h, ls, "mlp_first", 2.0 * N * 5 * e0, (double)N * (28 + 4 * e0));
      hipLaunchKernelGGL(line_mlp1_kernel<true>, dim3(cdiv(N * 8, 256)), dim3(256), 0, ls, sublines, resp, angle_sub, N, cx, cy,
                         scale, h->lW1, h->lb1, w.l1);
      LT_LAUNCH_CHECK();
    }
    if ((e = run_gemm(h, ls, w.l1, e0, nullptr, 0, 0, h->lW2, h->lb2, nullptr, 0, w.l2, e1, N, e1, e0, ACT_RELU))) return e;
    if ((e = run_gemm(h, ls, w.l2, e1, nullptr, 0, 0, h->lW3, h->lb3, nullptr, 0, w.l3, e2, N, e2, e1, ACT_RELU))) return e;
  }
  if ((e = run_gemm(h, ls, w.l3, e2, nullptr, 0, 0, h->lW4, h->lb4, nullptr, 0, w.l4, e3, N, e3, e2, ACT_RELU))) return e;
  }
  if ((e = run_gemm(h, ls, w.l4, e3, nullptr, 0, 0, h->lW5, h->lb5, nullptr, 0, w.lpos, D, N, D, e3, ACT_NONE))) return e;
  if ((e = pipe_boundary(ts.pipe, CUT_MLP, st))) return e;
  // ---- CLS-row attention pooling + value/last-MLP projection
  if (ts.cpnt) {
#ifdef LINETR_EXPERIMENTS
    static const bool two_pass = LT_XENV("LINETR_POOL_TWO_PASS") != nullptr;  // tuning aid: the LDS two-pass variant
#else
    constexpr bool two_pass = false;
#endif
    if (!two_pass) {
      // algorithmic bytes: the dense map once (or the four taps of every token, whichever is less), one a4 row per token, the pooled rows out
      const double tap_bytes = std::min((double)n_images * ts.Hc * ts.Wc * D * 4, (double)rows * D * 4 * 4);
      ProfScope ps(h, st, "cls_pool_online", 2.0 * rows * (2.0 * HEADS * D * 2), tap_bytes + (double)rows * D * 4 + (double)N * HEADS * POOLW * 4);
      // few sub-lines (a single pair): four waves per sub-line, so that the chip is covered and the token chain is a quarter as long
      if (N <= 2048 && !LT_XENV("LINETR_POOL_NO_SPLIT"))
        hipLaunchKernelGGL(cls_pool_online_kernel<4>, dim3(N), dim3(256), 0, st, ts.recs, ts.sub2line_g, ts.cpnt,
                           w.a4, ts.first_pad, N, T, ts.nhwc, ts.Hc, ts.Wc, ts.align_corners, h->pool, w.pooled, 0);
      else
        hipLaunchKernelGGL(cls_pool_online_kernel<1>, dim3(cdiv(N, 4)), dim3(256), 0, st, ts.recs, ts.sub2line_g, ts.cpnt,
                           w.a4, ts.first_pad, N, T, ts.nhwc, ts.Hc, ts.Wc, ts.align_corners, h->pool, w.pooled,
                           LT_XENV("LINETR_POOL_FORWARD") ? 0 : 1);
    }
#ifdef LINETR_EXPERIMENTS
    else {
      ProfScope ps(h, st, "cls_pool_fused", 2.0 * rows * (2.0 * HEADS * D * 2), (double)rows * D * 4 * 5);
      const size_t lds = ((size_t)(T + 1) * D + HEADS * (T + 2)) * sizeof(float);
      if (lds > 160 * 1024) return fail(LINETR_E_ARG, "describe: max_tokens too large for the fused pooling kernel");
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(cls_pool_fused_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(cls_pool_fused_kernel, dim3(N), dim3(256), lds, st, ts.recs, ts.sub2line_g, ts.cpnt, w.a4,
                         ts.first_pad, T, ts.nhwc, ts.Hc, ts.Wc, ts.align_corners, h->pool, w.pooled);
    }
#endif
    LT_LAUNCH_CHECK();
  } else {
    ProfScope ps(h, st, "cls_pool", 2.0 * rows * (2.0 * HEADS * D * 2), (double)rows * D * 8);
    hipLaunchKernelGGL(cls_pool_kernel, dim3(N), dim3(256), HEADS * (T + 1) * sizeof(float), st, ts.desc, w.a4, T,
                       h->pool, w.pooled);
    LT_LAUNCH_CHECK();
  }
  if ((e = run_gemm(h, st, w.pooled, HEADS * POOLW, nullptr, 0, 0, h->Watt, h->batt, nullptr, 0, w.att, D, N, DH, POOLW,
                    ACT_NONE, HEADS, POOLW, (int64_t)DH * POOLW, DH, DH))) return e;
  if ((e = pipe_boundary(ts.pipe, CUT_POOL, st))) return e;
#ifdef LINETR_EXPERIMENTS
  const bool chain = chain_wins(h, N) && !h->sig.empty();
#else
  constexpr bool chain = false;
#endif
  float *z = w.zA, *zn = w.zB;
#ifdef LINETR_EXPERIMENTS
  if (chain) {
    // [fc + LN] -> [w_1, GELU] -> [w_2 + residual + LN (+ line position)] -> [q/k/v of signature layer 0]: one launch
    ChainBuilder cb(h);
    NormSpec ns1; ns1.mode = 1; ns1.gamma = h->ln1g; ns1.beta = h->ln1b; ns1.eps = 1e-6f;
    NormSpec ns2; ns2.mode = 1; ns2.gamma = h->ln2g; ns2.beta = h->ln2b; ns2.add2 = w.lpos; ns2.eps = 1e-6f;
    cb.add(w.att, D, nullptr, 0, 0, h->Wfc, h->bfc, nullptr, w.o, N, D, D, ACT_NONE, &ns1);
    cb.add(w.o, D, nullptr, 0, 0, h->Wf1, h->bf1, nullptr, w.f1, N, c.d_inner, D, ACT_GELU);
    cb.add(w.f1, c.d_inner, nullptr, 0, 0, h->Wf2, h->bf2, w.o, w.zA, N, D, c.d_inner, ACT_NONE, &ns2);
    cb.add(w.zA, D, nullptr, 0, 0, h->sig[0].Wqkv, h->sig[0].bqkv, nullptr, w.qkv, N, 3 * D, D, ACT_NONE);
    if ((e = cb.run(st, "gemm_chain_bf16x6_cls"))) return e;
  } else
#endif
  {
  {  // o = LN(fc(att) + cls)  (line_attention.py:36-40; the CLS residual sits in the bias)
    NormSpec ns; ns.mode = 1; ns.gamma = h->ln1g; ns.beta = h->ln1b; ns.eps = 1e-6f;
    if ((e = run_gemm_norm(h, st, w.att, D, nullptr, 0, 0, h->Wfc, h->bfc, nullptr, w.fc, w.o, N, D, ns))) return e;
  }
  if ((e = run_gemm(h, st, w.o, D, nullptr, 0, 0, h->Wf1, h->bf1, nullptr, 0, w.f1, c.d_inner, N, c.d_inner, D, ACT_GELU))) return e;
  {  // sentence = line_pos + LN(w_2(gelu(w_1 o)) + o)  (line_attention.py:79-83, line_transformer.py:128)
    NormSpec ns; ns.mode = 1; ns.gamma = h->ln2g; ns.beta = h->ln2b; ns.add2 = w.lpos; ns.eps = 1e-6f;
    if ((e = run_gemm_norm(h, st, w.f1, c.d_inner, nullptr, 0, 0, h->Wf2, h->bf2, w.o, w.f2, w.zA, N, c.d_inner, ns))) return e;
  }
  }
  // ---- line signature network
  if ((e = pipe_boundary(ts.pipe, CUT_SENTENCE, st))) return e;
#ifdef LINETR_EXPERIMENTS
  if (pairnet && !chain) return pairnet_run(h, st, w.zA, d_line_desc, h_cu, n_images, N, w.pn);
#endif
  // LINETR_SIG_PATH=st (experiment): activations stay in HBM as split-tile images and every K step travels by LDS-DMA
  // (lt_gemm_st.h, lt_attn_st.h).  Measured at cfg3 on one box: the ST GEMMs are 5-7 % faster than the register-staged
  // ones in isolation, but inside the step the 6-byte activations cost more at the kernel boundaries (the L2 write-back of
  // 273 MB instead of 182 MB of fresh activations per layer) than the main loops save: 2.92 vs 2.69 ms per step.
#ifdef LINETR_EXPERIMENTS
  const char* sig_path = LT_XENV("LINETR_SIG_PATH");      // read per call (tests switch it)
  if (!chain && h->precision == LINETR_PREC_BF16X6 && !h->sig.empty() && sig_path && !strcmp(sig_path, "st"))
    return sig_network_st(h, st, w, h_cu, cu_dev, n_images, N, max_n, d_line_desc);
#endif
  const int qtiles = cdiv(max_n, ATT_QT);
  const int64_t sig_bn_off = 4 * (int64_t)(e0 + e1 + e2 + e3);   // the signature layers' slots behind the two encoders' in the packed statistics
  // layers but the last: W1 -> ReLU -> W2 + residual in one kernel, hidden activations in registers (lt_mlp_fused.h)
#ifdef LINETR_EXPERIMENTS
  const bool fused_sig_mlp = h->precision != LINETR_PREC_F32 && N >= 4096 && !LT_XENV("LINETR_NO_FUSED_SIG_MLP") &&
                             LT_XENV("LINETR_FUSED_SIG_MLP") != nullptr;   // opt-in: measured slower
#endif
  // single-pair sizes (the 32-query attention below is taken): x_out and the NEXT layer's q/k/v come out of ONE contraction over
  // [z ; hid] (SigLayer::Wnext) -- one dependent launch less per layer where a launch costs more than its flops
  const bool small_attn = h->precision != LINETR_PREC_F32 && !LT_XENV("LINETR_ATTN_F32") && !LT_XENV("LINETR_ATTN_4WAVE") &&
                          !LT_XENV("LINETR_NO_SMALL_ATTN") && (int64_t)n_images * HEADS * cdiv(max_n, 256) < 64;
  const bool fold_next = !chain && small_attn && N <= SIG_FOLD_MAX_ROWS && !LT_XENV("LINETR_NO_SIG_FOLD");
  const float* qkv_in = w.qkv;     // where the current layer's q/k/v sit, and their row stride
  int ldq = 3 * D, ldz = D;        // (z's row stride: D, or 4 D when z is the head of a [x_out | q/k/v] row)
  float *zq = w.zqA, *zq_next = w.zqB;
  const float* zc = z;             // the layer's input rows
  for (size_t l = 0; l < h->sig.size(); ++l) {
    const SigLayer& S = h->sig[l];
    if (fold_next) {
      if (l == 0) {
        if ((e = run_gemm(h, st, zc, ldz, nullptr, 0, 0, S.Wqkv, S.bqkv, nullptr, 0, w.qkv, 3 * D, N, 3 * D, D, ACT_NONE))) return e;
      }
      {
        double fl = 0;
        for (int i = 0; i < n_images; ++i) { double n = h_cu[i + 1] - h_cu[i]; fl += 2.0 * 2.0 * n * n * D; }
        ProfScope ps(h, st, "sig_attn_bf16x6", fl, (double)N * D * 16);
        hipLaunchKernelGGL(sig_attn_small_kernel, dim3(n_images, HEADS, cdiv(max_n, 32)), dim3(256), 0, st, qkv_in, cu_dev, w.msgp, ldq);
        LT_LAUNCH_CHECK();
      }
      if ((e = run_gemm(h, st, zc, ldz, w.msgp, D, D, S.W1, S.b1, nullptr, 0, w.hid, 2 * D, N, 2 * D, 2 * D, ts.bn ? ACT_NONE : ACT_RELU))) return e;
      if (ts.bn && (e = bn_train_layer(st, *ts.bn, w.hid, N, 2 * D, 2 * D, h->bn_g[8 + l], h->bn_b[8 + l], sig_bn_off + (int64_t)l * 4 * D))) return e;
      if (l + 1 == h->sig.size()) break;
      if ((e = run_gemm(h, st, zc, ldz, w.hid, 2 * D, D, S.Wnext, S.bnext, nullptr, 0, zq, 4 * D, N, 4 * D, 3 * D, ACT_NONE))) return e;
      zc = zq; ldz = 4 * D; qkv_in = zq + D; ldq = 4 * D;
      std::swap(zq, zq_next);
      if ((e = pipe_boundary(ts.pipe, CUT_SIG0 + (int)l, st))) return e;
      continue;
    }
    // q/k/v projection + attention of an (image, head) in one launch (lt_attn_fused.h): images of up to 256 sub-lines, and
    // enough (image, head) blocks to fill the chip; q, k, v never reach HBM
    const bool no_fqa = LT_XENV("LINETR_NO_FUSED_QKV_ATTN") != nullptr;     // A/B switch (experiments build; read per call)
    if (!chain && !no_fqa && h->precision == LINETR_PREC_BF16X6 && max_n <= 256 && (int64_t)n_images * HEADS >= 128) {
      auto it = h->split.find(S.Wqkv);
      if (it == h->split.end() || !it->second.offst) return fail(LINETR_E_ARG, "signature layer: q/k/v weight has no split-tile image");
      double fl = 2.0 * N * 3.0 * D * D;
      for (int i = 0; i < n_images; ++i) { double n = h_cu[i + 1] - h_cu[i]; fl += 2.0 * 2.0 * n * n * D; }
      static unsigned long long attr_done = 0;
      const unsigned long long dev_bit = current_device_bit();
      if (!(attr_done & dev_bit)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sig_qkv_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FQA_LDS);
        attr_done |= dev_bit;
      }
      ProfScope ps(h, st, "sig_qkv_attn_bf16x6", fl, (double)N * D * 8);
      hipLaunchKernelGGL(sig_qkv_attn_kernel, dim3(n_images, HEADS), dim3(512), FQA_LDS, st, z, h->split_arena + it->second.offst,
                         S.bqkv, cu_dev, w.msgp);
      LT_LAUNCH_CHECK();
    } else {
    if (!chain)
      if ((e = run_gemm(h, st, z, D, nullptr, 0, 0, S.Wqkv, S.bqkv, nullptr, 0, w.qkv, 3 * D, N, 3 * D, D, ACT_NONE))) return e;
    {
      double fl = 0;
      for (int i = 0; i < n_images; ++i) { double n = h_cu[i + 1] - h_cu[i]; fl += 2.0 * 2.0 * n * n * D; }
      // exact-fp32 MFMA attention in f32 mode, fp32-faithful split-bf16 (6 products) otherwise
      static const bool force_f32_attn = LT_XENV("LINETR_ATTN_F32") != nullptr;
      if (h->precision == LINETR_PREC_F32 || force_f32_attn) {
        ProfScope ps(h, st, "sig_attn", fl, (double)N * D * 16);
        hipLaunchKernelGGL(sig_attn_kernel, dim3(n_images, HEADS, qtiles), dim3(256), 0, st, w.qkv, cu_dev, w.msgp);
      } else {
        ProfScope ps(h, st, "sig_attn_bf16x6", fl, (double)N * D * 16);
        static const bool attn4 = LT_XENV("LINETR_ATTN_4WAVE") != nullptr;   // tuning aid: 128-query blocks
        // few (image, head) pairs: 64-query blocks, so that a single pair still spreads over 32 CUs instead of 8
        static const bool no_small_attn = LT_XENV("LINETR_NO_SMALL_ATTN") != nullptr;   // tuning aid
        // few (image, head) pairs: 32-query blocks whose 4 waves also split the KV range (a single pair spreads over 56 CUs
        // and the critical path is 2 KV chunks instead of 7)
        if (!attn4 && !no_small_attn && (int64_t)n_images * HEADS * cdiv(max_n, 256) < 64)
          // (r04: an eight-wave form -- one key chunk per wave, K fragments straight from global memory -- measured level with this
          // one, 12.6 us per launch for a cfg2 pair either way, and was not kept)
          hipLaunchKernelGGL(sig_attn_small_kernel, dim3(n_images, HEADS, cdiv(max_n, 32)), dim3(256), 0, st, w.qkv, cu_dev, w.msgp, 3 * D);
        else if (attn4 || max_n <= 128)
          hipLaunchKernelGGL(sig_attn_split_kernel<4>, dim3(n_images, HEADS, qtiles), dim3(256), 0, st, w.qkv, cu_dev, w.msgp);
        else
          hipLaunchKernelGGL(sig_attn_split_kernel<8>, dim3(n_images, HEADS, cdiv(max_n, 256)), dim3(512), 0, st, w.qkv, cu_dev,
                             w.msgp);
      }
      LT_LAUNCH_CHECK();
    }
    }
#ifdef LINETR_EXPERIMENTS
    if (chain) {
      ChainBuilder cb(h);
      const bool w1_alone = LT_XENV("LINETR_CHAIN_W1_ALONE") != nullptr;     // A/B: W1 as its own launch (all 256 CUs)
      if (w1_alone && l + 1 < h->sig.size()) {
        if ((e = run_gemm(h, st, z, D, w.msgp, D, D, S.W1, S.b1, nullptr, 0, w.hid, 2 * D, N, 2 * D, 2 * D, ACT_RELU))) return e;
      } else
      cb.add(z, D, w.msgp, D, D, S.W1, S.b1, nullptr, w.hid, N, 2 * D, 2 * D, ACT_RELU);
      if (l + 1 < h->sig.size()) {
        // W1 -> W2 + residual -> the NEXT layer's q/k/v projection
        cb.add(w.hid, 2 * D, nullptr, 0, 0, S.W2, S.b2, z, zn, N, D, 2 * D, ACT_NONE);
        cb.add(zn, D, nullptr, 0, 0, h->sig[l + 1].Wqkv, h->sig[l + 1].bqkv, nullptr, w.qkv, N, 3 * D, D, ACT_NONE);
        if ((e = cb.run(st, "gemm_chain_bf16x6_sig"))) return e;
        std::swap(z, zn);
        continue;
      }
      // last layer: W1 -> [final projection with W2 folded in] -> L2 normalisation
      NormSpec nl2; nl2.mode = 2;
      cb.add(z, D, w.hid, 2 * D, D, h->Wfin2, h->bfin2, nullptr, d_line_desc, N, D, 3 * D, ACT_NONE, &nl2);
      if ((e = cb.run(st, "gemm_chain_bf16x6_final"))) return e;
      return LINETR_OK;
    }
    if (fused_sig_mlp && l + 1 < h->sig.size()) {
      if ((e = run_sig_mlp(h, st, z, w.msgp, S, zn, N))) return e;
      std::swap(z, zn);
      continue;
    }
#endif
    if ((e = run_gemm(h, st, z, D, w.msgp, D, D, S.W1, S.b1, nullptr, 0, w.hid, 2 * D, N, 2 * D, 2 * D, ts.bn ? ACT_NONE : ACT_RELU))) return e;
    if (ts.bn && (e = bn_train_layer(st, *ts.bn, w.hid, N, 2 * D, 2 * D, h->bn_g[8 + l], h->bn_b[8 + l], sig_bn_off + (int64_t)l * 4 * D))) return e;
    if (l + 1 == h->sig.size()) break;   // the last layer's second MLP GEMM is folded into the final projection below
    if ((e = run_gemm(h, st, w.hid, 2 * D, nullptr, 0, 0, S.W2, S.b2, z, D, zn, D, N, D, 2 * D, ACT_NONE))) return e;
    std::swap(z, zn);
    if ((e = pipe_boundary(ts.pipe, CUT_SIG0 + (int)l, st))) return e;
  }
  NormSpec l2; l2.mode = 2;      // F.normalize(final_proj(.), dim=1)  (line_transformer.py:245-246)
  if (h->sig.empty()) {
    if ((e = run_gemm_norm(h, st, z, D, nullptr, 0, 0, h->Wfin, h->bfin, nullptr, zn, d_line_desc, N, D, l2))) return e;
  } else {
    // final_proj(z + W2 hid + b2) = [Wfin | Wfin W2] [z ; hid] + (Wfin b2 + bfin): one K = 768 GEMM instead of two launches
    if (!fold_next) { zc = z; ldz = D; }
    if ((e = run_gemm_norm(h, st, zc, ldz, w.hid, 2 * D, D, h->Wfin2, h->bfin2, nullptr, zn, d_line_desc, N, 3 * D, l2))) return e;
  }
  return LINETR_OK;
}

int check_cu(const int32_t* h_cu, int n_images) {
  if (!h_cu || n_images < 1) return fail(LINETR_E_ARG, "null / empty cu_sub");
  if (h_cu[0] != 0) return fail(LINETR_E_ARG, "cu_sub[0] != 0");
  for (int i = 0; i < n_images; ++i)
    if (h_cu[i + 1] < h_cu[i]) return fail(LINETR_E_ARG, "cu_sub not monotone");
  return 0;
}
}  // namespace

extern "C" int linetr_forward(LinetrHandle* h, const LinetrTokens* tok, const int32_t* h_cu, const int32_t* d_cu,
                              int32_t n_images, int32_t T, float* d_line_desc, void* d_ws, int64_t ws_bytes,
                              void* stream) {
  if (!h || !tok) return fail(LINETR_E_ARG, "forward: null argument");
  if (h->cfg.bn_batch_stats) return fail(LINETR_E_ARG, "forward: a training-mode handle (bn_batch_stats = 1) runs linetr_forward_train only");
  if (int e = check_cu(h_cu, n_images)) return e;
  const int N = h_cu[n_images];
  if (N <= 0) return LINETR_OK;
  if (!tok->sublines || !tok->pnt || !tok->resp || !tok->angle_sub || !tok->desc || !tok->score || !d_line_desc)
    return fail(LINETR_E_ARG, "forward: null tensor");
  if (ws_bytes < linetr_forward_workspace_bytes(h, N, T)) return fail(LINETR_E_WORKSPACE, "forward: workspace too small");
  if ((int64_t)N * T > INT32_MAX / 8) return fail(LINETR_E_ARG, "forward: batch too large");
  hipStream_t st = (hipStream_t)stream;
  LT_HIP(hipSetDevice(h->device));
  FwdWs w = fwd_layout(h, N, (int64_t)N * T, std::max(N, 1), (char*)d_ws);
  const int* cu_dev = d_cu;
  if (!cu_dev) {
    LT_HIP(hipMemcpyAsync(w.cu, h_cu, (n_images + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    cu_dev = w.cu;
  }
  TokenStage ts;
  ts.pnt = tok->pnt; ts.score = tok->score; ts.desc = tok->desc; ts.rows = (int64_t)N * T;
  return forward_core(h, st, ts, tok->sublines, tok->resp, tok->angle_sub, h_cu, cu_dev, n_images, N, T, d_line_desc, w);
}

// Training-time forward (train.py:127,163-164): linetr_forward's dense token path with BatchNorm on batch
// statistics (lt_bntrain.h).  The workspace is linetr_forward's plus the statistics scratch.
extern "C" int64_t linetr_bn_stats_floats(const LinetrHandle* h) {
  if (!h) return -1;
  const LinetrModelConfig& c = h->cfg;
  return 4 * (int64_t)(c.enc_channels[0] + c.enc_channels[1] + c.enc_channels[2] + c.enc_channels[3]) + (int64_t)c.n_sig_layers * 4 * D;
}

extern "C" int64_t linetr_forward_train_workspace_bytes(const LinetrHandle* h, int32_t N, int32_t T) {
  if (!h) return -1;
  return linetr_forward_workspace_bytes(h, N, T) + (int64_t)BN_MAX_BLOCKS * 2 * BN_MAX_CHANNELS * 8 + 2 * BN_MAX_CHANNELS * 4 + 512;
}

extern "C" int linetr_forward_train(LinetrHandle* h, const LinetrTokens* tok, const int32_t* h_cu, const int32_t* d_cu, int32_t n_images,
                                    int32_t T, float momentum, float* d_bn_running, float* d_bn_batch, float* d_line_desc, void* d_ws,
                                    int64_t ws_bytes, void* stream) {
  if (!h || !tok) return fail(LINETR_E_ARG, "forward_train: null argument");
  if (!h->cfg.bn_batch_stats) return fail(LINETR_E_ARG, "forward_train: the handle was created for inference (bn_batch_stats = 0)");
  if (!d_bn_running) return fail(LINETR_E_ARG, "forward_train: null running statistics");
  if (!(momentum >= 0.f && momentum <= 1.f)) return fail(LINETR_E_ARG, "forward_train: momentum out of [0, 1]");
  if (int e = check_cu(h_cu, n_images)) return e;
  const int N = h_cu[n_images];
  if (N <= 0) return LINETR_OK;
  if (!tok->sublines || !tok->pnt || !tok->resp || !tok->angle_sub || !tok->desc || !tok->score || !d_line_desc)
    return fail(LINETR_E_ARG, "forward_train: null tensor");
  if (ws_bytes < linetr_forward_train_workspace_bytes(h, N, T)) return fail(LINETR_E_WORKSPACE, "forward_train: workspace too small");
  if ((int64_t)N * T > INT32_MAX / 8) return fail(LINETR_E_ARG, "forward_train: batch too large");
  hipStream_t st = (hipStream_t)stream;
  LT_HIP(hipSetDevice(h->device));
  FwdWs w = fwd_layout(h, N, (int64_t)N * T, std::max(N, 1), (char*)d_ws);
  BnTrain bt;
  bt.running = d_bn_running; bt.batch = d_bn_batch; bt.momentum = momentum;
  bt.partial = (double*)((char*)d_ws + align_up(w.total, 256));
  bt.affine = (float*)((char*)bt.partial + (int64_t)BN_MAX_BLOCKS * 2 * BN_MAX_CHANNELS * 8);
  const int* cu_dev = d_cu;
  if (!cu_dev) {
    LT_HIP(hipMemcpyAsync(w.cu, h_cu, (n_images + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    cu_dev = w.cu;
  }
  TokenStage ts;
  ts.pnt = tok->pnt; ts.score = tok->score; ts.desc = tok->desc; ts.rows = (int64_t)N * T;
  ts.bn = &bt;
  return forward_core(h, st, ts, tok->sublines, tok->resp, tok->angle_sub, h_cu, cu_dev, n_images, N, T, d_line_desc, w);
}

// =============================================================================================
// fused tokenise + describe
// =============================================================================================

namespace {
struct DescWs { float *nhwc, *cpnt, *cscore, *sublines, *resp, *angle_sub; int* s2l_g; int64_t fwd_off, total; };
DescWs desc_layout(int n_images, int height, int width, int N, int64_t rows, char* base) {
  DescWs d;
  int64_t off = 0;
  auto take = [&](int64_t bytes) { char* p = base + off; off += align_up(bytes, 256); return p; };
  const int64_t P = (int64_t)(height / 8) * (width / 8);
  d.nhwc = (float*)take(n_images * P * D * 4);
  d.cpnt = (float*)take(rows * 8);
  d.cscore = (float*)take(rows * 4);
  d.sublines = (float*)take((int64_t)N * 16);
  d.resp = (float*)take((int64_t)N * 4);
  d.angle_sub = (float*)take((int64_t)N * 8);
  d.s2l_g = (int*)take((int64_t)N * 4);
  d.fwd_off = off;
  d.total = off;
  return d;
}
}  // namespace

extern "C" int64_t linetr_describe_workspace_bytes(const LinetrHandle* h, int32_t n_images, int32_t height, int32_t width,
                                                   int32_t N, int64_t n_real) {
  if (!h) return -1;
  const int64_t rows = n_real + n_images;
  return desc_layout(n_images, height, width, std::max(N, 1), rows, nullptr).total +
         fwd_layout(h, std::max(N, 1), rows, n_images, nullptr).total;
}

namespace {
// linetr_describe proper.  pipe != nullptr (linetr_describe_submit): the launch sequence moves from stream to stream at the plan's
// cut points; `st` is the first stage's stream.
int describe_impl(LinetrHandle* h, const LinetrLineRec* d_recs, int32_t K, int32_t N, int64_t n_real,
                  const int32_t* h_cu, const int32_t* d_cu, int32_t n_images, double td, int32_t T,
                  const float* d_dense_desc, const float* d_dense_score, int32_t height, int32_t width,
                  int32_t align_corners, int32_t dense_is_nhwc, const LinetrTokens& out, int32_t* d_sub2line,
                  float* d_line_desc, void* d_ws, int64_t ws_bytes, hipStream_t st, PipeStages* pipe) {
  if (!h) return fail(LINETR_E_ARG, "describe: null handle");
  if (h->cfg.bn_batch_stats) return fail(LINETR_E_ARG, "describe: a training-mode handle (bn_batch_stats = 1) runs linetr_forward_train only");
  if (int e = check_cu(h_cu, n_images)) return e;
  if (h_cu[n_images] != N) return fail(LINETR_E_ARG, "describe: cu_sub does not end at N");
  if (K <= 0 || N <= 0) return LINETR_OK;
  if (!d_recs || !d_dense_desc || !d_dense_score || !d_line_desc || !d_ws) return fail(LINETR_E_ARG, "describe: null pointer");
  if (T < 1 || T > 4096 || height % 8 || width % 8) return fail(LINETR_E_ARG, "describe: bad max_tokens / image size");
  if (n_real < N || n_real > (int64_t)N * T) return fail(LINETR_E_ARG, "describe: implausible real-token count");
  if (out.desc && !out.pnt) return fail(LINETR_E_ARG, "describe: out.desc requires out.pnt");
  K2sTable k2s;
  if (int e = k2s_table(out, n_images, K, N, h_cu, k2s, "describe")) return e;
  if (ws_bytes < linetr_describe_workspace_bytes(h, n_images, height, width, N, n_real))
    return fail(LINETR_E_WORKSPACE, "describe: workspace too small");
  const int64_t rows = n_real + n_images;
  if (rows > INT32_MAX / 8) return fail(LINETR_E_ARG, "describe: batch too large");
  LT_HIP(hipSetDevice(h->device));
  DescWs dw = desc_layout(n_images, height, width, N, rows, (char*)d_ws);
  FwdWs w = fwd_layout(h, N, rows, n_images, (char*)d_ws + dw.fwd_off);
  const int Hc = height / 8, Wc = width / 8, P = Hc * Wc;
  const int* cu_dev = d_cu;
  if (!cu_dev) {
    LT_HIP(hipMemcpyAsync(w.cu, h_cu, (n_images + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    cu_dev = w.cu;
  }
  float* sublines = out.sublines ? out.sublines : dw.sublines;
  float* resp = out.resp ? out.resp : dw.resp;
  float* angle_sub = out.angle_sub ? out.angle_sub : dw.angle_sub;
  const float* nhwc_map = dense_is_nhwc ? d_dense_desc : dw.nhwc;
  {
    ProfScope ps(h, st, "line_fill", 0, (double)K * 80 + (double)N * 8);
    hipLaunchKernelGGL(line_fill_kernel, dim3(cdiv(K, 256)), dim3(256), 0, st, d_recs, K, (double)width - 0.6,
                       (double)height - 0.6, out.klines, out.length, out.angles, dw.s2l_g, d_sub2line);
    LT_LAUNCH_CHECK();
  }
  {
    ProfScope ps(h, st, "tokenize", 0, (double)n_real * 16);
    // (the last cdiv(n_images, 64) blocks write the per-image padding rows of the compact token list)
    hipLaunchKernelGGL(tokenize_kernel, dim3(N + cdiv(n_images, 64) + (out.mat ? K : 0)), dim3(64), 0, st, d_recs, dw.s2l_g, N, td, T,
                       height, width, (double)width - 0.6, (double)height - 0.6, d_dense_score, sublines, out.pnt, out.mask, resp,
                       angle_sub, out.score, dw.cpnt, dw.cscore,
                       n_images, (int64_t)n_real, out.mat, k2s);
    LT_LAUNCH_CHECK();
  }
  if (!dense_is_nhwc) {
    ProfScope ps(h, st, "nchw_to_nhwc", 0, 2.0 * n_images * P * D * 4);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(cdiv(P, 64), D / 64, n_images), dim3(256), 0, st, d_dense_desc,
                       dw.nhwc, D, P);
    LT_LAUNCH_CHECK();
  }
  if (out.desc) {  // the reference's dense tensor was asked for as well
    const int64_t ntok = (int64_t)N * T;
    ProfScope ps(h, st, "sample_desc", 0, (double)ntok * D * 4 * 2);
    hipLaunchKernelGGL(sample_desc_kernel, dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, st, out.pnt, dw.s2l_g, d_recs,
                       ntok, T, nhwc_map, Hc, Wc, align_corners, out.desc);
    LT_LAUNCH_CHECK();
  }
  TokenStage ts;
  ts.cpnt = dw.cpnt; ts.cscore = dw.cscore; ts.nhwc = nhwc_map; ts.recs = d_recs; ts.sub2line_g = dw.s2l_g;
  ts.rows = rows; ts.first_pad = n_real; ts.Hc = Hc; ts.Wc = Wc; ts.align_corners = align_corners;
  ts.pipe = pipe;
  if (int e = pipe_boundary(pipe, CUT_TOKENS, st)) return e;
  return forward_core(h, st, ts, sublines, resp, angle_sub, h_cu, cu_dev, n_images, N, T, d_line_desc, w);
}

// the streams and events of the describe pipeline, made at the first submit: all or nothing
int pipe_ready(LinetrHandle* h) {
  LinetrHandle::Pipe& p = h->pipe;
  if (p.stream[0]) return LINETR_OK;
  if (p.failed) return fail(LINETR_E_HIP, "describe_submit: the pipeline's streams could not be created");
  constexpr int NS = LinetrHandle::PIPE_STREAMS, NL = LinetrHandle::PIPE_SLOTS, NE = NL * (NS + 1);
  hipStream_t s[NS] = {};
  hipEvent_t ev[NE] = {};
  bool ok = true;
  for (int i = 0; ok && i < NS; ++i) ok = hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; ok && i < NE; ++i) ok = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    for (hipStream_t x : s) if (x) (void)hipStreamDestroy(x);
    (void)hipGetLastError();
    p.failed = true;
    return fail(LINETR_E_HIP, "describe_submit: the pipeline's streams could not be created");
  }
  for (int i = 0; i < NS; ++i) p.stream[i] = s[i];
  for (int l = 0, k = 0; l < NL; ++l) {
    p.fork[l] = ev[k++]; p.done[l] = ev[k++];
    for (int c = 0; c < NS - 1; ++c) p.cut[l][c] = ev[k++];
  }
  return LINETR_OK;
}

// How a batch is cut into stages when the caller keeps `n_slots` batches in flight.
struct PipePlan { int n_cuts = 0; int cut[LinetrHandle::PIPE_STREAMS - 1] = {0, 0, 0}; };
PipePlan pipe_plan(const LinetrHandle* h, int n_slots) {
  PipePlan pl;
  // measured on MI355X (profiles/r06_pipeline_sweep.txt): balanced stages win over "front | signature network" (the front is a third
  // of a batch), and three stages over two; a fourth slot only lets the host run one more batch ahead
  const int n_sig = (int)h->sig.size();
  if (n_slots == 2 || n_sig < 5) { pl.n_cuts = 1; pl.cut[0] = n_sig >= 3 ? CUT_SIG0 + 2 : CUT_SENTENCE; }
  else { pl.n_cuts = 2; pl.cut[0] = CUT_SIG0; pl.cut[1] = CUT_SIG0 + 3; }
  return pl;
}
}  // namespace

extern "C" int linetr_describe(LinetrHandle* h, const LinetrLineRec* d_recs, int32_t K, int32_t N, int64_t n_real,
                               const int32_t* h_cu, const int32_t* d_cu, int32_t n_images, double td, int32_t T,
                               const float* d_dense_desc, const float* d_dense_score, int32_t height, int32_t width,
                               int32_t align_corners, int32_t dense_is_nhwc, LinetrTokens out, int32_t* d_sub2line,
                               float* d_line_desc, void* d_ws, int64_t ws_bytes, void* stream) {
  return describe_impl(h, d_recs, K, N, n_real, h_cu, d_cu, n_images, td, T, d_dense_desc, d_dense_score, height, width, align_corners,
                       dense_is_nhwc, out, d_sub2line, d_line_desc, d_ws, ws_bytes, (hipStream_t)stream, nullptr);
}

extern "C" int32_t linetr_pipeline_max_slots(void) { return LinetrHandle::PIPE_SLOTS; }

extern "C" int linetr_describe_submit(LinetrHandle* h, const LinetrLineRec* d_recs, int32_t K, int32_t N, int64_t n_real,
                                      const int32_t* h_cu, const int32_t* d_cu, int32_t n_images, double td, int32_t T,
                                      const float* d_dense_desc, const float* d_dense_score, int32_t height, int32_t width,
                                      int32_t align_corners, int32_t dense_is_nhwc, LinetrTokens out, int32_t* d_sub2line,
                                      float* d_line_desc, void* d_ws, int64_t ws_bytes, int32_t slot, int32_t n_slots, void* stream) {
  if (!h) return fail(LINETR_E_ARG, "describe_submit: null handle");
  if (n_slots < 1 || n_slots > LinetrHandle::PIPE_SLOTS || slot < 0 || slot >= n_slots)
    return fail(LINETR_E_ARG, "describe_submit: need 0 <= slot < n_slots <= %d", LinetrHandle::PIPE_SLOTS);
  LT_HIP(hipSetDevice(h->device));
  if (int e = pipe_ready(h)) return e;
  LinetrHandle::Pipe& p = h->pipe;
  hipStream_t st = (hipStream_t)stream;
  // host run-ahead is bounded to the batches in flight: the slot's previous batch (n_slots submits ago) must have left the GPU
  // before its workspace is handed to the device again
  if (p.submitted[slot]) LT_HIP(hipEventSynchronize(p.done[slot]));
  const PipePlan plan = pipe_plan(h, n_slots);
  PipeStages stg;
  for (int i = 0; i < LinetrHandle::PIPE_STREAMS; ++i) stg.stream[i] = p.stream[i];
  stg.n_cuts = plan.n_cuts;
  for (int i = 0; i < plan.n_cuts; ++i) { stg.cut[i] = plan.cut[i]; stg.ev[i] = p.cut[slot][i]; }
  // fork: the first stage starts behind everything the caller has queued so far (the upload of d_recs, the producer of the maps)
  LT_HIP(hipEventRecord(p.fork[slot], st));
  LT_HIP(hipStreamWaitEvent(stg.stream[0], p.fork[slot], 0));
  const int e = describe_impl(h, d_recs, K, N, n_real, h_cu, d_cu, n_images, td, T, d_dense_desc, d_dense_score, height, width,
                              align_corners, dense_is_nhwc, out, d_sub2line, d_line_desc, d_ws, ws_bytes, stg.stream[0], &stg);
  if (e) {   // leave the streams idle and the slot free: a failed submit must not leave half a batch behind
    for (hipStream_t x : p.stream) (void)hipStreamSynchronize(x);
    p.submitted[slot] = false;
    return e;
  }
  // the stream of the last stage reached (an empty batch queues nothing: the event then completes with what that stream already holds)
  LT_HIP(hipEventRecord(p.done[slot], stg.stream[stg.next]));
  p.submitted[slot] = true;
  return LINETR_OK;
}

extern "C" int linetr_describe_join(LinetrHandle* h, int32_t slot, void* stream) {
  if (!h) return fail(LINETR_E_ARG, "describe_join: null handle");
  if (slot < 0 || slot >= LinetrHandle::PIPE_SLOTS) return fail(LINETR_E_ARG, "describe_join: bad slot");
  if (!h->pipe.stream[0] || !h->pipe.submitted[slot]) return fail(LINETR_E_ARG, "describe_join: nothing was submitted to this slot");
  LT_HIP(hipStreamWaitEvent((hipStream_t)stream, h->pipe.done[slot], 0));
  return LINETR_OK;
}

extern "C" int linetr_debug_posenc(LinetrHandle* h, int32_t which, const float* d_in0, const float* d_in1,
                                      const float* d_in2, int64_t rows, float* d_out, void* stream) {
  if (!h || !d_in0 || !d_in1 || !d_out || (which == 1 && !d_in2) || which < 0 || which > 1)
    return fail(LINETR_E_ARG, "debug_posenc: bad argument");
  if (!fused_mlp_enabled(h->cfg)) return fail(LINETR_E_ARG, "debug_posenc: needs keyline_encoder [32,64,128,256]");
  if (rows <= 0) return LINETR_OK;
  LT_HIP(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  const LinetrModelConfig& c = h->cfg;
  const float cx = c.norm_width / 2.f, cy = c.norm_height / 2.f;
  const float scale = (float)std::max(c.norm_width, c.norm_height) * 0.7f;
  const int rpw = mlp123_rows_per_wave(rows);
  const dim3 grid((unsigned)cdiv((int)cdiv((int)rows, rpw), 4));
  if (which == 0)
    hipLaunchKernelGGL(mlp123_kernel<true>, grid, dim3(256), 0, st, d_in0, d_in1, (const float*)nullptr, rows, rpw, cx, cy, scale,
                       h->wW1, h->wb1, h->wW2, h->wb2, h->wW3, h->wb3, d_out);
  else
    hipLaunchKernelGGL(mlp123_kernel<false>, grid, dim3(256), 0, st, d_in0, d_in1, d_in2, rows, rpw, cx, cy, scale, h->lW1, h->lb1,
                       h->lW2, h->lb2, h->lW3, h->lb3, d_out);
  LT_LAUNCH_CHECK();
  return LINETR_OK;
}

extern "C" int linetr_debug_gemm(LinetrHandle* h, const float* A, int32_t lda, const float* W, const float* bias,
                                 const float* R, float* Y, int32_t ldy, int32_t M, int32_t N, int32_t K, int32_t act,
                                 int32_t cache_weights, void* stream) {
  if (lda < K || ldy < N || lda % 4 || ldy % 4) return fail(LINETR_E_ARG, "debug_gemm: bad leading dimension");
  if (!h || !A || !W || !Y) return fail(LINETR_E_ARG, "debug_gemm: null argument");
  if (K % 32 || N % 64) return fail(LINETR_E_ARG, "debug_gemm: N %% 64 == 0 and K %% 32 == 0 required");
  LT_HIP(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  if (h->precision == LINETR_PREC_F32) return run_gemm(h, st, A, lda, nullptr, 0, 0, W, bias, R, ldy, Y, ldy, M, N, K, act);
  // caller-provided weights: split them into a scratch buffer (optionally cached by pointer)
  auto it = h->debug_split.find(W);
  unsigned char* buf = it != h->debug_split.end() ? it->second : nullptr;
  const int64_t b2 = align_up((int64_t)N * K * 4, 256), b3 = align_up((int64_t)N * K * 6, 256);
  const bool ws = N == WS_N && K == WS_K;             // the weight-stationary kernel's shape: it reads a split-tile image
  const int64_t bst = ws ? align_up(st_bytes(N, K), 256) : 0, off_st = align_up(b2 + b3 + b2, 1024);
  if (!buf) {
    LT_HIP(hipMalloc((void**)&buf, off_st + bst));
    const int64_t n4 = (int64_t)N * K / 4;
    hipLaunchKernelGGL(split_rows_kernel<2>, dim3((unsigned)cdiv((int)n4, 256)), dim3(256), 0, st, W, buf, (int64_t)N, K);
    hipLaunchKernelGGL(split_rows_kernel<3>, dim3((unsigned)cdiv((int)n4, 256)), dim3(256), 0, st, W, buf + b2, (int64_t)N, K);
    hipLaunchKernelGGL((split_rows_kernel<2, 1>), dim3((unsigned)cdiv((int)n4, 256)), dim3(256), 0, st, W, buf + b2 + b3, (int64_t)N, K);
    if (ws) {
      const int64_t thr = st_row_blocks(N) * (K / 16) * 32;
      hipLaunchKernelGGL(to_st_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, st, W, K, N, K / 16, buf + off_st);
    }
    LT_LAUNCH_CHECK();
    if (cache_weights) h->debug_split[W] = buf;
  }
  h->split[W] = {0, (size_t)b2, N, K, (size_t)(b2 + b3), ws ? (size_t)off_st : 0};
  unsigned char* keep = h->split_arena;
  h->split_arena = buf;  // the lookup inside run_gemm resolves relative to split_arena
  int e = run_gemm(h, st, A, lda, nullptr, 0, 0, W, bias, R, ldy, Y, ldy, M, N, K, act);
  h->split_arena = keep;
  h->split.erase(W);
  if (!cache_weights) {
    (void)hipStreamSynchronize(st);
    (void)hipFree(buf);
  }
  return e;
}

#ifdef LINETR_EXPERIMENTS
// ---- split-tile (ST) format and GEMM (lt_gemm_st.h), for the unit tests and micro-benchmarks
extern "C" int64_t linetr_st_bytes(int64_t rows, int32_t K) { return (K % 16 || rows < 0) ? -1 : st_bytes(rows, K); }

extern "C" int linetr_debug_to_st(LinetrHandle* h, const float* d_X, int32_t ld, int32_t rows, int32_t K, void* d_st,
                                  void* stream) {
  if (!h || !d_X || !d_st || K % 16 || ld < K || ld % 4 || rows < 1) return fail(LINETR_E_ARG, "debug_to_st: bad argument");
  LT_HIP(hipSetDevice(h->device));
  const int64_t thr = st_row_blocks(rows) * (K / 16) * 32;
  hipLaunchKernelGGL(to_st_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_X, ld, rows, K / 16,
                     (unsigned char*)d_st);
  LT_LAUNCH_CHECK();
  return LINETR_OK;
}

extern "C" int linetr_debug_from_st(LinetrHandle* h, const void* d_st, int32_t rows, int32_t K, float* d_X, int32_t ld,
                                    void* stream) {
  if (!h || !d_X || !d_st || K % 16 || ld < K || ld % 4 || rows < 1) return fail(LINETR_E_ARG, "debug_from_st: bad argument");
  LT_HIP(hipSetDevice(h->device));
  const int64_t thr = st_row_blocks(rows) * (K / 16) * 32;
  hipLaunchKernelGGL(from_st_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)d_st, rows, K / 16, d_X, ld);
  LT_LAUNCH_CHECK();
  return LINETR_OK;
}

extern "C" int linetr_debug_gemm_st(LinetrHandle* h, const void* d_A1, int32_t K1, const void* d_A2, int32_t K2,
                                    const void* d_W, const float* d_bias, const void* d_R, void* d_Yst, float* d_Y,
                                    int32_t ldy, int32_t M, int32_t N, int32_t act, void* stream) {
  if (!h || !d_A1 || !d_W || (!d_Yst && !d_Y) || K1 % 16 || K2 % 16 || N > 4096)
    return fail(LINETR_E_ARG, "debug_gemm_st: bad argument");
  LT_HIP(hipSetDevice(h->device));
  StGemmArgs a;
  a.A1 = (const unsigned char*)d_A1; a.nk1 = K1 / 16;
  a.A2 = (const unsigned char*)d_A2; a.nk2 = d_A2 ? K2 / 16 : 0;
  a.W = (const unsigned char*)d_W; a.bias = d_bias ? d_bias : h->zeros; a.R = (const unsigned char*)d_R;
  a.Yst = (unsigned char*)d_Yst; a.Y = d_Y; a.ldy = ldy; a.M = M; a.N = N; a.act = act;
  return gemm_st_launch(a, (hipStream_t)stream);
}

#endif  // LINETR_EXPERIMENTS
