// liblinetr_hip.so, translation unit 1 of 4: lifetime (float64 weight preparation), host pre-filter, the collective entry point
// and the profiling entry points of the C ABI declared in include/linetr_hip.h.  No device code lives here.
#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <numeric>
#include <thread>

#include "lt_handle.h"
#ifdef LINETR_EXPERIMENTS
#include "lt_mlp_fused.h"   // sig_mlp_kperm
#endif

using namespace lt;

namespace {

// Persistent host worker pool (the batched pre-filter used to create and join 7 std::threads per call).
// Leaked on purpose: the workers are detached and live until process exit, so there is no static-destruction order
// problem when the library is unloaded from an interpreter that is shutting down.
class WorkPool {
 public:
  static WorkPool& get() {
    static WorkPool* p = new WorkPool();
    return *p;
  }
  int size() const { return n_workers_ + 1; }
  // runs fn(0..n-1), the calling thread takes part; one parallel region at a time
  void run(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (n == 1 || n_workers_ == 0) { for (int i = 0; i < n; ++i) fn(i); return; }
    std::lock_guard<std::mutex> region(region_);
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = &fn; n_jobs_ = n; next_ = 0; pending_ = n; ++gen_;
    }
    cv_work_.notify_all();
    drain();
    std::unique_lock<std::mutex> lk(m_);
    cv_done_.wait(lk, [&] { return pending_ == 0; });
    job_ = nullptr;
  }

 private:
  WorkPool() {
    // one process per GPU: the ranks of a node share its cores (LOCAL_WORLD_SIZE is set by torch.distributed.run);
    // LINETR_HOST_THREADS overrides (0 = run the pre-filter on the calling thread)
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    if (const char* lw = getenv("LOCAL_WORLD_SIZE")) hw /= (unsigned)std::max(1, atoi(lw));
    n_workers_ = (int)std::min(15u, hw > 1 ? hw / 2 : 0u);
    if (const char* ht = getenv("LINETR_HOST_THREADS")) n_workers_ = std::max(0, std::min(63, atoi(ht) - 1));
    for (int i = 0; i < n_workers_; ++i) std::thread([this] { loop(); }).detach();
  }
  void drain() {
    for (;;) {
      int i;
      const std::function<void(int)>* f;
      {
        std::lock_guard<std::mutex> lk(m_);
        if (!job_ || next_ >= n_jobs_) return;
        i = next_++;
        f = job_;
      }
      (*f)(i);
      std::lock_guard<std::mutex> lk(m_);
      if (--pending_ == 0) cv_done_.notify_all();
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_work_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
      }
      drain();
    }
  }
  std::mutex m_, region_;
  std::condition_variable cv_work_, cv_done_;
  const std::function<void(int)>* job_ = nullptr;
  int n_jobs_ = 0, next_ = 0, pending_ = 0, n_workers_ = 0;
  uint64_t gen_ = 0;
};

// ---- float64 weight preparation ---------------------------------------------------------------

struct TensorMap {
  std::map<std::string, std::pair<const float*, int64_t>> t;
  const float* get(const std::string& k, int64_t numel, int& err) const {
    auto it = t.find(k);
    if (it == t.end() || it->second.first == nullptr) {
      err = fail(LINETR_E_WEIGHTS, "state_dict tensor '%s' missing", k.c_str());
      return nullptr;
    }
    if (it->second.second != numel) {
      err = fail(LINETR_E_WEIGHTS, "state_dict tensor '%s' has %lld elements, expected %lld", k.c_str(),
                 (long long)it->second.second, (long long)numel);
      return nullptr;
    }
    return it->second.first;
  }
};

struct Arena {
  std::vector<float> host;
  size_t put(const std::vector<double>& v) {
    size_t off = (host.size() + 63) / 64 * 64;
    host.resize(off + v.size());
    for (size_t i = 0; i < v.size(); ++i) host[off + i] = (float)v[i];
    return off;
  }
};

// Conv1d(k=1)+BatchNorm1d(eval) -> one affine map (models/line_transformer.py:9-20)
void fold_bn(const float* W, const float* b, const float* g, const float* beta, const float* mean,
             const float* var, int out, int in, std::vector<double>& Wf, std::vector<double>& bf) {
  Wf.resize((size_t)out * in);
  bf.resize(out);
  for (int o = 0; o < out; ++o) {
    const double s = (double)g[o] / std::sqrt((double)var[o] + 1e-5);
    for (int i = 0; i < in; ++i) Wf[(size_t)o * in + i] = (double)W[(size_t)o * in + i] * s;
    bf[o] = ((double)b[o] - (double)mean[o]) * s + (double)beta[o];
  }
}

std::vector<double> to_d(const float* p, size_t n) { return std::vector<double>(p, p + n); }

}  // namespace

// =============================================================================================
// lifetime
// =============================================================================================

extern "C" int linetr_abi_version(void) { return LINETR_ABI_VERSION; }
extern "C" const char* linetr_last_error(void) { return g_err.c_str(); }

extern "C" int linetr_create(const LinetrModelConfig* cfg, int32_t n_tensors, const char* const* names,
                             const float* const* h_data, const int64_t* numel, int32_t device,
                             LinetrHandle** out) {
  if (!cfg || !out || !names || !h_data || !numel) return fail(LINETR_E_ARG, "null argument");
  if (cfg->d_model != D || cfg->n_heads != HEADS)
    return fail(LINETR_E_ARG, "only descriptor_dim=256 / n_heads=4 are supported");
  if (cfg->d_inner % 128 != 0 || cfg->n_sig_layers < 0 || cfg->n_desc_layers < 1)
    return fail(LINETR_E_ARG, "bad d_inner / layer counts");
  const int e0 = cfg->enc_channels[0], e1 = cfg->enc_channels[1], e2 = cfg->enc_channels[2], e3 = cfg->enc_channels[3];
  if (e0 != 32 || e1 % 64 || e2 % 64 || e3 % 64 || e3 != D)
    return fail(LINETR_E_ARG, "keyline_encoder must be [32, 64k, 64k, 256] (got %d,%d,%d,%d)", e0, e1, e2, e3);
  int ndev = 0;
  LT_HIP(hipGetDeviceCount(&ndev));
  if (ndev <= 0 || device >= ndev) return fail(LINETR_E_HIP, "no usable HIP device (count=%d)", ndev);
  LT_HIP(hipSetDevice(device));

  TensorMap tm;
  for (int i = 0; i < n_tensors; ++i) tm.t[names[i]] = {h_data[i], numel[i]};
  int err = 0;
  Arena ar;
  auto H = std::make_unique<LinetrHandle>();
  H->cfg = *cfg;
  H->device = device;
  struct Fix { const float** dst; size_t off; };
  std::vector<Fix> fix;
  auto place = [&](const float** dst, const std::vector<double>& v) { fix.push_back({dst, ar.put(v)}); };
  struct GemmW { const float** dst; int64_t rows; int K; bool st; };
  std::vector<GemmW> gemm_w;
  // st: the weight also gets a split-tile image (lt_st_image.h) -- the q/k/v projections, which the fused projection +
  // attention kernel streams by LDS-DMA (lt_attn_fused.h); in the experiments build every eligible weight gets one
  auto place_w = [&](const float** dst, const std::vector<double>& v, int64_t rows, int K, bool st = false) {
    place(dst, v);
    gemm_w.push_back({dst, rows, K, st});
  };

  // training-mode handle: gamma / beta of the 8 + n_sig_layers BatchNorm layers (the vectors are sized up front: place() keeps
  // pointers to their elements until the arena is uploaded)
  int bn_slot = 0;
  if (cfg->bn_batch_stats) {
    // the statistics scratch of linetr_forward_train is sized for BN_MAX_CHANNELS channels per BatchNorm layer
    if (std::max(std::max(e0, e1), std::max(std::max(e2, e3), 2 * D)) > lt::BN_MAX_CHANNELS)
      return fail(LINETR_E_ARG, "create: a training-mode handle (bn_batch_stats = 1) takes keyline_encoder widths up to %d (got %d, %d, %d, %d)",
                  lt::BN_MAX_CHANNELS, e0, e1, e2, e3);
    H->bn_g.reserve(8 + cfg->n_sig_layers); H->bn_b.reserve(8 + cfg->n_sig_layers); H->bn_c.reserve(8 + cfg->n_sig_layers);
  }
  // ---- positional encoders: 4 x (conv + BN + ReLU) + linear ------------------------------------
  const int ch_w[6] = {3, e0, e1, e2, e3, D}, ch_l[6] = {5, e0, e1, e2, e3, D};
  std::vector<double> W5w, b5w;  // last (linear) layer of the word encoder, consumed algebraically
  for (int enc = 0; enc < 2; ++enc) {
    const std::string pre = enc == 0 ? "klenc.word_position_enc.encoder." : "klenc.line_position_enc.encoder.";
    const int* ch = enc == 0 ? ch_w : ch_l;
    const float** Wdst[4] = {enc == 0 ? &H->wW1 : &H->lW1, enc == 0 ? &H->wW2 : &H->lW2,
                             enc == 0 ? &H->wW3 : &H->lW3, enc == 0 ? &H->wW4 : &H->lW4};
    const float** bdst[4] = {enc == 0 ? &H->wb1 : &H->lb1, enc == 0 ? &H->wb2 : &H->lb2,
                             enc == 0 ? &H->wb3 : &H->lb3, enc == 0 ? &H->wb4 : &H->lb4};
    for (int i = 0; i < 4; ++i) {
      const std::string c = pre + std::to_string(3 * i), bn = pre + std::to_string(3 * i + 1);
      const float* W = tm.get(c + ".weight", (int64_t)ch[i + 1] * ch[i], err);
      const float* b = tm.get(c + ".bias", ch[i + 1], err);
      const float* g = tm.get(bn + ".weight", ch[i + 1], err);
      const float* be = tm.get(bn + ".bias", ch[i + 1], err);
      const float* mu = tm.get(bn + ".running_mean", ch[i + 1], err);
      const float* va = tm.get(bn + ".running_var", ch[i + 1], err);
      if (err) return err;
      std::vector<double> Wf, bf;
      if (cfg->bn_batch_stats) {     // training-mode handle: the convolution as it is, BatchNorm applied by lt_bntrain.h
        Wf = to_d(W, (size_t)ch[i + 1] * ch[i]); bf = to_d(b, ch[i + 1]);
        H->bn_g.push_back(nullptr); H->bn_b.push_back(nullptr); H->bn_c.push_back(ch[i + 1]);
        place(&H->bn_g[bn_slot], to_d(g, ch[i + 1])); place(&H->bn_b[bn_slot], to_d(be, ch[i + 1]));
        ++bn_slot;
      } else
      fold_bn(W, b, g, be, mu, va, ch[i + 1], ch[i], Wf, bf);
      // layers 2-4 also get split-tile images: the one-kernel MLP of lt_tokmlp.h keeps layers 2 / 3 in LDS and layer 4 in registers as
      // such, and the weight-stationary GEMM of lt_gemm_ws.h reads layer 4's planes from one
      if (i == 0) place(Wdst[i], Wf); else place_w(Wdst[i], Wf, ch[i + 1], ch[i], true);
      place(bdst[i], bf);
    }
    const float* W = tm.get(pre + "12.weight", (int64_t)D * e3, err);
    const float* b = tm.get(pre + "12.bias", D, err);
    if (err) return err;
    if (enc == 0) { W5w = to_d(W, (size_t)D * D); b5w = to_d(b, D); }
    else { place_w(&H->lW5, to_d(W, (size_t)D * D), D, e3); place(&H->lb5, to_d(b, D)); }
  }

  // ---- line-descriptive layer: only the last one matters (line_transformer.py:123-125) ----------
  {
    const std::string p = "klenc.desc_layers." + std::to_string(cfg->n_desc_layers - 1) + ".";
    const float* cls = tm.get("klenc.cls_token", D, err);
    const float* Wq = tm.get(p + "slf_attn.w_qs.weight", D * D, err);
    const float* bq = tm.get(p + "slf_attn.w_qs.bias", D, err);
    const float* Wk = tm.get(p + "slf_attn.w_ks.weight", D * D, err);
    const float* bk = tm.get(p + "slf_attn.w_ks.bias", D, err);
    const float* Wv = tm.get(p + "slf_attn.w_vs.weight", D * D, err);
    const float* bv = tm.get(p + "slf_attn.w_vs.bias", D, err);
    const float* Wfc = tm.get(p + "slf_attn.fc.weight", D * D, err);
    const float* bfc = tm.get(p + "slf_attn.fc.bias", D, err);
    const float* g1 = tm.get(p + "slf_attn.layer_norm.weight", D, err);
    const float* b1 = tm.get(p + "slf_attn.layer_norm.bias", D, err);
    const int DI = cfg->d_inner;
    const float* W1 = tm.get(p + "pos_ffn.w_1.weight", (int64_t)DI * D, err);
    const float* bb1 = tm.get(p + "pos_ffn.w_1.bias", DI, err);
    const float* W2 = tm.get(p + "pos_ffn.w_2.weight", (int64_t)D * DI, err);
    const float* bb2 = tm.get(p + "pos_ffn.w_2.bias", D, err);
    const float* g2 = tm.get(p + "pos_ffn.layer_norm.weight", D, err);
    const float* b2 = tm.get(p + "pos_ffn.layer_norm.bias", D, err);
    if (err) return err;
    // CLS query, pre-scaled by 1/sqrt(64) (line_attention.py:14)
    std::vector<double> q(D);
    for (int o = 0; o < D; ++o) {
      double s = bq[o];
      for (int i = 0; i < D; ++i) s += (double)Wq[o * D + i] * cls[i];
      q[o] = s / 8.0;
    }
    std::vector<double> U(HEADS * D, 0.0), U2(HEADS * D, 0.0);
    for (int h = 0; h < HEADS; ++h) {
      double c = 0.0;
      for (int d = 0; d < DH; ++d) {
        const int o = h * DH + d;  // descriptive heads are head-major (line_attention.py:55-57)
        c += q[o] * bk[o];
        for (int i = 0; i < D; ++i) U[h * D + i] += q[o] * Wk[o * D + i];
      }
      for (int i = 0; i < D; ++i) {  // U2 = W5^T u_h
        double s = 0.0;
        for (int o = 0; o < D; ++o) s += W5w[(size_t)o * D + i] * U[h * D + o];
        U2[h * D + i] = s;
      }
      double ub5 = 0.0, ucls = 0.0;
      for (int i = 0; i < D; ++i) { ub5 += U[h * D + i] * b5w[i]; ucls += U[h * D + i] * cls[i]; }
      H->pool.c_tok[h] = (float)(ub5 + c);
      H->pool.s_cls[h] = (float)(ucls + c);
    }
    place(&H->pool.U, U);
    place(&H->pool.U2, U2);
    // value path after pooling: att_h = Wv_h dbar + (Wv_h W5) abar + p0 * Wv_h (cls - b5) + (Wv_h b5 + bv_h)
    std::vector<double> Watt((size_t)HEADS * DH * POOLW, 0.0), batt(HEADS * DH);
    for (int h = 0; h < HEADS; ++h)
      for (int d = 0; d < DH; ++d) {
        const int o = h * DH + d;
        double* row = &Watt[((size_t)h * DH + d) * POOLW];
        double r = 0.0, bb = bv[o];
        for (int i = 0; i < D; ++i) {
          row[i] = Wv[o * D + i];
          r += (double)Wv[o * D + i] * ((double)cls[i] - b5w[i]);
          bb += (double)Wv[o * D + i] * b5w[i];
        }
        for (int i = 0; i < D; ++i) {
          double s = 0.0;
          for (int m = 0; m < D; ++m) s += (double)Wv[o * D + m] * W5w[(size_t)m * D + i];
          row[D + i] = s;
        }
        row[2 * D] = r;
        batt[o] = bb;
      }
    place_w(&H->Watt, Watt, HEADS * DH, POOLW);
    place(&H->batt, batt);
    place_w(&H->Wfc, to_d(Wfc, D * D), D, D);
    std::vector<double> bfc2(D);
    for (int i = 0; i < D; ++i) bfc2[i] = (double)bfc[i] + cls[i];  // residual of the CLS row is the constant token
    place(&H->bfc, bfc2);
    place(&H->ln1g, to_d(g1, D)); place(&H->ln1b, to_d(b1, D));
    place_w(&H->Wf1, to_d(W1, (size_t)DI * D), DI, D); place(&H->bf1, to_d(bb1, DI));
    place_w(&H->Wf2, to_d(W2, (size_t)D * DI), D, DI); place(&H->bf2, to_d(bb2, D));
    place(&H->ln2g, to_d(g2, D)); place(&H->ln2b, to_d(b2, D));
  }

  // ---- signature layers --------------------------------------------------------------------------
  H->sig.resize(cfg->n_sig_layers);
  std::vector<std::vector<double>> sig_qkv_d, sig_bqkv_d, sig_w2_d, sig_b2_d;   // kept for the W2 + next-projection fold below
  for (int l = 0; l < cfg->n_sig_layers; ++l) {
    const std::string p = "selfattn.layers." + std::to_string(l) + ".";
    const float* Wp[3];
    const float* bp[3];
    for (int j = 0; j < 3; ++j) {
      Wp[j] = tm.get(p + "attn.proj." + std::to_string(j) + ".weight", D * D, err);
      bp[j] = tm.get(p + "attn.proj." + std::to_string(j) + ".bias", D, err);
    }
    const float* Wm = tm.get(p + "attn.merge.weight", D * D, err);
    const float* bm = tm.get(p + "attn.merge.bias", D, err);
    const float* W1 = tm.get(p + "mlp.0.weight", 4 * D * D, err);
    const float* b1 = tm.get(p + "mlp.0.bias", 2 * D, err);
    const float* g = tm.get(p + "mlp.1.weight", 2 * D, err);
    const float* be = tm.get(p + "mlp.1.bias", 2 * D, err);
    const float* mu = tm.get(p + "mlp.1.running_mean", 2 * D, err);
    const float* va = tm.get(p + "mlp.1.running_var", 2 * D, err);
    const float* W2 = tm.get(p + "mlp.3.weight", 2 * D * D, err);
    const float* b2 = tm.get(p + "mlp.3.bias", D, err);
    if (err) return err;
    // reference channel c = d*4 + h (line_transformer.py:151)  ->  head-major c' = h*64 + d;
    // q additionally scaled by 1/sqrt(64) (:134), an exact power of two.
    std::vector<double> Wqkv((size_t)3 * D * D), bqkv(3 * D);
    for (int j = 0; j < 3; ++j)
      for (int h = 0; h < HEADS; ++h)
        for (int d = 0; d < DH; ++d) {
          const int src = d * HEADS + h, dst = j * D + h * DH + d;
          const double sc = j == 0 ? 0.125 : 1.0;
          for (int i = 0; i < D; ++i) Wqkv[(size_t)dst * D + i] = (double)Wp[j][src * D + i] * sc;
          bqkv[dst] = (double)bp[j][src] * sc;
        }
    std::vector<double> Wm2((size_t)D * D);
    for (int o = 0; o < D; ++o)
      for (int h = 0; h < HEADS; ++h)
        for (int d = 0; d < DH; ++d) Wm2[(size_t)o * D + h * DH + d] = Wm[o * D + d * HEADS + h];
    std::vector<double> W1f, b1f;
    if (cfg->bn_batch_stats) {
      W1f = to_d(W1, (size_t)4 * D * D); b1f = to_d(b1, 2 * D);
      H->bn_g.push_back(nullptr); H->bn_b.push_back(nullptr); H->bn_c.push_back(2 * D);
      place(&H->bn_g[bn_slot], to_d(g, 2 * D)); place(&H->bn_b[bn_slot], to_d(be, 2 * D));
      ++bn_slot;
    } else
    fold_bn(W1, b1, g, be, mu, va, 2 * D, 2 * D, W1f, b1f);
    // fold the attention's merge conv into the MLP's first layer (both linear, nothing in between):
    //   W1 [x ; Wm a + bm] + b1 = W1a x + (W1b Wm) a + (W1b bm + b1)            (line_transformer.py:154,:166)
    std::vector<double> W1m((size_t)2 * D * 2 * D);
    for (int o = 0; o < 2 * D; ++o) {
      const double* w1b = &W1f[(size_t)o * 2 * D + D];
      for (int i = 0; i < D; ++i) W1m[(size_t)o * 2 * D + i] = W1f[(size_t)o * 2 * D + i];
      for (int i = 0; i < D; ++i) {
        double sacc = 0.0;
        for (int m = 0; m < D; ++m) sacc += w1b[m] * Wm2[(size_t)m * D + i];
        W1m[(size_t)o * 2 * D + D + i] = sacc;
      }
      double bacc = b1f[o];
      for (int m = 0; m < D; ++m) bacc += w1b[m] * (double)bm[m];
      b1f[o] = bacc;
    }
    SigLayer& S = H->sig[l];
    sig_qkv_d.push_back(Wqkv); sig_bqkv_d.push_back(bqkv);
    sig_w2_d.push_back(to_d(W2, (size_t)2 * D * D)); sig_b2_d.push_back(to_d(b2, D));
    place_w(&S.Wqkv, Wqkv, 3 * D, D, true); place(&S.bqkv, bqkv);
    place_w(&S.W1, W1m, 2 * D, 2 * D); place(&S.b1, b1f);
    place_w(&S.W2, to_d(W2, (size_t)2 * D * D), D, 2 * D); place(&S.b2, to_d(b2, D));
#ifdef LINETR_EXPERIMENTS
    {
      std::vector<double> W2perm((size_t)2 * D * D);
      for (int o = 0; o < D; ++o)
        for (int k = 0; k < 2 * D; ++k) W2perm[(size_t)o * 2 * D + k] = W2[(size_t)o * 2 * D + sig_mlp_kperm(k)];
      place_w(&S.W2p, W2perm, D, 2 * D);
    }
#endif
  }
  // x_out = z + W2 hid + b2 (line_transformer.py:180-183) and the next layer's q/k/v projection is linear in x_out (:141-143), so
  //   [x_out ; qkv_next] = [[I, W2], [Wqkv, Wqkv W2]] [z ; hid] + [b2 ; Wqkv b2 + bqkv]                      (float64, once)
  // one contraction instead of two dependent ones -- 2.4x the flops, which only pays at single-pair sizes (linetr_net.hip)
  for (int l = 0; l + 1 < cfg->n_sig_layers; ++l) {
    const std::vector<double>&W2 = sig_w2_d[l], &b2 = sig_b2_d[l], &Wq = sig_qkv_d[l + 1], &bq = sig_bqkv_d[l + 1];
    std::vector<double> Wn((size_t)4 * D * 3 * D, 0.0), bn(4 * D);
    for (int o = 0; o < D; ++o) {
      Wn[(size_t)o * 3 * D + o] = 1.0;
      for (int j = 0; j < 2 * D; ++j) Wn[(size_t)o * 3 * D + D + j] = W2[(size_t)o * 2 * D + j];
      bn[o] = b2[o];
    }
    for (int r = 0; r < 3 * D; ++r) {
      double* row = &Wn[(size_t)(D + r) * 3 * D];
      const double* wq = &Wq[(size_t)r * D];
      for (int i = 0; i < D; ++i) row[i] = wq[i];
      double bacc = bq[r];
      for (int m2 = 0; m2 < D; ++m2) {          // row[D + j] += wq[m] * W2[m][j]: contiguous in j
        const double a = wq[m2];
        const double* w2r = &W2[(size_t)m2 * 2 * D];
        for (int j = 0; j < 2 * D; ++j) row[D + j] += a * w2r[j];
        bacc += a * b2[m2];
      }
      bn[D + r] = bacc;
    }
    place_w(&H->sig[l].Wnext, Wn, 4 * D, 3 * D);
    place(&H->sig[l].bnext, bn);
  }
  {
    const float* W = tm.get("final_proj.weight", D * D, err);
    const float* b = tm.get("final_proj.bias", D, err);
    if (err) return err;
    place_w(&H->Wfin, to_d(W, D * D), D, D);
    place(&H->bfin, to_d(b, D));
    if (cfg->n_sig_layers > 0) {
      // x_out = z + W2 hid + b2 (line_transformer.py:180-183) and final_proj is linear (:245), so
      //   final_proj(x_out) = [Wfin | Wfin W2] [z ; hid] + (Wfin b2 + bfin)          (float64, once)
      const std::string p = "selfattn.layers." + std::to_string(cfg->n_sig_layers - 1) + ".";
      const float* W2 = tm.get(p + "mlp.3.weight", 2 * D * D, err);
      const float* b2 = tm.get(p + "mlp.3.bias", D, err);
      if (err) return err;
      std::vector<double> Wf((size_t)D * 3 * D), bf(D);
      for (int o = 0; o < D; ++o) {
        for (int i = 0; i < D; ++i) Wf[(size_t)o * 3 * D + i] = W[o * D + i];
        for (int j = 0; j < 2 * D; ++j) {
          double sacc = 0.0;
          for (int m = 0; m < D; ++m) sacc += (double)W[o * D + m] * (double)W2[(size_t)m * 2 * D + j];
          Wf[(size_t)o * 3 * D + D + j] = sacc;
        }
        double bacc = b[o];
        for (int m = 0; m < D; ++m) bacc += (double)W[o * D + m] * (double)b2[m];
        bf[o] = bacc;
      }
      place_w(&H->Wfin2, Wf, D, 3 * D);
      place(&H->bfin2, bf);
    }
  }

  LT_HIP(hipMalloc((void**)&H->arena, ar.host.size() * sizeof(float)));
  LT_HIP(hipMemcpy(H->arena, ar.host.data(), ar.host.size() * sizeof(float), hipMemcpyHostToDevice));
  for (auto& f : fix) *f.dst = H->arena + f.off;
  {  // split-precision copies of the GEMM weights: made on the device (linetr_net.hip owns the kernels)
    std::vector<GemmWSpec> specs;
    for (auto& w : gemm_w) specs.push_back({*w.dst, w.rows, w.K, w.st});
    if (int e = make_split_copies(H.get(), specs)) return e;
  }
  if (const char* e = getenv("LINETR_PRECISION")) {
    if (!strcmp(e, "f32")) H->precision = LINETR_PREC_F32;
    else if (!strcmp(e, "bf16x3")) H->precision = LINETR_PREC_BF16X3;
    else if (!strcmp(e, "bf16x6")) H->precision = LINETR_PREC_BF16X6;
    else if (!strcmp(e, "f16x3")) H->precision = LINETR_PREC_F16X3;
    else return fail(LINETR_E_ARG, "LINETR_PRECISION must be f32, bf16x3, bf16x6 or f16x3 (got '%s')", e);
  }
  *out = H.release();
  return LINETR_OK;
}

extern "C" int linetr_set_precision(LinetrHandle* h, int32_t mode) {
  if (!h || mode < LINETR_PREC_F32 || mode > LINETR_PREC_F16X3) return fail(LINETR_E_ARG, "bad precision mode");
  h->precision = mode;
  return LINETR_OK;
}
extern "C" int linetr_get_precision(const LinetrHandle* h) { return h ? h->precision : LINETR_E_ARG; }

extern "C" void linetr_destroy(LinetrHandle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  for (auto& p : h->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto e : h->event_pool) (void)hipEventDestroy(e);
  for (hipStream_t x : h->pipe.stream)
    if (x) { (void)hipStreamSynchronize(x); (void)hipStreamDestroy(x); }
  for (int s = 0; s < LinetrHandle::PIPE_SLOTS; ++s) {
    for (hipEvent_t e : {h->pipe.fork[s], h->pipe.done[s]})
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->pipe.cut[s])
      if (e) (void)hipEventDestroy(e);
  }
  if (h->arena) (void)hipFree(h->arena);
  if (h->split_arena) (void)hipFree(h->split_arena);
  if (h->zeros) (void)hipFree(h->zeros);
  if (h->sk_ws) (void)hipFree(h->sk_ws);
  if (h->sk_flags) (void)hipFree(h->sk_flags);
  if (h->pn_abort) (void)hipHostFree(h->pn_abort);
  for (auto& kv : h->debug_split) (void)hipFree(kv.second);
  delete h;
}

// =============================================================================================
// host pre-filter
// =============================================================================================

static int pack_one(LinetrLineRec& r, double td, int T, int image, int line_local, int& sub_cursor, int& tok_cursor) {
  if (!(td > 0) || T < 1) return fail(LINETR_E_ARG, "token_distance must be > 0 and max_tokens >= 1");
  const double nt = std::ceil(r.length / td);          // line_process.py:109
  if (!(nt >= 1) || nt > 1e7) return fail(LINETR_E_ARG, "key-line %d has a non-positive / absurd token count", line_local);
  r.n_tok = (int)nt;
  r.n_sub = (r.n_tok + T - 1) / T;                      // :121
  r.first_sub = sub_cursor;
  r.image = image;
  r.line_local = line_local;
  r.first_tok = tok_cursor;
  sub_cursor += r.n_sub;
  tok_cursor += r.n_tok;
  // the reference asserts every walked distance <= geometric length (:44-45)
  if (r.n_tok >= 2) {
    const double dx = r.ep[0] - r.sp[0], dy = r.ep[1] - r.sp[1];
    const double geo = std::sqrt(dx * dx + dy * dy);
    if (!(geo >= (double)(r.n_tok - 2) * td))
      return fail(LINETR_E_ASSERT, "distance should be smaller than line length! (key-line %d)", line_local);
  }
  return 0;
}

static void angle_of(LinetrLineRec& r) {  // line_process.py:28-41
  double th = std::atan2(r.ep[0] - r.sp[0], r.ep[1] - r.sp[1]);
  if (th < 0) th += M_PI;
#ifdef __GLIBC__
  // one libm call for both (glibc's sincos returns exactly what its sin and cos return)
  ::sincos(2 * th, &r.angle[1], &r.angle[0]);
#else
  r.angle[0] = std::cos(2 * th);
  r.angle[1] = std::sin(2 * th);
#endif
}

extern "C" int linetr_pack_lines(const double* h_klines, const double* h_length, const double* h_angles, int32_t K,
                                 double td, int32_t T, int32_t image_index, int32_t sub_base, int32_t tok_base,
                                 LinetrLineRec* h_recs, int32_t* n_out) {
  if (K < 0 || (K > 0 && (!h_klines || !h_length || !h_angles || !h_recs))) return fail(LINETR_E_ARG, "null argument");
  int cur = sub_base, tcur = tok_base;
  for (int k = 0; k < K; ++k) {
    LinetrLineRec& r = h_recs[k];
    r.sp[0] = h_klines[k * 4 + 0]; r.sp[1] = h_klines[k * 4 + 1];
    r.ep[0] = h_klines[k * 4 + 2]; r.ep[1] = h_klines[k * 4 + 3];
    r.length = h_length[k];
    r.angle[0] = h_angles[k * 2]; r.angle[1] = h_angles[k * 2 + 1];
    if (int e = pack_one(r, td, T, image_index, k, cur, tcur)) return e;
  }
  if (n_out) *n_out = cur - sub_base;
  return LINETR_OK;
}

// filter + sort of one image (records carry geometry/length/angle only).  `emit(n)` is called once with the number of surviving
// lines and returns where to write them: straight into the caller's record array on the single-thread path (a record is 80 bytes;
// the earlier form copied every survivor three times).
struct KeptLine { double sp[2], ep[2], length; };
// images of the last pre-filter call on this thread whose sorted candidates held equal lengths (linetr_prefilter_tied_images)
static thread_local std::vector<int32_t> g_tied_images;
template <class Emit>
static void prefilter_core(const double* L, int32_t K, int32_t height, int32_t width, int32_t border,
                           double min_length, int32_t max_keylines, const double* vm, bool* tied, Emit emit) {
  static thread_local std::vector<KeptLine> keep;
  static thread_local std::vector<std::pair<double, int>> order;
  keep.clear();
  keep.reserve(K);
  const double xmax = ((double)width - 0.001) - (double)border;   // width-eps-border, line_process.py:72-74
  const double ymax = ((double)height - 0.001) - (double)border;
  for (int k = 0; k < K; ++k) {
    const double* l = L + (size_t)k * 6;
    KeptLine r;
    if (l[0] < l[2]) { r.sp[0] = l[0]; r.sp[1] = l[1]; r.ep[0] = l[2]; r.ep[1] = l[3]; }   // :212-217
    else { r.sp[0] = l[2]; r.sp[1] = l[3]; r.ep[0] = l[0]; r.ep[1] = l[1]; }
    // lineLength * 2 ** octave (:220); an integral octave is an exact power of two either way: skip the pow call
    const double oct = l[5];
    r.length = (oct == std::floor(oct) && std::fabs(oct) < 64.0) ? l[4] * std::ldexp(1.0, (int)oct) : l[4] * std::pow(2.0, oct);
    const bool inside = r.sp[0] >= border && r.sp[0] < width - border && r.sp[1] >= border && r.sp[1] < height - border &&
                        r.ep[0] >= border && r.ep[0] < width - border && r.ep[1] >= border && r.ep[1] < height - border;
    if (!inside) continue;                                                                   // :62-70
    r.sp[0] = std::min(r.sp[0], xmax); r.ep[0] = std::min(r.ep[0], xmax);
    r.sp[1] = std::min(r.sp[1], ymax); r.ep[1] = std::min(r.ep[1], ymax);
    if (vm) {                                                                                // :76-80
      const int64_t sx = (int64_t)std::floor(r.sp[0]), sy = (int64_t)std::floor(r.sp[1]);
      const int64_t ex = (int64_t)std::floor(r.ep[0]), ey = (int64_t)std::floor(r.ep[1]);
      auto at = [&](int64_t y, int64_t x) {  // numpy-style wrap of negative indices
        if (y < 0) y += height;
        if (x < 0) x += width;
        return vm[y * width + x];
      };
      if (at(sy, sx) + at(ey, ex) == 0.0) continue;
    }
    if (!(r.length > min_length)) continue;                                                  // :8
    keep.push_back(r);
  }
  // ascending stable order by length, read backwards (:15-16): ties come out in descending input order, as before
  order.resize(keep.size());
  for (size_t i = 0; i < keep.size(); ++i) order[i] = {keep[i].length, (int)i};
  std::stable_sort(order.begin(), order.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first < b.first; });
  // equal lengths anywhere among the candidates (a tie across the [:max_keylines] cut changes the SET, not only the order)
  *tied = false;
  for (size_t i = 1; i < order.size(); ++i)
    if (order[i].first == order[i - 1].first) { *tied = true; break; }
  int64_t n_keep = (int64_t)order.size();
  if (max_keylines < 0) n_keep = std::max<int64_t>(0, n_keep + max_keylines);                // python slice [:m]
  else n_keep = std::min<int64_t>(n_keep, max_keylines);
  LinetrLineRec* out = emit(n_keep);
  if (!out) return;
  for (int64_t i = 0; i < n_keep; ++i) {
    const KeptLine& kl = keep[order[order.size() - 1 - i].second];
    LinetrLineRec r{};
    r.sp[0] = kl.sp[0]; r.sp[1] = kl.sp[1]; r.ep[0] = kl.ep[0]; r.ep[1] = kl.ep[1]; r.length = kl.length;
    angle_of(r);                                                                             // :20
    out[i] = r;
  }
}

extern "C" int linetr_prefilter(const double* L, int32_t K, int32_t height, int32_t width, int32_t border,
                                double min_length, int32_t max_keylines, const double* vm, double td, int32_t T,
                                int32_t image_index, int32_t sub_base, int32_t tok_base, LinetrLineRec* h_recs,
                                int32_t capacity, int32_t* k_out, int32_t* n_out) {
  if (K < 0 || (K > 0 && !L) || !k_out || !n_out) return fail(LINETR_E_ARG, "null argument");
  int64_t n_sel = -1;
  bool tied = false;
  g_tied_images.clear();
  prefilter_core(L, K, height, width, border, min_length, max_keylines, vm, &tied, [&](int64_t n) -> LinetrLineRec* {
    n_sel = n;
    return n <= capacity ? h_recs : nullptr;
  });
  if (tied) g_tied_images.push_back(0);
  if (n_sel > capacity) return fail(LINETR_E_CAPACITY, "prefilter: %lld lines exceed capacity %d", (long long)n_sel, capacity);
  int cur = sub_base, tcur = tok_base;
  for (int64_t i = 0; i < n_sel; ++i)
    if (int e = pack_one(h_recs[i], td, T, image_index, (int)i, cur, tcur)) return e;
  *k_out = (int)n_sel;
  *n_out = cur - sub_base;
  return LINETR_OK;
}

extern "C" int linetr_prefilter_batch(const double* L, const int32_t* off, int32_t B, int32_t height, int32_t width,
                                      int32_t border, double min_length, int32_t max_keylines,
                                      const double* const* vms, double td, int32_t T, int32_t n_threads,
                                      LinetrLineRec* h_recs, int32_t capacity, int32_t* cu_k, int32_t* cu_n) {
  if (B < 0 || !off || !cu_k || !cu_n || (B > 0 && off[B] > 0 && !L)) return fail(LINETR_E_ARG, "null argument");
  WorkPool& pool = WorkPool::get();
  int nt = n_threads > 0 ? n_threads : pool.size();
  nt = std::max(1, std::min(nt, B / 4));  // not worth a hand-off for fewer than 4 images per chunk
  // contiguous chunks of images, a few per thread so that uneven images balance out
  const int chunks = nt == 1 ? 1 : std::min(B, nt * 2);
  cu_k[0] = cu_n[0] = 0;
  int cur = 0, tcur = 0;
  int64_t k = 0;
  g_tied_images.clear();
  if (chunks == 1) {
    // a single pair / a few images: no hand-off, and the survivors are written where they stay
    for (int i = 0; i < B; ++i) {
      int64_t n_sel = -1;
      bool tied = false;
      prefilter_core(L + (size_t)off[i] * 6, off[i + 1] - off[i], height, width, border, min_length, max_keylines,
                     vms ? vms[i] : nullptr, &tied, [&](int64_t n) -> LinetrLineRec* {
                       n_sel = n;
                       return k + n <= capacity ? h_recs + k : nullptr;
                     });
      if (tied) g_tied_images.push_back(i);
      if (k + n_sel > capacity) return fail(LINETR_E_CAPACITY, "prefilter_batch: more than %d surviving lines", capacity);
      for (int64_t j = 0; j < n_sel; ++j)
        if (int e = pack_one(h_recs[k + j], td, T, i, (int)j, cur, tcur)) return e;
      k += n_sel;
      cu_k[i + 1] = (int)k;
      cu_n[i + 1] = cur;
    }
    return LINETR_OK;
  }
  std::vector<std::vector<LinetrLineRec>> sel(B);
  std::vector<char> tied_img(B, 0);
  auto work = [&](int c) {
    const int i0 = (int)((int64_t)B * c / chunks), i1 = (int)((int64_t)B * (c + 1) / chunks);
    for (int i = i0; i < i1; ++i) {
      bool tied = false;
      prefilter_core(L + (size_t)off[i] * 6, off[i + 1] - off[i], height, width, border, min_length, max_keylines,
                     vms ? vms[i] : nullptr, &tied, [&](int64_t n) -> LinetrLineRec* {
                       sel[i].resize(n);
                       return sel[i].data();
                     });
      tied_img[i] = tied;
    }
  };
  pool.run(chunks, work);
  for (int i = 0; i < B; ++i)
    if (tied_img[i]) g_tied_images.push_back(i);
  for (int i = 0; i < B; ++i) {
    if (k + (int64_t)sel[i].size() > capacity)
      return fail(LINETR_E_CAPACITY, "prefilter_batch: more than %d surviving lines", capacity);
    for (size_t j = 0; j < sel[i].size(); ++j) {
      if (int e = pack_one(sel[i][j], td, T, i, (int)j, cur, tcur)) return e;
      h_recs[k++] = sel[i][j];
    }
    cu_k[i + 1] = (int)k;
    cu_n[i + 1] = cur;
  }
  return LINETR_OK;
}

extern "C" int32_t linetr_prefilter_tied_images(int32_t* h_images, int32_t capacity) {
  const int32_t n = (int32_t)g_tied_images.size();
  for (int32_t i = 0; i < n && i < capacity && h_images; ++i) h_images[i] = g_tied_images[i];
  return n;
}

// =============================================================================================
// multi-GPU collective (C-ABI form of parallel.allgather_descriptors)
// =============================================================================================

// ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t, ncclComm_t, hipStream_t)
typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
static allgather_fn g_allgather = nullptr;
static std::mutex g_allgather_mu;

extern "C" int linetr_set_allgather_fn(void* fn) {
  std::lock_guard<std::mutex> lk(g_allgather_mu);
  g_allgather = reinterpret_cast<allgather_fn>(fn);
  return LINETR_OK;
}

extern "C" int linetr_allgather_desc(void* nccl_comm, const void* d_slab, void* d_out, int64_t slab_bytes, void* stream) {
  if (!nccl_comm || !d_slab || !d_out || slab_bytes <= 0) return fail(LINETR_E_ARG, "allgather_desc: bad argument");
  allgather_fn fn = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_allgather_mu);
    fn = g_allgather;
    if (!fn) {   // not given by the caller: the first ncclAllGather the process-wide symbol resolution finds
      void* sym = dlsym(RTLD_DEFAULT, "ncclAllGather");
      for (const char* name : {"librccl.so", "librccl.so.1"}) {
        if (sym) break;
        if (void* lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD)) sym = dlsym(lib, "ncclAllGather");   // only an ALREADY loaded RCCL
      }
      fn = g_allgather = reinterpret_cast<allgather_fn>(sym);
    }
  }
  if (!fn) return fail(LINETR_E_HIP, "allgather_desc: no RCCL (ncclAllGather) is loaded in this process");
  const int rc = fn(d_slab, d_out, (size_t)slab_bytes, /*ncclChar*/ 0, nccl_comm, (hipStream_t)stream);
  if (rc != 0) return fail(LINETR_E_HIP, "allgather_desc: ncclAllGather failed with ncclResult_t %d", rc);
  return LINETR_OK;
}

// =============================================================================================
// profiling
// =============================================================================================

extern "C" int linetr_set_profiling(LinetrHandle* h, int32_t on) {
  if (!h) return fail(LINETR_E_ARG, "null handle");
  for (auto& p : h->pending) { h->event_pool.push_back(p.a); h->event_pool.push_back(p.b); }
  h->pending.clear();
  h->classes.clear();
  h->profiling = on != 0;
  return LINETR_OK;
}

extern "C" int linetr_get_profile(LinetrHandle* h, LinetrProfileEntry* out, int32_t max_entries, int32_t* n_out) {
  if (!h || !n_out) return fail(LINETR_E_ARG, "null argument");
  LT_HIP(hipSetDevice(h->device));
  for (auto& p : h->pending) {
    LT_HIP(hipEventSynchronize(p.b));
    float ms = 0.f;
    LT_HIP(hipEventElapsedTime(&ms, p.a, p.b));
    h->classes[p.cls].ms += ms;
    h->event_pool.push_back(p.a);
    h->event_pool.push_back(p.b);
  }
  h->pending.clear();
  const int n = std::min<int>((int)h->classes.size(), max_entries);
  for (int i = 0; i < n; ++i) {
    out[i].name = h->classes[i].name;
    out[i].calls = h->classes[i].calls;
    out[i].ms = h->classes[i].ms;
    out[i].flops = h->classes[i].flops;
    out[i].bytes = h->classes[i].bytes;
  }
  *n_out = (int)h->classes.size();
  return LINETR_OK;
}
