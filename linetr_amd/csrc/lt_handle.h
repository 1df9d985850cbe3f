// The library handle and what every translation unit of liblinetr_hip.so shares: the prepared-weight table, the per-kernel-class
// HIP-event profiler and the declarations of the few host functions that cross translation units.
//   linetr_core.hip   lifetime (float64 weight preparation), host pre-filter, collective, profiling entry points
//   linetr_net.hip    tokenise / forward / describe + the GEMM dispatcher and every model kernel
//   linetr_match.hip  matcher, dense-map producer, slab packing
//   linetr_pair.hip   (experiments build only) the single-pair persistent signature network
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "lt_common.h"

struct SigLayer {
  const float *Wqkv, *bqkv, *W1, *b1, *W2, *b2;  // merge conv folded into W1
  const float* W2p = nullptr;                    // W2 with K permuted inside 16-groups (lt_mlp_fused.h)
  // [x_out | q/k/v of the NEXT layer] = Wnext [z ; hid] + bnext: W2 + residual and the next projection as ONE contraction
  // ([4D x 3D]; all layers but the last).  Used for single-pair sizes only, where a dependent launch costs more than its flops.
  const float *Wnext = nullptr, *bnext = nullptr;
};

struct ProfClass {
  const char* name;
  int calls = 0;
  double flops = 0, bytes = 0;
  float ms = 0;
};

struct LinetrHandle {
  LinetrModelConfig cfg;
  int device = 0;
  float* arena = nullptr;  // all prepared weights, one allocation
  // word / line positional encoders (BN folded)
  const float *wW1, *wb1, *wW2, *wb2, *wW3, *wb3, *wW4, *wb4;
  const float *lW1, *lb1, *lW2, *lb2, *lW3, *lb3, *lW4, *lb4, *lW5, *lb5;
  // line-descriptive layer (CLS-row algebra)
  lt::ClsPoolConst pool;
  const float *Watt, *batt, *Wfc, *bfc, *ln1g, *ln1b, *Wf1, *bf1, *Wf2, *bf2, *ln2g, *ln2b;
  std::vector<SigLayer> sig;
  // training-mode handle (cfg.bn_batch_stats): gamma / beta / channel count of every BatchNorm layer, in the order of the packed
  // statistics arrays of linetr_forward_train (word encoder x4, line encoder x4, one per signature layer)
  std::vector<const float*> bn_g, bn_b;
  std::vector<int> bn_c;
  const float *Wfin, *bfin;
  const float *Wfin2 = nullptr, *bfin2 = nullptr;   // final projection with the last signature layer's second MLP GEMM folded in
  // split-bf16 copies of every GEMM weight (2 and 3 planes), keyed by the fp32 pointer
  int precision = LINETR_PREC_BF16X6;
  unsigned char* split_arena = nullptr;
  struct SplitW { size_t off2, off3; int64_t rows; int K; size_t offh = 0; size_t offst = 0; };  // bf16x2 planes, bf16x3 planes, fp16x2 planes, ST image (lt_st_image.h; 0 = none)
  std::map<const float*, SplitW> split;
  std::map<const float*, unsigned char*> debug_split;  // linetr_debug_gemm(cache_weights=1)
  // software pipeline of CONSECUTIVE describe calls (linetr_describe_submit / linetr_describe_join): a batch is cut into stages at
  // fixed points of the network (PipePlan), stage k of every batch runs on stream k, so stage k of batch i + 1 overlaps stage k + 1 of
  // batch i -- HBM-bound kernels under MFMA-bound ones, and the CUs a GEMM's last tile round leaves empty under another launch.
  // A batch in flight owns slot i mod LT_PIPE_SLOTS (its workspace + events).
  static constexpr int PIPE_SLOTS = 4, PIPE_STREAMS = 4;
  struct Pipe {
    hipStream_t stream[PIPE_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork[PIPE_SLOTS] = {}, done[PIPE_SLOTS] = {}, cut[PIPE_SLOTS][PIPE_STREAMS - 1] = {};
    bool submitted[PIPE_SLOTS] = {false, false, false, false};
    bool failed = false;
  } pipe;
  // stream-K workspace of the 128x256 GEMM (partial accumulator tiles + flags, one slot per CU; lt_gemm_split.h)
  float* zeros = nullptr;   // 4096 zero floats: the "no bias" vector of the split-tile GEMM (experiments/csrc/lt_gemm_st.h)
  float* sk_ws = nullptr;
  unsigned* sk_flags = nullptr;
  unsigned sk_epoch = 0;
  // single-pair persistent signature network (lt_pairnet.h, linetr_pair.hip)
  unsigned* pn_abort = nullptr;       // host-mapped word a timed-out launch raises (read by the host before the next launch)
  unsigned* pn_abort_dev = nullptr;   // the same word as the device sees it
  bool pn_disabled = false;
  unsigned long long* pn_stamps = nullptr;   // diagnostics buffer (experiments build: linetr_debug_pairnet_stamps)
  int n_cu = 0;
  // profiling
  bool profiling = false;
  std::vector<ProfClass> classes;
  struct Pending { int cls; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
};

namespace lt {

// widest BatchNorm layer the statistics scratch of linetr_forward_train is sized for (2 D of the signature MLPs at D = 256, lt_bntrain.h);
// linetr_create refuses a training-mode handle with a wider layer
constexpr int BN_MAX_CHANNELS = 512;

inline int prof_class(LinetrHandle* h, const char* name) {
  for (size_t i = 0; i < h->classes.size(); ++i)
    if (h->classes[i].name == name || strcmp(h->classes[i].name, name) == 0) return (int)i;
  ProfClass c;
  c.name = name;
  h->classes.push_back(c);
  return (int)h->classes.size() - 1;
}

inline hipEvent_t prof_event(LinetrHandle* h) {
  if (!h->event_pool.empty()) {
    hipEvent_t e = h->event_pool.back();
    h->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);   // a null event only loses this kernel's timing sample
  return e;
}

// RAII bracket around one kernel launch
struct ProfScope {
  LinetrHandle* h;
  hipStream_t st;
  int cls = -1;
  hipEvent_t a{}, b{};
  ProfScope(LinetrHandle* h_, hipStream_t st_, const char* name, double flops, double bytes) : h(h_), st(st_) {
    if (!h || !h->profiling) return;
    cls = prof_class(h, name);
    h->classes[cls].calls++;
    h->classes[cls].flops += flops;
    h->classes[cls].bytes += bytes;
    a = prof_event(h);
    b = prof_event(h);
    (void)hipEventRecord(a, st);
  }
  ~ProfScope() {
    if (cls < 0) return;
    (void)hipEventRecord(b, st);
    h->pending.push_back({cls, a, b});
  }
};

// one GEMM weight of the prepared arena that needs split-precision copies (made on the device by linetr_net.hip)
struct GemmWSpec { const float* W; int64_t rows; int K; bool st; };
int make_split_copies(LinetrHandle* H, const std::vector<GemmWSpec>& weights);

// Experiments build: the line-signature network of a single pair (a few small images) as ONE persistent launch (lt_pairnet.h):
//   pairnet_fits      does this batch take the path (precision, image count, row count)?  h_cu may be NULL (size check only)
//   pairnet_ws_bytes  bytes of workspace it needs for N rows (0 when N is out of range)
//   pairnet_prepare   zeroes the arrival counters on the stream (call it EARLY, well ahead of the launch)
//   pairnet_run       z0 [N,256] -> line_desc [N,256]
constexpr int PN_MAX_ROWS = 1024;
bool pairnet_fits(LinetrHandle* h, int n_images, int N, const int32_t* h_cu);
int64_t pairnet_ws_bytes(const LinetrHandle* h, int N);
int pairnet_prepare(LinetrHandle* h, hipStream_t st, int N, void* ws);
int pairnet_run(LinetrHandle* h, hipStream_t st, const float* z0, float* out, const int32_t* h_cu, int n_images, int N, void* ws);

}  // namespace lt
