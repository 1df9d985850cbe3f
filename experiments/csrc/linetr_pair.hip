// Experiments build only (liblinetr_hip_experiments.so): host side of the single-pair persistent signature network (lt_pairnet.h).
// Measured on MI355X against the launch chain it would replace and NOT shipped: 279-340 us against ~225 us for the
// signature network of one cfg2 pair.  LINETR_PAIRNET=1 takes the path; tools/pairnet_timeline.py prints its stage timeline.
#ifdef LINETR_EXPERIMENTS
#include <algorithm>

#include "lt_handle.h"
#include "lt_pairnet.h"

using namespace lt;

namespace {
int n_cus(LinetrHandle* h) {
  if (h->n_cu > 0) return h->n_cu;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, h->device) != hipSuccess) { (void)hipGetLastError(); return 0; }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(pair_net_kernel), PN_THREADS, 0) != hipSuccess) {
    (void)hipGetLastError();
    per_cu = 0;
  }
  h->n_cu = per_cu >= 1 ? prop.multiProcessorCount : -1;   // -1: the kernel does not fit a CU on this device, never take the path
  return h->n_cu;
}
}  // namespace

// a batch the persistent network takes: default precision, a few small images, the reference's layer shapes
bool lt::pairnet_fits(LinetrHandle* h, int n_images, int N, const int32_t* h_cu) {
  const char* on = LT_XENV("LINETR_PAIRNET");            // opt-in (read per call: the tests switch it)
  if (!on || on[0] != '1' || h->pn_disabled || h->precision != LINETR_PREC_BF16X6) return false;
  const int L = (int)h->sig.size();
  if (L < 1 || L > PN_MAX_LAYERS || n_images < 1 || n_images > PN_MAX_IMAGES || N < 1 || N > PN_MAX_ROWS) return false;
  if (h_cu) {
    int rt = 0;
    for (int i = 0; i < n_images; ++i) rt += cdiv(h_cu[i + 1] - h_cu[i], 32);
    if (rt > PN_MAX_RT) return false;
  }
  return n_cus(h) > 0;
}

int64_t lt::pairnet_ws_bytes(const LinetrHandle* h, int N) {
  const int L = (int)h->sig.size();
  if (L < 1 || L > PN_MAX_LAYERS || N > PN_MAX_ROWS) return 0;
  return align_up(pn_ws_floats(std::max(N, 1), L) * 4, 256) + align_up(pn_cnt_bytes(L), 256);
}

// the counters live behind the activations; they are zeroed on the stream at the START of forward_core, long before the launch
int lt::pairnet_prepare(LinetrHandle* h, hipStream_t st, int N, void* ws) {
  const int L = (int)h->sig.size();
  if (!h->pn_abort) {
    unsigned* p = nullptr;
    LT_HIP(hipHostMalloc((void**)&p, 64, hipHostMallocMapped));
    *p = 0u;
    void* d = nullptr;
    LT_HIP(hipHostGetDevicePointer(&d, p, 0));
    h->pn_abort = p;
    h->pn_abort_dev = (unsigned*)d;
  }
  if (*h->pn_abort != 0u) {   // a previous launch gave up waiting: its output was not complete
    h->pn_disabled = true;
    *h->pn_abort = 0u;
    return fail(LINETR_E_HIP, "the previous single-pair network launch timed out waiting for a producer block (device shared with "
                              "another long-running kernel?); its descriptors were incomplete.  The path is retired for this handle");
  }
  char* cnt = (char*)ws + align_up(pn_ws_floats(std::max(N, 1), L) * 4, 256);
  LT_HIP(hipMemsetAsync(cnt, 0, (size_t)pn_cnt_bytes(L), st));
  return LINETR_OK;
}

int lt::pairnet_run(LinetrHandle* h, hipStream_t st, const float* z0, float* out, const int32_t* h_cu, int n_images, int N, void* ws) {
  const int L = (int)h->sig.size();
  PairNetArgs a{};
  a.n_images = n_images; a.n_layers = L; a.N = N;
  int rt = 0;
  double flops = 0;
  for (int i = 0; i < n_images; ++i) {
    a.img_row0[i] = h_cu[i];
    a.img_rt0[i] = rt;
    const double n = h_cu[i + 1] - h_cu[i];
    rt += cdiv(h_cu[i + 1] - h_cu[i], 32);
    flops += L * 2.0 * 2.0 * n * n * D;
  }
  a.img_row0[n_images] = h_cu[n_images];
  a.img_rt0[n_images] = rt;
  a.n_rt = rt;
  flops += 2.0 * N * ((double)L * (3.0 * D * D + 4.0 * D * D) + (L - 1) * 2.0 * D * D + 3.0 * D * D);
  auto sp = [&](const float* W) -> const unsigned char* {
    auto it = h->split.find(W);
    return it == h->split.end() ? nullptr : h->split_arena + it->second.off3;
  };
  for (int l = 0; l < L; ++l) {
    const SigLayer& S = h->sig[l];
    a.layer[l] = PnLayer{sp(S.Wqkv), sp(S.W1), sp(S.W2), S.bqkv, S.b1, S.b2};
    if (!a.layer[l].Wqkv || !a.layer[l].W1 || !a.layer[l].W2) return fail(LINETR_E_ARG, "pair network: weight has no split copy");
  }
  a.Wfin = sp(h->Wfin2); a.bfin = h->bfin2;
  if (!a.Wfin) return fail(LINETR_E_ARG, "pair network: final projection has no split copy");
  a.z0 = z0; a.out = out;
  a.ws = (float*)ws;
  a.cnt = (int*)((char*)ws + align_up(pn_ws_floats(std::max(N, 1), L) * 4, 256));
  a.abort_word = h->pn_abort_dev;
  a.stamps = h->pn_stamps;
  // one block per CU and never more: every block must be resident for the arrival counters to be reached
  const int most_units = rt * 24;        // the widest stage (q/k/v: 24 column tiles per row tile)
  const int grid = std::max(1, std::min(n_cus(h), most_units));
  ProfScope ps(h, st, "pair_net_bf16x6", flops, (double)N * D * 8);
  hipLaunchKernelGGL(pair_net_kernel, dim3(grid), dim3(PN_THREADS), 0, st, a);
  LT_LAUNCH_CHECK();
  return LINETR_OK;
}

// diagnostics: per-block, per-stage wall-clock stamps of the next pair-network launches are written to d_buf
// ([blocks][stages][8] uint64, zeroed by the caller; lt_pairnet.h), NULL switches them off again.  tools/pairnet_timeline.py
extern "C" int linetr_debug_pairnet_stamps(LinetrHandle* h, unsigned long long* d_buf) {
  if (!h) return fail(LINETR_E_ARG, "null handle");
  h->pn_stamps = d_buf;
  return LINETR_OK;
}
#endif  // LINETR_EXPERIMENTS
