// liblinetr_hip.so, translation unit 3 of 4: the descriptor-distance matcher, the dense-map producer and the slab packing of the
// multi-GPU path (C ABI in include/linetr_hip.h; kernels in lt_match.h, lt_producer.h).
#include <algorithm>
#include <numeric>

#include "lt_handle.h"
#include "lt_match.h"
#include "lt_producer.h"

using namespace lt;

// =============================================================================================
// matcher
// =============================================================================================

namespace {
// Pinned staging ring for the small host tables the matcher uploads (PairDesc array, identity maps).  A slot is
// reused only after the copy that read it has completed (event), so no entry point has to synchronise the stream.
struct PinnedRing {
  struct Slot { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool busy = false; };
  Slot slots[8];
  int next = 0;
  std::mutex m;
  // returns a host pointer of >= bytes, or nullptr; *slot_out identifies the slot for commit()
  void* acquire(size_t bytes, int* slot_out) {
    std::lock_guard<std::mutex> lk(m);
    Slot& s = slots[next];
    *slot_out = next;
    next = (next + 1) % 8;
    if (s.busy) { (void)hipEventSynchronize(s.ev); s.busy = false; }
    if (s.cap < bytes) {
      if (s.p) (void)hipHostFree(s.p);
      s.p = nullptr; s.cap = 0;
      const size_t cap = std::max<size_t>(align_up((int64_t)bytes * 2, 4096), 16384);
      if (hipHostMalloc(&s.p, cap, hipHostMallocDefault) != hipSuccess) { s.p = nullptr; return nullptr; }
      s.cap = cap;
    }
    if (!s.ev && hipEventCreateWithFlags(&s.ev, hipEventDisableTiming) != hipSuccess) { s.ev = nullptr; return nullptr; }
    return s.p;
  }
  int commit(int slot, hipStream_t st) {   // call after the last async copy out of the slot has been enqueued
    std::lock_guard<std::mutex> lk(m);
    LT_HIP(hipEventRecord(slots[slot].ev, st));
    slots[slot].busy = true;
    return 0;
  }
};
PinnedRing& staging_ring() {   // one ring per device (its events belong to the device current at creation); leaked: see WorkPool
  static PinnedRing* rings[64] = {};
  static std::mutex m;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(m);
  PinnedRing*& r = rings[dev & 63];
  if (!r) r = new PinnedRing();
  return *r;
}

// Scratch of the one-launch single-pair matcher (pair_match_fused_kernel): column keys, row results, arrival counter.  One slot per
// (device, stream): launches on a stream are ordered and the kernel leaves its slot clean, so a slot is never shared by two
// launches in flight.  A slot is initialised (column and row keys all-ones, counter 0) by memsets queued on ITS stream in front of
// its first launch -- nothing is synchronised.  At most PAIR_SLOTS_MAX slots are kept (they are never freed: a stream handle may still
// have work in flight); a caller beyond that, or one whose allocation fails, gets nullptr and takes the three launches.
constexpr size_t PAIR_SLOTS_MAX = 256;
PairSlot* pair_slot(hipStream_t st) {
  static std::mutex m;
  static std::map<std::pair<int, hipStream_t>, PairSlot> slots;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(m);
  auto it = slots.find({dev, st});
  if (it != slots.end()) return &it->second;
  if (slots.size() >= PAIR_SLOTS_MAX) return nullptr;
  char* mem = nullptr;
  const size_t bytes = (size_t)PF_MAX_K * 16 + 256;
  if (hipMalloc((void**)&mem, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (hipMemsetAsync(mem, 0xff, (size_t)PF_MAX_K * 16, st) != hipSuccess ||
      hipMemsetAsync(mem + (size_t)PF_MAX_K * 16, 0, 256, st) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(mem);
    return nullptr;
  }
  PairSlot ps;
  ps.col_best = (unsigned long long*)mem;
  ps.row_res = (unsigned long long*)(mem + (size_t)PF_MAX_K * 8);
  ps.counter = (unsigned*)(mem + (size_t)PF_MAX_K * 16);
  return &(slots[{dev, st}] = ps);
}

// the one-launch matcher needs up to pair_fused_lds(PF_MAX_N1, PF_MAX_N1) bytes of dynamic LDS: raised once per device; a device that
// refuses keeps the three launches
bool fused_pair_lds_ok() {
  static unsigned long long done = 0, bad = 0;
  const unsigned long long dev_bit = current_device_bit();
  if (!(done & dev_bit)) {
    const int want = (int)pair_fused_lds(PF_MAX_N1, PF_MAX_N1);
    const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(pair_match_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(pair_match_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); bad |= dev_bit; }
    done |= dev_bit;
  }
  return !(bad & dev_bit);
}

// does a single-pair call have the sizes of the one-launch matcher (pair_match_fused_kernel)?
bool fused_pair_applies(int n0, int k0, int n1, int k1) {
  (void)n0;
  return k0 > 0 && k1 > 0 && n1 <= PF_MAX_N1 && k0 <= PF_MAX_K && k1 <= PF_MAX_K;
}

// the scratch slot of a single-pair call that takes the one-launch matcher, or nullptr (sizes beyond it, no LDS, no slot): then the
// three launches run.  Idempotent per (device, stream): linetr_match_points asks first (it skips the identity maps the one-launch
// path never reads) and linetr_match_gathered gets the same answer.
PairSlot* fused_pair_slot(int n0, int k0, int n1, int k1, hipStream_t st) {
  return fused_pair_applies(n0, k0, n1, k1) && fused_pair_lds_ok() ? pair_slot(st) : nullptr;
}

// ints of argmin scratch one pair needs (layout in lt_match.h)
int64_t pair_scratch_ints(int k0, int k1) { return 2 * (int64_t)k0 + 2 * (int64_t)k1 + 1 + 2 * (int64_t)cdiv(std::max(k0, 1), PM_ROWS) * k1 + 8; }
}  // namespace

extern "C" int64_t linetr_match_workspace_bytes(int32_t n_pairs, int64_t sum_n0n1, int64_t sum_k0k1, int64_t sum_k) {
  (void)sum_k0k1;
  // scratch bound: sum over pairs of pair_scratch_ints(k0,k1) <= 4 sum_k + 2 (sum_k0k1 / PM_ROWS + sum_k) + 9 P, and
  // k0 k1 <= n0 n1
  const int64_t scratch = 6 * sum_k + 2 * (sum_n0n1 / PM_ROWS + 1) + 9 * (int64_t)n_pairs;
  return align_up((int64_t)n_pairs * sizeof(PairDesc), 256) + align_up(sum_n0n1 * 4, 256) + align_up(scratch * 4, 256) + 256;
}

extern "C" int linetr_match_gathered(LinetrHandle* h, int32_t P, const int32_t* dims, const float* d_desc0,
                                     const int64_t* off_n0, const int32_t* d_s2l0, const int64_t* off_s0,
                                     const float* d_desc1, const int64_t* off_n1, const int32_t* d_s2l1,
                                     const int64_t* off_s1, float thr, int32_t mutual, float* d_dk, const int64_t* off_dk,
                                     int32_t* d_match01, const int64_t* off_k0, void* d_ws, int64_t ws_bytes, void* stream);

extern "C" int linetr_match(LinetrHandle* h, int32_t P, const int32_t* dims, const float* d_desc0, const int64_t* off_n0,
                            const int32_t* d_s2l0, const float* d_desc1, const int64_t* off_n1, const int32_t* d_s2l1,
                            float thr, int32_t mutual, float* d_dk, const int64_t* off_dk, int32_t* d_match01,
                            const int64_t* off_k0, void* d_ws, int64_t ws_bytes, void* stream) {
  return linetr_match_gathered(h, P, dims, d_desc0, off_n0, d_s2l0, nullptr, d_desc1, off_n1, d_s2l1, nullptr, thr, mutual,
                               d_dk, off_dk, d_match01, off_k0, d_ws, ws_bytes, stream);
}

extern "C" int linetr_match_gathered(LinetrHandle* h, int32_t P, const int32_t* dims, const float* d_desc0,
                                     const int64_t* off_n0, const int32_t* d_s2l0, const int64_t* off_s0,
                                     const float* d_desc1, const int64_t* off_n1, const int32_t* d_s2l1,
                                     const int64_t* off_s1, float thr, int32_t mutual, float* d_dk, const int64_t* off_dk,
                                     int32_t* d_match01, const int64_t* off_k0, void* d_ws, int64_t ws_bytes, void* stream) {
  if (P < 0) return fail(LINETR_E_ARG, "match: bad argument");
  if (P == 0) return LINETR_OK;
  if (!dims || !off_n0 || !off_n1 || !off_dk || !off_k0 || !d_ws) return fail(LINETR_E_ARG, "match: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (h) LT_HIP(hipSetDevice(h->device));
  PairTable tab{};
  int slot = -1;
  PairDesc* pd = tab.inl;
  if (P > PT_INLINE) {
    pd = (PairDesc*)staging_ring().acquire((size_t)P * sizeof(PairDesc), &slot);
    if (!pd) return fail(LINETR_E_HIP, "match: pinned staging allocation failed");
  } else {
    tab.n_inline = P;
  }
  int64_t od = 0, os = 0, sum_k = 0;
  int max_n0 = 0, max_n1 = 0, max_k1 = 0, max_chunks = 0;
  double flops = 0;
  for (int p = 0; p < P; ++p) {
    PairDesc& d = pd[p];
    d.n0 = dims[p * 4 + 0]; d.k0 = dims[p * 4 + 1]; d.n1 = dims[p * 4 + 2]; d.k1 = dims[p * 4 + 3];
    if (d.n0 < 0 || d.n1 < 0 || d.k0 < 0 || d.k1 < 0 || d.k0 > d.n0 || d.k1 > d.n1)
      return fail(LINETR_E_ARG, "match: bad dims for pair %d", p);
    d.off_n0 = off_n0[p]; d.off_n1 = off_n1[p]; d.off_dk = off_dk[p]; d.off_k0 = off_k0[p];
    d.off_s0 = off_s0 ? off_s0[p] : off_n0[p];
    d.off_s1 = off_s1 ? off_s1[p] : off_n1[p];
    d.off_d = od; od += (int64_t)d.n0 * d.n1;
    d.chunks = cdiv(std::max(d.k0, 1), PM_ROWS);
    d.pad_ = 0;
    d.off_seg = os; os += pair_scratch_ints(d.k0, d.k1);
    sum_k += d.k0 + d.k1;
    max_n0 = std::max(max_n0, d.n0); max_n1 = std::max(max_n1, d.n1);
    max_k1 = std::max(max_k1, d.k1); max_chunks = std::max(max_chunks, d.chunks);
    flops += 2.0 * d.n0 * d.n1 * D;
  }
  if (ws_bytes < linetr_match_workspace_bytes(P, od, 0, sum_k)) return fail(LINETR_E_WORKSPACE, "match: workspace too small");
  char* base = (char*)d_ws;
  PairDesc* d_pd = (PairDesc*)base;
  float* d_dist = (float*)(base + align_up((int64_t)P * sizeof(PairDesc), 256));
  int* d_scr = (int*)((char*)d_dist + align_up(od * 4, 256));
  if (slot >= 0) {
    LT_HIP(hipMemcpyAsync(d_pd, pd, P * sizeof(PairDesc), hipMemcpyHostToDevice, st));
    if (int e = staging_ring().commit(slot, st)) return e;
    tab.ptr = d_pd;
  }
  if (max_n0 > 0 && max_n1 > 0) {
    if (!d_desc0 || !d_desc1 || !d_s2l0 || !d_s2l1 || !d_dk) return fail(LINETR_E_ARG, "match: null tensor");
    // A single pair of ordinary size: ONE launch (pair_match_fused_kernel, lt_match.h).  (r03 built a one-launch form that staged
    // 8 K steps through LDS with a load round trip exposed at each and lost to the three launches, 0.10 vs 0.08 ms; this one keeps
    // whole operand rows in registers -- two exposed round trips in all -- and combines the column argmin with one 64-bit atomicMin.)
    PairSlot* ps_ = P == 1 ? fused_pair_slot(pd[0].n0, pd[0].k0, pd[0].n1, pd[0].k1, st) : nullptr;
    if (ps_) {    // (no slot / no LDS: the three launches below)
      const PairDesc& d = pd[0];
      const size_t lds = pair_fused_lds(d.n1, d.k1);
      ProfScope ps(h, st, "pair_match_fused", flops, 4.0 * ((double)(d.n0 + d.n1) * D + (double)d.k0 * d.k1));
      // one sub-line per key-line on both sides (known from the counts alone: the maps are onto): Dk = D, columns split over two
      // blocks when a wave would otherwise multiply two tiles
      const bool ident = d.n0 == d.k0 && d.n1 == d.k1;
      const float* a0 = d_desc0 + d.off_n0 * D; const float* a1 = d_desc1 + d.off_n1 * D;
      if (ident) {
        const int n_ct = cdiv(d.n1, 16);
        hipLaunchKernelGGL(pair_match_fused_kernel<true>, dim3(cdiv(d.k0, PM_ROWS), cdiv(n_ct, 8)), dim3(512), lds, st, a0, a1,
                           d_s2l0 + d.off_s0, d_s2l1 + d.off_s1, d.n0, d.k0, d.n1, d.k1, thr, mutual, d_dk + d.off_dk,
                           d_match01 + d.off_k0, *ps_);
      } else {
        hipLaunchKernelGGL(pair_match_fused_kernel<false>, dim3(cdiv(d.k0, PM_ROWS)), dim3(512), lds, st, a0, a1, d_s2l0 + d.off_s0,
                           d_s2l1 + d.off_s1, d.n0, d.k0, d.n1, d.k1, thr, mutual, d_dk + d.off_dk, d_match01 + d.off_k0, *ps_);
      }
      LT_LAUNCH_CHECK();
      return LINETR_OK;
    }
    ProfScope ps(h, st, "pair_dist", flops, 0);
    hipLaunchKernelGGL(pair_dist_kernel, dim3(cdiv(max_n1, 64), cdiv(max_n0, 64), P), dim3(256), 0, st, tab, d_desc0,
                       d_desc1, d_dist);
    LT_LAUNCH_CHECK();
  }
  {
    ProfScope ps(h, st, "pair_match", 0, 0);
    if (max_k1 > 0) {
      const int seg1_global = max_k1 > PM_MAX_K1;    // the reference has no limit (max_keylines / max_keypoints = -1)
      if (seg1_global) {
        hipLaunchKernelGGL(pair_seg1_kernel, dim3(cdiv(max_n1, 2048), P), dim3(256), 0, st, tab, d_s2l1, d_scr);
        LT_LAUNCH_CHECK();
      }
      const int cache_dk = max_k1 <= PM_CACHE_K1;
      hipLaunchKernelGGL(pair_pool_kernel, dim3(max_chunks, P), dim3(256), pair_pool_lds(max_k1, seg1_global, cache_dk), st, tab, d_s2l0,
                         d_s2l1, d_dist, d_dk, d_scr, seg1_global, cache_dk);
      LT_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(pair_final_kernel, dim3(P), dim3(256), 0, st, tab, thr, mutual, d_match01, d_scr);
    LT_LAUNCH_CHECK();
  }
  return LINETR_OK;   // fully asynchronous: the pair table travels in the kernel arguments or in ring-owned pinned memory
}

extern "C" int linetr_match_points(LinetrHandle* h, const float* d0_cn, int32_t n0, const float* d1_cn, int32_t n1,
                                   float thr, int32_t mutual, float* d_dist, int32_t* d_match01, void* d_ws,
                                   int64_t ws_bytes, void* stream) {
  if (n0 < 0 || n1 < 0) return fail(LINETR_E_ARG, "match_points: bad argument");
  if (n0 == 0) return LINETR_OK;
  hipStream_t st = (hipStream_t)stream;
  if (h) LT_HIP(hipSetDevice(h->device));
  // scratch: row-major copies + identity sub2line maps + the generic matcher's workspace
  const int64_t need_t = align_up((int64_t)n0 * D * 4, 256) + align_up((int64_t)std::max(n1, 1) * D * 4, 256) +
                         align_up((int64_t)n0 * 4, 256) + align_up((int64_t)std::max(n1, 1) * 4, 256);
  const int64_t need_m = linetr_match_workspace_bytes(1, (int64_t)n0 * n1, 0, n0 + n1);
  if (ws_bytes < need_t + need_m) return fail(LINETR_E_WORKSPACE, "match_points: workspace too small (need %lld)", (long long)(need_t + need_m));
  char* base = (char*)d_ws;
  float* r0 = (float*)base; base += align_up((int64_t)n0 * D * 4, 256);
  float* r1 = (float*)base; base += align_up((int64_t)std::max(n1, 1) * D * 4, 256);
  int* id0 = (int*)base; base += align_up((int64_t)n0 * 4, 256);
  int* id1 = (int*)base; base += align_up((int64_t)std::max(n1, 1) * 4, 256);
  if (!fused_pair_slot(n0, n0, n1, n1, st)) {      // (the one-launch identity path never reads the maps: no upload on the latency path)
    int slot = 0;
    const int m = std::max(n0, n1);
    int* iota = (int*)staging_ring().acquire((size_t)m * sizeof(int), &slot);
    if (!iota) return fail(LINETR_E_HIP, "match_points: pinned staging allocation failed");
    std::iota(iota, iota + m, 0);
    LT_HIP(hipMemcpyAsync(id0, iota, n0 * 4, hipMemcpyHostToDevice, st));
    if (n1 > 0) LT_HIP(hipMemcpyAsync(id1, iota, n1 * 4, hipMemcpyHostToDevice, st));
    if (int e = staging_ring().commit(slot, st)) return e;
  }
  hipLaunchKernelGGL(transpose_cn_kernel, dim3(cdiv(n0, 32), D / 32), dim3(32, 8), 0, st, d0_cn, r0, D, n0);
  if (n1 > 0) hipLaunchKernelGGL(transpose_cn_kernel, dim3(cdiv(n1, 32), D / 32), dim3(32, 8), 0, st, d1_cn, r1, D, n1);
  LT_LAUNCH_CHECK();
  const int32_t dims[4] = {n0, n0, n1, n1};
  const int64_t zero = 0;
  return linetr_match(h, 1, dims, r0, &zero, id0, r1, &zero, id1, thr, mutual, d_dist, &zero, d_match01, &zero, base,
                      ws_bytes - need_t, stream);
}

// The matching tail of Matching.forward (models/matching.py:67-84) in ONE call: point matcher (nn_matcher on the two [256,n] SuperPoint
// descriptor sets), line matcher (get_dist_matrix + subline2keyline + nn_matcher_distmat on the two images' line descriptors) and the
// four device -> host copies of their results into ONE caller-provided pinned block, all asynchronous on `stream`.  Either branch may
// be switched off (np0 = 0 / k0 = 0).  Layout of the pinned block (byte offsets returned in h_offsets[4], every segment 256-aligned):
//   point distances [np0][np1] f32 | point match01 [np0] i32 | key-line distances Dk [k0][k1] f32 | line match01 [k0] i32
namespace {
struct TailLayout { int64_t o_pd, o_pm, o_ld, o_lm, out_total, ws_points, ws_lines, ws_total; };
TailLayout tail_layout(int np0, int np1, int n0, int k0, int n1, int k1) {
  TailLayout L{};
  int64_t o = 0;
  L.o_pd = o; o += align_up((int64_t)np0 * np1 * 4, 256);
  L.o_pm = o; o += align_up((int64_t)np0 * 4, 256);
  L.o_ld = o; o += align_up((int64_t)k0 * k1 * 4, 256);
  L.o_lm = o; o += align_up((int64_t)k0 * 4, 256);
  L.out_total = o;
  L.ws_points = np0 > 0 && np1 > 0 ? 4 * (int64_t)(np0 + np1) * (D + 1) + 2048 + linetr_match_workspace_bytes(1, (int64_t)np0 * np1, 0, np0 + np1) : 0;
  L.ws_lines = k0 > 0 && k1 > 0 ? linetr_match_workspace_bytes(1, (int64_t)n0 * n1, 0, k0 + k1) : 0;
  L.ws_total = align_up(L.out_total, 256) + align_up(L.ws_points, 256) + align_up(L.ws_lines, 256) + 256;
  return L;
}
}  // namespace

extern "C" int64_t linetr_pair_tail_workspace_bytes(int32_t np0, int32_t np1, int32_t n0, int32_t k0, int32_t n1, int32_t k1) {
  return tail_layout(std::max(np0, 0), std::max(np1, 0), std::max(n0, 0), std::max(k0, 0), std::max(n1, 0), std::max(k1, 0)).ws_total;
}

extern "C" int64_t linetr_pair_tail_output_bytes(int32_t np0, int32_t np1, int32_t k0, int32_t k1, int64_t* h_offsets) {
  const TailLayout L = tail_layout(std::max(np0, 0), std::max(np1, 0), 0, std::max(k0, 0), 0, std::max(k1, 0));
  if (h_offsets) { h_offsets[0] = L.o_pd; h_offsets[1] = L.o_pm; h_offsets[2] = L.o_ld; h_offsets[3] = L.o_lm; }
  return L.out_total;
}

extern "C" int linetr_pair_tail(LinetrHandle* h, const float* d_pdesc0_cn, int32_t np0, const float* d_pdesc1_cn, int32_t np1, float thr_p,
                                const float* d_ldesc0, int32_t n0, const int32_t* d_s2l0, int32_t k0, const float* d_ldesc1, int32_t n1,
                                const int32_t* d_s2l1, int32_t k1, float thr_l, int32_t mutual, void* h_pinned_out, int64_t pinned_bytes,
                                void* d_ws, int64_t ws_bytes, void* stream) {
  if (np0 < 0 || np1 < 0 || n0 < 0 || n1 < 0 || k0 < 0 || k1 < 0) return fail(LINETR_E_ARG, "pair_tail: bad dims");
  const bool points = np0 > 0 && np1 > 0, lines = k0 > 0 && k1 > 0;
  const TailLayout L = tail_layout(np0, np1, n0, k0, n1, k1);
  if (!h_pinned_out || pinned_bytes < L.out_total) return fail(LINETR_E_CAPACITY, "pair_tail: output block too small (need %lld)", (long long)L.out_total);
  if (!d_ws || ws_bytes < L.ws_total) return fail(LINETR_E_WORKSPACE, "pair_tail: workspace too small (need %lld)", (long long)L.ws_total);
  hipStream_t st = (hipStream_t)stream;
  if (h) LT_HIP(hipSetDevice(h->device));
  char* dout = (char*)d_ws;                                       // device image of the output block
  char* wsp = dout + align_up(L.out_total, 256);
  char* wsl = wsp + align_up(L.ws_points, 256);
  char* hout = (char*)h_pinned_out;
  if (points) {
    if (!d_pdesc0_cn || !d_pdesc1_cn) return fail(LINETR_E_ARG, "pair_tail: null point descriptors");
    if (int e = linetr_match_points(h, d_pdesc0_cn, np0, d_pdesc1_cn, np1, thr_p, mutual, (float*)(dout + L.o_pd), (int32_t*)(dout + L.o_pm), wsp,
                                    L.ws_points, stream)) return e;
    // requested now: the 1 MB distance matrix travels while the line matcher runs
    LT_HIP(hipMemcpyAsync(hout + L.o_pd, dout + L.o_pd, (size_t)np0 * np1 * 4, hipMemcpyDeviceToHost, st));
    LT_HIP(hipMemcpyAsync(hout + L.o_pm, dout + L.o_pm, (size_t)np0 * 4, hipMemcpyDeviceToHost, st));
  }
  if (lines) {
    if (!d_ldesc0 || !d_ldesc1 || !d_s2l0 || !d_s2l1) return fail(LINETR_E_ARG, "pair_tail: null line descriptors / maps");
    const int32_t dims[4] = {n0, k0, n1, k1};
    const int64_t zero = 0;
    if (int e = linetr_match(h, 1, dims, d_ldesc0, &zero, d_s2l0, d_ldesc1, &zero, d_s2l1, thr_l, mutual, (float*)(dout + L.o_ld), &zero,
                             (int32_t*)(dout + L.o_lm), &zero, wsl, L.ws_lines, stream)) return e;
    LT_HIP(hipMemcpyAsync(hout + L.o_ld, dout + L.o_ld, (size_t)k0 * k1 * 4, hipMemcpyDeviceToHost, st));
    LT_HIP(hipMemcpyAsync(hout + L.o_lm, dout + L.o_lm, (size_t)k0 * 4, hipMemcpyDeviceToHost, st));
  }
  return LINETR_OK;
}

extern "C" int64_t linetr_match_distmat_workspace_bytes(int32_t n0, int32_t n1) {
  n0 = std::max(n0, 0); n1 = std::max(n1, 0);
  return 256 + align_up((int64_t)n0 * 4, 256) + align_up((int64_t)std::max(n1, 1) * 4, 256) +
         align_up((int64_t)n0 * std::max(n1, 1) * 4, 256) + align_up(pair_scratch_ints(n0, n1) * 4, 256);
}

extern "C" int linetr_match_distmat(LinetrHandle* h, const float* d_dist, int32_t n0, int32_t n1, float thr,
                                    int32_t mutual, int32_t* d_match01, void* d_ws, int64_t ws_bytes, void* stream) {
  if (n0 < 0 || n1 < 0) return fail(LINETR_E_ARG, "match_distmat: bad argument");
  if (n0 == 0) return LINETR_OK;
  hipStream_t st = (hipStream_t)stream;
  if (h) LT_HIP(hipSetDevice(h->device));
  // scratch: PairDesc | identity maps | Dk copy | argmin ints
  const int64_t o_id0 = 256, o_id1 = o_id0 + align_up((int64_t)n0 * 4, 256);
  const int64_t o_dk = o_id1 + align_up((int64_t)std::max(n1, 1) * 4, 256);
  const int64_t o_scr = o_dk + align_up((int64_t)n0 * std::max(n1, 1) * 4, 256);
  const int64_t need = linetr_match_distmat_workspace_bytes(n0, n1);
  if (!d_ws || ws_bytes < need) return fail(LINETR_E_WORKSPACE, "match_distmat: workspace too small (need %lld)", (long long)need);
  char* base = (char*)d_ws;
  const int m = std::max(n0, n1);
  int slot = 0;
  int* iota = (int*)staging_ring().acquire((size_t)m * sizeof(int), &slot);
  if (!iota) return fail(LINETR_E_HIP, "match_distmat: pinned staging allocation failed");
  PairTable tab{};
  tab.n_inline = 1;
  PairDesc* pd = tab.inl;
  pd->n0 = pd->k0 = n0; pd->n1 = pd->k1 = n1;
  pd->chunks = cdiv(n0, PM_ROWS);
  std::iota(iota, iota + m, 0);
  LT_HIP(hipMemcpyAsync(base + o_id0, iota, n0 * 4, hipMemcpyHostToDevice, st));
  if (n1 > 0) LT_HIP(hipMemcpyAsync(base + o_id1, iota, n1 * 4, hipMemcpyHostToDevice, st));
  if (int e = staging_ring().commit(slot, st)) return e;
  if (n1 > 0) {
    const int seg1_global = n1 > PM_MAX_K1;
    if (seg1_global) {
      hipLaunchKernelGGL(pair_seg1_kernel, dim3(cdiv(n1, 2048), 1), dim3(256), 0, st, tab, (const int*)(base + o_id1), (int*)(base + o_scr));
      LT_LAUNCH_CHECK();
    }
    const int cache_dk = n1 <= PM_CACHE_K1;
    hipLaunchKernelGGL(pair_pool_kernel, dim3(pd->chunks, 1), dim3(256), pair_pool_lds(n1, seg1_global, cache_dk),
                       st, tab, (const int*)(base + o_id0), (const int*)(base + o_id1), d_dist, (float*)(base + o_dk),
                       (int*)(base + o_scr), seg1_global, cache_dk);
    LT_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(pair_final_kernel, dim3(1), dim3(256), 0, st, tab, thr, mutual, d_match01, (int*)(base + o_scr));
  LT_LAUNCH_CHECK();
  return LINETR_OK;
}

extern "C" int64_t linetr_match_distmat_f64_workspace_bytes(int32_t n0, int32_t n1) {
  n0 = std::max(n0, 0); n1 = std::max(n1, 0);
  return align_up((int64_t)n0 * 8, 256) + align_up((int64_t)n0 * 4, 256) + align_up((int64_t)std::max(n1, 1) * 4, 256) + 256;
}

extern "C" int linetr_match_distmat_f64(LinetrHandle* h, const double* d_dist, int32_t n0, int32_t n1, double thr, int32_t mutual,
                                        int32_t* d_match01, void* d_ws, int64_t ws_bytes, void* stream) {
  if (n0 < 0 || n1 < 0) return fail(LINETR_E_ARG, "match_distmat_f64: bad argument");
  if (n0 == 0) return LINETR_OK;
  if (!d_match01 || !d_ws || (n1 > 0 && !d_dist)) return fail(LINETR_E_ARG, "match_distmat_f64: null pointer");
  if (ws_bytes < linetr_match_distmat_f64_workspace_bytes(n0, n1)) return fail(LINETR_E_WORKSPACE, "match_distmat_f64: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (h) LT_HIP(hipSetDevice(h->device));
  if (n1 == 0) { LT_HIP(hipMemsetAsync(d_match01, 0xff, (size_t)n0 * 4, st)); return LINETR_OK; }
  char* base = (char*)d_ws;
  double* row_min = (double*)base;
  int* row_arg = (int*)(base + align_up((int64_t)n0 * 8, 256));
  int* col_arg = (int*)((char*)row_arg + align_up((int64_t)n0 * 4, 256));
  hipLaunchKernelGGL(argmin_rows_f64_kernel, dim3(cdiv(n0, 4)), dim3(256), 0, st, d_dist, n0, n1, row_arg, row_min);
  hipLaunchKernelGGL(argmin_cols_f64_kernel, dim3(cdiv(n1, 256)), dim3(256), 0, st, d_dist, n0, n1, col_arg);
  hipLaunchKernelGGL(match_final_f64_kernel, dim3(cdiv(n0, 256)), dim3(256), 0, st, (const int*)row_arg, (const double*)row_min,
                     (const int*)col_arg, n0, thr, mutual, d_match01);
  LT_LAUNCH_CHECK();
  return LINETR_OK;
}

// subline2keyline (models/line_transformer.py:277-282) alone: Dk = A0 D A1^T for the segmented-mean matrices the tokeniser
// produces, given as sub-line -> key-line maps (non-decreasing).  Same kernel as linetr_match's pooling stage.
extern "C" int64_t linetr_pool_distmat_workspace_bytes(int32_t k0, int32_t k1) {
  return align_up(pair_scratch_ints(std::max(k0, 0), std::max(k1, 0)) * 4, 256) + 256;
}

extern "C" int linetr_pool_distmat(LinetrHandle* h, const float* d_dist, int32_t n0, int32_t n1, const int32_t* d_s2l0, int32_t k0,
                                   const int32_t* d_s2l1, int32_t k1, float* d_dk, void* d_ws, int64_t ws_bytes, void* stream) {
  if (n0 < 0 || n1 < 0 || k0 < 0 || k1 < 0 || k0 > n0 || k1 > n1) return fail(LINETR_E_ARG, "pool_distmat: bad dims");
  if (k0 == 0 || k1 == 0) return LINETR_OK;
  if (!d_dist || !d_s2l0 || !d_s2l1 || !d_dk || !d_ws) return fail(LINETR_E_ARG, "pool_distmat: null pointer");
  if (ws_bytes < linetr_pool_distmat_workspace_bytes(k0, k1)) return fail(LINETR_E_WORKSPACE, "pool_distmat: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  if (h) LT_HIP(hipSetDevice(h->device));
  PairTable tab{};
  tab.n_inline = 1;
  PairDesc* pd = tab.inl;
  pd->n0 = n0; pd->k0 = k0; pd->n1 = n1; pd->k1 = k1;
  pd->chunks = cdiv(k0, PM_ROWS);
  const int seg1_global = k1 > PM_MAX_K1;
  ProfScope ps(h, st, "pair_pool", 0, 4.0 * ((double)n0 * n1 + (double)k0 * k1));
  if (seg1_global) {
    hipLaunchKernelGGL(pair_seg1_kernel, dim3(cdiv(n1, 2048), 1), dim3(256), 0, st, tab, d_s2l1, (int*)d_ws);
    LT_LAUNCH_CHECK();
  }
  const int cache_dk = k1 <= PM_CACHE_K1;
  hipLaunchKernelGGL(pair_pool_kernel, dim3(pd->chunks, 1), dim3(256), pair_pool_lds(k1, seg1_global, cache_dk), st,
                     tab, d_s2l0, d_s2l1, d_dist, d_dk, (int*)d_ws, seg1_global, cache_dk);
  LT_LAUNCH_CHECK();
  return LINETR_OK;
}

// subline2keyline on the two mat_klines2sublines MATRICES, as the reference's call sites hand them over (models/matching.py:80:
// data['mat_klines2sublines0'][0], a [K,N] float32 tensor).  A tokeniser's matrix is reduced to its sub-line -> key-line map on the
// device and pooled by the segmented-mean kernel; any other matrix is multiplied out as given.  Asynchronous on `stream`.
namespace {
struct DensePoolLayout { int64_t o_map0, o_map1, o_val0, o_val1, o_verdict, o_tmp, o_pool, total; };
DensePoolLayout dense_pool_layout(int k0, int n0, int k1, int n1) {
  DensePoolLayout L{};
  int64_t o = 0;
  L.o_verdict = o; o += 256;
  L.o_map0 = o; o += align_up((int64_t)std::max(n0, 1) * 4, 256);
  L.o_map1 = o; o += align_up((int64_t)std::max(n1, 1) * 4, 256);
  L.o_val0 = o; o += align_up((int64_t)std::max(n0, 1) * 4, 256);
  L.o_val1 = o; o += align_up((int64_t)std::max(n1, 1) * 4, 256);
  L.o_tmp = o; o += align_up((int64_t)std::max(k0, 1) * std::max(n1, 1) * 4, 256);
  L.o_pool = o; o += linetr_pool_distmat_workspace_bytes(std::min(k0, n0), std::min(k1, n1));
  L.total = o;
  return L;
}
}  // namespace

extern "C" int64_t linetr_pool_distmat_dense_workspace_bytes(int32_t k0, int32_t n0, int32_t k1, int32_t n1) {
  return dense_pool_layout(std::max(k0, 0), std::max(n0, 0), std::max(k1, 0), std::max(n1, 0)).total;
}

extern "C" int linetr_pool_distmat_dense(LinetrHandle* h, const float* d_dist, int32_t n0, int32_t n1, const float* d_A0, int32_t k0,
                                         const float* d_A1, int32_t k1, float* d_dk, void* d_ws, int64_t ws_bytes, void* stream) {
  if (n0 < 0 || n1 < 0 || k0 < 0 || k1 < 0) return fail(LINETR_E_ARG, "pool_distmat_dense: bad dims");
  if (k0 > 65535) return fail(LINETR_E_ARG, "pool_distmat_dense: more than 65535 key-lines in image 0 (grid limit of the as-given product)");
  if (k0 == 0 || k1 == 0) return LINETR_OK;
  if (!d_dk || !d_ws || ((n0 > 0 && n1 > 0) && (!d_dist || !d_A0 || !d_A1))) return fail(LINETR_E_ARG, "pool_distmat_dense: null pointer");
  const DensePoolLayout L = dense_pool_layout(k0, n0, k1, n1);
  if (ws_bytes < L.total) return fail(LINETR_E_WORKSPACE, "pool_distmat_dense: workspace too small (need %lld)", (long long)L.total);
  hipStream_t st = (hipStream_t)stream;
  if (h) LT_HIP(hipSetDevice(h->device));
  char* base = (char*)d_ws;
  int* verdict = (int*)(base + L.o_verdict);
  if (n0 == 0 || n1 == 0) {                          // an empty inner dimension: the product is a zero matrix
    LT_HIP(hipMemsetAsync(d_dk, 0, (size_t)k0 * k1 * 4, st));
    return LINETR_OK;
  }
  // K > N cannot be a tokeniser's matrix (every key-line owns at least one sub-line): the verdict starts non-zero and the
  // segmented-mean launch is skipped
  const bool poolable = k0 <= n0 && k1 <= n1;
  LT_HIP(hipMemsetAsync(verdict, poolable ? 0 : 0x01, 4, st));
  if (poolable) {
    int* map0 = (int*)(base + L.o_map0); int* map1 = (int*)(base + L.o_map1);
    float* val0 = (float*)(base + L.o_val0); float* val1 = (float*)(base + L.o_val1);
    hipLaunchKernelGGL(mat_to_map_kernel, dim3(cdiv(n0, 256)), dim3(256), 0, st, d_A0, k0, n0, map0, val0, verdict);
    hipLaunchKernelGGL(mat_to_map_kernel, dim3(cdiv(n1, 256)), dim3(256), 0, st, d_A1, k1, n1, map1, val1, verdict);
    hipLaunchKernelGGL(mat_check_kernel, dim3(cdiv(n0, 256)), dim3(256), 0, st, (const int*)map0, (const float*)val0, k0, n0, verdict);
    hipLaunchKernelGGL(mat_check_kernel, dim3(cdiv(n1, 256)), dim3(256), 0, st, (const int*)map1, (const float*)val1, k1, n1, verdict);
    hipLaunchKernelGGL(map_sanitize_kernel, dim3(cdiv(n0, 256)), dim3(256), 0, st, map0, k0, n0, (const int*)verdict);
    hipLaunchKernelGGL(map_sanitize_kernel, dim3(cdiv(n1, 256)), dim3(256), 0, st, map1, k1, n1, (const int*)verdict);
    LT_LAUNCH_CHECK();
    if (int e = linetr_pool_distmat(h, d_dist, n0, n1, map0, k0, map1, k1, d_dk, base + L.o_pool, ws_bytes - L.o_pool, stream)) return e;
  }
  float* tmp = (float*)(base + L.o_tmp);
  hipLaunchKernelGGL(dense_pool_left_kernel, dim3(cdiv(n1, 256), k0), dim3(256), 0, st, d_A0, d_dist, tmp, n0, n1, (const int*)verdict);
  hipLaunchKernelGGL(dense_pool_right_kernel, dim3(cdiv(k1, 4), k0), dim3(256), 0, st, (const float*)tmp, d_A1, d_dk, n1, k1, (const int*)verdict);
  LT_LAUNCH_CHECK();
  return LINETR_OK;
}

// One rank's all-gather slab in one launch (layout: lt_match.h pack_slab_kernel, linetr_amd/parallel.py)
extern "C" int linetr_pack_slab(const float* d_line_desc, int32_t N, const int32_t* d_cu_n, const int32_t* d_cu_k, int32_t n_images,
                                const int32_t* d_sub2line, int32_t n_images_cap, int32_t rows_cap, int32_t zero_tail, float* d_slab,
                                void* stream) {
  if (N < 0 || n_images < 0 || n_images > n_images_cap || N > rows_cap) return fail(LINETR_E_CAPACITY, "pack_slab: %d images / %d rows exceed the slab capacity (%d / %d)", n_images, N, n_images_cap, rows_cap);
  if (!d_slab || (N > 0 && !d_line_desc) || (n_images > 0 && !d_cu_n)) return fail(LINETR_E_ARG, "pack_slab: null pointer");
  const int hr = (1 + 2 * n_images_cap + D - 1) / D, mr = (rows_cap + D - 1) / D;
  const int64_t items = (int64_t)(zero_tail ? rows_cap : N) * (D / 4) + (d_sub2line ? N : 0) + 1 + 2 * (int64_t)n_images_cap;
  hipLaunchKernelGGL(pack_slab_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_line_desc, N, d_cu_n,
                     d_cu_k, n_images, d_sub2line, n_images_cap, rows_cap, hr, mr, zero_tail, d_slab);
  LT_LAUNCH_CHECK();
  return LINETR_OK;
}

extern "C" int linetr_superpoint_heads(LinetrHandle* h, const float* d_score_logits, const float* d_desc_raw, int32_t B,
                                       int32_t Hc, int32_t Wc, float* d_dense_score, float* d_dense_desc_nhwc,
                                       float* d_dense_desc_nchw, void* stream) {
  if (B < 0 || Hc <= 0 || Wc <= 0) return fail(LINETR_E_ARG, "superpoint_heads: bad shape B=%d Hc=%d Wc=%d", B, Hc, Wc);
  if (d_dense_score && !d_score_logits) return fail(LINETR_E_ARG, "superpoint_heads: score output without score logits");
  if ((d_dense_desc_nhwc || d_dense_desc_nchw) && !d_desc_raw)
    return fail(LINETR_E_ARG, "superpoint_heads: descriptor output without the raw descriptor head");
  if (B == 0) return LINETR_OK;
  if (h) LT_HIP(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  const int HW = Hc * Wc;
  const dim3 grid((unsigned)cdiv(HW, 64), (unsigned)B);
  if (d_dense_desc_nhwc || d_dense_desc_nchw) {
    const double by = (double)B * HW * D * 4.0 * (1 + (d_dense_desc_nhwc ? 1 : 0) + (d_dense_desc_nchw ? 1 : 0));
    ProfScope ps(h, st, "sp_desc_head", 3.0 * B * HW * D, by);
    static const int cells = LT_XENV("LINETR_SP_CELLS") ? atoi(LT_XENV("LINETR_SP_CELLS")) : 32;   // tuning aid
    if (cells == 64) hipLaunchKernelGGL(sp_desc_head_kernel<64>, grid, dim3(256), 0, st, d_desc_raw, d_dense_desc_nhwc, d_dense_desc_nchw, HW);
    else hipLaunchKernelGGL(sp_desc_head_kernel<32>, dim3((unsigned)cdiv(HW, 32), (unsigned)B), dim3(256), 0, st, d_desc_raw, d_dense_desc_nhwc, d_dense_desc_nchw, HW);
    LT_LAUNCH_CHECK();
  }
  if (d_dense_score) {
    ProfScope ps(h, st, "sp_score_head", 0, (double)B * HW * (65 + 64) * 4.0);
    hipLaunchKernelGGL(sp_score_head_kernel, grid, dim3(256), 0, st, d_score_logits, d_dense_score, Hc, Wc);
    LT_LAUNCH_CHECK();
  }
  return LINETR_OK;
}
