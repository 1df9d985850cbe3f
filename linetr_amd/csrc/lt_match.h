// Descriptor-distance matcher on the device: get_dist_matrix (models/line_process.py:198-201),
// subline2keyline (models/line_transformer.py:277-282) and nn_matcher_distmat
// (models/nn_matcher.py:3-31).
#pragma once
#include "lt_common.h"

namespace lt {

struct PairDesc {          // one image pair (device copy lives in the workspace)
  int n0, k0, n1, k1;
  int64_t off_n0, off_n1;  // row offsets into desc0 / desc1
  int64_t off_s0, off_s1;  // element offsets into sub2line0 / sub2line1 (== off_n0 / off_n1 unless the caller says otherwise)
  int64_t off_d;           // offset of D [n0,n1] in the workspace
  int64_t off_dk;          // offset of Dk [k0,k1] in the output
  int64_t off_k0;          // offset of match01 [k0]
  int64_t off_seg;         // offset of the segment tables / argmin scratch (ints) in the workspace
  int32_t chunks;          // row chunks of pair_pool_kernel (cdiv(k0, PM_ROWS))
  int32_t pad_;
};

// Pair table handed to the matcher kernels: up to PT_INLINE descriptors travel inside the kernel arguments (no H2D copy
// on the latency path of a single pair), larger batches through a device array.
constexpr int PT_INLINE = 8;
struct PairTable {
  const PairDesc* ptr;
  int n_inline;
  PairDesc inl[PT_INLINE];
  __device__ __forceinline__ PairDesc get(int i) const { return n_inline ? inl[i] : ptr[i]; }
};

constexpr int PM_ROWS = 16;         // key-line rows of Dk per block of pair_pool_kernel
constexpr int PM_MAX_K1 = 12000;    // up to here the seg1 table of a pair lives in the block's LDS (48 KB); beyond, in the workspace
constexpr int PM_CACHE_K1 = 896;    // up to here the block's PM_ROWS pooled rows are kept in LDS as well (61 KB with the tables: under the 64 KB a launch gets without opting in)
// dynamic LDS of pair_pool_kernel
inline size_t pair_pool_lds(int max_k1, int seg1_global, int cache_dk) {
  return (size_t)((seg1_global ? 0 : max_k1 + 1) + PM_ROWS + 2 + (cache_dk ? PM_ROWS * max_k1 : 0)) * sizeof(int);
}

// D[a][b] = max(2 - 2 * <d0[a], d1[b]>, 0), fp32 MFMA, 64x64 tile per block, K = 256.
// grid (tiles_b, tiles_a, pair)
__global__ __launch_bounds__(256) void pair_dist_kernel(const PairTable pairs,
                                                        const float* __restrict__ desc0,
                                                        const float* __restrict__ desc1, float* __restrict__ dist) {
  constexpr int LS = 36;
  __shared__ __attribute__((aligned(16))) float As[64 * LS];
  __shared__ __attribute__((aligned(16))) float Bs[64 * LS];
  const PairDesc pd = pairs.get(blockIdx.z);
  const int a0 = blockIdx.y * 64, b0 = blockIdx.x * 64;
  if (a0 >= pd.n0 || b0 >= pd.n1) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave >> 1, wb = wave & 1;
  const float* A = desc0 + pd.off_n0 * D;
  const float* B = desc1 + pd.off_n1 * D;
  const int lrow = tid >> 3, lc4 = (tid & 7) * 4;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // rows of this thread's staging pieces; the NEXT K tile travels in registers while the current one is multiplied
  int ra[2], rb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    ra[i] = a0 + lrow + i * 32; ra[i] = ra[i] < pd.n0 ? ra[i] : pd.n0 - 1;
    rb[i] = b0 + lrow + i * 32; rb[i] = rb[i] < pd.n1 ? rb[i] : pd.n1 - 1;
  }
  f32x4 va[2], vb[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      va[i] = *reinterpret_cast<const f32x4*>(A + (int64_t)ra[i] * D + k0 + lc4);
      vb[i] = *reinterpret_cast<const f32x4*>(B + (int64_t)rb[i] * D + k0 + lc4);
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < D; k0 += 32) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<f32x4*>(&As[(lrow + i * 32) * LS + lc4]) = va[i];
      *reinterpret_cast<f32x4*>(&Bs[(lrow + i * 32) * LS + lc4]) = vb[i];
    }
    if (k0 + 32 < D) fetch(k0 + 32);
    __syncthreads();
    const float* ap = &As[(wa * 32 + (lane & 31)) * LS + (lane >> 5) * 4];
    const float* bp = &Bs[(wb * 32 + (lane & 31)) * LS + (lane >> 5) * 4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ap + kk * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(bp + kk * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
    }
  }
  float* Dp = dist + pd.off_d;
  const int col = b0 + wb * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = a0 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < pd.n0 && col < pd.n1) Dp[(int64_t)row * pd.n1 + col] = fmaxf(2.f - 2.f * acc[r], 0.f);
  }
}

// ---------------------------------------------------------------------------------------------
// Key-line pooling + mutual nearest neighbour (subline2keyline + nn_matcher_distmat), two launches (a single block per
// pair took 150 us for one 200 x 200 pair):
//   pair_pool_kernel   grid (row chunks, pairs): PM_ROWS key-line rows of Dk each -- segmented mean, row argmin, and the
//                      chunk's partial column argmin;
//   pair_final_kernel  grid (pairs): column partials combined in chunk order (first index wins), threshold + mutual check.
// Deterministic: fixed summation order, no atomics; np.argmin's first-minimum rule on rows and columns.
// Sub-lines of a key-line are contiguous and key-line ids are non-decreasing, so a segment start is a lower bound.
// Scratch ints of a pair at off_seg: row_arg[k0] | row_min[k0] | col_arg[k1] | part_val[chunks][k1] | part_arg[chunks][k1] | seg1[k1+1]
// (seg1: only used when an image has more than PM_MAX_K1 key-lines / key-points and the table does not fit the LDS)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int seg_lower_bound(const int* __restrict__ m, int n, int key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (m[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// seg1 table of a pair in global scratch (images with more than PM_MAX_K1 key-lines / key-points): grid (blocks, pairs)
__global__ __launch_bounds__(256) void pair_seg1_kernel(const PairTable pairs, const int* __restrict__ s2l1, int* __restrict__ scratch) {
  const PairDesc pd = pairs.get(blockIdx.y);
  if (pd.k1 <= 0) return;
  int* seg1 = scratch + pd.off_seg + 2 * (int64_t)pd.k0 + pd.k1 + 2 * (int64_t)pd.chunks * pd.k1;
  const int* m1 = s2l1 + pd.off_s1;
  for (int n = blockIdx.x * 256 + threadIdx.x; n < pd.n1; n += gridDim.x * 256)
    if (n == 0 || m1[n] != m1[n - 1]) seg1[m1[n]] = n;
  if (blockIdx.x == 0 && threadIdx.x == 0) seg1[pd.k1] = pd.n1;
}

__global__ __launch_bounds__(256) void pair_pool_kernel(const PairTable pairs, const int* __restrict__ s2l0,
                                                        const int* __restrict__ s2l1, const float* __restrict__ dist,
                                                        float* __restrict__ dk_out, int* __restrict__ scratch, int seg1_global, int cache_dk) {
  extern __shared__ int pm_lds[];                    // seg1[k1+1] | seg0[PM_ROWS+2] | (cache_dk) pooled rows [PM_ROWS][k1]   (seg1_global: no seg1)
  const PairDesc pd = pairs.get(blockIdx.y);
  const int chunk = blockIdx.x;
  if (chunk >= pd.chunks || pd.k1 <= 0) return;
  const int tid = threadIdx.x;
  const int i0 = chunk * PM_ROWS, rows = min(PM_ROWS, pd.k0 - i0);
  // large images: the seg1 table does not fit the LDS; pair_seg1_kernel has built it in the pair's scratch region (one writer,
  // an earlier launch on the same stream) and the blocks of this launch only read it
  int* seg1 = seg1_global ? scratch + pd.off_seg + 2 * (int64_t)pd.k0 + pd.k1 + 2 * (int64_t)pd.chunks * pd.k1 : pm_lds;
  int* seg0 = seg1_global ? pm_lds : pm_lds + pd.k1 + 1;
  const int* m0 = s2l0 + pd.off_s0;
  const int* m1 = s2l1 + pd.off_s1;
  // segment starts: sub-lines of a key-line are contiguous and key-line ids non-decreasing, so a start is where the id
  // changes -- one coalesced pass (a chain of dependent global loads per binary search cost ~5 us of pure latency)
  if (!seg1_global)
    for (int n = tid; n < pd.n1; n += 256)
      if (n == 0 || m1[n] != m1[n - 1]) seg1[m1[n]] = n;
  for (int n = tid; n < pd.n0; n += 256)
    if (n == 0 || m0[n] != m0[n - 1]) {
      const int k = m0[n] - i0;
      if (k >= 0 && k <= rows) seg0[k] = n;
    }
  if (tid == 0) {
    if (!seg1_global) seg1[pd.k1] = pd.n1;
    if (i0 + rows == pd.k0) seg0[rows] = pd.n0;
  }
  __syncthreads();
  const float* Dp = dist + pd.off_d;
  float* Dk = dk_out + pd.off_dk + (int64_t)i0 * pd.k1;
  const int total = rows * pd.k1;
  // the pooled rows also stay in LDS when they fit (cache_dk: the caller sized the dynamic LDS for PM_ROWS x k1 floats): both
  // argmin passes below then read LDS instead of the Dk rows this block has just stored
  float* dkc = cache_dk ? reinterpret_cast<float*>(pm_lds + (seg1_global ? 0 : pd.k1 + 1) + PM_ROWS + 2) : nullptr;
  // four elements per thread and pass: their distance loads are independent and in flight together (one element per pass exposed a
  // full L2 round trip per iteration: 13 dependent round trips for a 199 x 199 pair)
  for (int e0 = tid; e0 < total; e0 += 256 * 4) {
    float v[4];
    int seg[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = min(e0 + u * 256, total - 1);
      const int r = e / pd.k1, j = e - r * pd.k1;
      seg[u][0] = seg0[r]; seg[u][1] = seg0[r + 1]; seg[u][2] = seg1[j]; seg[u][3] = seg1[j + 1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = Dp[(int64_t)seg[u][0] * pd.n1 + seg[u][2]];      // the whole answer for 1 x 1 segments
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int a0 = seg[u][0], a1 = seg[u][1], b0 = seg[u][2], b1 = seg[u][3];
      if (a1 - a0 != 1 || b1 - b0 != 1) {  // (A0 @ D) @ A1^T with A rows = 1/num_sublines
        const float w0 = 1.f / (float)(a1 - a0), w1 = 1.f / (float)(b1 - b0);
        float acc = 0.f;
        for (int b = b0; b < b1; ++b) {
          float t = 0.f;
          for (int a = a0; a < a1; ++a) t += w0 * Dp[(int64_t)a * pd.n1 + b];
          acc += t * w1;
        }
        v[u] = acc;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * 256;
      if (e < total) {
        Dk[e] = v[u];
        if (dkc) dkc[e] = v[u];
      }
    }
  }
  __syncthreads();
  const float* Dr = dkc ? dkc : Dk;       // where the argmin passes read the pooled rows
  int* row_arg = scratch + pd.off_seg;
  float* row_min = reinterpret_cast<float*>(row_arg + pd.k0);
  float* part_val = reinterpret_cast<float*>(row_arg + 2 * pd.k0 + pd.k1) + (int64_t)chunk * pd.k1;
  int* part_arg = row_arg + 2 * pd.k0 + pd.k1 + (int64_t)pd.chunks * pd.k1 + (int64_t)chunk * pd.k1;
  {
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < rows; r += 4) {
      float best = INFINITY; int arg = 0x7fffffff;
      for (int j = lane; j < pd.k1; j += 64) {
        const float v = fmaxf(Dr[(int64_t)r * pd.k1 + j], 0.f);
        if (v < best) { best = v; arg = j; }       // ascending j per lane: strict < keeps the first
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oa = __shfl_xor(arg, o, 64);
        if (ov < best || (ov == best && oa < arg)) { best = ov; arg = oa; }
      }
      if (lane == 0) { row_arg[i0 + r] = arg == 0x7fffffff ? 0 : arg; row_min[i0 + r] = best; }   // all-inf / NaN row: index 0
    }
  }
  for (int j = tid; j < pd.k1; j += 256) {
    float best = INFINITY; int arg = 0;
    for (int r = 0; r < rows; ++r) {
      const float v = fmaxf(Dr[(int64_t)r * pd.k1 + j], 0.f);
      if (v < best) { best = v; arg = i0 + r; }
    }
    part_val[j] = best;
    part_arg[j] = arg;
  }
}

__global__ __launch_bounds__(256) void pair_final_kernel(const PairTable pairs, float thr, int mutual,
                                                         int* __restrict__ match01, int* __restrict__ scratch) {
  const PairDesc pd = pairs.get(blockIdx.x);
  const int tid = threadIdx.x;
  int* row_arg = scratch + pd.off_seg;
  const float* row_min = reinterpret_cast<const float*>(row_arg + pd.k0);
  int* col_arg = row_arg + 2 * pd.k0;
  const float* part_val = reinterpret_cast<const float*>(col_arg + pd.k1);
  const int* part_arg = col_arg + pd.k1 + (int64_t)pd.chunks * pd.k1;
  for (int j = tid; j < pd.k1; j += 256) {
    float best = INFINITY; int arg = 0;
    int c = 0;
    for (; c + 4 <= pd.chunks; c += 4) {             // four independent loads in flight; ascending rows: strict < keeps the first
      float v[4]; int a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { v[u] = part_val[(int64_t)(c + u) * pd.k1 + j]; a[u] = part_arg[(int64_t)(c + u) * pd.k1 + j]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) if (v[u] < best) { best = v[u]; arg = a[u]; }
    }
    for (; c < pd.chunks; ++c) {
      const float v = part_val[(int64_t)c * pd.k1 + j];
      if (v < best) { best = v; arg = part_arg[(int64_t)c * pd.k1 + j]; }
    }
    col_arg[j] = arg;
  }
  __syncthreads();
  int* mo = match01 + pd.off_k0;
  for (int i = tid; i < pd.k0; i += 256) {
    int j = -1;
    if (pd.k1 > 0) {
      const int a = row_arg[i];
      bool keep = row_min[i] < thr;
      if (mutual) keep = keep && (col_arg[a] == i);
      if (keep) j = a;
    }
    mo[i] = j;
  }
}

// ---------------------------------------------------------------------------------------------
// subline2keyline (models/line_transformer.py:277-282) for a mat_klines2sublines given as the MATRIX the reference passes around
// ([K,N] float32) instead of this library's sub-line -> key-line map: the matrix is reduced to the map when it is one the
// tokeniser writes (line_process.py:163-167: one non-zero per column, key-lines in order, every entry of a row the float32 of
// 1 / num_sublines) and pooled by pair_pool_kernel; ANY other matrix is multiplied out as given (two plain fp32 passes).
// Everything is decided on the device (verdict word in the workspace): the entry point stays asynchronous.
//   verdict bits: 1 a column without exactly one non-zero, 2 rows not in order / a key-line without sub-lines, 4 an entry that
//   is not float32(1 / num_sublines)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mat_to_map_kernel(const float* __restrict__ A, int K, int N, int* __restrict__ map,
                                                         float* __restrict__ vals, int* __restrict__ verdict) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  int cnt = 0, row = 0;
  float v = 0.f;
  for (int k = 0; k < K; ++k) {                       // adjacent threads read adjacent columns: coalesced rows
    const float a = A[(int64_t)k * N + n];
    if (a != 0.f) { ++cnt; row = k; v = a; }          // (a NaN entry counts as non-zero and fails the value check below)
  }
  map[n] = row;
  vals[n] = v;
  if (cnt != 1) atomicOr(verdict, 1);
}

__global__ __launch_bounds__(256) void mat_check_kernel(const int* __restrict__ map, const float* __restrict__ vals, int K, int N,
                                                        int* __restrict__ verdict) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const int prev = n ? map[n - 1] : -1, cur = map[n];
  int bad = 0;
  if (cur != prev && cur != prev + 1) bad |= 2;       // key-line ids start at 0 and grow in steps of one
  if (n == N - 1 && cur != K - 1) bad |= 2;
  if (cur != prev) {                                  // first sub-line of a key-line: every entry must be float32(1 / count)
    int len = 1;
    while (n + len < N && map[n + len] == cur) ++len;
    const float w = (float)(1.0 / (double)len);       // the tokeniser's value (line_process.py:165: a Python float stored as float32)
    for (int i = 0; i < len; ++i) if (!(vals[n + i] == w)) bad |= 4;
  }
  if (bad) atomicOr(verdict, bad);
}

// a matrix that is not a tokeniser's: give pair_pool_kernel a well-formed map to run on (its output is overwritten below)
__global__ __launch_bounds__(256) void map_sanitize_kernel(int* __restrict__ map, int K, int N, const int* __restrict__ verdict) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n < N && *verdict) map[n] = n < K - 1 ? n : K - 1;
}

// tmp[i][m] = sum_n A0[i][n] * D[n][m]   (the reference's left product, as given); grid (cdiv(n1,256), k0)
__global__ __launch_bounds__(256) void dense_pool_left_kernel(const float* __restrict__ A0, const float* __restrict__ Dm, float* __restrict__ tmp,
                                                              int n0, int n1, const int* __restrict__ verdict) {
  if (*verdict == 0) return;
  const int m = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (m >= n1) return;
  float acc = 0.f;
  for (int n = 0; n < n0; ++n) acc += A0[(int64_t)i * n0 + n] * Dm[(int64_t)n * n1 + m];
  tmp[(int64_t)i * n1 + m] = acc;
}

// Dk[i][j] = sum_m tmp[i][m] * A1[j][m]; grid (cdiv(k1,4), k0): one wave per (i, j), lanes over m (coalesced), wave reduction
__global__ __launch_bounds__(256) void dense_pool_right_kernel(const float* __restrict__ tmp, const float* __restrict__ A1, float* __restrict__ Dk,
                                                               int n1, int k1, const int* __restrict__ verdict) {
  if (*verdict == 0) return;
  const int lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6), i = blockIdx.y;
  if (j >= k1) return;
  float acc = 0.f;
  for (int m = lane; m < n1; m += 64) acc += tmp[(int64_t)i * n1 + m] * A1[(int64_t)j * n1 + m];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) Dk[(int64_t)i * k1 + j] = acc;
}

// ---------------------------------------------------------------------------------------------
// ONE pair in ONE launch (the latency path of Matching.forward / LineTransformer's matching tail: get_dist_matrix +
// subline2keyline + nn_matcher_distmat, models/line_process.py:198-201, models/line_transformer.py:277-282,
// models/nn_matcher.py:3-31).  grid = cdiv(k0, PM_ROWS) blocks of 8 waves; a block owns PM_ROWS key-lines of image 0:
//   1. segment tables of both images from the sub-line -> key-line maps (one coalesced pass each);
//   2. the distance rows of its sub-lines against ALL sub-lines of image 1, 16 rows at a time, by exact-fp32 MFMA
//      (v_mfma_f32_16x16x4_f32) with NO staging: a lane fetches the 64 consecutive channels of its own row that the K order
//      {64 (lane / 16) + s} assigns to it -- 16 independent dwordx4 loads, all in flight at once, operands straight in
//      registers -- and two column tiles are multiplied interleaved (the MFMA's 40-cycle dependent latency hides behind the
//      other accumulator).  Two exposed memory round trips in the whole kernel (the r03 attempt paid one per K step);
//   3. t[r][b] = sum_a w0 D[a][b] as the rows arrive, then Dk[r][j] = sum_b t[r][b] w1 -- the summation order of pair_pool_kernel;
//   4. row argmin per key-line; the column argmin across blocks through ONE 64-bit atomicMin per column on the packed
//      (distance bits, row) key: distances are >= +0, so the unsigned order of the key is (distance, first row) -- np.argmin's rule;
//   5. the last block to arrive (agent-scope counter) applies the threshold and the mutual check and leaves the slot clean
//      (column keys all-ones, counter 0) for the next call.  Cross-block traffic is 8-byte agent-scope atomics on both sides
//      (MI355X_MICROARCH.md, "Valid forms"): no fences.
// Deterministic: min is order-independent, every sum has a fixed order.
// ---------------------------------------------------------------------------------------------
struct PairSlot {                 // library-owned scratch of the one-launch matcher, one per (device, stream); self-cleaning
  unsigned long long* col_best;   // [PF_MAX_K] packed (distance bits << 32 | row), all-ones when idle
  unsigned long long* row_res;    // [PF_MAX_K] packed (row minimum bits << 32 | column)
  unsigned* counter;              // arrivals, 0 when idle
};
constexpr int PF_MAX_K = 4096;    // key-lines per image the slot holds
constexpr int PF_MAX_N1 = 1024;   // sub-lines of image 1: t [PM_ROWS][n1] + a 16-row distance tile must fit the LDS
inline size_t pair_fused_lds(int n1, int k1) {
  const int n1p = (n1 + 15) / 16 * 16;
  return (size_t)((k1 + 1) + (PM_ROWS + 2) + 2 * PM_ROWS * n1p) * sizeof(float) + 64;
}

typedef float f32x4v __attribute__((ext_vector_type(4)));

// IDENT: every key-line of both images has exactly one sub-line (n == k on both sides: the point matcher always, line pairs whose
// lines are all shorter than token_distance x max_tokens): Dk IS D, so the segment tables, the t stage and the pooling stage
// disappear, the row operands are requested at the first instruction, and the columns are split over gridDim.y blocks (the row
// argmin is then combined across blocks like the column argmin, by atomicMin on the packed key).
template <bool IDENT>
__global__ __launch_bounds__(512) void pair_match_fused_kernel(const float* __restrict__ desc0, const float* __restrict__ desc1,
                                                               const int* __restrict__ m0, const int* __restrict__ m1, int n0, int k0,
                                                               int n1, int k1, float thr, int mutual, float* __restrict__ dk_out,
                                                               int* __restrict__ match01, PairSlot slot) {
  extern __shared__ int pf_lds[];
  const int n1p = (n1 + 15) / 16 * 16;
  int* seg1 = pf_lds;                                            // [k1 + 1]
  int* seg0 = seg1 + k1 + 1;                                     // [PM_ROWS + 1] (+1 pad)
  float* tbuf = reinterpret_cast<float*>(seg0 + PM_ROWS + 2);    // [PM_ROWS][n1p]
  float* dt = tbuf + PM_ROWS * n1p;                              // [16][n1p]; later the pooled rows [PM_ROWS][k1]
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.x * PM_ROWS, rows = min(PM_ROWS, k0 - i0);
  const int lr = lane & 15, lg = lane >> 4;
  const int n_ct = n1p / 16;                                     // column tiles of 16 sub-lines
  // column tiles of this block (IDENT: split over gridDim.y) and of this wave
  const int ct_lo = IDENT ? (int)blockIdx.y * ((n_ct + (int)gridDim.y - 1) / (int)gridDim.y) : 0;
  const int ct_hi = IDENT ? min(n_ct, ct_lo + (n_ct + (int)gridDim.y - 1) / (int)gridDim.y) : n_ct;
  // ---- column operands of this wave's first two tiles: independent of everything else, requested first ------------------
  f32x4v b0[16], b1[16];
  const int ct0 = ct_lo + wave, ct1 = ct0 + 8;
  {
    const int c0 = min(ct0 * 16 + lr, n1 - 1), c1 = min(ct1 * 16 + lr, n1 - 1);
    const f32x4v* p0 = reinterpret_cast<const f32x4v*>(desc1 + (int64_t)c0 * D + 64 * lg);
    const f32x4v* p1 = reinterpret_cast<const f32x4v*>(desc1 + (int64_t)c1 * D + 64 * lg);
    if (ct0 < ct_hi) {
#pragma unroll
      for (int q = 0; q < 16; ++q) b0[q] = p0[q];
    }
    if (ct1 < ct_hi) {
#pragma unroll
      for (int q = 0; q < 16; ++q) b1[q] = p1[q];
    }
  }
  int a_lo = i0, a_hi = i0 + rows;
  if constexpr (!IDENT) {
    // ---- 1. segment starts (sub-lines of a key-line are contiguous, ids non-decreasing: a start is where the id changes) -----
    for (int n = tid; n < n1; n += 512)
      if (n == 0 || m1[n] != m1[n - 1]) seg1[m1[n]] = n;
    for (int n = tid; n < n0; n += 512)
      if (n == 0 || m0[n] != m0[n - 1]) {
        const int k = m0[n] - i0;
        if (k >= 0 && k <= rows) seg0[k] = n;
      }
    if (tid == 0) {
      seg1[k1] = n1;
      if (i0 + rows == k0) seg0[rows] = n0;
    }
    for (int e = tid; e < PM_ROWS * n1p; e += 512) tbuf[e] = 0.f;
    __syncthreads();
    a_lo = seg0[0]; a_hi = seg0[rows];
  }
  // ---- 2. + 3a. distance rows, 16 sub-lines of image 0 at a time ---------------------------------------------------------
  for (int a0 = a_lo; a0 < a_hi; a0 += 16) {
    f32x4v av[16];
    {
      const int ar = min(a0 + lr, n0 - 1);
      const f32x4v* pa = reinterpret_cast<const f32x4v*>(desc0 + (int64_t)ar * D + 64 * lg);
#pragma unroll
      for (int q = 0; q < 16; ++q) av[q] = pa[q];
    }
    for (int ct = ct0; ct < ct_hi; ct += 16) {
      const int ctb = ct + 8;
      const bool two = ctb < ct_hi;                              // wave-uniform
      if (ct != ct0 || a0 != a_lo) {                             // beyond the prefetched pair (or a later row tile): fetch now
        const int c0 = min(ct * 16 + lr, n1 - 1), c1 = min(ctb * 16 + lr, n1 - 1);
        const f32x4v* p0 = reinterpret_cast<const f32x4v*>(desc1 + (int64_t)c0 * D + 64 * lg);
        const f32x4v* p1 = reinterpret_cast<const f32x4v*>(desc1 + (int64_t)c1 * D + 64 * lg);
#pragma unroll
        for (int q = 0; q < 16; ++q) b0[q] = p0[q];
        if (two) {
#pragma unroll
          for (int q = 0; q < 16; ++q) b1[q] = p1[q];
        }
      }
      f32x4v c0v = {0.f, 0.f, 0.f, 0.f}, c1v = {0.f, 0.f, 0.f, 0.f};
      if (two) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            c0v = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][e], b0[q][e], c0v, 0, 0, 0);
            c1v = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][e], b1[q][e], c1v, 0, 0, 0);
          }
      } else {
        // one tile: two accumulators over alternating K steps (a single chain waits the MFMA's 40-cycle dependent latency on every
        // step), added at the end in a fixed order
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            c0v = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][e], b0[q][e], c0v, 0, 0, 0);
            c1v = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][e + 1], b0[q][e + 1], c1v, 0, 0, 0);
          }
#pragma unroll
        for (int i = 0; i < 4; ++i) c0v[i] += c1v[i];
      }
      // C/D layout of the 16 x 16 tile: lane = column lr, registers = rows 4 lg + i
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dt[(4 * lg + i) * n1p + ct * 16 + lr] = fmaxf(2.f - 2.f * c0v[i], 0.f);
        if (two) dt[(4 * lg + i) * n1p + ctb * 16 + lr] = fmaxf(2.f - 2.f * c1v[i], 0.f);
      }
    }
    __syncthreads();
    if constexpr (!IDENT) {
      // t[r][b] += w0 D[a][b], a ascending inside its key-line (pair_pool_kernel's order)
      const int a_end = min(a0 + 16, a_hi);
      for (int b = tid; b < n1; b += 512) {
        int r = 0;
        for (int a = a0; a < a_end; ++a) {
          while (a >= seg0[r + 1]) ++r;
          const float w0 = 1.f / (float)(seg0[r + 1] - seg0[r]);
          tbuf[r * n1p + b] += w0 * dt[(a - a0) * n1p + b];
        }
      }
      __syncthreads();
    }
  }
  // ---- 3b. Dk rows of this block (global + LDS: `dks`, row stride `ldk`, columns j_lo .. j_hi) --------------------------------
  const float* dks = dt;
  int ldk = n1p, j_lo = 0, j_hi = k1;
  if constexpr (IDENT) {
    j_lo = ct_lo * 16; j_hi = min(k1, ct_hi * 16);
    const int w = j_hi - j_lo;
    for (int e = tid; e < rows * w; e += 512) {
      const int r = e / w, j = j_lo + e - r * w;
      dk_out[(int64_t)(i0 + r) * k1 + j] = dt[r * n1p + j];
    }
  } else {
    float* dkw = dt;
    ldk = k1;
    const int total = rows * k1;
    for (int e = tid; e < total; e += 512) {
      const int r = e / k1, j = e - r * k1;
      const int bb0 = seg1[j], bb1 = seg1[j + 1];
      const float w1 = 1.f / (float)(bb1 - bb0);
      float acc = 0.f;
      for (int b = bb0; b < bb1; ++b) acc += tbuf[r * n1p + b] * w1;
      dk_out[(int64_t)(i0 + r) * k1 + j] = acc;
      dkw[e] = acc;              // (e = r k1 + j: element e of a row tile is never read as D again -- the t stage is complete)
    }
    __syncthreads();
  }
  // ---- 4. argmins: both combined across blocks by a 64-bit atomicMin on (distance bits, index) ----------------------------------
  for (int r = wave; r < rows; r += 8) {
    float best = INFINITY; int arg = 0x7fffffff;
    for (int j = j_lo + lane; j < j_hi; j += 64) {
      const float v = fmaxf(dks[r * ldk + j], 0.f) + 0.f;          // (+0: a -0 would order after every positive as a key)
      if (v < best) { best = v; arg = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(best, o, 64);
      const int oa = __shfl_xor(arg, o, 64);
      if (ov < best || (ov == best && oa < arg)) { best = ov; arg = oa; }
    }
    if (lane == 0 && arg != 0x7fffffff) {                          // (an all-inf row leaves the all-ones key: column 0xffffffff, rejected below)
      const unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)arg;
      __hip_atomic_fetch_min(slot.row_res + i0 + r, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  for (int j = j_lo + tid; j < j_hi; j += 512) {
    float best = INFINITY; int arg = 0;
    for (int r = 0; r < rows; ++r) {
      const float v = fmaxf(dks[r * ldk + j], 0.f) + 0.f;
      if (v < best) { best = v; arg = i0 + r; }
    }
    const unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)arg;
    __hip_atomic_fetch_min(slot.col_best + j, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- 5. arrival; the last block finishes -----------------------------------------------------------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0)
    s_last = __hip_atomic_fetch_add(slot.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x * gridDim.y - 1;
  __syncthreads();
  if (!s_last) return;
  for (int i = tid; i < k0; i += 512) {
    // read AND reset through one atomic exchange: the value every block's atomicMin left, wherever it was executed
    const unsigned long long rr = __hip_atomic_exchange(slot.row_res + i, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned a = (unsigned)(rr & 0xffffffffull);
    bool keep = a < (unsigned)k1 && __uint_as_float((unsigned)(rr >> 32)) < thr;
    const int aa = a < (unsigned)k1 ? (int)a : 0;
    if (mutual) {
      const unsigned long long cb = __hip_atomic_load(slot.col_best + aa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      keep = keep && ((int)(unsigned)(cb & 0xffffffffull) == i);
    }
    match01[i] = keep ? aa : -1;
  }
  __syncthreads();
  for (int j = tid; j < k1; j += 512) __hip_atomic_store(slot.col_best + j, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) __hip_atomic_store(slot.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
// nn_matcher_distmat (models/nn_matcher.py:3-31) on a float64 matrix, compared IN float64 like NumPy does with one: clip(min = 0),
// first-index argmin of rows and columns, strict `<` against the threshold, optional mutual check.  (The reference's Matching
// passes float32, which the kernels above serve; this is the form a caller with a float64 matrix gets, instead of a rounding to
// float32 that could merge two distinct distances.)  Cold path: plain kernels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmin_rows_f64_kernel(const double* __restrict__ d, int n0, int n1, int* __restrict__ row_arg,
                                                              double* __restrict__ row_min) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n0) return;
  double best = INFINITY; int arg = 0x7fffffff;
  for (int j = lane; j < n1; j += 64) {
    const double v = fmax(d[(int64_t)r * n1 + j], 0.0);
    if (v < best) { best = v; arg = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(best, o, 64);
    const int oa = __shfl_xor(arg, o, 64);
    if (ov < best || (ov == best && oa < arg)) { best = ov; arg = oa; }
  }
  if (lane == 0) { row_arg[r] = arg == 0x7fffffff ? 0 : arg; row_min[r] = best; }
}

__global__ __launch_bounds__(256) void argmin_cols_f64_kernel(const double* __restrict__ d, int n0, int n1, int* __restrict__ col_arg) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n1) return;
  double best = INFINITY; int arg = 0;
  for (int r = 0; r < n0; ++r) {                       // ascending rows, strict <: the first minimum
    const double v = fmax(d[(int64_t)r * n1 + j], 0.0);
    if (v < best) { best = v; arg = r; }
  }
  col_arg[j] = arg;
}

__global__ __launch_bounds__(256) void match_final_f64_kernel(const int* __restrict__ row_arg, const double* __restrict__ row_min,
                                                              const int* __restrict__ col_arg, int n0, double thr, int mutual,
                                                              int* __restrict__ match01) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n0) return;
  const int a = row_arg[i];
  bool keep = row_min[i] < thr;
  if (mutual) keep = keep && col_arg[a] == i;
  match01[i] = keep ? a : -1;
}

// [256][n] (SuperPoint 'descriptors' layout) -> [n][256]
__global__ void transpose_cn_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int n) {
  __shared__ float tile[32][33];
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    int p = p0 + threadIdx.x;
    tile[i][threadIdx.x] = p < n ? in[(int64_t)(c0 + i) * n + p] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    int p = p0 + i;
    if (p < n) out[(int64_t)p * C + c0 + threadIdx.x] = tile[threadIdx.x][i];
  }
}

// ---------------------------------------------------------------------------------------------
// One rank's slab of the multi-GPU all-gather (linetr_amd/parallel.py): float32 [hr + mr + rows_cap][256] =
// int32 header {n_images, n_0.., k_0..} | int32 sub2line[rows_cap] | line_desc rows (tail rows zeroed when asked).
// Counts come from the device prefix sums linetr_describe already holds: nothing is copied from the host.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_slab_kernel(const float* __restrict__ desc, int N, const int* __restrict__ cu_n,
                                                        const int* __restrict__ cu_k, int n_images, const int* __restrict__ s2l,
                                                        int n_images_cap, int rows_cap, int hr, int mr, int zero_tail,
                                                        float* __restrict__ slab) {
  const int64_t row_items = (int64_t)(zero_tail ? rows_cap : N) * (D / 4);
  const int64_t map_items = s2l ? N : 0;
  const int64_t hdr_items = 1 + 2 * (int64_t)n_images_cap;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < row_items) {
    const int64_t r = i / (D / 4);
    const f32x4 v = r < N ? reinterpret_cast<const f32x4*>(desc)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    reinterpret_cast<f32x4*>(slab + (int64_t)(hr + mr) * D)[i] = v;
  } else if (i < row_items + map_items) {
    const int64_t j = i - row_items;
    reinterpret_cast<int*>(slab + (int64_t)hr * D)[j] = s2l[j];
  } else if (i < row_items + map_items + hdr_items) {
    const int j = (int)(i - row_items - map_items);
    int* hdr = reinterpret_cast<int*>(slab);
    int v = 0;
    if (j == 0) v = n_images;
    else if (j <= n_images_cap) { const int im = j - 1; v = im < n_images ? cu_n[im + 1] - cu_n[im] : 0; }
    else { const int im = j - 1 - n_images_cap; v = (cu_k && im < n_images) ? cu_k[im + 1] - cu_k[im] : 0; }
    hdr[j] = v;
  }
}

}  // namespace lt
