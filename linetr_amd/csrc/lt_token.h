// Key-line -> sub-line tokenisation, descriptor sampling and score gather on the device.
// Replaces the Python double loop of models/line_process.py:100-196 (line_tokenizer),
// :86-98 (sample_descriptors) and :174-179 (score gather).
#pragma once
#include "lt_common.h"

namespace lt {

// ---------------------------------------------------------------------------------------------
// NCHW -> NHWC copy of the dense descriptor map so that each bilinear tap is one coalesced 1 KiB
// row (SuperPoint emits [1,256,H/8,W/8], models/superpoint.py:193).  32x32 LDS-tiled transpose.
// grid (P/64 ceil, C/64, B), block 256
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int C, int P) {
  // 64 channels x 64 positions per block, dwordx4 on both sides: reads 16 B along p (NCHW rows), writes 16 B along c
  // (NHWC rows).  LDS tile [64 c][64 p + 1]: the odd stride keeps the 4-way column gather conflict-free.
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const float* src = in + (int64_t)b * C * P;
  float* dst = out + (int64_t)b * C * P;
  const int tid = threadIdx.x;
  {
    const int pq = (tid & 15) * 4;          // 16 threads cover 64 positions
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = (tid >> 4) + i * 16;
      const float* row = src + (int64_t)(c0 + c) * P + p0 + pq;
      f32x4 v;
      if (p0 + pq + 3 < P && (((uintptr_t)row) & 15) == 0) v = *reinterpret_cast<const f32x4*>(row);
      else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = p0 + pq + k < P ? row[k] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) tile[c][pq + k] = v[k];
    }
  }
  __syncthreads();
  {
    const int cq = (tid & 15) * 4;          // 16 threads cover 64 channels of one position
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = (tid >> 4) + i * 16;
      if (p0 + p < P) {
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = tile[cq + k][p];
        *reinterpret_cast<f32x4*>(dst + (int64_t)(p0 + p) * C + c0 + cq) = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// per key-line: float32 copies of klines/length/angles (models/line_process.py:182-184, with the
// end-point clip of :114-116 already applied) and the sub-line -> key-line maps.
// ---------------------------------------------------------------------------------------------
__global__ void line_fill_kernel(const LinetrLineRec* __restrict__ recs, int K, double wclip, double hclip,
                                 float* __restrict__ klines, float* __restrict__ length,
                                 float* __restrict__ angles, int* __restrict__ sub2line_g,
                                 int* __restrict__ sub2line_l) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const LinetrLineRec r = recs[k];
  if (klines) {
    klines[k * 4 + 0] = (float)r.sp[0];
    klines[k * 4 + 1] = (float)r.sp[1];
    klines[k * 4 + 2] = (float)fmin(r.ep[0], wclip);
    klines[k * 4 + 3] = (float)fmin(r.ep[1], hclip);
    length[k] = (float)r.length;
    angles[k * 2 + 0] = (float)r.angle[0];
    angles[k * 2 + 1] = (float)r.angle[1];
  }
  for (int s = 0; s < r.n_sub; ++s) {
    sub2line_g[r.first_sub + s] = k;
    if (sub2line_l) sub2line_l[r.first_sub + s] = r.line_local;
  }
}

// mat_klines2sublines of a small batch: where every image's [K_i][N_i] block starts (floats), its row length and the batch-wide index
// of its first sub-line.  Travels in the kernel arguments.
constexpr int K2S_MAX_IMAGES = 8;
struct K2sImage { int64_t off; int n_sub; int sub_base; };
struct K2sTable { K2sImage img[K2S_MAX_IMAGES]; };

// point at arclength `d` from sp along sp->ep in the reference's slope form, float64, no FMA
// contraction (models/line_process.py:43-57).
__device__ __forceinline__ void walk_along(const double sp[2], const double ep[2], double d, double& x, double& y) {
#pragma clang fp contract(off)
  const double vx = ep[0] - sp[0], vy = ep[1] - sp[1];
  double dx, dy;
  if (vx != 0.0) {
    const double m = vy / vx;
    dx = sqrt(d * d / (1.0 + m * m));
    dy = m * dx;
  } else {
    dx = 0.0;
    dy = vy > 0.0 ? d : -d;
  }
  x = dx + sp[0];
  y = dy + sp[1];
}

// ---------------------------------------------------------------------------------------------
// One 64-thread block per sub-line; thread t < T produces token t.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tokenize_kernel(
    const LinetrLineRec* __restrict__ recs, const int* __restrict__ sub2line_g, int N, double td, int T,
    int height, int width, double clip_x, double clip_y, const float* __restrict__ dense_score, float* __restrict__ sublines,
    float* __restrict__ pnt, float* __restrict__ mask, float* __restrict__ resp,
    float* __restrict__ angle_sub, float* __restrict__ score, float* __restrict__ cpnt,
    float* __restrict__ cscore, int n_pad_images, int64_t first_pad, float* __restrict__ mat_k2s, const K2sTable k2s) {
#pragma clang fp contract(off)
  const int n = blockIdx.x;
  if (n >= N) {
    const int pad_blocks = (n_pad_images + 63) / 64;
    if (n - N < pad_blocks) {
      // blocks past the sub-lines: one shared padding token per image for the compact token list -- coordinate (0, 0), score
      // dense_score[img][0][0] (the reference pads with zeros and gathers the score at the rounded coordinate, line_process.py:174-179)
      const int i = (n - N) * 64 + threadIdx.x;
      if (i < n_pad_images) {
        cpnt[(first_pad + i) * 2 + 0] = 0.f;
        cpnt[(first_pad + i) * 2 + 1] = 0.f;
        cscore[first_pad + i] = dense_score[(int64_t)i * height * width];
      }
      return;
    }
    // one more block per key-line (calls of up to K2S_MAX_IMAGES images): its row of the image's mat_klines2sublines [K_i][N_i],
    // 1 / num_sublines over the key-line's own sub-lines and 0 elsewhere (line_process.py:160-165; Python's 1/num is a float64
    // quotient stored as float32)
    const int k = n - N - pad_blocks;
    const LinetrLineRec r = recs[k];
    const K2sImage im = k2s.img[r.image & (K2S_MAX_IMAGES - 1)];
    float* row = mat_k2s + im.off + (int64_t)r.line_local * im.n_sub;
    const float w = (float)(1.0 / (double)r.n_sub);
    const int j0 = r.first_sub - im.sub_base;
    for (int j = threadIdx.x; j < im.n_sub; j += 64) row[j] = (j >= j0 && j < j0 + r.n_sub) ? w : 0.f;
    return;
  }
  const LinetrLineRec r = recs[sub2line_g[n]];
  const int j = n - r.first_sub;  // sub-line index inside its key-line
  // end-point clip of line_process.py:115-116: the limits come from the `image_shape` ARGUMENT of line_tokenizer (width - 0.6,
  // height - 0.6), which need not be the score map's shape (the dataset builder passes (640, 480) for a 480 x 640 image)
  const double epc[2] = {fmin(r.ep[0], clip_x), fmin(r.ep[1], clip_y)};
  for (int t = threadIdx.x; t < T; t += 64) {
    const int ti = j * T + t;
    double x = 0.0, y = 0.0;
    float mk = 0.f;
    if (ti < r.n_tok - 1) {
      walk_along(r.sp, r.ep, (double)ti * td, x, y);  // :110-113
      mk = 1.f;
    } else if (ti == r.n_tok - 1) {
      x = epc[0];                                      // :114-117
      y = epc[1];
      mk = 1.f;
    }
    const float fx = (float)x, fy = (float)y;          // :157 .float()
    const int64_t o = (int64_t)n * T + t;
    if (pnt) { pnt[o * 2 + 0] = fx; pnt[o * 2 + 1] = fy; }
    if (mask) mask[(int64_t)n * (T + 1) + 1 + t] = mk;
    // score gather :174-179 -- torch.round (half to even), clip to the map, index [y][x]
    int ix = (int)rintf(fx), iy = (int)rintf(fy);
    ix = ix < width - 1 ? ix : width - 1;
    iy = iy < height - 1 ? iy : height - 1;
    ix = ix < 0 ? 0 : ix;   // negative coordinates cannot occur after remove_borders; guard the load anyway
    iy = iy < 0 ? 0 : iy;
    const float sc = dense_score[(int64_t)r.image * height * width + (int64_t)iy * width + ix];
    if (score) score[o] = sc;
    if (cpnt && mk != 0.f) {  // compact list of real tokens (fused path)
      const int64_t c = (int64_t)r.first_tok + ti;
      cpnt[c * 2 + 0] = fx;
      cpnt[c * 2 + 1] = fy;
      cscore[c] = sc;
    }
  }
  if (threadIdx.x == 0) {
    if (mask) mask[(int64_t)n * (T + 1)] = 1.f;  // CLS slot :135
    double s[2], e[2];
    if (j == 0) { s[0] = r.sp[0]; s[1] = r.sp[1]; }
    else walk_along(r.sp, r.ep, (double)(j * T - 1) * td, s[0], s[1]);        // :125-128
    if (j == r.n_sub - 1) { e[0] = epc[0]; e[1] = epc[1]; }
    else walk_along(r.sp, r.ep, (double)((j + 1) * T - 1) * td, e[0], e[1]);
    sublines[(int64_t)n * 4 + 0] = (float)s[0];
    sublines[(int64_t)n * 4 + 1] = (float)s[1];
    sublines[(int64_t)n * 4 + 2] = (float)e[0];
    sublines[(int64_t)n * 4 + 3] = (float)e[1];
    const double dx = e[0] - s[0], dy = e[1] - s[1];
    const double geo = sqrt(dx * dx + dy * dy);                               // :148-149
    resp[n] = (float)(geo / (td * (double)T));
    angle_sub[(int64_t)n * 2 + 0] = (float)r.angle[0];                        // :151
    angle_sub[(int64_t)n * 2 + 1] = (float)r.angle[1];
  }
}

// ---------------------------------------------------------------------------------------------
// sample_descriptors (models/line_process.py:86-98): bilinear grid_sample (zero padding) of the NHWC
// descriptor map + L2 normalisation.  One wave64 per token, lane = 4 channels (dwordx4, coalesced
// 1 KiB per tap).  fp32 arithmetic in the order PyTorch's CPU grid_sampler uses.
// ---------------------------------------------------------------------------------------------
// --- the same sampling split into "issue the 4 tap loads" and "finish", so a caller can keep the next
// token's loads in flight while it works on the current one -----------------------------------------------
struct TapSet { f32x4 v[4]; float w[4]; };   // nw, ne, sw, se values and weights

// grid_sample's un-normalisation in the operation order of PyTorch's CPU kernel (the reference runs it on the CPU; its vectorised
// grid sampler computes (g + 1) * (size / 2) - 0.5 with ONE rounding -- the compiler contracts the multiply-add of its vector
// operators -- and (g + 1) * ((size - 1) / 2) with align_corners).  r06: established by matching torch 2.10's CPU output bit for bit
// on 20 000 points per map size (60 x 80 and 120 x 160 cells: 100 % identical samples with this form; the former
// ((g + 1) * size - 1) / 2 left 1-3 % of the samples up to 2e-6 off, 4e-6 after normalisation on 960 x 1280 images).
__device__ __forceinline__ float grid_unnormalize(float g, int size, int align_corners) {
#pragma clang fp contract(off)
  if (align_corners) return (g + 1.f) * ((float)(size - 1) / 2.f);
  return __builtin_fmaf(g + 1.f, (float)size / 2.f, -0.5f);
}

// Bilinear tap geometry of one point (sample_descriptors' grid arithmetic, models/line_process.py:86-98, and
// grid_sample's unnormalisation): 4 cell offsets (clamped, in cells) and weights (0 for taps outside the map: zero padding).
__device__ __forceinline__ void tap_coords(float px, float py, int Hc, int Wc, int align_corners, int (&off)[4],
                                           float (&wt)[4]) {
#pragma clang fp contract(off)
  const float s = 8.f;
  float gx = ((px - s / 2) + 0.5f) / ((float)Wc * s - s / 2 - 0.5f);
  float gy = ((py - s / 2) + 0.5f) / ((float)Hc * s - s / 2 - 0.5f);
  gx = gx * 2.f - 1.f;
  gy = gy * 2.f - 1.f;
  const float ix = grid_unnormalize(gx, Wc, align_corners), iy = grid_unnormalize(gy, Hc, align_corners);
  const float x_w = floorf(ix), y_n = floorf(iy);
  const float w = ix - x_w, e = 1.f - w, nn = iy - y_n, ss = 1.f - nn;
  const float wv[4] = {ss * e, ss * w, nn * e, nn * w};
  const int x0 = (int)x_w, y0 = (int)y_n;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
    const bool in = xx >= 0 && xx < Wc && yy >= 0 && yy < Hc;
    const int yc = min(max(yy, 0), Hc - 1), xc = min(max(xx, 0), Wc - 1);
    off[k] = yc * Wc + xc;
    wt[k] = in ? wv[k] : 0.f;
  }
}

// loads of the 4 taps (this lane's 4 channels); the address is always valid, out-of-map taps carry weight 0
__device__ __forceinline__ void taps_load(const float* __restrict__ nhwc_img, const int (&off)[4], const float (&wt)[4],
                                          int lane, TapSet& t) {
  const float* base = nhwc_img + lane * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    t.v[k] = *reinterpret_cast<const f32x4*>(base + (int64_t)off[k] * D);
    t.w[k] = wt[k];
  }
}

__device__ __forceinline__ void taps_issue(float px, float py, const float* __restrict__ nhwc_img, int Hc, int Wc,
                                           int align_corners, int lane, TapSet& t) {
  int off[4];
  float wt[4];
  tap_coords(px, py, Hc, Wc, align_corners, off, wt);
  taps_load(nhwc_img, off, wt, lane, t);
}

// RCP = true: one division + 4 multiplies instead of 4 divisions (<= 1 ulp apart); only for consumers that never
// expose the sampled descriptor itself (the pooling kernel), the reference's desc_sublines keeps the exact form.
template <bool RCP = false>
__device__ __forceinline__ f32x4 taps_finish(const TapSet& t) {
#pragma clang fp contract(off)
  f32x4 o;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float v = t.v[0][c] * t.w[0];                  // nw * w_nw, then one fused multiply-add per further tap: the CPU kernel's order
    v = __builtin_fmaf(t.v[1][c], t.w[1], v);
    v = __builtin_fmaf(t.v[2][c], t.w[2], v);
    v = __builtin_fmaf(t.v[3][c], t.w[3], v);
    o[c] = v;
    sq += v * v;
  }
  sq = wave_sum(sq);
  const float nrm = fmaxf(sqrtf(sq), 1e-12f);
  if constexpr (RCP) {
    const float inv = 1.f / nrm;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = o[c] * inv;
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = o[c] / nrm;
  }
  return o;
}

// bilinear sample (zero padding) + L2 normalisation of one token; every lane of the wave returns its 4 channels
__device__ __forceinline__ f32x4 sample_one(float px, float py, const float* __restrict__ nhwc_img, int Hc, int Wc,
                                            int align_corners, int lane) {
#pragma clang fp contract(off)
  const float s = 8.f;
  // keypoints - s/2 + 0.5 ; /= (w*s - s/2 - 0.5) ; *2 - 1          (:88-92)
  float gx = ((px - s / 2) + 0.5f) / ((float)Wc * s - s / 2 - 0.5f);
  float gy = ((py - s / 2) + 0.5f) / ((float)Hc * s - s / 2 - 0.5f);
  gx = gx * 2.f - 1.f;
  gy = gy * 2.f - 1.f;
  const float ix = grid_unnormalize(gx, Wc, align_corners), iy = grid_unnormalize(gy, Hc, align_corners);
  const float x_w = floorf(ix), y_n = floorf(iy);
  const float w = ix - x_w, e = 1.f - w, nn = iy - y_n, ss = 1.f - nn;
  const float nw = ss * e, ne = ss * w, sw = nn * e, se = nn * w;
  const int x0 = (int)x_w, y0 = (int)y_n, x1 = x0 + 1, y1 = y0 + 1;
  const float* base = nhwc_img + lane * 4;
  auto tap = [&](int yy, int xx) -> f32x4 {
    if (xx < 0 || xx >= Wc || yy < 0 || yy >= Hc) return f32x4{0.f, 0.f, 0.f, 0.f};
    return *reinterpret_cast<const f32x4*>(base + ((int64_t)yy * Wc + xx) * D);
  };
  const f32x4 v_nw = tap(y0, x0), v_ne = tap(y0, x1), v_sw = tap(y1, x0), v_se = tap(y1, x1);
  f32x4 o;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float v = v_nw[c] * nw;                        // (nw_val * nw) + (ne_val * ne) + ... as the CPU kernel evaluates it: fused
    v = __builtin_fmaf(v_ne[c], ne, v);
    v = __builtin_fmaf(v_sw[c], sw, v);
    v = __builtin_fmaf(v_se[c], se, v);
    o[c] = v;
    sq += v * v;
  }
  sq = wave_sum(sq);
  const float nrm = fmaxf(sqrtf(sq), 1e-12f);  // F.normalize eps
#pragma unroll
  for (int c = 0; c < 4; ++c) o[c] = o[c] / nrm;
  return o;
}

__global__ __launch_bounds__(256) void sample_desc_kernel(
    const float* __restrict__ pnt, const int* __restrict__ sub2line_g, const LinetrLineRec* __restrict__ recs,
    int64_t n_tokens, int T, const float* __restrict__ nhwc, int Hc, int Wc, int align_corners,
    float* __restrict__ desc) {
  const int64_t tok = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= n_tokens) return;
  const int lane = threadIdx.x & 63;
  const int n = (int)(tok / T);
  const int img = recs ? recs[sub2line_g[n]].image : 0;   // no records: all points belong to one image (linetr_sample_descriptors)
  const f32x4 o = sample_one(pnt[tok * 2 + 0], pnt[tok * 2 + 1], nhwc + (int64_t)img * Hc * Wc * D, Hc, Wc,
                             align_corners, lane);
  *reinterpret_cast<f32x4*>(desc + tok * D + lane * 4) = o;
}

}  // namespace lt
