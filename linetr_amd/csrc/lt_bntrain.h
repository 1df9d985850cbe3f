// BatchNorm1d in TRAINING mode (batch statistics) for the training-time forward, SURVEY.md section 8(f) row 4:
// the reference's MLP stacks are Conv1d(k=1) + BatchNorm1d + ReLU (models/line_transformer.py:9-20) and train.py:127 puts the
// model in train mode, so every BatchNorm normalises with the mean / biased variance of the CURRENT batch -- over all B * N (* T)
// positions -- and moves its running statistics towards them (momentum 0.1, unbiased variance).  The eval path folds the running
// statistics into the convolution at linetr_create; here the convolution stays unfolded (LinetrModelConfig::bn_batch_stats), its
// pre-activations z [rows][C] are written by the GEMM, and three small kernels do the rest:
//   bn_partial_kernel    per-channel sum / sum of squares of a row chunk, float64 accumulators (torch's CPU kernel accumulates a float
//                        batch in double as well), one partial row per block -- no atomics, so the result is deterministic
//   bn_finalize_kernel   partials combined in block order -> mean, biased var, alpha = gamma / sqrt(var + eps), beta' = beta - mean alpha
//                        (the affine form torch's batch_norm_cpu_transform_input applies), running statistics updated in place
//   bn_apply_relu_kernel z <- max(z alpha + beta', 0)
// HBM-bound elementwise work (one read for the statistics, one read + write for the transform); coalesced along the channel axis.
#pragma once
#include "lt_common.h"

namespace lt {

constexpr int BN_MAX_BLOCKS = 512;     // row chunks of bn_partial_kernel (two per CU)
// (BN_MAX_CHANNELS, lt_handle.h: the widest BatchNorm layer the statistics scratch is sized for)

// what a training-time forward hands down to forward_core (linetr_forward_train)
struct BnTrain {
  float* running = nullptr;     // packed [layer][mean[C] | var[C]], updated in place
  float* batch = nullptr;       // packed like `running`: this batch's mean | BIASED variance (may be null)
  float momentum = 0.1f;
  double* partial = nullptr;    // [BN_MAX_BLOCKS][2 * 512] scratch
  float* affine = nullptr;      // [2 * 512] scratch: alpha | beta'
};

// z [rows][C] (row stride ld), C in {32, 64, 128, 256, 512}; partial [gridDim.x][2 C]
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* __restrict__ z, int64_t rows, int C, int ld,
                                                         double* __restrict__ partial) {
  __shared__ double red[2][256];
  const int tid = threadIdx.x;
  const int cw = C < 256 ? C : 256;            // channels side by side in a block pass
  const int rp = 256 / cw;                     // rows in parallel
  const int r_in = tid / cw, c_in = tid % cw;
  const int64_t chunk = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * chunk, r1 = r0 + chunk < rows ? r0 + chunk : rows;
  for (int c = c_in; c < C; c += 256) {
    double s = 0.0, q = 0.0;
    for (int64_t r = r0 + r_in; r < r1; r += rp) {
      const double v = (double)z[r * ld + c];
      s += v;
      q += v * v;
    }
    red[0][tid] = s;
    red[1][tid] = q;
    __syncthreads();
    if (r_in == 0) {
      for (int k = 1; k < rp; ++k) { s += red[0][k * cw + c_in]; q += red[1][k * cw + c_in]; }   // fixed order
      partial[(int64_t)blockIdx.x * 2 * C + c] = s;
      partial[(int64_t)blockIdx.x * 2 * C + C + c] = q;
    }
    __syncthreads();
  }
}

// one thread per channel
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ partial, int n_blocks, int C, int64_t rows,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float momentum, float* __restrict__ running /*mean[C] | var[C]*/,
                                                          float* __restrict__ batch /*mean[C] | biased var[C], or null*/,
                                                          float* __restrict__ affine /*alpha[C] | beta'[C]*/) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int b = 0; b < n_blocks; ++b) { s += partial[(int64_t)b * 2 * C + c]; q += partial[(int64_t)b * 2 * C + C + c]; }
  const double n = (double)rows;
  const double mean = s / n;
  double var = q / n - mean * mean;            // biased (what the normalisation uses)
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  const double alpha = (double)gamma[c] * invstd;
  affine[c] = (float)alpha;
  affine[C + c] = (float)((double)beta[c] - mean * alpha);
  if (batch) { batch[c] = (float)mean; batch[C + c] = (float)var; }
  // running statistics: (1 - momentum) old + momentum new, the variance unbiased (torch.nn.BatchNorm1d)
  const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
  running[c] = (float)((1.0 - (double)momentum) * (double)running[c] + (double)momentum * mean);
  running[C + c] = (float)((1.0 - (double)momentum) * (double)running[C + c] + (double)momentum * unbiased);
}

// in place, four channels per thread (C % 4 == 0, ld % 4 == 0)
__global__ __launch_bounds__(256) void bn_apply_relu_kernel(float* __restrict__ z, int64_t rows, int C, int ld,
                                                            const float* __restrict__ affine) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c4 = C / 4;
  if (i >= rows * c4) return;
  const int64_t r = i / c4;
  const int c = (int)(i - r * c4) * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(z + r * ld + c);
  const f32x4 a = *reinterpret_cast<const f32x4*>(affine + c), b = *reinterpret_cast<const f32x4*>(affine + C + c);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma clang fp contract(off)
    v[k] = fmaxf(v[k] * a[k] + b[k], 0.f);
  }
  *reinterpret_cast<f32x4*>(z + r * ld + c) = v;
}

// BatchNorm(train) + ReLU of layer `layer` on z [rows][C]; `off` = float offset of the layer inside the packed statistics
inline int bn_train_layer(hipStream_t st, const BnTrain& bt, float* z, int64_t rows, int C, int ld, const float* gamma,
                          const float* beta, int64_t off) {
  if (rows <= 0) return 0;
  if (C > BN_MAX_CHANNELS || C % 4) return fail(LINETR_E_ARG, "BatchNorm(train): %d channels (the statistics scratch holds %d, multiples of 4)", C, BN_MAX_CHANNELS);
  const int nb = (int)std::min<int64_t>(BN_MAX_BLOCKS, std::max<int64_t>(1, rows / 64));
  hipLaunchKernelGGL(bn_partial_kernel, dim3(nb), dim3(256), 0, st, (const float*)z, rows, C, ld, bt.partial);
  LT_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, (const double*)bt.partial, nb, C, rows, gamma, beta,
                     1e-5f, bt.momentum, bt.running + off, bt.batch ? bt.batch + off : nullptr, bt.affine);
  LT_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_apply_relu_kernel, dim3((unsigned)((rows * (C / 4) + 255) / 256)), dim3(256), 0, st, z, rows, C, ld,
                     (const float*)bt.affine);
  LT_LAUNCH_CHECK();
  return 0;
}

}  // namespace lt
