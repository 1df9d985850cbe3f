// Micro-benchmark: per-CU fetch rate of the GEMM's A-tile access pattern.
//   pattern 0: row-major matrix [M][K], block reads a [256 rows x 32 floats] tile per step (8 lanes x 16 B per row)
//   pattern 1: tile-major: the same 32 KB per step, but contiguous
// 512 threads per block, 1 block per CU, data consumed by a dummy reduction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(512) void fetch_kernel(const float* __restrict__ A, int K, int steps, int row_tiles, float* out) {
  const int tid = threadIdx.x;
  const int rt = blockIdx.x % row_tiles;
  f32x4 acc = {0, 0, 0, 0};
  for (int s = 0; s < steps; ++s) {
    const int k0 = (s * 32) % K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* p;
      if (PATTERN == 0) {
        const int r = rt * 256 + (tid >> 3) + i * 64;
        p = A + (size_t)r * K + k0 + (tid & 7) * 4;
      } else {
        p = A + ((size_t)rt * (K / 32) + (k0 / 32)) * 8192 + (i * 512 + tid) * 4;
      }
      const f32x4 v = *reinterpret_cast<const f32x4*>(p);
      acc += v;
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

int main() {
  const int K = 4096;
  const int tiles_list[5] = {2, 8, 24, 48, 256};    // 8 MB (L2), 32 MB (L2 aggregate), 96 MB / 192 MB (Infinity Cache), 1 GB (HBM)
  for (int ti = 0; ti < 5; ++ti) {
    const int row_tiles = tiles_list[ti];
    const int big = ti;
    const size_t elems = (size_t)row_tiles * 256 * K;
    float *A, *out;
    hipMalloc(&A, elems * 4); hipMalloc(&out, 64);
    hipMemset(A, 0, elems * 4);
    for (int pat = 0; pat < 2; ++pat) {
      const int steps = 1024, blocks = 256;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (pat == 0) hipLaunchKernelGGL(fetch_kernel<0>, dim3(blocks), dim3(512), 0, 0, A, K, steps, row_tiles, out);
        else hipLaunchKernelGGL(fetch_kernel<1>, dim3(blocks), dim3(512), 0, 0, A, K, steps, row_tiles, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double bytes = (double)blocks * steps * 32768;
      printf("working set %4d MB pattern=%s : %.1f us, %.2f TB/s total, %.1f GB/s per CU (%.1f B/clk @2.1GHz)\n", row_tiles * 4,
             pat ? "tile-major" : "row-major ", ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.1);
    }
    hipFree(A); hipFree(out);
  }
  return 0;
}
