// Practical MFMA ceiling on MI355X: register-resident v_mfma_f32_32x32x16_bf16 loop on every CU, with the shader
// clock (s_memtime) and the 100 MHz wall clock read around it, so the achieved rate can be split into
// "clock under load" x "pipe utilisation".  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>   // 0: bf16 32x32x16, 1: fp32 32x32x2
__global__ __launch_bounds__(512) void mfma_loop(const unsigned* __restrict__ seed, float* out, long long* clk, int iters) {
  const int tid = threadIdx.x;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  unsigned sa[4], sb[4], sc[4], sd[4];
  for (int i = 0; i < 4; ++i) { sa[i] = seed[(tid * 16 + i) & 4095]; sb[i] = seed[(tid * 16 + 4 + i) & 4095];
                                sc[i] = seed[(tid * 16 + 8 + i) & 4095]; sd[i] = seed[(tid * 16 + 12 + i) & 4095]; }
  bf16x8 a0 = __builtin_bit_cast(bf16x8, sa), a1 = __builtin_bit_cast(bf16x8, sb);
  bf16x8 b0 = __builtin_bit_cast(bf16x8, sc), b1 = __builtin_bit_cast(bf16x8, sd);
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
      }
    } else {
      const float fa0 = __builtin_bit_cast(float, sa[0]), fa1 = __builtin_bit_cast(float, sb[0]);
      const float fb0 = __builtin_bit_cast(float, sc[0]), fb1 = __builtin_bit_cast(float, sd[0]);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1, acc[3], 0, 0, 0);
      }
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + tid] = s;
  if (tid == 0) { clk[blockIdx.x * 2] = c1 - c0; clk[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
  std::vector<unsigned> h(4096);
  // bf16 pairs with modest exponents (values ~ +-[0.5, 2)) so nothing overflows and the data toggles like real data
  srand(1);
  for (auto& v : h) {
    auto one = []() { unsigned m = rand() & 0x7f, e = 126 + (rand() & 1), s = rand() & 1; return (s << 15) | (e << 7) | m; };
    v = one() | (one() << 16);
  }
  unsigned* d_seed; float* d_out; long long* d_clk;
  hipMalloc(&d_seed, 4096 * 4); hipMemcpy(d_seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipMalloc(&d_out, 4096 * 512 * 4); hipMalloc(&d_clk, 4096 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind) {
    for (int threads : {256, 512}) {
      for (int blocks : {256, 512}) {
        const int iters = kind == 0 ? 20000 : 2500;
        for (int rep = 0; rep < 3; ++rep) {
          hipEventRecord(e0);
          if (kind == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(threads), 0, 0, d_seed, d_out, d_clk, iters);
          else hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(threads), 0, 0, d_seed, d_out, d_clk, iters);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          long long c[2]; hipMemcpy(c, d_clk, 16, hipMemcpyDeviceToHost);
          const double flop_per = kind == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
          const double fl = (double)blocks * (threads / 64) * iters * 16 * flop_per;
          const double mhz = (double)c[0] / ((double)c[1] / 100.0);   // shader clocks per us of the 100 MHz wall clock
          if (rep == 2)
            printf("%s blocks=%d waves/block=%d: %.3f ms  %.1f TF  shader clock %.0f MHz  cycles/MFMA/SIMD %.1f\n",
                   kind == 0 ? "bf16 32x32x16" : "fp32 32x32x2 ", blocks, threads / 64, ms, fl / ms / 1e9, mhz,
                   (double)c[0] / ((double)iters * 16 * ((threads / 64 + 3) / 4) * (blocks > 256 ? 1 : 1)));
        }
      }
    }
  }
  return 0;
}
