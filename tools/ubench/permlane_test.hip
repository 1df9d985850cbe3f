#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
  float x = (float)threadIdx.x;
  unsigned lo = __builtin_bit_cast(unsigned, x), hi = lo;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
  unsigned r[2] = {lo, hi};
  out[threadIdx.x] = __builtin_bit_cast(float, r[0]);
  out[64 + threadIdx.x] = __builtin_bit_cast(float, r[1]);
}
int main() {
  float* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[128]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("r0: lane0=%g lane1=%g lane31=%g lane32=%g lane33=%g lane63=%g\n", h[0], h[1], h[31], h[32], h[33], h[63]);
  printf("r1: lane0=%g lane1=%g lane31=%g lane32=%g lane33=%g lane63=%g\n", h[64], h[65], h[95], h[96], h[97], h[127]);
  return 0;
}
