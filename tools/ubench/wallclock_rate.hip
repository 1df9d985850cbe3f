// What does wall_clock64() (s_memrealtime) tick at?  Spin for N ticks, time with HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long ticks, long long* out) {
  const long long t0 = wall_clock64(), c0 = clock64();
  while (wall_clock64() - t0 < ticks) {}
  out[0] = wall_clock64() - t0; out[1] = clock64() - c0;
}
int main() {
  long long* d; hipMalloc(&d, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (long long ticks : {100000LL, 1000000LL}) {
    hipEventRecord(e0); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, 0, ticks, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%lld wall ticks (%lld s_memtime ticks) took %.3f ms by HIP events -> wall_clock64 = %.1f MHz, s_memtime = %.1f MHz\n", h[0], h[1], ms, h[0] / ms / 1e3, h[1] / ms / 1e3);
  }
  return 0;
}
