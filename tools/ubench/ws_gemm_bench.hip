// Ablation bench of the weight-stationary K = 128 -> 256 GEMM (csrc/lt_gemm_ws.h) at cfg3's token count (291 208 rows).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/ws_gemm_bench.hip -o tools/ubench/ws_gemm_bench
#include "../../linetr_amd/csrc/lt_common.h"
#include "../../linetr_amd/csrc/lt_gemm.h"
#include "../../linetr_amd/csrc/lt_gemm_split.h"
#include "../../linetr_amd/csrc/lt_gemm_ws.h"
namespace lt {
inline bool small_gemm_wins(const GemmArgs&, int) { return false; }
inline bool split16_wins(const GemmArgs&, int) { return false; }
}
using namespace lt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void fill_kernel(float* p, int64_t n, unsigned seed, float scale) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed;
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  p[i] = ((int)(x & 0xffffff) - 0x800000) * (scale / 0x800000);
}
template <int DBG>
static float run(const WsGemmArgs& a, int grid, int iters) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws_kernel<DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_ws_kernel<DBG>, dim3(grid), dim3(512), WS_LDS, 0, a);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_ws_kernel<DBG>, dim3(grid), dim3(512), WS_LDS, 0, a);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / iters;
}
int main() {
  const int M = 291208;
  float *A, *Wf, *b, *Y; unsigned char* Wst;
  CK(hipMalloc((void**)&A, (size_t)M * 128 * 4)); CK(hipMalloc((void**)&Wf, 256 * 128 * 4)); CK(hipMalloc((void**)&b, 256 * 4));
  CK(hipMalloc((void**)&Y, (size_t)M * 256 * 4)); CK(hipMalloc((void**)&Wst, st_bytes(256, 128)));
  hipLaunchKernelGGL(fill_kernel, dim3((M * 128 + 255) / 256), dim3(256), 0, 0, A, (int64_t)M * 128, 1u, 1.f);
  hipLaunchKernelGGL(fill_kernel, dim3(128), dim3(256), 0, 0, Wf, (int64_t)256 * 128, 2u, 0.09f);
  hipLaunchKernelGGL(fill_kernel, dim3(1), dim3(256), 0, 0, b, (int64_t)256, 3u, 0.1f);
  const int64_t thr = st_row_blocks(256) * (128 / 16) * 32;
  hipLaunchKernelGGL(to_st_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, 0, Wf, 128, 256, 128 / 16, Wst);
  CK(hipDeviceSynchronize());
  WsGemmArgs a; a.A = A; a.lda = 128; a.Wst = Wst; a.bias = b; a.Y = Y; a.ldy = 256; a.M = M; a.act = ACT_RELU;
  const int it = 20;
  for (int grid : {256, 512}) {
    const float t0 = run<0>(a, grid, it), t1 = run<1>(a, grid, it), t2 = run<2>(a, grid, it), t3 = run<3>(a, grid, it), t4 = run<4>(a, grid, it),
                t5 = run<5>(a, grid, it), t7 = run<7>(a, grid, it);
    printf("ws gemm 291208 x 256 x 128, grid %d: full %.1f us (%.0f TF-eq, %.2f TB/s) | no stores %.1f | no loads %.1f | no loads, no stores %.1f | no MFMAs %.1f | "
           "no MFMAs, no stores %.1f | skeleton %.1f\n", grid, t0, 2.0 * M * 256 * 128 / t0 * 1e-6, (double)M * 1536 / t0 * 1e-6, t1, t2, t3, t4, t5, t7);
  }
  printf("grid 512: no loads, no stores %.1f us | the same without the B-fragment reads (MFMAs + barriers + LDS stores only) %.1f | "
         "full without fragment reads %.1f\n", run<3>(a, 512, it), run<35>(a, 512, it), run<32>(a, 512, it));
  for (int r = 0; r < 1; ++r)
    printf("stores: plain %.1f us | write-through (sc1) %.1f | nontemporal %.1f   (grid 512)\n", run<0>(a, 512, it), run<8>(a, 512, it), run<16>(a, 512, it));
  return 0;
}
