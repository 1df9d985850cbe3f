// Ablation bench of the split-tile GEMM (experiments/csrc/lt_gemm_st.h): the same kernel with MFMAs / in-loop DMA / fragment reads /
// epilogue compiled out, timed with HIP events on the signature network's shapes.  Operands are ST images of random floats.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilinetr_amd/csrc -Iinclude experiments/ubench/st_gemm_bench.hip -o experiments/ubench/st_gemm_bench
#include "../../linetr_amd/csrc/lt_common.h"
#include "../csrc/lt_gemm_st.h"
namespace lt {
inline bool small_gemm_wins(const GemmArgs&, int) { return false; }
inline bool split16_wins(const GemmArgs&, int) { return false; }
}
using namespace lt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_kernel(float* p, int64_t n, unsigned seed) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed;
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  p[i] = ((int)(x & 0xffffff) - 0x800000) * (1.f / 0x800000);
}
static unsigned char* make_st(int rows, int K, unsigned seed) {
  float* f; unsigned char* st;
  if (hipMalloc((void**)&f, (size_t)rows * K * 4) != hipSuccess || hipMalloc((void**)&st, st_bytes(rows, K)) != hipSuccess) return nullptr;
  const int64_t n = (int64_t)rows * K;
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, f, n, seed);
  const int64_t thr = st_row_blocks(rows) * (K / 16) * 32;
  hipLaunchKernelGGL(to_st_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, 0, f, K, rows, K / 16, st);
  (void)hipDeviceSynchronize();
  (void)hipFree(f);
  return st;
}
template <int DBG>
static float run(const StGemmArgs& a, int iters) {
  constexpr int lds = 4 * STG_SLOT;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_st_kernel<DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const dim3 grid((a.N / STG_BN) * cdiv(a.M, STG_BM));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_st_kernel<DBG>, grid, dim3(512), lds, 0, a);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(gemm_st_kernel<DBG>, grid, dim3(512), lds, 0, a);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / iters;
}
int main() {
  struct Shape { int M, N, K1, K2; const char* name; };
  const Shape shapes[] = {{25472, 768, 256, 0, "qkv"}, {25472, 512, 256, 256, "W1"}, {25472, 256, 512, 0, "W2"},
                          {32768, 512, 512, 0, "2 rounds"}, {32768, 256, 4096, 0, "1 round long K"}, {8192, 4096, 4096, 0, "big"}};
  float* zeros; CK(hipMalloc((void**)&zeros, 4096 * 4)); CK(hipMemset(zeros, 0, 4096 * 4));
  for (const Shape& s : shapes) {
    StGemmArgs a;
    a.A1 = make_st(s.M, s.K1, 1); a.nk1 = s.K1 / 16;
    if (s.K2) { a.A2 = make_st(s.M, s.K2, 2); a.nk2 = s.K2 / 16; }
    a.W = make_st(s.N, s.K1 + s.K2, 3);
    unsigned char* y; CK(hipMalloc((void**)&y, st_bytes(s.M, s.N)));
    float* yf; CK(hipMalloc((void**)&yf, 4096));
    a.Yst = y; a.Y = yf; a.bias = zeros; a.M = s.M; a.N = s.N;
    if (!a.A1 || !a.W) { printf("alloc failed\n"); return 1; }
    const double fl = 2.0 * s.M * s.N * (s.K1 + s.K2);
    const int it = 20;
    const float t0 = run<0>(a, it), t1 = run<1>(a, it), t2 = run<2>(a, it), t4 = run<4>(a, it), t8 = run<8>(a, it),
                t3 = run<3>(a, it), t6 = run<6>(a, it), t5 = run<5>(a, it), t15 = run<15>(a, it);
    printf("%-16s %6d x %4d x %4d: full %7.1f us (%5.1f TF-eq) | no MFMA %7.1f | no DMA %7.1f | no reads %7.1f | no epilogue %7.1f | "
           "no MFMA+DMA %7.1f | no DMA+reads (MFMA only) %7.1f | no MFMA+reads (DMA only) %7.1f | skeleton %7.1f\n",
           s.name, s.M, s.N, s.K1 + s.K2, t0, fl / t0 / 1e6, t1, t2, t4, t8, t3, t6, t5, t15);
    fflush(stdout);
    (void)hipFree((void*)a.A1); (void)hipFree((void*)a.A2); (void)hipFree((void*)a.W); (void)hipFree(y); (void)hipFree(yf);
  }
  return 0;
}
