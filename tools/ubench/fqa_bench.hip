// Ablation bench of the fused q/k/v projection + attention kernel (csrc/lt_attn_fused.h) at cfg3's shape: 128 images x 199
// sub-lines, 4 heads.  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/fqa_bench.hip -o tools/ubench/fqa_bench
#include "../../linetr_amd/csrc/lt_common.h"
#define LT_FQA_STAMPS 1
#include "../../linetr_amd/csrc/lt_attn_fused.h"
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
static float run(const float* z, const unsigned char* W, const float* b, const int* cu, float* out, int n_img, int iters) {
  const dim3 grid(n_img, HEADS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sig_qkv_attn_kernel<DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, FQA_LDS);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(sig_qkv_attn_kernel<DBG>, grid, dim3(512), FQA_LDS, 0, z, W, b, cu, out);
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(sig_qkv_attn_kernel<DBG>, grid, dim3(512), FQA_LDS, 0, z, W, b, cu, out);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) printf("launch error\n");
  return ms * 1e3f / iters;
}
int main() {
  const int n_img = 128, per = 199, N = n_img * per;
  float *z, *Wf, *b, *out; unsigned char* Wst; int* cu;
  CK(hipMalloc((void**)&z, (size_t)N * D * 4)); CK(hipMalloc((void**)&Wf, (size_t)3 * D * D * 4)); CK(hipMalloc((void**)&b, 3 * D * 4));
  CK(hipMalloc((void**)&out, (size_t)N * D * 4)); CK(hipMalloc((void**)&Wst, st_bytes(3 * D, D))); CK(hipMalloc((void**)&cu, (n_img + 1) * 4));
  hipLaunchKernelGGL(fill_kernel, dim3((N * D + 255) / 256), dim3(256), 0, 0, z, (int64_t)N * D, 1u, 1.f);
  hipLaunchKernelGGL(fill_kernel, dim3((3 * D * D + 255) / 256), dim3(256), 0, 0, Wf, (int64_t)3 * D * D, 2u, 0.06f);
  hipLaunchKernelGGL(fill_kernel, dim3(3), dim3(256), 0, 0, b, (int64_t)3 * D, 3u, 0.1f);
  const int64_t thr = st_row_blocks(3 * D) * (D / 16) * 32;
  hipLaunchKernelGGL(to_st_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, 0, Wf, D, 3 * D, D / 16, Wst);
  std::vector<int> h(n_img + 1);
  for (int i = 0; i <= n_img; ++i) h[i] = i * per;
  CK(hipMemcpy(cu, h.data(), (n_img + 1) * 4, hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());
  const int it = 20;
  const float t0 = run<0>(z, Wst, b, cu, out, n_img, it), t1 = run<1>(z, Wst, b, cu, out, n_img, it), t2 = run<2>(z, Wst, b, cu, out, n_img, it),
              t3 = run<3>(z, Wst, b, cu, out, n_img, it), t4 = run<4>(z, Wst, b, cu, out, n_img, it), t10 = run<10>(z, Wst, b, cu, out, n_img, it),
              t11 = run<11>(z, Wst, b, cu, out, n_img, it), t6 = run<6>(z, Wst, b, cu, out, n_img, it);
  printf("fused q/k/v + attention, 128 images x 199 x 4 heads: full %.1f us | no projection MFMAs %.1f | no attention phase %.1f | neither %.1f | "
         "projection without barriers / DMA waits %.1f | projection only (no staging, no attention) %.1f | skeleton (no MFMAs, no staging, no attention) %.1f | "
         "projection without sync, no attention %.1f\n", t0, t1, t2, t3, t4, t10, t11, t6);
  // phase time stamps of wave 0 (wall_clock64 = 100 MHz): one launch with DBG = 16
  (void)run<16>(z, Wst, b, cu, out, n_img, 1);
  static unsigned long long st[512][8];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(lt::fqa_stamps), sizeof(st)));
  double ph[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long t_min = ~0ull, t_max = 0;
  for (int bk = 0; bk < 512; ++bk) {
    for (int k = 0; k < 6; ++k) ph[k] += (double)(st[bk][k + 1] - st[bk][k]) / 100.0;      // us
    t_min = st[bk][0] < t_min ? st[bk][0] : t_min;
    t_max = st[bk][6] > t_max ? st[bk][6] : t_max;
  }
  printf("mean per block (us): prologue %.2f | projection %.2f | q/k/v conversion %.2f | stage + half 0 %.2f | stage + half 1 %.2f | epilogue %.2f | "
         "first start -> last end %.1f us\n", ph[0] / 512, ph[1] / 512, ph[2] / 512, ph[3] / 512, ph[4] / 512, ph[5] / 512, (double)(t_max - t_min) / 100.0);
  return 0;
}
