// Do ds_read_b128 fragment reads overlap with MFMAs on a CDNA4 CU?  8 waves per CU (2 per SIMD), each iteration =
// the GEMM's per-K-step work: NR fragment reads (row stride 208 B, conflict-free) + NM v_mfma_f32_32x32x16_bf16.
//   mode 0: MFMAs only      mode 1: reads only      mode 2: interleaved M R M R ...     mode 3: R x NR then M x NM
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_mfma_overlap lds_mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NR, int NM, bool BARRIER>
__global__ __launch_bounds__(512) void k(float* out, long long* clk, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 384 * 208 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u;
  __syncthreads();
  const unsigned char* base = lds + ((wave & 3) * 64 + (lane & 31)) * 208 + (lane >> 5) * 16;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 f[12];
  for (int i = 0; i < 12; ++i) f[i] = *reinterpret_cast<const bf16x8*>(base + (i % 3) * 64 + (i / 3) * 32 * 208 % (320 * 208));
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (MODE == 3 && m == 0) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[r % 12]) : "v"((unsigned)(size_t)base), "n"((r % 3) * 64 + (r / 3) * 32));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if (MODE != 1) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[(m * 5) % 6], f[6 + (m * 7) % 6], acc[m & 3], 0, 0, 0);
      if ((MODE == 1 || MODE == 2) && m < NR) {
        // a fresh destination each time; the value is consumed NR slots later at the earliest
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[(m + 6) % 12]) : "v"((unsigned)(size_t)base), "n"((m % 3) * 64 + (m / 3) * 32));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 1 || MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (BARRIER) __syncthreads();
  }
  const long long c1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 12; ++i) s += (float)f[i][0];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) clk[blockIdx.x] = c1 - c0;
}

// Same work as the GEMM's K tile, but with compiler-visible LDS loads (so the compiler inserts the s_waitcnt the real
// kernel has): half A = 24 MFMAs on F0 while F1 is read, [barrier], half B = 24 MFMAs on F1 while F0 is read.
// SPAN: the 12 reads of a half are spread over the first SPAN MFMA slots.  DRAIN=false: no barrier (no lgkmcnt(0)).
template <int SPAN, bool BARRIER, int NFSETS>
__global__ __launch_bounds__(512) void kreal(float* out, long long* clk, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 384 * 208 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u;
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 F[2][12];
  auto rd = [&](int k, int buf, int s, bf16x8 (&f)[12]) {
    const unsigned char* Ab = lds + (buf * 384 + wm * 64 + (lane & 31)) * 208 + (lane >> 5) * 16 + s * 32;
    const unsigned char* Bb = lds + (buf * 384 + 256 + wn * 64 + (lane & 31)) * 208 + (lane >> 5) * 16 + s * 32;
    if (k < 6) f[k] = *reinterpret_cast<const bf16x8*>(Ab + (k / 3) * 32 * 208 + (k % 3) * 64);
    else f[k] = *reinterpret_cast<const bf16x8*>(Bb + ((k - 6) / 3) * 32 * 208 + (k % 3) * 64);
  };
  constexpr int TPA[6] = {2, 1, 0, 1, 0, 0}, TPB[6] = {0, 1, 2, 0, 1, 0};
  auto mm = [&](int m, const bf16x8 (&f)[12]) {
    const int t = m / 4, i = (m % 4) / 2, j = m % 2;
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[i * 3 + TPA[t]], f[6 + j * 3 + TPB[t]], acc[i][j], 0, 0, 0);
  };
#pragma unroll
  for (int k = 0; k < 12; ++k) rd(k, 0, 0, F[0]);
  const long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      mm(m, F[0]);
#pragma unroll
      for (int k = 0; k < 12; ++k) if (k * SPAN / 12 == m) rd(k, buf, 1, F[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BARRIER) __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      mm(m, F[1]);
#pragma unroll
      for (int k = 0; k < 12; ++k) if (k * SPAN / 12 == m) rd(k, buf ^ 1, 0, F[0]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long c1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) clk[blockIdx.x] = c1 - c0;
}

template <int SPAN, bool BARRIER>
void run_real(const char* name, float* out, long long* clk) {
  const int iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kreal<SPAN, BARRIER, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 384 * 208);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kreal<SPAN, BARRIER, 2>), dim3(256), dim3(512), 2 * 384 * 208, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-44s span=%2d barrier=%d : %7.1f ns per K tile (48 MFMAs + 24 reads per wave; MFMA pipe alone ~1440 ns)\n", name, SPAN, (int)BARRIER, ms * 1e6 / iters);
}

template <int MODE, int NR, int NM, bool BARRIER>
void run(const char* name, float* out, long long* clk) {
  const int iters = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NR, NM, BARRIER>), hipFuncAttributeMaxDynamicSharedMemorySize, 384 * 208);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NR, NM, BARRIER>), dim3(256), dim3(512), 384 * 208, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("%-34s NR=%2d NM=%2d barrier=%d : %7.1f s_memtime ticks/iter, %7.1f ns/iter  (MFMA pipe alone: %d clk, LDS at 128 B/clk: %d clk)\n",
         name, NR, NM, (int)BARRIER, (double)c / iters, ms * 1e6 / iters, NM * 32 * 2, NR * 8 * 8);
}

int main() {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
  run<0, 12, 24, false>("MFMA only", out, clk);
  run<1, 12, 24, false>("ds_read_b128 only", out, clk);
  run<2, 12, 24, false>("interleaved M R", out, clk);
  run<3, 12, 24, false>("R x12 then M x24", out, clk);
  run<2, 12, 24, true>("interleaved M R + barrier", out, clk);
  run<3, 12, 24, true>("R x12 then M x24 + barrier", out, clk);
  run<0, 8, 12, false>("MFMA only", out, clk);
  run<1, 8, 12, false>("ds_read_b128 only", out, clk);
  run<2, 8, 12, false>("interleaved M R", out, clk);
  run<3, 8, 12, true>("R x8 then M x12 + barrier", out, clk);
  run_real<24, true>("compiler-visible reads (real waits)", out, clk);
  run_real<24, false>("compiler-visible reads (real waits)", out, clk);
  run_real<12, true>("compiler-visible reads, front-loaded", out, clk);
  run_real<12, false>("compiler-visible reads, front-loaded", out, clk);
  return 0;
}
