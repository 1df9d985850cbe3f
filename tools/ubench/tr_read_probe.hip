// What does ds_read_b64_tr_b16 deliver?  LDS is filled with the identity pattern (element i of the bf16 array holds the
// 16-bit value i); every lane issues one transposing read at a per-lane address and prints the four 16-bit values it got.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/tr_read_probe.hip -o tools/ubench/tr_read_probe && tools/ubench/tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(unsigned short* out, int row_stride_elems) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  // the attention-V recipe: within a 16-lane group lane li addresses row (li >> 2), 4-element column chunk (li & 3);
  // lane groups 0/1 take columns 0-15 / 16-31, the upper half-wave starts 4 rows further down
  const int li = l & 15;
  const int row = (li >> 2) + 4 * (l >> 5);
  const int col = ((l >> 4) & 1) * 16 + (li & 3) * 4;
  const unsigned addr = (unsigned)(size_t)(&lds[row * row_stride_elems + col]);   // low 32 bits of a generic LDS pointer = LDS offset
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  out[l * 4 + 0] = (unsigned short)(v[0] & 0xffff);
  out[l * 4 + 1] = (unsigned short)(v[0] >> 16);
  out[l * 4 + 2] = (unsigned short)(v[1] & 0xffff);
  out[l * 4 + 3] = (unsigned short)(v[1] >> 16);
}

int main() {
  unsigned short* d;
  if (hipMalloc(&d, 64 * 4 * 2) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  for (int stride : {64, 200}) {
    probe<<<1, 64>>>(d, stride);
    hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
    if (e1 != hipSuccess || e2 != hipSuccess) { printf("launch: %s / %s\n", hipGetErrorString(e1), hipGetErrorString(e2)); return 1; }
    unsigned short h[256];
    hipError_t e3 = hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    if (e3 != hipSuccess) { printf("memcpy: %s\n", hipGetErrorString(e3)); return 1; }
    printf("row stride %d elements: lane -> (row,col) of the 4 values received\n", stride);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l * 4 + j] / stride, h[l * 4 + j] % stride);
      printf("\n");
    }
  }
  return 0;
}
