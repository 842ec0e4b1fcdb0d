// Which SIMD does wavefront w of an 8-wave (and 12-/16-wave) workgroup land on?  Reads HW_REG_HW_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; hipMalloc(&d, 1024 * 16 * 4);
  for (int threads : {512, 768, 1024}) {
    hipMemset(d, 0, 1024 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(4), dim3(threads), 0, 0, d);
    hipDeviceSynchronize();
    unsigned h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; ++b) {
      printf("threads=%d block %d: ", threads, b);
      for (int w = 0; w < threads / 64; ++w) printf("w%d:simd%u/wave%u/cu%u ", w, (h[b * 16 + w] >> 4) & 3, h[b * 16 + w] & 15, (h[b * 16 + w] >> 8) & 15);
      printf("\n");
    }
  }
  return 0;
}
