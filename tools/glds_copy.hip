#include <hip/hip_runtime.h>
__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__global__ void k(const float* src, float* dst, int n) {
  extern __shared__ float sm[];
  for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4) glds16(src + i, sm + i);
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = sm[i];
}
int main() {
  int n = 160 * 164; float *s, *d; hipMalloc(&s, n * 4); hipMalloc(&d, n * 4);
  float* h = new float[n]; for (int i = 0; i < n; ++i) h[i] = i * .5f;
  hipMemcpy(s, h, n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, n * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), n * 4, 0, s, d, n);
  float* o = new float[n]; hipMemcpy(o, d, n * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < n; ++i) bad += o[i] != h[i];
  printf("bad %d of %d\n", bad, n); return bad != 0;
}
