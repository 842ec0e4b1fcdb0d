#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, float* dst, int n, int off) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (4 * i + 3 + off < n) {
    f32x4 v = *reinterpret_cast<const f32x4u*>(src + off + 4 * i);
    *reinterpret_cast<f32x4u*>(dst + off + 4 * i) = v * 2.f;
  }
}
int main() {
  int n = 1 << 20; float *s, *d; hipMalloc(&s, n * 4); hipMalloc(&d, n * 4);
  float* h = new float[n]; for (int i = 0; i < n; ++i) h[i] = i;
  hipMemcpy(s, h, n * 4, hipMemcpyHostToDevice); hipMemset(d, 0, n * 4);
  for (int off = 1; off <= 3; ++off) {
    hipLaunchKernelGGL(k, dim3(n / 1024), dim3(256), 0, 0, s, d, n, off);
    float* o = new float[n]; hipMemcpy(o, d, n * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = off; i + 4 < n - 8; ++i) bad += o[i] != 2.f * h[i];
    printf("off %d bad %d\n", off, bad);
  }
  return 0;
}
