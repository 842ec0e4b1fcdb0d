// Microbenchmark: issue rate of v_mfma_f32_16x16x4_f32 as a function of the number of independent accumulator chains
// per wave (NA) and of waves per SIMD.  One workgroup per CU; prints cycles per MFMA per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NA>
__global__ void k(float* out, long long* cyc, int iters) {
  f32x4 acc[NA];
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NA>
void run(int threads) {
  float* out; long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipLaunchKernelGGL(k<NA>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(k<NA>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("NA=%d waves/SIMD=%d : %.1f cycles per MFMA per wave  (pipe: %.1f cycles per MFMA)\n", NA, threads / 256,
         (double)c / (iters * 8.0 * NA), (double)c / (iters * 8.0 * NA) / (threads / 256));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<1>(256); run<2>(256); run<3>(256); run<4>(256); run<9>(256);
  run<1>(512); run<2>(512); run<3>(512); run<4>(512); run<9>(512);
  return 0;
}
