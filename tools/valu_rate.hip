// Microbenchmark: issue cost of v_add_f32 / v_fma_f32 vs the packed v_pk_add_f32 / v_pk_fma_f32 (2 flops per lane per op) per
// wave64 instruction, with 16 independent chains per wavefront, at 1 and 2 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters, float seed) {
  float a[16]; f32x2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] * .5f}; }
  const float c = seed * 1.0001f; const f32x2 c2 = {c, c + 1.f};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (MODE == 0) a[i] = a[i] + c;
        else if (MODE == 1) a[i] = __builtin_fmaf(a[i], c, c);
        else if (MODE == 2) p[i] = p[i] + c2;
        else p[i] = __builtin_elementwise_fma(p[i], c2, c2);
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(int threads, const char* name) {
  float* out; long long* cyc; (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 8);
  const int iters = 4000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
  (void)hipDeviceSynchronize();
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-14s waves/SIMD=%d : %.2f cycles per instruction per wave (SIMD: %.2f)\n", name, threads / 256,
         (double)c / (iters * 64.0), (double)c / (iters * 64.0) / (threads / 256));
  (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
  run<0>(256, "v_add_f32"); run<1>(256, "v_fma_f32"); run<2>(256, "v_pk_add_f32"); run<3>(256, "v_pk_fma_f32");
  run<0>(512, "v_add_f32"); run<1>(512, "v_fma_f32"); run<2>(512, "v_pk_add_f32"); run<3>(512, "v_pk_fma_f32");
  return 0;
}
