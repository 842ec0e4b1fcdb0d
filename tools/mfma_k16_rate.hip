// What the K = 16 bf16 MFMA (v_mfma_f32_16x16x16_bf16, the shape of a Winograd-domain contraction over 16 channels) sustains per
// SIMD next to the K = 32 form and the fp32 16x16x4, and whether VALU work of the SAME wavefront fits between them
// (tools/mfma_k16_rate.hip):   hipcc --offload-arch=gfx950 -O3 -o /tmp/k16 tools/mfma_k16_rate.hip && /tmp/k16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE 0: 16x16x16 bf16   1: 16x16x32 bf16   2: 16x16x4 f32.   NV: independent v_fma per MFMA issued by the same wavefront
template <int MODE, int NA, int NV>
__global__ void k(float* out, int iters) {
  f32x4 acc[NA];
  for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
  const float y = 1.0001f;
  unsigned ua[4] = {threadIdx.x * 0x3f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  const bf16x8 a8 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ua));
  const s16x4 a4 = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(ua));
  const float af = threadIdx.x * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, a4, acc[i], 0, 0, 0);
        if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, a8, acc[i], 0, 0, 0);
        if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, af, acc[i], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[(v + i) & 7]) : "v"(y));
      }
  }
  float s = 0.f;
  for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE, int NA, int NV>
void run(int waves_per_simd) {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 3000, threads = 256 * waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NA, NV>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double n = (double)iters * 6 * NA * waves_per_simd;
  const char* nm[3] = {"16x16x16 bf16", "16x16x32 bf16", "16x16x4  f32 "};
  printf("%s  %d wave(s)/SIMD  %2d chains  %d v_fma per MFMA: %.1f ns per MFMA per SIMD = %.1f cycles @2.4 GHz\n", nm[MODE], waves_per_simd, NA, NV,
         best * 1e6 / n, best * 1e6 / n * 2.4);
  hipFree(out);
}
int main() {
  run<0, 8, 0>(1); run<0, 8, 0>(2); run<1, 8, 0>(1); run<1, 8, 0>(2); run<2, 8, 0>(1); run<2, 8, 0>(2);
  run<0, 8, 1>(1); run<0, 8, 2>(1); run<0, 8, 4>(1); run<0, 8, 1>(2); run<0, 8, 2>(2); run<0, 8, 4>(2);
  run<1, 8, 2>(2); run<1, 8, 4>(2); run<1, 8, 8>(2);
  run<2, 8, 2>(2); run<2, 8, 4>(2); run<2, 8, 8>(2);
  return 0;
}
