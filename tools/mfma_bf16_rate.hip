// Microbenchmark (round 6): what v_mfma_f32_16x16x32_bf16 sustains on one CU under the conditions of the split-bf16 GEMMs --
// NA independent accumulator chains per wavefront, W MFMA wavefronts per SIMD, optionally one more wavefront per SIMD doing
// plain VALU work (the loader role's split) -- timed over the whole launch with HIP events (all 256 CUs busy).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_rate.hip -o /tmp/mfma_bf16_rate && /tmp/mfma_bf16_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NA, int MW>
__global__ void k(float* out, int iters, int valu_per_iter, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const long long t0 = __builtin_readcyclecounter();
  if (wave >= 4 * MW) {                       // VALU role: 8 independent chains
    float x[8], y = 1.0001f;
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it)
      for (int v = 0; v < valu_per_iter; v += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(y));
      }
    float sx = 0.f;
    for (int i = 0; i < 8; ++i) sx += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sx;
    if (blockIdx.x == 0 && threadIdx.x == 256 * MW) cyc[1] = __builtin_readcyclecounter() - t0;
    return;
  }
  f32x4 acc[NA];
  u32x4 ua = {threadIdx.x * 0x3f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, ub = {0x3f803f80u, threadIdx.x, 0x3f803f80u, 0x3f803f80u};
  const bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
#pragma unroll
  for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int i = 0; i < NA; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NA; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = __builtin_readcyclecounter() - t0;
}
template <int NA, int MW>
void run(int valu_waves, int valu_per_iter) {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  long long* cyc; hipMalloc(&cyc, 16); hipMemset(cyc, 0, 16);
  const int iters = 4000, threads = 256 * MW + 256 * valu_waves;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NA, MW>), dim3(256), dim3(threads), 0, 0, out, iters, valu_per_iter, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long hc[2]; hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
  printf("   [wave 0 (MFMA role): %lld cycles = %.1f per MFMA of the SIMD; VALU wave: %lld cycles = %.1f per v_fma]\n", hc[0],
         (double)hc[0] / (iters * 6. * NA * MW), hc[1], valu_per_iter ? (double)hc[1] / ((double)iters * valu_per_iter) : 0.);
  const double mfma_per_simd = (double)iters * 6 * NA * MW;
  printf("NA=%2d MFMA waves/SIMD=%d VALU wave/SIMD=%d (%3d v_fma per %2d MFMAs of a wave): %.3f ms  %.1f ns per MFMA per SIMD = %.1f cycles @2.4GHz  -> %.0f TFLOP/s bf16\n",
         NA, MW, valu_waves, valu_per_iter, 6 * NA, ms, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4,
         mfma_per_simd * 1024 * 16384. / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  run<1, 1>(0, 0); run<2, 1>(0, 0); run<4, 1>(0, 0); run<10, 1>(0, 0);
  run<2, 2>(0, 0); run<10, 2>(0, 0);
  run<10, 2>(1, 0); run<10, 2>(1, 56); run<10, 2>(1, 120); run<10, 2>(1, 240); run<10, 1>(1, 120); run<10, 1>(1, 240);
  return 0;
}
