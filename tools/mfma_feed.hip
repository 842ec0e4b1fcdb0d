// Microbenchmark: v_mfma_f32_16x16x4_f32 whose B operand is produced by a VALU op right before it (Winograd input
// transform pattern) vs. operands produced in a batch ahead of the MFMAs.  One workgroup per CU; cycles per MFMA per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters, float seed) {
  f32x4 acc[16];
  float a = threadIdx.x * 1e-3f, t[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = seed * (i + 1) + i * i + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{(float)i, 0.f, seed, 0.f};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {            // constant operands
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, t[i], acc[i], 0, 0, 0);
    } else if (MODE == 1) {     // VALU result consumed by the very next MFMA
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float v = t[i] - t[(i + 3) & 15];
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] += 1.0f;
    } else if (MODE == 4 || MODE == 5) {   // Winograd round: 8 ds_read_b64 (+wait), ~NV VALU ops, then 16 MFMAs
      extern __shared__ float lds[];
      constexpr int NV = MODE == 4 ? 32 : 96;
      float2 raw[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) raw[i] = *reinterpret_cast<const float2*>(lds + ((threadIdx.x * 2 + 40 * i + it * 8) & 8190));
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = (i & 1 ? raw[i >> 1].y : raw[i >> 1].x) - t[(i + 3) & 15];
#pragma unroll
      for (int rep = 0; rep < (NV - 16) / 16; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = v[i] * 1.0001f + v[(i + 5) & 15];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[i], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE == 3) {     // 16 operands ahead, then 4 chains x 4 dependent MFMAs starting from 0 (Winograd da1 round)
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = t[i] - t[(i + 3) & 15];
      __builtin_amdgcn_sched_barrier(0);
      f32x4 m[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) m[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) m[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[4 * g + c], m[c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] += m[c];
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] += 1.0f;
    } else {                    // 16 operands computed ahead, then 16 MFMAs
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = t[i] - t[(i + 3) & 15];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[i], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) t[i] += 1.0f;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(int threads) {
  float* out; long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 32768, 0, out, cyc, iters, 1.0f);
  hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("mode=%d waves/SIMD=%d : %.1f cycles per MFMA per wave (pipe: %.1f)\n", MODE, threads / 256,
         (double)c / (iters * 16.0), (double)c / (iters * 16.0) / (threads / 256));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>(256); run<1>(256); run<2>(256); run<3>(256); run<4>(256); run<5>(256);
  run<0>(512); run<1>(512); run<2>(512); run<3>(512); run<4>(512); run<5>(512);
  return 0;
}
