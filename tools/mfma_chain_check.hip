// Reproducer attempt for the finding of round 6 (mono_fwd_x_k<split>): v_mfma_f32_16x16x32_bf16 and v_mfma_f32_16x16x16_bf16 chained
// on ONE accumulator against the same products on two accumulators added at the end (tools/mfma_chain_check.hip):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mcc tools/mfma_chain_check.hip && /tmp/mcc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(const u32x4* A32, const u32x2* A16, const u32x4* B32, const u32x2* B16, float* out, int reps) {
  const int lane = threadIdx.x;
  f32x4 c = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const u32x4 a = A32[(r * 6 + t) * 64 + lane], b = B32[(r * 6 + t) * 64 + lane];
      const u32x2 a2 = A16[(r * 6 + t) * 64 + lane], b2 = B16[(r * 6 + t) * 64 + lane];
      if (MODE == 0) {          // chained on one accumulator
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a2), __builtin_bit_cast(s16x4, b2), c, 0, 0, 0);
      } else if (MODE == 1) {   // two accumulators
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a2), __builtin_bit_cast(s16x4, b2), c2, 0, 0, 0);
      } else {                  // the K = 16 part through the K = 32 instruction, upper half zero
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, u32x4{a2[0], a2[1], 0u, 0u}), __builtin_bit_cast(bf16x8, u32x4{b2[0], b2[1], 0u, 0u}), c, 0, 0, 0);
      }
    }
  }
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i] + c2[i];
}
int main() {
  const int reps = 8, n = reps * 6 * 64;
  unsigned *h32 = new unsigned[n * 4 * 2], *h16 = new unsigned[n * 2 * 2];
  srand(1);
  auto bf = []() -> unsigned { const float x = (rand() % 17 - 8) * 0.25f; unsigned u; std::memcpy(&u, &x, 4); return u >> 16; };
  for (int i = 0; i < n * 8; ++i) h32[i] = bf() | (bf() << 16);
  for (int i = 0; i < n * 4; ++i) h16[i] = bf() | (bf() << 16);
  unsigned *d32, *d16; float* out; hipMalloc(&d32, n * 32); hipMalloc(&d16, n * 16); hipMalloc(&out, 1024);
  hipMemcpy(d32, h32, n * 32, hipMemcpyHostToDevice); hipMemcpy(d16, h16, n * 16, hipMemcpyHostToDevice);
  float res[3][256];
  hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, (u32x4*)d32, (u32x2*)d16, (u32x4*)(d32 + n * 4), (u32x2*)(d16 + n * 2), out, reps);
  hipMemcpy(res[0], out, 1024, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, (u32x4*)d32, (u32x2*)d16, (u32x4*)(d32 + n * 4), (u32x2*)(d16 + n * 2), out, reps);
  hipMemcpy(res[1], out, 1024, hipMemcpyDeviceToHost);
  hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, (u32x4*)d32, (u32x2*)d16, (u32x4*)(d32 + n * 4), (u32x2*)(d16 + n * 2), out, reps);
  hipMemcpy(res[2], out, 1024, hipMemcpyDeviceToHost);
  double e01 = 0, e21 = 0, mx = 0;
  for (int i = 0; i < 256; ++i) { e01 = fmax(e01, fabs(res[0][i] - res[1][i])); e21 = fmax(e21, fabs(res[2][i] - res[1][i])); mx = fmax(mx, fabs(res[1][i])); }
  printf("exact small-integer operands (every product and sum exact in fp32); max |result| %g\n", mx);
  printf("chained K=32 -> K=16 on one accumulator  vs two accumulators: max |diff| %g\n", e01);
  printf("K=16 part through the K=32 instruction   vs two accumulators: max |diff| %g\n", e21);
  return 0;
}
