// Microbenchmark: ALU cost of one extra wave-instruction of a given kind executed NEXT TO f32 MFMAs (phase-separated:
// a batch of NV independent ops, then 16 v_mfma_f32_16x16x4_f32), 2 or 4 wavefronts per SIMD, whole-CU timing.
// cost per op = (SIMD cycles per 16-MFMA round - 16 * 32.5) / NV.   Kinds: 0 v_add_f32, 1 v_fma_f32, 2 v_pk_add_f32,
// 3 v_pk_fma_f32, 4 v_cndmask (select), 5 v_max_f32, 6 ds_read_b64, 7 ds_read_b128, 8 v_mul_f32, 9 v_pk_mul_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int NV, int MAXT>
__global__ __launch_bounds__(MAXT) void k(float* out, long long* cyc, int iters, float seed) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 1e-3f;
  __syncthreads();
  f32x4 acc[16];
  float a = threadIdx.x * 1e-3f;
  f32x2 v[16], w[16];                 // register PAIRS throughout: packed kinds need no moves to form their operands
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = f32x2{seed * (i + 1) + threadIdx.x, seed * i}; w[i] = f32x2{seed + i, seed - i}; }
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{(float)i, 0.f, seed, 0.f};
  const float* lp = lds + ((threadIdx.x * 4) & 2047);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    f32x2 n[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) n[i] = v[i];
    // NV independent ops: op k reads v[], w[] (previous round) and writes (half of) n[k mod 16]
#pragma unroll
    for (int k2 = 0; k2 < NV; ++k2) {
      const int i = k2 & 15, j = (i + 7) & 15;
      const bool hi = (k2 >> 4) & 1;            // second pass over the array writes the .y halves for scalar kinds
      float dst = 0.f;
      const float s0 = hi ? v[j].y : v[j].x, s1 = hi ? w[i].y : w[i].x, s2 = hi ? v[i].y : v[i].x;
      if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(dst) : "v"(s0), "v"(s1));
      else if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(dst) : "v"(s0), "v"(s1), "v"(s2));
      else if (KIND == 8) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(dst) : "v"(s0), "v"(s1));
      else if (KIND == 5) asm volatile("v_max_f32 %0, %1, %2" : "=v"(dst) : "v"(s0), "v"(s1));
      else if (KIND == 4) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(dst) : "v"(s0), "v"(s1));
      else if (KIND == 10) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dst) : "v"(s0), "v"(s1));
      if (KIND == 0 || KIND == 1 || KIND == 8 || KIND == 5 || KIND == 4 || KIND == 10) { if (hi) n[i].y = dst; else n[i].x = dst; }
      if (KIND == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(n[i]) : "v"(v[j]), "v"(w[i]));
      else if (KIND == 9) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(n[i]) : "v"(v[j]), "v"(w[i]));
      else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(n[i]) : "v"(v[j]), "v"(w[i]), "v"(v[i]));
      else if (KIND == 6) n[i] = *reinterpret_cast<const volatile f32x2*>(lp + 64 * (k2 & 15) + ((it & 3) << 1));
      else if (KIND == 7) {
        const f32x4 r = *reinterpret_cast<const volatile f32x4*>(lp + 64 * (k2 & 15) + ((it & 3) << 2));
        n[i] = f32x2{r.x, r.y}; n[(i + 1) & 15] = f32x2{r.z, r.w};
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, n[i].x, acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = n[i];
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int KIND, int NV, int MAXT>
double run() {
  float* out; long long* cyc; (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 64 * 8);
  const int iters = 2000, threads = MAXT;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<KIND, NV, MAXT>), dim3(256), dim3(threads), 32768, 0, out, cyc, iters, 1.0f);
  (void)hipDeviceSynchronize();
  long long c[64]; (void)hipMemcpy(c, cyc, (threads / 64) * 16, hipMemcpyDeviceToHost);
  long long lo = c[0], hi = c[1];
  for (int w = 0; w < threads / 64; ++w) { lo = c[2 * w] < lo ? c[2 * w] : lo; hi = c[2 * w + 1] > hi ? c[2 * w + 1] : hi; }
  (void)hipFree(out); (void)hipFree(cyc);
  return (double)(hi - lo) / iters / (threads / 256);        // SIMD cycles per round (16 MFMAs + NV ops) per wave-round
}
template <int KIND>
void kind(const char* name) {
  const double b2 = run<KIND, 0, 512>(), b4 = run<KIND, 0, 1024>();
  const double a2 = run<KIND, 16, 512>(), c2 = run<KIND, 32, 512>(), a4 = run<KIND, 16, 1024>(), c4 = run<KIND, 32, 1024>();
  printf("%-14s 2 waves/SIMD: base %.0f, +16 ops %.1f cyc/op, +32 ops %.1f cyc/op | 4 waves/SIMD: base %.0f, +16 %.1f, +32 %.1f\n",
         name, b2, (a2 - b2) / 16, (c2 - b2) / 32, b4, (a4 - b4) / 16, (c4 - b4) / 32);
}
int main() {
  kind<0>("v_add_f32"); kind<10>("v_sub_f32"); kind<1>("v_fma_f32"); kind<8>("v_mul_f32"); kind<5>("v_max_f32"); kind<4>("v_cndmask_b32");
  kind<2>("v_pk_add_f32"); kind<9>("v_pk_mul_f32"); kind<3>("v_pk_fma_f32"); kind<6>("ds_read_b64"); kind<7>("ds_read_b128");
  return 0;
}
