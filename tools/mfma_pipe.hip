// Microbenchmark: can ONE wavefront hide its own operand VALU / LDS reads in the issue shadow of its f32 MFMAs when the
// operands are produced one batch AHEAD (software pipelining: VALU of batch k+1 interleaved between the MFMAs of batch k,
// no VALU->MFMA dependence closer than 16 MFMAs)?  tools/mfma_feed.hip measured the dependent / phase-separated forms.
//   mode 0: 16 MFMAs only                               (floor: 32 cycles per v_mfma_f32_16x16x4_f32 per SIMD)
//   mode 1: 32 scalar VALU, sched_barrier, 16 MFMAs     (the round-1 kernels' phase-separated form)
//   mode 2: per MFMA 2 scalar VALU of the NEXT batch    (sched_group_barrier interleave)
//   mode 3: per MFMA 1 v_pk_add_f32 of the next batch
//   mode 4: mode 2 + one ds_read_b64 per 2 MFMAs (the next batch's 4x4 patch)
//   mode 5: per MFMA 3 scalar VALU + ds_read_b64 per 2 MFMAs (VALU-heavier batch: output transform folded in)
//   mode 6: per MFMA 4 scalar VALU
//   mode 9 / 10 / 11: phase-separated 32 v_pk_add_f32 / 32 v_pk_fma_f32 / 64 v_fma_f32 per 16 MFMAs
//   mode 7 / 8: mode 1 with 64 / 96 VALU per 16 MFMAs (4 / 6 per MFMA: the conv kernels' real ratio)
// One workgroup per CU, 256 (1 wave/SIMD) or 512 (2 waves/SIMD) threads; prints cycles per MFMA per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int MAXT>
__global__ __launch_bounds__(MAXT) void k(float* out, long long* cyc, int iters, float seed) {
  extern __shared__ float lds[];
  f32x4 acc[16];
  float a = threadIdx.x * 1e-3f;
  float v[16], t[16];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 1e-3f;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) { t[i] = seed * (i + 1) + i * i + threadIdx.x; v[i] = t[i] * .5f; }
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{(float)i, 0.f, seed, 0.f};
  const float* lp = lds + (threadIdx.x * 2 & 1023);
  f32x2 raw[8];                       // LDS patch read one batch ahead of its use (two batches ahead of its MFMAs)
#pragma unroll
  for (int i = 0; i < 8; ++i) raw[i] = *reinterpret_cast<const f32x2*>(lp + 40 * i);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[i], acc[i], 0, 0, 0);
    } else if (MODE == 9 || MODE == 10 || MODE == 11) {
      float n[16];
      if (MODE == 11) {
#pragma unroll
        for (int i = 0; i < 16; ++i) n[i] = t[i];
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
          for (int i = 0; i < 16; ++i) n[i] = __builtin_fmaf(n[(i + 5) & 15], a, t[(i + rep) & 15]);
      } else {
        f32x2 np[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) np[i] = f32x2{t[2 * i], t[2 * i + 1]};
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const f32x2 y = {v[(2 * i + 3 + rep) & 15], v[(2 * i + 4 + rep) & 15]};
            f32x2 r;
            if (MODE == 9) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(np[(i + 3) & 7]), "v"(y));
            else asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(np[(i + 3) & 7]), "v"(y), "v"(np[i]));
            np[i] = r;
          }
#pragma unroll
        for (int i = 0; i < 8; ++i) { n[2 * i] = np[i].x; n[2 * i + 1] = np[i].y; }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, n[i], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = n[i];
    } else if (MODE == 1 || MODE == 7 || MODE == 8) {
      float n[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float u = t[i] - v[(i + 3) & 15]; n[i] = u + t[(i + 7) & 15]; }
      if (MODE >= 7) {
#pragma unroll
        for (int rep = 0; rep < (MODE == 7 ? 1 : 2); ++rep)
#pragma unroll
          for (int i = 0; i < 16; ++i) { const float u = n[i] - t[(i + 5) & 15]; n[i] = u + n[(i + 9) & 15]; }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, n[i], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = n[i];
    } else {
      constexpr int NV = MODE == 2 || MODE == 4 ? 2 : MODE == 5 ? 3 : MODE == 6 ? 4 : 1;
      constexpr bool LDSR = MODE == 4 || MODE == 5;
      float n[16];
      f32x2 rawn[8];
      if (LDSR) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rawn[i] = *reinterpret_cast<const f32x2*>(lp + 40 * i + ((it & 7) << 3));
      }
      if (MODE == 3) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          f32x2 x = {t[i], t[i + 1]}, y = {v[(i + 3) & 15], v[(i + 4) & 15]}, r;
          asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
          f32x2 r2;
          asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r2) : "v"(r), "v"(x));
          n[i] = r2.x; n[i + 1] = r2.y;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float u = t[i] - v[(i + 3) & 15];
          if (LDSR) u += (i & 1) ? raw[i >> 1].y : raw[i >> 1].x;
          u = u + t[(i + 7) & 15];
          if (NV >= 3) u = u - t[(i + 9) & 15];
          if (NV >= 4) u = u + t[(i + 11) & 15];
          n[i] = u;
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[i], acc[i], 0, 0, 0);
      // schedule: [1 MFMA, NV VALU (+ a DS read every other MFMA)] x 16
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (LDSR && (i & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, NV + (LDSR ? 1 : 0), 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = n[i];
      if (LDSR) {
#pragma unroll
        for (int i = 0; i < 8; ++i) raw[i] = rawn[i];
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  // every wavefront of workgroup 0 reports its own interval: the host takes last end - first start (a single
  // wavefront's own interval says nothing about the SIMD it shares: the older wavefront wins arbitration)
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { cyc[2 * (threadIdx.x >> 6)] = t0; cyc[2 * (threadIdx.x >> 6) + 1] = t1; }
}
template <int MODE, int MAXT>
void run() {
  const int threads = MAXT;
  float* out; long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 64 * 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, MAXT>), dim3(256), dim3(threads), 32768, 0, out, cyc, iters, 1.0f);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<MODE, MAXT>), dim3(256), dim3(threads), 32768, 0, out, cyc, iters, 1.0f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c[64]; hipMemcpy(c, cyc, (threads / 64) * 16, hipMemcpyDeviceToHost);
  long long lo = c[0], hi = c[1], own = 0;
  for (int w = 0; w < threads / 64; ++w) { lo = c[2 * w] < lo ? c[2 * w] : lo; hi = c[2 * w + 1] > hi ? c[2 * w + 1] : hi; own += c[2 * w + 1] - c[2 * w]; }
  const double per_simd = (double)(hi - lo) / (iters * 16.0) / (threads / 256);
  printf("mode=%d waves/SIMD=%d : SIMD cycles per MFMA %.1f (wave's own interval per MFMA %.1f; kernel %.3f ms = %.1f TFLOP/s MFMA)\n",
         MODE, threads / 256, per_simd, (double)own / (threads / 64) / (iters * 16.0), ms,
         256.0 * (threads / 64) * iters * 16 * 2048 / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0, 256>(); run<1, 256>(); run<2, 256>(); run<3, 256>(); run<4, 256>(); run<5, 256>(); run<6, 256>(); run<7, 256>(); run<8, 256>();
  run<0, 512>(); run<1, 512>(); run<2, 512>(); run<3, 512>(); run<4, 512>(); run<5, 512>(); run<6, 512>(); run<7, 512>(); run<8, 512>();
  run<9, 512>(); run<10, 512>(); run<11, 512>(); run<9, 1024>(); run<10, 1024>(); run<11, 1024>();
  run<0, 768>(); run<1, 768>(); run<7, 768>(); run<8, 768>(); run<0, 1024>(); run<1, 1024>(); run<7, 1024>(); run<8, 1024>();
  return 0;
}
