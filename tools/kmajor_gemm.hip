// Prototype: split-K fp32 GEMM with BOTH operands k-major:  C[128 x N] = A[K x 128]^T * B[K x N]  (fc1 weight gradient of the
// MNISTCNN: A = dY [78 400 x 128], B = X [78 400 x 2304]).  One 8-wavefront workgroup per CU owns a 128 x 128 output tile over
// a K range; K-slabs of 64 rows stream through two LDS stages in their global layout ([k][128], 512-B rows: no transpose on
// the way in).  Fragments: ONE ds_read_b128 at [k = 4 kk + q][m = 4 j .. 4 j + 3] feeds the A operand of FOUR MFMAs -- the
// MFMA's row i is mapped to m = 4 i + t for tile t (any row permutation is as good as another as long as the epilogue
// knows it); the B side likewise with 2 tiles per ds_read_b64.  Wavefront (wm, wn) of a 2 x 4 grid: 64 m x 32 n.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifndef TB
#define TB 2
#endif
#ifndef BKK
#define BKK 64
#endif
constexpr int MT = 128, BN = 64 * TB, BK = BKK, WAVES = 8;      // wavefront (wm, wn) of a 2 x 4 grid: 64 m x 16 TB n
#ifndef PITCH
#define PITCH 128
#endif
constexpr int LPA = 128, LPB = BN;           // LDS row pitches (floats)
constexpr int SLAB = BK * (LPA + LPB);       // A rows then B rows

__global__ __launch_bounds__(64 * WAVES, 1) void kmajor_k(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ part, long long K, int N, int nsplit, long long kps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int wm = wave >> 2, wn = wave & 3;           // 2 x 4 wavefronts: 64 m x 32 n each
  const int ntile = (N + BN - 1) / BN;
  const int tile = blockIdx.x % ntile, split = blockIdx.x / ntile;
  if (split >= nsplit) return;
  const long long k0 = split * kps, k1 = k0 + kps < K ? k0 + kps : K;
  const int n0 = tile * BN;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A) + k0 * MT, 0, (int)((k1 - k0) * MT * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B) + k0 * N, 0, (int)((k1 - k0) * N * 4), 0x00020000);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  // pieces of a slab: 64 rows x 32 quads for A, the same for B: 4 + 4 per thread; piece P -> row P / 32, quad P % 32
  constexpr int PA = BK * 32 / 512, PB = BK * (BN / 4) / 512;        // 16-B pieces per thread and slab
  f32x4 pa[PA], pb[PB];
  auto fetch = [&](int s) {
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int P = p * 512 + tid, row = P >> 5, c = P & 31;
      pa[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (unsigned)((s * BK + row) * MT + 4 * c) * 4u, 0, 0));
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int P = p * 512 + tid, row = P / (BN / 4), c = P % (BN / 4);
      const int gn = n0 + 4 * c;
      pb[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, gn < N ? (unsigned)((s * BK + row) * N + gn) * 4u : 0xfffffff0u, 0, 0));
    }
  };
  auto stash = [&](int stage) {
    float* a = smem + stage * SLAB;
    float* b = a + BK * LPA;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int P = p * 512 + tid, row = P >> 5, c = P & 31;
      *reinterpret_cast<f32x4*>(a + row * LPA + 4 * c) = pa[p];
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int P = p * 512 + tid, row = P / (BN / 4), c = P % (BN / 4);
      *reinterpret_cast<f32x4*>(b + row * LPB + 4 * c) = pb[p];
    }
  };
  f32x4 acc[4][TB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < TB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nslab = (int)((k1 - k0 + BK - 1) / BK);           // rows past k1 read as zeros (range check)
  fetch(0);
  stash(0);
  for (int s = 0; s < nslab; ++s) {
    __syncthreads();
    if (s + 1 < nslab) fetch(s + 1);
    const float* As = smem + (s & 1) * SLAB;
    const float* Bs = As + BK * LPA;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      const f32x4 af = *reinterpret_cast<const f32x4*>(As + (4 * kk + q) * LPA + 64 * wm + 4 * j);
      float bfv[TB];
      if constexpr (TB == 4) {
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(Bs + (4 * kk + q) * LPB + 64 * wn + 4 * j);
#pragma unroll
        for (int b = 0; b < 4; ++b) bfv[b] = t4[b];
      } else {
        const f32x2 t2 = *reinterpret_cast<const f32x2*>(Bs + (4 * kk + q) * LPB + 32 * wn + 2 * j);
        bfv[0] = t2[0]; bfv[1] = t2[1];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bfv[b], acc[a][b], 0, 0, 0);
    }
    if (s + 1 < nslab) stash((s + 1) & 1);
  }
  // D tile (a, b): lane (q, j) holds rows i = 4 q + r -> m = 64 wm + 4 i + a, column j -> n = 16 TB wn + TB j + b
  float* out = part + (long long)split * MT * N;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 64 * wm + 4 * (4 * q + r) + a, n = n0 + 16 * TB * wn + TB * j;
      if (n < N) {
        if constexpr (TB == 4) *reinterpret_cast<f32x4*>(out + (long long)m * N + n) = f32x4{acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
        else *reinterpret_cast<f32x2*>(out + (long long)m * N + n) = f32x2{acc[a][0][r], acc[a][1][r]};
      }
    }
}

__global__ void reduce_k(const float* __restrict__ part, float* __restrict__ C, int nsplit, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total / 4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<const f32x4*>(part)[i];
    for (int s = 1; s < nsplit; ++s) v += reinterpret_cast<const f32x4*>(part + s * total)[i];
    reinterpret_cast<f32x4*>(C)[i] = v;
  }
}

int main(int argc, char** argv) {
  const long long K = argc > 1 ? atoll(argv[1]) : 78400;
  const int N = argc > 2 ? atoi(argv[2]) : 2304;
  const int nsplit = argc > 3 ? atoi(argv[3]) : 14;
  std::vector<float> hA((size_t)K * MT), hB((size_t)K * N);
  srand(1);
  for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
  float *A, *B, *P, *C;
  (void)hipMalloc(&A, hA.size() * 4); (void)hipMalloc(&B, hB.size() * 4);
  (void)hipMalloc(&P, (size_t)nsplit * MT * N * 4); (void)hipMalloc(&C, (size_t)MT * N * 4);
  (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  const int ntile = (N + BN - 1) / BN;
  const long long kps = ((K + nsplit - 1) / nsplit + BK - 1) / BK * BK;
  const size_t lds = 2 * (size_t)SLAB * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kmajor_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  auto run = [&]() {
    hipLaunchKernelGGL(kmajor_k, dim3(ntile * nsplit), dim3(64 * WAVES), lds, 0, A, B, P, K, N, nsplit, kps);
    hipLaunchKernelGGL(reduce_k, dim3(256), dim3(256), 0, 0, P, C, nsplit, (long long)MT * N);
  };
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 300; ++r) run();
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 50; ++r) run();
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  printf("launch status: %s\n", hipGetErrorString(hipGetLastError()));
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 50;
  printf("K=%lld N=%d splits=%d (kps %lld) LDS %zu BN %d BK %d: %.4f ms (GEMM + reduction) = %.1f TFLOP/s\n", K, N, nsplit, kps, lds, BN, BK, ms,
         2.0 * MT * N * K / (ms * 1e-3) / 1e12);
  std::vector<float> hC((size_t)MT * N);
  (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  for (int t = 0; t < 400; ++t) {
    const int m = rand() % MT, n = (t & 1) ? rand() % N : N - 1 - rand() % 130;
    double s = 0;
    for (long long k = 0; k < K; ++k) s += (double)hA[k * MT + m] * hB[(size_t)k * N + n];
    maxerr = fmax(maxerr, fabs(s - hC[(size_t)m * N + n])); maxref = fmax(maxref, fabs(s));
  }
  printf("max abs err on 400 sampled entries: %.3e (max |ref| %.1f)\n", maxerr, maxref);
  return 0;
}
