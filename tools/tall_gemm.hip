// Prototype: persistent tall-skinny fp32 GEMM  C[M x N] = relu(A[M x K] * B[N x K]^T + bias),  N <= 128, K % 32 == 0,
// both operands k-contiguous (MNISTCNN fc1 forward: 78 400 x 128 x 2304).  One 8-wave workgroup per CU owns a
// (16 TMW * 4) x 128 row block; K-slabs of 32 stream through a double-buffered LDS ring by global_load_lds (no staging
// registers, no ds_write); fragments are ds_read_b128 (4 k per lane, XOR-swizzled chunk positions: conflict-free).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef WROWS
#define WROWS 4
#endif
constexpr int BK = 32, BN = 128, WAVES = 2 * WROWS;     // WROWS = 2: 160-row blocks, 4 wavefronts, TWO workgroups per CU

__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// LDS slab: rows of 32 floats = 8 chunks of 16 B; chunk c of row r sits at chunk position c ^ ((r >> 1) & 7)
template <int TMW>
__global__ __launch_bounds__(64 * WAVES, 8 / WAVES) void tall_gemm_k(const float* __restrict__ A, const float* __restrict__ B,
                                                            const float* __restrict__ bias, float* __restrict__ C,
                                                            long long M, int N, int K, int relu, long long* cyc) {
  constexpr int BM = 16 * WROWS * TMW;          // WROWS wave-rows x TMW tiles x 16
  constexpr int SLAB = (BM + BN) * BK;          // floats per stage
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;      // 4 x 2 wavefront grid
  const long long ntiles = (M + BM - 1) / BM;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long m0 = tile * BM;
    f32x4 acc[TMW][4];
#pragma unroll
    for (int a = 0; a < TMW; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this thread's pieces of a slab: piece p covers 16-B positions [p * 512 + tid]; position P -> row P / 8, chunk
    // position P % 8, holding logical chunk (P % 8) ^ ((row >> 1) & 7).  Global -> registers while the MFMAs of the
    // current slab run, registers -> LDS (one ds_write_b128 per piece) behind them.  (global_load_lds costs 60-185 issue
    // cycles per 1 KB piece next to MFMAs -- 7 per wavefront and slab were 17 % of the loop.)
    constexpr int NT_ = 64 * WAVES;
    constexpr int PIECES = (BM + BN) * 8 / NT_;
    const float* src[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const int P = p * NT_ + tid;
      const int row = P >> 3, c = (P & 7) ^ ((row >> 1) & 7);
      if (row < BM) {
        long long gm = m0 + row; if (gm >= M) gm = M - 1;
        src[p] = A + gm * K + 4 * c;
      } else {
        int gn = row - BM; if (gn >= N) gn = N - 1;
        src[p] = B + (long long)gn * K + 4 * c;
      }
    }
    f32x4 pre[PIECES];
    auto fetch = [&](int k0) {
#pragma unroll
      for (int p = 0; p < PIECES; ++p) pre[p] = *reinterpret_cast<const f32x4*>(src[p] + k0);
    };
    auto stash = [&](int stage) {
      float* base = smem + stage * SLAB + tid * 4;
#pragma unroll
      for (int p = 0; p < PIECES; ++p) *reinterpret_cast<f32x4*>(base + p * NT_ * 4) = pre[p];
    };
    fetch(0);
    stash(0);
    const int nslab = K / BK;
    const long long tc0 = __builtin_readcyclecounter();
    for (int s = 0; s < nslab; ++s) {
      __syncthreads();                             // slab s visible; everybody done reading the other stage
      if (s + 1 < nslab) fetch((s + 1) * BK);
      const float* As = smem + (s & 1) * SLAB;
      const float* Bs = As + BM * BK;
#ifndef VARIANT
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        f32x4 af[TMW], bf[4];
#pragma unroll
        for (int a = 0; a < TMW; ++a) {
          const int row = (wm * TMW + a) * 16 + j;
          af[a] = *reinterpret_cast<const f32x4*>(As + row * BK + 4 * ((4 * kg + q) ^ ((row >> 1) & 7)));
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int row = (wn * 4 + b) * 16 + j;
          bf[b] = *reinterpret_cast<const f32x4*>(Bs + row * BK + 4 * ((4 * kg + q) ^ ((row >> 1) & 7)));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int a = 0; a < TMW; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][r], bf[b][r], acc[a][b], 0, 0, 0);
      }
      if (s + 1 < nslab) stash((s + 1) & 1);
#else
      // both k-groups' fragments requested up front; the next slab goes to LDS between the two MFMA blocks
      f32x4 af[2][TMW], bf[2][4];
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
#pragma unroll
        for (int a = 0; a < TMW; ++a) {
          const int row = (wm * TMW + a) * 16 + j;
          af[kg][a] = *reinterpret_cast<const f32x4*>(As + row * BK + 4 * ((4 * kg + q) ^ ((row >> 1) & 7)));
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int row = (wn * 4 + b) * 16 + j;
          bf[kg][b] = *reinterpret_cast<const f32x4*>(Bs + row * BK + 4 * ((4 * kg + q) ^ ((row >> 1) & 7)));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int a = 0; a < TMW; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][a][r], bf[0][b][r], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < nslab) stash((s + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int a = 0; a < TMW; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][a][r], bf[1][b][r], acc[a][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if (cyc && blockIdx.x == 0 && lane == 0) cyc[wave] = __builtin_readcyclecounter() - tc0;
    __syncthreads();                               // all reads of the last slab done before the next tile's first issue
    // epilogue: D layout of a 16x16 tile: lane (q, j) holds rows 4q..4q+3 (of the A tile), column j (of the B tile)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int n = (wn * 4 + b) * 16 + j;
      const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
      for (int a = 0; a < TMW; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long m = m0 + (wm * TMW + a) * 16 + 4 * q + r;
          if (m < M && n < N) {
            float v = acc[a][b][r] + bv;
            if (relu) v = fmaxf(v, 0.f);
            C[m * N + n] = v;
          }
        }
    }
  }
}

int main(int argc, char** argv) {
  const long long M = argc > 1 ? atoll(argv[1]) : 78400;
  const int N = argc > 2 ? atoi(argv[2]) : 128, K = argc > 3 ? atoi(argv[3]) : 2304;
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hb(N);
  srand(1);
  for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hb) v = (rand() % 2001 - 1000) * 1e-3f;
  float *A, *B, *b, *C;
  (void)hipMalloc(&A, hA.size() * 4); (void)hipMalloc(&B, hB.size() * 4); (void)hipMalloc(&b, N * 4); (void)hipMalloc(&C, (size_t)M * N * 4);
  (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(b, hb.data(), N * 4, hipMemcpyHostToDevice);
  long long* cyc; (void)hipMalloc(&cyc, 64);
  constexpr int TMW = 5;
  const size_t lds = 2 * (size_t)(16 * WROWS * TMW + BN) * BK * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tall_gemm_k<TMW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const long long ntiles = (M + 16 * WROWS * TMW - 1) / (16 * WROWS * TMW);
  const unsigned grid = ntiles < 256 * (8 / WAVES) ? (unsigned)ntiles : 256 * (8 / WAVES);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 300; ++r) hipLaunchKernelGGL(tall_gemm_k<TMW>, dim3(grid), dim3(64 * WAVES), lds, 0, A, B, b, C, M, N, K, 1, cyc);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 50; ++r) hipLaunchKernelGGL(tall_gemm_k<TMW>, dim3(grid), dim3(64 * WAVES), lds, 0, A, B, b, C, M, N, K, 1, cyc);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  printf("launch status: %s\n", hipGetErrorString(hipGetLastError()));
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 50;   // 300 launches first: a GPU that has just been handed 700 MB of operands runs its first ~50 ms of kernels at ~2.15 GHz
  printf("M=%lld N=%d K=%d tiles=%lld: %.4f ms = %.1f TFLOP/s\n", M, N, K, ntiles, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
  long long hc[8]; (void)hipMemcpy(hc, cyc, 64, hipMemcpyDeviceToHost);
  printf("main loop cycles per slab (wave 0..7):"); for (int w = 0; w < 8; ++w) printf(" %lld", hc[w] / (K / BK)); printf("  (MFMA-bound: %d)\n", 2 * 160 * 32);
  std::vector<float> hC((size_t)M * N);
  (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int t = 0; t < 2000; ++t) {
    const long long m = (t < 700) ? (M - 1 - t) : (((long long)rand() << 8) ^ rand()) % M;
    const int n = rand() % N;
    double s = hb[n];
    for (int k = 0; k < K; ++k) s += (double)hA[m * K + k] * hB[(size_t)n * K + k];
    if (s < 0) s = 0;
    maxerr = fmax(maxerr, fabs(s - hC[m * N + n]));
  }
  printf("max abs err on 2000 sampled entries: %.3e\n", maxerr);
  return 0;
}
