// Prototype: persistent short-K fp32 GEMM  C[M x N] = A[M x K] * B[K x N],  K = 128 (fc1 data gradient: 78 400 x 2304 x 128,
// A = g_h1 k-contiguous, B = Wfc1 n-contiguous).  A workgroup keeps its 128-row block of A in LDS for all N tiles and
// streams 128-column tiles of B (double-buffered halves of K) -- the generic kernel re-reads the A tile per N tile and
// drains its pipeline every 4 slabs (77 TFLOP/s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 128, BM = 128, BN = 128, WAVES = 8, LDB = BN + 4;
// LDS: A block k-chunked like tall_gemm (rows of K floats = 32 chunks of 16 B, chunk c of row r at c ^ (r>>1 & 7) within
// each group of 8 chunks); B tile stored [n][k] (transposed on the way in) the same way so that both fragments are b128
__global__ __launch_bounds__(64 * WAVES, 1) void wide_gemm_k(const float* __restrict__ A, const float* __restrict__ B,
                                                            float* __restrict__ C, long long M, int N) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [BM][K]
  float* Bs = smem + BM * K;              // 2 x [K/2][LDB]  (double-buffered K halves, n-contiguous as in global memory)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;            // 4 x 2 wavefronts: 32 rows x 64 cols each = 2 x 4 tiles of 16x16
#ifdef V2
  constexpr int TA = 4, TB = 2;
  const int wm2 = wave >> 2, wn2 = wave & 3;          // 2 x 4 wavefronts: 64 rows x 32 cols each
  (void)wm; (void)wn;
#endif
  const long long nblk = (M + BM - 1) / BM;
  const int ntile = (N + BN - 1) / BN;
  // work units = (row block, N tile) pairs, dealt in contiguous ranges: 613 blocks over 256 CUs would leave the busiest
  // CU with 3 blocks against an average of 2.39; ranges of 43 units reload the A block at most 3 times
  const long long units = nblk * ntile;
  const long long u0 = units * blockIdx.x / gridDim.x, u1 = units * (blockIdx.x + 1) / gridDim.x;
  f32x4 pre[4];
  auto fetch = [&](long long u, int h) {
    const int t = (int)(u % ntile);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int P = p * 512 + tid, k = P >> 5, n4 = P & 31;          // k 0..63, 4 consecutive n
      int gn = 128 * t + 4 * n4; if (gn + 3 >= N) gn = N - 4;
      pre[p] = *reinterpret_cast<const f32x4*>(B + (long long)(64 * h + k) * N + gn);
    }
  };
  auto stash = [&](int buf) {             // [k][n] with row pitch LDB == 4 (mod 32): one ds_write_b128 per float4
    float* base = Bs + buf * (64 * LDB);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int P = p * 512 + tid, k = P >> 5, n4 = P & 31;
      *reinterpret_cast<f32x4*>(base + k * LDB + 4 * n4) = pre[p];
    }
  };
  long long cur_blk = -1;
  if (u0 < u1) { fetch(u0, 0); stash(0); }
  for (long long u = u0; u < u1; ++u) {
    const long long blk = u / ntile;
    const int t = (int)(u - blk * ntile);
    const long long m0 = blk * BM;
    if (blk != cur_blk) {                   // (re)load the A block: 128 x 128 floats = 4096 16-B chunks, 8 per thread
      __syncthreads();
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int P = p * 512 + tid, row = P >> 5, c = P & 31;
        long long gm = m0 + row; if (gm >= M) gm = M - 1;
        const f32x4 v = *reinterpret_cast<const f32x4*>(A + gm * K + 4 * c);
        *reinterpret_cast<f32x4*>(As + row * K + 4 * ((c & ~7) | ((c & 7) ^ ((row >> 1) & 7)))) = v;
      }
      cur_blk = blk;
    }
#ifdef V2
    f32x4 acc2[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
      for (int b = 0; b < TB; ++b) acc2[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __syncthreads();
      const long long nu = h == 1 ? u + 1 : u;
      const int nh = h ^ 1;
      const bool more = nu < u1;
      if (more) fetch(nu, nh);
      const float* Bh = Bs + h * (64 * LDB);
#ifdef V2
      // wavefront (wm2, wn2) of a 2 x 4 grid: 64 rows x 32 columns = TA x TB tiles; the B-matrix fragment is the MFMA's FIRST
      // operand, so a lane's four accumulator registers are four consecutive n of one row m: float4 stores
      f32x4 af[2][TA], bf[2][TB];
      auto frags = [&](int kg, int slot) {
#pragma unroll
        for (int a = 0; a < TA; ++a) {
          const int row = (wm2 * TA + a) * 16 + j, c = 16 * h + 4 * kg + q;
          af[slot][a] = *reinterpret_cast<const f32x4*>(As + row * K + 4 * ((c & ~7) | ((c & 7) ^ ((row >> 1) & 7))));
        }
#pragma unroll
        for (int b = 0; b < TB; ++b) {
          const float* pb = Bh + (16 * kg + 4 * q) * LDB + (wn2 * TB + b) * 16 + j;
#pragma unroll
          for (int r = 0; r < 4; ++r) bf[slot][b][r] = pb[r * LDB];
        }
      };
      frags(0, 0);
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        if (kg < 3) frags(kg + 1, (kg + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int b = 0; b < TB; ++b)
              acc2[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[kg & 1][b][r], af[kg & 1][a][r], acc2[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) stash(nh);
    }
#ifdef NOSTORE
    if (M < 0)                 // measurement only: never true, the accumulators stay live
#endif
#pragma unroll
    for (int a = 0; a < TA; ++a) {
      const long long m = m0 + (wm2 * TA + a) * 16 + j;
#pragma unroll
      for (int b = 0; b < TB; ++b) {
        const int n = 128 * t + (wn2 * TB + b) * 16 + 4 * q;
        if (m < M && n < N) *reinterpret_cast<f32x4*>(C + m * N + n) = acc2[a][b];
      }
    }
  }
}
#else
      f32x4 af[2][2], bf[2][4];
      auto frags = [&](int kg, int slot) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int row = (wm * 2 + a) * 16 + j, c = 16 * h + 4 * kg + q;
          af[slot][a] = *reinterpret_cast<const f32x4*>(As + row * K + 4 * ((c & ~7) | ((c & 7) ^ ((row >> 1) & 7))));
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float* pb = Bh + (16 * kg + 4 * q) * LDB + (wn * 4 + b) * 16 + j;
#pragma unroll
          for (int r = 0; r < 4; ++r) bf[slot][b][r] = pb[r * LDB];
        }
      };
      frags(0, 0);
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        if (kg < 3) frags(kg + 1, (kg + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kg & 1][a][r], bf[kg & 1][b][r], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) stash(nh);
    }
#ifdef NOSTORE
    if (M < 0)
#endif
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int n = 128 * t + (wn * 4 + b) * 16 + j;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long m = m0 + (wm * 2 + a) * 16 + 4 * q + r;
          if (m < M && n < N) C[m * N + n] = acc[a][b][r];
        }
    }
  }
}
#endif
#ifdef V3
// V3: the B halves are requested so that no waited load sits behind the previous unit's stores in the (in-order) vmcnt queue:
// (u+1, h=1) at the top of (u, h=1), (u+2, h=0) at the END of (u, h=1) in front of the stores of unit u; they go to LDS at the
// top of (u+1, h=0) and (u+1, h=1).  Two register sets in flight.
__global__ __launch_bounds__(64 * WAVES, 1) void wide_gemm3_k(const float* __restrict__ A, const float* __restrict__ B,
                                                             float* __restrict__ C, long long M, int N) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + BM * K;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const long long nblk = (M + BM - 1) / BM;
  const int ntile = (N + BN - 1) / BN;
  const long long units = nblk * ntile;
  const long long u0 = units * blockIdx.x / gridDim.x, u1 = units * (blockIdx.x + 1) / gridDim.x;
  f32x4 pre[2][4];
  auto fetch = [&](f32x4 (&dst)[4], long long u, int h) {
    const int t = (int)(u % ntile);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int P = p * 512 + tid, k = P >> 5, n4 = P & 31;
      int gn = 128 * t + 4 * n4; if (gn + 3 >= N) gn = N - 4;
      dst[p] = *reinterpret_cast<const f32x4*>(B + (long long)(64 * h + k) * N + gn);
    }
  };
  auto stash = [&](const f32x4 (&src)[4], int buf) {
    float* base = Bs + buf * (64 * LDB);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int P = p * 512 + tid, k = P >> 5, n4 = P & 31;
      *reinterpret_cast<f32x4*>(base + k * LDB + 4 * n4) = src[p];
    }
  };
  long long cur_blk = -1;
  if (u0 < u1) {
    fetch(pre[0], u0, 0); stash(pre[0], 0);
    fetch(pre[1], u0, 1);
    fetch(pre[0], u0 + 1 < u1 ? u0 + 1 : u0, 0);

  }
  for (long long u = u0; u < u1; ++u) {
    const long long blk = u / ntile;
    const int t = (int)(u - blk * ntile);
    const long long m0 = blk * BM;
    if (blk != cur_blk) {
      __syncthreads();
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int P = p * 512 + tid, row = P >> 5, c = P & 31;
        long long gm = m0 + row; if (gm >= M) gm = M - 1;
        const f32x4 v = *reinterpret_cast<const f32x4*>(A + gm * K + 4 * c);
        *reinterpret_cast<f32x4*>(As + row * K + 4 * ((c & ~7) | ((c & 7) ^ ((row >> 1) & 7)))) = v;
      }
      cur_blk = blk;
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __syncthreads();
      if (h == 0) {
        stash(pre[1], 1);                                   // (u, 1): requested at the top of (u-1, 1)
      } else {
        // (u+1, 0): requested at the end of (u-1, 1).  Unconditional (a clamped unit past the end, never used): a request
        // under a branch makes hipcc lose count of the queue and wait for vmcnt(0..3), i.e. for the stores
        stash(pre[0], 0);
        fetch(pre[1], u + 1 < u1 ? u + 1 : u, 1);
      }
      const float* Bh = Bs + h * (64 * LDB);
      f32x4 af[2][2], bf[2][4];
      auto frags = [&](int kg, int slot) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int row = (wm * 2 + a) * 16 + j, c = 16 * h + 4 * kg + q;
          af[slot][a] = *reinterpret_cast<const f32x4*>(As + row * K + 4 * ((c & ~7) | ((c & 7) ^ ((row >> 1) & 7))));
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float* pb = Bh + (16 * kg + 4 * q) * LDB + (wn * 4 + b) * 16 + j;
#pragma unroll
          for (int r = 0; r < 4; ++r) bf[slot][b][r] = pb[r * LDB];
        }
      };
      frags(0, 0);
#pragma unroll
      for (int kg = 0; kg < 4; ++kg) {
        if (kg < 3) frags(kg + 1, (kg + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kg & 1][a][r], bf[kg & 1][b][r], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (h == 1 && kg == 0) { fetch(pre[0], u + 2 < u1 ? u + 2 : u, 0); __builtin_amdgcn_sched_barrier(0); }   // (u+2, 0)
      }
    }
    // Every request of this unit is complete BEFORE its stores are issued (a use of the registers: hipcc waits here, where the
    // loads are 3 / 1.5 k-groups old, and knows afterwards that they have arrived).  Otherwise the waits land in front of the
    // LDS writes of the next unit, behind 32 stores in the in-order vmcnt queue, and hipcc -- which derives the count from
    // the shortest path into the loop -- makes them wait for the stores.
#pragma unroll
#ifndef NOTOUCH
    for (int p = 0; p < 4; ++p) asm volatile("" : "+v"(pre[0][p]), "+v"(pre[1][p]));
#endif
    __builtin_amdgcn_sched_barrier(0);
    // unconditional buffer stores, the rows past M and the columns past N dropped by the range check of the descriptor: with
    // the stores under a branch hipcc cannot count them and waits for vmcnt(0..3) at the next LDS write -- i.e. for the stores
    const long long rows = M - m0 < BM ? M - m0 : BM;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(C + m0 * N, 0, (int)(rows * N * 4), 0x00020000);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int n = 128 * t + (wn * 4 + b) * 16 + j;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (wm * 2 + a) * 16 + 4 * q + r;
          const unsigned off = n < N ? (unsigned)(row * N + n) * 4u : 0xfffffff0u;
#ifdef PLAINST
          if (m0 + row < M && n < N) C[(m0 + row) * N + n] = acc[a][b][r];
#else
          const float val = acc[a][b][r];
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rs, off, 0, 0);
#endif
        }
    }
  }
}
#define wide_gemm_k wide_gemm3_k
#endif

int main() {
  const long long M = 78400; const int N = 2304;
  std::vector<float> hA((size_t)M * K), hB((size_t)K * N);
  srand(1);
  for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
  float *A, *B, *C;
  (void)hipMalloc(&A, hA.size() * 4); (void)hipMalloc(&B, hB.size() * 4); (void)hipMalloc(&C, (size_t)M * N * 4);
  (void)hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
  const size_t lds = (size_t)(BM * K + 2 * 64 * LDB) * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_gemm_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 300; ++r) hipLaunchKernelGGL(wide_gemm_k, dim3(256), dim3(64 * WAVES), lds, 0, A, B, C, M, N);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 50; ++r) hipLaunchKernelGGL(wide_gemm_k, dim3(256), dim3(64 * WAVES), lds, 0, A, B, C, M, N);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  printf("launch status: %s\n", hipGetErrorString(hipGetLastError()));
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 50;      // behind 300 warm-up launches: a GPU that has just started runs at ~2.15 GHz
  printf("M=%lld N=%d K=%d: %.4f ms = %.1f TFLOP/s (LDS %zu B)\n", M, N, K, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12, lds);
  std::vector<float> hC((size_t)M * N);
  (void)hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int t = 0; t < 3000; ++t) {
    const long long m = (t < 700) ? (M - 1 - t) : (((long long)rand() << 8) ^ rand()) % M;
    const int n = (t & 1) ? rand() % N : N - 1 - (rand() % 130);
    double s = 0;
    for (int k = 0; k < K; ++k) s += (double)hA[m * K + k] * hB[(size_t)k * N + n];
    maxerr = fmax(maxerr, fabs(s - hC[m * N + n]));
  }
  printf("max abs err on 3000 sampled entries: %.3e\n", maxerr);
  return 0;
}
