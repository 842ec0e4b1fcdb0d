// fp32 MFMA GEMM with fused weight-mask, bias, output-mask, ReLU and ReLU-gate epilogues.
// One kernel serves every dense contraction of the conditioners (MADE masked linears,
// Coupling/DAG MLPs, MNISTCNN fc layers) and of their backward (dX, dW) through generic
// element strides, so no transposed copies are ever materialised.
//
// gfx950 mapping: 256-thread workgroup = 4 wavefronts in a 2x2 grid; each wavefront owns
// a (BM/2)x(BN/2) sub-tile as 32x32 accumulators of v_mfma_f32_32x32x2_f32 (exact fp32,
// k-ordered fma chain -> 1e-6-level agreement with the reference's fp32 addmm).  A and B
// K-slabs (BK=16) are staged k-major in LDS so that a fragment read is 32 consecutive
// dwords per half-wave (conflict-free ds_read_b32); the next slab is prefetched into
// registers while the MFMAs of the current one run (issue-early / write-late).
#include "gnf_common.h"
#include "gnf_gemm.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));


constexpr int BK = 16;
constexpr int PAD = 4;

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_k(GemmArgs g) {
  constexpr int WM = BM / 2, WN = BN / 2;   // wavefront sub-tile
  constexpr int TM = WM / 32, TN = WN / 32; // 32x32 MFMA tiles per wavefront
  constexpr int LA = BM * BK / 256, LB = BN * BK / 256;
  __shared__ float As[BK][BM + PAD];
  __shared__ float Bs[BK][BN + PAD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;
  const int64_t m0 = (int64_t)blockIdx.x * BM, n0 = (int64_t)blockIdx.y * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
  const int64_t kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
  float* __restrict__ Cz = g.C + (int64_t)blockIdx.z * g.c_split_stride;

  // thread -> (row/col, k) mapping for the global loads, chosen so that consecutive lanes
  // walk the contiguous dimension of the operand
  const bool a_kfast = (g.sak == 1 && g.sam != 1);
  const bool b_kfast = (g.sbk == 1 && g.sbn != 1);

  float ra[LA], rb[LB];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto load_tile = [&](int64_t k0) {
#pragma unroll
    for (int it = 0; it < LA; ++it) {
      const int idx = tid + it * 256;
      const int m = a_kfast ? idx / BK : idx % BM;
      const int k = a_kfast ? idx % BK : idx / BM;
      const int64_t gm = m0 + m, gk = k0 + k;
      ra[it] = (gm < g.M && gk < kend) ? g.A[gm * g.sam + gk * g.sak] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < LB; ++it) {
      const int idx = tid + it * 256;
      const int n = b_kfast ? idx / BK : idx % BN;
      const int k = b_kfast ? idx % BK : idx / BN;
      const int64_t gn = n0 + n, gk = k0 + k;
      float v = 0.f;
      if (gn < g.N && gk < kend) {
        const int64_t off = gk * g.sbk + gn * g.sbn;
        v = g.B[off];
        if (g.Bmask) v *= g.Bmask[off];
      }
      rb[it] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int it = 0; it < LA; ++it) {
      const int idx = tid + it * 256;
      const int m = a_kfast ? idx / BK : idx % BM;
      const int k = a_kfast ? idx % BK : idx / BM;
      As[k][m] = ra[it];
    }
#pragma unroll
    for (int it = 0; it < LB; ++it) {
      const int idx = tid + it * 256;
      const int n = b_kfast ? idx / BK : idx % BN;
      const int k = b_kfast ? idx % BK : idx / BN;
      Bs[k][n] = rb[it];
    }
  };

  const int fi = lane & 31, fk = lane >> 5;
  load_tile(kbeg);
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();            // previous slab fully consumed
    store_tile();
    __syncthreads();
    if (k0 + BK < kend) load_tile(k0 + BK);   // in flight during the MFMAs below
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = As[2 * ks + fk][wm + 32 * i + fi];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = Bs[2 * ks + fk][wn + 32 * j + fi];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }

  // epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t n = n0 + wn + 32 * j + fi;
      if (n >= g.N) continue;
      const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (m >= g.M) continue;
        float v = acc[i][j][r] + bv;
        if (g.Cmask) v *= g.Cmask[m * g.scmm + n * g.scmn];
        if (g.flags & GNF_GEMM_RELU) v = fmaxf(v, 0.f);
        if (g.gate) v = g.gate[m * g.sgm + n * g.sgn] > 0.f ? v : 0.f;
        float* cp = Cz + m * g.scm + n * g.scn;
        *cp = (g.flags & GNF_GEMM_ACCUM) ? *cp + v : v;
      }
    }
}


// split-K epilogue: C = epi(sum_z partial[z]) with the same options as the fused epilogue
__global__ void gemm_reduce_k(const float* __restrict__ part, int64_t nsp, GemmArgs g) {
  const int64_t total = g.M * g.N;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = idx / g.N, n = idx - m * g.N;
    float v = 0.f;
    for (int64_t z = 0; z < nsp; ++z) v += part[z * total + idx];
    if (g.bias) v += g.bias[n];
    if (g.Cmask) v *= g.Cmask[m * g.scmm + n * g.scmn];
    if (g.flags & GNF_GEMM_RELU) v = fmaxf(v, 0.f);
    if (g.gate) v = g.gate[m * g.sgm + n * g.sgn] > 0.f ? v : 0.f;
    g.C[m * g.scm + n * g.scn] = v;
  }
}

}  // namespace

static int64_t k_per_split(int64_t K, int splits) {
  if (splits < 1) splits = 1;
  int64_t kps = ((K + splits - 1) / splits + BK - 1) / BK * BK;
  return kps < BK ? BK : kps;
}

int64_t gnf_gemm_num_splits(int64_t K, int splits) {
  const int64_t kps = k_per_split(K, splits);
  return K > 0 ? (K + kps - 1) / kps : 1;
}

// Internal launcher shared with gnf_monotonic.hip (split-K weight-gradient GEMMs).
int gnf_gemm_launch(GemmArgs g, int splits, hipStream_t s) {
  g.k_per_split = k_per_split(g.K, splits);
  // an accumulating launch must cover every partial written by the first one: exactly `splits` of them
  const int64_t nsp = (g.flags & GNF_GEMM_ACCUM) ? splits : gnf_gemm_num_splits(g.K, splits);
  // 128x128 tiles once there is >= 2 workgroups per CU of them, else 64x64
  const int64_t big = ((g.M + 127) / 128) * ((g.N + 127) / 128) * nsp;
  const int bt = big >= 512 ? 128 : 64;
  const int64_t gx = (g.M + bt - 1) / bt, gy = (g.N + bt - 1) / bt;
  if (gy > 65535 || nsp > 65535) return GNF_ESHAPE;
  const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)nsp);
  if (bt == 128) hipLaunchKernelGGL((gemm_k<128, 128>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_k<64, 64>), grid, dim3(256), 0, s, g);
  GNF_LAUNCH_CHECK();
  return 0;
}

// split-K plan of the public entry: few output tiles and a long K -> spread K over the chip
static int plan_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + 63) / 64) * ((N + 63) / 64);
  if (tiles >= 256 || K < 2048) return 1;
  int64_t s = (768 + tiles - 1) / tiles;
  if (s > K / 512) s = K / 512;
  if (s > 512) s = 512;
  return s < 2 ? 1 : (int)s;
}

extern "C" int64_t gnf_gemm_ws_bytes(int64_t M, int64_t N, int64_t K) {
  const int s = plan_splits(M, N, K);
  return s > 1 ? (int64_t)s * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int gnf_gemm(const float* A, int64_t sam, int64_t sak, const float* B, const float* Bmask, int64_t sbk,
                        int64_t sbn, float* C, int64_t scm, int64_t scn, const float* bias, const float* Cmask,
                        int64_t scmm, int64_t scmn, const float* gate, int64_t sgm, int64_t sgn, int flags,
                        int64_t M, int64_t N, int64_t K, float* ws, int64_t ws_bytes, gnf_stream_t stream) {
  if (!A || !B || !C || M < 0 || N < 0 || K < 0) return GNF_EINVAL;
  if (M == 0 || N == 0) return 0;
  GemmArgs g{A, sam, sak, B, Bmask, sbk, sbn, C, scm, scn, bias, Cmask, scmm, scmn, gate, sgm, sgn,
             flags & GNF_GEMM_RELU, M, N, K, 0, 0};
  const int splits = plan_splits(M, N, K);
  if (splits > 1 && ws && ws_bytes >= gnf_gemm_ws_bytes(M, N, K)) {
    GemmArgs p = g;                       // partial products only: epilogue runs in the reduction
    p.C = ws; p.scm = N; p.scn = 1; p.c_split_stride = M * N;
    p.bias = nullptr; p.Cmask = nullptr; p.gate = nullptr; p.flags = 0;
    int rc = gnf_gemm_launch(p, splits, (hipStream_t)stream);
    if (rc) return rc;
    const int64_t nsp = gnf_gemm_num_splits(K, splits);
    int64_t grid = (M * N + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(gemm_reduce_k, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, ws, nsp, g);
    GNF_LAUNCH_CHECK();
    return 0;
  }
  return gnf_gemm_launch(g, 1, (hipStream_t)stream);
}
