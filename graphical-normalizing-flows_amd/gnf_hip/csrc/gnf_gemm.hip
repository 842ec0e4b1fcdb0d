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


// ---------------------------------------------------------------------------------------------
// Vectorised variant for operands whose contiguous dimension is 16-B aligned (every hot shape of
// the flows): 128-bit global loads, K-slab 32.  A k-contiguous operand is staged row-major
// [rows][32+1] (odd stride: conflict-free fragment reads and scattered writes), a row-contiguous one
// k-major [32][rows+4] with ds_write_b128.  Same MFMA tiling and epilogue as gemm_k.
// ---------------------------------------------------------------------------------------------
typedef float f32x4v __attribute__((ext_vector_type(4)));
// global side: only dword alignment is assumed.  gfx950 executes global_load_dwordx4 at any dword address (checked by
// tools/unaligned_vec.hip), so operands with odd leading strides (MADE's K = 630, the 30-wide embedding outputs) take
// the vector path too; the LDS side keeps its 16-B aligned layout.
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
constexpr int BKV = 32;

template <int BX, bool KF>
struct Slab {            // staging of one operand: BX rows/cols x 32 k
  static constexpr int NV = BX * BKV / 4 / 256;                       // float4 per thread
  static constexpr int SZ = KF ? BX * (BKV + 1) : BKV * (BX + 4);
  // coordinates of this thread's it-th float4: x = row/col inside the tile, k inside the slab
  static __device__ __forceinline__ void coord(int tid, int it, int& x, int& k) {
    const int idx = tid + it * 256;
    if (KF) { x = idx / (BKV / 4); k = 4 * (idx % (BKV / 4)); }
    else { x = 4 * (idx % (BX / 4)); k = idx / (BX / 4); }
  }
  static __device__ __forceinline__ f32x4v load(const float* __restrict__ P, const float* __restrict__ Mk,
                                                 int64_t sx, int64_t sk, int64_t gx, int64_t gk, int64_t X,
                                                 int64_t kend) {
    f32x4v v = {0.f, 0.f, 0.f, 0.f};
    if (KF) {
      if (gx < X) {
        const int64_t o = gx * sx + gk;
        if (gk + 3 < kend) {
          v = *reinterpret_cast<const f32x4u*>(P + o);
          if (Mk) v *= *reinterpret_cast<const f32x4u*>(Mk + o);
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (gk + c < kend) v[c] = Mk ? P[o + c] * Mk[o + c] : P[o + c];
        }
      }
    } else {
      if (gk < kend) {
        const int64_t o = gk * sk + gx;
        if (gx + 3 < X) {
          v = *reinterpret_cast<const f32x4u*>(P + o);
          if (Mk) v *= *reinterpret_cast<const f32x4u*>(Mk + o);
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (gx + c < X) v[c] = Mk ? P[o + c] * Mk[o + c] : P[o + c];
        }
      }
    }
    return v;
  }
  static __device__ __forceinline__ void store(float* S, int x, int k, const f32x4v& v) {
    if (KF) {
#pragma unroll
      for (int c = 0; c < 4; ++c) S[x * (BKV + 1) + k + c] = v[c];
    } else {
      *reinterpret_cast<f32x4v*>(S + k * (BX + 4) + x) = v;
    }
  }
  static __device__ __forceinline__ float frag(const float* S, int x, int k) {
    return KF ? S[x * (BKV + 1) + k] : S[k * (BX + 4) + x];
  }
};

// WGM = wavefronts along M: 2 (2x2 grid, the default) or 1 (1x4: every wavefront owns all BM rows and a quarter of the
// columns -- the 160x128 tile that splits M = 78 400 into 490 tiles = 1.9 per CU instead of 613 = 2.4 of the 128x128 one)
template <int BM, int BN, bool AKF, bool BKF, int WGM = 2>
__global__ __launch_bounds__(256) void gemm_vec_k(GemmArgs g) {
  constexpr int WM = BM / WGM, WN = BN / (4 / WGM), TM = WM / 32, TN = WN / 32;
  static_assert(WM % 32 == 0 && WN % 32 == 0, "wavefront sub-tile is a multiple of the 32x32 MFMA tile");
  using SA = Slab<BM, AKF>;
  using SB = Slab<BN, BKF>;
  __shared__ __attribute__((aligned(16))) float As[SA::SZ];
  __shared__ __attribute__((aligned(16))) float Bs[SB::SZ];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WGM == 2 ? (wave >> 1) * WM : 0, wn = WGM == 2 ? (wave & 1) * WN : wave * WN;
  const int64_t m0 = (int64_t)blockIdx.x * BM, n0 = (int64_t)blockIdx.y * BN;
  int64_t kbeg = (int64_t)blockIdx.z * g.k_per_split;
  int64_t kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
  float* __restrict__ Cz = g.C + (int64_t)blockIdx.z * g.c_split_stride;
  const float* __restrict__ Ag = g.A;
  const float* __restrict__ Bg = g.B;
  int64_t Mr = g.M;
  if (g.grp) {                               // grouped launch: blockIdx.z selects a row range and its own B
    const int64_t first = g.grp[2 * blockIdx.z], cnt = g.grp[2 * blockIdx.z + 1];
    if (g.grp_k) {                           // K range of the shared operands, own output
      kbeg = first;
      kend = first + cnt;
    } else {
      Mr = cnt;
      if (m0 >= Mr) return;                  // workgroup-uniform
      Ag += first * g.sam;
      Bg += (int64_t)blockIdx.z * g.b_grp_stride;
      Cz = g.C + first * g.scm;
      kbeg = 0;
      kend = g.K;
    }
  }

  f32x4v ra[SA::NV], rb[SB::NV];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto load_tile = [&](int64_t k0) {
#pragma unroll
    for (int it = 0; it < SA::NV; ++it) {
      int x, k;
      SA::coord(tid, it, x, k);
      ra[it] = SA::load(Ag, nullptr, g.sam, g.sak, m0 + x, k0 + k, Mr, kend);
    }
#pragma unroll
    for (int it = 0; it < SB::NV; ++it) {
      int x, k;
      SB::coord(tid, it, x, k);
      rb[it] = SB::load(Bg, g.Bmask, g.sbn, g.sbk, n0 + x, k0 + k, g.N, kend);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int it = 0; it < SA::NV; ++it) {
      int x, k;
      SA::coord(tid, it, x, k);
      SA::store(As, x, k, ra[it]);
    }
#pragma unroll
    for (int it = 0; it < SB::NV; ++it) {
      int x, k;
      SB::coord(tid, it, x, k);
      SB::store(Bs, x, k, rb[it]);
    }
  };

  const int fi = lane & 31, fk = lane >> 5;
  load_tile(kbeg);
  for (int64_t k0 = kbeg; k0 < kend; k0 += BKV) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (k0 + BKV < kend) load_tile(k0 + BKV);
#pragma unroll
    for (int ks = 0; ks < BKV / 2; ++ks) {
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = SA::frag(As, wm + 32 * i + fi, 2 * ks + fk);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = SB::frag(Bs, wn + 32 * j + fi, 2 * ks + fk);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
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
        if (m >= Mr) continue;
        float v = acc[i][j][r] + bv;
        if (g.Cmask) v *= g.Cmask[m * g.scmm + n * g.scmn];
        if (g.flags & GNF_GEMM_RELU) v = fmaxf(v, 0.f);
        if (g.gate) v = g.gate[m * g.sgm + n * g.sgn] > 0.f ? v : 0.f;
        float* cp = Cz + m * g.scm + n * g.scn;
        *cp = (g.flags & GNF_GEMM_ACCUM) ? *cp + v : v;
      }
    }
}

// ---------------------------------------------------------------------------------------------
// Short-K, wide-N, tall-M product  C[M x N] = A[M x 128] * B[128 x N]  with A k-contiguous and B n-contiguous: the data
// gradient of the MNISTCNN fc1 layer (78 400 x 2304 x 128).  The tiled kernels above drain their pipeline every 4 slabs
// (K = 128), re-read the A tile for each of the 18 N tiles, and 613 row blocks over 256 CUs leave the busiest CU with
// 3 against an average of 2.39 (77 TFLOP/s).  Here one 8-wave workgroup per CU walks a contiguous range of (row block,
// N tile) UNITS: the 128 x 128 A block stays in LDS while the range stays inside a block, the B tiles stream through a
// double buffer of K halves (global -> registers during the MFMAs -> one ds_write_b128 per float4), the fragments of MFMA
// group kg+1 are requested ahead of the MFMAs of group kg.  A: 16-B chunks XOR-swizzled per row (conflict-free b128
// fragment reads); B: [k][n] with pitch 132 == 4 (mod 32): MFMA step r contracts k = 16 kg + 4 q + r, four conflict-free
// b32 reads per fragment.  90 TFLOP/s (tools/wide_gemm.hip): 0.60 -> 0.51 ms.
// ---------------------------------------------------------------------------------------------
typedef float f32x4w __attribute__((ext_vector_type(4)));
typedef float f32x4wu __attribute__((ext_vector_type(4), aligned(4)));
constexpr int WK = 128, WBM = 128, WBN = 128, WWAVES = 8, WLDB = WBN + 4;
constexpr size_t kWideLds = (size_t)(WBM * WK + 2 * 64 * WLDB) * sizeof(float);

// Addressing: one buffer descriptor per operand / row block, 32-bit lane offsets (the 64-bit `m * scm + n` of the first
// version cost 100 VALU and 110 SALU instructions per unit and wavefront: 0.459 ms against the 0.419 of the stand-alone form
// with compile-time strides); rows past M and columns past N are dropped (stores) or read as 0 (loads) by the range check, so
// no memory instruction sits under a branch.  Request order: half (u+1, 1) at the top of (u, 1), half (u+2, 0) inside (u, 1),
// BOTH complete before the stores of unit u are issued (a use of their registers: hipcc waits there, where the loads are
// old, and not in front of the next unit's LDS writes, behind 32 stores in the in-order vmcnt queue).
__global__ __launch_bounds__(64 * WWAVES, 1) void gemm_wide_k(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  float* As = wsm;                        // [WBM][WK], chunk-swizzled
  float* Bs = wsm + WBM * WK;             // 2 x [64][WLDB]
  typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
  const int64_t M = g.M;
  const int N = (int)g.N, sam = (int)g.sam, sbk = (int)g.sbk, scm = (int)g.scm;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;            // 4 x 2 wavefronts: 32 rows x 64 cols each = 2 x 4 tiles of 16x16
  const int64_t nblk = (M + WBM - 1) / WBM;
  const int ntile = (N + WBN - 1) / WBN;
  const int64_t units = nblk * ntile;
  const int64_t u0 = units * blockIdx.x / gridDim.x, u1 = units * (blockIdx.x + 1) / gridDim.x;
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0, (int)(((int64_t)(WK - 1) * sbk + N) * 4), 0x00020000);
  f32x4w pre[2][4];
  auto fetch = [&](f32x4w (&dst)[4], int64_t u, int h) {
    const int t = (int)(u % ntile);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int P = p * 512 + tid, k = P >> 5, n4 = P & 31;          // k 0..63, 4 consecutive n
      const int gn = WBN * t + 4 * n4;                               // N % 4 == 0: a quad is inside the row or past it
      const unsigned off = gn < N ? (unsigned)((64 * h + k) * sbk + gn) * 4u : 0xfffffff0u;
      dst[p] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsB, off, 0, 0));
    }
  };
  auto stash = [&](const f32x4w (&src)[4], int buf) {
    float* base = Bs + buf * (64 * WLDB);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int P = p * 512 + tid, k = P >> 5, n4 = P & 31;
      *reinterpret_cast<f32x4w*>(base + k * WLDB + 4 * n4) = src[p];
    }
  };
  int64_t cur_blk = -1;
  if (u0 < u1) {
    fetch(pre[0], u0, 0);
    stash(pre[0], 0);
    fetch(pre[1], u0, 1);
    fetch(pre[0], u0 + 1 < u1 ? u0 + 1 : u0, 0);
  }
  for (int64_t u = u0; u < u1; ++u) {
    const int64_t blk = u / ntile, m0 = blk * WBM;
    const int t = (int)(u - blk * ntile);
    const int rows = (int)(M - m0 < WBM ? M - m0 : WBM);
    if (blk != cur_blk) {                   // (re)load the A block: 4096 16-B chunks, 8 per thread
      __syncthreads();
      const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A) + m0 * sam, 0,
                                                                           ((rows - 1) * sam + WK) * 4, 0x00020000);
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int P = p * 512 + tid, row = P >> 5, c = P & 31;
        const unsigned off = row < rows ? (unsigned)(row * sam + 4 * c) * 4u : 0xfffffff0u;      // rows past M: zeros
        const f32x4w v = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
        *reinterpret_cast<f32x4w*>(As + row * WK + 4 * ((c & ~7) | ((c & 7) ^ ((row >> 1) & 7)))) = v;
      }
      cur_blk = blk;
    }
    f32x4w acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4w{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      __syncthreads();
      if (h == 0) {
        stash(pre[1], 1);                                   // (u, 1): requested at the top of (u-1, 1)
      } else {
        stash(pre[0], 0);                                   // (u+1, 0): requested inside (u-1, 1)
        fetch(pre[1], u + 1 < u1 ? u + 1 : u, 1);           // (a clamped unit past the end is never used)
      }
      const float* Bh = Bs + h * (64 * WLDB);
      f32x4w af[2][2], bf[2][4];
      auto frags = [&](int kg, int slot) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int row = (wm * 2 + a) * 16 + j, c = 16 * h + 4 * kg + q;
          af[slot][a] = *reinterpret_cast<const f32x4w*>(As + row * WK + 4 * ((c & ~7) | ((c & 7) ^ ((row >> 1) & 7))));
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float* pb = Bh + (16 * kg + 4 * q) * WLDB + (wn * 4 + b) * 16 + j;
#pragma unroll
          for (int r = 0; r < 4; ++r) bf[slot][b][r] = pb[r * WLDB];
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
        if (h == 1 && kg == 0) {
          fetch(pre[0], u + 2 < u1 ? u + 2 : u, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) asm volatile("" : "+v"(pre[0][p]), "+v"(pre[1][p]));     // both requests complete HERE
    __builtin_amdgcn_sched_barrier(0);
    const __amdgpu_buffer_rsrc_t rsC =
        __builtin_amdgcn_make_buffer_rsrc(g.C + m0 * scm, 0, ((rows - 1) * scm + N) * 4, 0x00020000);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int n = WBN * t + (wn * 4 + b) * 16 + j;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (wm * 2 + a) * 16 + 4 * q + r;
          const unsigned off = (n < N && row < rows) ? (unsigned)(row * scm + n) * 4u : 0xfffffff0u;
          const float val = acc[a][b][r];      // (a scalar first: bit-casting the vector element stored element 0 four times)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rsC, off, 0, 0);
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Tall-skinny product with both operands k-contiguous, N <= 128, K % 32 == 0, bias / ReLU epilogue only: the fc1 forward
// of the MNISTCNN embedding (78 400 x 128 x 2304, models/MLP.py:44).  One 8-wavefront workgroup per CU owns a
// (64 TMW) x 128 row block (TMW = 5: 245 blocks for 78 400 rows); K-slabs of 32 go global -> registers under the MFMAs of the
// current slab, registers -> LDS (one ds_write_b128 per 16-B piece) behind them, two LDS stages, ONE workgroup barrier per
// slab; fragments are ds_read_b128 (4 k per lane, chunk position XOR-swizzled by the row: conflict-free).  Wavefront
// (wm, wn) of the 4 x 2 grid holds TMW x 4 accumulator tiles: every A fragment feeds 4 MFMAs, every B fragment TMW.
// Main loop 87 % of the MFMA-bound cycle count; 0.368 ms = 126 TFLOP/s against 0.427 ms of gemm_vec_k<160,128> on a warm
// GPU (tools/tall_gemm.hip is the stand-alone form; the round-3 "no gain" was measured on a GPU that had not ramped its
// clocks up: 10 launches right behind a 700 MB host copy).
// ---------------------------------------------------------------------------------------------
// WR = 4 (shipped): one 8-wavefront workgroup per CU with (64 TMW)-row blocks.  WR = 2 (GNF_GEMM_TALL=2): TWO 4-wavefront
// workgroups per CU with (32 TMW)-row blocks -- their barriers are independent, so while one waits at its slab boundary the
// other feeds the MFMA pipe: 0.381 -> 0.364 ms in the stand-alone form (tools/tall_gemm.hip -DWROWS=2), but once the loads
// of this kernel were lean (below) the two forms measure the same here: 0.368 / 0.373 ms, 6.55-6.57 ms per step either way.
constexpr int TBK = 32, TBN = 128;
template <int TMW, int WR>
constexpr size_t tall_lds() { return 2 * (size_t)(16 * WR * TMW + TBN) * TBK * sizeof(float); }

template <int TMW, int WR>
__global__ __launch_bounds__(128 * WR, 4 / WR) void gemm_tall_k(GemmArgs g) {
  constexpr int BM = 16 * WR * TMW;             // WR wavefront rows x TMW tiles x 16
  constexpr int NT = 128 * WR;                  // threads
  constexpr int SLAB = (BM + TBN) * TBK;        // floats per stage
  extern __shared__ __attribute__((aligned(16))) float tsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t M = g.M, K = g.K;
  const int N = (int)g.N;
  const int64_t ntiles = (M + BM - 1) / BM;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t m0 = tile * BM;
    f32x4w acc[TMW][4];
#pragma unroll
    for (int a = 0; a < TMW; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4w{0.f, 0.f, 0.f, 0.f};
    // this thread's 16-B pieces of a slab: piece p covers position P = p * NT + tid -> row P / 8, chunk position P % 8,
    // holding the row's logical chunk (P % 8) ^ ((row >> 1) & 7).  The first PA pieces are rows of A, the rest rows of B
    // (BM * 8 is a multiple of NT).  One buffer descriptor per operand (A: per row block), a 32-bit lane offset that does
    // not depend on the slab (its k offset rides in the scalar offset of the load): nothing is computed per slab, rows past
    // M / N read as zeros.  (With 64-bit pointers per piece the two-workgroup form carried 22 v_lshl_add_u64 and 85 v_mov
    // per slab and wavefront: 0.404 ms against 0.363 of the stand-alone form.)
    constexpr int PIECES = (BM + TBN) * 8 / NT, PA = BM * 8 / NT;
    static_assert((BM + TBN) * 8 % NT == 0 && BM * 8 % NT == 0, "whole pieces per thread, A and B pieces apart");
    const int rows = (int)(M - m0 < BM ? M - m0 : BM);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A) + m0 * g.sam, 0,
                                                                         (int)(((int64_t)(rows - 1) * g.sam + K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B), 0,
                                                                         (int)(((int64_t)(N - 1) * g.sbn + K) * 4), 0x00020000);
    unsigned voff[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const int P = p * NT + tid;
      const int row = P >> 3, c = (P & 7) ^ ((row >> 1) & 7);
      if (p < PA) voff[p] = row < rows ? (unsigned)(row * (int)g.sam + 4 * c) * 4u : 0xfffffff0u;
      else voff[p] = row - BM < N ? (unsigned)((row - BM) * (int)g.sbn + 4 * c) * 4u : 0xfffffff0u;
    }
    f32x4w pre[PIECES];
    auto fetch = [&](int k0) {
#pragma unroll
      for (int p = 0; p < PIECES; ++p)
        pre[p] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(p < PA ? rsA : rsB, voff[p], k0 * 4, 0));
    };
    auto stash = [&](int stage) {
      float* base = tsm + stage * SLAB + tid * 4;
#pragma unroll
      for (int p = 0; p < PIECES; ++p) *reinterpret_cast<f32x4w*>(base + p * NT * 4) = pre[p];
    };
    fetch(0);
    stash(0);
    const int nslab = (int)(K / TBK);
    for (int s = 0; s < nslab; ++s) {
      __syncthreads();                             // slab s visible; everybody done reading the other stage
      if (s + 1 < nslab) fetch((s + 1) * TBK);
      const float* As = tsm + (s & 1) * SLAB;
      const float* Bs = As + BM * TBK;
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        f32x4w af[TMW], bf[4];
#pragma unroll
        for (int a = 0; a < TMW; ++a) {
          const int row = (wm * TMW + a) * 16 + j;
          af[a] = *reinterpret_cast<const f32x4w*>(As + row * TBK + 4 * ((4 * kg + q) ^ ((row >> 1) & 7)));
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int row = (wn * 4 + b) * 16 + j;
          bf[b] = *reinterpret_cast<const f32x4w*>(Bs + row * TBK + 4 * ((4 * kg + q) ^ ((row >> 1) & 7)));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int a = 0; a < TMW; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][r], bf[b][r], acc[a][b], 0, 0, 0);
      }
      if (s + 1 < nslab) stash((s + 1) & 1);
    }
    __syncthreads();                               // all reads of the last slab done before the next block's first stash
    // D layout of a 16x16 tile: lane (q, j) holds rows 4q..4q+3 of the A tile, column j of the B tile
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int n = (wn * 4 + b) * 16 + j;
      const float bv = (g.bias && n < N) ? g.bias[n] : 0.f;
#pragma unroll
      for (int a = 0; a < TMW; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t m = m0 + (wm * TMW + a) * 16 + 4 * q + r;
          if (m < M && n < N) {
            float v = acc[a][b][r] + bv;
            if (g.flags & GNF_GEMM_RELU) v = fmaxf(v, 0.f);
            g.C[m * g.scm + n * g.scn] = v;
          }
        }
    }
  }
}

// the tile height (TMW) that wastes the least of the chip on a tall M, 0 when every choice idles > 10 % of it
static int tall_tmw(int64_t M, int wr) {
  int best = 0;
  double best_waste = .10;
  const int64_t slots = 256 * (4 / wr);
  for (int tmw = 5; tmw >= 3; --tmw) {
    const int64_t bm = 16 * wr * tmw, tiles = (M + bm - 1) / bm, rounds = (tiles + slots - 1) / slots;
    const double waste = 1. - (double)M / (double)(rounds * slots * bm);
    if (waste < best_waste) { best_waste = waste; best = tmw; }
  }
  return best;
}

// ---------------------------------------------------------------------------------------------
// Split-K product with BOTH operands k-major, M <= 128:  C[M x N] = A[K x M]^T B[K x N]  -- the weight gradient of a
// <= 128-wide layer over a tall batch (MNISTCNN fc1: dW = dY^T X, 128 x 2304 x 78 400, models/MLP.py:44 backward).  One
// 8-wavefront workgroup owns a 128 x 128 output tile over one K range (blockIdx = tile + ntile * split: 18 x 14 = 252
// workgroups for fc1); K-slabs of 64 rows go through two LDS stages IN THEIR GLOBAL LAYOUT ([k][128], 512-B rows: nothing is
// transposed on the way in).  Fragments: ONE ds_read_b128 at [k = 4 kk + q][m = 4 j .. 4 j + 3] is the A operand of FOUR MFMAs
// -- row i of tile t is m = 4 i + t (any row permutation is as good as another as long as the epilogue knows it), the B
// side likewise with two tiles per ds_read_b64; wavefront (wm, wn) of a 2 x 4 grid holds 64 m x 32 n.  Buffer descriptors
// per K range: rows past the range read as zeros, columns past N are neither read nor stored.  Partials go to
// C + split * c_split_stride as float2 (the caller reduces them).  0.372 + 0.006 ms against 0.417 + 0.013 of
// gemm_vec_k<128,128> with 36 splits (tools/kmajor_gemm.hip is the stand-alone form).
// ---------------------------------------------------------------------------------------------
constexpr int KMK = 64, KMN = 128, KMLDS = 2 * KMK * (128 + KMN) * (int)sizeof(float);

__global__ __launch_bounds__(512, 1) void gemm_kmajor_k(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float ksm[];
  typedef float f32x2w __attribute__((ext_vector_type(2)));
  constexpr int SLAB = KMK * (128 + KMN);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int wm = wave >> 2, wn = wave & 3;
  const int M = (int)g.M, N = (int)g.N, sak = (int)g.sak, sbk = (int)g.sbk;
  const int ntile = (N + KMN - 1) / KMN;
  const int tile = blockIdx.x % ntile, split = blockIdx.x / ntile;
  const int64_t k0 = (int64_t)split * g.k_per_split;
  if (k0 >= g.K) return;
  const int kn = (int)(g.K - k0 < g.k_per_split ? g.K - k0 : g.k_per_split);
  const int n0 = tile * KMN;
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A) + k0 * sak, 0, ((kn - 1) * sak + M) * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.B) + k0 * sbk, 0, ((kn - 1) * sbk + N) * 4, 0x00020000);
  // 16-B pieces of a slab: 64 rows x 32 quads per operand, 4 + 4 per thread
  f32x4w pa[4], pb[4];
  auto fetch = [&](int s) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int P = p * 512 + tid, row = s * KMK + (P >> 5), c = P & 31;
      pa[p] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(
                                             rsA, 4 * c < M ? (unsigned)(row * sak + 4 * c) * 4u : 0xfffffff0u, 0, 0));
      const int gn = n0 + 4 * c;
      pb[p] = __builtin_bit_cast(f32x4w, __builtin_amdgcn_raw_buffer_load_b128(
                                             rsB, gn < N ? (unsigned)(row * sbk + gn) * 4u : 0xfffffff0u, 0, 0));
    }
  };
  auto stash = [&](int stage) {
    float* a = ksm + stage * SLAB;
    float* b = a + KMK * 128;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int P = p * 512 + tid, row = P >> 5, c = P & 31;
      *reinterpret_cast<f32x4w*>(a + row * 128 + 4 * c) = pa[p];
      *reinterpret_cast<f32x4w*>(b + row * KMN + 4 * c) = pb[p];
    }
  };
  f32x4w acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = f32x4w{0.f, 0.f, 0.f, 0.f};
  const int nslab = (kn + KMK - 1) / KMK;
  fetch(0);
  stash(0);
  for (int s = 0; s < nslab; ++s) {
    __syncthreads();                               // slab s visible; everybody done reading the other stage
    if (s + 1 < nslab) fetch(s + 1);
    const float* As = ksm + (s & 1) * SLAB;
    const float* Bs = As + KMK * 128;
#pragma unroll
    for (int kk = 0; kk < KMK / 4; ++kk) {
      const f32x4w af = *reinterpret_cast<const f32x4w*>(As + (4 * kk + q) * 128 + 64 * wm + 4 * j);
      const f32x2w bf = *reinterpret_cast<const f32x2w*>(Bs + (4 * kk + q) * KMN + 32 * wn + 2 * j);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        acc[a][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[0], acc[a][0], 0, 0, 0);
        acc[a][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[1], acc[a][1], 0, 0, 0);
      }
    }
    if (s + 1 < nslab) stash((s + 1) & 1);
  }
  // D tile (a, b): lane (q, j) holds rows i = 4 q + r -> m = 64 wm + 4 i + a, column j -> n = 32 wn + 2 j + b
  float* out = g.C + (int64_t)split * g.c_split_stride;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 64 * wm + 4 * (4 * q + r) + a, n = n0 + 32 * wn + 2 * j;
      if (m < M && n < N) *reinterpret_cast<f32x2w*>(out + (int64_t)m * g.scm + n) = f32x2w{acc[a][0][r], acc[a][1][r]};
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
  int64_t kps = ((K + splits - 1) / splits + BKV - 1) / BKV * BKV;
  return kps < BKV ? BKV : kps;
}

int64_t gnf_gemm_num_splits(int64_t K, int splits) {
  const int64_t kps = k_per_split(K, splits);
  return K > 0 ? (K + kps - 1) / kps : 1;
}

// which kernel the most recent gnf_gemm_launch of THIS thread dispatched to (measurement / tests: tools/bench_kernels.py names
// the kernel behind each row, the dedicated-kernel tests check that their shapes reach it); no effect on any result
static thread_local const char* g_last_kernel = "";
extern "C" const char* gnf_gemm_last_kernel(void) { return g_last_kernel; }

// Internal launcher shared with gnf_monotonic.hip (split-K weight-gradient GEMMs).
int gnf_gemm_launch(GemmArgs g, int splits, hipStream_t s) {
  g.k_per_split = k_per_split(g.K, splits);
  // an accumulating launch must cover every partial written by the first one: exactly `splits` of them
  const int64_t nsp = (g.flags & GNF_GEMM_ACCUM) ? splits : gnf_gemm_num_splits(g.K, splits);
  // 128x128 tiles once there are >= 2 workgroups per CU of them, unless 64x64 tiles waste much less padding
  // (e.g. the 160x160 weight gradients of the wide integrand nets: 192^2 vs 256^2)
  const int64_t t128 = ((g.M + 127) / 128) * ((g.N + 127) / 128), t64 = ((g.M + 63) / 64) * ((g.N + 63) / 64);
  const bool pad_heavy = t128 * 4 * 4 > t64 * 5;            // 128-tiling computes > 1.25x the 64-tiling's area
  const int bt = (t128 * nsp >= 512 && !pad_heavy) ? 128 : 64;
  const int64_t gx = (g.M + bt - 1) / bt, gy = (g.N + bt - 1) / bt;
  if (gy > 65535 || nsp > 65535) return GNF_ESHAPE;
  const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)nsp);
  // both operands k-major, M <= 128, a long K per split, plain partial / plain output (fc1 weight gradient)
  if (g.sam == 1 && g.sbn == 1 && g.scn == 1 && g.M <= 128 && g.M % 4 == 0 && !g.grp && !g.bias && !g.Bmask && !g.Cmask && !g.gate &&
      g.flags == 0 && g.N % 4 == 0 && g.sak % 4 == 0 && g.sbk % 4 == 0 && g.scm % 2 == 0 && g.k_per_split >= 512 &&
      g.sak >= g.M && g.sbk >= g.N &&       // rows past a K range must lie past the descriptor's extent (they read as zeros)
      (((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0 && (((uintptr_t)g.C | (uintptr_t)(g.c_split_stride * 4)) & 7) == 0 &&
      g.k_per_split * (g.sak > g.sbk ? g.sak : g.sbk) < (1 << 28) && ((g.N + KMN - 1) / KMN) * nsp >= 64 && nsp <= 4096) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kmajor_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              KMLDS);
    hipLaunchKernelGGL(gemm_kmajor_k, dim3((unsigned)(((g.N + KMN - 1) / KMN) * nsp)), dim3(512), KMLDS, s, g);
    g_last_kernel = "gemm_kmajor_k";
    GNF_LAUNCH_CHECK();
    return 0;
  }
  // short K, wide N, tall M, no epilogue options (fc1 data gradient): the persistent unit-range kernel
  if (g.K == WK && nsp == 1 && !g.grp && g.sak == 1 && g.sbn == 1 && g.scn == 1 && !g.bias && !g.Bmask && !g.Cmask &&
      !g.gate && g.flags == 0 && g.N % 4 == 0 && g.N >= 4 * WBN && g.M >= 32 * WBM &&
      // 32-bit lane offsets inside a row block / inside B, 16-B aligned quads
      g.sam % 4 == 0 && g.sbk % 4 == 0 && (((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0 && g.N < (1 << 24) &&
      (int64_t)WBM * g.sam < (1 << 28) && (int64_t)WBM * g.scm < (1 << 28) && (int64_t)WK * g.sbk < (1 << 28)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wide_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kWideLds);
    hipLaunchKernelGGL(gemm_wide_k, dim3(256), dim3(64 * WWAVES), kWideLds, s, g);
    g_last_kernel = "gemm_wide_k";
    GNF_LAUNCH_CHECK();
    return 0;
  }
  // vector path: A and B each have a unit-stride dimension (any leading stride, dword alignment)
  const bool akf = g.sak == 1, amf = !akf && g.sam == 1;
  const bool bkf = g.sbk == 1, bnf = !bkf && g.sbn == 1;
  const bool vec = (akf || amf) && (bkf || bnf) && g.k_per_split % 4 == 0;
#define GNF_VEC_LAUNCH(BT)                                                                              \
  do {                                                                                                  \
    if (akf && bkf) hipLaunchKernelGGL((gemm_vec_k<BT, BT, true, true>), grid, dim3(256), 0, s, g);      \
    else if (akf) hipLaunchKernelGGL((gemm_vec_k<BT, BT, true, false>), grid, dim3(256), 0, s, g);       \
    else if (bkf) hipLaunchKernelGGL((gemm_vec_k<BT, BT, false, true>), grid, dim3(256), 0, s, g);       \
    else hipLaunchKernelGGL((gemm_vec_k<BT, BT, false, false>), grid, dim3(256), 0, s, g);               \
  } while (0)
  // one column of <= 128-wide tiles over a tall M, long K (fc1 forward: 78 400 x 128 x 2304): the persistent tall kernel when
  // one of its block heights fills the chip
  // (decided on N itself, not on the generic kernels' tile choice: that takes 64 x 64 tiles below 512 tiles of 128 x 128, and
  // with it this branch was skipped for every M < 65 536 -- found in round 5 by asserting gnf_gemm_last_kernel in the tests)
  if (vec && akf && bkf && g.N <= 128 && nsp == 1 && !g.grp && !g.Bmask && !g.Cmask && !g.gate &&
      !(g.flags & ~GNF_GEMM_RELU) && g.N > 96 && g.K % TBK == 0 && g.K >= 8 * TBK && g.sam % 4 == 0 && g.sbn % 4 == 0 &&
      (((uintptr_t)g.A | (uintptr_t)g.B) & 15) == 0 && 320 * g.sam + g.K < (1 << 28) && 128 * g.sbn + g.K < (1 << 28)) {
    const int wr = 4;                               // (the two-group form, 2 x 4 wavefronts per CU, lost its A/B in round 4)
    const int tmw = tall_tmw(g.M, wr);
    if (tmw) {
      const int64_t bm = 16 * wr * tmw, tiles = (g.M + bm - 1) / bm, slots = 256 * (4 / wr);
      const unsigned tgrid = (unsigned)(tiles < slots ? tiles : slots);
#define GNF_TALL_LAUNCH(T, W)                                                                                     \
  do {                                                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tall_k<T, W>),                                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)tall_lds<T, W>());                  \
    hipLaunchKernelGGL((gemm_tall_k<T, W>), dim3(tgrid), dim3(128 * W), (tall_lds<T, W>()), s, g);                 \
  } while (0)
      if (tmw == 5) GNF_TALL_LAUNCH(5, 4); else if (tmw == 4) GNF_TALL_LAUNCH(4, 4); else GNF_TALL_LAUNCH(3, 4);
#undef GNF_TALL_LAUNCH
      g_last_kernel = "gemm_tall_k";
      GNF_LAUNCH_CHECK();
      return 0;
    }
  }
  // otherwise pick the tile height that leaves the fewest tile-rows on the busiest CU
  if (vec && akf && bkf && bt == 128 && gy == 1 && nsp == 1 && !g.grp) {
    const int64_t t160 = (g.M + 159) / 160;
    if (((t160 + 255) / 256) * 160 < ((gx + 255) / 256) * 128) {
      hipLaunchKernelGGL((gemm_vec_k<160, 128, true, true, 1>), dim3((unsigned)t160, 1, 1), dim3(256), 0, s, g);
      g_last_kernel = "gemm_vec_k<160,128>";
      GNF_LAUNCH_CHECK();
      return 0;
    }
  }
  if (vec) {
    if (bt == 128) GNF_VEC_LAUNCH(128); else GNF_VEC_LAUNCH(64);
    g_last_kernel = bt == 128 ? "gemm_vec_k<128,128>" : "gemm_vec_k<64,64>";
  } else {
    if (bt == 128) hipLaunchKernelGGL((gemm_k<128, 128>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_k<64, 64>), grid, dim3(256), 0, s, g);
    g_last_kernel = bt == 128 ? "gemm_k<128,128>" : "gemm_k<64,64>";
  }
#undef GNF_VEC_LAUNCH
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_gemm_grouped_launch(GemmArgs g, int ngroups, hipStream_t s) {
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  const bool akf = g.sak == 1 && g.sam % 4 == 0, amf = g.sam == 1 && g.sak % 4 == 0;
  const bool bkf = g.sbk == 1 && g.sbn % 4 == 0, bnf = g.sbn == 1 && g.sbk % 4 == 0;
  if (!g.grp || ngroups < 1 || ngroups > 65535 || !(akf || amf) || !(bkf || bnf) || !al16(g.A) || !al16(g.B) ||
      g.b_grp_stride % 4 || g.Bmask || g.Cmask || g.gate)
    return GNF_ESHAPE;
  if (g.grp_k) {
    if (akf || bkf) return GNF_ESHAPE;       // K ranges start anywhere: K must be the strided dimension of both operands
    g.k_per_split = BKV;                     // unused (the table gives the range)
  } else {
    if (g.K % 4) return GNF_ESHAPE;
    g.k_per_split = (g.K + BKV - 1) / BKV * BKV;
    g.c_split_stride = 0;
  }
  // 128 x 128 tiles only where they tile the problem without padding: the column count (row groups: the row counts sit in the
  // device table) or both extents (K groups) a multiple of 128.  The sparse front's d pd = g Wg^T (N = 400) and dWg = pd^T g
  // (M = 400) pad 400 to 512 with 128-wide tiles and to 448 with 64-wide ones, on top of every group's own ragged last row
  // tile: 2.683 -> 2.635 ms per frozen-gate step (profiles/r05_sparse_bwd2.txt); the forward (N = 128) keeps the large tiles.
  const int bt = g.grp_k ? ((g.M % 128 == 0 && g.N % 128 == 0) ? 128 : 64) : ((g.M >= 1024 && g.N % 128 == 0) ? 128 : 64);
  const dim3 grid((unsigned)((g.M + bt - 1) / bt), (unsigned)((g.N + bt - 1) / bt), (unsigned)ngroups);
#define GNF_GRP_LAUNCH(BT)                                                                              \
  do {                                                                                                  \
    if (akf && bkf) hipLaunchKernelGGL((gemm_vec_k<BT, BT, true, true>), grid, dim3(256), 0, s, g);      \
    else if (akf) hipLaunchKernelGGL((gemm_vec_k<BT, BT, true, false>), grid, dim3(256), 0, s, g);       \
    else if (bkf) hipLaunchKernelGGL((gemm_vec_k<BT, BT, false, true>), grid, dim3(256), 0, s, g);       \
    else hipLaunchKernelGGL((gemm_vec_k<BT, BT, false, false>), grid, dim3(256), 0, s, g);               \
  } while (0)
  if (bt == 128) GNF_GRP_LAUNCH(128); else GNF_GRP_LAUNCH(64);
#undef GNF_GRP_LAUNCH
  GNF_LAUNCH_CHECK();
  return 0;
}

// split-K plan of the public entry: few output tiles and a long K -> spread K over the chip
static int plan_splits(int64_t M, int64_t N, int64_t K, bool kmajor_ok = true) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;  // empty operands (a zero-row batch): nothing to split
  // Fewer 64x64 output tiles than ~3/4 of the CUs: split K until the chip is covered, at least 128 of K per split.
  // (The MADE layers at B = 100 are 2 x 16 tiles with K = 1024: unsplit they ran on 32 of the 256 CUs.)
  const int64_t tiles = ((M + 63) / 64) * ((N + 63) / 64);
  // Long-K weight-gradient shapes with a small output (fc1 dW: 128 x 2304 x 78400): 128x128 tiles read the tall
  // operands half as often as 64x64 ones (the launch is L2/HBM-bound on them); enough splits to fill the chip.
  const int64_t t128 = ((M + 127) / 128) * ((N + 127) / 128);
  // (t128 up to 128: the 1890 x 630 x 50 000 weight gradient of cfg5's last MADE layer is 75 such tiles -- as 300 unsplit
  // 64 x 64 tiles it streamed 7.7 GB of operands in 2.63 ms, 45 TFLOP/s.)
  // M <= 128 over a very long K (fc1 weight gradient 128 x 2304 x 78 400): gemm_kmajor_k runs ONE 128-KB-LDS workgroup per
  // CU, so as many (tile, split) pairs as CUs: 18 tiles x 14 splits (the generic kernels took 36 splits: 3 x the partials)
  if (M <= 128 && N % 4 == 0 && N >= 512 && K >= 16384 && kmajor_ok) {
    int64_t s = 256 / ((N + 127) / 128);
    if (s > K / 512) s = K / 512;
    if (s >= 2) return (int)s;
  }
  if (K >= 8192 && t128 <= 128 && t128 * 16 <= tiles * 5) {
    int64_t s128 = (640 + t128 - 1) / t128;
    if (s128 > K / 512) s128 = K / 512;
    if (s128 >= 2) return (int)(s128 > 512 ? 512 : s128);
  }
  if (K < 512 || tiles >= (K < 2048 ? 192 : 256)) return 1;
  int64_t s = ((K < 2048 ? 512 : 768) + tiles - 1) / tiles;
  const int64_t kmin = K < 2048 ? 128 : 512;
  if (s > K / kmin) s = K / kmin;
  if (s > 512) s = 512;
  return s < 2 ? 1 : (int)s;
}

// (sized without knowing the strides: the larger of the k-major plan and the generic one)
// what the fp32-MFMA kernels of this file want for their split-K partials (measurement / tests: a workspace of exactly this size
// keeps a call on the fp32 kernels where gnf_gemm_ws_bytes would also admit the split-bf16 ones)
extern "C" int64_t gnf_gemm_f32_ws_bytes(int64_t M, int64_t N, int64_t K) {
  const int a = plan_splits(M, N, K, true), b = plan_splits(M, N, K, false);
  const int s = a > b ? a : b;
  return s > 1 ? (int64_t)(s + 1) * M * N * (int64_t)sizeof(float) : 0;     // partials + one row for their sum
}

extern "C" int64_t gnf_gemm_ws_bytes(int64_t M, int64_t N, int64_t K) {
  const int64_t w = gnf_gemm_f32_ws_bytes(M, N, K);
  // the split-bf16 kernels' pre-split planes of the small operand (round 6), when the switch is on
  const int64_t w2 = gnf_gemm_split_enabled() ? gnf_gemm_split_ws_bytes(M, N, K) : 0;
  return w > w2 ? w : w2;
}

extern "C" int gnf_gemm(const float* A, int64_t sam, int64_t sak, const float* B, const float* Bmask, int64_t sbk,
                        int64_t sbn, float* C, int64_t scm, int64_t scn, const float* bias, const float* Cmask,
                        int64_t scmm, int64_t scmn, const float* gate, int64_t sgm, int64_t sgn, int flags,
                        int64_t M, int64_t N, int64_t K, float* ws, int64_t ws_bytes, gnf_stream_t stream) {
  if (M < 0 || N < 0 || K < 0) return GNF_EINVAL;
  if (M == 0 || N == 0) return 0;          // nothing to write; operands may be NULL
  if (!C || ((!A || !B) && K > 0)) return GNF_EINVAL;   // K == 0 (an empty batch as the contraction): C = epilogue(0)
  // Round 6: the three fc1-shaped products (tall M x <= 128 x long K; tall M x wide N x 128) run on the bf16 matrix pipe with
  // fp32 accuracy (gnf_gemm_split.hip: exact three-way operand splits, six cross terms, fp32 accumulate; measured against
  // fp64: more accurate than the fp32-MFMA kernels below on every case, profiles/r06_split_bf16_error.txt).
  // GNF_TRUE_F32=1 keeps everything on v_mfma_f32_*.
  if (K > 0 && !Bmask && !Cmask && !gate) {
    const int rc = gnf_gemm_split_try(A, sam, sak, B, sbk, sbn, C, scm, scn, bias, (flags & GNF_GEMM_RELU) ? 1 : 0, M, N, K, ws,
                                      ws_bytes, (hipStream_t)stream);
    if (rc == 0) { g_last_kernel = gnf_gemm_split_last_kernel(); return 0; }
    if (rc != 1) return rc;
  }
  GemmArgs g{A, sam, sak, B, Bmask, sbk, sbn, C, scm, scn, bias, Cmask, scmm, scmn, gate, sgm, sgn,
             flags & GNF_GEMM_RELU, M, N, K, 0, 0};
  // the few long splits of the k-major plan only for operands gemm_kmajor_k takes (the launcher's predicate on strides and
  // alignment); anything else gets the generic kernels' tuned count
  const bool km_ok = sam == 1 && sbn == 1 && scn == 1 && M % 4 == 0 && sak % 4 == 0 && sbk % 4 == 0 && sak >= M && sbk >= N &&
                     !bias && !Bmask && !Cmask && !gate && !(flags & GNF_GEMM_RELU) &&
                     (((uintptr_t)A | (uintptr_t)B) & 15) == 0;
  const int splits = plan_splits(M, N, K, km_ok);
  if (splits > 1 && ws && ws_bytes >= gnf_gemm_f32_ws_bytes(M, N, K)) {
    GemmArgs p = g;                       // partial products only: epilogue runs in the reduction
    p.C = ws; p.scm = N; p.scn = 1; p.c_split_stride = M * N;
    p.bias = nullptr; p.Cmask = nullptr; p.gate = nullptr; p.flags = 0;
    int rc = gnf_gemm_launch(p, splits, (hipStream_t)stream);
    if (rc) return rc;
    int64_t nsp = gnf_gemm_num_splits(K, splits);
    const float* part = ws;
    if (M * N < 16384 && nsp > 8) {
      // small output, many partials (the 60 x 60 weight gradients of the POWER embedding MLP over 60 000 rows): one
      // thread per output walking the partials serially is latency-bound; sum them with the parallel row-sum kernel first
      float* sum = ws + nsp * M * N;
      rc = gnf_rowsum_launch(ws, sum, nsp, M * N, 0, (hipStream_t)stream);
      if (rc) return rc;
      part = sum;
      nsp = 1;
    }
    int64_t grid = (M * N + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(gemm_reduce_k, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, part, nsp, g);
    GNF_LAUNCH_CHECK();
    return 0;
  }
  return gnf_gemm_launch(g, 1, (hipStream_t)stream);
}
