// Linear / masked-Linear layers of the conditioner MLPs at SMALL batch (M <= 128 rows): MADE's masked linears
// (models/Conditionners/AutoregressiveConditioner.py:14-25,85-96: `F.linear(x, mask * W, b)`), CouplingMLP / DAGMLP
// (CouplingConditioner.py:6-19, DAGConditioner.py:7-20) and their autograd.
//
// BASELINE cfg3 (MNIST, B = 100, MADE 784 -> 1024^3 -> 1568) is the case: 2 x 100 x 1024 x 1024 flop against 4.2 MB of
// weights per layer -- 48 flop per weight byte, weight-streaming bound.  The tiled GEMM (gnf_gemm.hip) stages 64 x 64
// tiles through LDS slab by slab and read the fp32 MASK as a second 4.2 MB stream: 15-20 us per layer where the bytes
// need ~1 us.  Here
//  * the mask is never read when it is degree-structured (MADE's is by construction): mask[o][i] = deg_in[i] <= deg_out[o]
//    (< for the output layer) is evaluated on two small vectors while the weight fragment is in flight;
//  * nothing goes through LDS on the way in: every operand is loaded straight into its MFMA register layout
//    (v_mfma_f32_16x16x4_f32, exact fp32).  With k visited in the order the registers hold it (K-step r of a 16-wide
//    chunk pairs lane slot q with k = 16 c + 4 q + r) one dwordx4 per lane feeds four K-steps;
//  * the chip is filled by (16-wide output tile) x (16-row batch tile) workgroups -- 448 of them at cfg3 -- whose four
//    wavefronts split the contraction and meet in 4 KB of LDS; outputs are final (bias / ReLU / ReLU gate / mask fused):
//    no split-K partials in HBM, no reduction launch.
//      forward   y  = act(x (W o mask)^T + b)                     lin_fwd_skinny_k    grid (N/16, M/16)
//      data grad gx = (g (W o mask)) o [a > 0]                    lin_bwdx_skinny_k   grid (K/32, M/16)
//      weight grad gW = (g^T a) o mask,  gb = colsum g            lin_bwdw_skinny_k   grid (N/64, K/64)
// Everything else (M > 128, K not a multiple of 16) goes to the tiled GEMM unchanged.
#include "gnf_common.h"
#include "gnf_gemm.h"
#include "gnf_linear_tall.h"
#include <cstdlib>

extern "C" int64_t gnf_gemm_ws_bytes(int64_t M, int64_t N, int64_t K);
extern "C" int gnf_gemm(const float* A, int64_t sam, int64_t sak, const float* B, const float* Bmask, int64_t sbk,
                        int64_t sbn, float* C, int64_t scm, int64_t scn, const float* bias, const float* Cmask,
                        int64_t scmm, int64_t scmn, const float* gate, int64_t sgm, int64_t sgn, int flags, int64_t M,
                        int64_t N, int64_t K, float* ws, int64_t ws_bytes, gnf_stream_t stream);
extern "C" int64_t gnf_colsum_ws_bytes(int64_t M, int64_t N);
extern "C" int gnf_colsum(const float* a, int64_t lda, float* out, int64_t M, int64_t N, float* ws, gnf_stream_t stream);

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // dwordx4 at any dword address
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// mask(o, i) of a layer with weight [N out][K in]: the full 0/1 tensor (row-major [N][K]) or, when drow != NULL, the
// degree rule dcol[i] <= drow[o] (strict: <).  Neither: no mask.
struct LinMask {
  const float* full;
  const float* drow;
  const float* dcol;
  int strict;
};
__device__ __forceinline__ float deg_keep(float dc, float dr, int strict) { return (strict ? dc < dr : dc <= dr) ? 1.f : 0.f; }

// mask kind of a kernel instantiation
constexpr int MK_NONE = 0, MK_FULL = 1, MK_DEG = 2;

// ------------------------------------------------------------------------------------------------------------- forward
// Workgroup (nt, mp): y[16 TM mp .. +16 TM, 16 nt .. +16] over the whole K -- TM batch tiles per workgroup, so that a
// weight fragment feeds TM MFMAs and the weight stream through the CU's L1 return path (what bounds these kernels:
// 64 B/clk, every workgroup pulls its own operand slabs from L2) shrinks by TM.  A 16-wavefront workgroup is alone on its
// CU: TM is chosen so that the grid is one round of the 256 CUs.  Wavefront w of NW contracts the 16-wide chunks [c0, c1)
// of its share; CB chunks are requested in one round -- with K / (16 NW) <= CB the whole kernel is ONE memory latency plus
// 8 CB MFMAs per wavefront.
// Degree masks: the 16 deg_in values of a chunk are wave-uniform -> SCALAR loads (s_load_dwordx16 through the constant
// cache, no vector-memory traffic at all; as a dwordx4 per lane this stream cost as much as the weights), the lane's four
// (slot q) picked with three selects each.
template <int NW, int MK, int CB, int TM>
__global__ __launch_bounds__(64 * NW) void lin_fwd_skinny_k(const float* __restrict__ x, const float* __restrict__ W, LinMask mk,
                                                            const float* __restrict__ bias, int relu, float* __restrict__ y,
                                                            int M, int N, int K) {
  __shared__ f32x4 red[NW][TM][64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int nt = blockIdx.x, mp = blockIdx.y;
  const int nchunks = K / 16;
  const int c0 = nchunks * wave / NW, c1 = nchunks * (wave + 1) / NW;
  const int n = 16 * nt + j;
  const bool nok = n < N;
  const float* xr[TM];
  bool mok[TM];
  f32x4 acc[TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int m = 16 * (TM * mp + t) + j;
    mok[t] = m < M;
    xr[t] = x + (int64_t)(mok[t] ? m : 0) * K + 4 * q;
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* wr = W + (int64_t)(nok ? n : 0) * K + 4 * q;
  const float* fr = MK == MK_FULL ? mk.full + (int64_t)(nok ? n : 0) * K + 4 * q : nullptr;
  const float* __restrict__ dcol = mk.dcol;
  float dr = 0.f;
  for (int cb = c0; cb < c1; cb += CB) {
    f32x4 a[TM][CB], b[CB], d[MK == MK_NONE ? 1 : CB];
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const int c = cb + u < c1 ? cb + u : c1 - 1;
#pragma unroll
      for (int t = 0; t < TM; ++t) a[t][u] = *reinterpret_cast<const f32x4u*>(xr[t] + 16 * c);
      b[u] = *reinterpret_cast<const f32x4u*>(wr + 16 * c);
      if (MK == MK_FULL) d[u] = *reinterpret_cast<const f32x4u*>(fr + 16 * c);
      if (MK == MK_DEG) {                              // c is wave-uniform: 16 scalar loads, then the lane's slot
        const float* dc = dcol + 16 * c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s0 = dc[r], s1 = dc[4 + r], s2 = dc[8 + r], s3 = dc[12 + r];
          d[u][r] = q == 0 ? s0 : (q == 1 ? s1 : (q == 2 ? s2 : s3));
        }
      }
    }
    if (MK == MK_DEG && cb == c0) dr = mk.drow[nok ? n : 0];      // behind the fragment requests: not a round trip of its own
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      if (cb + u < c1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float bv = b[u][r];
          if (MK == MK_DEG) bv *= deg_keep(d[u][r], dr, mk.strict);
          else if (MK == MK_FULL) bv *= d[u][r];
          bv = nok ? bv : 0.f;
#pragma unroll
          for (int t = 0; t < TM; ++t) acc[t] = mfma(mok[t] ? a[t][u][r] : 0.f, bv, acc[t]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TM; ++t) red[wave][t][lane] = acc[t];
  __syncthreads();
  if (wave < TM) {                                     // wavefront t finishes batch tile t
    f32x4 s = red[0][wave][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += red[w][wave][lane];
    if (nok) {
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mo = 16 * (TM * mp + wave) + 4 * q + r;
        if (mo < M) {
          float v = s[r] + bv;
          if (relu) v = fmaxf(v, 0.f);
          y[(int64_t)mo * N + n] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------- data gradient
// gx[m][i] = sum_o g[m][o] W[o][i] mask(o, i), then the ReLU gate of the layer's input.  Workgroup (kt, mt): 32 input
// columns as TWO interleaved 16-column tiles (tile c = columns 32 kt + 2 j + c), so that a lane's dwordx2 of a weight
// row serves both tiles and four rows of one request are four full 128-B lines; the contraction runs over the out units
// (K is even: a column pair never straddles the edge).
template <int NW, int MK, int CB>
__device__ __forceinline__ void lin_bwdx_body(f32x4 (*red)[2][64], int kt, int mt, const float* __restrict__ g,
                                              const float* __restrict__ W, LinMask mk, const float* __restrict__ gate,
                                              float* __restrict__ gx, int M, int N, int K) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int nchunks = N / 16;                          // chunks of out units
  const int c0 = nchunks * wave / NW, c1 = nchunks * (wave + 1) / NW;
  const int m = 16 * mt + j;
  const bool mok = m < M;
  const int i0 = 32 * kt + 2 * j;                      // this lane's two input columns (tile 0: i0, tile 1: i0 + 1)
  const bool iok = i0 < K;
  const int ic = iok ? i0 : 0;                         // clamped column pair for the loads
  const float* gr = g + (int64_t)(mok ? m : 0) * N + 4 * q;
  const float dc0 = MK == MK_DEG ? mk.dcol[ic] : 0.f, dc1 = MK == MK_DEG ? mk.dcol[ic + 1] : 0.f;
  f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  for (int cb = c0; cb < c1; cb += CB) {
    f32x4 a[CB], dg[MK == MK_DEG ? CB : 1];
    f32x2u w[CB][4], f[MK == MK_FULL ? CB : 1][4];
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      const int c = cb + u < c1 ? cb + u : c1 - 1;
      a[u] = *reinterpret_cast<const f32x4u*>(gr + 16 * c);
      if (MK == MK_DEG) dg[u] = *reinterpret_cast<const f32x4u*>(mk.drow + 16 * c + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t off = (int64_t)(16 * c + 4 * q + r) * K + ic;
        w[u][r] = *reinterpret_cast<const f32x2u*>(W + off);
        if (MK == MK_FULL) f[u][r] = *reinterpret_cast<const f32x2u*>(mk.full + off);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < CB; ++u) {
      if (cb + u < c1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float w0 = w[u][r][0], w1 = w[u][r][1];
          if (MK == MK_DEG) { w0 *= deg_keep(dc0, dg[u][r], mk.strict); w1 *= deg_keep(dc1, dg[u][r], mk.strict); }
          else if (MK == MK_FULL) { w0 *= f[u][r][0]; w1 *= f[u][r][1]; }
          const float av = mok ? a[u][r] : 0.f;
          acc0 = mfma(av, iok ? w0 : 0.f, acc0);
          acc1 = mfma(av, iok ? w1 : 0.f, acc1);
        }
      }
    }
  }
  red[wave][0][lane] = acc0;
  red[wave][1][lane] = acc1;
  __syncthreads();
  if (wave == 0 && iok) {
    f32x4 s0 = red[0][0][lane], s1 = red[0][1][lane];
#pragma unroll
    for (int w = 1; w < NW; ++w) { s0 += red[w][0][lane]; s1 += red[w][1][lane]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mo = 16 * mt + 4 * q + r;
      if (mo >= M) continue;
      float v0 = s0[r], v1 = s1[r];
      const int64_t o = (int64_t)mo * K + i0;
      if (gate) {
        const f32x2u gt = *reinterpret_cast<const f32x2u*>(gate + o);
        v0 = gt[0] > 0.f ? v0 : 0.f;
        v1 = gt[1] > 0.f ? v1 : 0.f;
      }
      *reinterpret_cast<f32x2u*>(gx + o) = f32x2u{v0, v1};
    }
  }
}
template <int NW, int MK, int CB>
__global__ __launch_bounds__(64 * NW) void lin_bwdx_skinny_k(const float* __restrict__ g, const float* __restrict__ W, LinMask mk,
                                                             const float* __restrict__ gate, float* __restrict__ gx, int M,
                                                             int N, int K) {
  __shared__ f32x4 red[NW][2][64];
  lin_bwdx_body<NW, MK, CB>(red, blockIdx.x, blockIdx.y, g, W, mk, gate, gx, M, N, K);
}

// ----------------------------------------------------------------------------------------------------- weight gradient
// gW[o][i] = mask(o, i) sum_m g[m][o] a[m][i]: the contraction is the (small) batch.  Workgroup (ob, ib): 64 x 64 outputs
// as 4 x 4 INTERLEAVED tiles (out tile c: rows o0 + 4 j + c; in tile c: columns i0 + 4 j + c): a lane's dwordx4 of an
// activation row serves all four in tiles and its four results of one accumulator register are four consecutive
// columns, i.e. 16-B stores.  Eight wavefronts: (wave & 3) = out tile, (wave >> 2) = half of the batch (the halves meet
// in LDS), so that every operand of a wavefront is requested in ONE round (<= SB K-steps of 4 rows).  Column block 0
// also sums g over the batch (the bias gradient).
template <int MK, int SB>
__device__ __forceinline__ void lin_bwdw_body(f32x4 (*red)[4][64], float (*redb)[64], int bx, int by, const float* __restrict__ g,
                                              const float* __restrict__ a, LinMask mk, float* __restrict__ gW,
                                              float* __restrict__ gb, int M, int N, int K) {
  const int lane = threadIdx.x & 63, wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = wave8 & 3, half = wave8 >> 2;
  const int q = lane >> 4, j = lane & 15;
  const int o0 = 64 * bx, i0 = 64 * by;
  const int oa = o0 + 4 * j + wave;                    // out unit this lane feeds as the A operand
  const int ib = i0 + 4 * j;                           // first of this lane's four input columns
  const bool oaok = oa < N, ibok = ib + 3 < K;
  const int ksteps_all = (M + 3) / 4;
  const int s0 = half ? (ksteps_all + 1) / 2 : 0, s1 = half ? ksteps_all : (ksteps_all + 1) / 2;
  f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const float* gcol = g + (oaok ? oa : 0);
  const float* arow = a + (ib < K ? ib : 0);
  for (int sb = s0; sb < s1; sb += SB) {
    float ga[SB];
    f32x4 av[SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int mrow = 4 * (sb + u) + q;
      const int mc = mrow < M ? mrow : M - 1;
      ga[u] = gcol[(int64_t)mc * N];
      if (ibok) av[u] = *reinterpret_cast<const f32x4u*>(arow + (int64_t)mc * K);
      else {
#pragma unroll
        for (int c = 0; c < 4; ++c) av[u][c] = arow[(int64_t)mc * K + (ib + c < K ? c : 0)];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int mrow = 4 * (sb + u) + q;
      const bool ok = sb + u < s1 && mrow < M;
      const float gv = (ok && oaok) ? ga[u] : 0.f;
      bsum += gv;
      if (sb + u < s1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = mfma(gv, (ok && ib + c < K) ? av[u][c] : 0.f, acc[c]);
      }
    }
  }
  if (half) {
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wave][c][lane] = acc[c];
    redb[wave][lane] = bsum;
  }
  __syncthreads();
  if (half) return;
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] += red[wave][c][lane];
  bsum += redb[wave][lane];
  // lane (q, j), register r of accumulator c: out unit o0 + 4 (4 q + r) + wave, input column i0 + 4 j + c
  f32x4 dcv = f32x4{0.f, 0.f, 0.f, 0.f};
  if (MK == MK_DEG) {
#pragma unroll
    for (int c = 0; c < 4; ++c) dcv[c] = mk.dcol[ib + c < K ? ib + c : 0];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = o0 + 4 * (4 * q + r) + wave;
    if (o >= N) continue;
    f32x4 v = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    const int64_t off = (int64_t)o * K + ib;
    if (MK == MK_DEG) {
      const float dr = mk.drow[o];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] *= deg_keep(dcv[c], dr, mk.strict);
    } else if (MK == MK_FULL) {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] *= mk.full[ib + c < K ? off + c : (int64_t)o * K];   // never past row o (ib may be >= K)
    }
    if (ibok) *reinterpret_cast<f32x4u*>(gW + off) = v;
    else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (ib + c < K) gW[off + c] = v[c];
    }
  }
  if (gb && by == 0) {                          // bias gradient: sum over the batch slots q of this lane's out unit
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (q == 0 && oaok) gb[oa] = bsum;
  }
}
template <int MK, int SB>
__global__ __launch_bounds__(512) void lin_bwdw_skinny_k(const float* __restrict__ g, const float* __restrict__ a, LinMask mk,
                                                         float* __restrict__ gW, float* __restrict__ gb, int M, int N, int K) {
  __shared__ f32x4 red[4][4][64];
  __shared__ float redb[4][64];
  lin_bwdw_body<MK, SB>(red, redb, blockIdx.x, blockIdx.y, g, a, mk, gW, gb, M, N, K);
}

// ------------------------------------------------------------------------------------------- both gradients, one launch
// The two gradients of a layer read the same g and write disjoint outputs; at these sizes each is a few memory round
// trips plus a launch, so they run side by side, two workgroups per CU (<= 128 registers: the data-gradient role asks
// for 4 chunks per round instead of 8).  Workgroups [0, nbx) take the data-gradient role -- the longer one when the
// contraction over the out units needs several rounds -- so that on a grid above the 512 resident workgroups (1568 out
// units: 624) it is the short weight-gradient workgroups that start late.
template <int MK>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void lin_bwd_both_k(const float* __restrict__ g, const float* __restrict__ W,
                                                      const float* __restrict__ a, LinMask mk, const float* __restrict__ gate,
                                                      float* __restrict__ gx, float* __restrict__ gW, float* __restrict__ gb,
                                                      int M, int N, int K, int nbx, int wgx, int xgx) {
  __shared__ f32x4 red[8 * 2 * 64];
  __shared__ float redb[4][64];
  const int bid = blockIdx.x;
  if (bid < nbx) lin_bwdx_body<8, MK, 4>(reinterpret_cast<f32x4 (*)[2][64]>(red), bid % xgx, bid / xgx, g, W, mk, gate, gx, M, N, K);
  else lin_bwdw_body<MK, 16>(reinterpret_cast<f32x4 (*)[4][64]>(red), redb, (bid - nbx) % wgx, (bid - nbx) / wgx, g, a, mk, gW, gb, M, N, K);
}

template <int NW, int CB, int TM>
void launch_fwd(int kind, hipStream_t s, const float* x, const float* W, LinMask mk, const float* b, int relu, float* y, int M,
                int N, int K) {
  const dim3 grid((unsigned)((N + 15) / 16), (unsigned)((M + 16 * TM - 1) / (16 * TM)));
  if (kind == MK_DEG) hipLaunchKernelGGL((lin_fwd_skinny_k<NW, MK_DEG, CB, TM>), grid, dim3(64 * NW), 0, s, x, W, mk, b, relu, y, M, N, K);
  else if (kind == MK_FULL) hipLaunchKernelGGL((lin_fwd_skinny_k<NW, MK_FULL, CB, TM>), grid, dim3(64 * NW), 0, s, x, W, mk, b, relu, y, M, N, K);
  else hipLaunchKernelGGL((lin_fwd_skinny_k<NW, MK_NONE, CB, TM>), grid, dim3(64 * NW), 0, s, x, W, mk, b, relu, y, M, N, K);
}
template <int NW, int CB>
void launch_bwdx(int kind, dim3 grid, hipStream_t s, const float* g, const float* W, LinMask mk, const float* gate, float* gx,
                 int M, int N, int K) {
  if (kind == MK_DEG) hipLaunchKernelGGL((lin_bwdx_skinny_k<NW, MK_DEG, CB>), grid, dim3(64 * NW), 0, s, g, W, mk, gate, gx, M, N, K);
  else if (kind == MK_FULL) hipLaunchKernelGGL((lin_bwdx_skinny_k<NW, MK_FULL, CB>), grid, dim3(64 * NW), 0, s, g, W, mk, gate, gx, M, N, K);
  else hipLaunchKernelGGL((lin_bwdx_skinny_k<NW, MK_NONE, CB>), grid, dim3(64 * NW), 0, s, g, W, mk, gate, gx, M, N, K);
}
inline int mask_kind(const float* mask, const float* deg_out) { return deg_out ? MK_DEG : (mask ? MK_FULL : MK_NONE); }

constexpr int kSkinnyM = 128;
bool skinny_ok(int64_t M) {
  return M >= 1 && M <= kSkinnyM;
}

}  // namespace

extern "C" {

// workspace of the three entry points below (the tiled-GEMM fallbacks split K through it; the skinny kernels need none)
int64_t gnf_linear_ws_bytes(int64_t M, int64_t N, int64_t K) {
  int64_t w = gnf_gemm_ws_bytes(M, N, K);
  const int64_t w2 = gnf_gemm_ws_bytes(M, K, N), w3 = gnf_gemm_ws_bytes(N, K, M), w4 = gnf_colsum_ws_bytes(M, N);
  const int64_t w6 = gnf_colsum_ws_bytes(M, K);
  w = w > w2 ? w : w2; w = w > w3 ? w : w3; w = w > w4 ? w : w4; w = w > w6 ? w : w6;
  if (gnf_linear_tall_ok(M, N, K)) {
    const int64_t w5 = gnf_linear_tall_ws_floats(M, N, K) * (int64_t)sizeof(float);
    w = w > w5 ? w : w5;
  }
  return w;
}

int gnf_linear_fwd(const float* x, const float* W, const float* b, const float* mask, const float* deg_out,
                   const float* deg_in, int strict, int relu, float* y, int64_t M, int64_t N, int64_t K, float* ws,
                   int64_t ws_bytes, gnf_stream_t stream) {
  if (M < 0 || N <= 0 || K <= 0) return GNF_EINVAL;
  if (M == 0) return 0;
  if (!x || !W || !y || (!deg_out) != (!deg_in)) return GNF_EINVAL;
  if (skinny_ok(M) && K % 16 == 0) {
    const LinMask mk{mask, deg_out, deg_in, strict};
    // 16 wavefronts: K = 1024 is 4 chunks of 16 per wavefront, requested in one round.  Two batch tiles per workgroup
    // unless that grid is more than one round of the CUs (1568 out units at B = 100: 392 workgroups) -- then four.
    const int64_t two = (N + 15) / 16 * ((M + 31) / 32);
    if (two <= 256 || M <= 32) launch_fwd<16, 4, 2>(mask_kind(mask, deg_out), (hipStream_t)stream, x, W, mk, b, relu, y, (int)M, (int)N, (int)K);
    else launch_fwd<16, 4, 4>(mask_kind(mask, deg_out), (hipStream_t)stream, x, W, mk, b, relu, y, (int)M, (int)N, (int)K);
    GNF_LAUNCH_CHECK();
    return 0;
  }
  if (!mask && !deg_out && gnf_linear_tall_fwd_ok(M, N, K))
    return gnf_linear_tall_fwd(x, W, b, relu, y, M, N, K, (hipStream_t)stream);
  if (deg_out && !mask) return GNF_EINVAL;             // the tiled GEMM multiplies a mask tensor in
  return gnf_gemm(x, K, 1, W, mask, 1, K, y, N, 1, b, nullptr, 0, 0, nullptr, 0, 0, relu ? GNF_GEMM_RELU : 0, M, N, K, ws,
                  ws_bytes, stream);
}

int gnf_linear_bwd_x(const float* g, const float* W, const float* mask, const float* deg_out, const float* deg_in,
                     int strict, const float* gate, float* gx, int64_t M, int64_t N, int64_t K, float* ws,
                     int64_t ws_bytes, gnf_stream_t stream) {
  if (M < 0 || N <= 0 || K <= 0) return GNF_EINVAL;
  if (M == 0) return 0;
  if (!g || !W || !gx || (!deg_out) != (!deg_in)) return GNF_EINVAL;
  if (skinny_ok(M) && N % 16 == 0 && K % 2 == 0) {
    const LinMask mk{mask, deg_out, deg_in, strict};
    const dim3 grid((unsigned)((K + 31) / 32), (unsigned)((M + 15) / 16));
    if (N <= 1024) launch_bwdx<8, 8>(mask_kind(mask, deg_out), grid, (hipStream_t)stream, g, W, mk, gate, gx, (int)M, (int)N, (int)K);
    else launch_bwdx<16, 4>(mask_kind(mask, deg_out), grid, (hipStream_t)stream, g, W, mk, gate, gx, (int)M, (int)N, (int)K);
    GNF_LAUNCH_CHECK();
    return 0;
  }
  if (deg_out && !mask) return GNF_EINVAL;
  return gnf_gemm(g, N, 1, W, mask, K, 1, gx, K, 1, nullptr, nullptr, 0, 0, gate, K, 1, 0, M, K, N, ws, ws_bytes, stream);
}

int gnf_linear_bwd_w(const float* g, const float* a, const float* mask, const float* deg_out, const float* deg_in,
                     int strict, float* gW, float* gb, int64_t M, int64_t N, int64_t K, float* ws, int64_t ws_bytes,
                     gnf_stream_t stream) {
  if (M < 0 || N <= 0 || K <= 0) return GNF_EINVAL;
  if (!gW || (M > 0 && (!g || !a)) || (!deg_out) != (!deg_in)) return GNF_EINVAL;
  if (skinny_ok(M)) {
    const LinMask mk{mask, deg_out, deg_in, strict};
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((K + 63) / 64));
    const int kind = mask_kind(mask, deg_out);
    if (kind == MK_DEG) hipLaunchKernelGGL((lin_bwdw_skinny_k<MK_DEG, 16>), grid, dim3(512), 0, (hipStream_t)stream, g, a, mk, gW, gb, (int)M, (int)N, (int)K);
    else if (kind == MK_FULL) hipLaunchKernelGGL((lin_bwdw_skinny_k<MK_FULL, 16>), grid, dim3(512), 0, (hipStream_t)stream, g, a, mk, gW, gb, (int)M, (int)N, (int)K);
    else hipLaunchKernelGGL((lin_bwdw_skinny_k<MK_NONE, 16>), grid, dim3(512), 0, (hipStream_t)stream, g, a, mk, gW, gb, (int)M, (int)N, (int)K);
    GNF_LAUNCH_CHECK();
    return 0;
  }
  if (deg_out && !mask) return GNF_EINVAL;
  int rc = gnf_gemm(g, 1, N, a, nullptr, K, 1, gW, K, 1, nullptr, mask, K, 1, nullptr, 0, 0, 0, N, K, M, ws, ws_bytes, stream);
  if (rc || !gb) return rc;
  if (M == 0) return (int)hipMemsetAsync(gb, 0, sizeof(float) * N, (hipStream_t)stream);
  if (ws_bytes < gnf_colsum_ws_bytes(M, N)) return GNF_EWS;
  return gnf_colsum(g, N, gb, M, N, ws, stream);
}

// 1 when gnf_linear_bwd produces `gxsum` in the launch that produces gx (else it costs a column-sum pass over gx)
int gnf_linear_gxsum_fused(int64_t M, int64_t N, int64_t K, int masked) {
  return (!masked && gnf_linear_tall_ok(M, N, K)) ? 1 : 0;
}

// both gradients of one layer (gnf_linear_bwd_w + gnf_linear_bwd_x, same arguments); small and tall-narrow batches: ONE
// launch.  gxsum (may be NULL): column sums of gx, i.e. the bias gradient of the layer that produced `a`.
int gnf_linear_bwd(const float* g, const float* W, const float* a, const float* mask, const float* deg_out,
                   const float* deg_in, int strict, const float* gate, float* gx, float* gW, float* gb, float* gxsum,
                   int64_t M, int64_t N, int64_t K, float* ws, int64_t ws_bytes, gnf_stream_t stream) {
  if (M < 0 || N <= 0 || K <= 0) return GNF_EINVAL;
  if (!gW || (M > 0 && (!g || !a || !W || !gx)) || (!deg_out) != (!deg_in)) return GNF_EINVAL;
  if (M > 0 && skinny_ok(M) && N % 16 == 0 && K % 2 == 0) {
    const LinMask mk{mask, deg_out, deg_in, strict};
    const int wgx = (int)((N + 63) / 64), wgy = (int)((K + 63) / 64), xgx = (int)((K + 31) / 32), xgy = (int)((M + 15) / 16);
    const int nbx = xgx * xgy;
    const dim3 grid((unsigned)(nbx + wgx * wgy));
    const int kind = mask_kind(mask, deg_out);
    if (kind == MK_DEG) hipLaunchKernelGGL((lin_bwd_both_k<MK_DEG>), grid, dim3(512), 0, (hipStream_t)stream, g, W, a, mk, gate, gx, gW, gb, (int)M, (int)N, (int)K, nbx, wgx, xgx);
    else if (kind == MK_FULL) hipLaunchKernelGGL((lin_bwd_both_k<MK_FULL>), grid, dim3(512), 0, (hipStream_t)stream, g, W, a, mk, gate, gx, gW, gb, (int)M, (int)N, (int)K, nbx, wgx, xgx);
    else hipLaunchKernelGGL((lin_bwd_both_k<MK_NONE>), grid, dim3(512), 0, (hipStream_t)stream, g, W, a, mk, gate, gx, gW, gb, (int)M, (int)N, (int)K, nbx, wgx, xgx);
    GNF_LAUNCH_CHECK();
    if (gxsum) {
      if (ws_bytes < gnf_colsum_ws_bytes(M, K)) return GNF_EWS;
      return gnf_colsum(gx, K, gxsum, M, K, ws, stream);
    }
    return 0;
  }
  if (!mask && !deg_out && gnf_linear_tall_ok(M, N, K) && (!gate || gate == a)) {
    if (!ws || ws_bytes < gnf_linear_tall_ws_floats(M, N, K) * (int64_t)sizeof(float)) return GNF_EWS;
    return gnf_linear_tall_bwd(g, W, a, gate, gx, gW, gb, gxsum, M, N, K, ws, (hipStream_t)stream);
  }
  const int rc = gnf_linear_bwd_w(g, a, mask, deg_out, deg_in, strict, gW, gb, M, N, K, ws, ws_bytes, stream);
  if (rc) return rc;
  if (M == 0)                                          // an empty batch: zero weight gradients, no rows of gx
    return gxsum ? (int)hipMemsetAsync(gxsum, 0, sizeof(float) * K, (hipStream_t)stream) : 0;
  const int rx = gnf_linear_bwd_x(g, W, mask, deg_out, deg_in, strict, gate, gx, M, N, K, ws, ws_bytes, stream);
  if (rx || !gxsum) return rx;
  if (ws_bytes < gnf_colsum_ws_bytes(M, K)) return GNF_EWS;
  return gnf_colsum(gx, K, gxsum, M, K, ws, stream);
}

}  // extern "C"
