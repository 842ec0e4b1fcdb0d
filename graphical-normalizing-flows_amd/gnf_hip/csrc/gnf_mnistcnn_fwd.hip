// Convolutional front of the MNISTCNN embedding net (models/MLP.py:36-41) over the B*d masked
// images of the DAG conditioner: conv3x3(1->16) + ReLU + conv3x3(16->16) + maxpool2, forward
// and backward, as implicit GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32).  This is where
// the MNIST d=784 Monotonic+DAG step spends its flops (1.42 MMAC per image, 78 400 images
// per 100 samples); the fc layers behind it run on the GEMM of gnf_gemm.hip.
//
// One workgroup processes one 28x28 image at a time, entirely out of LDS:
//   M = 16 output channels (all of them), N = 16 output positions (a 2-row x 8-column patch,
//   so 2x2 pool windows never straddle tiles), K = taps / (input channel, tap) pairs.
//   The weights are the A operand and live in registers for the whole kernel; the B operand
//   is gathered from the LDS image with per-K-step immediate offsets.  Row / channel strides
//   (40 and == 16 mod 32 dwords) make the 64-lane gather bank-conflict free:
//   lane (q, j) -> channel 4g+q (bank +16q), row j>>3 (bank +8), column j&7.
// Backward recomputes conv1 (3 % of the flops) instead of storing 43 KB of activations per
// image, takes the pool argmax saved by the forward (1 byte per pooled value), and keeps the
// weight-gradient accumulators in registers across all images of a workgroup.
//
// This file: the FORWARD kernels (Winograd, and the direct tie-exact one).  Built with -fno-slp-vectorize: the SLP
// vectoriser packs the scalar adds of the output transform into v_pk_add_f32 and pays for it with 77 v_mov per 64 MFMAs
// to form the register pairs (1.44 -> 1.40 ms at cfg4); the backward (gnf_mnistcnn.hip) is faster WITH it.
#include "gnf_mnistcnn.h"

namespace {

#ifdef GNF_CNN_TIMING
__device__ float g_fwd_timing[64];
__device__ long long g_fwd_start[1024];
#define TSTAMP(k) do { const long long t__ = __builtin_readcyclecounter(); tacc[k] += t__ - tlast; tlast = t__; } while (0)
#else
#define TSTAMP(k)
#endif

__global__ __launch_bounds__(64 * FWD_WAVES) void cnn_fwd_k(CnnArgs a) {
#ifdef GNF_CNN_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = __builtin_readcyclecounter();
  const long long tstart0 = tlast;
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* e_s = smem;
  float* a1_s = smem + ESZ;
  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: scalar branches / SGPR math

  // weights as MFMA A operands (row i = j = output channel, K slot q), resident in registers
  float w1f[3];
  int off1[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = 4 * s + q;
    w1f[s] = tap < 9 ? a.W1[j * 9 + tap] : 0.f;
    const int tt = tap < 9 ? tap : 0;
    off1[s] = (tt / 3) * ROWE + tt % 3;
  }
  float w2f[36];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < 9; ++k) w2f[g * 9 + k] = a.W2[(j * NCH + 4 * g + q) * 9 + k];
  f32x4 b1v, b2v;
#pragma unroll
  for (int r = 0; r < 4; ++r) { b1v[r] = a.b1[4 * q + r]; b2v[r] = a.b2[4 * q + r]; }

  for (int i = tid; i < ESZ; i += blockDim.x) e_s[i] = 0.f;

  constexpr int NT = 64 * FWD_WAVES, EPT = (IMG * IMG + NT - 1) / NT;   // image elements per thread
  float pre[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int i = tid + k * NT;
    pre[k] = (blockIdx.x < a.n && i < IMG * IMG) ? a.e[(int64_t)blockIdx.x * (IMG * IMG) + i] : 0.f;
  }
  for (int64_t img = blockIdx.x; img < a.n; img += gridDim.x) {
    __syncthreads();                                   // previous image fully consumed
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int i = tid + k * NT;
      if (i < IMG * IMG) e_s[(i / IMG) * ROWE + i % IMG] = pre[k];
    }
    __syncthreads();
    {                                                  // next image's pixels: in flight under the MFMAs
      const int64_t nx = img + gridDim.x;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int i = tid + k * NT;
        pre[k] = (nx < a.n && i < IMG * IMG) ? a.e[nx * (IMG * IMG) + i] : 0.f;
      }
    }
    TSTAMP(0);
    conv1_tiles<FWD_WAVES>(e_s, a1_s, w1f, off1, b1v, wave, q, j);
    __syncthreads();
    TSTAMP(1);
    // conv2 (implicit GEMM, K = 16 channels x 9 taps) + 2x2 max pool: 36 tiles, two in flight per wave
#pragma nounroll
    for (int t = wave; t < 36; t += 2 * FWD_WAVES) {
      const int tB = t + FWD_WAVES < 36 ? t + FWD_WAVES : t;
      const int yA = 2 * (t / 3) + (j >> 3), xA = 8 * (t % 3) + (j & 7);
      const int yB = 2 * (tB / 3) + (j >> 3), xB = 8 * (tB % 3) + (j & 7);
      const float* pA = a1_s + q * CH + yA * ROW + xA;
      const float* pB = a1_s + q * CH + yB * ROW + xB;
      f32x4 accA = b2v, accB = b2v;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int o = 4 * g * CH + ky * ROW + kx;
            accA = mfma(w2f[g * 9 + ky * 3 + kx], pA[o], accA);
            accB = mfma(w2f[g * 9 + ky * 3 + kx], pB[o], accB);
          }
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const f32x4 acc = half ? accB : accA;
        const int tt = half ? tB : t;
        if (half && tB == t) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v00 = acc[r];
          const float v01 = dpp_xor1(v00), v10 = dpp_xor8(v00), v11 = dpp_xor8(v01);
          if ((j & 9) == 0) {                          // top-left lane of a 2x2 window; first max wins ties
            float best = v00; int bi = 0;
            if (v01 > best) { best = v01; bi = 1; }
            if (v10 > best) { best = v10; bi = 2; }
            if (v11 > best) { best = v11; bi = 3; }
            const int64_t o = img * NPOOL + (4 * q + r) * (PO * PO) + (tt / 3) * PO + 4 * (tt % 3) + ((j & 7) >> 1);
            a.pooled[o] = best;
            a.arg[o] = (unsigned char)bi;
          }
        }
      }
    }
    TSTAMP(2);
  }
#ifdef GNF_CNN_TIMING
  if (blockIdx.x == 7 && (tid & 63) == 0)
    for (int k = 0; k < 3; ++k) g_fwd_timing[wave * 8 + k] = (float)tacc[k];
  if (tid == 0) { g_fwd_start[blockIdx.x] = tstart0; g_fwd_start[512 + blockIdx.x] = __builtin_readcyclecounter(); }
#endif
}

// ---------------------------------------------------------------------------------------------
// Forward with conv2 as Winograd F(2x2,3x3): 2.25x fewer MFMAs than the implicit GEMM above.
//   Y = A^T [ sum_c (G w G^T)_c (.) (B^T d_c B) ] A  per 2x2 output tile, d = 4x4 input patch.
// The 2x2 output tile IS the pool window, so pooling happens in registers of one lane.
// GEMM view per transform point xi (16 of them): M_xi[o][tile] = sum_c U_xi[o][c] V_xi[c][tile];
//   A operand = U_xi (transformed weights, 64 registers per lane, resident for the whole kernel),
//   B operand = V_xi computed by the lane itself from its 4x4 patch (8 ds_read_b64 + 32 adds feed 16 MFMAs),
//   D: lane (q,j) holds out-channels 4q..4q+3 of tile j for all 16 xi -> the output transform is lane-local.
// A workgroup handles two images per iteration (18 groups of 16 tiles over 8 wavefronts: 5/5/4/4 per SIMD).
// ---------------------------------------------------------------------------------------------
// LDS geometry of the Winograd forward (round 4): a1 as [image][16][26][WROW = 28] with channel stride WCH = 736.
//  * the patch gather reads row pairs of 16 consecutive tiles of ONE channel per 16-lane group (hipcc fuses the two b64
//    halves of a patch row into one ds_read2_b64: 2 x 4 groups of 16 contiguous lanes over 32 banks): tiles are 2 dwords
//    apart, 12 per tile row, so the row wrap 2 WROW - 22 must be 2 (mod 32): WROW = 28 (the [26][40] image of rounds 1-3
//    put the wrap at 26 mod 32: bank conflicts were 47 % of the LDS-active cycles, profiles/r03_cnn_pmc.json);
//    WCH = 32 (mod 64) keeps the two channels of a 32-lane group apart where the halves stay plain ds_read_b64.
//  * 47 KB per image instead of 66.5; the input images are unpadded (pitch 28 = WROW: one offset serves e and a1).
constexpr int WROW = IMG, WCH = 736, A1SZ = NCH * WCH, WESZ = IMG * IMG;
static_assert(WCH >= 25 * WROW + C1 && WCH % 64 == 32 && (2 * WROW - 22) % 32 == 2, "forward a1 layout");

// 2x2 max pool of one (channel, tile): the four outputs of the tile
__device__ __forceinline__ void pool_store4(const CnnArgs& a, int64_t o, float v00, float v01, float v10, float v11) {
  float best = v00; int bi = 0;                        // first max wins ties (torch max_pool2d order)
  if (v01 > best) { best = v01; bi = 1; }
  if (v10 > best) { best = v10; bi = 2; }
  if (v11 > best) { best = v11; bi = 3; }
  a.pooled[o] = best;
  a.arg[o] = (unsigned char)bi;
}
// column half of the output transform (s0 / s1 = the two rows of A^T M) + the pool
__device__ __forceinline__ void pool_store(const CnnArgs& a, int64_t o, const float (&s0)[4], const float (&s1)[4]) {
  pool_store4(a, o, s0[0] + s0[1] + s0[2], s0[1] - s0[2] - s0[3], s1[0] + s1[1] + s1[2], s1[1] - s1[2] - s1[3]);
}

// conv1 units (64 positions each, 11 per image) per wavefront: image 0 on wavefronts 0-3, image 1 on wavefronts 4-7
__device__ constexpr int FC1U[4] = {3, 3, 3, 2};

__global__ __launch_bounds__(64 * FWD_WAVES) void cnn_fwd_wino_k(CnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* e_s = smem;                       // [2][28 x 28]
  float* a1_s = smem + 2 * WESZ;           // [2][16][26][WROW]
  float* xch = a1_s + 2 * A1SZ;            // [2 items][2 halves][16][64]: partial output transforms of the split items
  float* w1_s = xch + 2 * 2 * 16 * 64;     // W1 as [tap][channel]: conv1's A operands, re-read per call
  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = FWD_WAVES, NT = 64 * FWD_WAVES;

  if (tid < 9 * NCH) w1_s[tid] = a.W1[(tid & 15) * 9 + (tid >> 4)];
  f32x4 b1v, b2v;
#pragma unroll
  for (int r = 0; r < 4; ++r) { b1v[r] = a.b1[4 * q + r]; b2v[r] = a.b2[4 * q + r]; }

  // U = G w G^T of W2[o = j][c = 4g+q], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; fp64 once, rounded to fp32
  float uw[64];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float* w = a.W2 + (j * NCH + 4 * g + q) * 9;
    double gw[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double w0 = w[c], w1 = w[3 + c], w2 = w[6 + c];
      gw[0][c] = w0; gw[1][c] = 0.5 * (w0 + w1 + w2); gw[2][c] = 0.5 * (w0 - w1 + w2); gw[3][c] = w2;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uw[(4 * r + 0) * 4 + g] = (float)gw[r][0];
      uw[(4 * r + 1) * 4 + g] = (float)(0.5 * (gw[r][0] + gw[r][1] + gw[r][2]));
      uw[(4 * r + 2) * 4 + g] = (float)(0.5 * (gw[r][0] - gw[r][1] + gw[r][2]));
      uw[(4 * r + 3) * 4 + g] = (float)gw[r][2];
    }
  }

  // this wavefront's conv1 units: image slot wave / 4, units [c1u0, c1u0 + FC1U[wave & 3])
  const int c1s = wave >> 2;
  int c1u0 = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) c1u0 += w < (wave & 3) ? FC1U[w] : 0;
  const int c1n = FC1U[wave & 3];

  int64_t prev_ob = -1;                    // output offset of this wavefront's split item of the previous pair (or none)
  auto finish_split = [&](const float* xb0, int64_t ob) {      // xb0: [2 halves][16][64] of the item
    const float* xa = xb0 + lane;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = xa[(r * 4 + k) * 64] + xa[16 * 64 + (r * 4 + k) * 64];
      pool_store4(a, ob + r * (PO * PO), v[0], v[1], v[2], v[3]);
    }
  };
  const int64_t npair = (a.n + 1) >> 1;
  constexpr int EPT = (2 * IMG * IMG + NT - 1) / NT;          // pixels of an image pair per thread
  float pre[EPT];
  auto fetch = [&](int64_t pair) {
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int i = tid + k * NT;
      const int64_t o = pair * (2 * IMG * IMG) + i;
      pre[k] = (pair < npair && i < 2 * IMG * IMG && o < a.n * (IMG * IMG)) ? a.e[o] : 0.f;
    }
  };
  fetch(blockIdx.x);
  for (int64_t pair = blockIdx.x; pair < npair; pair += gridDim.x) {
    __syncthreads();                                   // previous pair fully consumed
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int i = tid + k * NT;
      if (i < 2 * IMG * IMG) e_s[i] = pre[k];          // the two unpadded images are contiguous in LDS as in memory
    }
    __syncthreads();
    fetch(pair + gridDim.x);                           // next pair's pixels: in flight under the MFMAs

    // conv1 + ReLU of both images: 2 x 11 units of 64 positions on v_mfma_f32_16x16x1_4b_f32 (gnf_mnistcnn.h)
    if (c1n == 3) conv1_units<3, WCH>(e_s + c1s * WESZ, a1_s + c1s * A1SZ, c1u0, w1_s + j, b1v, q, j, lane);
    else conv1_units<2, WCH>(e_s + c1s * WESZ, a1_s + c1s * A1SZ, c1u0, w1_s + j, b1v, q, j, lane);
    __syncthreads();

    // the split items of the PREVIOUS pair: the partner's half arrived before the barrier at the top of this iteration
    if (wave < 4 && !(wave & 1) && prev_ob >= 0) finish_split(xch + (wave >> 1) * 2 * 16 * 64, prev_ob);
    prev_ob = -1;

    // 18 items (image, group of 16 Winograd tiles) over 4 SIMDs: 16 whole ones, two per wavefront, and the last two
    // SPLIT by xi_y halves over the wavefronts 0..3 (one per SIMD), so that every SIMD carries 4.5 items instead of
    // 5 / 5 / 4 / 4 (two SIMDs idle for one item in five).  A half item accumulates 8 of the 16 transform points, applies
    // its rows of the output transform A^T M (row sums are additive in xi_y) and leaves 32 partial values per lane in
    // LDS; the wavefront holding xi_y = 0, 1 adds its partner's at the top of the next iteration (above) and finishes.
#pragma nounroll
    for (int item = wave; item < 16; item += NW) {
      const int s = item >= 9, grp = item - 9 * s;     // wave-uniform
      const int64_t img = 2 * pair + s;
      if (img >= a.n) continue;
      const int t = 16 * grp + j, ty = t / 12, tx = t - 12 * ty;
      const float* base = a1_s + s * A1SZ + q * WCH + 2 * ty * WROW + 2 * tx;
      f32x4 acc[16];
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) acc[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[5] = b2v;                                    // xi = (1,1) reaches all four outputs with weight +1: the bias
      // per 4 input channels: 16 operands first, then 16 back-to-back MFMAs (VALU and MFMA of one wavefront do not
      // overlap, tools/mfma_feed.hip); the next channel group's patch is loaded before the MFMAs so that its LDS
      // latency hides under them
      f32x2 plo[4], phi[4];                            // patch rows as two column pairs
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        plo[rr] = *reinterpret_cast<const f32x2*>(base + rr * WROW);
        phi[rr] = *reinterpret_cast<const f32x2*>(base + rr * WROW + 2);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float vv[16];
        wino_in(plo, phi, vv);
        if (g < 3) {
          const float* p = base + 4 * (g + 1) * WCH;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            plo[rr] = *reinterpret_cast<const f32x2*>(p + rr * WROW);
            phi[rr] = *reinterpret_cast<const f32x2*>(p + rr * WROW + 2);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) acc[xi] = mfma(uw[xi * 4 + g], vv[xi], acc[xi]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // output transform A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]) + 2x2 max pool, lane-local
      const int64_t ob = img * NPOOL + 4 * q * (PO * PO) + t;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s0[4], s1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          s0[c] = acc[c][r] + acc[4 + c][r] + acc[8 + c][r];
          s1[c] = acc[4 + c][r] - acc[8 + c][r] - acc[12 + c][r];
        }
        pool_store(a, ob + r * (PO * PO), s0, s1);
      }
    }
    if (wave < 4) {
      const int item = 16 + (wave >> 1), hf = wave & 1;          // item 16 / 17 = groups 7 / 8 of image 1; xi_y in {2 hf, 2 hf + 1}
      const int grp = item - 9;
      const int64_t img = 2 * pair + 1;
      if (img < a.n) {
        const int t = 16 * grp + j, ty = t / 12, tx = t - 12 * ty;
        const float* base = a1_s + A1SZ + q * WCH + 2 * ty * WROW + 2 * tx;
        f32x4 acc[8];
#pragma unroll
        for (int xi = 0; xi < 8; ++xi) acc[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (hf == 0) acc[5] = b2v;
        f32x2 plo[4], phi[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          plo[rr] = *reinterpret_cast<const f32x2*>(base + rr * WROW);
          phi[rr] = *reinterpret_cast<const f32x2*>(base + rr * WROW + 2);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float vv[16];
          wino_in(plo, phi, vv);
          if (g < 3) {
            const float* p = base + 4 * (g + 1) * WCH;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              plo[rr] = *reinterpret_cast<const f32x2*>(p + rr * WROW);
              phi[rr] = *reinterpret_cast<const f32x2*>(p + rr * WROW + 2);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (hf == 0) {
#pragma unroll
            for (int xi = 0; xi < 8; ++xi) acc[xi] = mfma(uw[xi * 4 + g], vv[xi], acc[xi]);
          } else {
#pragma unroll
            for (int xi = 0; xi < 8; ++xi) acc[xi] = mfma(uw[(8 + xi) * 4 + g], vv[8 + xi], acc[xi]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // this half's share of A^T M A: rows s0 += xi_y 0, 1, 2;  s1 += xi_y 1, -2, -3, then the (linear) column half
        float* xb = xch + ((wave >> 1) * 2 + hf) * 16 * 64 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p0[4], p1[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            p0[c] = hf == 0 ? acc[c][r] + acc[4 + c][r] : acc[c][r];
            p1[c] = hf == 0 ? acc[4 + c][r] : -acc[c][r] - acc[4 + c][r];
          }
          xb[(r * 4 + 0) * 64] = p0[0] + p0[1] + p0[2];
          xb[(r * 4 + 1) * 64] = p0[1] - p0[2] - p0[3];
          xb[(r * 4 + 2) * 64] = p1[0] + p1[1] + p1[2];
          xb[(r * 4 + 3) * 64] = p1[1] - p1[2] - p1[3];
        }
        if (hf == 0) prev_ob = img * NPOOL + 4 * q * (PO * PO) + t;
      }
    }
  }
  __syncthreads();                                     // the last pair's halves
  if (wave < 4 && !(wave & 1) && prev_ob >= 0) finish_split(xch + (wave >> 1) * 2 * 16 * 64, prev_ob);
}

constexpr size_t kFwdLds = (size_t)(ESZ + NCH * CH) * sizeof(float);
constexpr size_t kWinoLds = (size_t)(2 * WESZ + 2 * A1SZ + 2 * 2 * 16 * 64 + 9 * NCH) * sizeof(float);   // + the split items' exchange, the W1 table
static_assert(kWinoLds <= 160 * 1024, "one workgroup per CU");
constexpr unsigned kWinoGrid = 256;                  // one 8-wave workgroup per CU, two images per iteration
constexpr unsigned kFwdGrid = 512;                   // direct kernel: 512 measured faster than 256

}  // namespace

extern "C" {

int gnf_mnistcnn_conv_fwd(const float* e, const float* W1, const float* b1, const float* W2, const float* b2,
                          float* pooled, unsigned char* argmax, int64_t n_img, int exact_ties, gnf_stream_t stream) {
  if (!W1 || !b1 || !W2 || !b2 || n_img < 0) return GNF_EINVAL;
  if (n_img == 0) return 0;                // image-sized arrays may be NULL for an empty batch
  if (!e || !pooled || !argmax) return GNF_EINVAL;
  CnnArgs a{};
  a.e = e; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.pooled = pooled; a.arg = argmax; a.n = n_img;
  if (exact_ties) {                                 // direct implicit GEMM: bit-equal outputs for equal patches
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cnn_fwd_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kFwdLds);
    const unsigned grid = n_img < kFwdGrid ? (unsigned)n_img : kFwdGrid;
    hipLaunchKernelGGL(cnn_fwd_k, dim3(grid), dim3(64 * FWD_WAVES), kFwdLds, (hipStream_t)stream, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cnn_fwd_wino_k),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLds);
    const int64_t npair = (n_img + 1) / 2;
    const unsigned grid = npair < kWinoGrid ? (unsigned)npair : kWinoGrid;
    hipLaunchKernelGGL(cnn_fwd_wino_k, dim3(grid), dim3(64 * FWD_WAVES), kWinoLds, (hipStream_t)stream, a);
  }
  GNF_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
