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
#include "gnf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int IMG = 28, C1 = 26, C2 = 24, PO = 12, NCH = 16;
constexpr int ROWE = 36, ESZ = IMG * ROWE;            // padded input image
constexpr int ROW = 40, CH = C1 * ROW;                // conv1 activations: [16][26][40], CH = 1040 == 16 mod 32
constexpr int ROWD = 40, CHD = 28 * ROWD + 16;        // dY2 with a 2-wide zero border: [16][28][40], CHD = 1136 == 16 mod 32
constexpr int CS = 688;                               // per-tap planes T: [9][688] flat 26x26 positions
constexpr int NPOOL = NCH * PO * PO;                  // 2304
constexpr int FWD_WAVES = 8, BWD_WAVES = 8;
constexpr int PROW = NCH * 144 + NCH * 16 + NCH;      // per-wave gradient partial row: dW2 | dW1+db1 | db2

static_assert(CH % 32 == 16 && CHD % 32 == 16, "channel strides must sit 16 banks apart");

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Packed fp32 VALU (2 flops per lane per op) for the Winograd transforms.  One row (t0,t1,t2,t3) of B^T d held as
// A = (t0,t1), B = (t2,t3) gives the four outputs of (B^T d) B in two instructions:
//   A - B                       = (t0 - t2, t1 - t3) = (v0, v3)
//   (A.hi + B.lo, -A.hi + B.lo) = (t1 + t2, t2 - t1) = (v1, v2)      [op_sel picks the halves, neg_hi negates src0]
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_v12(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// B^T d B of a 4x4 patch given as 4 rows x 2 column pairs; out[xi = 4*xi_y + xi_x]
__device__ __forceinline__ void wino_in(const f32x2 (&lo)[4], const f32x2 (&hi)[4], float (&v)[16]) {
  f32x2 tl[4], th[4];                                 // B^T d: rows d0-d2, d1+d2, d2-d1, d1-d3
  tl[0] = lo[0] - lo[2]; th[0] = hi[0] - hi[2];
  tl[1] = lo[1] + lo[2]; th[1] = hi[1] + hi[2];
  tl[2] = lo[2] - lo[1]; th[2] = hi[2] - hi[1];
  tl[3] = lo[1] - lo[3]; th[3] = hi[1] - hi[3];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const f32x2 v03 = tl[rr] - th[rr], v12 = pk_v12(tl[rr], th[rr]);
    v[4 * rr + 0] = v03.x; v[4 * rr + 1] = v12.x; v[4 * rr + 2] = v12.y; v[4 * rr + 3] = v03.y;
  }
}

// lane^1 (quad_perm [1,0,3,2]) and lane^8 (row_ror:8 inside a 16-lane row) without touching LDS
__device__ __forceinline__ float dpp_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_xor8(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));
}

struct CnnArgs {
  const float* e; const float* W1; const float* b1; const float* W2; const float* b2;
  float* pooled; unsigned char* arg;                                // forward outputs
  const float* gp; const unsigned char* argin; float* ge; float* part;   // backward
  int64_t n;
};

// conv1 + ReLU of the image in e_s into a1_s; 43 tiles of 16 consecutive positions of the 26x26 grid.
// All operand reads of a wave's (up to 6) tiles are issued first, then 6 independent 3-step MFMA chains,
// then the stores: the phase is latency-bound, so nothing may serialise behind a single chain.
template <int NW>
__device__ __forceinline__ void conv1_tiles(const float* e_s, float* a1_s, const float (&w1f)[3], const int (&off1)[3],
                                            const f32x4& b1v, int wave, int q, int j) {
  constexpr int NTL = (43 + NW - 1) / NW, NB = NTL;           // tiles per wave, processed NB at a time
#pragma unroll
  for (int k0 = 0; k0 < NTL; k0 += NB) {
    int po[NB];
    f32x4 acc[NB];
    float ev[NB][3];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int pos = 16 * (wave + NW * (k0 + k)) + j;
      const int pc = pos < C1 * C1 ? pos : 0;
      const int y = pc / C1, x = pc - y * C1;
      po[k] = pos < C1 * C1 ? y * ROW + x : -1;
#pragma unroll
      for (int s = 0; s < 3; ++s) ev[k][s] = e_s[y * ROWE + x + off1[s]];
      acc[k] = b1v;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int k = 0; k < NB; ++k) acc[k] = mfma(w1f[s], ev[k][s], acc[k]);
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if (po[k] >= 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a1_s[(4 * q + r) * CH + po[k]] = fmaxf(acc[k][r], 0.f);
      }
  }
}

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
constexpr int A1SZ = NCH * CH;

__global__ __launch_bounds__(64 * FWD_WAVES) void cnn_fwd_wino_k(CnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* e_s = smem;                       // [2][ESZ]
  float* a1_s = smem + 2 * ESZ;            // [2][16][26][ROW]
  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = FWD_WAVES, NT = 64 * FWD_WAVES;

  float w1f[3];
  int off1[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = 4 * s + q;
    w1f[s] = tap < 9 ? a.W1[j * 9 + tap] : 0.f;
    const int tt = tap < 9 ? tap : 0;
    off1[s] = (tt / 3) * ROWE + tt % 3;
  }
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

  for (int i = tid; i < 2 * ESZ; i += NT) e_s[i] = 0.f;
  // conv1 tile k of this wavefront: (image slot, 16 positions) -> e_s offset (high half) | a1_s offset (low half,
  // 0xFFFF: nothing to store).  a1 offsets of slot 1 exceed 16 bits, so the slot is folded in as s * A1SZ at use.
  int c1off[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const int tile = wave + NW * k;
    const int sl = tile >= 43, pos = 16 * (tile - 43 * sl) + j;
    const bool ok = tile < 86 && pos < C1 * C1;
    const int pc = ok ? pos : 0;
    const int y = pc / C1, x = pc - y * C1;
    c1off[k] = (((tile < 86 ? sl : 0) * ESZ + y * ROWE + x) << 16) | (ok ? (sl << 15) | (y * ROW + x) : 0xFFFF);
  }

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
      if (i < 2 * IMG * IMG) {
        const int s = i >= IMG * IMG, p = i - s * (IMG * IMG);
        e_s[s * ESZ + (p / IMG) * ROWE + p % IMG] = pre[k];
      }
    }
    __syncthreads();
    fetch(pair + gridDim.x);                           // next pair's pixels: in flight under the MFMAs

    // conv1 + ReLU of both images: 86 tiles of 16 consecutive positions, up to 6 in flight per wavefront; the tile ->
    // LDS offsets are the same for every image pair and come packed from c1off (computed once per kernel)
#pragma unroll
    for (int k0 = 0; k0 < 12; k0 += 6) {
      f32x4 acc[6];
      float ev[6][3];
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float* pe = e_s + (c1off[k0 + k] >> 16);
#pragma unroll
        for (int t = 0; t < 3; ++t) ev[k][t] = pe[off1[t]];
        acc[k] = b1v;
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] = mfma(w1f[t], ev[k][t], acc[k]);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int pk = c1off[k0 + k] & 0xFFFF;
        if (pk != 0xFFFF) {
          const int po = (pk >> 15) * A1SZ + (pk & 0x7FFF);
#pragma unroll
          for (int r = 0; r < 4; ++r) a1_s[(4 * q + r) * CH + po] = fmaxf(acc[k][r], 0.f);
        }
      }
    }
    __syncthreads();

#pragma nounroll
    for (int item = wave; item < 18; item += NW) {
      const int s = item >= 9, grp = item - 9 * s;     // wave-uniform
      const int64_t img = 2 * pair + s;
      if (img >= a.n) continue;
      const int t = 16 * grp + j, ty = t / 12, tx = t - 12 * ty;
      const float* base = a1_s + s * A1SZ + q * CH + 2 * ty * ROW + 2 * tx;
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
        plo[rr] = *reinterpret_cast<const f32x2*>(base + rr * ROW);
        phi[rr] = *reinterpret_cast<const f32x2*>(base + rr * ROW + 2);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float vv[16];
        wino_in(plo, phi, vv);
        if (g < 3) {
          const float* p = base + 4 * (g + 1) * CH;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            plo[rr] = *reinterpret_cast<const f32x2*>(p + rr * ROW);
            phi[rr] = *reinterpret_cast<const f32x2*>(p + rr * ROW + 2);
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
        const float v00 = s0[0] + s0[1] + s0[2], v01 = s0[1] - s0[2] - s0[3];
        const float v10 = s1[0] + s1[1] + s1[2], v11 = s1[1] - s1[2] - s1[3];
        float best = v00; int bi = 0;                  // first max wins ties (torch max_pool2d order)
        if (v01 > best) { best = v01; bi = 1; }
        if (v10 > best) { best = v10; bi = 2; }
        if (v11 > best) { best = v11; bi = 3; }
        a.pooled[ob + r * (PO * PO)] = best;
        a.arg[ob + r * (PO * PO)] = (unsigned char)bi;
      }
    }
  }
}

// dY2 planes: channel pairs sit PS dwords apart, the two channels of a pair CHD (== 16 mod 32) apart, so that
//  - the dW2 A-operand read (16 channels x 2 consecutive positions per 32-lane group) and
//  - the da1 B-operand gather (2 channels x 16 consecutive positions per 32-lane group)
// are both bank-conflict free.
constexpr int PS = 2 * CHD + 2;
__device__ __forceinline__ int based(int c) { return (c >> 1) * PS + (c & 1) * CHD; }
constexpr int DSZ = (NCH / 2) * PS;

__global__ __launch_bounds__(64 * BWD_WAVES) void cnn_bwd_k(CnnArgs a) {
#ifdef GNF_CNN_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = __builtin_readcyclecounter();
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* e_s = smem;
  float* a1_s = smem + ESZ;            // conv1 activations; reused for the per-tap planes T after dW2
  float* d_s = a1_s + NCH * CH;             // dY2 with a 2-wide zero border
  float* T_s = a1_s;
  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // provably wave-uniform: scalar branches / SGPR math
  constexpr int NW = BWD_WAVES, NT = 64 * BWD_WAVES;

  float w1f[3];
  int off1[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = 4 * s + q;
    w1f[s] = tap < 9 ? a.W1[j * 9 + tap] : 0.f;
    const int tt = tap < 9 ? tap : 0;
    off1[s] = (tt / 3) * ROWE + tt % 3;
  }
  f32x4 b1v;
#pragma unroll
  for (int r = 0; r < 4; ++r) b1v[r] = a.b1[4 * q + r];
  // W2^T as A operand of the data gradient: row i = j = input channel, K slot q -> output channel 4g+q
  float w2t[36];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < 9; ++k) w2t[g * 9 + k] = a.W2[((4 * g + q) * NCH + j) * 9 + k];
  // W1^T as A operand of the per-tap planes: row i = j = tap, K slot q, step r -> channel 4q+r
  float w1t[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) w1t[r] = j < 9 ? a.W1[(4 * q + r) * 9 + j] : 0.f;
  // per-lane column offsets of the im2col B operand of dW2: column c = 16 nt + j = ic*9 + ky*3 + kx
  int colo[9];
#pragma unroll
  for (int nt = 0; nt < 9; ++nt) {
    const int c = 16 * nt + j;
    colo[nt] = (c / 9) * CH + ((c % 9) / 3) * ROW + c % 3;
  }

  f32x4 gW2[9];
#pragma unroll
  for (int nt = 0; nt < 9; ++nt) gW2[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // dW1 / db1 as per-lane partials over this lane's positions: channel 4q+r, tap k (k = 9: bias)
  float gW1p[4][10];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < 10; ++k) gW1p[r][k] = 0.f;
  float gb2 = 0.f;                                           // thread tid accumulates channel tid/32

  for (int i = tid; i < ESZ; i += NT) e_s[i] = 0.f;
  for (int i = tid; i < DSZ; i += NT) d_s[i] = 0.f;

  // software prefetch: the next image's pixels, pooled gradients and argmax bytes are loaded into
  // registers while the MFMA phases of the current image run
  constexpr int EPT = (IMG * IMG + NT - 1) / NT, WPT = (PO * PO + 31) / 32;
  float epre[EPT], gpre[WPT];
  int apre[WPT];
  auto prefetch = [&](int64_t im) {
    const bool on = im < a.n;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int i = tid + k * NT;
      epre[k] = (on && i < IMG * IMG) ? a.e[im * (IMG * IMG) + i] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
      const int w = (tid & 31) + 32 * k;
      const bool ok = on && w < PO * PO;
      const int64_t o = im * NPOOL + (tid >> 5) * (PO * PO) + w;
      gpre[k] = ok ? a.gp[o] : 0.f;
      apre[k] = ok ? (int)a.argin[o] : 0;
    }
  };
  prefetch(blockIdx.x);
  int woff[WPT];                                            // this thread's pool windows: fixed LDS offsets
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int w = (tid & 31) + 32 * k;
    woff[k] = w < PO * PO ? based(tid >> 5) + (2 * (w / PO) + 2) * ROWD + 2 * (w % PO) + 2 : -1;
  }
  int eoff[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int i = tid + k * NT;
    eoff[k] = i < IMG * IMG ? (i / IMG) * ROWE + i % IMG : -1;
  }

  for (int64_t img = blockIdx.x; img < a.n; img += gridDim.x) {
    __syncthreads();
    // ---- P0: image, and dY2 = pool-backward scatter of g_pooled (one write per conv2 position)
#pragma unroll
    for (int k = 0; k < EPT; ++k)
      if (eoff[k] >= 0) e_s[eoff[k]] = epre[k];
#pragma unroll
    for (int k = 0; k < WPT; ++k)
      if (woff[k] >= 0) {
        const float g = gpre[k];
        const int am = apre[k];
        gb2 += g;
        float* p = d_s + woff[k];
        p[0] = am == 0 ? g : 0.f;
        p[1] = am == 1 ? g : 0.f;
        p[ROWD] = am == 2 ? g : 0.f;
        p[ROWD + 1] = am == 3 ? g : 0.f;
      }
    __syncthreads();
    prefetch(img + gridDim.x);
    TSTAMP(0);
    // ---- P1: recompute conv1 + ReLU (43 flat tiles in pairs: pair p = wave + 8k' owns tiles 2p, 2p+1); the ReLU gate
    //      bits stay in registers: the da1 tiles below use the same tile -> lane mapping
    unsigned gate = 0u;
    {
      int po[6];
      f32x4 acc[6];
      float ev[6][3];
#pragma unroll
      for (int k = 0; k < 6; ++k) {                           // all operand reads first ...
        const int pos = 16 * (2 * (wave + NW * (k >> 1)) + (k & 1)) + j;     // tile 2p + (k&1), pair p = wave + 8(k>>1)
        const int pc = pos < C1 * C1 ? pos : 0;
        const int y = pc / C1, x = pc - y * C1;
        po[k] = pos < C1 * C1 ? y * ROW + x : -1;
#pragma unroll
        for (int s = 0; s < 3; ++s) ev[k][s] = e_s[y * ROWE + x + off1[s]];
        acc[k] = b1v;
      }
#pragma unroll
      for (int s = 0; s < 3; ++s)                             // ... then 6 independent MFMA chains
#pragma unroll
        for (int k = 0; k < 6; ++k) acc[k] = mfma(w1f[s], ev[k][s], acc[k]);
#pragma unroll
      for (int k = 0; k < 6; ++k)
        if (po[k] >= 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            a1_s[(4 * q + r) * CH + po[k]] = fmaxf(acc[k][r], 0.f);
            if (acc[k][r] > 0.f) gate |= 1u << (4 * k + r);
          }
        }
    }
    __syncthreads();
    TSTAMP(1);
    // ---- P3: dW2[oc][c] += sum_pos dY2[oc][pos] * a1[ic(c)][pos + tap(c)];  K = 576 positions, 72 (3 rows) per wave.
    //      K slots of one 32-lane group are columns x and x+3: their im2col gathers (kx = 0..2) hit disjoint banks.
    {
      // K slot q covers column xs + 3(q&1) + 12(q>>1): the lane-dependent part folds into per-lane base
      // pointers, the step-dependent part (row s/6, xs in {0,1,2,6,7,8}) is an immediate offset -> no VALU
      // address arithmetic inside the 18 unrolled steps.
      const int lq = 3 * (q & 1) + 12 * (q >> 1);
      const float* ap = d_s + based(j) + (3 * wave + 2) * ROWD + lq + 2;
      const float* bp = a1_s + 3 * wave * ROW + lq;
#pragma unroll
      for (int s = 0; s < 18; ++s) {
        const int xs = (s % 6) < 3 ? (s % 6) : (s % 6) + 3;
        const float av = ap[(s / 6) * ROWD + xs];
#pragma unroll
        for (int nt = 0; nt < 9; ++nt) gW2[nt] = mfma(av, bp[(s / 6) * ROW + xs + colo[nt]], gW2[nt]);
      }
    }
    __syncthreads();                                           // a1 as im2col operand is done: region becomes T
    TSTAMP(2);
    // ---- P4: dpre1 = conv2^T(dY2) * gate on the 26x26 grid (same tile pairs, both tiles in flight).  Epilogue on the
    //      VALU, straight from the accumulator registers: dW1/db1 partials and the per-tap planes
    //      T[tap][pos] = sum_oc W1[oc][tap] dpre1[oc][pos] (lane sum over its 4 channels, then over the 4 lane slots)
#pragma nounroll
    for (int kk = 0; kk < 3; ++kk) {
      const int tA = 2 * (wave + NW * kk);
      if (tA >= 43) break;                                      // wave-uniform (scalar): waves 6,7 own two pairs
      const int posA = 16 * tA + j, posB = posA + 16;
      const bool okA = posA < C1 * C1, okB = posB < C1 * C1;
      const int pcA = okA ? posA : 0, pcB = okB ? posB : 0;
      const int yA = pcA / C1, xA = pcA - yA * C1, yB = pcB / C1, xB = pcB - yB * C1;
      const float* pA = d_s + based(q) + (yA + 2) * ROWD + xA + 2;
      const float* pB = d_s + based(q) + (yB + 2) * ROWD + xB + 2;
      f32x4 accA = f32x4{0.f, 0.f, 0.f, 0.f}, accB = accA;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int o = g * 2 * PS - ky * ROWD - kx;
            accA = mfma(w2t[g * 9 + ky * 3 + kx], pA[o], accA);
            accB = mfma(w2t[g * 9 + ky * 3 + kx], pB[o], accB);     // tile 43 does not exist: masked below
          }
#ifdef GNF_CNN_EXP_NOEPI
      asm volatile("" ::"v"(accA[0]), "v"(accA[1]), "v"(accA[2]), "v"(accA[3]), "v"(accB[0]), "v"(accB[1]), "v"(accB[2]), "v"(accB[3]));
      if (false)
#endif
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const bool ok = half ? okB : okA;
        const int y = half ? yB : yA, x = half ? xB : xA, pos = half ? posB : posA;
        f32x4 dp = half ? accB : accA;
#pragma unroll
        for (int r = 0; r < 4; ++r) dp[r] = (ok && ((gate >> (8 * kk + 4 * half + r)) & 1u)) ? dp[r] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) gW1p[r][9] += dp[r];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float ev = e_s[(y + tap / 3) * ROWE + x + tap % 3];
#pragma unroll
          for (int r = 0; r < 4; ++r) gW1p[r][tap] = fmaf(dp[r], ev, gW1p[r][tap]);
        }
        // T = W1^T[tap x oc] * dpre1[oc x pos]: dpre1 in the C/D layout IS the B operand (K slot q, step r -> oc 4q+r)
        f32x4 tt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) tt = mfma(w1t[r], dp[r], tt);
        // branch-free store: lanes without a valid (tap, pos) write to a scratch word behind the planes
#pragma unroll
        for (int r = 0; r < 4; ++r) T_s[(ok && 4 * q + r < 9) ? (4 * q + r) * CS + pos : 9 * CS + lane] = tt[r];
      }
    }
    __syncthreads();
    TSTAMP(3);
    // ---- de[y][x] = sum_tap T[tap][y-ky][x-kx]
    for (int i = tid; i < IMG * IMG; i += NT) {
      const int y = i / IMG, x = i - y * IMG;
      float s = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yy = y - ky, xx = x - kx;
          if (yy >= 0 && yy < C1 && xx >= 0 && xx < C1) s += T_s[(ky * 3 + kx) * CS + yy * C1 + xx];
        }
      a.ge[img * (IMG * IMG) + i] = s;
    }
    TSTAMP(4);
  }
#ifdef GNF_CNN_TIMING
  if (blockIdx.x == 7 && (tid & 63) == 0)
    for (int k = 0; k < 6; ++k) a.part[((int64_t)gridDim.x * NW + 1) * PROW + wave * 8 + k] = (float)tacc[k];
#endif

  // ---- per-wave partial row: dW2 [16][144] | dW1+db1 [16][16] | db2 [16]
  float* prow = a.part + ((int64_t)blockIdx.x * NW + wave) * PROW;
#pragma unroll
  for (int nt = 0; nt < 9; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[(4 * q + r) * 144 + 16 * nt + j] = gW2[nt][r];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      float v = gW1p[r][k];                                    // sum over the 16 position lanes
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (j == 0) prow[NCH * 144 + (4 * q + r) * 16 + k] = v;
    }
  if (j == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int k = 10; k < 16; ++k) prow[NCH * 144 + (4 * q + r) * 16 + k] = 0.f;
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) gb2 += __shfl_xor(gb2, off, 64);    // over the 32 threads of a channel
  // wave w owns channels 2w, 2w+1 (lanes 0 and 32); the other 14 db2 slots of its row are zero
  if ((lane & 31) == 0) prow[NCH * 144 + NCH * 16 + 2 * wave + (lane >> 5)] = gb2;
  if (lane < NCH && (lane >> 1) != wave) prow[NCH * 144 + NCH * 16 + lane] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// Backward with both conv2-sized contractions in the Winograd F(2x2,3x3) domain (2.25x fewer MFMAs):
//   forward   Y = A^T [ sum_c U[o][c] (.) V[c] ] A,  U = G w G^T,  V = B^T d B   (d = 4x4 patch of a1)
//   dW2:      dU[o][c] = sum_tiles (A dY A^T)[o] (.) V[c],  dw = G^T dU G once at the end of the kernel.
//             The 2x2 output tile is the pool window, so dY has ONE non-zero g at the saved argmax (py,px):
//             A dY A^T = g * alpha_py alpha_px^T with alpha_0 = (1,1,1,0), alpha_1 = (0,1,-1,-1) -- read
//             straight from g_pooled/argmax, no transform.  GEMM per xi: M = o, N = c, K = tiles (144/image).
//   da1:      a 3x3 valid correlation of the zero-bordered dY2 (28x28) with the flipped kernel
//             w'[c][o][a][b] = W2[o][c][2-a][2-b] -> 13x13 tiles of 2x2; U' = G w' G^T lives in LDS (A operand),
//             the lane transforms its own dY2 patch (B operand); the output transform, ReLU gate, dW1/db1
//             partials and the per-tap planes T are lane-local as in the direct kernel.
// conv1 is recomputed in the same tile -> lane mapping (lane (q,j): channels 4q..4q+3 of tile j, 4 sub-positions),
// so its ReLU gate bits stay in registers for da1.
// ---------------------------------------------------------------------------------------------
constexpr int NG4 = 11;                              // groups of 16 da1 tiles (13 x 13 = 169 tiles of 2x2)
constexpr int USZ = 16 * 4 * 64;                     // U' as [xi_y][g][lane][xi_x]
// LDS layouts of this kernel.  Every gather is a ds_read_b64 whose 16-lane (ds_read2) / 32-lane groups must spread
// over the 64 banks; with the direct kernel's strides the dW2 gather was 4-way and the da1 gather 2-way conflicted
// and both phases LDS-bound (measured: 4.4k of 11.5k and ~5k of 17k cycles per image).
//  a1 [16][26][ROWB]: the dW2 gather reads 16 CHANNELS x one column pair per 16-lane group -> CHB*j mod 64 must be
//     16 distinct even banks (CHB = 2 * odd).
//  dY2 (zero-bordered 28x28) per channel as [14 tile rows][TRD] with the odd image row at +ROD: the da1 gather reads
//     16 consecutive 2x2 TILES (13 per tile row) of 2 channels per 32-lane group -> consecutive tiles are +2 dwords,
//     the tile-row wrap TRD - 24 == 2 (mod 64), the two channels of a pair CHDW == 32 (mod 64) apart; channel pairs
//     PSD == 2 (mod 64) apart keep the 16-channel Z read of dW2 conflict-free too.
constexpr int ROWB = 28, CHB = 730;
constexpr int TRD = 90, ROD = 44, CHDW = 1312, PSD = 2 * CHDW + 2, DSZW = (NCH / 2) * PSD;
static_assert(CHB >= C1 * ROWB && (CHB % 4) == 2, "a1 channel stride");
static_assert((TRD - 24) % 64 == 2 && CHDW % 64 == 32 && PSD % 64 == 2 && 13 * TRD + ROD + 28 <= CHDW, "dY2 layout");
__device__ __forceinline__ int dofs(int c) { return (c >> 1) * PSD + (c & 1) * CHDW; }

__global__ __launch_bounds__(64 * BWD_WAVES) void cnn_bwd_wino_k(CnnArgs a) {
#ifdef GNF_CNN_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = __builtin_readcyclecounter();
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* e_s = smem;
  float* a1_s = smem + ESZ;                 // conv1 activations; reused for the per-tap planes T after dW2
  float* d_s = a1_s + NCH * CHB;            // dY2 with a 2-wide zero border, tile-row layout
  float* u_s = d_s + DSZW;                  // U' as [g][xi_y][lane][xi_x]
  float* T_s = a1_s;
  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = BWD_WAVES, NT = 64 * BWD_WAVES;
  static_assert((ESZ % 4 == 0) && ((NCH * CHB) % 4 == 0) && (DSZW % 4 == 0), "u_s must be 16-B aligned");
  static_assert(9 * CS + 64 <= NCH * CHB, "T planes alias the a1 region");

  float w1f[3];
  int off1[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = 4 * s + q;
    w1f[s] = tap < 9 ? a.W1[j * 9 + tap] : 0.f;
    const int tt = tap < 9 ? tap : 0;
    off1[s] = (tt / 3) * ROWE + tt % 3;
  }
  f32x4 b1v;
#pragma unroll
  for (int r = 0; r < 4; ++r) b1v[r] = a.b1[4 * q + r];
  float w1t[4];                              // W1^T as A operand of the per-tap planes: row = tap j, K slot q, step r
#pragma unroll
  for (int r = 0; r < 4; ++r) w1t[r] = j < 9 ? a.W1[(4 * q + r) * 9 + j] : 0.f;

  // U'[c = j][o = 4g+q] = G w' G^T, w'[a][b] = W2[o][c][2-a][2-b]; fp64 once, stored for one ds_read_b128 per (g, xi_y)
  if (wave < 4) {
    const int g = wave;
    const float* w = a.W2 + ((4 * g + q) * NCH + j) * 9;
    double gw[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double w0 = w[8 - c], w1 = w[5 - c], w2 = w[2 - c];         // rows a = 0,1,2 of the flipped kernel, column b = c
      gw[0][c] = w0; gw[1][c] = 0.5 * (w0 + w1 + w2); gw[2][c] = 0.5 * (w0 - w1 + w2); gw[3][c] = w2;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 u;
      u[0] = (float)gw[r][0];
      u[1] = (float)(0.5 * (gw[r][0] + gw[r][1] + gw[r][2]));
      u[2] = (float)(0.5 * (gw[r][0] - gw[r][1] + gw[r][2]));
      u[3] = (float)gw[r][2];
      *reinterpret_cast<f32x4*>(u_s + ((g * 4 + r) * 64 + lane) * 4) = u;
    }
  }

  // dW2 in the Winograd domain, split by xi_y over the two wavefronts of a SIMD: wavefronts 0-3 own xi_y in {0,1},
  // wavefronts 4-7 xi_y in {2,3}; each covers all 36 K-steps of an image with its three partners (s = wave&3 mod 4).
  // dU[4*(xi_y & 1) + xi_x][r] = dU_xi[o = 4q+r][c = j]
  const int hy = wave >> 2;                  // wave-uniform
  f32x4 dU[8];
#pragma unroll
  for (int xi = 0; xi < 8; ++xi) dU[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x2 gW1p[2][10];                         // dW1 / db1 per-lane partials: channels 4q+2h, 4q+2h+1 (packed), tap k (9: bias)
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 10; ++k) gW1p[h][k] = f32x2{0.f, 0.f};
  float gb2 = 0.f;                           // thread tid accumulates channel tid/32

  for (int i = tid; i < ESZ; i += NT) e_s[i] = 0.f;
  for (int i = tid; i < DSZW; i += NT) d_s[i] = 0.f;

  constexpr int EPT = (IMG * IMG + NT - 1) / NT, WPT = (PO * PO + 31) / 32;
  float epre[EPT], gpre[WPT];
  unsigned apre[WPT];                        // raw loads only: any arithmetic here would wait for the data and make
  auto prefetch = [&](int64_t im) {          // the prefetch synchronous
    const bool on = im < a.n;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int i = tid + k * NT;
      epre[k] = (on && i < IMG * IMG) ? a.e[im * (IMG * IMG) + i] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
      const int w = (tid & 31) + 32 * k;
      const bool ok = on && w < PO * PO;
      const int64_t o = im * NPOOL + (tid >> 5) * (PO * PO) + w;
      gpre[k] = ok ? a.gp[o] : 0.f;
      apre[k] = ok ? (unsigned)a.argin[o] : 0u;
    }
  };
  prefetch(blockIdx.x);
  // this thread's pool windows (channel tid/32, window (tid&31) + 32k) and pixels: fixed LDS offsets
  const int dwin = dofs(tid >> 5) + TRD + 2;
  int woff[WPT];
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int w = (tid & 31) + 32 * k;
    woff[k] = w < PO * PO ? dwin + (w / PO) * TRD + 2 * (w % PO) : -1;
  }

  for (int64_t img = blockIdx.x; img < a.n; img += gridDim.x) {
    __syncthreads();
    // ---- P0: image, and dY2 = pool-backward scatter of g_pooled (one write per conv2 position)
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int i = tid + k * NT;
      if (i < IMG * IMG) e_s[(i / IMG) * ROWE + i % IMG] = epre[k];
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k)
      if (woff[k] >= 0) {
        const float g = gpre[k];
        const int am = (int)apre[k];
        gb2 += g;
        float* p = d_s + woff[k];
        p[0] = am == 0 ? g : 0.f;
        p[1] = am == 1 ? g : 0.f;
        p[ROD] = am == 2 ? g : 0.f;
        p[ROD + 1] = am == 3 ? g : 0.f;
      }
    __syncthreads();
    prefetch(img + gridDim.x);
    TSTAMP(0);
    // ---- P1: conv1 + ReLU in the 2x2-tile layout: group grp = wave + 8k, 4 sub-positions = 4 MFMA chains
    unsigned gate = 0u;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int grp = wave + NW * k;                            // wave-uniform
      if (grp < NG4) {
        const int t = 16 * grp + j;
        const bool ok = t < 169;
        const int tc = ok ? t : 0, ty = tc / 13, tx = tc - 13 * ty;
        f32x4 acc[4];
        float ev[4][3];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
          for (int s = 0; s < 3; ++s) ev[p][s] = e_s[(2 * ty + (p >> 1)) * ROWE + 2 * tx + (p & 1) + off1[s]];
          acc[p] = b1v;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
          for (int p = 0; p < 4; ++p) acc[p] = mfma(w1f[s], ev[p][s], acc[p]);
        if (ok) {
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              a1_s[(4 * q + r) * CHB + (2 * ty + (p >> 1)) * ROWB + 2 * tx + (p & 1)] = fmaxf(acc[p][r], 0.f);
              if (acc[p][r] > 0.f) gate |= 1u << (16 * k + 4 * p + r);
            }
        }
      }
    }
    __syncthreads();
    TSTAMP(1);
    // ---- P3: dU_xi[o][c] += sum_tiles Z_xi[o][tile] V_xi[c][tile]; K-step s = 4 tiles; this wavefront's xi_y half.
    //      Z = A dY A^T from the 2x2 window of dY2 in LDS (A = [[1,0],[1,1],[1,-1],[0,-1]]); V = B^T d B needs the
    //      patch rows hy..hy+2 only: xi_y 0,1 = d0-d2, d1+d2;  xi_y 2,3 = d2-d1, d1-d3
#pragma nounroll
    for (int s = wave & 3; s < 36; s += 4) {
      const int T = 4 * s + q, ty = T / 12, tx = T - 12 * ty;
      const float* p = a1_s + j * CHB + (2 * ty + hy) * ROWB + 2 * tx;
      const float* pz = d_s + dofs(j) + (ty + 1) * TRD + 2 * tx + 2;
      const float2 y0 = *reinterpret_cast<const float2*>(pz);
      const float2 y1 = *reinterpret_cast<const float2*>(pz + ROD);
      f32x2 dl[3], dh[3];                                        // patch rows hy..hy+2 as two column pairs
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        dl[rr] = *reinterpret_cast<const f32x2*>(p + rr * ROWB);
        dh[rr] = *reinterpret_cast<const f32x2*>(p + rr * ROWB + 2);
      }
      f32x2 tal, tah, tbl, tbh;                                  // the two rows of B^T d of this half
      float za[2], zb[2];
      if (hy == 0) {
        tal = dl[0] - dl[2]; tah = dh[0] - dh[2]; tbl = dl[1] + dl[2]; tbh = dh[1] + dh[2];
        za[0] = y0.x; za[1] = y0.y; zb[0] = y0.x + y1.x; zb[1] = y0.y + y1.y;
      } else {
        tal = dl[1] - dl[0]; tah = dh[1] - dh[0]; tbl = dl[0] - dl[2]; tbh = dh[0] - dh[2];
        za[0] = y0.x - y1.x; za[1] = y0.y - y1.y; zb[0] = -y1.x; zb[1] = -y1.y;
      }
      // all operands first, then 8 back-to-back MFMAs: VALU and MFMA of ONE wavefront do not overlap
      // (tools/mfma_feed.hip); a VALU op in front of every MFMA would stall the pipe for both wavefronts
      float vv[8], zz[8];
      {
        const f32x2 a03 = tal - tah, a12 = pk_v12(tal, tah), b03 = tbl - tbh, b12 = pk_v12(tbl, tbh);
        vv[0] = a03.x; vv[1] = a12.x; vv[2] = a12.y; vv[3] = a03.y;
        vv[4] = b03.x; vv[5] = b12.x; vv[6] = b12.y; vv[7] = b03.y;
      }
      zz[0] = za[0]; zz[1] = za[0] + za[1]; zz[2] = za[0] - za[1]; zz[3] = -za[1];
      zz[4] = zb[0]; zz[5] = zb[0] + zb[1]; zz[6] = zb[0] - zb[1]; zz[7] = -zb[1];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int xi = 0; xi < 8; ++xi) dU[xi] = mfma(zz[xi], vv[xi], dU[xi]);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                           // a1 as dW2 operand is done: region becomes T
    TSTAMP(2);
    // ---- P4: dpre1 = conv2^T(dY2) * gate on 2x2 tiles: per 4 input channels (g) 16 operands, then 16 MFMAs
#pragma nounroll
    for (int k = 0; k < 2; ++k) {
      const int grp = wave + NW * k;                            // wave-uniform
      if (grp < NG4) {
        const int t = 16 * grp + j;
        const bool ok = t < 169;
        const int tc = ok ? t : 0, ty = tc / 13, tx = tc - 13 * ty;
        const float* pd = d_s + ty * TRD + 2 * tx;
        f32x4 m[16];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) m[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int RO4[4] = {0, ROD, TRD, TRD + ROD};           // patch row a at (a>>1)*TRD + (a&1)*ROD
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float* pp = pd + dofs(4 * g + q);
          f32x2 plo[4], phi[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            plo[rr] = *reinterpret_cast<const f32x2*>(pp + RO4[rr]);
            phi[rr] = *reinterpret_cast<const f32x2*>(pp + RO4[rr] + 2);
          }
          f32x4 uf[4];
#pragma unroll
          for (int xy = 0; xy < 4; ++xy) uf[xy] = *reinterpret_cast<const f32x4*>(u_s + ((g * 4 + xy) * 64 + lane) * 4);
          float vv[16];
          wino_in(plo, phi, vv);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int xi = 0; xi < 16; ++xi) m[xi] = mfma(uf[xi >> 2][xi & 3], vv[xi], m[xi]);
          __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 dp[4];                                            // dpre1 at sub-position p, channels 4q+r
#pragma unroll
        for (int r = 0; r < 4; ++r) {                           // A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]]
          float s0[4], s1[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            s0[c] = m[c][r] + m[4 + c][r] + m[8 + c][r];
            s1[c] = m[4 + c][r] - m[8 + c][r] - m[12 + c][r];
          }
          const float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
          const float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
          const unsigned gb = ok ? (gate >> (16 * k + r)) : 0u;
          dp[0][r] = (gb & 1u) ? y00 : 0.f;
          dp[1][r] = (gb & 16u) ? y01 : 0.f;
          dp[2][r] = (gb & 256u) ? y10 : 0.f;
          dp[3][r] = (gb & 4096u) ? y11 : 0.f;
        }
        // dW1 / db1 partials against the 4x4 image patch of this tile
        float ep[4][4];
        {
          const float* pe = e_s + 2 * ty * ROWE + 2 * tx;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const float2 lo = *reinterpret_cast<const float2*>(pe + rr * ROWE);
            const float2 hi = *reinterpret_cast<const float2*>(pe + rr * ROWE + 2);
            ep[rr][0] = lo.x; ep[rr][1] = lo.y; ep[rr][2] = hi.x; ep[rr][3] = hi.y;
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {                           // packed over the channel pair (2 flops per lane per op)
          f32x2 d2[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) d2[p] = f32x2{dp[p][2 * h], dp[p][2 * h + 1]};
          gW1p[h][9] += (d2[0] + d2[1]) + (d2[2] + d2[3]);
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            f32x2 acc = gW1p[h][tap];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const float ev = ep[(p >> 1) + ky][(p & 1) + kx];
              acc = __builtin_elementwise_fma(d2[p], f32x2{ev, ev}, acc);
            }
            gW1p[h][tap] = acc;
          }
        }
        // T[tap][pos] = sum_oc W1[oc][tap] dpre1[oc][pos]: dpre1 in the C/D layout IS the B operand
        f32x4 tq[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) tq[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int p = 0; p < 4; ++p) tq[p] = mfma(w1t[r], dp[p][r], tq[p]);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int pos = (2 * ty + (p >> 1)) * C1 + 2 * tx + (p & 1);
#pragma unroll
          for (int r = 0; r < 4; ++r) T_s[(ok && 4 * q + r < 9) ? (4 * q + r) * CS + pos : 9 * CS + lane] = tq[p][r];
        }
      }
    }
    __syncthreads();
    TSTAMP(3);
    // ---- de[y][x] = sum_tap T[tap][y-ky][x-kx]
    for (int i = tid; i < IMG * IMG; i += NT) {
      const int y = i / IMG, x = i - y * IMG;
      float s = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yy = y - ky, xx = x - kx;
          if (yy >= 0 && yy < C1 && xx >= 0 && xx < C1) s += T_s[(ky * 3 + kx) * CS + yy * C1 + xx];
        }
      a.ge[img * (IMG * IMG) + i] = s;
    }
    TSTAMP(4);
  }
#ifdef GNF_CNN_TIMING
  if (blockIdx.x == 7 && (tid & 63) == 0)
    for (int k = 0; k < 6; ++k) a.part[((int64_t)gridDim.x * NW + 1) * PROW + wave * 8 + k] = (float)tacc[k];
#endif

  // ---- per-wave partial row: dW2 [16][144] | dW1+db1 [16][16] | db2 [16]
  float* prow = a.part + ((int64_t)blockIdx.x * NW + wave) * PROW;
#pragma unroll
  for (int r = 0; r < 4; ++r) {          // this half's share of dw = G^T dU G for (o = 4q+r, c = j); the halves add up
    float ar[3][4];                      // G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const float ua = dU[x][r], ub = dU[4 + x][r];            // xi_y = 2hy, 2hy+1
      if (hy == 0) { ar[0][x] = ua + 0.5f * ub; ar[1][x] = 0.5f * ub; ar[2][x] = 0.5f * ub; }
      else { ar[0][x] = 0.5f * ua; ar[1][x] = -0.5f * ua; ar[2][x] = 0.5f * ua + ub; }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float* o = prow + (4 * q + r) * 144 + j * 9 + 3 * i;
      o[0] = ar[i][0] + 0.5f * (ar[i][1] + ar[i][2]);
      o[1] = 0.5f * (ar[i][1] - ar[i][2]);
      o[2] = 0.5f * (ar[i][1] + ar[i][2]) + ar[i][3];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      float v = gW1p[r >> 1][k][r & 1];                        // sum over the 16 tile lanes
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (j == 0) prow[NCH * 144 + (4 * q + r) * 16 + k] = v;
    }
  if (j == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int k = 10; k < 16; ++k) prow[NCH * 144 + (4 * q + r) * 16 + k] = 0.f;
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) gb2 += __shfl_xor(gb2, off, 64);    // over the 32 threads of a channel
  if ((lane & 31) == 0) prow[NCH * 144 + NCH * 16 + 2 * wave + (lane >> 5)] = gb2;
  if (lane < NCH && (lane >> 1) != wave) prow[NCH * 144 + NCH * 16 + lane] = 0.f;
}

// unpack the summed partial row into the parameter-shaped gradients
__global__ void cnn_unpack_k(const float* __restrict__ vec, float* gW1, float* gb1, float* gW2, float* gb2) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= PROW) return;
  const float s = vec[n];
  if (n < NCH * 144) gW2[n] = s;
  else if (n < NCH * 144 + NCH * 16) {
    const int k = n - NCH * 144, oc = k >> 4, c = k & 15;
    if (c < 9) gW1[oc * 9 + c] = s;
    else if (c == 9) gb1[oc] = s;
  } else gb2[n - NCH * 144 - NCH * 16] = s;
}

constexpr size_t kFwdLds = (size_t)(ESZ + NCH * CH) * sizeof(float);
constexpr size_t kWinoLds = (size_t)(2 * ESZ + 2 * A1SZ) * sizeof(float);
constexpr unsigned kWinoGrid = 256;                  // one 8-wave workgroup per CU, two images per iteration
constexpr size_t kBwdLds = (size_t)(ESZ + NCH * CH + DSZ) * sizeof(float);
constexpr size_t kBwdWinoLds = (size_t)(ESZ + NCH * CHB + DSZW + USZ) * sizeof(float);
// one 8-wave workgroup per CU: at its 128 VGPRs a second one is not admitted (measured with tools/census.hip and
// the occupancy API; the 96-VGPR variant that admits two spills and is slower)
constexpr unsigned kFwdGrid = 512, kBwdGrid = 256;   // 512 measured faster than 256 for the forward

}  // namespace

extern "C" {

#ifdef GNF_CNN_TIMING
int gnf_debug_occupancy_lds(int lds) {
  int nb = -1;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cnn_fwd_k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, cnn_fwd_k, 64 * FWD_WAVES, (size_t)lds);
  return nb;
}
int gnf_debug_occupancy(int which) {
  int nb = -1;
  if (which == 0) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cnn_fwd_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFwdLds);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, cnn_fwd_k, 64 * FWD_WAVES, kFwdLds);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cnn_bwd_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdLds);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, cnn_bwd_k, 64 * BWD_WAVES, kBwdLds);
  }
  return nb;
}
int gnf_debug_fwd_start(long long* host1024) { return (int)hipMemcpyFromSymbol(host1024, HIP_SYMBOL(g_fwd_start), 1024 * sizeof(long long)); }
int gnf_debug_fwd_timing(float* host64) { return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_fwd_timing), 64 * sizeof(float)); }
#endif

int gnf_mnistcnn_conv_fwd(const float* e, const float* W1, const float* b1, const float* W2, const float* b2,
                          float* pooled, unsigned char* argmax, int64_t n_img, int exact_ties, gnf_stream_t stream) {
  if (!e || !W1 || !b1 || !W2 || !b2 || !pooled || !argmax || n_img < 0) return GNF_EINVAL;
  if (n_img == 0) return 0;
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

int64_t gnf_mnistcnn_conv_bwd_ws_bytes(int64_t n_img) {
  (void)n_img;
  return ((int64_t)kBwdGrid * BWD_WAVES + 2) * PROW * (int64_t)sizeof(float);
}

int gnf_mnistcnn_conv_bwd(const float* e, const float* W1, const float* b1, const float* W2, const float* g_pooled,
                          const unsigned char* argmax, float* ge, float* gW1, float* gb1, float* gW2, float* gb2,
                          void* ws, int64_t ws_bytes, int64_t n_img, gnf_stream_t stream) {
  if (!e || !W1 || !b1 || !W2 || !g_pooled || !argmax || !ge || !gW1 || !gb1 || !gW2 || !gb2 || !ws || n_img < 0)
    return GNF_EINVAL;
  if (ws_bytes < gnf_mnistcnn_conv_bwd_ws_bytes(n_img)) return GNF_EWS;
  CnnArgs a{};
  a.e = e; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.gp = g_pooled; a.argin = argmax; a.ge = ge; a.part = (float*)ws;
  a.n = n_img;
  // fixed grid: every workgroup (also one without images) writes its partial rows
#ifdef GNF_CNN_DIRECT_BWD
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cnn_bwd_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kBwdLds);
  hipLaunchKernelGGL(cnn_bwd_k, dim3(kBwdGrid), dim3(64 * BWD_WAVES), kBwdLds, (hipStream_t)stream, a);
#else
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cnn_bwd_wino_k),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdWinoLds);
  hipLaunchKernelGGL(cnn_bwd_wino_k, dim3(kBwdGrid), dim3(64 * BWD_WAVES), kBwdWinoLds, (hipStream_t)stream, a);
#endif
  GNF_LAUNCH_CHECK();
  const int64_t rows = (int64_t)kBwdGrid * BWD_WAVES;
  float* vec = (float*)ws + rows * PROW;
  const int rc = gnf_rowsum_launch((const float*)ws, vec, rows, PROW, 0, (hipStream_t)stream);
  if (rc) return rc;
  hipLaunchKernelGGL(cnn_unpack_k, dim3((PROW + 255) / 256), dim3(256), 0, (hipStream_t)stream, vec, gW1, gb1, gW2,
                     gb2);
  GNF_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
