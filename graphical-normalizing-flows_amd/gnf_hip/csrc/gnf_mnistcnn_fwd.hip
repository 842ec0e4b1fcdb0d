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
// This file: the FORWARD kernels (Winograd, and the direct tie-exact one).  Both conv units are built with
// -fno-slp-vectorize (gnf_hip/build.py): the SLP vectoriser packs the scalar adds of the output transform into v_pk_add_f32
// and pays for it with 77 v_mov per 64 MFMAs to form the register pairs (forward 1.44 -> 1.40 ms at cfg4; the round-4
// backward 2.89 -> 2.86 ms -- the round-3 backward was faster WITH it).
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

// ---------------------------------------------------------------------------------------------
// Round 4: ONE image per barrier interval, two a1 buffers -- the next image's conv1 runs UNDER this image's Winograd items
// (rounds 1-3: two images per iteration, conv1 in a barrier interval of its own).  Per interval i
// (p = i & 1):   items of image i from a1[p]  |  conv1(i+1): e[1-p] -> a1[1-p]  |  stage e(i+2) -> e[p]  |  the
// previous image's split item is finished.  9 items + 11 conv1 units (~0.3 item each) over 4 SIMDs: every wavefront one
// whole item (group w), group 8 split by xi_y halves over wavefronts 0 and 1, the units dealt to the others:
//   SIMD 0 (w0, w4): 1.5 + 1 items + 2 units     SIMD 2 (w2, w6): 2 items + 4 units
//   SIMD 1 (w1, w5): 1.5 + 1 items + 2 units     SIMD 3 (w3, w7): 2 items + 3 units
// (1/1/5/4 units measured: the same within the +-3 % run-to-run spread of this kernel, as is the round-3 structure with two
// images per iteration and conv1 in a barrier interval of its own: 1.34-1.42 ms in every variant on one box -- the kernel
// is bound by its ALU work, 64 MFMAs + ~260 other VALU instructions per item at two wavefronts per SIMD.)
// and of the two wavefronts of a SIMD one starts with its units (latency-bound), the other with its item (MFMA-bound).
// Outputs leave through per-image buffer descriptors (one 32-bit lane offset, no 64-bit address arithmetic).
// ---------------------------------------------------------------------------------------------
#ifndef GNF_FWD_F1U
#define GNF_FWD_F1U {0, 0, 3, 2, 1, 1, 2, 2}     // by barrier-wait timing (tools/time_cnn_phases.py): {0,0,2,2,2,2,2,1} left SIMD 0 (wavefronts 0 + 4: the split item, its finish and two units) 10 % behind the others: 1.331 -> 1.272 ms
#endif
#ifndef GNF_FWD_SPLIT_WAVE
#define GNF_FWD_SPLIT_WAVE 0
#endif
__device__ constexpr int F1U[8] = GNF_FWD_F1U;                       // conv1 units of the next image per wavefront
__device__ constexpr int F1PRO[8] = {2, 2, 2, 1, 1, 1, 1, 1};        // the first image: dealt evenly
constexpr int fsum(const int (&v)[8]) { int t = 0; for (int i = 0; i < 8; ++i) t += v[i]; return t; }
static_assert(fsum(F1U) == 11 && fsum(F1PRO) == 11, "conv1 units");
constexpr int fmax8(const int (&v)[8]) { int t = 0; for (int i = 0; i < 8; ++i) t = v[i] > t ? v[i] : t; return t; }
static_assert(fmax8(F1U) <= 3 && fmax8(F1PRO) <= 3, "at most three units per wavefront (conv1_do, the packed offsets)");

__global__ __launch_bounds__(64 * FWD_WAVES) void cnn_fwd_wino_k(CnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* e_s = smem;                       // [2][28 x 28]
  float* a1_s = smem + 2 * WESZ;           // [2][16][26][WROW]
  float* xch = a1_s + 2 * A1SZ;            // [2 images][2 halves][16][64]: partial output transforms of the split item
  float* w1_s = xch + 2 * 2 * 16 * 64;     // W1 as [tap][channel]: conv1's A operands, re-read per call
  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = FWD_WAVES, NT = 64 * FWD_WAVES;
  typedef __amdgpu_buffer_rsrc_t rsrc_t;

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

  auto rsrc_of = [&](const void* base, int64_t im, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(base)) + im * bytes, 0,
                                             im < a.n ? bytes : 0, 0x00020000);
  };
  constexpr int EPT = (IMG * IMG + NT - 1) / NT;
  float pre[EPT];
  auto fetch = [&](int64_t im) {           // raw loads only; out-of-range lanes / images read 0
    const rsrc_t rs = rsrc_of(a.e, im, IMG * IMG * 4);
#pragma unroll
    for (int k = 0; k < EPT; ++k) pre[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (tid + k * NT) * 4, 0, 0));
  };
  auto stage = [&](float* dst) {
#pragma unroll
    for (int k = 0; k < EPT; ++k)
      if (tid + k * NT < IMG * IMG) dst[tid + k * NT] = pre[k];
  };
  auto half2 = [](const f32x4& v, int pr) { return pr ? f32x2{v[2], v[3]} : f32x2{v[0], v[1]}; };
  // 2x2 max pool of (channel 4q+r, tile t) into the image's outputs; first max wins ties (torch max_pool2d order)
  auto pool_out = [&](rsrc_t rp, rsrc_t ra, int t, int r, float v00, float v01, float v10, float v11) {
    float best = v00; int bi = 0;
    if (v01 > best) { best = v01; bi = 1; }
    if (v10 > best) { best = v10; bi = 2; }
    if (v11 > best) { best = v11; bi = 3; }
    const int o = (4 * q + r) * (PO * PO) + t;
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, best), rp, o * 4, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bi, ra, o, 0, 0);
  };
  auto finish_split = [&](const float* xb0, int64_t im) {      // xb0: [2 halves][16][64] of the item (group 8)
    const rsrc_t rp = rsrc_of(a.pooled, im, NPOOL * 4), ra = rsrc_of(a.arg, im, NPOOL);
    const float* xa = xb0 + lane;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = xa[(r * 4 + k) * 64] + xa[16 * 64 + (r * 4 + k) * 64];
      pool_out(rp, ra, 16 * 8 + j, r, v[0], v[1], v[2], v[3]);
    }
  };
  auto conv1_do = [&](const float* e_rd, float* a1_wr, int u0, int nu) {     // nu wave-uniform
    if (nu == 3) conv1_units<3, WCH>(e_rd, a1_wr, u0, w1_s + j, b1v, q, j, lane);
    else if (nu == 2) conv1_units<2, WCH>(e_rd, a1_wr, u0, w1_s + j, b1v, q, j, lane);
    else if (nu == 1) conv1_units<1, WCH>(e_rd, a1_wr, u0, w1_s + j, b1v, q, j, lane);
  };
  // one whole item: group grp of 16 tiles of the image in a1
  auto whole_item = [&](const float* a1, int grp, int64_t im) {
    const int t = 16 * grp + j, ty = t / 12, tx = t - 12 * ty;
    const float* base = a1 + q * WCH + 2 * ty * WROW + 2 * tx;
    f32x4 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) acc[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[5] = b2v;                                      // xi = (1,1) reaches all four outputs with weight +1: the bias
    f32x2 plo[4], phi[4];                              // patch rows as two column pairs
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
    const rsrc_t rp = rsrc_of(a.pooled, im, NPOOL * 4), ra = rsrc_of(a.arg, im, NPOOL);
    // output transform A^T M A (A^T = [[1,1,1,0],[0,1,-1,-1]]) + pool, two channels per instruction: the halves of an
    // accumulator are an aligned register pair, so the 24 additions per channel are 12 v_pk_add_f32 per channel PAIR (this
    // unit is built without the SLP vectoriser; same operation order per component, so the same bits)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      f32x2 s0[4], s1[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x2 m0 = half2(acc[c], pr), m1 = half2(acc[4 + c], pr), m2 = half2(acc[8 + c], pr), m3 = half2(acc[12 + c], pr);
        s0[c] = m0 + m1 + m2;
        s1[c] = m1 - m2 - m3;
      }
      const f32x2 y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
      const f32x2 y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
      pool_out(rp, ra, t, 2 * pr, y00.x, y01.x, y10.x, y11.x);
      pool_out(rp, ra, t, 2 * pr + 1, y00.y, y01.y, y10.y, y11.y);
    }
  };
  // half of the split item (group 8): xi_y in {2 hf, 2 hf + 1}; partial output transform into xb
  auto half_item = [&](const float* a1, int hf, float* xb0) {
    const int t = 16 * 8 + j, ty = t / 12, tx = t - 12 * ty;
    const float* base = a1 + q * WCH + 2 * ty * WROW + 2 * tx;
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
    float* xb = xb0 + hf * 16 * 64 + lane;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      f32x2 p0[4], p1[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x2 m0 = half2(acc[c], pr), m1 = half2(acc[4 + c], pr);
        p0[c] = hf == 0 ? m0 + m1 : m0;
        p1[c] = hf == 0 ? m1 : -m0 - m1;
      }
      const f32x2 y0 = p0[0] + p0[1] + p0[2], y1 = p0[1] - p0[2] - p0[3], y2 = p1[0] + p1[1] + p1[2], y3 = p1[1] - p1[2] - p1[3];
      xb[(2 * pr * 4 + 0) * 64] = y0.x; xb[(2 * pr * 4 + 1) * 64] = y1.x; xb[(2 * pr * 4 + 2) * 64] = y2.x; xb[(2 * pr * 4 + 3) * 64] = y3.x;
      xb[((2 * pr + 1) * 4 + 0) * 64] = y0.y; xb[((2 * pr + 1) * 4 + 1) * 64] = y1.y;
      xb[((2 * pr + 1) * 4 + 2) * 64] = y2.y; xb[((2 * pr + 1) * 4 + 3) * 64] = y3.y;
    }
  };

  // ---- prologue: images 0 and 1 staged, conv1 of image 0 dealt evenly, image 2 requested
  const int64_t img0 = blockIdx.x, gs = gridDim.x;
  fetch(img0); stage(e_s);
  fetch(img0 + gs); stage(e_s + WESZ);
  fetch(img0 + 2 * gs);
  __syncthreads();
  {
    int u0 = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) u0 += w < wave ? F1PRO[w] : 0;
    if (img0 < a.n) conv1_do(e_s, a1_s, u0, F1PRO[wave]);
  }
  int c1u0 = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) c1u0 += w < wave ? F1U[w] : 0;
  // the steady-state units of this wavefront are the same for every image: their LDS offsets once, packed (conv1_units_pre;
  // 1.272 -> 1.244 ms: ~40 instructions of address arithmetic per unit and image were 17 % of the kernel's VALU work)
  unsigned upo[3][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) conv1_offsets<WCH>(c1u0 + (k < F1U[wave] ? k : 0), q, j, lane, upo[k]);
  auto conv1_steady = [&](const float* e_rd, float* a1_wr) {
    const int nu = F1U[wave];
    if (nu == 3) conv1_units_pre<3, WCH>(e_rd, a1_wr, upo, w1_s + j, b1v);
    else if (nu == 2) conv1_units_pre<2, WCH>(e_rd, a1_wr, upo, w1_s + j, b1v);
    else if (nu == 1) conv1_units_pre<1, WCH>(e_rd, a1_wr, upo, w1_s + j, b1v);
  };
#ifndef GNF_FWD_UNITS_FIRST
#define GNF_FWD_UNITS_FIRST (wave >= 4)
#endif
  const bool units_first = GNF_FWD_UNITS_FIRST;        // one wavefront of every SIMD starts with conv1, the other with its item

  int par = 0;
  int64_t prev = -1;                                   // image whose split item waits in xch[par ^ 1]
#ifdef GNF_CNN_TIMING      // measurement build (tools/time_cnn_phases.py): cycle counts per phase and wavefront
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = __builtin_readcyclecounter();
#define FSTAMP(k) do { const long long t__ = __builtin_readcyclecounter(); tacc[k] += t__ - tlast; tlast = t__; } while (0)
#else
#define FSTAMP(k)
#endif
  for (int64_t img = img0; img < a.n; img += gs, par ^= 1) {
    FSTAMP(5);
    __syncthreads();                                   // a1[par] complete; a1[par^1], e[par] free; the previous halves in xch
    FSTAMP(0);
    const float* a1p = a1_s + par * A1SZ;
    float* a1n = a1_s + (par ^ 1) * A1SZ;
    const bool has_next = img + gs < a.n;
    stage(e_s + par * WESZ);                           // image i+2 (requested one interval ago)
    fetch(img + 3 * gs);
    if (wave == GNF_FWD_SPLIT_WAVE && prev >= 0) finish_split(xch + (par ^ 1) * 2 * 16 * 64, prev);
    FSTAMP(1);
    if (has_next && units_first) conv1_steady(e_s + (par ^ 1) * WESZ, a1n);
    FSTAMP(2);
    whole_item(a1p, wave, img);
    FSTAMP(3);
    if (wave < 2) half_item(a1p, wave, xch + par * 2 * 16 * 64);
    FSTAMP(4);
    if (has_next && !units_first) conv1_steady(e_s + (par ^ 1) * WESZ, a1n);
    FSTAMP(2);
    prev = img;
  }
  __syncthreads();                                     // the last image's halves
  if (wave == GNF_FWD_SPLIT_WAVE && prev >= 0) finish_split(xch + (par ^ 1) * 2 * 16 * 64, prev);
#ifdef GNF_CNN_TIMING
  if (blockIdx.x == 7 && lane == 0)                    // (overwrites the head of image 0's output: measurement build only)
    for (int k = 0; k < 8; ++k) a.pooled[wave * 8 + k] = (float)tacc[k];
#endif
#undef FSTAMP
}

constexpr size_t kFwdLds = (size_t)(ESZ + NCH * CH) * sizeof(float);
constexpr size_t kWinoLds = (size_t)(2 * WESZ + 2 * A1SZ + 2 * 2 * 16 * 64 + 9 * NCH) * sizeof(float);   // + the split items' exchange, the W1 table
static_assert(kWinoLds <= 160 * 1024, "one workgroup per CU");
constexpr unsigned kWinoGrid = 256;                  // one 8-wave workgroup per CU
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
    const unsigned grid = n_img < kWinoGrid ? (unsigned)n_img : kWinoGrid;
    hipLaunchKernelGGL(cnn_fwd_wino_k, dim3(grid), dim3(64 * FWD_WAVES), kWinoLds, (hipStream_t)stream, a);
  }
  GNF_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
