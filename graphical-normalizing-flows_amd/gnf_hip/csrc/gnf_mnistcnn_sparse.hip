// Sparse masked-image front of the MNISTCNN embedding net for a DETERMINISTIC DAG gate (SURVEY.md 8(f)1;
// models/Conditionners/DAGConditioner.py:142-169 feeding models/MLP.py:36-48).
//
// With the MNIST adjacency prior (NormalizingFlowFactories.py:35-46, kernel = 2) row i of the importance matrix is
// non-zero only inside the 5x5 window around pixel i, and a zero entry never becomes non-zero again (its gradient is
// 0).  Under a deterministic gate the masked copy e_i = x * P[i] is therefore EXACTLY zero outside that window, so
//   conv1 + ReLU deviates from the constant relu(b1) only on 7x7 positions,
//   conv2 (no activation, MLP.py:38-40) deviates from its constant background bg[c] only on 9x9 positions,
//   the 12x12 pooled map deviates from bg[c] on at most 5x5 cells.
// Those cells always lie inside the 5x5 cell block at (r0, c0) = clamp(floor((pixel row/col - 6) / 2), 0, 7), i.e. in
// what the same CNN computes from the 14x14 CROP of e_i at (2 r0, 2 c0):  conv 14 -> 12 -> 10, pool -> 5x5 x 16 ch.
// fc1 is linear, so  fc1(pooled) = [b + W . bg]  +  W[:, block(r0, c0)] . (pooled_block - bg):  a 400-wide GEMM against
// one of 64 column blocks of the fc1 weight instead of the 2304-wide one.  Per masked copy: 0.24 MMAC instead of 1.72.
// The stochastic (Gumbel) gate leaks ~1e-6 e^(g1-g2) of every pixel through, occasionally O(1): it stays on the dense
// kernels (gnf_mnistcnn.hip).
//
// gfx950 mapping (forward, and the backward w.r.t. the network parameters for training with a FROZEN deterministic
// gate -- gradients w.r.t. x or P are not produced here, such a step stays on the dense pair):
//   crop kernel: ONE WAVEFRONT per masked copy, everything in that wave's 10 KB of LDS -> 12 waves per CU.
//     conv1 as 9 x 3 and conv2 as 7 x 36 v_mfma_f32_16x16x4_f32 (direct convolution as implicit GEMM: M = 16 output
//     channels, N = 16 positions, K = taps / (tap, input channel)); the weights are the A operand and stay in registers
//     for the whole kernel, the B operand is one ds_read_b32 per MFMA at an immediate offset.  conv2's N index is
//     (pool cell, position in the cell), so the 2x2 max-pool is two quad DPP ops on the accumulator and one lane per
//     quad stores 4 channels of a cell as a float4: rows of `pd` are [cell][channel].
//   fc1: the 64 column blocks are gathered once per call into k-major images Wg[64][400][F]; masked copies arrive
//     sorted by block, so ONE grouped launch of the fp32 MFMA GEMM (gnf_gemm.hip) does bias + ReLU for all of them.
//   backward: d pd = g . Wg^T and dWg = pd^T . g as grouped GEMMs (row groups / K groups), dWg scattered back to the
//     fc1 weight; the crop backward kernel (one wavefront per masked copy again) recomputes conv1, rebuilds the
//     max-pooled cotangent from the saved argmax, and contracts dW2 (K = the 100 conv2 positions), d a1 (K = (output
//     channel, tap), only on the 8x8 box of conv1 positions around the pixel's window) and dW1/db1 (K = the box,
//     against the crop and a constant-1 image) on MFMA with
//     the accumulators in registers across all copies of the wave.  What the constant background contributes (bg
//     depends on b1, W2, b2) is a closed form of the column sums and is added by a one-workgroup epilogue.
#include <cstdlib>
#include "gnf_common.h"
#include "gnf_gemm.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int IMG = 28, NPIX = IMG * IMG, NCH = 16;
constexpr int CROP = 14, ES = 16, ESZ = CROP * ES;     // crop of the masked image, row stride 16
constexpr int A1 = 12, PL = A1 * A1;                   // conv1 activations [16][12][12]; PL = 144 == 16 mod 64 banks
constexpr int NCELL = 25, KD = NCELL * NCH;            // 5x5 pool cells x 16 channels = 400 deviation features
constexpr int NORIG = 64;                              // crop origins (r0, c0) in 0..7 x 0..7
constexpr int WAVES = 4;
constexpr int WLDS = ESZ + NCH * PL;                   // floats of LDS per wavefront

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// Winograd F(2x2,3x3) input transform B^T d B of a 4x4 patch given as 4 rows x 2 column pairs; out[xi = 4 xi_y + xi_x] (the dense
// kernels' routine, gnf_mnistcnn.h: two v_pk_add_f32 per row of B^T d)
__device__ __forceinline__ f32x2 pk_v12(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void wino_in(const f32x2 (&lo)[4], const f32x2 (&hi)[4], float (&v)[16]) {
  f32x2 tl[4], th[4];
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
__device__ __forceinline__ float quad_max(float v) {   // max over the 4 lanes of a quad
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));   // [1,0,3,2]
  return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false))); // [2,3,0,1]
}
__host__ __device__ __forceinline__ int crop_origin(int p) {   // cell row / column origin of pixel coordinate p
  int o = (p - 6) >> 1;                                        // arithmetic shift: floor
  return o < 0 ? 0 : (o > 7 ? 7 : o);
}

// bg[c] = b2[c] + sum_{c',tap} W2[c][c'][tap] relu(b1[c']);  hbg[n] = bfc1[n] + sum_{c,pos} Wfc1[n][c*144+pos] bg[c]
__global__ void sparse_bg_k(const float* __restrict__ b1, const float* __restrict__ W2, const float* __restrict__ b2,
                            const float* __restrict__ Wfc1, const float* __restrict__ bfc1, int F,
                            float* __restrict__ bg, float* __restrict__ hbg) {
  __shared__ float sbg[NCH];
  if (threadIdx.x < NCH) {
    const int c = threadIdx.x;
    float s = b2[c];
    for (int ci = 0; ci < NCH; ++ci) {
      const float a = fmaxf(b1[ci], 0.f);
      for (int t = 0; t < 9; ++t) s = fmaf(W2[(c * NCH + ci) * 9 + t], a, s);
    }
    sbg[c] = s;
    if (blockIdx.x == 0) bg[c] = s;
  }
  __syncthreads();
  // every workgroup recomputes the 16 background values (2304 fma) and reduces 4 rows of Wfc1
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6, wave = blockIdx.x * nw + (threadIdx.x >> 6);
  if (!hbg) return;
  for (int n = wave; n < F; n += nw * gridDim.x) {
    float s = 0.f;
    for (int k = lane; k < NCH * 144; k += 64) s = fmaf(Wfc1[(int64_t)n * (NCH * 144) + k], sbg[k / 144], s);
    s = group_sum<64>(s);
    if (lane == 0) hbg[n] = bfc1[n] + s;
  }
}

// Wg[g][k = cell*16 + c][n] = Wfc1[n][c*144 + (r0 + cy)*12 + c0 + cx],  g = r0*8 + c0, cell = cy*5 + cx
__global__ void sparse_gather_fc1_k(const float* __restrict__ Wfc1, int F, float* __restrict__ Wg) {
  const int64_t total = (int64_t)NORIG * KD * F;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(idx % F);
    const int k = (int)((idx / F) % KD), g = (int)(idx / ((int64_t)F * KD));
    const int c = k & 15, cell = k >> 4, cy = cell / 5, cx = cell - 5 * cy;
    const int r0 = g >> 3, c0 = g & 7;
    Wg[idx] = Wfc1[(int64_t)n * (NCH * 144) + c * 144 + (r0 + cy) * 12 + c0 + cx];
  }
}

struct SparseArgs {
  const float* x; const float* P; const int32_t* pix;
  const float* W1; const float* b1; const float* W2; const float* b2; const float* bg;
  float* pd; unsigned char* arg; int64_t B, items;
};

__device__ __forceinline__ int quad_min(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false));
  return min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));
}

// SAVE: also record, per (cell, channel), which of the 4 positions won the max-pool (first maximum in scan order, what
// torch's max_pool2d backward uses) -- one byte each, same [cell][channel] layout as pd
//
// SPLIT (inference on few copies -- a level of a sampling pass is ~700 of them, less than three per CU): the FOUR wavefronts
// of a workgroup share one masked copy -- conv1's nine position tiles dealt 3/2/2/2, conv2's seven 2/2/2/1, two barriers --
// so a copy's ~280 dependent MFMAs are a quarter as deep (the one-wavefront form is the throughput form: no barrier at all).
// One wavefront per copy WITHOUT the argmax record (!SPLIT && !SAVE: evaluation of many copies under no_grad): conv2 in the
// WINOGRAD F(2x2,3x3) domain, as in the dense kernels.  NOT for the training form: a deterministic gate leaves exactly constant
// regions in a crop, whose pool windows are EXACT ties; the direct form gives bit-equal outputs for equal patches, so its first
// maximum is torch's (max_pool2d's backward routes the cotangent there) -- the four outputs of a Winograd tile round differently
// and the tie breaks elsewhere (conv1.weight gradient 1.3e-3 off in test_sparse_front_parameter_gradients).  Values agree to
// rounding either way, which is all the inference form returns (the dense path has the same split: exact_ties).  The
// 10 x 10 conv2 output of the crop is 5 x 5 tiles of 2 x 2 = the 25 pool cells: per transform point xi a 16 x 16 x 16 product
// M_xi[o][tile] = sum_c U_xi[o][c] V_xi[c][tile] over two N tiles (cells 0..15, 16..24) -- 128 MFMAs per copy instead of 7 x 36, 64
// ds_read_b64 of a1 instead of 252 ds_read_b32, and the D layout leaves all 16 xi of one (4 channels, cell) in ONE lane, so the
// output transform, the bias (added at xi = (1,1)), the max-pool and its first-max argmax are lane-local (no DPP).  The filter
// transforms U = G w G^T are computed in fp64 once per wavefront.  64 + 64 registers of U' and accumulators: two workgroups per CU.
template <bool SAVE, bool SPLIT>
__global__ __launch_bounds__(64 * WAVES, (SPLIT || SAVE) ? 3 : 2) void sparse_crop_k(SparseArgs a) {
  constexpr bool WINO = !SPLIT && !SAVE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  float* e_s = smem + (SPLIT ? 0 : wave * WLDS);
  float* a1_s = e_s + ESZ;
  constexpr int N1 = SPLIT ? 3 : 9, N2 = SPLIT ? 2 : 7, NT = SPLIT ? 1 : 4;
  const int nb1 = SPLIT ? (wave ? 2 * wave + 1 : 0) : 0, n1 = SPLIT ? (wave ? 2 : 3) : 9;   // conv1 tiles [nb1, nb1 + n1)
  const int nb2 = SPLIT ? 2 * wave : 0, n2 = SPLIT ? (wave == 3 ? 1 : 2) : 7;              // conv2 tiles [nb2, nb2 + n2)

  // ---- weights as MFMA A operands: lane (j = output channel, q = K slot)
  float wa1[3], wa2[WINO ? 1 : 36], uw[WINO ? 64 : 1];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = 4 * s + q;
    wa1[s] = tap < 9 ? a.W1[j * 9 + tap] : 0.f;
  }
  if constexpr (!WINO) {
#pragma unroll
    for (int s = 0; s < 36; ++s) wa2[s] = a.W2[(j * NCH + 4 * (s & 3) + q) * 9 + (s >> 2)];
  } else {
    // U = G w G^T of W2[o = j][c = 4 g + q], G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]: uw[xi * 4 + g], xi = 4 xi_y + xi_x
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
  }
  f32x4 bias1, bias2, bgv;
#pragma unroll
  for (int r = 0; r < 4; ++r) { bias1[r] = a.b1[4 * q + r]; bias2[r] = a.b2[4 * q + r]; bgv[r] = a.bg[4 * q + r]; }

  // ---- per-lane LDS offsets, independent of the masked copy
  int eo[N1][3];                  // conv1 B operand: e_s offset of (position 16 nb + j, tap 4 s + q)
#pragma unroll
  for (int k = 0; k < N1; ++k) {
    const int nb = nb1 + (k < n1 ? k : 0);
    const int p = 16 * nb + j, y = p / A1, x = p - A1 * y;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int tap = 4 * s + q, ty = tap < 9 ? tap / 3 : 0, tx = tap < 9 ? tap - 3 * ty : 0;
      eo[k][s] = (y + ty) * ES + x + tx;
    }
  }
  int po[N2];                     // conv2 B operand: a1_s offset of (cell 4 nb + j/4, sub-position j%4, channel slot q)
#pragma unroll
  for (int k = 0; k < N2; ++k) {
    const int nb = nb2 + (k < n2 ? k : 0);
    int cell = 4 * nb + (j >> 2);
    cell = cell < NCELL ? cell : NCELL - 1;
    const int cy = cell / 5, cx = cell - 5 * cy;
    po[k] = q * PL + (2 * cy + ((j >> 1) & 1)) * A1 + 2 * cx + (j & 1);
  }
  int wb[2];                      // Winograd: a1_s offset of the 4x4 patch of (cell 16 nt + j, channel slot q)
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    int cell = 16 * nt + j;
    cell = cell < NCELL ? cell : NCELL - 1;
    const int cy = cell / 5, cx = cell - 5 * cy;
    wb[nt] = q * PL + 2 * cy * A1 + 2 * cx;
  }
  const int a1w = 4 * q * PL + j;                              // conv1 D store: channel 4q + r, position 16 nb + j

  // ---- the crop of e = x * P[pixel] of one masked copy: 196 values, 4 per lane (the last lane group idles on the 4th;
  //      SPLIT: one per thread of the workgroup)
  int ce[NT], cl[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int idx = lane + 64 * (SPLIT ? wave : t), ey = idx / CROP, ex = idx - CROP * ey;
    ce[t] = ey * IMG + ex;                                     // offset inside the image, relative to the crop corner
    cl[t] = idx < CROP * CROP ? ey * ES + ex : -1;
  }
  float xv[NT], pv[NT];
  auto fetch = [&](int64_t item) {
    const int64_t r = item / a.B, b = item - r * a.B;
    const int pix = a.pix[r];
    const int corner = 2 * crop_origin(pix / IMG) * IMG + 2 * crop_origin(pix % IMG);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const bool ok = cl[t] >= 0;
      xv[t] = ok ? a.x[b * NPIX + corner + ce[t]] : 0.f;
      pv[t] = ok ? a.P[(int64_t)pix * NPIX + corner + ce[t]] : 0.f;
    }
  };

  const int64_t stride = SPLIT ? (int64_t)gridDim.x : (int64_t)gridDim.x * WAVES;
  int64_t item = SPLIT ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * WAVES + wave;
  if (item < a.items) fetch(item);
  for (; item < a.items; item += stride) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (cl[t] >= 0) e_s[cl[t]] = xv[t] * pv[t];
    if (item + stride < a.items) fetch(item + stride);         // next copy's loads fly during this one's MFMAs
    if (SPLIT) __syncthreads();

    // ---- conv1 + ReLU -> a1_s
#pragma unroll
    for (int k = 0; k < N1; ++k) {
      if (!SPLIT || k < n1) {
        f32x4 d = bias1;
#pragma unroll
        for (int s = 0; s < 3; ++s) d = mfma(wa1[s], e_s[eo[k][s]], d);
#pragma unroll
        for (int r = 0; r < 4; ++r) a1_s[a1w + r * PL + 16 * (nb1 + k)] = fmaxf(d[r], 0.f);
      }
    }
    if (SPLIT) __syncthreads();
    // ---- conv2 + 2x2 max-pool + bias - background -> pd[item][cell][channel]
    float* prow = a.pd + item * KD + 4 * q;
    if constexpr (WINO) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x4 acc[16];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) acc[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[5] = bias2;                                          // xi = (1,1) reaches all four outputs with weight +1
        const float* pb = a1_s + wb[nt];
        f32x2 plo[4], phi[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          plo[rr] = *reinterpret_cast<const f32x2*>(pb + rr * A1);
          phi[rr] = *reinterpret_cast<const f32x2*>(pb + rr * A1 + 2);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float vv[16];
          wino_in(plo, phi, vv);
          if (g < 3) {                                           // the next channel step's patch: in flight under the MFMAs
            const float* p = pb + 4 * (g + 1) * PL;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              plo[rr] = *reinterpret_cast<const f32x2*>(p + rr * A1);
              phi[rr] = *reinterpret_cast<const f32x2*>(p + rr * A1 + 2);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int xi = 0; xi < 16; ++xi) acc[xi] = mfma(uw[xi * 4 + g], vv[xi], acc[xi]);
          __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                            // A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]]; pool
          float s0[4], s1[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            s0[c] = acc[c][r] + acc[4 + c][r] + acc[8 + c][r];
            s1[c] = acc[4 + c][r] - acc[8 + c][r] - acc[12 + c][r];
          }
          const float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
          const float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
          v[r] = fmaxf(fmaxf(y00, y01), fmaxf(y10, y11)) - bgv[r];
        }
        const int cell = 16 * nt + j;
        if (cell < NCELL) *reinterpret_cast<f32x4*>(prow + cell * NCH) = v;
      }
    }
#pragma unroll
    for (int k = 0; k < (WINO ? 0 : N2); ++k) {
      if (!SPLIT || k < n2) {
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;              // two chains: consecutive MFMAs are independent
        const float* bp = a1_s + po[k];
#pragma unroll
        for (int s = 0; s < 36; s += 2) {
          const int t0 = s >> 2, t1 = (s + 1) >> 2;
          d0 = mfma(wa2[s], bp[4 * (s & 3) * PL + (t0 / 3) * A1 + t0 % 3], d0);
          d1 = mfma(wa2[s + 1], bp[4 * ((s + 1) & 3) * PL + (t1 / 3) * A1 + t1 % 3], d1);
        }
        f32x4 v;
        unsigned am = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pre = d0[r] + d1[r], mx = quad_max(pre);
          v[r] = mx + bias2[r] - bgv[r];
          if (SAVE) am |= (unsigned)quad_min(pre == mx ? (j & 3) : 4) << (8 * r);
        }
        const int cell = 4 * (nb2 + k) + (j >> 2);
        if ((j & 3) == 0 && cell < NCELL) {
          *reinterpret_cast<f32x4*>(prow + cell * NCH) = v;
          if (SAVE) *reinterpret_cast<unsigned*>(a.arg + item * KD + cell * NCH + 4 * q) = am;
        }
      }
    }
    if (SPLIT) __syncthreads();                                // (a persistent workgroup: the images are rewritten next)
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the crop network w.r.t. W1, b1, W2 (b2 and the background terms are closed forms, see sparse_finish_k).
// One wavefront (= one workgroup) per masked copy.  LDS: the crop e (row stride 16) followed by a constant-1 image of
// the same shape, conv1 activations / their cotangent [16][12 rows x stride 14] (plane 172 == 44 mod 64: the 16 lanes
// of an operand read sit 4 banks apart), the conv2 cotangent with a 2-wide zero border [16][14][14] (plane 196 == 4).
// ---------------------------------------------------------------------------------------------------------------
constexpr int A1S = 14, BPL = 172, DPL = 196, ONES = ESZ, EB = 2 * ESZ;
constexpr int BLDS = EB + NCH * BPL + NCH * DPL;            // 6336 floats = 25 KB per wavefront: 6 per CU
constexpr int PROW = NCH * NCH * 9 + NCH * 9 + 2 * NCH;     // partial row: dW2 | dW1 | db1 (box) | sum of d a1 (box)
constexpr int BWD_GRID = 256 * 8;             // upper bound of the crop-backward grid (partials)

struct SparseBwdArgs {
  const float* x; const float* P; const int32_t* pix;
  const float* W1; const float* b1; const float* W2;
  const float* dpd; const unsigned char* arg;
  float* part; int64_t B, items;
};

// TWO wavefronts per masked copy (round 5; the one-wavefront form of rounds 1-4 -- same arithmetic, profiles/r05_sparse_bwd2.txt --
// was deleted in round 6).  One wavefront per copy held 256 registers and 25 KB of LDS:
// six copies per CU, 1.5 wavefronts per SIMD, and its ~450 MFMAs per copy are one dependent stream (0.61 of the MFMA-bound
// time).  Here the copy's work is two streams that meet at barriers:
//   wavefront 0: stage e  | conv1 -> a1, bump F_a1 | wait F_dy | dW2 taps 0..6 (210 MFMAs)                                   | barrier
//   wavefront 1: stage dY, bump F_dy | d a1 MFMAs of the box (144, need dY only) | wait F_a1 | gates, dW2 taps 7..8, dW1/db1 | barrier
// (F_a1, F_dy: two LDS counters -- ONE workgroup barrier per copy, at its end, before the images are rewritten)
// The gated d a1 of the box goes to its own 4 KB buffer (in place it would race with the other wavefront's dW2 reads of a1):
// 29 KB per copy, five copies = ten wavefronts per CU.  Same arithmetic per accumulator chain: the same bits per copy; the
// partial rows are summed by the same reduction.
constexpr int BOXS = NCH * 64;                               // gated d a1 of the 8 x 8 box, [16][64]
constexpr int BLDS2 = BLDS + BOXS + 4;                      // + two counters

#ifndef GNF_SPARSE_BWD2_T0
#define GNF_SPARSE_BWD2_T0 7               // 7 + 2 taps: conv1 + 7 taps = 237 MFMAs against d a1 + 2 taps + dW1 + the gates ~ 250
#endif
static_assert(GNF_SPARSE_BWD2_T0 >= 5 && GNF_SPARSE_BWD2_T0 <= 8, "role 1 keeps at least one tap and at most four");
__global__ __launch_bounds__(128, 2) void sparse_crop_bwd2_k(SparseBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), q = lane >> 4, j = lane & 15;
  float* e_s = smem;
  float* a1_s = smem + EB;
  float* dy_s = a1_s + NCH * BPL;
  float* bx_s = dy_s + NCH * DPL;
  unsigned* flag = reinterpret_cast<unsigned*>(bx_s + BOXS);      // [0]: copies whose a1 is complete, [1]: copies whose dY is staged
  if (threadIdx.x < 2) flag[threadIdx.x] = 0u;
  for (int i = threadIdx.x; i < ESZ; i += 128) e_s[ONES + i] = 1.f;
  for (int i = threadIdx.x; i < NCH * DPL; i += 128) dy_s[i] = 0.f;
  for (int i = threadIdx.x; i < NCH * BPL; i += 128) a1_s[i] = 0.f;

  // ---- role 0: conv1 weights, positions;  role 1: W2 as A[c_in][(tap, c_out)], box positions, cotangent scatter offsets.
  //      The loop constants of the two roles SHARE their registers (a wavefront has one role for its whole life; declared
  //      apart, the allocator keeps both sets live across the copy loop: 256 registers and scratch):
  //      wreg: wa1[3] | wf[36];   ireg: pe[9], pao[9], ce[4], cl[4] | dyo[7], pb[4];   freg: xv[4], pv[4] | gv[7], av[7] (as bits)
  float wreg[36];
  int ireg[26];
  float freg[14];
  f32x4 bias1 = {0.f, 0.f, 0.f, 0.f};
  int to[3] = {0, 0, 0};
#define WA1(s_) wreg[s_]
#define WF(s_) wreg[s_]
#define PE(n_) ireg[n_]
#define PAO(n_) ireg[9 + (n_)]
#define CE(t_) ireg[18 + (t_)]
#define CL(t_) ireg[22 + (t_)]
#define DYO(t_) ireg[t_]
#define PB(n_) ireg[7 + (n_)]
#define XV(t_) freg[t_]
#define PV(t_) freg[4 + (t_)]
#define GV(t_) freg[t_]
#define AV(t_) __float_as_int(freg[7 + (t_)])
  if (wave == 0) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int tap = 4 * s + q, ty = tap < 9 ? tap / 3 : 0, tx = tap < 9 ? tap - 3 * ty : 0;
      WA1(s) = tap < 9 ? a.W1[j * 9 + tap] : 0.f;
      to[s] = ty * ES + tx;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) bias1[r] = a.b1[4 * q + r];
#pragma unroll
    for (int nb = 0; nb < 9; ++nb) {
      const int p = 16 * nb + j, y = p / A1, x = p - A1 * y;
      PAO(nb) = y * A1S + x;
      PE(nb) = y * ES + x;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = lane + 64 * t, ey = idx / CROP, ex = idx - CROP * ey;
      CE(t) = ey * IMG + ex;
      CL(t) = idx < CROP * CROP ? ey * ES + ex : -1;
    }
  } else {
#pragma unroll
    for (int s = 0; s < 36; ++s) WF(s) = a.W2[((4 * (s & 3) + q) * NCH + j) * 9 + (s >> 2)];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
      const int idx = lane + 64 * t, cell = idx >> 4, c = idx & 15, cy = cell / 5, cx = cell - 5 * cy;
      DYO(t) = idx < KD ? c * DPL + (2 * cy + 2) * A1S + 2 * cx + 2 : -1;
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) PB(nb) = (2 * nb + (j >> 3)) * A1S + (j & 7);
  }
  const int eb = j < 9 ? (j / 3) * ES + j % 3 + q : (j == 9 ? ONES + q : q);   // dW1 B operand: tap j | ones (db1) | unused

  constexpr int T0 = GNF_SPARSE_BWD2_T0;     // dW2 taps of role 0 (0 .. T0-1); role 1 takes T0 .. 8 in acc2[0 .. 8-T0]
  f32x4 acc2[T0], acc1a = {0.f, 0.f, 0.f, 0.f}, acc1b = acc1a, dsum = acc1a;
#pragma unroll
  for (int t = 0; t < T0; ++t) acc2[t] = acc1a;

  auto fetch = [&](int64_t item) {
    if (wave == 0) {
      const int64_t r = item / a.B, b = item - r * a.B;
      const int pix = a.pix[r];
      const int corner = 2 * crop_origin(pix / IMG) * IMG + 2 * crop_origin(pix % IMG);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool ok = CL(t) >= 0;
        XV(t) = ok ? a.x[b * NPIX + corner + CE(t)] : 0.f;
        PV(t) = ok ? a.P[(int64_t)pix * NPIX + corner + CE(t)] : 0.f;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        const bool ok = DYO(t) >= 0;
        GV(t) = ok ? a.dpd[item * KD + lane + 64 * t] : 0.f;
        freg[7 + t] = __int_as_float(ok ? (int)a.arg[item * KD + lane + 64 * t] : 0);
      }
    }
  };
  __syncthreads();                                               // the constant images exist

  int64_t item = blockIdx.x;
  unsigned nth = 0;
  if (item < a.items) fetch(item);
  for (; item < a.items; item += gridDim.x) {
    if (wave == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (CL(t) >= 0) e_s[CL(t)] = XV(t) * PV(t);
    } else {
#pragma unroll
      for (int t = 0; t < 7; ++t)
        if (DYO(t) >= 0) {
          f32x2 r0, r1;
          r0.x = AV(t) == 0 ? GV(t) : 0.f; r0.y = AV(t) == 1 ? GV(t) : 0.f;
          r1.x = AV(t) == 2 ? GV(t) : 0.f; r1.y = AV(t) == 3 ? GV(t) : 0.f;
          *reinterpret_cast<f32x2*>(dy_s + DYO(t)) = r0;
          *reinterpret_cast<f32x2*>(dy_s + DYO(t) + A1S) = r1;
        }
    }
    if (item + gridDim.x < a.items) fetch(item + gridDim.x);
    int boxo, boxe;
    {
      const int pix = __builtin_amdgcn_readfirstlane(a.pix[item / a.B]);
      const int yi = pix / IMG, xi = pix - IMG * yi;
      int oy = yi - 4 - 2 * crop_origin(yi), ox = xi - 4 - 2 * crop_origin(xi);
      oy = oy < 0 ? 0 : (oy > 4 ? 4 : oy);
      ox = ox < 0 ? 0 : (ox > 4 ? 4 : ox);
      boxo = oy * A1S + ox;
      boxe = oy * ES + ox;
    }
    ++nth;                                                       // this copy's ordinal: what the two flags count up to
    if (wave == 1 && lane == 0) __hip_atomic_fetch_add(flag + 1, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);   // dY staged
    if (wave == 0) {
      // ---- conv1 + ReLU (recomputed) -> a1_s
#pragma unroll
      for (int nb = 0; nb < 9; ++nb) {
        f32x4 d = bias1;
#pragma unroll
        for (int s = 0; s < 3; ++s) d = mfma(WA1(s), e_s[PE(nb) + to[s]], d);
#pragma unroll
        for (int r = 0; r < 4; ++r) a1_s[(4 * q + r) * BPL + PAO(nb)] = fmaxf(d[r], 0.f);
      }
    } else {
      // ---- d a1[c_in][pos] = sum_{c_out,tap} W2[c_out][c_in][tap] dY2[c_out][pos - tap] on the 8 x 8 box (dY only)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;
        const float* bp = dy_s + q * DPL + boxo + PB(nb) + 2 * A1S + 2;
#pragma unroll
        for (int s = 0; s < 36; s += 2) {
          const int t0 = s >> 2, t1 = (s + 1) >> 2;
          d0 = mfma(WF(s), bp[4 * (s & 3) * DPL - (t0 / 3) * A1S - t0 % 3], d0);
          d1 = mfma(WF(s + 1), bp[4 * ((s + 1) & 3) * DPL - (t1 / 3) * A1S - t1 % 3], d1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {                            // ungated for now: a1 of this copy is still being written
          const float v = d0[r] + d1[r];
          dsum[r] += v;
          bx_s[(4 * q + r) * 64 + 16 * nb + j] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // a1 complete (role 0 says so, role 1 waits before it reads the gates); dY staged (the other way round, before dW2).
    // No workgroup barrier here: role 0 goes from its own conv1 straight into its dW2 taps while role 1 is in its 144 MFMAs
    // (LDS executes a wavefront's operations in order, so the counter bump lands behind the writes it announces).
    if (wave == 0) {
      if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__hip_atomic_load(flag + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < nth) __builtin_amdgcn_s_sleep(1);
    } else {
      while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < nth) __builtin_amdgcn_s_sleep(1);
    }
    // ---- dW2[c_out][c_in][tap] += sum_pos dY2[c_out][pos] a1[c_in][pos + tap]: taps 0..T0-1 (role 0), T0..8 (role 1)
    {
      const float* ap = dy_s + j * DPL + 2 * A1S + 2 + q;
      const float* bp = a1_s + j * BPL + q;
      if (wave == 0) {
#pragma unroll
        for (int s = 0; s < 30; ++s) {
          const int o = (s / 3) * A1S + 4 * (s % 3);
          const float av2 = ap[o];
#pragma unroll
          for (int t = 0; t < T0; ++t) acc2[t] = mfma(av2, bp[o + (t / 3) * A1S + t % 3], acc2[t]);
          if (s % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        // the gates first: d a1 of the box, gated by a1 > 0, into the box buffer [c][16 nb + j]
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {                          // (this lane wrote the entry itself: no barrier in between)
            float* g = bx_s + (4 * q + r) * 64 + 16 * nb + j;
            *g = a1_s[(4 * q + r) * BPL + boxo + PB(nb)] > 0.f ? *g : 0.f;
          }
#pragma unroll
        for (int s = 0; s < 30; ++s) {
          const int o = (s / 3) * A1S + 4 * (s % 3);
          const float av2 = ap[o];
#pragma unroll
          for (int t = T0; t < 9; ++t) acc2[t - T0] = mfma(av2, bp[o + (t / 3) * A1S + t % 3], acc2[t - T0]);
          if (s % 3 == 2) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- dW1[c][tap] += sum_box da1[c][pos] e[pos + tap];  db1[c] += sum_box da1[c][pos]  (column 9: constant 1)
        //      box position of K-step s, slot q: row s / 2, column 4 (s % 2) + q  ->  buffer index 8 (s / 2) + 4 (s % 2) + q
        const float* apb = bx_s + j * 64 + q;
        const float* bpe = e_s + boxe + eb;
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
          acc1a = mfma(apb[8 * (s / 2) + 4 * (s % 2)], bpe[(s / 2) * ES + 4 * (s % 2)], acc1a);
          acc1b = mfma(apb[8 * ((s + 1) / 2) + 4 * ((s + 1) % 2)], bpe[((s + 1) / 2) * ES + 4 * ((s + 1) % 2)], acc1b);
        }
      }
    }
    __syncthreads();                                             // B3: both roles are done with this copy's images
  }

  float* prow = a.part + (int64_t)blockIdx.x * PROW;
  if (wave == 0) {
#pragma unroll
    for (int t = 0; t < T0; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) prow[((4 * q + r) * NCH + j) * 9 + t] = acc2[t][r];
  } else {
#pragma unroll
    for (int t = T0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) prow[((4 * q + r) * NCH + j) * 9 + t] = acc2[t - T0][r];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = acc1a[r] + acc1b[r];
      if (j < 9) prow[NCH * NCH * 9 + (4 * q + r) * 9 + j] = v;
      else if (j == 9) prow[NCH * NCH * 9 + NCH * 9 + 4 * q + r] = v;
      const float ds = group_sum<16>(dsum[r]);
      if (j == 0) prow[NCH * NCH * 9 + NCH * 9 + NCH + 4 * q + r] = ds;
    }
  }
}
#undef WA1
#undef WF
#undef PE
#undef PAO
#undef CE
#undef CL
#undef DYO
#undef PB
#undef XV
#undef PV
#undef GV
#undef AV

// gWfc1[n][c*144 + py*12 + px] = bg[c] S[n] + sum over the <= 25 crop origins whose 5x5 block holds cell (py, px), each
// origin's gradient being the sum of its K chunks (oc[2g], oc[2g+1] = first chunk, number of chunks of origin g)
__global__ void sparse_scatter_fc1_k(const float* __restrict__ dWg, const int32_t* __restrict__ oc,
                                     const float* __restrict__ S, const float* __restrict__ bg, int F,
                                     float* __restrict__ gW) {
  const int64_t total = (int64_t)F * NCH * 144;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(idx % F), col = (int)(idx / F);
    const int c = col / 144, pos = col - 144 * c, py = pos / 12, px = pos - 12 * py;
    float s = bg[c] * S[n];
    for (int r0 = max(0, py - 4); r0 <= min(7, py); ++r0)
      for (int c0 = max(0, px - 4); c0 <= min(7, px); ++c0) {
        const int g = r0 * 8 + c0, k = ((py - r0) * 5 + px - c0) * NCH + c;
        for (int ch = oc[2 * g]; ch < oc[2 * g] + oc[2 * g + 1]; ++ch) s += dWg[((int64_t)ch * KD + k) * F + n];
      }
    gW[(int64_t)n * (NCH * 144) + col] = s;
  }
}

// One workgroup.  With S = column sums of g (= d bfc1), T[c] = sum over copies and cells of d pd[.., c] and
// Wsum[n][c] = sum_pos Wfc1[n][c*144+pos]:   d b2 = S . Wsum,   d bg = d b2 - T,  and through
// bg[c] = b2[c] + sum_{c',tap} W2[c][c'][tap] relu(b1[c']):   dW2 += d bg (x) relu(b1),   d b1 += [b1 > 0] W2^T d bg
// (+ the crop positions outside the boxes, see below).
__global__ __launch_bounds__(1024) void sparse_finish_k(const float* __restrict__ red, const float* __restrict__ S,
                                                        const float* __restrict__ T, const float* __restrict__ Wfc1,
                                                        const float* __restrict__ b1, const float* __restrict__ W2, int F,
                                                        float* __restrict__ gW1, float* __restrict__ gb1,
                                                        float* __restrict__ gW2, float* __restrict__ gb2) {
  __shared__ float wsum[16][NCH + 1];
  __shared__ float dbg[NCH], gb2s[NCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = 0.f;
  for (int idx = tid; idx < F * 144; idx += 1024) {
    const int n = idx / 144, pos = idx - 144 * n;
    const float sn = S[n];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = fmaf(sn, Wfc1[(int64_t)n * (NCH * 144) + c * 144 + pos], acc[c]);
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const float v = group_sum<64>(acc[c]);
    if (lane == 0) wsum[wave][c] = v;
  }
  __syncthreads();
  if (tid < NCH) {
    float v = 0.f, t = 0.f;
    for (int w = 0; w < 16; ++w) v += wsum[w][tid];
    for (int cell = 0; cell < NCELL; ++cell) t += T[cell * NCH + tid];
    gb2[tid] = v;
    gb2s[tid] = v;
    dbg[tid] = v - t;
  }
  __syncthreads();
  for (int i = tid; i < NCH * NCH * 9; i += 1024) {
    const int c = i / 144, ci = (i / 9) % NCH;
    gW2[i] = red[i] + dbg[c] * fmaxf(b1[ci], 0.f);
  }
  if (tid < NCH * 9) gW1[tid] = red[NCH * NCH * 9 + tid];
  if (tid < NCH) {
    // outside the boxes the conv1 gate is the constant [b1 > 0] and the d a1 of ALL positions sum to W2^T (column sums of
    // dY2 = T); together with the background path (d bg = d b2 - T) that is W2sum^T d b2 minus what the boxes hold
    float v = 0.f;
    if (b1[tid] > 0.f) {
      for (int c = 0; c < NCH; ++c)
        for (int t = 0; t < 9; ++t) v = fmaf(gb2s[c], W2[(c * NCH + tid) * 9 + t], v);
      v -= red[NCH * NCH * 9 + NCH * 9 + NCH + tid];
    }
    gb1[tid] = red[NCH * NCH * 9 + NCH * 9 + tid] + v;
  }
}

// ---- fc1 + ReLU + fc2 of the masked copies in ONE launch (MLP.py:43-47; inference: the levels of a sampling pass) ----------
// A level of the MNIST schedule holds ~700 masked copies in ~8 crop origins: the grouped 64 x 64-tile GEMM + the tall-layer
// launch for fc2 cost 17 + 6 us of pure latency per level (100 dependent LDS-staged k-steps per tile).  Here a workgroup owns
// 16 rows of one origin: 8 wavefronts = 2 halves of the 128 fc1 columns x 4 quarters of K = 400, every operand of a wavefront
// requested up front (A: the 16 x 400 tile of pd through LDS, one conflict-free ds_read_b32 per k-step; B: one dwordx4 of the
// k-major weight image straight into registers = the same k for four column sets, so a k-step is 4 MFMAs on one A register),
// the quarters summed in fixed order through LDS
// with bias + ReLU, and fc2 (<= 32 outputs) contracted from that LDS tile by wavefronts 0 and 1 against weights they asked for
// at the top of the kernel.
constexpr int kCropSplitMax = 256 * 8;                         // masked copies up to which a copy gets a whole workgroup
constexpr int FC_F = 128, FC_ROWS = 16, FC_WAVES = 8, FC_KQ = 4;
constexpr int FC_KSTEPS = KD / 4 / FC_KQ;                       // 25 MFMA k-steps per wavefront
constexpr int FC_HP = FC_F + 4;                                 // LDS pitch of a 128-wide row (bank shift 4 per row)
static_assert(FC_KSTEPS * 4 * FC_KQ == KD, "K = 400 splits into 4 quarters of 25 k-steps");

struct FcArgs {
  const float* pd; const float* Wg; const float* hbg; const int32_t* groups;
  const float* W2; const float* b2; float* out; int out_d;
};

__global__ __launch_bounds__(64 * FC_WAVES) void sparse_fc12_k(FcArgs a) {
  const int g = blockIdx.y;
  const int first = a.groups[2 * g], rows = a.groups[2 * g + 1];
  const int r0 = blockIdx.x * FC_ROWS;
  if (r0 >= rows) return;                                       // (uniform per workgroup: in front of every barrier)
  // the 16 x 400 tile of pd (pitch 404 = 20 mod 64 banks: the 64 (row, k) words of an A fragment lie in 64 banks), later
  // the four K-quarter partial tiles in the same bytes
  __shared__ __attribute__((aligned(16))) float buf[FC_KQ * FC_ROWS * FC_HP];
  __shared__ __attribute__((aligned(16))) float h1s[FC_ROWS][FC_HP];
  constexpr int AP = KD + 4;
  static_assert(FC_ROWS * AP <= FC_KQ * FC_ROWS * FC_HP, "the pd tile fits the partial-sum buffer");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, q = lane >> 4;
  const int ch = wave & 1, kq = wave >> 1;
  const int nvalid = rows - r0 < FC_ROWS ? rows - r0 : FC_ROWS;

  // fc2's weights of wavefronts 0 / 1: lane (n = j, q) holds W2[n][16 T + 4 q .. + 3]
  const int n2 = 16 * (wave & 1) + j;
  f32x4 wv[FC_F / 16];
  if (wave < 2) {
    const float* w2p = a.W2 + (int64_t)(n2 < a.out_d ? n2 : a.out_d - 1) * FC_F + 4 * q;
#pragma unroll
    for (int T = 0; T < FC_F / 16; ++T) wv[T] = *reinterpret_cast<const f32x4*>(w2p + 16 * T);
  }
  const float* bp = a.Wg + ((int64_t)g * KD + kq * (FC_KSTEPS * 4) + q) * FC_F + 64 * ch + 4 * j;
  f32x4 bv[FC_KSTEPS];
#pragma unroll
  for (int t = 0; t < FC_KSTEPS; ++t) bv[t] = *reinterpret_cast<const f32x4*>(bp + (int64_t)4 * t * FC_F);
  // pd rows of the tile: 100 dwordx4 per row, coalesced (as A fragments straight from memory every wave-load touched 16 rows)
  {
    const float* src = a.pd + ((int64_t)first + r0) * KD;
#pragma unroll
    for (int it = 0; it < (FC_ROWS * KD / 4 + 64 * FC_WAVES - 1) / (64 * FC_WAVES); ++it) {
      const int v = it * 64 * FC_WAVES + threadIdx.x, rr = v / (KD / 4), c4 = (v - rr * (KD / 4)) * 4;
      if (rr < FC_ROWS) {
        const int rs = rr < nvalid ? rr : nvalid - 1;            // rows past the end repeat the last one (not stored)
        *reinterpret_cast<f32x4*>(&buf[rr * AP + c4]) = *reinterpret_cast<const f32x4*>(src + (int64_t)rs * KD + c4);
      }
    }
  }
  __syncthreads();
  float av[FC_KSTEPS];
  {
    const float* ap = &buf[j * AP + kq * (FC_KSTEPS * 4) + q];
#pragma unroll
    for (int t = 0; t < FC_KSTEPS; ++t) av[t] = ap[4 * t];
  }
  f32x4 acc[4] = {};
#pragma unroll
  for (int t = 0; t < FC_KSTEPS; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = mfma(av[t], bv[t][i], acc[i]);
  }
  __syncthreads();                                               // every A fragment has been read: the bytes become `part`
  // D of product i: row 4 q + r, column 64 ch + 4 j + i  ->  a lane owns 4 consecutive columns of 4 rows
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    *reinterpret_cast<f32x4*>(&buf[(kq * FC_ROWS + 4 * q + r) * FC_HP + 64 * ch + 4 * j]) = v;
  }
  __syncthreads();
  {
    const int rr = threadIdx.x >> 5, c4 = (threadIdx.x & 31) * 4;
    f32x4 sum = *reinterpret_cast<const f32x4*>(&buf[rr * FC_HP + c4]);
#pragma unroll
    for (int k = 1; k < FC_KQ; ++k) sum += *reinterpret_cast<const f32x4*>(&buf[(k * FC_ROWS + rr) * FC_HP + c4]);
    const f32x4 bias = *reinterpret_cast<const f32x4*>(a.hbg + c4);
#pragma unroll
    for (int i = 0; i < 4; ++i) sum[i] = fmaxf(sum[i] + bias[i], 0.f);
    *reinterpret_cast<f32x4*>(&h1s[rr][c4]) = sum;
  }
  __syncthreads();
  if (wave < 2) {
    f32x4 o0 = {}, o1 = {};
#pragma unroll
    for (int T = 0; T < FC_F / 16; ++T) {                        // product (T, i): k = 16 T + 4 q + i in both operands
      const f32x4 hv = *reinterpret_cast<const f32x4*>(&h1s[j][16 * T + 4 * q]);
      o0 = mfma(hv[0], wv[T][0], o0);
      o1 = mfma(hv[1], wv[T][1], o1);
      o0 = mfma(hv[2], wv[T][2], o0);
      o1 = mfma(hv[3], wv[T][3], o1);
    }
    if (n2 < a.out_d) {
      const float b = a.b2[n2];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * q + r < nvalid) a.out[((int64_t)first + r0 + 4 * q + r) * a.out_d + n2] = o0[r] + o1[r] + b;
    }
  }
}

}  // namespace

extern "C" {

int64_t gnf_mnistcnn_sparse_ws_bytes(int64_t n_rows, int64_t F) {
  if (n_rows < 0 || F < 0) return 0;
  return (n_rows * KD + (int64_t)NORIG * KD * F + NCH + F + 64) * (int64_t)sizeof(float);
}

// the parameter-only part of the front: Wg [64][400][F] (the fc1 weight columns of each crop origin), bg [16] (conv2's output
// over the all-zero image), hbg [F] (fc1 of that background + bias) -- laid out as [Wg | bg | hbg]
static int sparse_tables(const float* b1, const float* W2, const float* b2, const float* Wfc1, const float* bfc1, int64_t F,
                         float* Wg, hipStream_t s) {
  float* bg = Wg + (int64_t)NORIG * KD * F;
  float* hbg = bg + NCH;
  hipLaunchKernelGGL(sparse_bg_k, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, s, b1, W2, b2, Wfc1, bfc1, (int)F, bg, hbg);
  GNF_LAUNCH_CHECK();
  hipLaunchKernelGGL(sparse_gather_fc1_k, dim3(2048), dim3(256), 0, s, Wfc1, (int)F, Wg);
  GNF_LAUNCH_CHECK();
  return 0;
}

static void sparse_crop_launch(const SparseArgs& a, hipStream_t s) {
  constexpr size_t lds = (size_t)WAVES * WLDS * sizeof(float);
  if (!a.arg && a.items <= kCropSplitMax) {                     // few copies: four wavefronts per copy (latency form)
    hipLaunchKernelGGL((sparse_crop_k<false, true>), dim3((unsigned)a.items), dim3(64 * WAVES), WLDS * sizeof(float), s, a);
    return;
  }
  int64_t grid = (a.items + WAVES - 1) / WAVES;
  const int per_cu = a.arg ? 3 : 2;                              // (Winograd inference form: 64 + 64 registers of U' and M)
  if (grid > 256 * per_cu) grid = 256 * per_cu;
  if (a.arg) hipLaunchKernelGGL((sparse_crop_k<true, false>), dim3((unsigned)grid), dim3(64 * WAVES), lds, s, a);
  else hipLaunchKernelGGL((sparse_crop_k<false, false>), dim3((unsigned)grid), dim3(64 * WAVES), lds, s, a);
}

// crop convolutions + grouped fc1 GEMM against tables that exist
static int sparse_front(const float* x, int64_t B, const float* P, const int32_t* pix, const int32_t* groups,
                        int64_t max_group_rows, const float* W1, const float* b1, const float* W2, const float* b2, int64_t F,
                        const float* Wg, float* h1, float* pd, unsigned char* argmax_save, int64_t items, hipStream_t s) {
  const float* bg = Wg + (int64_t)NORIG * KD * F;
  const float* hbg = bg + NCH;
  SparseArgs a{x, P, pix, W1, b1, W2, b2, bg, pd, argmax_save, B, items};
  sparse_crop_launch(a, s);
  GNF_LAUNCH_CHECK();

  GemmArgs g{};
  g.A = pd; g.sam = KD; g.sak = 1;
  g.B = Wg; g.sbk = F; g.sbn = 1; g.b_grp_stride = (int64_t)KD * F;
  g.C = h1; g.scm = F; g.scn = 1;
  g.bias = hbg; g.flags = GNF_GEMM_RELU;
  g.M = max_group_rows; g.N = F; g.K = KD;
  g.grp = groups;
  return gnf_gemm_grouped_launch(g, NORIG, s);
}

int gnf_mnistcnn_sparse_fwd(const float* x, int64_t B, const float* P, const int32_t* pix, int64_t R,
                            const int32_t* groups, int64_t max_group_rows,
                            const float* W1, const float* b1, const float* W2, const float* b2,
                            const float* Wfc1, const float* bfc1, int64_t F,
                            float* h1, float* pd_save, unsigned char* argmax_save,
                            void* ws, int64_t ws_bytes, gnf_stream_t stream) {
  if (!W1 || !b1 || !W2 || !b2 || !Wfc1 || !bfc1 || B < 0 || R < 0 || F <= 0 ||
      (pd_save == nullptr) != (argmax_save == nullptr))
    return GNF_EINVAL;
  if (F % 4 || F > 65535 * 64) return GNF_ESHAPE;
  const int64_t items = R * B;
  if (items == 0) return 0;                // batch- and row-sized arrays may be NULL for an empty call
  if (!x || !P || !pix || !groups || !h1) return GNF_EINVAL;
  if (!ws || ws_bytes < gnf_mnistcnn_sparse_ws_bytes(items, F) || max_group_rows <= 0 || max_group_rows > items)
    return GNF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* pd = pd_save ? pd_save : (float*)ws;   // [items][400]
  float* Wg = (float*)ws + items * KD;          // [64][400][F] | bg [16] | hbg [F]
  if (int rc = sparse_tables(b1, W2, b2, Wfc1, bfc1, F, Wg, s)) return rc;
  return sparse_front(x, B, P, pix, groups, max_group_rows, W1, b1, W2, b2, F, Wg, h1, pd, argmax_save, items, s);
}

int64_t gnf_mnistcnn_sparse_prep_bytes(int64_t F) {
  if (F < 0) return 0;
  return ((int64_t)NORIG * KD * F + NCH + F + 64) * (int64_t)sizeof(float);
}

int gnf_mnistcnn_sparse_prepare(const float* b1, const float* W2, const float* b2, const float* Wfc1, const float* bfc1,
                                int64_t F, void* prep, int64_t prep_bytes, gnf_stream_t stream) {
  if (!b1 || !W2 || !b2 || !Wfc1 || !bfc1 || !prep || F <= 0) return GNF_EINVAL;
  if (F % 4 || F > 65535 * 64) return GNF_ESHAPE;
  if (prep_bytes < gnf_mnistcnn_sparse_prep_bytes(F)) return GNF_EWS;
  return sparse_tables(b1, W2, b2, Wfc1, bfc1, F, (float*)prep, (hipStream_t)stream);
}

int gnf_mnistcnn_sparse_fwd_train(const float* x, int64_t B, const float* P, const int32_t* pix, int64_t R,
                                  const int32_t* groups, int64_t max_group_rows,
                                  const float* W1, const float* b1, const float* W2, const float* b2,
                                  const float* Wfc1, const float* bfc1, int64_t F,
                                  float* h1, float* pd_save, unsigned char* argmax_save,
                                  void* tables, int64_t tables_bytes, gnf_stream_t stream) {
  if (!W1 || !b1 || !W2 || !b2 || !Wfc1 || !bfc1 || B < 0 || R < 0 || F <= 0) return GNF_EINVAL;
  if (F % 4 || F > 65535 * 64) return GNF_ESHAPE;
  if (!tables || tables_bytes < gnf_mnistcnn_sparse_prep_bytes(F)) return GNF_EWS;
  hipStream_t s = (hipStream_t)stream;
  // the tables are built even for an empty call: the backward that follows reads them
  if (int rc = sparse_tables(b1, W2, b2, Wfc1, bfc1, F, (float*)tables, s)) return rc;
  const int64_t items = R * B;
  if (items == 0) return 0;
  if (!x || !P || !pix || !groups || !h1 || !pd_save || !argmax_save) return GNF_EINVAL;
  if (max_group_rows <= 0 || max_group_rows > items) return GNF_EINVAL;
  return sparse_front(x, B, P, pix, groups, max_group_rows, W1, b1, W2, b2, F, (const float*)tables, h1, pd_save, argmax_save,
                      items, s);
}

int gnf_mnistcnn_sparse_fwd_prepared(const float* x, int64_t B, const float* P, const int32_t* pix, int64_t R,
                                     const int32_t* groups, int64_t max_group_rows,
                                     const float* W1, const float* b1, const float* W2, const float* b2, int64_t F,
                                     const void* prep, float* h1, void* ws, int64_t ws_bytes, gnf_stream_t stream) {
  if (!W1 || !b1 || !W2 || !b2 || !prep || B < 0 || R < 0 || F <= 0) return GNF_EINVAL;
  if (F % 4 || F > 65535 * 64) return GNF_ESHAPE;
  const int64_t items = R * B;
  if (items == 0) return 0;
  if (!x || !P || !pix || !groups || !h1) return GNF_EINVAL;
  if (!ws || ws_bytes < items * KD * (int64_t)sizeof(float) || max_group_rows <= 0 || max_group_rows > items)
    return GNF_EINVAL;
  return sparse_front(x, B, P, pix, groups, max_group_rows, W1, b1, W2, b2, F, (const float*)prep, h1, (float*)ws, nullptr,
                      items, (hipStream_t)stream);
}

int gnf_mnistcnn_sparse_fwd_prepared_fc2(const float* x, int64_t B, const float* P, const int32_t* pix, int64_t R,
                                         const int32_t* groups, int64_t max_group_rows,
                                         const float* W1, const float* b1, const float* W2, const float* b2, int64_t F,
                                         const void* prep, const float* Wfc2, const float* bfc2, int64_t out_d,
                                         float* h2, void* ws, int64_t ws_bytes, gnf_stream_t stream) {
  if (!W1 || !b1 || !W2 || !b2 || !prep || !Wfc2 || !bfc2 || B < 0 || R < 0 || F <= 0 || out_d <= 0) return GNF_EINVAL;
  if (F != FC_F || out_d > 32) return GNF_ESHAPE;          // (other widths: gnf_mnistcnn_sparse_fwd_prepared + gnf_linear_fwd)
  const int64_t items = R * B;
  if (items == 0) return 0;
  if (!x || !P || !pix || !groups || !h2) return GNF_EINVAL;
  if (!ws || ws_bytes < items * KD * (int64_t)sizeof(float) || max_group_rows <= 0 || max_group_rows > items)
    return GNF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const float* Wg = (const float*)prep;
  const float* bg = Wg + (int64_t)NORIG * KD * F;
  SparseArgs a{x, P, pix, W1, b1, W2, b2, bg, (float*)ws, nullptr, B, items};
  sparse_crop_launch(a, s);
  GNF_LAUNCH_CHECK();
  FcArgs f{(const float*)ws, Wg, bg + NCH, groups, Wfc2, bfc2, h2, (int)out_d};
  hipLaunchKernelGGL(sparse_fc12_k, dim3((unsigned)((max_group_rows + FC_ROWS - 1) / FC_ROWS), NORIG), dim3(64 * FC_WAVES), 0, s, f);
  GNF_LAUNCH_CHECK();
  return 0;
}

static int64_t bwd_rows_n(int64_t F) {                 // widest row the two-level column sums see
  int64_t n = PROW;
  if (F > n) n = F;
  return n;
}

int64_t gnf_mnistcnn_sparse_bwd_ws_bytes(int64_t n_rows, int64_t F, int64_t n_kgroups) {
  if (n_rows < 0 || F < 0 || n_kgroups < 0) return 0;
  return (n_rows * KD + (int64_t)(NORIG + n_kgroups) * KD * F + NCH + KD + (int64_t)BWD_GRID * PROW + PROW +
          (int64_t)kRowsumChunks * bwd_rows_n(F) + 64) * (int64_t)sizeof(float);
}

int gnf_mnistcnn_sparse_bwd(const float* x, int64_t B, const float* P, const int32_t* pix, int64_t R,
                            const int32_t* groups, int64_t max_group_rows,
                            const int32_t* kgroups, int64_t n_kgroups, const int32_t* origin_chunks,
                            const float* W1, const float* b1, const float* W2, const float* b2,
                            const float* Wfc1, int64_t F,
                            const float* pd, const unsigned char* argmax, const float* g_h1,
                            float* gW1, float* gb1, float* gW2, float* gb2, float* gWfc1, float* gbfc1,
                            void* ws, int64_t ws_bytes, gnf_stream_t stream) {
  return gnf_mnistcnn_sparse_bwd_tables(x, B, P, pix, R, groups, max_group_rows, kgroups, n_kgroups, origin_chunks, W1, b1, W2, b2,
                                        Wfc1, F, nullptr, pd, argmax, g_h1, gW1, gb1, gW2, gb2, gWfc1, gbfc1, ws, ws_bytes, stream);
}

int gnf_mnistcnn_sparse_bwd_tables(const float* x, int64_t B, const float* P, const int32_t* pix, int64_t R,
                                   const int32_t* groups, int64_t max_group_rows,
                                   const int32_t* kgroups, int64_t n_kgroups, const int32_t* origin_chunks,
                                   const float* W1, const float* b1, const float* W2, const float* b2,
                                   const float* Wfc1, int64_t F, const void* tables,
                                   const float* pd, const unsigned char* argmax, const float* g_h1,
                                   float* gW1, float* gb1, float* gW2, float* gb2, float* gWfc1, float* gbfc1,
                                   void* ws, int64_t ws_bytes, gnf_stream_t stream) {
  if (n_kgroups < 0 || n_kgroups > 65535) return GNF_EINVAL;
  if (!W1 || !b1 || !W2 || !b2 || !Wfc1 || !gW1 || !gb1 || !gW2 || !gb2 || !gWfc1 || !gbfc1 || B < 0 || R < 0 || F <= 0)
    return GNF_EINVAL;
  if (R * B > 0 && (!kgroups || !origin_chunks || !x || !P || !pix || !groups || !pd || !argmax || !g_h1))
    return GNF_EINVAL;                       // an empty call only zeroes the weight gradients
  if (F % 4 || F > 65535 * 64) return GNF_ESHAPE;
  const int64_t items = R * B;
  hipStream_t s = (hipStream_t)stream;
  if (items == 0) {
    (void)hipMemsetAsync(gW1, 0, NCH * 9 * sizeof(float), s); (void)hipMemsetAsync(gb1, 0, NCH * sizeof(float), s);
    (void)hipMemsetAsync(gW2, 0, NCH * NCH * 9 * sizeof(float), s); (void)hipMemsetAsync(gb2, 0, NCH * sizeof(float), s);
    (void)hipMemsetAsync(gWfc1, 0, F * NCH * 144 * sizeof(float), s); (void)hipMemsetAsync(gbfc1, 0, F * sizeof(float), s);
    return 0;
  }
  if (!ws || ws_bytes < gnf_mnistcnn_sparse_bwd_ws_bytes(items, F, n_kgroups) || max_group_rows <= 0 ||
      max_group_rows > items || n_kgroups < 1)
    return GNF_EINVAL;
  float* dpd = (float*)ws;                              // [items][400]
  float* Wg = dpd + items * KD;                         // [64][400][F]
  float* dWg = Wg + (int64_t)NORIG * KD * F;            // [n_kgroups][400][F]
  float* bg = dWg + n_kgroups * KD * F;                 // [16]
  float* T = bg + NCH;                                  // [400]
  float* part = T + KD;                                 // [BWD_GRID][PROW]
  float* red = part + (int64_t)BWD_GRID * PROW;         // [PROW]
  float* rws = red + PROW;                              // two-level column-sum scratch
  int rc;

  if (tables) {                       // the forward's tables (same parameters): fc1 column blocks and the background response
    Wg = const_cast<float*>(static_cast<const float*>(tables));
    bg = Wg + (int64_t)NORIG * KD * F;
  } else {
    hipLaunchKernelGGL(sparse_bg_k, dim3(1), dim3(64), 0, s, b1, W2, b2, (const float*)nullptr, (const float*)nullptr, (int)F,
                       bg, (float*)nullptr);
    GNF_LAUNCH_CHECK();
    hipLaunchKernelGGL(sparse_gather_fc1_k, dim3(2048), dim3(256), 0, s, Wfc1, (int)F, Wg);
    GNF_LAUNCH_CHECK();
  }
  if ((rc = gnf_rowsum_tall_launch(g_h1, gbfc1, items, F, 0, rws, s))) return rc;           // S = d bfc1

  GemmArgs g{};                                         // d pd = g . Wg[origin]^T
  g.A = g_h1; g.sam = F; g.sak = 1;
  g.B = Wg; g.sbk = 1; g.sbn = F; g.b_grp_stride = (int64_t)KD * F;
  g.C = dpd; g.scm = KD; g.scn = 1;
  g.M = max_group_rows; g.N = KD; g.K = F;
  g.grp = groups;
  if ((rc = gnf_gemm_grouped_launch(g, NORIG, s))) return rc;
  if ((rc = gnf_rowsum_tall_launch(dpd, T, items, KD, 0, rws, s))) return rc;

  GemmArgs w{};                                         // dWg[chunk] = pd[rows of chunk]^T . g[rows of chunk]
  w.A = pd; w.sam = 1; w.sak = KD;
  w.B = g_h1; w.sbk = F; w.sbn = 1;
  w.C = dWg; w.scm = F; w.scn = 1; w.c_split_stride = (int64_t)KD * F;
  w.M = KD; w.N = F; w.K = items;
  w.grp = kgroups; w.grp_k = 1;
  if ((rc = gnf_gemm_grouped_launch(w, (int)n_kgroups, s))) return rc;
  hipLaunchKernelGGL(sparse_scatter_fc1_k, dim3(2048), dim3(256), 0, s, dWg, origin_chunks, gbfc1, bg, (int)F, gWfc1);
  GNF_LAUNCH_CHECK();

  SparseBwdArgs a{x, P, pix, W1, b1, W2, dpd, argmax, part, B, items};
  // exactly one resident wave of workgroups: a second, partial wave would idle most of the chip (every workgroup
  // walks the same number of copies)
  static int per_cu = 0, n_cu = 0;
  if (!per_cu) {
    int dev = 0, nb = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return GNF_EINVAL;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sparse_crop_bwd2_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)(BLDS2 * sizeof(float)));
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, sparse_crop_bwd2_k, 128, BLDS2 * sizeof(float)) != hipSuccess || nb < 1)
      return GNF_EINVAL;
    n_cu = prop.multiProcessorCount;
    per_cu = nb;
  }
  int64_t grid = (int64_t)per_cu * n_cu;
  if (grid > BWD_GRID) grid = BWD_GRID;
  if (grid > items) grid = items;
  hipLaunchKernelGGL(sparse_crop_bwd2_k, dim3((unsigned)grid), dim3(128), BLDS2 * sizeof(float), s, a);
  GNF_LAUNCH_CHECK();
  if ((rc = gnf_rowsum_launch(part, red, grid, PROW, 0, s))) return rc;
  hipLaunchKernelGGL(sparse_finish_k, dim3(1), dim3(1024), 0, s, red, gbfc1, T, Wfc1, b1, W2, (int)F, gW1, gb1, gW2, gb2);
  GNF_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
