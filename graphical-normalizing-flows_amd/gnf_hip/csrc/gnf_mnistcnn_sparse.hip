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
// gfx950 mapping (forward only; a training step needing gradients uses the dense pair):
//   crop kernel: ONE WAVEFRONT per masked copy, everything in that wave's 10 KB of LDS -> 12 waves per CU.
//     conv1 as 9 x 3 and conv2 as 7 x 36 v_mfma_f32_16x16x4_f32 (direct convolution as implicit GEMM: M = 16 output
//     channels, N = 16 positions, K = taps / (tap, input channel)); the weights are the A operand and stay in registers
//     for the whole kernel, the B operand is one ds_read_b32 per MFMA at an immediate offset.  conv2's N index is
//     (pool cell, position in the cell), so the 2x2 max-pool is two quad DPP ops on the accumulator and one lane per
//     quad stores 4 channels of a cell as a float4: rows of `pd` are [cell][channel].
//   fc1: the 64 column blocks are gathered once per call into k-major images Wg[64][400][F]; masked copies arrive
//     sorted by block, so ONE grouped launch of the fp32 MFMA GEMM (gnf_gemm.hip) does bias + ReLU for all of them.
#include "gnf_common.h"
#include "gnf_gemm.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
  float* pd; int64_t B, items;
};

__global__ __launch_bounds__(64 * WAVES, 3) void sparse_crop_k(SparseArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  float* e_s = smem + wave * WLDS;
  float* a1_s = e_s + ESZ;

  // ---- weights as MFMA A operands: lane (j = output channel, q = K slot)
  float wa1[3], wa2[36];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = 4 * s + q;
    wa1[s] = tap < 9 ? a.W1[j * 9 + tap] : 0.f;
  }
#pragma unroll
  for (int s = 0; s < 36; ++s) wa2[s] = a.W2[(j * NCH + 4 * (s & 3) + q) * 9 + (s >> 2)];
  f32x4 bias1, bias2, bgv;
#pragma unroll
  for (int r = 0; r < 4; ++r) { bias1[r] = a.b1[4 * q + r]; bias2[r] = a.b2[4 * q + r]; bgv[r] = a.bg[4 * q + r]; }

  // ---- per-lane LDS offsets, independent of the masked copy
  int eo[9][3];                   // conv1 B operand: e_s offset of (position 16 nb + j, tap 4 s + q)
#pragma unroll
  for (int nb = 0; nb < 9; ++nb) {
    const int p = 16 * nb + j, y = p / A1, x = p - A1 * y;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int tap = 4 * s + q, ty = tap < 9 ? tap / 3 : 0, tx = tap < 9 ? tap - 3 * ty : 0;
      eo[nb][s] = (y + ty) * ES + x + tx;
    }
  }
  int po[7];                      // conv2 B operand: a1_s offset of (cell 4 nb + j/4, sub-position j%4, channel slot q)
#pragma unroll
  for (int nb = 0; nb < 7; ++nb) {
    int cell = 4 * nb + (j >> 2);
    cell = cell < NCELL ? cell : NCELL - 1;
    const int cy = cell / 5, cx = cell - 5 * cy;
    po[nb] = q * PL + (2 * cy + ((j >> 1) & 1)) * A1 + 2 * cx + (j & 1);
  }
  const int a1w = 4 * q * PL + j;                              // conv1 D store: channel 4q + r, position 16 nb + j

  // ---- the crop of e = x * P[pixel] of one masked copy: 196 values, 4 per lane (the last lane group idles on the 4th)
  int ce[4], cl[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int idx = lane + 64 * t, ey = idx / CROP, ex = idx - CROP * ey;
    ce[t] = ey * IMG + ex;                                     // offset inside the image, relative to the crop corner
    cl[t] = idx < CROP * CROP ? ey * ES + ex : -1;
  }
  float xv[4], pv[4];
  auto fetch = [&](int64_t item) {
    const int64_t r = item / a.B, b = item - r * a.B;
    const int pix = a.pix[r];
    const int corner = 2 * crop_origin(pix / IMG) * IMG + 2 * crop_origin(pix % IMG);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bool ok = cl[t] >= 0;
      xv[t] = ok ? a.x[b * NPIX + corner + ce[t]] : 0.f;
      pv[t] = ok ? a.P[(int64_t)pix * NPIX + corner + ce[t]] : 0.f;
    }
  };

  const int64_t stride = (int64_t)gridDim.x * WAVES;
  int64_t item = (int64_t)blockIdx.x * WAVES + wave;
  if (item < a.items) fetch(item);
  for (; item < a.items; item += stride) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (cl[t] >= 0) e_s[cl[t]] = xv[t] * pv[t];
    if (item + stride < a.items) fetch(item + stride);         // next copy's loads fly during this one's MFMAs

    // ---- conv1 + ReLU -> a1_s
#pragma unroll
    for (int nb = 0; nb < 9; ++nb) {
      f32x4 d = bias1;
#pragma unroll
      for (int s = 0; s < 3; ++s) d = mfma(wa1[s], e_s[eo[nb][s]], d);
#pragma unroll
      for (int r = 0; r < 4; ++r) a1_s[a1w + r * PL + 16 * nb] = fmaxf(d[r], 0.f);
    }
    // ---- conv2 + 2x2 max-pool + bias - background -> pd[item][cell][channel]
    float* prow = a.pd + item * KD + 4 * q;
#pragma unroll
    for (int nb = 0; nb < 7; ++nb) {
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = d0;                // two chains: consecutive MFMAs are independent
      const float* bp = a1_s + po[nb];
#pragma unroll
      for (int s = 0; s < 36; s += 2) {
        const int t0 = s >> 2, t1 = (s + 1) >> 2;
        d0 = mfma(wa2[s], bp[4 * (s & 3) * PL + (t0 / 3) * A1 + t0 % 3], d0);
        d1 = mfma(wa2[s + 1], bp[4 * ((s + 1) & 3) * PL + (t1 / 3) * A1 + t1 % 3], d1);
      }
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = quad_max(d0[r] + d1[r]) + bias2[r] - bgv[r];
      const int cell = 4 * nb + (j >> 2);
      if ((j & 3) == 0 && cell < NCELL) *reinterpret_cast<f32x4*>(prow + cell * NCH) = v;
    }
  }
}

}  // namespace

extern "C" {

int64_t gnf_mnistcnn_sparse_ws_bytes(int64_t n_rows, int64_t F) {
  if (n_rows < 0 || F < 0) return 0;
  return (n_rows * KD + (int64_t)NORIG * KD * F + NCH + F + 64) * (int64_t)sizeof(float);
}

int gnf_mnistcnn_sparse_fwd(const float* x, int64_t B, const float* P, const int32_t* pix, int64_t R,
                            const int32_t* groups, int64_t max_group_rows,
                            const float* W1, const float* b1, const float* W2, const float* b2,
                            const float* Wfc1, const float* bfc1, int64_t F,
                            float* h1, void* ws, int64_t ws_bytes, gnf_stream_t stream) {
  if (!x || !P || !pix || !groups || !W1 || !b1 || !W2 || !b2 || !Wfc1 || !bfc1 || !h1 || B < 0 || R < 0 || F <= 0)
    return GNF_EINVAL;
  if (F % 4 || F > 65535 * 64) return GNF_ESHAPE;
  const int64_t items = R * B;
  if (items == 0) return 0;
  if (!ws || ws_bytes < gnf_mnistcnn_sparse_ws_bytes(items, F) || max_group_rows <= 0 || max_group_rows > items)
    return GNF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  float* pd = (float*)ws;                       // [items][400]
  float* Wg = pd + items * KD;                  // [64][400][F]
  float* bg = Wg + (int64_t)NORIG * KD * F;     // [16]
  float* hbg = bg + NCH;                        // [F]

  hipLaunchKernelGGL(sparse_bg_k, dim3((unsigned)((F + 3) / 4)), dim3(256), 0, s, b1, W2, b2, Wfc1, bfc1, (int)F, bg, hbg);
  GNF_LAUNCH_CHECK();
  hipLaunchKernelGGL(sparse_gather_fc1_k, dim3(2048), dim3(256), 0, s, Wfc1, (int)F, Wg);
  GNF_LAUNCH_CHECK();

  SparseArgs a{x, P, pix, W1, b1, W2, b2, bg, pd, B, items};
  constexpr size_t lds = (size_t)WAVES * WLDS * sizeof(float);
  int64_t grid = (items + WAVES - 1) / WAVES;
  if (grid > 256 * 3) grid = 256 * 3;
  hipLaunchKernelGGL(sparse_crop_k, dim3((unsigned)grid), dim3(64 * WAVES), lds, s, a);
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

}  // extern "C"
