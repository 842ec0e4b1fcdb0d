// Monotonic (UMNN) normalizer: Clenshaw-Curtis quadrature of the integrand MLP, forward,
// fused bisection inverse, and backward (models/Normalizers/MonotonicNormalizer.py:21-83 +
// the UMNN 1.0 NeuralIntegral algorithm, restated in oracle/gnf_oracle.py).
//
// Compute-bound (>1000 flop/B): the work is (S+2) evaluations per element of a small MLP,
// so everything is organised around v_mfma_f32_16x16x4_f32 (exact fp32):
//
//  * A wavefront owns a GROUP of 16 elements (b,i).  For one quadrature node the hidden
//    state is a [H x 16] matrix held in the MFMA C/D layout: lane (q=lane>>4, j=lane&15)
//    holds hidden units 16t+4q+r (t = tile, r = 0..3) of element j.
//  * A hidden layer is out[H x 16] = W[H x H] * act[H x 16].  The weights are the MFMA
//    A operand (one float4 per lane per (out-tile, in-tile) read from the padded weight
//    image: LDS-resident when it fits, L1/L2 otherwise), the activations are the B
//    operand.  Because the contraction index may be visited in any order, K-step (t,r)
//    pairs lane-slot q with hidden unit 16t+4q+r -- exactly what the C/D registers of the
//    previous layer already hold.  Layers therefore chain register-to-register: no LDS
//    round trip, no cross-lane shuffle between layers.
//  * Layer 1 is split: W1[:,1:]*h + b1 does not depend on the node, so it is computed once
//    per group (MFMA) and each node only adds the rank-1 term w1x * x_k on the VALU.
//  * The last layer (H -> 1) is a per-lane dot product + two __shfl_xor over the q axis.
//  * Two nodes are processed together so every A fragment feeds 8 MFMAs.
//
// Backward = recompute-in-kernel (no saved activations): the chain kernel re-runs the
// forward per node, back-propagates dpre through the transposed weights with the same
// register chaining, writes dx, dh, and accumulates bias / first / last layer gradients
// in per-lane registers.  Hidden->hidden weight gradients are reductions over (element,
// node) rows: the chain kernel stages act[l-1] and dpre[l] row-major to a workspace and
// the split-K fp32 MFMA GEMM (gnf_gemm.hip) contracts them, chunk by chunk.
#include "gnf_common.h"
#include "gnf_gemm.h"
#include "gnf_linear_tall.h"
#include "gnf_monotonic.h"
#include <cstdlib>

extern "C" int gnf_gemm_split_enabled(void);
// kernel family of the process' last forward / backward launch (gnf_monotonic_fwd_kernel / _bwd_kernel: reporting only; NOT
// thread-local -- autograd runs the backward on its own thread and the caller asks from another)
static const char* volatile g_fwd_kernel = "";
static const char* volatile g_bwd_kernel = "";

namespace {

using namespace gnfmono;

struct PackArgs {
  gnf_mono_net net;
  MonoLayout L;
};

__global__ void mono_pack_k(PackArgs a, float* __restrict__ pack) {
  const MonoLayout& L = a.L;
  const gnf_mono_net& N = a.net;
  const int H0 = N.dims[1];
  auto unit = [&](int p, int l) { return mono_unit_at(p, N.dims[l], L.perm); };   // unit of hidden layer l at padded position p
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < L.pack_floats; idx += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (L.NH > 1 && L.o_Wq[1] && idx >= L.o_Wq[1]) {   // bf16 split planes of a peeled narrow net's 48 x 48 main blocks
      const bool tr = idx >= L.o_WTq[1];
      const int l = 1 + (idx - (tr ? L.o_WTq[1] : L.o_Wq[1])) / kNarrowQ, k = (idx - L.o_Wq[1]) % kNarrowQ;
      const int plane = k / (3 * 384), kk = k % (3 * 384), mt = kk / 384, w = kk % 384;
      int lane, c0;                                    // the word holds contraction positions c0, c0 + 1
      if (w < 256) { lane = w >> 2; const int i = 2 * (w & 3); c0 = (i < 4 ? 0 : 16) + 4 * (lane >> 4) + (i & 3); }
      else { lane = (w - 256) >> 1; c0 = 32 + 4 * (lane >> 4) + 2 * ((w - 256) & 1); }
      const int row = 16 * mt + (lane & 15);
      float x[2] = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r_ = unit(row, tr ? l : l + 1), c_ = unit(c0 + i, tr ? l + 1 : l);
        if (r_ >= 0 && c_ >= 0) x[i] = tr ? N.W[l][(int64_t)c_ * N.dims[l] + r_] : N.W[l][(int64_t)r_ * N.dims[l] + c_];
      }
      unsigned word = 0;
      for (int pl = 0; pl <= plane; ++pl) {
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(word) : "v"(x[0]), "v"(x[1]));
        x[0] -= __uint_as_float(word << 16);
        x[1] -= __uint_as_float(word & 0xffff0000u);
      }
      reinterpret_cast<unsigned*>(pack)[idx] = word;
      continue;
    }
    if (L.NH > 1 && L.o_Wp[1] && idx >= L.o_Wp[1]) {   // bf16 split planes of the hidden->hidden matrices (see MonoLayout)
      const int per = 3 * L.HT * L.KT32 * 256;
      const bool tr = idx >= L.o_WTp[1];               // transposed planes behind the forward ones
      const int l = 1 + (idx - (tr ? L.o_WTp[1] : L.o_Wp[1])) / per, k = (idx - L.o_Wp[1]) % per;
      const int plane = k / (L.HT * L.KT32 * 256), kk = k % (L.HT * L.KT32 * 256);
      const int frag = kk >> 8, lane = (kk >> 2) & 63, w2 = kk & 3, mt = frag / L.KT32, t = frag - mt * L.KT32;
      const int q = lane >> 4, j = lane & 15;
      const int row = unit(16 * mt + j, tr ? l : l + 1);          // fragment row: out unit (transposed: in unit)
      float x[2] = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int cp = 32 * t + 8 * q + 2 * w2 + i;
        const int col = cp < L.HP ? unit(cp, tr ? l + 1 : l) : -1;  // contraction index: in unit (transposed: out unit)
        if (row >= 0 && col >= 0) x[i] = tr ? N.W[l][(int64_t)col * N.dims[l] + row] : N.W[l][(int64_t)row * N.dims[l] + col];
      }
      unsigned word = 0;
      for (int pl = 0; pl <= plane; ++pl) {           // hi, then the rounded remainders (exact subtractions)
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(word) : "v"(x[0]), "v"(x[1]));
        x[0] -= __uint_as_float(word << 16);
        x[1] -= __uint_as_float(word & 0xffff0000u);
      }
      reinterpret_cast<unsigned*>(pack)[idx] = word;
      continue;
    }
    if (idx >= L.total_floats) {                  // fragment-major copies of the hidden->hidden matrices (see MonoLayout)
      for (int l = 1; l < L.NH; ++l) {
        const bool tr = idx >= L.o_WTf[l];
        const int k = idx - (tr ? L.o_WTf[l] : L.o_Wf[l]);
        if (k < 0 || k >= L.HP * L.HP) continue;
        const int frag = k >> 8, lane = (k >> 2) & 63, r = k & 3, mt = frag / L.HT, t = frag - mt * L.HT;
        const int q = lane >> 4, j = lane & 15;
        const int row = unit(tr ? 16 * t + 4 * q + r : 16 * mt + j, l + 1);     // out unit
        const int col = unit(tr ? 16 * mt + j : 16 * t + 4 * q + r, l);         // in unit
        if (row >= 0 && col >= 0) v = N.W[l][(int64_t)row * N.dims[l] + col];
        break;
      }
      pack[idx] = v;
      continue;
    }
    if (idx < L.o_b1) { const int k = unit(idx - L.o_w1x, 1); if (k >= 0) v = N.W[0][(int64_t)k * N.dims[0]]; }
    else if (idx < L.o_wL) { const int k = unit(idx - L.o_b1, 1); if (k >= 0) v = N.b[0][k]; }
    else if (idx < L.o_bL) { const int k = unit(idx - L.o_wL, L.NH); if (k >= 0) v = N.W[L.NH][k]; }
    else if (idx < L.o_W1h) { if (idx == L.o_bL) v = N.b[L.NH][0]; }
    else if (idx < (L.NH > 1 ? L.o_W[1] : L.fwd_floats)) {
      const int k = idx - L.o_W1h; const int r = unit(k / L.LDH, 1), cc = k % L.LDH;
      if (r >= 0 && cc < L.c) v = N.W[0][(int64_t)r * N.dims[0] + 1 + cc];
    } else if (idx < L.fwd_floats) {
      for (int l = 1; l < L.NH; ++l) {
        if (idx >= L.o_W[l] && idx < L.o_b[l]) {
          const int k = idx - L.o_W[l]; const int r = unit(k / L.LDW, l + 1), cc = unit(k % L.LDW, l);
          if (r >= 0 && cc >= 0) v = N.W[l][(int64_t)r * N.dims[l] + cc];
        } else if (idx >= L.o_b[l] && idx < L.o_b[l] + L.HP) {
          const int k = unit(idx - L.o_b[l], l + 1); if (k >= 0) v = N.b[l][k];
        }
      }
    } else if (idx < L.o_W1hT) {
      for (int l = 1; l < L.NH; ++l) {
        if (idx >= L.o_WT[l] && idx < L.o_WT[l] + L.HP * L.LDW) {
          const int k = idx - L.o_WT[l]; const int r = unit(k / L.LDW, l), cc = unit(k % L.LDW, l + 1);   // r = in, cc = out
          if (r >= 0 && cc >= 0) v = N.W[l][(int64_t)cc * N.dims[l] + r];
        }
      }
    } else {
      const int k = idx - L.o_W1hT; const int r = k / L.LDW, cc = unit(k % L.LDW, 1);       // r = cond idx, cc = hidden
      if (r < L.c && cc >= 0) v = N.W[0][(int64_t)cc * N.dims[0] + 1 + r];
    }
    pack[idx] = v;
  }
}

// c1[t][r] = b1[hid] + sum_cc W1h[hid][cc] * h[elem][cc],  hid = 16t+4q+r   (MFMA, once per group)
template <int HT>
__device__ __forceinline__ void cond_bias(const float* wp, const MonoLayout& L, const float* __restrict__ h,
                                          int64_t hbase, int64_t h_sc, int q, int j, f32x4 (&c1)[HT]) {
#pragma unroll
  for (int t = 0; t < HT; ++t) c1[t] = ld4(wp + L.o_b1 + 16 * t + 4 * q);
  for (int s = 0; s < L.CP / 16; ++s) {
    float hv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cc = 16 * s + 4 * q + r;
      hv[r] = cc < L.c ? h[hbase + cc * h_sc] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      const f32x4 A = ld4(wp + L.o_W1h + (16 * t + j) * L.LDH + 16 * s + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) c1[t] = mfma(A[r], hv[r], c1[t]);
    }
  }
}

// f(xa;h), f(xb;h) for the 16 elements of the group: two nodes share every weight fragment.
template <int HT, class GetW>
__device__ __forceinline__ void eval2(const float* wp, const MonoLayout& L, const f32x4 (&c1)[HT], float xa,
                                      float xb, int q, int j, float& fa, float& fb, GetW&& getW) {
  f32x4 a0[HT], a1[HT];
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    const f32x4 wx = ld4(wp + L.o_w1x + 16 * t + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a0[t][r] = fmaxf(fmaf(wx[r], xa, c1[t][r]), 0.f);
      a1[t][r] = fmaxf(fmaf(wx[r], xb, c1[t][r]), 0.f);
    }
  }
  for (int l = 1; l < L.NH; ++l) {
    const float* W = getW(l);                  // hidden->hidden matrix of layer l (L2, resident LDS image, or swapped in)
    f32x4 o0[HT], o1[HT];
#pragma unroll
    for (int mt = 0; mt < HT; ++mt) { o0[mt] = ld4(wp + L.o_b[l] + 16 * mt + 4 * q); o1[mt] = o0[mt]; }
#pragma unroll
    for (int t = 0; t < HT; ++t) {
#pragma unroll
      for (int mt = 0; mt < HT; ++mt) {
        const f32x4 A = ld4(W + (16 * mt + j) * L.LDW + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o0[mt] = mfma(A[r], a0[t][r], o0[mt]);
          o1[mt] = mfma(A[r], a1[t][r], o1[mt]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { a0[t][r] = fmaxf(o0[t][r], 0.f); a1[t][r] = fmaxf(o1[t][r], 0.f); }
  }
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    const f32x4 wl = ld4(wp + L.o_wL + 16 * t + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) { s0 = fmaf(wl[r], a0[t][r], s0); s1 = fmaf(wl[r], a1[t][r], s1); }
  }
  const float bL = wp[L.o_bL];
  fa = elu_plus(qsum(s0) + bL);
  fb = elu_plus(qsum(s1) + bL);
}

// sum_k w_k f(xT (t_k+1)/2) over the S+1 quadrature nodes; optionally also f(xj) (Jacobian node)
template <int HT, bool WITH_JAC, class GetW>
__device__ __forceinline__ float quadrature(const float* wp, const MonoLayout& L, const f32x4 (&c1)[HT],
                                            const float* __restrict__ ccw, const float* __restrict__ cct, int S,
                                            float xT, float xj, int q, int j, float& fjac, GetW&& getW) {
  float acc = 0.f;
  const int total = S + 1 + (WITH_JAC ? 1 : 0);
  for (int k = 0; k < total; k += 2) {
    const int k1 = k + 1;
    const float wa = k <= S ? ccw[k] : 0.f;
    const float wb = k1 <= S ? ccw[k1] : 0.f;
    const float xa = k <= S ? xT * (cct[k] + 1.f) * .5f : xj;
    const float xb = k1 <= S ? xT * (cct[k1] + 1.f) * .5f : xj;
    float fa, fb;
    eval2<HT>(wp, L, c1, xa, xb, q, j, fa, fb, getW);
    acc = fmaf(wa, fa, acc);
    acc = fmaf(wb, fb, acc);
    if (WITH_JAC) {
      if (k == S + 1) fjac = fa;
      if (k1 == S + 1) fjac = fb;
    }
  }
  return acc;
}

// ---------------------------------------------------------------------------------------
// Peeled forward.  H = 50 (the reference's default integrand net) padded to 4 tiles of 16 spends a quarter of every
// hidden layer's MFMAs on a tile that holds TWO real units (and a quarter of the K steps likewise).  Here the HM = H/16
// full tiles stay on the MFMA (9 instead of 16 tile products per layer) and the EX = H mod 16 leftover units are
// carried as plain per-element scalars, replicated over the four q-lanes of an element: their contribution to the main
// units is a rank-EX update (W[:, U] from the padded image, one b128 per row), their own pre-activations are per-lane
// partial dot products over the lane's 4 HM units + a q-sum.  The pack (HP = 64 image) is unchanged.
// ---------------------------------------------------------------------------------------
template <int HM, int EX, class GetW>
__device__ __forceinline__ void eval2x(const float* wp, const MonoLayout& L, const f32x4 (&c1)[HM], const float (&c1x)[EX],
                                       float xa, float xb, int q, int j, float& fa, float& fb, GetW&& getW) {
  constexpr int U0 = 16 * HM;                  // first peeled unit
  f32x4 a0[HM], a1[HM];
  float x0[EX], x1[EX];
#pragma unroll
  for (int t = 0; t < HM; ++t) {
    const f32x4 wx = ld4(wp + L.o_w1x + 16 * t + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a0[t][r] = fmaxf(fmaf(wx[r], xa, c1[t][r]), 0.f);
      a1[t][r] = fmaxf(fmaf(wx[r], xb, c1[t][r]), 0.f);
    }
  }
  {
    const f32x4 wx = ld4(wp + L.o_w1x + U0);
#pragma unroll
    for (int e = 0; e < EX; ++e) {
      x0[e] = fmaxf(fmaf(wx[e], xa, c1x[e]), 0.f);
      x1[e] = fmaxf(fmaf(wx[e], xb, c1x[e]), 0.f);
    }
  }
  for (int l = 1; l < L.NH; ++l) {
    const float* W = getW(l);
    // every weight fragment of the peeled part is requested BEFORE the MFMA block, so that the LDS latency of these
    // 21 b128 reads hides under the 72 MFMAs instead of stalling the VALU tail read by read
    f32x4 wr[EX][HM], wxr[EX], wc[HM][4];
    const f32x4 bx = ld4(wp + L.o_b[l] + U0);
#pragma unroll
    for (int e = 0; e < EX; ++e) {
#pragma unroll
      for (int t = 0; t < HM; ++t) wr[e][t] = ld4(W + (U0 + e) * L.LDW + 16 * t + 4 * q);
      wxr[e] = ld4(W + (U0 + e) * L.LDW + U0);
    }
#pragma unroll
    for (int mt = 0; mt < HM; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) wc[mt][r] = ld4(W + (16 * mt + 4 * q + r) * L.LDW + U0);
    f32x4 o0[HM], o1[HM];
#pragma unroll
    for (int mt = 0; mt < HM; ++mt) { o0[mt] = ld4(wp + L.o_b[l] + 16 * mt + 4 * q); o1[mt] = o0[mt]; }
    __builtin_amdgcn_sched_barrier(0);           // keep the requests above the MFMA block (the scheduler would sink them)
#pragma unroll
    for (int t = 0; t < HM; ++t) {
#pragma unroll
      for (int mt = 0; mt < HM; ++mt) {
        const f32x4 A = ld4(W + (16 * mt + j) * L.LDW + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o0[mt] = mfma(A[r], a0[t][r], o0[mt]);
          o1[mt] = mfma(A[r], a1[t][r], o1[mt]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);           // ... and the VALU tail below it (VALU between MFMAs costs issue slots)
    // the peeled units' own pre-activations (from the OLD a / x), then their rank-EX contribution to the main units
    float y0[EX], y1[EX];
#pragma unroll
    for (int e = 0; e < EX; ++e) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int t = 0; t < HM; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0 = fmaf(wr[e][t][r], a0[t][r], s0); s1 = fmaf(wr[e][t][r], a1[t][r], s1); }
      float k0 = bx[e], k1 = bx[e];
#pragma unroll
      for (int e2 = 0; e2 < EX; ++e2) { k0 = fmaf(wxr[e][e2], x0[e2], k0); k1 = fmaf(wxr[e][e2], x1[e2], k1); }
      y0[e] = fmaxf(qsum(s0) + k0, 0.f);
      y1[e] = fmaxf(qsum(s1) + k1, 0.f);
    }
#pragma unroll
    for (int mt = 0; mt < HM; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v0 = o0[mt][r], v1 = o1[mt][r];
#pragma unroll
        for (int e = 0; e < EX; ++e) { v0 = fmaf(wc[mt][r][e], x0[e], v0); v1 = fmaf(wc[mt][r][e], x1[e], v1); }
        a0[mt][r] = fmaxf(v0, 0.f);
        a1[mt][r] = fmaxf(v1, 0.f);
      }
#pragma unroll
    for (int e = 0; e < EX; ++e) { x0[e] = y0[e]; x1[e] = y1[e]; }
  }
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int t = 0; t < HM; ++t) {
    const f32x4 wl = ld4(wp + L.o_wL + 16 * t + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) { s0 = fmaf(wl[r], a0[t][r], s0); s1 = fmaf(wl[r], a1[t][r], s1); }
  }
  const f32x4 wlx = ld4(wp + L.o_wL + U0);
  float t0 = wp[L.o_bL], t1 = t0;
#pragma unroll
  for (int e = 0; e < EX; ++e) { t0 = fmaf(wlx[e], x0[e], t0); t1 = fmaf(wlx[e], x1[e], t1); }
  fa = elu_plus(qsum(s0) + t0);
  fb = elu_plus(qsum(s1) + t1);
}

// The same evaluation with the 48 x 48 main block of every hidden->hidden layer on the bf16 matrix pipe (round 6; the method
// of gnf_gemm_split.hip / mono_fwd_wide_split_k): the lane splits the 12 activations it holds in its C/D registers exactly
// into three bf16 numbers each (x = hi + mid + lo) -- tiles 0 and 1 side by side are the 8-value B operand of a K = 32 MFMA,
// tile 2 that of a K = 16 MFMA, no LDS round trip --, the weights come pre-split from the pack (MonoLayout::o_Wq, LDS
// resident: one conflict-free lane-linear read per fragment), a product is its six leading cross terms, hi hi in one fp32
// accumulator and the five small terms in another.  72 MFMAs of 17 cycles per layer and node pair instead of 72 of 32.5, and
// VALU work next to bf16 MFMAs is not serialised the way it is next to fp32 MFMAs (profiles/r06_mfma_k16_rate.txt).  The
// peeled units, layer 1, the last layer and the quadrature are the fp32 code of eval2x.
typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
typedef unsigned u32x2n __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8n;
typedef __attribute__((ext_vector_type(4))) short s16x4n;
typedef float f32x2n __attribute__((ext_vector_type(2)));
// (v_cvt_pk_bf16_f32 through the conversion builtin, NOT inline asm: the results feed MFMAs a few instructions later, and the
// compiler's hazard recognizer inserts the VALU-write -> MFMA-read wait states only for instructions it can see.  With the asm
// form this kernel read stale operands: errors of 1e-2 .. 1 that moved with every rebuild)
typedef __bf16 bf16x2n __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16n(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2n{a, b}, bf16x2n));
}
__device__ __forceinline__ void split3_pairn(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16n(x0, x1);
  const f32x2n r = f32x2n{x0, x1} - f32x2n{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};   // exact
  m = cvt_pk_bf16n(r[0], r[1]);
  const f32x2n q = r - f32x2n{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};                // exact
  l = cvt_pk_bf16n(q[0], q[1]);
}
// the lane's 12 values of one node (tiles 0..2 of a 48-unit layer) -> B operands: K = 32 (tiles 0, 1) and K = 16 (tile 2), 3 planes
__device__ __forceinline__ void split_acts48(const f32x4 (&a)[3], u32x4n (&b32)[3], u32x2n (&b16)[3]) {
  unsigned h[6], m[6], l[6];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    split3_pairn(a[t][0], a[t][1], h[2 * t], m[2 * t], l[2 * t]);
    split3_pairn(a[t][2], a[t][3], h[2 * t + 1], m[2 * t + 1], l[2 * t + 1]);
  }
  b32[0] = u32x4n{h[0], h[1], h[2], h[3]}; b16[0] = u32x2n{h[4], h[5]};
  b32[1] = u32x4n{m[0], m[1], m[2], m[3]}; b16[1] = u32x2n{m[4], m[5]};
  b32[2] = u32x4n{l[0], l[1], l[2], l[3]}; b16[2] = u32x2n{l[4], l[5]};
}
__device__ __forceinline__ f32x4 mfma_bf32(const u32x4n& a, const u32x4n& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8n, a), __builtin_bit_cast(bf16x8n, b), c, 0, 0, 0);
}
// The 16-wide remainder of the contraction (tile 2) ALSO goes through the K = 32 instruction, upper half zero.  The legacy
// v_mfma_f32_16x16x16_bf16 takes the same 17 cycles on gfx950 (profiles/r06_mfma_k16_rate.txt), so nothing is lost -- and with
// it this kernel produced wrong sums in the optimised build (errors of 1e-2 .. 1; right in every instrumented build; the
// operand planes verified bit-identical by device printf).  Cause NOT isolated: the instruction alone and the two forms
// chained on one accumulator are correct in stand-alone kernels (tools/mfma_chain_check.hip).  The K = 32 form is correct in
// every build of this kernel; tests/test_gpu_mono_split.py::test_narrow_* pin it against fp64.
__device__ __forceinline__ f32x4 mfma_bf16k(const u32x2n& a, const u32x2n& b, f32x4 c) {
  return mfma_bf32(u32x4n{a[0], a[1], 0u, 0u}, u32x4n{b[0], b[1], 0u, 0u}, c);
}
// out[mt] (+)= W_main[16 mt + ., :48] x acts for two nodes; big: hi hi, sml: the five small terms.  Q: the matrix' planes (LDS)
// (ONE: a single accumulator class -- the backward kernel, whose registers are full: the same error level as the fp32 MFMA)
template <bool ONE = false>
__device__ __forceinline__ void block48_split(const unsigned* Q, int lane, const u32x4n (&b32)[2][3], const u32x2n (&b16)[2][3],
                                              f32x4 (&big)[2][3], f32x4 (&sml)[2][3]) {
#pragma unroll
  for (int mt = 0; mt < 3; ++mt) {
    u32x4n A32[3];
    u32x2n A16[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      A32[p] = *reinterpret_cast<const u32x4n*>(Q + (p * 3 + mt) * 384 + 4 * lane);
      A16[p] = *reinterpret_cast<const u32x2n*>(Q + (p * 3 + mt) * 384 + 256 + 2 * lane);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      auto term = [&](int pa, int pb, f32x4& c) {
        c = mfma_bf32(A32[pa], b32[u][pb], c);
        c = mfma_bf16k(A16[pa], b16[u][pb], c);
      };
      f32x4& cs = ONE ? big[u][mt] : sml[u][mt];
      term(2, 0, cs); term(0, 2, cs); term(1, 1, cs); term(1, 0, cs); term(0, 1, cs);
      term(0, 0, big[u][mt]);
    }
  }
}

template <int EX>
__device__ __forceinline__ void eval2x_split(const float* wp, const unsigned* wq, const MonoLayout& L, const f32x4 (&c1)[3],
                                             const float (&c1x)[EX], float xa, float xb, int q, int j, float& fa, float& fb) {
  constexpr int HM = 3, U0 = 48;
  const int lane = 16 * q + j;
  f32x4 a[2][HM];
  float x0[EX], x1[EX];
#pragma unroll
  for (int t = 0; t < HM; ++t) {
    const f32x4 wx = ld4(wp + L.o_w1x + 16 * t + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a[0][t][r] = fmaxf(fmaf(wx[r], xa, c1[t][r]), 0.f);
      a[1][t][r] = fmaxf(fmaf(wx[r], xb, c1[t][r]), 0.f);
    }
  }
  {
    const f32x4 wx = ld4(wp + L.o_w1x + U0);
#pragma unroll
    for (int e = 0; e < EX; ++e) {
      x0[e] = fmaxf(fmaf(wx[e], xa, c1x[e]), 0.f);
      x1[e] = fmaxf(fmaf(wx[e], xb, c1x[e]), 0.f);
    }
  }
  for (int l = 1; l < L.NH; ++l) {
    const float* W = wp + L.o_W[l];
    u32x4n b32[2][3];
    u32x2n b16[2][3];
    split_acts48(a[0], b32[0], b16[0]);
    split_acts48(a[1], b32[1], b16[1]);
    f32x4 big[2][HM], sml[2][HM];
#pragma unroll
    for (int mt = 0; mt < HM; ++mt) {
      big[0][mt] = ld4(wp + L.o_b[l] + 16 * mt + 4 * q);
      big[1][mt] = big[0][mt];
      sml[0][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      sml[1][mt] = sml[0][mt];
    }
    block48_split<false>(wq + (l - 1) * kNarrowQ, lane, b32, b16, big, sml);
    // (the peeled part's weights are read BEHIND the block here: ahead of it, as in eval2x, their 80 registers push the
    // kernel over 256 and a second workgroup off the CU -- the other workgroup's MFMAs cover these reads instead)
    __builtin_amdgcn_sched_barrier(0);
    f32x4 wr[EX][HM], wxr[EX], wc[HM][4];
    const f32x4 bx = ld4(wp + L.o_b[l] + U0);
#pragma unroll
    for (int e = 0; e < EX; ++e) {
#pragma unroll
      for (int t = 0; t < HM; ++t) wr[e][t] = ld4(W + (U0 + e) * L.LDW + 16 * t + 4 * q);
      wxr[e] = ld4(W + (U0 + e) * L.LDW + U0);
    }
#pragma unroll
    for (int mt = 0; mt < HM; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) wc[mt][r] = ld4(W + (16 * mt + 4 * q + r) * L.LDW + U0);
    // the peeled units' own pre-activations (from the OLD a / x), then their rank-EX contribution to the main units
    float y0[EX], y1[EX];
#pragma unroll
    for (int e = 0; e < EX; ++e) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int t = 0; t < HM; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0 = fmaf(wr[e][t][r], a[0][t][r], s0); s1 = fmaf(wr[e][t][r], a[1][t][r], s1); }
      float k0 = bx[e], k1 = bx[e];
#pragma unroll
      for (int e2 = 0; e2 < EX; ++e2) { k0 = fmaf(wxr[e][e2], x0[e2], k0); k1 = fmaf(wxr[e][e2], x1[e2], k1); }
      y0[e] = fmaxf(qsum(s0) + k0, 0.f);
      y1[e] = fmaxf(qsum(s1) + k1, 0.f);
    }
#pragma unroll
    for (int mt = 0; mt < HM; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v0 = big[0][mt][r] + sml[0][mt][r], v1 = big[1][mt][r] + sml[1][mt][r];
#pragma unroll
        for (int e = 0; e < EX; ++e) { v0 = fmaf(wc[mt][r][e], x0[e], v0); v1 = fmaf(wc[mt][r][e], x1[e], v1); }
        a[0][mt][r] = fmaxf(v0, 0.f);
        a[1][mt][r] = fmaxf(v1, 0.f);
      }
#pragma unroll
    for (int e = 0; e < EX; ++e) { x0[e] = y0[e]; x1[e] = y1[e]; }
  }
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int t = 0; t < HM; ++t) {
    const f32x4 wl = ld4(wp + L.o_wL + 16 * t + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) { s0 = fmaf(wl[r], a[0][t][r], s0); s1 = fmaf(wl[r], a[1][t][r], s1); }
  }
  const f32x4 wlx = ld4(wp + L.o_wL + U0);
  float t0 = wp[L.o_bL], t1 = t0;
#pragma unroll
  for (int e = 0; e < EX; ++e) { t0 = fmaf(wlx[e], x0[e], t0); t1 = fmaf(wlx[e], x1[e], t1); }
  fa = elu_plus(qsum(s0) + t0);
  fb = elu_plus(qsum(s1) + t1);
}

// first-layer pre-activations without the x term: HM main tiles + EX peeled scalars (the once-per-group MFMA runs
// over HM + 1 tiles; the peeled rows sit in lanes q = 0 of the last tile and are broadcast to the element's other lanes)
template <int HM, int EX>
__device__ __forceinline__ void cond_bias_x(const float* wp, const MonoLayout& L, const float* __restrict__ h,
                                            int64_t hbase, int64_t h_sc, int q, int j, f32x4 (&c1)[HM],
                                            float (&c1x)[EX]) {
  f32x4 full[HM + 1];
  cond_bias<HM + 1>(wp, L, h, hbase, h_sc, q, j, full);
#pragma unroll
  for (int t = 0; t < HM; ++t) c1[t] = full[t];
#pragma unroll
  for (int e = 0; e < EX; ++e) c1x[e] = __shfl(full[HM][e], j, 64);
}

// SP: the main blocks on the bf16 matrix pipe (eval2x_split; WM = 1, HM = 3, forward only): the planes of the NH - 1 matrices
// follow the fp32 image in LDS
template <int HM, int EX, int WM, bool INV, bool SP = false>
__global__ __launch_bounds__(64 * kWaves) void mono_fwd_x_k(MonoArgs a) {
  static_assert(WM == 0 || WM == 1, "peeled nets are narrow: the forward image is resident (1) or cached (0)");
  static_assert(!SP || (WM == 1 && HM == 3 && !INV), "split form: resident image, 48-unit main block, forward");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const MonoLayout& L = a.L;
  const float* wp = a.pack;
  const unsigned* wq = nullptr;
  if (WM == 1) {
    float* img = smem;
    if (SP) {                                          // the planes first, the fp32 image behind them
      const int nq = (L.NH - 1) * kNarrowQ;
      for (int i = threadIdx.x * 4; i < nq; i += blockDim.x * 4)
        *reinterpret_cast<f32x4*>(smem + i) = ld4(a.pack + L.o_Wq[1] + i);
      wq = reinterpret_cast<const unsigned*>(smem);
      img = smem + nq;
    }
    for (int i = threadIdx.x * 4; i < L.fwd_floats; i += blockDim.x * 4)
      *reinterpret_cast<f32x4*>(img + i) = ld4(a.pack + i);
    __syncthreads();
    wp = img;
  }
  auto getW = [&](int l) -> const float* { return wp + L.o_W[l]; };
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int64_t ngroups = (a.n + 15) / 16;
  const float fS = (float)a.S;
  for (int64_t grp = (int64_t)blockIdx.x * kWaves + wave; grp < ngroups; grp += (int64_t)gridDim.x * kWaves) {
    const int64_t e = grp * 16 + j;
    const bool valid = e < a.n;
    const int64_t ec = valid ? e : a.n - 1;
    const int64_t b = ec / a.d, i = ec - b * a.d;
    const int64_t hbase = b * a.h_sb + i * a.h_sd;
    f32x4 c1[HM];
    float c1x[EX];
    cond_bias_x<HM, EX>(wp, L, a.h, hbase, a.h_sc, q, j, c1, c1x);
    const float h0 = a.h[hbase];
    auto quad = [&](float xT, float xj, bool with_jac, float& fjac) {
      float acc = 0.f;
      const int total = a.S + 1 + (with_jac ? 1 : 0);
      for (int k = 0; k < total; k += 2) {
        const int k1 = k + 1;
        const float wa = k <= a.S ? a.ccw[k] : 0.f;
        const float wb = k1 <= a.S ? a.ccw[k1] : 0.f;
        const float xa = k <= a.S ? xT * (a.cct[k] + 1.f) * .5f : xj;
        const float xb = k1 <= a.S ? xT * (a.cct[k1] + 1.f) * .5f : xj;
        float fa, fb;
        if constexpr (SP) eval2x_split<EX>(wp, wq, L, c1, c1x, xa, xb, q, j, fa, fb);
        else eval2x<HM, EX>(wp, L, c1, c1x, xa, xb, q, j, fa, fb, getW);
        acc = fmaf(wa, fa, acc);
        acc = fmaf(wb, fb, acc);
        if (with_jac) {
          if (k == a.S + 1) fjac = fa;
          if (k1 == a.S + 1) fjac = fb;
        }
      }
      return acc;
    };
    float fj = 0.f;
    if (!INV) {
      const float xv = a.x[ec];
      const float xT = fS * (xv / fS);                 // xT = x0 + nb_steps*step, x0 = 0
      const float zs = quad(xT, xv, true, fj);
      if (valid && q == 0) {
        a.z[e] = zs * xT * .5f + h0;
        a.jac[e] = fj;
      }
    } else {
      const float zt = a.zt[ec];
      float xmax = 20.f, xmin = -20.f;
      for (int it = 0; it < 20; ++it) {
        const float xm = (xmax + xmin) * .5f;
        const float xT = fS * (xm / fS);
        const float zm = quad(xT, 0.f, false, fj) * xT * .5f + h0;
        if (zm > zt) xmax = xm; else xmin = xm;
      }
      if (valid && q == 0) store_inverse(a, e, (xmax + xmin) * .5f);
    }
  }
}

// WM 0: weight fragments from L1/L2.  1: the forward part of the pack LDS-resident (H <= 112).  2: wide nets -- one
// hidden->hidden matrix in LDS at a time, swapped in by the whole workgroup (two quadrature nodes share every residency);
// the wavefronts of a workgroup then run the same number of group iterations (a surplus wavefront recomputes the last
// group and writes nothing).
template <int HT, int WM, bool INV>
__global__ __launch_bounds__(64 * kWaves) void mono_fwd_k(MonoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const MonoLayout& L = a.L;
  const float* wp = a.pack;
  if (WM == 1) {
    for (int i = threadIdx.x * 4; i < L.fwd_floats; i += blockDim.x * 4)
      *reinterpret_cast<f32x4*>(smem + i) = ld4(a.pack + i);
    __syncthreads();
    wp = smem;
  }
  int resident = 0;                              // WM 2: layer whose matrix sits in LDS; workgroup-uniform
  auto getW = [&](int l) -> const float* {
    if (WM != 2) return wp + L.o_W[l];
    if (resident != l) {
      const int matf = L.HP * L.LDW, stride = blockDim.x * 4;
      __syncthreads();                            // previous matrix no longer read
      for (int i = threadIdx.x * 4; i < matf; i += stride) glds16(a.pack + L.o_W[l] + i, smem + i);   // all in flight
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      resident = l;
    }
    return smem;
  };
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int64_t ngroups = (a.n + 15) / 16;
  const float fS = (float)a.S;
  for (int64_t g0 = (int64_t)blockIdx.x * kWaves; g0 < ngroups; g0 += (int64_t)gridDim.x * kWaves) {
    const bool gvalid = g0 + wave < ngroups;
    if (WM != 2 && !gvalid) break;               // without workgroup barriers a surplus wavefront simply stops
    const int64_t grp = gvalid ? g0 + wave : ngroups - 1;
    const int64_t e = grp * 16 + j;
    const bool valid = gvalid && e < a.n;
    const int64_t ec = e < a.n ? e : a.n - 1;
    const int64_t b = ec / a.d, i = ec - b * a.d;
    const int64_t hbase = b * a.h_sb + i * a.h_sd;
    f32x4 c1[HT];
    cond_bias<HT>(wp, L, a.h, hbase, a.h_sc, q, j, c1);
    const float h0 = a.h[hbase];
    float dummy = 0.f;
    if (!INV) {
      const float xv = a.x[ec];
      const float xT = fS * (xv / fS);                 // xT = x0 + nb_steps*step, x0 = 0
      float fj = 0.f;
      const float zs = quadrature<HT, true>(wp, L, c1, a.ccw, a.cct, a.S, xT, xv, q, j, fj, getW);
      if (valid && q == 0) {
        a.z[e] = zs * xT * .5f + h0;
        a.jac[e] = fj;
      }
    } else {
      const float zt = a.zt[ec];
      float xmax = 20.f, xmin = -20.f;
      for (int it = 0; it < 20; ++it) {
        const float xm = (xmax + xmin) * .5f;
        const float xT = fS * (xm / fS);
        const float zm = quadrature<HT, false>(wp, L, c1, a.ccw, a.cct, a.S, xT, 0.f, q, j, dummy, getW) * xT * .5f + h0;
        if (zm > zt) xmax = xm; else xmin = xm;
      }
      if (valid && q == 0) store_inverse(a, e, (xmax + xmin) * .5f);
    }
  }
}

// Inverse for FEW elements (level-scheduled sampling evaluates a handful of variables per DAG level): the bisection
// is a chain of 20 dependent quadratures, so with fewer groups than SIMDs the launch is pure latency.  Here the (up to
// 8) wavefronts of a workgroup share ONE group of 16 elements and split the quadrature nodes (pairs w, w+nw, ...); the
// partial sums meet in LDS once per bisection step (double-buffered: one barrier per step) and are added in a fixed
// order, so all wavefronts take identical decisions.  WM as in mono_fwd_k (0 or 1).
// EPG (elements per group) = 4: the 16 columns of a tile are 4 elements x 4 node pairs instead of 16 elements x 1, i.e. a
// group is a quarter of the elements and needs a quarter of the wavefronts -- four times as many workgroups for the same
// MFMA work.  A DAG level of MNIST sampling is ~720 elements: 45 groups of 16 kept 45 of the 256 CUs busy, each with 12
// wavefronts queueing on its 4 MFMA pipes (a bisection step = 3 pair evaluations deep); 180 groups of 4 x 3 wavefronts
// put ONE pair evaluation per SIMD on 180 CUs.
constexpr int kSplitWaves = 8;
template <int HT, int WM, int EPG = 16>
__global__ __launch_bounds__(64 * kSplitWaves) void mono_inv_split_k(MonoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const MonoLayout& L = a.L;
  const float* wp = a.pack;
  float* psum = smem;                            // [2][nw][16]
  constexpr int NS = 16 / EPG;                   // node sub-slots of a wavefront: lane j -> element j % EPG, sub-slot j / EPG
  const int nw = blockDim.x >> 6;                // wavefronts sharing the group (<= kSplitWaves)
  if (WM == 1) {
    for (int i = threadIdx.x * 4; i < L.fwd_floats; i += blockDim.x * 4)
      *reinterpret_cast<f32x4*>(smem + i) = ld4(a.pack + i);
    __syncthreads();
    wp = smem;
    psum = smem + L.fwd_floats;
  }
  auto getW = [&](int l) -> const float* { return wp + L.o_W[l]; };
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int64_t ngroups = (a.n + EPG - 1) / EPG;
  const float fS = (float)a.S;
  int buf = 0;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t e = grp * EPG + (j % EPG);
    const int slot = wave * NS + j / EPG, nslots = nw * NS;        // this lane's node-pair slot
    const bool valid = e < a.n;
    const int64_t ec = valid ? e : a.n - 1;
    const int64_t b = ec / a.d, i = ec - b * a.d;
    const int64_t hbase = b * a.h_sb + i * a.h_sd;
    f32x4 c1[HT];
    cond_bias<HT>(wp, L, a.h, hbase, a.h_sc, q, j, c1);
    const float h0 = a.h[hbase];
    const float zt = a.zt[ec];
    float xmax = 20.f, xmin = -20.f;
    for (int it = 0; it < 20; ++it) {
      const float xm = (xmax + xmin) * .5f;
      const float xT = fS * (xm / fS);
      float acc = 0.f;
      for (int kb = 0; kb <= a.S; kb += 2 * nslots) {              // wave-uniform trip count; a lane past the rule idles at weight 0
        const int kk = kb + 2 * slot;
        const bool on = kk <= a.S;
        const int k = on ? kk : 0, k1 = k + 1;
        const float wa = on ? a.ccw[k] : 0.f;
        const float wb = (on && k1 <= a.S) ? a.ccw[k1] : 0.f;
        const float xa = xT * (a.cct[k] + 1.f) * .5f;
        const float xb = k1 <= a.S ? xT * (a.cct[k1] + 1.f) * .5f : xa;
        float fa, fb;
        eval2<HT>(wp, L, c1, xa, xb, q, j, fa, fb, getW);
        acc = fmaf(wa, fa, acc);
        acc = fmaf(wb, fb, acc);
      }
      if (q == 0) psum[(buf * nw + wave) * 16 + j] = acc;
      __syncthreads();
      // the element's node slots: wavefront-major, sub-slot-minor, the same order in every lane of the element
      const float* ps = psum + buf * nw * 16 + (j % EPG);
      float tot = 0.f;
      for (int w = 0; w < nw; ++w)
#pragma unroll
        for (int u = 0; u < NS; ++u) tot += ps[16 * w + EPG * u];
      const float zm = tot * xT * .5f + h0;
      buf ^= 1;
      if (zm > zt) xmax = xm; else xmin = xm;
    }
    if (valid && q == 0 && wave == 0 && j < EPG) store_inverse(a, e, (xmax + xmin) * .5f);
  }
}

// The same for the peeled narrow nets (H in {49, 50, 51}: three MFMA tiles + EX units on the VALU, eval2x) with up to
// TWELVE wavefronts per group: at S = 20 the 11 node pairs get a wavefront each, so a bisection step is ONE pair
// evaluation (144 MFMAs) instead of two of the padded form (2 x 256) -- the level kernel of MNIST sampling 256 -> see
// DESIGN.md section 6.  The partial sums are added in the fixed order of the wavefronts, as above.
constexpr int kSplitWavesX = 12;     // 3 wavefronts per SIMD: 168 registers (at 16 / 128 registers eval2x spills)
template <int HM, int EX, int WM, int EPG = 16>
__global__ __launch_bounds__(64 * kSplitWavesX) void mono_inv_split_x_k(MonoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const MonoLayout& L = a.L;
  const float* wp = a.pack;
  float* psum = smem;                            // [2][nw][16]
  constexpr int NS = 16 / EPG;                   // node sub-slots of a wavefront: lane j -> element j % EPG, sub-slot j / EPG
  const int nw = blockDim.x >> 6;                // wavefronts sharing the group (<= kSplitWavesX)
  if (WM == 1) {
    for (int i = threadIdx.x * 4; i < L.fwd_floats; i += blockDim.x * 4)
      *reinterpret_cast<f32x4*>(smem + i) = ld4(a.pack + i);
    __syncthreads();
    wp = smem;
    psum = smem + L.fwd_floats;
  }
  auto getW = [&](int l) -> const float* { return wp + L.o_W[l]; };
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int64_t ngroups = (a.n + EPG - 1) / EPG;
  const float fS = (float)a.S;
  int buf = 0;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t e = grp * EPG + (j % EPG);
    const int slot = wave * NS + j / EPG, nslots = nw * NS;        // this lane's node-pair slot
    const bool valid = e < a.n;
    const int64_t ec = valid ? e : a.n - 1;
    const int64_t b = ec / a.d, i = ec - b * a.d;
    const int64_t hbase = b * a.h_sb + i * a.h_sd;
    f32x4 c1[HM];
    float c1x[EX];
    cond_bias_x<HM, EX>(wp, L, a.h, hbase, a.h_sc, q, j, c1, c1x);
    const float h0 = a.h[hbase];
    const float zt = a.zt[ec];
    float xmax = 20.f, xmin = -20.f;
    for (int it = 0; it < 20; ++it) {
      const float xm = (xmax + xmin) * .5f;
      const float xT = fS * (xm / fS);
      float acc = 0.f;
      for (int kb = 0; kb <= a.S; kb += 2 * nslots) {              // wave-uniform trip count; a lane past the rule idles at weight 0
        const int kk = kb + 2 * slot;
        const bool on = kk <= a.S;
        const int k = on ? kk : 0, k1 = k + 1;
        const float wa = on ? a.ccw[k] : 0.f;
        const float wb = (on && k1 <= a.S) ? a.ccw[k1] : 0.f;
        const float xa = xT * (a.cct[k] + 1.f) * .5f;
        const float xb = k1 <= a.S ? xT * (a.cct[k1] + 1.f) * .5f : xa;
        float fa, fb;
        eval2x<HM, EX>(wp, L, c1, c1x, xa, xb, q, j, fa, fb, getW);
        acc = fmaf(wa, fa, acc);
        acc = fmaf(wb, fb, acc);
      }
      if (q == 0) psum[(buf * nw + wave) * 16 + j] = acc;
      __syncthreads();
      // the element's node slots: wavefront-major, sub-slot-minor, the same order in every lane of the element
      const float* ps = psum + buf * nw * 16 + (j % EPG);
      float tot = 0.f;
      for (int w = 0; w < nw; ++w)
#pragma unroll
        for (int u = 0; u < NS; ++u) tot += ps[16 * w + EPG * u];
      const float zm = tot * xT * .5f + h0;
      buf ^= 1;
      if (zm > zt) xmax = xm; else xmin = xm;
    }
    if (valid && q == 0 && wave == 0 && j < EPG) store_inverse(a, e, (xmax + xmin) * .5f);
  }
}

// Round 5, the level kernel of a sampling pass: TWO bisection steps per round.  Step 2 evaluates the midpoint of whichever
// half step 1 keeps, i.e. one of the two quarter points -- so a round evaluates the integrand at the nodes of THREE points
// (midpoint, both quarter points) at once and takes both decisions from the three quadrature sums: 10 dependent quadratures
// instead of 20, and the kernel is pure latency (180 workgroups of 3 wavefronts at S = 20, one pair evaluation deep).  The
// 3 (S + 1) evaluations of an element are laid out FLAT over the node-pair slots (eval2x takes any two abscissae of an
// element): 63 evaluations = 32 pairs = 8 wavefronts of 4 elements x 4 slots at S = 20 -- two pair evaluations per SIMD and
// round, where a first version with one point per wavefront triple (3 x 3 wavefronts, 36 slots for 33 pairs) had three and
// LOST to the sequential kernel (113 vs 103 us per level).  The raw integrand values meet in LDS; every lane of an element
// then forms the three sums in exactly the order of mono_inv_split_x_k (pairs (0,1), (2,3), ... as fma chains, added
// ascending), with the same midpoints: the result is that of the 20 sequential steps BIT FOR BIT (tested).  S <= 31.
// EPG = elements per workgroup: 4 (8 wavefronts at S = 20), or 2 (4 wavefronts: ONE per SIMD) when the call holds so few
// elements that every workgroup of two still finds a CU of its own -- a round is then one pair evaluation deep instead of two.
// MAXW = wavefronts the launch may use: 8 (S <= 20 with four elements: 256 registers, nothing spills) or kSplitWavesX (168
// registers: the round's uniform values then live in spilled SGPRs, ~230 v_readlane / s_nop slots per round).
template <int HM, int EX, int EPG, int MAXW>
__global__ __launch_bounds__(64 * MAXW) void mono_inv_ks_x_k(MonoArgs a) {
  constexpr int NS = 16 / EPG;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const MonoLayout& L = a.L;
  for (int i = threadIdx.x * 4; i < L.fwd_floats; i += blockDim.x * 4)
    *reinterpret_cast<f32x4*>(smem + i) = ld4(a.pack + i);
  const int S1 = a.S + 1, NV = 3 * S1;
  constexpr int SP = 32;                         // node slots per point: S + 1 <= 32 (the launch condition), the rest stay 0
  float* wq = smem + L.fwd_floats;               // [32] weights of the rule, zero past S
  float* tq = wq + SP;                           // [32] its nodes
  float* fbuf = tq + SP;                         // [2][3][32][EPG] integrand values of the round (slots past S: 0, never written)
  float* zs = fbuf + 2 * 3 * SP * EPG;           // [2][3][EPG] the quadrature sums of the round
  for (int i = threadIdx.x; i < SP; i += blockDim.x) { wq[i] = i < S1 ? a.ccw[i] : 0.f; tq[i] = i < S1 ? a.cct[i] : 0.f; }
  for (int i = threadIdx.x; i < 2 * 3 * SP * EPG; i += blockDim.x) fbuf[i] = 0.f;
  __syncthreads();
  const float* wp = smem;
  auto getW = [&](int l) -> const float* { return wp + L.o_W[l]; };
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15, el = j % EPG;
  // this lane's two evaluations: flat index v = point * (S + 1) + node
  const int v0 = 2 * (wave * NS + j / EPG), v1 = v0 + 1;
  const bool on0 = v0 < NV, on1 = v1 < NV;
  const int p0 = on0 ? v0 / S1 : 0, k0 = on0 ? v0 - p0 * S1 : 0;
  const int p1 = on1 ? v1 / S1 : p0, k1 = on1 ? v1 - p1 * S1 : k0;
  const float ta = (tq[k0] + 1.f), tb = (tq[k1] + 1.f);
  const int64_t ngroups = (a.n + EPG - 1) / EPG;
  const float fS = (float)a.S;
  int buf = 0;
  for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int64_t e = grp * EPG + el;
    const bool valid = e < a.n;
    const int64_t ec = valid ? e : a.n - 1;
    const int64_t b = ec / a.d, i = ec - b * a.d;
    const int64_t hbase = b * a.h_sb + i * a.h_sd;
    f32x4 c1[HM];
    float c1x[EX];
    cond_bias_x<HM, EX>(wp, L, a.h, hbase, a.h_sc, q, j, c1, c1x);
    const float h0 = a.h[hbase];
    const float zt = a.zt[ec];
    float xmax = 20.f, xmin = -20.f;
    for (int it = 0; it < 20; it += 2) {
      const float xm = (xmax + xmin) * .5f;
      float xp[3], xT[3];
      xp[0] = xm; xp[1] = (xm + xmin) * .5f; xp[2] = (xmax + xm) * .5f;      // step 2's midpoint for either outcome of step 1
#pragma unroll
      for (int p = 0; p < 3; ++p) xT[p] = fS * (xp[p] / fS);
      const float xTa = p0 == 0 ? xT[0] : (p0 == 1 ? xT[1] : xT[2]);
      const float xTb = p1 == 0 ? xT[0] : (p1 == 1 ? xT[1] : xT[2]);
      const float xa = xTa * ta * .5f, xb = xTb * tb * .5f;
      float fa, fb;
#ifdef GNF_INV_EXP_NOEVAL      // timing build: the round without its integrand evaluations (wrong results)
      fa = xa * c1[0][0]; fb = xb * c1x[0];
#else
      eval2x<HM, EX>(wp, L, c1, c1x, xa, xb, q, j, fa, fb, getW);
#endif
      float* fw = fbuf + buf * 3 * SP * EPG + el;
      if (q == 0) {
        if (on0) fw[(p0 * SP + k0) * EPG] = fa;
        if (on1) fw[(p1 * SP + k1) * EPG] = fb;
      }
      __syncthreads();
      // the three sums of each element: ONE lane per (point, element) -- lane (q = point, j = element) of wavefront 0 -- walks
      // the 21 values in the sequential kernel's order and leaves z in LDS (every lane of the workgroup forming all three
      // sums itself was 126 LDS reads per lane and round: slower than the 20 sequential steps)
      if (wave == 0 && q < 3 && j < EPG) {
        const float* f = fbuf + (buf * 3 + q) * SP * EPG + j;
        // 16 pairs at FIXED offsets: the slots past S hold w = 0 and f = 0, so their pairs add +0 (as a loop over a run-time
        // node count every pair waited for its own LDS round trip, and with clamped run-time indices the 64 uniform addresses
        // sat in SGPRs that spilled)
        float fv[SP], wv[SP];
#pragma unroll
        for (int kk = 0; kk < SP; ++kk) { fv[kk] = f[kk * EPG]; wv[kk] = wq[kk]; }
        float tot = 0.f;
#pragma unroll
        for (int kk = 0; kk < SP; kk += 2)                  // pair (kk, kk + 1) as the sequential kernel forms it
          tot += fmaf(wv[kk + 1], fv[kk + 1], fmaf(wv[kk], fv[kk], 0.f));
        zs[(buf * 3 + q) * EPG + j] = tot;
      }
      __syncthreads();
      float zp[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) zp[p] = zs[(buf * 3 + p) * EPG + el] * xT[p] * .5f + h0;
      buf ^= 1;
      if (zp[0] > zt) {
        xmax = xm;
        if (zp[1] > zt) xmax = xp[1]; else xmin = xp[1];
      } else {
        xmin = xm;
        if (zp[2] > zt) xmax = xp[2]; else xmin = xp[2];
      }
    }
    if (valid && q == 0 && wave == 0 && j < EPG) store_inverse(a, e, (xmax + xmin) * .5f);
  }
}

// ---------------------------------------------------------------------------------------
// Backward chain kernel.  Vector gradients kept as per-lane partials over the wave's whole
// persistent loop:  slot 0: d wL, 1: d w1x, 2+l: d b_l (l = 0..NH-1);  + scalar d bL.
// ---------------------------------------------------------------------------------------
// WMODE 0: weight fragments stream from L1/L2;  1: whole padded image LDS-resident (small nets);
//       2: one hidden->hidden matrix at a time in LDS, swapped in before each layer pass by the whole workgroup
//          (wide nets: the image does not fit, and L2 fragment loads starve the MFMA pipe)
// ONES (narrow nets whose hidden widths leave a padding column): the staged layer inputs carry a constant 1 in column
// HP-1, so the weight-gradient GEMM returns the bias gradient in that column and the 3 x HT x 4 bias partial registers
// go away -- together with reading W^T out of the untransposed LDS-resident matrices (WMODE 4, 54 KB instead of the
// 89 KB image with transposes) that lets two workgroups share a CU (2 wavefronts per SIMD).
// INDW (narrow nets, WMODE 1): nothing is staged in HBM at all.  Each wavefront keeps the layer inputs and the dpre of
// its 16 elements in LDS, element-major, reads them back as MFMA operands with K = the 16 elements, and carries the
// (NH-1) x HP x HP weight-gradient accumulators in registers for its whole persistent loop (one wavefront per SIMD:
// 512 registers); the constant-1 column gives the bias gradients as with ONES.
constexpr int kTS = 64;             // row pitch of the element-major LDS tiles (no padding: XOR-swizzled, see tile_w / tile_r)
// Element-major tile of 16 elements x 64 units.  A lane (q, j) of the MFMA C/D layout WRITES units 16t+4q..+3 of element j
// (one b128; the dpre / activation tiles are read back the same way as b128) and later READS, as an MFMA operand with
// K = elements, unit 16t+j of element 4s+q (one b32).  Unit u of element e is stored at column
//     16 ((u>>4) ^ (e&3)) + 4 (((u>>2)&3) ^ ((e>>1)&3)) + (u&3).
// Round 6 -- the lane groups the LDS really services (MI355X_MICROARCH.md): ds_write_b128 in groups of 8 CONTIGUOUS lanes on
// 32 banks, ds_read_b128 in four NON-contiguous groups of 16 on 64 banks, ds_read_b32 in two halves of 32 lanes on 32 banks.
//   write, lanes j = 8g .. 8g+7 of one q: (bit 0 of t ^ j, q ^ (j>>1)&3) is distinct over the eight -> 8 x 4 banks, all 32;
//   b128 read-back, group {j 0-3, 12-15 at q} + {j 4-11 at q+1}: slot 4 ((t ^ j)&3) + (q ^ (j>>1)&3) is distinct over the 16;
//   operand read, elements 4s + {0, 1} (resp. {2, 3}) in a 32-lane half: e&1 picks the 16-bank half, j the bank in it.
// (Rounds 1-5 used (e>>2) in place of ((e>>1)&3): right for 16 contiguous lanes on 64 banks, 2-way for the groups above.)
#ifdef GNF_MONO_OLD_LDS
__device__ __forceinline__ int tile_w(int j, int t, int q) { return j * kTS + 16 * (t ^ (j & 3)) + 4 * (q ^ (j >> 2)); }
__device__ __forceinline__ int tile_r(int s, int q, int t, int j) {
  return (4 * s + q) * kTS + 16 * (t ^ q) + 4 * ((j >> 2) ^ s) + (j & 3);
}
#else
__device__ __forceinline__ int tile_w(int j, int t, int q) { return j * kTS + 16 * (t ^ (j & 3)) + 4 * (q ^ ((j >> 1) & 3)); }
__device__ __forceinline__ int tile_r(int s, int q, int t, int j) {
  return (4 * s + q) * kTS + 16 * (t ^ q) + 4 * ((j >> 2) ^ ((2 * s + (q >> 1)) & 3)) + (j & 3);
}
#endif
template <int HT, int NH, int WMODE, bool ONES = false, bool INDW = false>
__global__ __launch_bounds__(64 * kWaves, (ONES && !INDW) ? 2 : 1) void mono_bwd_k(MonoArgs a) {
  static_assert(!INDW || (ONES && WMODE == 1 && HT <= 4), "in-kernel weight gradients: narrow nets, resident image");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // WMODE 0: weights from L2.  1: the whole pack (incl. transposes) in LDS (narrow nets).  2: ONE hidden->hidden matrix in
  // LDS at a time, swapped by the workgroup.  3: all hidden->hidden matrices resident in LDS.  Modes 2 and 3 keep only
  // the untransposed matrices: the backward reads W^T fragments as 4 ds_read_b32 at stride LDW (lanes j consecutive, the
  // two q of a 32-lane group 16 banks apart: conflict-free), so a node needs the matrices in the order
  // W1 .. W_{NH-1} | W_{NH-1} .. W1 and the one the forward loaded last / the backward used last is still there:
  // 2 swaps per node instead of 4 at NH = 3, none in mode 3.
  // 4: the forward part of the pack (small vectors, W1h, the untransposed hidden matrices) plus W1h^T resident, W^T read
  //    out of the untransposed matrices as in 3 (54 KB at H <= 64 instead of 89 KB: two workgroups per CU)
  constexpr bool WLDS = WMODE == 1, SWAP = WMODE == 2, RES = WMODE == 3, FRES = WMODE == 4, UNT = SWAP || RES || FRES;
  // hand-pipelined weight fragments for 6 <= HT <= 8 (H = 81..128, one wavefront per SIMD, both matrices resident): cfg2's
  // backward 2.38 -> 2.30 ms.  At HT = 10 the second fragment buffer spills (5 -> 65 registers) and buys nothing: that
  // kernel is bound by its staging traffic, not by LDS latency
  constexpr bool PIPE = (SWAP || RES) && HT >= 6 && HT <= 8;
  constexpr int HB = (HT + 1) / 2;                           // fragments per pipeline stage
  constexpr int NUT = (HT + HB - 1) / HB;                    // stages per k-tile
  const MonoLayout& L = a.L;
  const float* wp = a.pack;
  if (WLDS) {
    for (int i = threadIdx.x * 4; i < L.total_floats; i += blockDim.x * 4)
      *reinterpret_cast<f32x4*>(smem + i) = ld4(a.pack + i);
    __syncthreads();
    wp = smem;
  }
  const float* w1ht = wp + L.o_W1hT;
  if (FRES) {
    for (int i = threadIdx.x * 4; i < L.fwd_floats; i += blockDim.x * 4)
      *reinterpret_cast<f32x4*>(smem + i) = ld4(a.pack + i);
    for (int i = threadIdx.x * 4; i < L.CP * L.LDW; i += blockDim.x * 4)
      *reinterpret_cast<f32x4*>(smem + L.fwd_floats + i) = ld4(a.pack + L.o_W1hT + i);
    __syncthreads();
    wp = smem;
    w1ht = smem + L.fwd_floats;
  }
  const int matf = L.HP * L.LDW;                 // floats per hidden->hidden matrix
  auto copy_mat = [&](int off, float* dst) {     // every 1 KB piece in flight at once, no staging registers
    for (int i = threadIdx.x * 4; i < matf; i += blockDim.x * 4) glds16(a.pack + off + i, dst + i);
    __builtin_amdgcn_s_waitcnt(0);
  };
  if (RES) {
    for (int l = 1; l < NH; ++l) copy_mat(L.o_W[l], smem + (l - 1) * matf);
    __syncthreads();
  }
  int resident = 0;                              // SWAP: layer whose matrix is in LDS (0: none); workgroup-uniform
  // every wave of the workgroup calls get_mat at the same points with the same l
  auto get_mat = [&](int l) -> const float* {
    if (RES) return smem + (l - 1) * matf;
    if (FRES) return smem + L.o_W[l];
    if (resident != l) {
      __syncthreads();                            // previous matrix no longer read
      copy_mat(L.o_W[l], smem);
      __syncthreads();
      resident = l;
    }
    return smem;
  };
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int HP = 16 * HT;
  const int64_t ngroups = (a.ecount + 15) / 16;
  const float fS = (float)a.S;

  // bias gradients: per-lane register partials for small nets; for wide nets (register pressure) they are
  // column sums of the staged dpre / Dsum arrays, taken by the host-side row-sum launches instead
  constexpr bool BREG = HT <= 4 && !ONES && !INDW;
  f32x4 p_wL[HT], p_w1x[HT], p_b[BREG ? NH : 1][BREG ? HT : 1];
  float p_bL = 0.f;
#pragma unroll
  for (int t = 0; t < HT; ++t) {
    p_wL[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    p_w1x[t] = p_wL[t];
    if constexpr (BREG) {
#pragma unroll
      for (int l = 0; l < NH; ++l) p_b[l][t] = p_wL[t];
    }
  }

  // INDW: weight-gradient accumulators dW_l[out tile ti][in tile tn] (D layout: out = 16 ti + 4q + r, in = 16 tn + j) and
  // this wavefront's element-major tiles behind the weight image: inputs of layers 1..NH-1, then one dpre tile
  f32x4 accW[INDW ? NH - 1 : 1][INDW ? HT : 1][INDW ? HT : 1];
  float* tiles = smem + (L.total_floats + 3) / 4 * 4 + wave * (NH * 16 * kTS);
  if constexpr (INDW) {
#pragma unroll
    for (int l = 0; l < NH - 1; ++l)
#pragma unroll
      for (int ti = 0; ti < HT; ++ti)
#pragma unroll
        for (int tn = 0; tn < HT; ++tn) accW[l][ti][tn] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < NH * 16 * kTS; i += 64) tiles[i] = 0.f;
  }

  // all waves of a workgroup run the same number of iterations (SWAP needs workgroup barriers inside); a wave
  // without a group of its own recomputes the last group with zero cotangents and writes nothing new
  for (int64_t g0 = (int64_t)blockIdx.x * kWaves; g0 < ngroups; g0 += (int64_t)gridDim.x * kWaves) {
    const bool gvalid = g0 + wave < ngroups;
    const int64_t grp = gvalid ? g0 + wave : ngroups - 1;
    const int64_t el = grp * 16 + j;                 // element index inside the chunk
    const bool valid = gvalid && el < a.ecount;
    const int64_t e = a.e0 + (el < a.ecount ? el : a.ecount - 1);
    const int64_t b = e / a.d, i = e - b * a.d;
    const int64_t hbase = b * a.h_sb + i * a.h_sd;
    f32x4 c1[HT];
    cond_bias<HT>(wp, L, a.h, hbase, a.h_sc, q, j, c1);
    const float xv = a.x[e];
    const float xT = fS * (xv / fS);
    const float g_z = valid ? a.gz[e] : 0.f;
    const float g_j = (valid && a.gjac) ? a.gjac[e] : 0.f;
    const float cotq = g_z * xT * .5f;               // grad_out * (xT - x0)/2
    f32x4 Ds[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) Ds[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dx = 0.f;

    for (int k = 0; k < a.NK; ++k) {
      const bool isq = k <= a.S, isj = k == a.S + 1;
      const float xk = isq ? xT * (a.cct[k] + 1.f) * .5f : xv;
      const float cot = isq ? a.ccw[k] * cotq : (isj ? g_j : 0.f);
      const int64_t row = (grp * a.NK + k) * 16 + j;

      // ---- forward recompute, remembering ReLU masks; stage the inputs of layers 1..NH-1
      f32x4 act[HT];
      unsigned long long msk[NH];
      msk[0] = 0ull;
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 wx = ld4(wp + L.o_w1x + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pre = fmaf(wx[r], xk, c1[t][r]);
          act[t][r] = fmaxf(pre, 0.f);
          if (!INDW && pre > 0.f) msk[0] |= 1ull << (4 * t + r);
        }
      }
#pragma unroll
      for (int l = 1; l < NH; ++l) {
        // the matrix first, the staging stores behind it: a swap ends in s_waitcnt vmcnt(0), which would also wait for
        // the acknowledgement of stores issued just in front of it (the layer input stays in registers as the B operand)
        const float* W = UNT ? get_mat(l) : wp + L.o_W[l];
        if constexpr (INDW) {                   // layer input, element-major, into this wavefront's LDS tile
          float* ta = tiles + (l - 1) * 16 * kTS;
#pragma unroll
          for (int t = 0; t < HT; ++t) {
            f32x4 v = act[t];
            if (t == HT - 1 && q == 3) v[3] = 1.f;              // column HP-1 = 1: dW_l[:, HP-1] is the bias gradient
            *reinterpret_cast<f32x4*>(ta + tile_w(j, t, q)) = v;
          }
        } else {
          float* sa = a.SA[l] + row * HP + 4 * q;
          if (gvalid) {                         // a wave re-running the last group must not clobber its owner's rows
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              f32x4 v = act[t];
              if (ONES && t == HT - 1 && q == 3) v[3] = 1.f;    // column HP-1: bias gradient rides the dW GEMM
              *reinterpret_cast<f32x4*>(sa + 16 * t) = v;
            }
          }
        }
        f32x4 o[HT];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) o[mt] = ld4(wp + L.o_b[l] + 16 * mt + 4 * q);
        if constexpr (PIPE) {
          // wide nets, ONE wavefront per SIMD: nothing hides an LDS round trip, so the weight fragments are double-buffered
          // by hand in half rows of tiles -- the reads of half row u+1 are issued in front of the MFMAs of half row u
          // (2 x HB fragments live = the HT of the block-at-a-time form it replaces, which exposed every block's latency)
          f32x4 Af[2][HB];
          auto loadA = [&](int u, f32x4 (&dst)[HB]) {
            const int t = u / NUT, m0 = (u % NUT) * HB;
#pragma unroll
            for (int i = 0; i < HB; ++i)
              if (m0 + i < HT) dst[i] = ld4(W + (16 * (m0 + i) + j) * L.LDW + 16 * t + 4 * q);
          };
          loadA(0, Af[0]);
#pragma unroll
          for (int u = 0; u < NUT * HT; ++u) {
            if (u + 1 < NUT * HT) loadA(u + 1, Af[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int t = u / NUT, m0 = (u % NUT) * HB;
#pragma unroll
            for (int i = 0; i < HB; ++i)
              if (m0 + i < HT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[m0 + i] = mfma(Af[u & 1][i][r], act[t][r], o[m0 + i]);
              }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
#pragma unroll
          for (int mt = 0; mt < HT; ++mt) {
            const f32x4 A = ld4(W + (16 * mt + j) * L.LDW + 16 * t + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[mt] = mfma(A[r], act[t][r], o[mt]);
          }
          // wide nets: keep the scheduler from hoisting all HT^2 weight fragments (400+ VGPRs) at once
          if constexpr (WMODE != 0) __builtin_amdgcn_sched_barrier(0);   // LDS weights: bound the fragment hoisting (no spills)
        }
        }
        msk[l] = 0ull;
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            act[t][r] = fmaxf(o[t][r], 0.f);
            if (!INDW && o[t][r] > 0.f) msk[l] |= 1ull << (4 * t + r);
          }
      }
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 wl = ld4(wp + L.o_wL + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) s = fmaf(wl[r], act[t][r], s);
      }
      s = qsum(s) + wp[L.o_bL];
      const float f = elu_plus(s);
      if (isj) dx = g_z * f;                                   // Leibniz rule: dz/dx = f(x;h)

      // ---- backward through the last layer and the ELU
      const float dpl = cot * (s > 0.f ? 1.f : expf(s));
      if (q == 0) p_bL += dpl;
      f32x4 dp[HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 wl = ld4(wp + L.o_wL + 16 * t + 4 * q);     // re-read (L1/LDS hit) rather than held across the node
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p_wL[t][r] = fmaf(dpl, act[t][r], p_wL[t][r]);
          // INDW: the ReLU gates come from the activations themselves (registers here, the LDS tiles below) instead
          // of 64-bit masks built and tested bit by bit
          const bool on = INDW ? act[t][r] > 0.f : (bool)((msk[NH - 1] >> (4 * t + r)) & 1ull);
          dp[t][r] = on ? wl[r] * dpl : 0.f;
        }
      }
      // ---- hidden->hidden layers, top down
#pragma unroll
      for (int l = NH - 1; l >= 1; --l) {
        const float* WT = UNT ? get_mat(l) : wp + L.o_WT[l];      // UNT: the untransposed matrix, read transposed below
        if constexpr (INDW) {
          // dW_l[out][in] += sum over the 16 elements of dpre_l[elem][out] * input_l[elem][in]: K = elements, both
          // operands read back element-major (lane (q, j): element 4 s + q, unit 16 tile + j).  A wave re-running the
          // last group has zero cotangents, so it adds nothing.
          float* td = tiles + (NH - 1) * 16 * kTS;
#pragma unroll
          for (int t = 0; t < HT; ++t) *reinterpret_cast<f32x4*>(td + tile_w(j, t, q)) = dp[t];
          const float* ta = tiles + (l - 1) * 16 * kTS;
#pragma unroll
          for (int sK = 0; sK < 4; ++sK) {
            float fa[HT], fb[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) {
              fa[t] = td[tile_r(sK, q, t, j)];
              fb[t] = ta[tile_r(sK, q, t, j)];
            }
#pragma unroll
            for (int ti = 0; ti < HT; ++ti)
#pragma unroll
              for (int tn = 0; tn < HT; ++tn) accW[l - 1][ti][tn] = mfma(fa[ti], fb[tn], accW[l - 1][ti][tn]);
          }
        } else {
          float* sd = a.SD[l] + row * HP + 4 * q;
#pragma unroll
          for (int t = 0; t < HT; ++t) {
            if (gvalid) *reinterpret_cast<f32x4*>(sd + 16 * t) = dp[t];
            if constexpr (BREG) p_b[l][t] += dp[t];
          }
        }
        f32x4 da[HT];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) da[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PIPE) {
          f32x4 Af[2][HB];
          auto loadT = [&](int u, f32x4 (&dst)[HB]) {
            const int t = u / NUT, m0 = (u % NUT) * HB;
#pragma unroll
            for (int i = 0; i < HB; ++i)
              if (m0 + i < HT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[i][r] = WT[(16 * t + 4 * q + r) * L.LDW + 16 * (m0 + i) + j];
              }
          };
          loadT(0, Af[0]);
#pragma unroll
          for (int u = 0; u < NUT * HT; ++u) {
            if (u + 1 < NUT * HT) loadT(u + 1, Af[(u + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int t = u / NUT, m0 = (u % NUT) * HB;
#pragma unroll
            for (int i = 0; i < HB; ++i)
              if (m0 + i < HT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) da[m0 + i] = mfma(Af[u & 1][i][r], dp[t][r], da[m0 + i]);
              }
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
        for (int t = 0; t < HT; ++t) {
#pragma unroll
          for (int mt = 0; mt < HT; ++mt) {
            f32x4 A;
            if constexpr (UNT) {
#pragma unroll
              for (int r = 0; r < 4; ++r) A[r] = WT[(16 * t + 4 * q + r) * L.LDW + 16 * mt + j];
            } else {
              A = ld4(WT + (16 * mt + j) * L.LDW + 16 * t + 4 * q);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) da[mt] = mfma(A[r], dp[t][r], da[mt]);
          }
          if constexpr (WMODE != 0) __builtin_amdgcn_sched_barrier(0);   // LDS weights: bound the fragment hoisting (no spills)
        }
        }
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (INDW) continue;
            dp[t][r] = ((msk[l - 1] >> (4 * t + r)) & 1ull) ? da[t][r] : 0.f;
          }
        if constexpr (INDW) {                   // gate = (input of layer l) > 0, read back from its tile in the layout it was written in
          const float* tg = tiles + (l - 1) * 16 * kTS;
#pragma unroll
          for (int t = 0; t < HT; ++t) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(tg + tile_w(j, t, q));
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[t][r] = g[r] > 0.f ? da[t][r] : 0.f;
          }
        }
      }
      // ---- first layer: rank-1 in x_k, node-independent in h
      float sx = 0.f;
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 wx = ld4(wp + L.o_w1x + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Ds[t][r] += dp[t][r];
          p_w1x[t][r] = fmaf(dp[t][r], xk, p_w1x[t][r]);
          sx = fmaf(wx[r], dp[t][r], sx);
        }
      }
      if (isj) dx += qsum(sx);                                 // gjac * df/dx(x;h)
    }

    // ---- per-group epilogue: d b_0, staged Dsum (for d W1h), dh, dx
    float* ds = a.Dsum + (grp * 16 + j) * HP + 4 * q;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      if constexpr (BREG) p_b[0][t] += Ds[t];
      if (gvalid) *reinterpret_cast<f32x4*>(ds + 16 * t) = Ds[t];
    }
    if (valid && q == 0 && a.gx) a.gx[e] = dx;
    const int64_t gbase = b * a.g_sb + i * a.g_sd;
    for (int mt = 0; mt < L.CP / 16; ++mt) {
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 A = ld4(w1ht + (16 * mt + j) * L.LDW + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) o = mfma(A[r], Ds[t][r], o);
      }
      if (valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cc = 16 * mt + 4 * q + r;
          if (cc < L.c) a.gh[gbase + cc * a.g_sc] = o[r] + (cc == 0 ? g_z : 0.f);   // + gz: the "+ z0" term
        }
      }
    }
  }

  // ---- reduce the per-lane partials over the 16 element lanes, one row per wavefront
  float* prow = a.part + ((int64_t)blockIdx.x * kWaves + wave) * ((NH + 2) * HP + 4);
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int hid = 16 * t + 4 * q + r;
      float v = jsum(p_wL[t][r]);
      if (j == 0) prow[hid] = v;
      v = jsum(p_w1x[t][r]);
      if (j == 0) prow[HP + hid] = v;
#pragma unroll
      for (int l = 0; l < NH; ++l) {
        if constexpr (BREG) {
          v = jsum(p_b[l][t][r]);
          if (j == 0) prow[(2 + l) * HP + hid] = v;
        } else {
          if (j == 0) prow[(2 + l) * HP + hid] = 0.f;
        }
      }
    }
  if constexpr (INDW) {
    float* wrow = a.wpart + ((int64_t)blockIdx.x * kWaves + wave) * ((NH - 1) * HP * HP);
#pragma unroll
    for (int l = 0; l < NH - 1; ++l)
#pragma unroll
      for (int ti = 0; ti < HT; ++ti)
#pragma unroll
        for (int tn = 0; tn < HT; ++tn)
#pragma unroll
          for (int r = 0; r < 4; ++r) wrow[l * HP * HP + (16 * ti + 4 * q + r) * HP + 16 * tn + j] = accW[l][ti][tn][r];
  }
  const float vbl = jsum(p_bL);
  if (lane == 0) { prow[(NH + 2) * HP] = vbl; prow[(NH + 2) * HP + 1] = 0.f; prow[(NH + 2) * HP + 2] = 0.f; prow[(NH + 2) * HP + 3] = 0.f; }
}

// ---------------------------------------------------------------------------------------
// Narrow nets (H <= 64), everything in-kernel, TWO quadrature nodes per pass: every weight fragment read from LDS feeds 8
// MFMAs (as in the forward) and the two nodes' chains are independent.  Same outputs as mono_bwd_k<.., INDW>: dx, dh,
// Dsum rows, vector partials (d wL, d w1x, d bL; the biases come out of the weight-gradient accumulators' ones column
// and the Dsum column sums), and one (NH-1) x HP x HP accumulator row per wavefront.
// LDS: small vectors | W_l, b_l | W_l^T (W1h and W1h^T, used once per group, stay in L2), then per wavefront the
// element-major tiles of the layer inputs of BOTH nodes and one dpre tile (tile_w / tile_r swizzle).
// ---------------------------------------------------------------------------------------
template <int HT, int NH>
__global__ __launch_bounds__(64 * kWaves, 1) void mono_bwd_pair_k(MonoArgs a) {
  static_assert(HT <= 4 && NH >= 2, "narrow nets with at least one hidden->hidden layer");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const MonoLayout& L = a.L;
  constexpr int HP = 16 * HT;
  const int matf = HP * L.LDW;
  // LDS image: [0, o_W1h): w1x, b1, wL, bL;  then per hidden layer W_l (matf) + b_l (HP);  then per hidden layer W_l^T
  const int small = L.o_W1h;
  float* sW = smem + small;                      // W_l at sW + (l-1) * (matf + HP), b_l right behind it
  float* sWT = sW + (NH - 1) * (matf + HP);      // W_l^T at sWT + (l-1) * matf
  for (int i = threadIdx.x * 4; i < small; i += blockDim.x * 4) *reinterpret_cast<f32x4*>(smem + i) = ld4(a.pack + i);
  for (int l = 1; l < NH; ++l) {
    for (int i = threadIdx.x * 4; i < matf + HP; i += blockDim.x * 4)
      *reinterpret_cast<f32x4*>(sW + (l - 1) * (matf + HP) + i) = ld4(a.pack + L.o_W[l] + i);
    for (int i = threadIdx.x * 4; i < matf; i += blockDim.x * 4)
      *reinterpret_cast<f32x4*>(sWT + (l - 1) * matf + i) = ld4(a.pack + L.o_WT[l] + i);
  }
  __syncthreads();
  const float* wp = smem;                        // small vectors only (same offsets as in the pack)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  constexpr int NT = 2 * (NH - 1) + 1;           // tiles per wavefront: layer inputs of node 0 / node 1, one dpre tile
  float* tiles = sWT + (NH - 1) * matf + wave * (NT * 16 * kTS);
  for (int i = lane; i < NT * 16 * kTS; i += 64) tiles[i] = 0.f;
  float* td = tiles + (NT - 1) * 16 * kTS;
  const int64_t ngroups = (a.ecount + 15) / 16;
  const float fS = (float)a.S;

  f32x4 p_wL[HT], p_w1x[HT], accW[NH - 1][HT][HT];
  float p_bL = 0.f;
#pragma unroll
  for (int t = 0; t < HT; ++t) { p_wL[t] = f32x4{0.f, 0.f, 0.f, 0.f}; p_w1x[t] = p_wL[t]; }
#pragma unroll
  for (int l = 0; l < NH - 1; ++l)
#pragma unroll
    for (int ti = 0; ti < HT; ++ti)
#pragma unroll
      for (int tn = 0; tn < HT; ++tn) accW[l][ti][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int64_t grp = (int64_t)blockIdx.x * kWaves + wave; grp < ngroups; grp += (int64_t)gridDim.x * kWaves) {
    const int64_t el = grp * 16 + j;
    const bool valid = el < a.ecount;
    const int64_t e = a.e0 + (valid ? el : a.ecount - 1);
    const int64_t b = e / a.d, i = e - b * a.d;
    const int64_t hbase = b * a.h_sb + i * a.h_sd;
    f32x4 c1[HT];
    cond_bias<HT>(a.pack, L, a.h, hbase, a.h_sc, q, j, c1);      // W1h from L2, b1 too (once per group)
    const float xv = a.x[e];
    const float xT = fS * (xv / fS);
    const float g_z = valid ? a.gz[e] : 0.f;
    const float g_j = (valid && a.gjac) ? a.gjac[e] : 0.f;
    const float cotq = g_z * xT * .5f;
    f32x4 Ds[HT];
#pragma unroll
    for (int t = 0; t < HT; ++t) Ds[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dx = 0.f;

    for (int k = 0; k < a.NK; k += 2) {
      float xk[2], cot[2];
      bool isj[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ku = k + u;
        const bool isq = ku <= a.S;
        isj[u] = ku == a.S + 1;
        xk[u] = isq ? xT * (a.cct[ku] + 1.f) * .5f : xv;
        cot[u] = isq ? a.ccw[ku] * cotq : (isj[u] ? g_j : 0.f);
      }
      // ---- forward recompute of both nodes
      f32x4 act[2][HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 wx = ld4(wp + L.o_w1x + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          act[0][t][r] = fmaxf(fmaf(wx[r], xk[0], c1[t][r]), 0.f);
          act[1][t][r] = fmaxf(fmaf(wx[r], xk[1], c1[t][r]), 0.f);
        }
      }
#pragma unroll
      for (int l = 1; l < NH; ++l) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float* ta = tiles + (2 * (l - 1) + u) * 16 * kTS;
#pragma unroll
          for (int t = 0; t < HT; ++t) {
            f32x4 v = act[u][t];
            if (t == HT - 1 && q == 3) v[3] = 1.f;              // column HP-1 = 1: the bias gradient rides dW_l
            *reinterpret_cast<f32x4*>(ta + tile_w(j, t, q)) = v;
          }
        }
        const float* W = sW + (l - 1) * (matf + HP);
        f32x4 o[2][HT];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) { o[0][mt] = ld4(W + matf + 16 * mt + 4 * q); o[1][mt] = o[0][mt]; }
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
          for (int mt = 0; mt < HT; ++mt) {
            const f32x4 A = ld4(W + (16 * mt + j) * L.LDW + 16 * t + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              o[0][mt] = mfma(A[r], act[0][t][r], o[0][mt]);
              o[1][mt] = mfma(A[r], act[1][t][r], o[1][mt]);
            }
          }
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) { act[0][t][r] = fmaxf(o[0][t][r], 0.f); act[1][t][r] = fmaxf(o[1][t][r], 0.f); }
      }
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 wl = ld4(wp + L.o_wL + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) { s0 = fmaf(wl[r], act[0][t][r], s0); s1 = fmaf(wl[r], act[1][t][r], s1); }
      }
      const float bL = wp[L.o_bL];
      float sv[2] = {qsum(s0) + bL, qsum(s1) + bL}, dpl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (isj[u]) dx = g_z * elu_plus(sv[u]);                  // Leibniz rule: dz/dx = f(x;h)
        dpl[u] = cot[u] * (sv[u] > 0.f ? 1.f : expf(sv[u]));
        if (q == 0) p_bL += dpl[u];
      }
      // ---- backward through the last layer
      f32x4 dp[2][HT];
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 wl = ld4(wp + L.o_wL + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            p_wL[t][r] = fmaf(dpl[u], act[u][t][r], p_wL[t][r]);
            dp[u][t][r] = act[u][t][r] > 0.f ? wl[r] * dpl[u] : 0.f;
          }
      }
      // ---- hidden->hidden layers, top down
#pragma unroll
      for (int l = NH - 1; l >= 1; --l) {
        // dW_l += dpre_l^T * input_l, K = the 16 elements, node by node through the one dpre tile
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int t = 0; t < HT; ++t) *reinterpret_cast<f32x4*>(td + tile_w(j, t, q)) = dp[u][t];
          const float* ta = tiles + (2 * (l - 1) + u) * 16 * kTS;
#pragma unroll
          for (int sK = 0; sK < 4; ++sK) {
            float fa[HT], fb[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) { fa[t] = td[tile_r(sK, q, t, j)]; fb[t] = ta[tile_r(sK, q, t, j)]; }
#pragma unroll
            for (int ti = 0; ti < HT; ++ti)
#pragma unroll
              for (int tn = 0; tn < HT; ++tn) accW[l - 1][ti][tn] = mfma(fa[ti], fb[tn], accW[l - 1][ti][tn]);
          }
        }
        const float* WT = sWT + (l - 1) * matf;
        f32x4 da[2][HT];
#pragma unroll
        for (int mt = 0; mt < HT; ++mt) { da[0][mt] = f32x4{0.f, 0.f, 0.f, 0.f}; da[1][mt] = da[0][mt]; }
#pragma unroll
        for (int t = 0; t < HT; ++t)
#pragma unroll
          for (int mt = 0; mt < HT; ++mt) {
            const f32x4 A = ld4(WT + (16 * mt + j) * L.LDW + 16 * t + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              da[0][mt] = mfma(A[r], dp[0][t][r], da[0][mt]);
              da[1][mt] = mfma(A[r], dp[1][t][r], da[1][mt]);
            }
          }
        // gate = (input of layer l) > 0, read back from its tile in the layout it was written in
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float* tg = tiles + (2 * (l - 1) + u) * 16 * kTS;
#pragma unroll
          for (int t = 0; t < HT; ++t) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(tg + tile_w(j, t, q));
#pragma unroll
            for (int r = 0; r < 4; ++r) dp[u][t][r] = g[r] > 0.f ? da[u][t][r] : 0.f;
          }
        }
      }
      // ---- first layer: rank-1 in x_k, node-independent in h
      float sx0 = 0.f, sx1 = 0.f;
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 wx = ld4(wp + L.o_w1x + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Ds[t][r] += dp[0][t][r] + dp[1][t][r];
          p_w1x[t][r] = fmaf(dp[0][t][r], xk[0], p_w1x[t][r]);
          p_w1x[t][r] = fmaf(dp[1][t][r], xk[1], p_w1x[t][r]);
          sx0 = fmaf(wx[r], dp[0][t][r], sx0);
          sx1 = fmaf(wx[r], dp[1][t][r], sx1);
        }
      }
      if (isj[0]) dx += qsum(sx0);                               // gjac * df/dx(x;h)
      if (isj[1]) dx += qsum(sx1);
    }

    // ---- per-group epilogue: staged Dsum (for d W1h and d b_0), dh, dx
    float* ds = a.Dsum + (grp * 16 + j) * HP + 4 * q;
#pragma unroll
    for (int t = 0; t < HT; ++t) *reinterpret_cast<f32x4*>(ds + 16 * t) = Ds[t];
    if (valid && q == 0 && a.gx) a.gx[e] = dx;
    const int64_t gbase = b * a.g_sb + i * a.g_sd;
    for (int mt = 0; mt < L.CP / 16; ++mt) {
      f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 A = ld4(a.pack + L.o_W1hT + (16 * mt + j) * L.LDW + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) o = mfma(A[r], Ds[t][r], o);
      }
      if (valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cc = 16 * mt + 4 * q + r;
          if (cc < L.c) a.gh[gbase + cc * a.g_sc] = o[r] + (cc == 0 ? g_z : 0.f);
        }
      }
    }
  }

  float* prow = a.part + ((int64_t)blockIdx.x * kWaves + wave) * ((NH + 2) * HP + 4);
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int hid = 16 * t + 4 * q + r;
      float v = jsum(p_wL[t][r]);
      if (j == 0) prow[hid] = v;
      v = jsum(p_w1x[t][r]);
      if (j == 0) prow[HP + hid] = v;
#pragma unroll
      for (int l = 0; l < NH; ++l)
        if (j == 0) prow[(2 + l) * HP + hid] = 0.f;
    }
  const float vbl = jsum(p_bL);
  if (lane == 0) { prow[(NH + 2) * HP] = vbl; prow[(NH + 2) * HP + 1] = 0.f; prow[(NH + 2) * HP + 2] = 0.f; prow[(NH + 2) * HP + 3] = 0.f; }
  float* wrow = a.wpart + ((int64_t)blockIdx.x * kWaves + wave) * ((NH - 1) * HP * HP);
#pragma unroll
  for (int l = 0; l < NH - 1; ++l)
#pragma unroll
    for (int ti = 0; ti < HT; ++ti)
#pragma unroll
      for (int tn = 0; tn < HT; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) wrow[l * HP * HP + (16 * ti + 4 * q + r) * HP + 16 * tn + j] = accW[l][ti][tn][r];
}

// ---------------------------------------------------------------------------------------
// Two-node backward of a PEELED narrow net (see eval2x): the HM full tiles run on the MFMA exactly as in
// mono_bwd_pair_k with HT = HM (9 instead of 16 tile products per layer in each of the three contractions), the EX
// leftover units U = 16 HM + e are per-element scalars replicated over the four q-lanes:
//   forward      pre[U] = b + q-sum of the lane's partial dot with the main inputs + W[U][U'] x[U'];  main outputs get the
//                rank-EX update W[:, U] x[U]
//   dW, db       stay on the MFMA in the padded 4-tile form (K = the 16 elements: nothing to skip there, and per-lane
//                partial accumulators for the peeled rows / columns / biases would cost 132 registers: measured, they
//                spill); the peeled units and the constant 1 of the bias column are written into the fourth tile.
//                RP (row peel): the fourth OUT tile alone -- the rows dW[U][:], db[U] -- leaves the MFMA: 12 instead of
//                16 tile products per layer and node, 15 per-lane partials per peeled row (dpre[U] x the lane's own
//                inputs, which the data-gradient section reads back for its gates anyway) instead of 16 accumulator
//                registers per dropped tile: +28 registers at NH = 3, which fit
//   da           main inputs: W^T on the MFMA + rank-EX update W[U][:] dpre[U];  inputs U: q-sum of the lane's partial dot
//                of W[:, U] with the main dpre + the corner
// The pack, the LDS image, the per-wavefront element-major tiles and every output (accumulator rows in the padded
// [HP][HP] form with the bias gradient in column HP-1, Dsum, the partial vector row) are those of mono_bwd_pair_k.
// ---------------------------------------------------------------------------------------
// SP (round 6): the 48 x 48 main blocks of the recompute and of the data gradient on the bf16 matrix pipe (block48_split:
// activations / dpre split in the C/D registers, weights as LDS-resident bf16 planes of W_l and W_l^T from the pack); the
// weight-gradient contraction (K = the 16 elements) and everything peeled stay fp32.  LDS then holds, per layer, only the
// peeled rows of W_l and W_l^T and the bias in fp32 (the fp32 main blocks are gone: 157 -> 143 KB).
template <int HM, int NH, int EX, bool RP = false, bool SP = false>
__global__ __launch_bounds__(64 * kWaves, 1) void mono_bwd_pair_x_k(MonoArgs a) {
  static_assert(HM >= 1 && HM <= 3 && NH >= 2 && EX >= 1 && EX <= 3, "peeled narrow nets");
  static_assert(!SP || HM == 3, "split form: 48-unit main block");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const MonoLayout& L = a.L;
  #ifdef GNF_MONO_OLD_LDS
  constexpr int HT = HM + 1, HP = 16 * HT, LDW = HP + 4, U0 = 16 * HM;
#else
  constexpr int HT = HM + 1, HP = 16 * HT, LDW = HP + 8, U0 = 16 * HM;       // (= MonoLayout::LDW of a narrow net)
#endif
  constexpr int HTO = RP ? HM : HT;              // out tiles of the weight gradient on the MFMA
  constexpr int NPR = RP ? NH - 1 : 1;
  constexpr int matf = HP * LDW;
  const int small = L.o_W1h;
  float* sW = smem + small;                      // W_l at sW + (l-1) * (matf + HP), b_l right behind it
  float* sWT = sW + (NH - 1) * (matf + HP);      // W_l^T at sWT + (l-1) * matf
  // SP: per layer [4 peeled rows of W_l][4 peeled rows of W_l^T][b_l] = kRS floats at sW, the planes of W_1.., W_1^T.. behind them
  constexpr int kRS = 8 * LDW + HP;
  float* sQ = sW + (NH - 1) * kRS;               // SP: (NH-1) x kNarrowQ words of W_l planes, then as many of W_l^T planes
  for (int i = threadIdx.x * 4; i < small; i += blockDim.x * 4) *reinterpret_cast<f32x4*>(smem + i) = ld4(a.pack + i);
  if constexpr (SP) {
    for (int l = 1; l < NH; ++l) {
      float* rs = sW + (l - 1) * kRS;
      for (int i = threadIdx.x * 4; i < 4 * LDW; i += blockDim.x * 4) {
        *reinterpret_cast<f32x4*>(rs + i) = ld4(a.pack + L.o_W[l] + U0 * LDW + i);
        *reinterpret_cast<f32x4*>(rs + 4 * LDW + i) = ld4(a.pack + L.o_WT[l] + U0 * LDW + i);
      }
      for (int i = threadIdx.x * 4; i < HP; i += blockDim.x * 4) *reinterpret_cast<f32x4*>(rs + 8 * LDW + i) = ld4(a.pack + L.o_b[l] + i);
    }
    for (int i = threadIdx.x * 4; i < (NH - 1) * kNarrowQ; i += blockDim.x * 4) {
      *reinterpret_cast<f32x4*>(sQ + i) = ld4(a.pack + L.o_Wq[1] + i);
      *reinterpret_cast<f32x4*>(sQ + (NH - 1) * kNarrowQ + i) = ld4(a.pack + L.o_WTq[1] + i);
    }
  } else {
    for (int l = 1; l < NH; ++l) {
      for (int i = threadIdx.x * 4; i < matf + HP; i += blockDim.x * 4)
        *reinterpret_cast<f32x4*>(sW + (l - 1) * (matf + HP) + i) = ld4(a.pack + L.o_W[l] + i);
      for (int i = threadIdx.x * 4; i < matf; i += blockDim.x * 4)
        *reinterpret_cast<f32x4*>(sWT + (l - 1) * matf + i) = ld4(a.pack + L.o_WT[l] + i);
    }
  }
  __syncthreads();
  const float* wp = smem;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  constexpr int NT = 2 * (NH - 1) + 1;           // tiles per wavefront: layer inputs of node 0 / node 1, one dpre tile
  float* tiles = (SP ? sQ + 2 * (NH - 1) * kNarrowQ : sWT + (NH - 1) * matf) + wave * (NT * 16 * kTS);
  for (int i = lane; i < NT * 16 * kTS; i += 64) tiles[i] = 0.f;
  float* td = tiles + (NT - 1) * 16 * kTS;
  const int64_t ngroups = (a.ecount + 15) / 16;
  const float fS = (float)a.S;

  f32x4 p_wL[HM], p_w1x[HM], accW[NH - 1][HTO][HT];     // dW_l in the padded form (all 4 in tiles; 4 or, RP, 3 out tiles)
  float p_wLx[EX], p_w1xx[EX];                            // d wL[U], d w1x[U] (same value in all q lanes)
  float p_bL = 0.f;
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  // RP: per-lane partials of the peeled rows: dW[U][the lane's main inputs], dW[U][U'], db[U] (the last two: the same
  // value in all q lanes)
  f32x4 prw[NPR][EX][HM];
  float prx[NPR][EX][EX], prb[NPR][EX];
#pragma unroll
  for (int l = 0; l < NPR; ++l)
#pragma unroll
    for (int ee = 0; ee < EX; ++ee) {
#pragma unroll
      for (int t = 0; t < HM; ++t) prw[l][ee][t] = z4;
#pragma unroll
      for (int e2 = 0; e2 < EX; ++e2) prx[l][ee][e2] = 0.f;
      prb[l][ee] = 0.f;
    }
#pragma unroll
  for (int t = 0; t < HM; ++t) { p_wL[t] = z4; p_w1x[t] = z4; }
#pragma unroll
  for (int e = 0; e < EX; ++e) { p_wLx[e] = 0.f; p_w1xx[e] = 0.f; }
#pragma unroll
  for (int l = 0; l < NH - 1; ++l)
#pragma unroll
    for (int ti = 0; ti < HTO; ++ti)
#pragma unroll
      for (int tn = 0; tn < HT; ++tn) accW[l][ti][tn] = z4;

  for (int64_t grp = (int64_t)blockIdx.x * kWaves + wave; grp < ngroups; grp += (int64_t)gridDim.x * kWaves) {
    const int64_t el = grp * 16 + j;
    const bool valid = el < a.ecount;
    const int64_t e = a.e0 + (valid ? el : a.ecount - 1);
    const int64_t b = e / a.d, i = e - b * a.d;
    const int64_t hbase = b * a.h_sb + i * a.h_sd;
    f32x4 c1[HM];
    float c1x[EX];
    cond_bias_x<HM, EX>(a.pack, L, a.h, hbase, a.h_sc, q, j, c1, c1x);     // W1h from L2 (once per group)
    const float xv = a.x[e];
    const float xT = fS * (xv / fS);
    const float g_z = valid ? a.gz[e] : 0.f;
    const float g_j = (valid && a.gjac) ? a.gjac[e] : 0.f;
    const float cotq = g_z * xT * .5f;
    f32x4 Ds[HM];
    float Dsx[EX];
#pragma unroll
    for (int t = 0; t < HM; ++t) Ds[t] = z4;
#pragma unroll
    for (int ee = 0; ee < EX; ++ee) Dsx[ee] = 0.f;
    float dx = 0.f;

    for (int k = 0; k < a.NK; k += 2) {
      float xk[2], cot[2];
      bool isj[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ku = k + u;
        const bool isq = ku <= a.S;
        isj[u] = ku == a.S + 1;
        xk[u] = isq ? xT * (a.cct[ku] + 1.f) * .5f : xv;
        cot[u] = isq ? a.ccw[ku] * cotq : (isj[u] ? g_j : 0.f);
      }
      // ---- forward recompute of both nodes
      f32x4 act[2][HM];
      float ax[2][EX], axin[NH - 1][2][EX];          // axin[l-1]: peeled inputs of hidden layer l (for gates and dW)
#pragma unroll
      for (int t = 0; t < HM; ++t) {
        const f32x4 wx = ld4(wp + L.o_w1x + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          act[0][t][r] = fmaxf(fmaf(wx[r], xk[0], c1[t][r]), 0.f);
          act[1][t][r] = fmaxf(fmaf(wx[r], xk[1], c1[t][r]), 0.f);
        }
      }
      {
        const f32x4 wx = ld4(wp + L.o_w1x + U0);
#pragma unroll
        for (int ee = 0; ee < EX; ++ee) {
          ax[0][ee] = fmaxf(fmaf(wx[ee], xk[0], c1x[ee]), 0.f);
          ax[1][ee] = fmaxf(fmaf(wx[ee], xk[1], c1x[ee]), 0.f);
        }
      }
#pragma unroll
      for (int l = 1; l < NH; ++l) {
        const float* W = sW + (l - 1) * (matf + HP);         // (fp32 form only)
        // the peeled rows of W_l and W_l^T and the bias: inside the full images, or (SP) the compact per-layer slice
        const float* Wrow = SP ? sW + (l - 1) * kRS : W + U0 * LDW;
        const float* WTrow = SP ? sW + (l - 1) * kRS + 4 * LDW : sWT + (l - 1) * matf + U0 * LDW;
        const float* Wbias = SP ? sW + (l - 1) * kRS + 8 * LDW : W + matf;
        f32x4 wr[EX][HM], wxr[EX], wc[EX][HM];      // W[U][k] (k = lane's units), W[U][U'], W[o][U] (o = lane's units)
        f32x4 bx;
        auto load_peeled = [&]() {
          bx = ld4(Wbias + U0);
#pragma unroll
          for (int ee = 0; ee < EX; ++ee) {
#pragma unroll
            for (int t = 0; t < HM; ++t) {
              wr[ee][t] = ld4(Wrow + ee * LDW + 16 * t + 4 * q);
              wc[ee][t] = ld4(WTrow + ee * LDW + 16 * t + 4 * q);
            }
            wxr[ee] = ld4(Wrow + ee * LDW + U0);
          }
        };
        if constexpr (!SP) load_peeled();            // fp32 form: ahead of the MFMA block (their latency hides under it)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float* ta = tiles + (2 * (l - 1) + u) * 16 * kTS;
#pragma unroll
          for (int t = 0; t < HM; ++t) *reinterpret_cast<f32x4*>(ta + tile_w(j, t, q)) = act[u][t];
          f32x4 vx = z4;                                       // tile HM: the peeled units (lanes q = 0), column HP-1 = 1
#pragma unroll
          for (int ee = 0; ee < EX; ++ee) { vx[ee] = q == 0 ? ax[u][ee] : 0.f; axin[l - 1][u][ee] = ax[u][ee]; }
          if (q == 3) vx[3] = 1.f;
          *reinterpret_cast<f32x4*>(ta + tile_w(j, HM, q)) = vx;
        }
        f32x4 o[2][HM];
#pragma unroll
        for (int mt = 0; mt < HM; ++mt) { o[0][mt] = ld4(Wbias + 16 * mt + 4 * q); o[1][mt] = o[0][mt]; }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SP) {
          u32x4n b32[2][3];
          u32x2n b16[2][3];
          split_acts48(act[0], b32[0], b16[0]);
          split_acts48(act[1], b32[1], b16[1]);
          block48_split<true>(reinterpret_cast<const unsigned*>(sQ) + (l - 1) * kNarrowQ, lane, b32, b16, o, o);
          load_peeled();                             // SP: behind the block (registers: the split operands are live in it)
        } else {
#pragma unroll
          for (int t = 0; t < HM; ++t)
#pragma unroll
            for (int mt = 0; mt < HM; ++mt) {
              const f32x4 A = ld4(W + (16 * mt + j) * LDW + 16 * t + 4 * q);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                o[0][mt] = mfma(A[r], act[0][t][r], o[0][mt]);
                o[1][mt] = mfma(A[r], act[1][t][r], o[1][mt]);
              }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        float y[2][EX];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int ee = 0; ee < EX; ++ee) {
            float sdot = 0.f;
#pragma unroll
            for (int t = 0; t < HM; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) sdot = fmaf(wr[ee][t][r], act[u][t][r], sdot);
            float kk = bx[ee];
#pragma unroll
            for (int e2 = 0; e2 < EX; ++e2) kk = fmaf(wxr[ee][e2], ax[u][e2], kk);
            y[u][ee] = fmaxf(qsum(sdot) + kk, 0.f);
          }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int mt = 0; mt < HM; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = o[u][mt][r];
#pragma unroll
              for (int ee = 0; ee < EX; ++ee) v = fmaf(wc[ee][mt][r], ax[u][ee], v);
              act[u][mt][r] = fmaxf(v, 0.f);
            }
#pragma unroll
          for (int ee = 0; ee < EX; ++ee) ax[u][ee] = y[u][ee];
        }
      }
      float sv[2], dpl[2];
      {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int t = 0; t < HM; ++t) {
          const f32x4 wl = ld4(wp + L.o_wL + 16 * t + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) { s0 = fmaf(wl[r], act[0][t][r], s0); s1 = fmaf(wl[r], act[1][t][r], s1); }
        }
        const f32x4 wlx = ld4(wp + L.o_wL + U0);
        float t0 = wp[L.o_bL], t1 = t0;
#pragma unroll
        for (int ee = 0; ee < EX; ++ee) { t0 = fmaf(wlx[ee], ax[0][ee], t0); t1 = fmaf(wlx[ee], ax[1][ee], t1); }
        sv[0] = qsum(s0) + t0;
        sv[1] = qsum(s1) + t1;
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        if (isj[u]) dx = g_z * elu_plus(sv[u]);                  // Leibniz rule: dz/dx = f(x;h)
        dpl[u] = cot[u] * (sv[u] > 0.f ? 1.f : expf(sv[u]));
        if (q == 0) p_bL += dpl[u];
      }
      // ---- backward through the last layer
      f32x4 dp[2][HM];
      float dpx[2][EX];
#pragma unroll
      for (int t = 0; t < HM; ++t) {
        const f32x4 wl = ld4(wp + L.o_wL + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            p_wL[t][r] = fmaf(dpl[u], act[u][t][r], p_wL[t][r]);
            dp[u][t][r] = act[u][t][r] > 0.f ? wl[r] * dpl[u] : 0.f;
          }
      }
      {
        const f32x4 wlx = ld4(wp + L.o_wL + U0);
#pragma unroll
        for (int ee = 0; ee < EX; ++ee)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            p_wLx[ee] = fmaf(dpl[u], ax[u][ee], p_wLx[ee]);
            dpx[u][ee] = ax[u][ee] > 0.f ? wlx[ee] * dpl[u] : 0.f;
          }
      }
      // ---- hidden->hidden layers, top down
#pragma unroll
      for (int l = NH - 1; l >= 1; --l) {
        const float* WT = sWT + (l - 1) * matf;              // (fp32 form only)
        const float* Wrow = SP ? sW + (l - 1) * kRS : sW + (l - 1) * (matf + HP) + U0 * LDW;
        const float* WTrow = SP ? sW + (l - 1) * kRS + 4 * LDW : WT + U0 * LDW;
        // dW_l main block += dpre_l^T * input_l, K = the 16 elements, node by node through the one dpre tile
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int t = 0; t < HM; ++t) *reinterpret_cast<f32x4*>(td + tile_w(j, t, q)) = dp[u][t];
          if constexpr (!RP) {
            f32x4 vx = z4;
#pragma unroll
            for (int ee = 0; ee < EX; ++ee) vx[ee] = q == 0 ? dpx[u][ee] : 0.f;
            *reinterpret_cast<f32x4*>(td + tile_w(j, HM, q)) = vx;
          }
          const float* ta = tiles + (2 * (l - 1) + u) * 16 * kTS;
#ifdef GNF_MONO_EXP_NODW           // measurement only (wrong weight gradients): what the dW contraction costs
          if (a.NK > 1000)
#endif
#pragma unroll
          for (int sK = 0; sK < 4; ++sK) {
            float fa[HTO], fb[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) { if (t < HTO) fa[t < HTO ? t : 0] = td[tile_r(sK, q, t, j)]; fb[t] = ta[tile_r(sK, q, t, j)]; }
#pragma unroll
            for (int ti = 0; ti < HTO; ++ti)
#pragma unroll
              for (int tn = 0; tn < HT; ++tn) accW[l - 1][ti][tn] = mfma(fa[ti], fb[tn], accW[l - 1][ti][tn]);
          }
        }
        f32x4 da[2][HM];
#pragma unroll
        for (int mt = 0; mt < HM; ++mt) { da[0][mt] = z4; da[1][mt] = z4; }
        if constexpr (SP) {
          u32x4n b32[2][3];
          u32x2n b16[2][3];
          split_acts48(dp[0], b32[0], b16[0]);
          split_acts48(dp[1], b32[1], b16[1]);
          block48_split<true>(reinterpret_cast<const unsigned*>(sQ) + (NH - 1 + l - 1) * kNarrowQ, lane, b32, b16, da, da);
        } else {
#pragma unroll
          for (int t = 0; t < HM; ++t)
#pragma unroll
            for (int mt = 0; mt < HM; ++mt) {
              const f32x4 A = ld4(WT + (16 * mt + j) * LDW + 16 * t + 4 * q);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                da[0][mt] = mfma(A[r], dp[0][t][r], da[0][mt]);
                da[1][mt] = mfma(A[r], dp[1][t][r], da[1][mt]);
              }
            }
        }
        // peeled parts of da: the main inputs get a rank-EX update, the peeled inputs a q-summed partial dot
        f32x4 wr[EX][HM], wcT[EX][HM], wxr[EX];
#pragma unroll
        for (int ee = 0; ee < EX; ++ee) {
#pragma unroll
          for (int t = 0; t < HM; ++t) {
            wr[ee][t] = ld4(Wrow + ee * LDW + 16 * t + 4 * q);           // W[U][k],  k = the lane's main units
            wcT[ee][t] = ld4(WTrow + ee * LDW + 16 * t + 4 * q);         // W[o][U],  o = the lane's main units
          }
          wxr[ee] = ld4(Wrow + ee * LDW + U0);                            // W[U][U']
        }
        float dax[2][EX];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          f32x4 ain[HM];                               // main inputs of layer l, back from the tile they were saved in
          const float* tg = tiles + (2 * (l - 1) + u) * 16 * kTS;
#pragma unroll
          for (int t = 0; t < HM; ++t) ain[t] = *reinterpret_cast<const f32x4*>(tg + tile_w(j, t, q));
          if constexpr (RP) {                          // the peeled rows of dW_l, db_l: dpre_l[U] x (inputs, 1)
#pragma unroll
            for (int ee = 0; ee < EX; ++ee) {
#pragma unroll
              for (int t = 0; t < HM; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) prw[l - 1][ee][t][r] = fmaf(dpx[u][ee], ain[t][r], prw[l - 1][ee][t][r]);
#pragma unroll
              for (int e2 = 0; e2 < EX; ++e2) prx[l - 1][ee][e2] = fmaf(dpx[u][ee], axin[l - 1][u][e2], prx[l - 1][ee][e2]);
              prb[l - 1][ee] += dpx[u][ee];
            }
          }
          // da of the peeled inputs U' = sum_o W[o][U'] dp[o] + sum_U W[U][U'] dpx[U]
#pragma unroll
          for (int e2 = 0; e2 < EX; ++e2) {
            float sdot = 0.f;
#pragma unroll
            for (int t = 0; t < HM; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) sdot = fmaf(wcT[e2][t][r], dp[u][t][r], sdot);
            float kk = 0.f;
#pragma unroll
            for (int ee = 0; ee < EX; ++ee) kk = fmaf(wxr[ee][e2], dpx[u][ee], kk);
            dax[u][e2] = qsum(sdot) + kk;
          }
          // main inputs: MFMA result + rank-EX update, gated by (input > 0)
#pragma unroll
          for (int t = 0; t < HM; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float v = da[u][t][r];
#pragma unroll
              for (int ee = 0; ee < EX; ++ee) v = fmaf(wr[ee][t][r], dpx[u][ee], v);
              dp[u][t][r] = ain[t][r] > 0.f ? v : 0.f;
            }
#pragma unroll
          for (int e2 = 0; e2 < EX; ++e2) dpx[u][e2] = axin[l - 1][u][e2] > 0.f ? dax[u][e2] : 0.f;
        }
      }
      // ---- first layer: rank-1 in x_k, node-independent in h
      float sx0 = 0.f, sx1 = 0.f;
#pragma unroll
      for (int t = 0; t < HM; ++t) {
        const f32x4 wx = ld4(wp + L.o_w1x + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Ds[t][r] += dp[0][t][r] + dp[1][t][r];
          p_w1x[t][r] = fmaf(dp[0][t][r], xk[0], p_w1x[t][r]);
          p_w1x[t][r] = fmaf(dp[1][t][r], xk[1], p_w1x[t][r]);
          sx0 = fmaf(wx[r], dp[0][t][r], sx0);
          sx1 = fmaf(wx[r], dp[1][t][r], sx1);
        }
      }
      {
        const f32x4 wx = ld4(wp + L.o_w1x + U0);
        float tx0 = 0.f, tx1 = 0.f;
#pragma unroll
        for (int ee = 0; ee < EX; ++ee) {
          Dsx[ee] += dpx[0][ee] + dpx[1][ee];
          p_w1xx[ee] = fmaf(dpx[0][ee], xk[0], p_w1xx[ee]);
          p_w1xx[ee] = fmaf(dpx[1][ee], xk[1], p_w1xx[ee]);
          tx0 = fmaf(wx[ee], dpx[0][ee], tx0);
          tx1 = fmaf(wx[ee], dpx[1][ee], tx1);
        }
        if (isj[0]) dx += qsum(sx0) + tx0;                       // gjac * df/dx(x;h)
        if (isj[1]) dx += qsum(sx1) + tx1;
      }
    }

    // ---- per-group epilogue: staged Dsum (for d W1h and d b_0), dh, dx.  The peeled units form the tile HM of the
    //      padded Dsum row: lanes q = 0 hold them (columns U0..U0+3), the other lanes of the tile are zero
    f32x4 DsF[HT];
#pragma unroll
    for (int t = 0; t < HM; ++t) DsF[t] = Ds[t];
    DsF[HM] = z4;
#pragma unroll
    for (int ee = 0; ee < EX; ++ee) DsF[HM][ee] = q == 0 ? Dsx[ee] : 0.f;
    float* ds = a.Dsum + (grp * 16 + j) * HP + 4 * q;
#pragma unroll
    for (int t = 0; t < HT; ++t) *reinterpret_cast<f32x4*>(ds + 16 * t) = DsF[t];
    if (valid && q == 0 && a.gx) a.gx[e] = dx;
    const int64_t gbase = b * a.g_sb + i * a.g_sd;
    for (int mt = 0; mt < L.CP / 16; ++mt) {
      f32x4 o = z4;
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const f32x4 A = ld4(a.pack + L.o_W1hT + (16 * mt + j) * LDW + 16 * t + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) o = mfma(A[r], DsF[t][r], o);
      }
      if (valid) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cc = 16 * mt + 4 * q + r;
          if (cc < L.c) a.gh[gbase + cc * a.g_sc] = o[r] + (cc == 0 ? g_z : 0.f);
        }
      }
    }
  }

  // ---- partial vector row: d wL | d w1x | (hidden biases: zero here, they ride column HP-1 of the weight rows) | d bL
  float* prow = a.part + ((int64_t)blockIdx.x * kWaves + wave) * ((NH + 2) * HP + 4);
#pragma unroll
  for (int t = 0; t < HT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int hid = 16 * t + 4 * q + r;
      float v0 = 0.f, v1 = 0.f;
      if (t < HM) { v0 = jsum(p_wL[t < HM ? t : 0][r]); v1 = jsum(p_w1x[t < HM ? t : 0][r]); }
      else if (r < EX) { v0 = jsum(p_wLx[r < EX ? r : 0]); v1 = jsum(p_w1xx[r < EX ? r : 0]); if (q != 0) { v0 = 0.f; v1 = 0.f; } }
      if (j == 0) {
        prow[hid] = v0;
        prow[HP + hid] = v1;
#pragma unroll
        for (int l = 0; l < NH; ++l) prow[(2 + l) * HP + hid] = 0.f;
      }
    }
  const float vbl = jsum(p_bL);
  if (lane == 0) { prow[(NH + 2) * HP] = vbl; prow[(NH + 2) * HP + 1] = 0.f; prow[(NH + 2) * HP + 2] = 0.f; prow[(NH + 2) * HP + 3] = 0.f; }
  // ---- accumulator rows in the padded [HP][HP] form (row = out unit, column = in unit, column HP-1 = bias gradient).
  //      wcomb: through LDS (every wavefront is done with the image and its tiles), the workgroup's four rows added in
  //      wavefront order -- 256 partial rows (8 MB at NH = 3) for the row-sum launch instead of 1024 (33 MB)
  constexpr int WSZ = (NH - 1) * HP * HP;
  if (a.wcomb) __syncthreads();
  float* wrow = a.wcomb ? smem + wave * WSZ : a.wpart + ((int64_t)blockIdx.x * kWaves + wave) * WSZ;
#pragma unroll
  for (int l = 0; l < NH - 1; ++l)
#pragma unroll
    for (int ti = 0; ti < HTO; ++ti)
#pragma unroll
      for (int tn = 0; tn < HT; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) wrow[l * HP * HP + (16 * ti + 4 * q + r) * HP + 16 * tn + j] = accW[l][ti][tn][r];
  if constexpr (RP) {
    // the fourth row tile: rows U0 + ee from the per-lane partials (summed over the 16 elements of a lane slot), the
    // padding rows zero; every location is written by exactly one lane
#pragma unroll
    for (int l = 0; l < NH - 1; ++l) {
      float* wl = wrow + l * HP * HP;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tn = 0; tn < HT; ++tn)
          if (!(q == 0 && r < EX)) wl[(U0 + 4 * q + r) * HP + 16 * tn + j] = 0.f;
#pragma unroll
      for (int ee = 0; ee < EX; ++ee) {
#pragma unroll
        for (int t = 0; t < HM; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = jsum(prw[l][ee][t][r]);
            if (j == 0) wl[(U0 + ee) * HP + 16 * t + 4 * q + r] = v;
          }
        float vt = 0.f;                                  // columns U0 .. HP-1: peeled inputs, padding, bias gradient
#pragma unroll
        for (int e2 = 0; e2 < EX; ++e2) { const float v = jsum(prx[l][ee][e2]); if (j == e2) vt = v; }
        const float vb = jsum(prb[l][ee]);
        if (j == 15) vt = vb;
        if (q == 0) wl[(U0 + ee) * HP + U0 + j] = vt;
      }
    }
  }
  if (a.wcomb) {
    __syncthreads();
    float* grow = a.wpart + (int64_t)blockIdx.x * WSZ;
    for (int i = 4 * threadIdx.x; i < WSZ; i += 4 * 64 * kWaves) {
      f32x4 v = ld4(smem + i);
#pragma unroll
      for (int w = 1; w < kWaves; ++w) v += ld4(smem + w * WSZ + i);
      *reinterpret_cast<f32x4*>(grow + i) = v;
    }
  }
}

struct UnpackArgs {
  gnf_mono_net net; MonoLayout L;
  float* gW[GNF_MONO_MAX_LAYERS]; float* gb[GNF_MONO_MAX_LAYERS];
  const float* dWpad[kMaxNH];   // [HP][HP] (out,in) for l = 1..NH-1
  const float* dW1h;            // [HP][c]
  const float* vec;             // [(NH+2)*HP + 4]
  int ones;                     // hidden-layer bias gradients sit in column HP-1 of dWpad
};

__global__ void mono_unpack_k(UnpackArgs u) {
  const MonoLayout& L = u.L;
  const gnf_mono_net& N = u.net;
  const int HP = L.HP, NH = L.NH;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  const int H0 = N.dims[1], in0 = N.dims[0];
  auto pos = [&](int unit, int l) { return mono_pos_of(unit, N.dims[l], L.perm); };     // padded position of a unit of layer l
  for (int k = tid; k < H0 * in0; k += nth) {
    const int r = k / in0, cc = k % in0;
    u.gW[0][k] = cc == 0 ? u.vec[HP + pos(r, 1)] : u.dW1h[pos(r, 1) * L.c + (cc - 1)];
  }
  for (int k = tid; k < H0; k += nth) u.gb[0][k] = u.vec[2 * HP + pos(k, 1)];
  for (int l = 1; l < NH; ++l) {
    const int out = N.dims[l + 1], in = N.dims[l];
    for (int k = tid; k < out * in; k += nth) u.gW[l][k] = u.dWpad[l][pos(k / in, l + 1) * HP + pos(k % in, l)];
    for (int k = tid; k < out; k += nth)
      u.gb[l][k] = u.ones ? u.dWpad[l][pos(k, l + 1) * HP + HP - 1] : u.vec[(2 + l) * HP + pos(k, l + 1)];
  }
  for (int k = tid; k < N.dims[NH]; k += nth) u.gW[NH][k] = u.vec[pos(k, NH)];
  if (tid == 0) u.gb[NH][0] = u.vec[(NH + 2) * HP];
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
int pick_ht(const gnf_mono_net* net) {
  if (!net || net->nl < 2 || net->nl > GNF_MONO_MAX_LAYERS) return -1;
  if (net->dims[net->nl] != 1 || net->dims[0] < 1) return -1;
  int hmax = 0;
  for (int l = 1; l < net->nl; ++l) hmax = net->dims[l] > hmax ? net->dims[l] : hmax;
  const int supported[] = {2, 4, 7, 10, 16};
  for (int ht : supported)
    if (hmax <= 16 * ht) return ht;
  return -1;
}

constexpr int kLdsBudget = 160 * 1024;

// layout of a net's pack + whether its kernels peel the leftover units of the last tile (all hidden widths H with
// H / 16 == 3 and H mod 16 in {1, 2, 3}: the reference's default [50, 50, 50]).  (The A/B switches of rounds 3-5 --
// GNF_MONO_PEEL, GNF_MONO_WIDE_EXP, GNF_MONO_ROWPEEL, GNF_MONO_KPERM, GNF_MONO_INV_EPG4, GNF_MONO_WIDE / _WIDE_FWD -- are gone with
// the instantiations only they reached: every one of them lost on every recorded box, profiles/r03-r05_mono_*.txt.)
MonoLayout net_layout(const gnf_mono_net* net, int HT) {
  MonoLayout L = make_layout(HT, net->nl - 1, net->dims[0] - 1);
  if (HT == 4) {
    const int H = net->dims[1];
    bool same = true;
    for (int l = 1; l < net->nl; ++l) same = same && net->dims[l] == H;
    if (same && H / 16 == 3 && H % 16 >= 1 && H % 16 <= 3) { L.HM = 3; L.EX = H % 16; }
  }
  if (HT >= 7) {                  // wide nets: K order of the last unit tile (see MonoLayout::perm)
    L.perm = 1;
    for (int l = 1; l < net->nl; ++l) {
      const int H = net->dims[l], T = (H - 1) / 16;
      L.ksv[l] = 4 * T + (H - 16 * T + 3) / 4;
    }
  }
  return L;
}

template <bool INV>
int launch_fwd(const MonoArgs& a, hipStream_t s) {
  const int HT = a.L.HT;
  const int64_t ngroups = (a.n + 15) / 16;
  int64_t grid = (ngroups + kWaves - 1) / kWaves;
  const size_t lds = (size_t)a.L.fwd_floats * sizeof(float);
  const size_t lds_one = (size_t)a.L.HP * a.L.LDW * sizeof(float);
  const bool wlds = lds <= (size_t)150 * 1024;                       // whole forward image resident
  const bool swap = !wlds && a.L.NH > 1 && lds_one <= (size_t)150 * 1024;   // else one matrix at a time, else L1/L2
  if (INV && !swap && ngroups <= 512 && a.S >= 7) {
    // fewer groups than a resident wave of workgroups: split the quadrature nodes over the workgroup's wavefronts
    const int pairs = (a.S + 2) / 2;
    const bool quarter = ngroups < 256;                               // groups of 4 elements x 4 node pairs (see above)
    // level-sized problems on a peeled net: two bisection steps per round (mono_inv_ks_x_k) when the 3 (S + 1) evaluations of
    // an element fit the workgroup's pair slots (S <= 31) and the weight image is LDS-resident; GNF_MONO_INV_PTS=1 keeps the
    // one-step kernel (A/B and the bit-equality test)
    static const bool one_pt = getenv("GNF_MONO_INV_PTS") && getenv("GNF_MONO_INV_PTS")[0] == '1';
    const int epg = a.n <= 2 * 256 ? 2 : 4;                           // elements per workgroup (see mono_inv_ks_x_k)
    const int ks_waves = ((3 * (a.S + 1) + 1) / 2 * epg + 15) / 16;   // node-pair columns / 16 per wavefront
    if (a.L.EX > 0 && quarter && wlds && !one_pt && ks_waves <= kSplitWavesX && a.S <= 31) {   // (32 node slots per point)
      const size_t lds_k = lds + (size_t)(2 * 32 + 2 * 3 * 32 * epg + 2 * 3 * epg) * sizeof(float);
      const unsigned gq = (unsigned)((a.n + epg - 1) / epg);
#define GNF_INVKS_W(EX_, EPG_, MW_)                                                                            \
      {                                                                                                        \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_inv_ks_x_k<3, EX_, EPG_, MW_>),          \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k);                     \
        hipLaunchKernelGGL((mono_inv_ks_x_k<3, EX_, EPG_, MW_>), dim3(gq), dim3(64 * ks_waves), lds_k, s, a);  \
      }
#define GNF_INVKS(EX_, EPG_) { if (ks_waves <= 8) GNF_INVKS_W(EX_, EPG_, 8) else GNF_INVKS_W(EX_, EPG_, kSplitWavesX) }
      if (a.L.EX <= 2) { if (epg == 2) GNF_INVKS(2, 2) else GNF_INVKS(2, 4) }
      else { if (epg == 2) GNF_INVKS(3, 2) else GNF_INVKS(3, 4) }
#undef GNF_INVKS
#undef GNF_INVKS_W
      GNF_LAUNCH_CHECK();
      return 0;
    }
    if (a.L.EX > 0 && quarter) {
      const int nwq = (pairs + 3) / 4 < kSplitWavesX ? (pairs + 3) / 4 : kSplitWavesX;
      const size_t lds_q = (wlds ? lds : 0) + 2 * nwq * 16 * sizeof(float);
      const unsigned gq = (unsigned)((a.n + 3) / 4);
#define GNF_INVXQ(EX_)                                                                                         \
      if (wlds) {                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_inv_split_x_k<3, EX_, 1, 4>),            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);                     \
        hipLaunchKernelGGL((mono_inv_split_x_k<3, EX_, 1, 4>), dim3(gq), dim3(64 * nwq), lds_q, s, a);         \
      } else {                                                                                                 \
        hipLaunchKernelGGL((mono_inv_split_x_k<3, EX_, 0, 4>), dim3(gq), dim3(64 * nwq), lds_q, s, a);         \
      }
      if (a.L.EX <= 2) { GNF_INVXQ(2) } else { GNF_INVXQ(3) }
#undef GNF_INVXQ
      GNF_LAUNCH_CHECK();
      return 0;
    }
    if (a.L.EX > 0) {                                                 // peeled narrow net: up to 16 wavefronts per group
      const int nwx = pairs < kSplitWavesX ? pairs : kSplitWavesX;
      const size_t lds_x = (wlds ? lds : 0) + 2 * nwx * 16 * sizeof(float);
#define GNF_INVX(EX_)                                                                                          \
      if (wlds) {                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_inv_split_x_k<3, EX_, 1>),               \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_x);                     \
        hipLaunchKernelGGL((mono_inv_split_x_k<3, EX_, 1>), dim3((unsigned)ngroups), dim3(64 * nwx), lds_x, s, a); \
      } else {                                                                                                 \
        hipLaunchKernelGGL((mono_inv_split_x_k<3, EX_, 0>), dim3((unsigned)ngroups), dim3(64 * nwx), lds_x, s, a); \
      }
      if (a.L.EX <= 2) { GNF_INVX(2) } else { GNF_INVX(3) }
#undef GNF_INVX
      GNF_LAUNCH_CHECK();
      return 0;
    }
    if (quarter) {
      const int nwq = (pairs + 3) / 4 < kSplitWaves ? (pairs + 3) / 4 : kSplitWaves;
      const size_t lds_q = (wlds ? lds : 0) + 2 * nwq * 16 * sizeof(float);
      const unsigned gq = (unsigned)((a.n + 3) / 4);
#define GNF_INVQ_CASE(HT_)                                                                                     \
  case HT_:                                                                                                   \
    if (wlds) {                                                                                               \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_inv_split_k<HT_, 1, 4>),                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_q);                      \
      hipLaunchKernelGGL((mono_inv_split_k<HT_, 1, 4>), dim3(gq), dim3(64 * nwq), lds_q, s, a);               \
    } else {                                                                                                  \
      hipLaunchKernelGGL((mono_inv_split_k<HT_, 0, 4>), dim3(gq), dim3(64 * nwq), lds_q, s, a);               \
    }                                                                                                         \
    break;
      switch (HT) {
        GNF_INVQ_CASE(2) GNF_INVQ_CASE(4) GNF_INVQ_CASE(7) GNF_INVQ_CASE(10) GNF_INVQ_CASE(16)
        default: return GNF_ESHAPE;
      }
#undef GNF_INVQ_CASE
      GNF_LAUNCH_CHECK();
      return 0;
    }
    const int nw = pairs < kSplitWaves ? pairs : kSplitWaves;
    const size_t lds_split = (wlds ? lds : 0) + 2 * nw * 16 * sizeof(float);
#define GNF_INV_CASE(HT_)                                                                                      \
  case HT_:                                                                                                   \
    if (wlds) {                                                                                               \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_inv_split_k<HT_, 1>),                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_split);                  \
      hipLaunchKernelGGL((mono_inv_split_k<HT_, 1>), dim3((unsigned)ngroups), dim3(64 * nw), lds_split, s, a); \
    } else {                                                                                                  \
      hipLaunchKernelGGL((mono_inv_split_k<HT_, 0>), dim3((unsigned)ngroups), dim3(64 * nw), lds_split, s, a); \
    }                                                                                                         \
    break;
    switch (HT) {
      GNF_INV_CASE(2) GNF_INV_CASE(4) GNF_INV_CASE(7) GNF_INV_CASE(10) GNF_INV_CASE(16)
      default: return GNF_ESHAPE;
    }
#undef GNF_INV_CASE
    GNF_LAUNCH_CHECK();
    return 0;
  }
  const int64_t per_cu = ((wlds && lds > (size_t)kLdsBudget / 2) || swap) ? 1 : 2;   // resident workgroups per CU
  if (grid > 256 * per_cu) grid = 256 * per_cu;                      // persistent
  if (a.L.EX > 0 && !swap) {                                          // peeled narrow net (HM = 3: H in {49, 50, 51})
    if constexpr (!INV) {
      const size_t lds_sp = lds + (size_t)(a.L.NH - 1) * kNarrowQ * 4;
      if (wlds && a.L.NH > 1 && !a.f32only && gnf_gemm_split_enabled() && lds_sp <= (size_t)kLdsBudget / 2) {
        g_fwd_kernel = "mono_fwd_x_k<split>";
#define GNF_FWDXS(EX_)                                                                                         \
        {                                                                                                      \
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_fwd_x_k<3, EX_, 1, false, true>),      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sp);                  \
          hipLaunchKernelGGL((mono_fwd_x_k<3, EX_, 1, false, true>), dim3((unsigned)grid), dim3(64 * kWaves), lds_sp, s, a); \
        }
        if (a.L.EX <= 2) GNF_FWDXS(2) else GNF_FWDXS(3)
#undef GNF_FWDXS
        GNF_LAUNCH_CHECK();
        return 0;
      }
    }
#define GNF_FWDX(EX_)                                                                                          \
    if (wlds) {                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_fwd_x_k<3, EX_, 1, INV>),                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
      hipLaunchKernelGGL((mono_fwd_x_k<3, EX_, 1, INV>), dim3((unsigned)grid), dim3(64 * kWaves), lds, s, a);  \
    } else {                                                                                                   \
      hipLaunchKernelGGL((mono_fwd_x_k<3, EX_, 0, INV>), dim3((unsigned)grid), dim3(64 * kWaves), 0, s, a);    \
    }
    if (a.L.EX <= 2) { GNF_FWDX(2) } else { GNF_FWDX(3) }
#undef GNF_FWDX
    GNF_LAUNCH_CHECK();
    return 0;
  }
#define GNF_FWD_CASE(HT_)                                                                                     \
  case HT_:                                                                                                  \
    if (wlds) {                                                                                              \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_fwd_k<HT_, 1, INV>),                     \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                             \
      hipLaunchKernelGGL((mono_fwd_k<HT_, 1, INV>), dim3((unsigned)grid), dim3(64 * kWaves), lds, s, a);     \
    } else if (swap) {                                                                                       \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_fwd_k<HT_, 2, INV>),                     \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_one);                         \
      hipLaunchKernelGGL((mono_fwd_k<HT_, 2, INV>), dim3((unsigned)grid), dim3(64 * kWaves), lds_one, s, a); \
    } else {                                                                                                 \
      hipLaunchKernelGGL((mono_fwd_k<HT_, 0, INV>), dim3((unsigned)grid), dim3(64 * kWaves), 0, s, a);       \
    }                                                                                                        \
    break;
  switch (HT) {
    GNF_FWD_CASE(2) GNF_FWD_CASE(4) GNF_FWD_CASE(7) GNF_FWD_CASE(10) GNF_FWD_CASE(16)
    default: return GNF_ESHAPE;
  }
#undef GNF_FWD_CASE
  GNF_LAUNCH_CHECK();
  return 0;
}

template <int HT, int NH>
int launch_bwd_one(const MonoArgs& a, unsigned grid, hipStream_t s) {
  const size_t lds_all = (size_t)a.L.total_floats * sizeof(float);
  const size_t lds_one = (size_t)a.L.HP * a.L.LDW * sizeof(float);
  if constexpr (HT <= 4 && NH > 1) {
    if (a.indw == 2) {                          // two nodes per pass, image without W1h / W1h^T, 5 tiles per wavefront
      const size_t lds_pair = ((size_t)a.L.o_W1h + (size_t)(NH - 1) * (2 * a.L.HP * a.L.LDW + a.L.HP) +
                               (size_t)kWaves * (2 * (NH - 1) + 1) * 16 * kTS) * sizeof(float);
      if constexpr (HT == 4) {
        if (a.L.EX > 0) {                       // peeled: 3 tiles on the MFMA, the H mod 16 leftover units on the VALU
          // row peel (the weight gradient's fourth out tile on the VALU) where its partials fit the register file
          constexpr bool kRP = NH <= 3;          // NH = 4: 122 registers spilled
          if constexpr (NH <= 3) {               // the split-bf16 chain (SP): the reference's [H]^3 / [H]^2 nets
            const size_t lds_sp = ((size_t)a.L.o_W1h + (size_t)(NH - 1) * (8 * a.L.LDW + a.L.HP + 2 * kNarrowQ) +
                                   (size_t)kWaves * (2 * (NH - 1) + 1) * 16 * kTS) * sizeof(float);
            const size_t lds_wcomb = (size_t)kWaves * (NH - 1) * a.L.HP * a.L.HP * sizeof(float);
            if (!a.f32only && gnf_gemm_split_enabled() && lds_sp <= (size_t)kLdsBudget && (!a.wcomb || lds_wcomb <= lds_sp)) {
              g_bwd_kernel = "mono_bwd_pair_x_k<split>";
              if (a.L.EX <= 2) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_bwd_pair_x_k<3, NH, 2, kRP, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sp);
                hipLaunchKernelGGL((mono_bwd_pair_x_k<3, NH, 2, kRP, true>), dim3(grid), dim3(64 * kWaves), lds_sp, s, a);
              } else {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_bwd_pair_x_k<3, NH, 3, false, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sp);
                hipLaunchKernelGGL((mono_bwd_pair_x_k<3, NH, 3, false, true>), dim3(grid), dim3(64 * kWaves), lds_sp, s, a);
              }
              GNF_LAUNCH_CHECK();
              return 0;
            }
          }
          if (a.L.EX <= 2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_bwd_pair_x_k<3, NH, 2, kRP>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair);
            hipLaunchKernelGGL((mono_bwd_pair_x_k<3, NH, 2, kRP>), dim3(grid), dim3(64 * kWaves), lds_pair, s, a);
          } else {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_bwd_pair_x_k<3, NH, 3>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair);
            hipLaunchKernelGGL((mono_bwd_pair_x_k<3, NH, 3>), dim3(grid), dim3(64 * kWaves), lds_pair, s, a);
          }
          GNF_LAUNCH_CHECK();
          return 0;
        }
      }
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_bwd_pair_k<HT, NH>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair);
      hipLaunchKernelGGL((mono_bwd_pair_k<HT, NH>), dim3(grid), dim3(64 * kWaves), lds_pair, s, a);
      GNF_LAUNCH_CHECK();
      return 0;
    }
    if (a.indw) {                               // whole image + per-wavefront element-major tiles, one workgroup per CU
      const size_t lds_in = ((size_t)(a.L.total_floats + 3) / 4 * 4 + (size_t)kWaves * NH * 16 * kTS) * sizeof(float);
      if (lds_in > (size_t)kLdsBudget) return GNF_ESHAPE;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_bwd_k<HT, NH, 1, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_in);
      hipLaunchKernelGGL((mono_bwd_k<HT, NH, 1, true, true>), dim3(grid), dim3(64 * kWaves), lds_in, s, a);
      GNF_LAUNCH_CHECK();
      return 0;
    }
    if (a.ones) {                               // everything but the transposed hidden matrices resident, two workgroups per CU
      const size_t lds_res = (size_t)(a.L.fwd_floats + a.L.CP * a.L.LDW) * sizeof(float);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_bwd_k<HT, NH, 4, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_res);
      hipLaunchKernelGGL((mono_bwd_k<HT, NH, 4, true>), dim3(grid), dim3(64 * kWaves), lds_res, s, a);
      GNF_LAUNCH_CHECK();
      return 0;
    }
  }
  if constexpr (HT <= 4) {                      // whole image resident
    if (lds_all > (size_t)150 * 1024) return GNF_ESHAPE;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_bwd_k<HT, NH, 1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_all);
    hipLaunchKernelGGL((mono_bwd_k<HT, NH, 1>), dim3(grid), dim3(64 * kWaves), lds_all, s, a);
  } else if constexpr (HT <= 10) {
    const size_t lds_res = lds_one * (NH > 1 ? NH - 1 : 1);
    if (NH > 1 && lds_res <= (size_t)150 * 1024) {          // all hidden->hidden matrices resident (H <= 112, 3 hidden layers)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_bwd_k<HT, NH, 3>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_res);
      hipLaunchKernelGGL((mono_bwd_k<HT, NH, 3>), dim3(grid), dim3(64 * kWaves), lds_res, s, a);
    } else {                                                 // one matrix at a time
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_bwd_k<HT, NH, 2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_one);
      hipLaunchKernelGGL((mono_bwd_k<HT, NH, 2>), dim3(grid), dim3(64 * kWaves), lds_one, s, a);
    }
  } else {
    hipLaunchKernelGGL((mono_bwd_k<HT, NH, 0>), dim3(grid), dim3(64 * kWaves), 0, s, a);
  }
  GNF_LAUNCH_CHECK();
  return 0;
}

template <int HT>
int launch_bwd_nh(const MonoArgs& a, unsigned grid, hipStream_t s) {
  switch (a.L.NH) {
    case 1: return launch_bwd_one<HT, 1>(a, grid, s);
    case 2: return launch_bwd_one<HT, 2>(a, grid, s);
    case 3: return launch_bwd_one<HT, 3>(a, grid, s);
    case 4: return launch_bwd_one<HT, 4>(a, grid, s);
    default: return GNF_ESHAPE;
  }
}

int launch_bwd(const MonoArgs& a, unsigned grid, hipStream_t s) {
  switch (a.L.HT) {
    case 2: return launch_bwd_nh<2>(a, grid, s);
    case 4: return launch_bwd_nh<4>(a, grid, s);
    case 7: return launch_bwd_nh<7>(a, grid, s);
    case 10: return launch_bwd_nh<10>(a, grid, s);
    case 16: return launch_bwd_nh<16>(a, grid, s);
    default: return GNF_ESHAPE;
  }
}

// ---------------------------------------------------------------------------------------
// Weight gradient of one hidden->hidden layer of a WIDE net from the staged rows:
//   dW[i][k] (+)= sum_rows dpre[row][i] * act[row][k],   db[i] (+)= sum_rows dpre[row][i]
// A tall-skinny A^T B: the output is only HP x HP (112^2 / 160^2), the contraction runs over 10^6..10^8 rows.  The
// generic GEMM tiles the output 64x64 and so streams both staged arrays once per tile row/column (3x at HP = 160);
// here a workgroup owns the WHOLE output for its slice of rows, so every staged byte is read exactly once
// (HBM-bound by construction), and the bias gradient rides along as one extra MFMA column against a constant 1.
// Wavefront w owns output row tiles {w, w+4, w+8}; row slab of 32 staged rows in LDS, next slab prefetched in registers.
// ---------------------------------------------------------------------------------------
constexpr int kDwRows = 16;
template <int HT>
__global__ __launch_bounds__(256, 2) void mono_dw_k(const float* __restrict__ Y, const float* __restrict__ X,
                                                 float* __restrict__ Cpart, float* __restrict__ bpart, int64_t rows,
                                                 int64_t rows_per_wg, int accum) {
  constexpr int HP = 16 * HT, LD = HP + 4, MT = (HT + 3) / 4;
  constexpr int NV4 = kDwRows * HP / 4, V4 = (NV4 + 255) / 256;          // float4 per slab and array / per thread
  __shared__ __attribute__((aligned(16))) float Ys[kDwRows * LD];
  __shared__ __attribute__((aligned(16))) float Xs[kDwRows * LD];
  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;

  f32x4 acc[MT][HT + 1];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n <= HT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 py[V4], px[V4];
  auto fetch = [&](int64_t rb) {
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const int idx = tid + v * 256, rr = idx / (HP / 4), c4 = idx - rr * (HP / 4);
      const int64_t row = rb + rr;
      if (idx < NV4 && row < r1) { py[v] = ld4(Y + row * HP + 4 * c4); px[v] = ld4(X + row * HP + 4 * c4); }
      else { py[v] = f32x4{0.f, 0.f, 0.f, 0.f}; px[v] = py[v]; }
    }
  };
  if (r0 < r1) fetch(r0);
  for (int64_t rb = r0; rb < r1; rb += kDwRows) {
    __syncthreads();                                  // previous slab consumed
#pragma unroll
    for (int v = 0; v < V4; ++v) {
      const int idx = tid + v * 256, rr = idx / (HP / 4), c4 = idx - rr * (HP / 4);
      if (idx < NV4) {
        *reinterpret_cast<f32x4*>(Ys + rr * LD + 4 * c4) = py[v];
        *reinterpret_cast<f32x4*>(Xs + rr * LD + 4 * c4) = px[v];
      }
    }
    __syncthreads();
    if (rb + kDwRows < r1) fetch(rb + kDwRows);       // in flight under the MFMAs
#pragma unroll 2
    for (int ks = 0; ks < kDwRows / 4; ++ks) {
      float a[MT], b[HT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int mt = wave + 4 * m;
        a[m] = mt < HT ? Ys[(4 * ks + q) * LD + 16 * mt + j] : 0.f;
      }
#pragma unroll
      for (int n = 0; n < HT; ++n) b[n] = Xs[(4 * ks + q) * LD + 16 * n + j];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (wave + 4 * m < HT) {
#pragma unroll
          for (int n = 0; n < HT; ++n) acc[m][n] = mfma(a[m], b[n], acc[m][n]);
          acc[m][HT] = mfma(a[m], 1.0f, acc[m][HT]);            // column sums of dpre: the bias gradient
        }
      }
    }
  }
  float* C = Cpart + (int64_t)blockIdx.x * HP * HP;
  float* bp = bpart + (int64_t)blockIdx.x * HP;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mt = wave + 4 * m;
    if (mt < HT) {
#pragma unroll
      for (int n = 0; n < HT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* cp = C + (16 * mt + 4 * q + r) * HP + 16 * n + j;
          *cp = accum ? *cp + acc[m][n][r] : acc[m][n][r];
        }
      if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* cp = bp + 16 * mt + 4 * q + r;
          *cp = accum ? *cp + acc[m][HT][r] : acc[m][HT][r];
        }
      }
    }
  }
}

constexpr int kDwGrid = 512;                  // workgroups (= partials) of mono_dw_k

constexpr unsigned kBwdGrid = 256 * 2;        // persistent workgroups of the chain kernel
constexpr int kSplits = 256;                  // split-K partials per weight-gradient GEMM
constexpr int64_t kWsTarget = 6ll << 30;      // staging budget the ws_bytes query asks for

struct BwdPlan {
  int64_t chunk_elems;      // elements per chain-kernel launch (multiple of 16)
  int64_t o_SA[kMaxNH], o_SD[kMaxNH], o_Dsum, o_part, o_gpart[kMaxNH], o_bpart[kMaxNH], o_hpart, o_dW[kMaxNH], o_dW1h,
      o_vec, o_rs, o_wpart;
  int64_t total_floats;
};

// narrow nets whose hidden widths leave a padding column (and whose image + tiles fit the LDS): the chain kernel
// accumulates the hidden-layer weight gradients itself, nothing but Dsum is staged
bool use_indw(const gnf_mono_net* net, const MonoLayout& L) {
  if (L.HT > 4 || L.NH < 2) return false;
  for (int l = 1; l < L.NH; ++l)
    if (net->dims[l] >= L.HP) return false;
  const size_t lds = ((size_t)(L.total_floats + 3) / 4 * 4 + (size_t)kWaves * L.NH * 16 * kTS) * sizeof(float);
  static const bool off = getenv("GNF_MONO_INDW") && getenv("GNF_MONO_INDW")[0] == '0';     // A/B switch (measurement)
  return lds <= (size_t)160 * 1024 && !off;
}
constexpr unsigned kInDwGrid = 256;           // one workgroup per CU (512 registers per wavefront)

// indw: 0 staged weight gradients; 1 narrow in-kernel (one accumulator row per wavefront); 2 wide in-kernel
// (gnf_monotonic_wide.hip: one accumulator row per workgroup)
BwdPlan plan_bwd(const MonoLayout& L, int S, int64_t n, int64_t ws_floats, int indw = 0) {
  BwdPlan P;
  const int64_t HP = L.HP, NK = (S + 2 + 1) / 2 * 2;
  const int64_t vecw = (L.NH + 2) * HP + 4;
  int64_t fixed = 0;
  P.o_part = fixed; fixed += (int64_t)kBwdGrid * kWaves * vecw;
  const int64_t npart = (L.HT == 7 || L.HT == 10) ? kDwGrid : kSplits;
  for (int l = 1; l < L.NH; ++l) { P.o_gpart[l] = fixed; fixed += npart * HP * HP; }
  for (int l = 1; l < L.NH; ++l) { P.o_bpart[l] = fixed; fixed += (int64_t)kDwGrid * HP; }
  P.o_hpart = fixed; fixed += (int64_t)kSplits * HP * L.c;
  for (int l = 1; l < L.NH; ++l) { P.o_dW[l] = fixed; fixed += HP * HP; }
  P.o_dW1h = fixed; fixed += HP * L.c;
  P.o_vec = fixed; fixed += vecw;
  P.o_rs = fixed; fixed += (int64_t)kRowsumChunks * HP;     // scratch of the tall row-sums (bias gradients)
  P.o_wpart = fixed;
  if (indw) fixed += (int64_t)kInDwGrid * (indw == 2 ? 1 : kWaves) * (L.NH - 1) * HP * HP;
  const int64_t per_elem = indw ? HP : (int64_t)(L.NH - 1) * 2 * NK * HP + HP;     // SA+SD per hidden layer, Dsum
  int64_t ce = (n + 15) / 16 * 16;
  if (ws_floats > 0) {
    const int64_t room = ws_floats - fixed;
    int64_t fit = room > 0 ? room / per_elem / 16 * 16 : 0;
    if (fit < ce) ce = fit;
  }
  P.chunk_elems = ce;
  int64_t o = fixed;
  for (int l = 1; l < L.NH; ++l) {
    P.o_SA[l] = o; if (!indw) o += ce * NK * HP;
    P.o_SD[l] = o; if (!indw) o += ce * NK * HP;
  }
  P.o_Dsum = o; o += ce * HP;
  P.total_floats = o;
  return P;
}

}  // namespace

extern "C" {

int64_t gnf_monotonic_pack_floats(const gnf_mono_net* net) {
  const int HT = pick_ht(net);
  if (HT < 0) return GNF_ESHAPE;
  return net_layout(net, HT).pack_floats;
}

int gnf_monotonic_pack(const gnf_mono_net* net, float* pack, gnf_stream_t stream) {
  const int HT = pick_ht(net);
  if (HT < 0) return GNF_ESHAPE;
  if (!pack) return GNF_EINVAL;
  PackArgs a;
  a.net = *net;
  a.L = net_layout(net, HT);
  hipLaunchKernelGGL(mono_pack_k, dim3((a.L.pack_floats + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, pack);
  GNF_LAUNCH_CHECK();
  return 0;
}

const char* gnf_monotonic_fwd_kernel(void) { return g_fwd_kernel; }

static int mono_fwd_any(const float* pack, const gnf_mono_net* net, const float* x, const float* h, int64_t h_sb,
                        int64_t h_sd, int64_t h_sc, const float* cc_w, const float* cc_t, int S, float* z,
                        float* jac, int64_t B, int64_t d, gnf_stream_t stream, bool true_f32);

int gnf_monotonic_fwd(const float* pack, const gnf_mono_net* net, const float* x, const float* h, int64_t h_sb,
                      int64_t h_sd, int64_t h_sc, const float* cc_w, const float* cc_t, int S, float* z,
                      float* jac, int64_t B, int64_t d, gnf_stream_t stream) {
  return mono_fwd_any(pack, net, x, h, h_sb, h_sd, h_sc, cc_w, cc_t, S, z, jac, B, d, stream, false);
}
int gnf_monotonic_fwd_f32(const float* pack, const gnf_mono_net* net, const float* x, const float* h, int64_t h_sb,
                          int64_t h_sd, int64_t h_sc, const float* cc_w, const float* cc_t, int S, float* z,
                          float* jac, int64_t B, int64_t d, gnf_stream_t stream) {
  return mono_fwd_any(pack, net, x, h, h_sb, h_sd, h_sc, cc_w, cc_t, S, z, jac, B, d, stream, true);
}

static int mono_fwd_any(const float* pack, const gnf_mono_net* net, const float* x, const float* h, int64_t h_sb,
                        int64_t h_sd, int64_t h_sc, const float* cc_w, const float* cc_t, int S, float* z,
                        float* jac, int64_t B, int64_t d, gnf_stream_t stream, bool true_f32) {
  const int HT = pick_ht(net);
  if (HT < 0) return GNF_ESHAPE;
  if (!pack || !cc_w || !cc_t || S < 1 || B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;                    // batch-sized arrays may be NULL for an empty batch
  if (!x || !h || !z || !jac) return GNF_EINVAL;
  MonoArgs a{};
  a.pack = pack; a.L = net_layout(net, HT);
  a.x = x; a.h = h; a.h_sb = h_sb; a.h_sd = h_sd; a.h_sc = h_sc;
  a.ccw = cc_w; a.cct = cc_t; a.S = S; a.z = z; a.jac = jac; a.n = B * d; a.d = d;
  a.f32only = true_f32;
  if (gnf_mono_fwd_wide_ok(a.L)) return gnf_mono_fwd_wide_launch(a, (hipStream_t)stream, true_f32, &g_fwd_kernel);
  g_fwd_kernel = "mono_fwd_k";                        // (launch_fwd overrides it for the split form of the peeled nets)
  return launch_fwd<false>(a, (hipStream_t)stream);
}

int gnf_monotonic_inv_scatter(const float* pack, const gnf_mono_net* net, const float* z, const float* h, int64_t h_sb,
                              int64_t h_sd, int64_t h_sc, const float* cc_w, const float* cc_t, int S, float* x,
                              const int32_t* x_row_off, int64_t x_sd, int64_t B, int64_t d, gnf_stream_t stream) {
  const int HT = pick_ht(net);
  if (HT < 0) return GNF_ESHAPE;
  if (!pack || !cc_w || !cc_t || S < 1 || B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;
  if (!z || !h || !x) return GNF_EINVAL;
  MonoArgs a{};
  a.pack = pack; a.L = net_layout(net, HT);
  a.h = h; a.h_sb = h_sb; a.h_sd = h_sd; a.h_sc = h_sc;
  a.ccw = cc_w; a.cct = cc_t; a.S = S; a.zt = z; a.xo = x; a.n = B * d; a.d = d;
  a.xo_row = x_row_off; a.xo_sd = x_sd;
  return launch_fwd<true>(a, (hipStream_t)stream);
}

int gnf_monotonic_inv(const float* pack, const gnf_mono_net* net, const float* z, const float* h, int64_t h_sb,
                      int64_t h_sd, int64_t h_sc, const float* cc_w, const float* cc_t, int S, float* x, int64_t B,
                      int64_t d, gnf_stream_t stream) {
  return gnf_monotonic_inv_scatter(pack, net, z, h, h_sb, h_sd, h_sc, cc_w, cc_t, S, x, nullptr, 0, B, d, stream);
}

int64_t gnf_monotonic_bwd_ws_bytes(const gnf_mono_net* net, int S, int64_t B, int64_t d) {
  const int HT = pick_ht(net);
  if (HT < 0) return GNF_ESHAPE;
  const MonoLayout L = net_layout(net, HT);
  const int indw = use_indw(net, L) ? 1 : (gnf_mono_bwd_wide_ok(L) ? 2 : 0);
  if (B < 1) B = 1;                          // an empty batch needs no workspace; keep the plan arithmetic away from 0
  const BwdPlan full = plan_bwd(L, S, B * d, 0, indw);
  const int64_t want = full.total_floats * (int64_t)sizeof(float);
  if (want <= kWsTarget) return want;
  // bounded staging: at least one 16-element group per persistent wavefront
  const BwdPlan fixed_only = plan_bwd(L, S, 16, 0, indw);
  const int64_t min_bytes = fixed_only.total_floats * (int64_t)sizeof(float);
  return kWsTarget > min_bytes ? kWsTarget : min_bytes;
}

static thread_local bool g_bwd_true_f32 = false;
const char* gnf_monotonic_bwd_kernel(void) { return g_bwd_kernel; }

int gnf_monotonic_bwd_f32(const float* pack, const gnf_mono_net* net, const float* x, const float* h, int64_t h_sb,
                          int64_t h_sd, int64_t h_sc, const float* cc_w, const float* cc_t, int S, const float* gz,
                          const float* gjac, float* gx, float* gh, int64_t g_sb, int64_t g_sd, int64_t g_sc,
                          float* const* gW, float* const* gb, void* ws, int64_t ws_bytes, int64_t B, int64_t d,
                          gnf_stream_t stream) {
  g_bwd_true_f32 = true;
  const int rc = gnf_monotonic_bwd(pack, net, x, h, h_sb, h_sd, h_sc, cc_w, cc_t, S, gz, gjac, gx, gh, g_sb, g_sd, g_sc, gW, gb, ws,
                                   ws_bytes, B, d, stream);
  g_bwd_true_f32 = false;
  return rc;
}

int gnf_monotonic_bwd(const float* pack, const gnf_mono_net* net, const float* x, const float* h, int64_t h_sb,
                      int64_t h_sd, int64_t h_sc, const float* cc_w, const float* cc_t, int S, const float* gz,
                      const float* gjac, float* gx, float* gh, int64_t g_sb, int64_t g_sd, int64_t g_sc,
                      float* const* gW, float* const* gb, void* ws, int64_t ws_bytes, int64_t B, int64_t d,
                      gnf_stream_t stream) {
  const int HT = pick_ht(net);
  if (HT < 0) return GNF_ESHAPE;
  if (!pack || !cc_w || !cc_t || !gW || !gb || S < 1 || B < 0 || d <= 0) return GNF_EINVAL;
  const int NH = net->nl - 1;
  if (NH > 4) return GNF_ESHAPE;
  if (B == 0) {                              // empty batch: zero parameter gradients, nothing else to write
    for (int l = 0; l <= NH; ++l) {
      if (!gW[l] || !gb[l]) return GNF_EINVAL;
      if (hipMemsetAsync(gW[l], 0, sizeof(float) * net->dims[l] * net->dims[l + 1], (hipStream_t)stream) != hipSuccess ||
          hipMemsetAsync(gb[l], 0, sizeof(float) * net->dims[l + 1], (hipStream_t)stream) != hipSuccess)
        return GNF_EINVAL;
    }
    return 0;
  }
  if (!x || !h || !gz || !gh || !ws) return GNF_EINVAL;
  if (h_sb != d * h_sd) return GNF_ESHAPE;   // element stride must collapse (caller makes h contiguous)
  hipStream_t s = (hipStream_t)stream;
  const MonoLayout L = net_layout(net, HT);
  const int64_t n = B * d;
  const bool wide = gnf_mono_bwd_wide_ok(L);       // wide nets: hidden state in LDS, weight gradients in registers
  const bool indw = use_indw(net, L) || wide;
  const BwdPlan P = plan_bwd(L, S, n, ws_bytes / (int64_t)sizeof(float), wide ? 2 : (indw ? 1 : 0));
  if (P.chunk_elems < 16) return GNF_EWS;
  float* w = (float*)ws;
  const int64_t HP = L.HP, NK = (S + 2 + 1) / 2 * 2;
  const int64_t vecw = (NH + 2) * HP + 4;

  MonoArgs a{};
  a.pack = pack; a.L = L; a.x = x; a.h = h; a.h_sb = h_sb; a.h_sd = h_sd; a.h_sc = h_sc;
  a.ccw = cc_w; a.cct = cc_t; a.S = S; a.n = n; a.d = d;
  a.gz = gz; a.gjac = gjac; a.gx = gx; a.gh = gh; a.g_sb = g_sb; a.g_sd = g_sd; a.g_sc = g_sc;
  for (int l = 1; l < NH; ++l) { a.SA[l] = w + P.o_SA[l]; a.SD[l] = w + P.o_SD[l]; }
  a.Dsum = w + P.o_Dsum; a.part = w + P.o_part; a.NK = (int)NK;
  a.ones = HT <= 4 && NH > 1 && !wide;       // (the two-role kernels carry the bias gradients in their partial vector rows)
  for (int l = 1; l < NH; ++l) a.ones = a.ones && net->dims[l] < HP;
  a.indw = wide ? 3 : (int)indw;
  a.f32only = g_bwd_true_f32;
  if (indw && !wide) {                            // the two-node kernel when its LDS plan fits (A/B: GNF_MONO_INDW=1 keeps one node)
    const size_t lds_pair = ((size_t)L.o_W1h + (size_t)(NH - 1) * (2 * L.HP * L.LDW + L.HP) +
                             (size_t)kWaves * (2 * (NH - 1) + 1) * 16 * kTS) * sizeof(float);
    static const bool one = getenv("GNF_MONO_INDW") && getenv("GNF_MONO_INDW")[0] == '1';
    if (lds_pair <= (size_t)160 * 1024 && !one) a.indw = 2;
    // the peeled two-node kernel can add its wavefronts' weight-gradient rows in LDS when they fit its allocation
    a.wcomb = a.indw == 2 && HT == 4 && L.EX > 0 && (size_t)kWaves * (NH - 1) * HP * HP * sizeof(float) <= lds_pair;
  }
  a.wpart = w + P.o_wpart;
  const unsigned bwd_grid = wide ? gnf_mono_bwd_wide_grid(L, n < P.chunk_elems ? n : P.chunk_elems)
                                 : (indw ? kInDwGrid : kBwdGrid);

  auto rowsum = [&](const float* src, float* out, int64_t Pn, int64_t N, int acc) -> int {
    return gnf_rowsum_launch(src, out, Pn, N, acc, s);
  };
  const int64_t nchunks = (n + P.chunk_elems - 1) / P.chunk_elems;
  const int64_t part_rows = (int64_t)bwd_grid * kWaves;
  const int64_t wpart_rows = (wide || a.wcomb) ? (int64_t)bwd_grid : part_rows;
  int64_t nsp_w = 1, nsp_h = 1;
  int rc = 0;
  for (int64_t ck = 0; ck < nchunks; ++ck) {
    a.e0 = ck * P.chunk_elems;
    a.ecount = n - a.e0 < P.chunk_elems ? n - a.e0 : P.chunk_elems;
    if (wide) {                        // its wavefronts write only their own half of the partial rows
      if (hipMemsetAsync(a.part, 0, sizeof(float) * part_rows * vecw, s) != hipSuccess) return GNF_EINVAL;
      if ((rc = gnf_mono_bwd_wide_launch(a, bwd_grid, s, g_bwd_true_f32, &g_bwd_kernel))) return rc;
    } else {
      g_bwd_kernel = "mono_bwd_k";
      if ((rc = launch_bwd(a, bwd_grid, s))) return rc;
    }
    const int64_t groups = (a.ecount + 15) / 16;
    const int64_t rows = groups * NK * 16;
    const int accum = ck > 0 ? GNF_GEMM_ACCUM : 0;     // chunk 0 is the largest: it defines the split count
    // d W_l (+)= dpre_l^T * act_{l-1}   (split-K partials, accumulated across chunks)
    // (indw: the chain kernel's accumulator rows -> dW, accumulated across chunks, ride in the launch that sums the partial
    // vector rows below)
    for (int l = 1; l < NH && !indw; ++l) {
      if (HT == 7 || HT == 10) {       // wide nets: one pass over the staged rows, bias gradient fused
        const int64_t rpw = ((rows + kDwGrid - 1) / kDwGrid + kDwRows - 1) / kDwRows * kDwRows;
        nsp_w = kDwGrid;
        float* cp = w + P.o_gpart[l];
        float* bp = w + P.o_bpart[l];
        const int acc = ck > 0;
        if (HT == 7) hipLaunchKernelGGL((mono_dw_k<7>), dim3(kDwGrid), dim3(256), 0, s, a.SD[l], a.SA[l], cp, bp, rows, rpw, acc);
        else hipLaunchKernelGGL((mono_dw_k<10>), dim3(kDwGrid), dim3(256), 0, s, a.SD[l], a.SA[l], cp, bp, rows, rpw, acc);
        GNF_LAUNCH_CHECK();
        continue;
      }
      GemmArgs g{};
      g.A = a.SD[l]; g.sam = 1; g.sak = HP;
      g.B = a.SA[l]; g.sbk = HP; g.sbn = 1;
      g.C = w + P.o_gpart[l]; g.scm = HP; g.scn = 1;
      g.M = HP; g.N = HP; g.K = rows; g.c_split_stride = HP * HP; g.flags = accum;
      if (ck == 0) nsp_w = gnf_gemm_num_splits(rows, kSplits);
      if ((rc = gnf_gemm_launch(g, ck == 0 ? kSplits : (int)nsp_w, s))) return rc;
    }
    // d W1[:,1:] = Dsum^T h and d b1 = colsum Dsum: tall weight-gradient launches (64 hidden units each) + their
    // reductions on contiguous conditioner outputs instead of a split-K tiled GEMM, its reduction and a two-stage column
    // sum (cfg4: 17 + 5 + 13 us; cfg5, where Dsum is 2 GB: 3.6 + 0.8 ms at 0.7 TB/s)
    const bool tall_w1 = nchunks == 1 && h_sc == 1 && gnf_linear_tall_wgrad_ok(a.ecount, HP < 64 ? HP : 64, L.c, HP, h_sd);
    if (!tall_w1) {  // d W1[:,1:] (+)= Dsum^T * h
      GemmArgs g{};
      g.A = a.Dsum; g.sam = 1; g.sak = HP;
      g.B = h + a.e0 * h_sd; g.sbk = h_sd; g.sbn = h_sc;
      g.C = w + P.o_hpart; g.scm = L.c; g.scn = 1;
      g.M = HP; g.N = L.c; g.K = a.ecount; g.c_split_stride = HP * L.c; g.flags = accum;
      if (ck == 0) nsp_h = gnf_gemm_num_splits(a.ecount, kSplits);
      if ((rc = gnf_gemm_launch(g, ck == 0 ? kSplits : (int)nsp_h, s))) return rc;
    }
    if (indw) {
      if ((rc = gnf_rowsum2_launch(a.wpart, w + P.o_dW[1], wpart_rows, (int64_t)(NH - 1) * HP * HP, ck > 0, a.part,
                                   w + P.o_vec, part_rows, vecw, ck > 0, s)))
        return rc;
    } else if ((rc = rowsum(a.part, w + P.o_vec, part_rows, vecw, ck > 0))) return rc;
    if (tall_w1) {
      nsp_h = 0;                       // (no split-K partials to sum behind the loop)
      for (int64_t u0 = 0; u0 < HP; u0 += 64) {
        const int64_t nu = HP - u0 < 64 ? HP - u0 : 64;
        if ((rc = gnf_linear_tall_wgrad(a.Dsum + u0, HP, h + a.e0 * h_sd, h_sd, w + P.o_dW1h + u0 * L.c,
                                        (HT > 4 || a.ones || wide) ? w + P.o_vec + 2 * HP + u0 : nullptr, 1, a.ecount, nu, L.c,
                                        w + P.o_hpart, w + P.o_rs, s)))
          return rc;
      }
    } else if (HT > 4 || a.ones || wide) {     // first-layer bias gradient (and wide nets' others): column sums of staged arrays
      if ((rc = gnf_rowsum_tall_launch(a.Dsum, w + P.o_vec + 2 * HP, groups * 16, HP, 1, w + P.o_rs, s))) return rc;
    }
    // widest nets (H = 161..256): bias gradients of the hidden->hidden layers = column sums of the staged dpre rows --
    // on BOTH first-layer paths (inside the else-branch above they were skipped whenever tall_w1 held: zero db_l)
    if (HT > 10)
      for (int l = 1; l < NH; ++l)
        if ((rc = gnf_rowsum_tall_launch(a.SD[l], w + P.o_vec + (2 + l) * HP, rows, HP, 1, w + P.o_rs, s))) return rc;
  }
  if ((HT == 7 || HT == 10) && !wide)  // bias gradients of the hidden->hidden layers: partial column sums of mono_dw_k
    for (int l = 1; l < NH; ++l)
      if ((rc = rowsum(w + P.o_bpart[l], w + P.o_vec + (2 + l) * HP, kDwGrid, HP, 1))) return rc;
  UnpackArgs u{};
  u.net = *net; u.L = L;
  for (int l = 0; l <= NH; ++l) { u.gW[l] = gW[l]; u.gb[l] = gb[l]; if (!gW[l] || !gb[l]) return GNF_EINVAL; }
  for (int l = 1; l < NH; ++l) {
    if (!indw && (rc = rowsum(w + P.o_gpart[l], w + P.o_dW[l], nsp_w, HP * HP, 0))) return rc;
    u.dWpad[l] = w + P.o_dW[l];
  }
  if (nsp_h > 0 && (rc = rowsum(w + P.o_hpart, w + P.o_dW1h, nsp_h, HP * L.c, 0))) return rc;
  u.dW1h = w + P.o_dW1h; u.vec = w + P.o_vec; u.ones = a.ones;
  hipLaunchKernelGGL(mono_unpack_k, dim3(64), dim3(256), 0, s, u);
  GNF_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
