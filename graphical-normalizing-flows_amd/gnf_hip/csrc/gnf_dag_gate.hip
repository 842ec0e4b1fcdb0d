// DAG conditioner gate (models/Conditionners/DAGConditioner.py:94-166): builds the masked
// copies e[b,i,:] = x[b,:] * gate(importance(A[i,:])) that feed the embedding net, and the
// backward onto A and x.  HBM-bound: the forward's only real traffic is the write of e
// ([B*d, d] fp32); importance, Gumbel noise (Philox) and the gate are computed in
// registers, so the two [B,d,d] noise tensors of the reference are never materialised.
#include "gnf_common.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// importance p(A) and dp/dA   (DAG:118-124, :151-153)
__device__ __forceinline__ float importance(float a, int mode, float h_thresh, float* dp_da) {
  if (mode == 0) { *dp_da = 1.f; return a; }
  if (mode == 3) {
    const float a2 = a * a;
    const bool on = a2 > h_thresh;
    *dp_da = on ? 2.f * a : 0.f;
    return on ? a2 : 0.f;
  }
  const float s = sigmoidf(2.f * a * a);
  const float G = 2.f * (s - .5f);
  const float dG = 8.f * a * s * (1.f - s);
  if (mode == 2) {
    const bool on = G > h_thresh;
    *dp_da = on ? dG : 0.f;
    return on ? G : 0.f;
  }
  *dp_da = dG;
  return G;
}

// Per-(i,j) table, built once per call (d*d entries, negligible next to the B*d*d gate work):
//   P  = importance p,   dP = dp/dA,
//   ET = ((1-p+eps)/(p+eps))^(1/T)         Gumbel gate:  z1/(z1+z2) = 1/(1 + ET * (ln u1/ln u2)^(1/T))
//   Q  = (1/(p+eps) + 1/(1-p+eps))/T       d gate/dp = gate (1-gate) Q
// so the per-element work is the noise plus two logarithms (the reference's four logs, two exps and the
// sigmoid collapse algebraically; same value to fp32 rounding, no overflow for large Gumbel draws).
__global__ void dag_gate_tab_k(const float* __restrict__ A, float* __restrict__ tab, int imp_mode, float h_thresh,
                               float T, int64_t dd) {
  const int64_t ij = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ij >= dd) return;
  float dpda;
  const float p = importance(A[ij], imp_mode, h_thresh, &dpda);
  const float eps = 1e-6f;
  const float pa = p + eps, pb = 1.f - p + eps;
  tab[ij] = p;
  tab[dd + ij] = dpda;
  tab[2 * dd + ij] = expf((logf(pb) - logf(pa)) / T);
  tab[3 * dd + ij] = (1.f / pa + 1.f / pb) / T;
}

// Column plan of the backward (round 5): dL/dA[i,j] = dP/dA[i,j] * sum_b ..., so wherever dP/dA is exactly zero -- every entry
// of A that is zero under the soft / hard thresholds: 97.2 % of the MNIST prior (NormalizingFlowFactories.py:35-46) -- the
// cotangent of e[b,i,j] is multiplied by an exact 0 and never needed (as long as x wants no gradient).  One wavefront per
// row i compacts the columns with dP/dA != 0 in ascending order:
//   plan[i] = their number (also when it exceeds KC),   cols[i][k] (int16, behind the d counts) = the k-th such column.
// A consumer uses the lists only if NO row holds more than KC columns (the flag word behind them) and runs its dense path otherwise:
// the decision is taken on the device from the same table the forward used -- nothing is cached on the host.
constexpr int KC = GNF_DAG_PLAN_KC;
// the table of dag_gate_tab_k AND the plan in one launch: one workgroup per row i
__global__ __launch_bounds__(kBlock) void dag_gate_tab_plan_k(const float* __restrict__ A, float* __restrict__ tab,
                                                              int imp_mode, float h_thresh, float T, int64_t d,
                                                              int32_t* __restrict__ plan) {
  __shared__ int wcnt[kBlock / 64];
  const int64_t i = blockIdx.x, dd = d * d;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int16_t* cols = reinterpret_cast<int16_t*>(plan + d) + i * KC;
  int n = 0;                                               // columns found so far (the same in every thread)
  for (int64_t j0 = 0; j0 < d; j0 += kBlock) {
    const int64_t j = j0 + tid, ij = i * d + j;
    bool on = false;
    if (j < d) {
      float dpda;
      const float p = importance(A[ij], imp_mode, h_thresh, &dpda);
      const float eps = 1e-6f;
      const float pa = p + eps, pb = 1.f - p + eps;
      tab[ij] = p;
      tab[dd + ij] = dpda;
      tab[2 * dd + ij] = expf((logf(pb) - logf(pa)) / T);
      tab[3 * dd + ij] = (1.f / pa + 1.f / pb) / T;
      on = dpda != 0.f;
    }
    const uint64_t m = __ballot(on);
    if (lane == 0) wcnt[w] = __popcll(m);
    __syncthreads();
    int base = n, tot = n;
#pragma unroll
    for (int ww = 0; ww < kBlock / 64; ++ww) { base += ww < w ? wcnt[ww] : 0; tot += wcnt[ww]; }
    const int k = base + __popcll(m & ((1ull << lane) - 1ull));
    if (on && k < KC) cols[k] = (int16_t)j;
    n = tot;
    __syncthreads();
  }
  if (tid >= n && tid < KC) cols[tid] = (int16_t)-1;
  if (tid == 0) plan[i] = n;
}

// plan[d + d*KC/2] = 1 when a row of the plan holds more than KC columns (consumers then run their dense code), else 0:
// written by workgroup (0, 0) of the gate forward that follows the table launch in the same stream
__device__ __forceinline__ int64_t plan_flag_index(int64_t d) { return d + (d * KC + 1) / 2; }
__device__ __forceinline__ void plan_write_flag(int32_t* __restrict__ plan, int64_t d) {
  int over = 0;
  for (int64_t t = threadIdx.x; t < d; t += blockDim.x) over |= plan[t] > KC;
  over = __syncthreads_or(over);
  if (threadIdx.x == 0) plan[plan_flag_index(d)] = over ? 1 : 0;
}

// Noise of the FOUR adjacent columns 4 jq .. 4 jq + 3 of row (b*d + i), one value per column:
//   gate_mode 1 (Gumbel-softmax): v = exp(g2 - g1) = E1 / E2, the ratio of two independent Exp(1) variates.
//     * injected uniforms (parity tests, the reference's draw order u1 then u2):  v = ln u1 / ln u2;
//     * Philox:  E1 / (E1 + E2) is EXACTLY uniform on (0,1), so v = V / (1 - V) with ONE uniform V has exactly the law of
//       the reference's ratio -- one 32-bit word and one division per element instead of two words and two logarithms.
//       One Philox4x32-10 call (the dominant VALU cost of these kernels: 40 quarter-rate integer multiplies) then serves
//       four columns: the counter is the column QUAD (row * ceil(d/4) + jq).
//   gate_mode 2 (noise gate): one standard normal per column; Philox: both Box-Muller outputs of each uniform pair.
// Forward and backward use the same mapping, so the backward regenerates the forward's noise.
// gate_mode 1: the ratio is handed on as numerator v / denominator w (the gate then needs ONE division, see gumbel_gate)
struct Draw4 { float v[4]; float w[4]; };

// 23-bit uniform centred in its cell: never exactly 0 or 1.  (With 24 bits the + .5f is a rounding tie for words >= 2^23 and
// rounds to even: 16777215.5 -> 16777216, i.e. V = 1.0f about four times per cfg4 step and V / (1 - V) = inf in the
// single-uniform Gumbel ratio.  With 23 bits k + .5 is exactly representable for every k < 2^23.)
__device__ __forceinline__ float u01_open(uint32_t w) {
  return ((float)(w >> 9) + .5f) * (1.0f / 8388608.0f);
}

__device__ __forceinline__ Draw4 draw4(int gate_mode, const float* u1, const float* u2, uint64_t seed, uint64_t offset,
                                       int64_t row, int64_t jq, int64_t d) {
  Draw4 n;
#pragma unroll
  for (int h = 0; h < 4; ++h) { n.v[h] = 0.f; n.w[h] = 1.f; }
  if (gate_mode == 0) return n;
  const int64_t j0 = 4 * jq;
  if (u1) {
#pragma unroll
    for (int h = 0; h < 4; ++h)
      if (j0 + h < d) {
        const float a = u1[row * d + j0 + h];
        if (gate_mode == 1) { n.v[h] = log2f(a); n.w[h] = log2f(u2[row * d + j0 + h]); }      // ln u1 / ln u2 = exp(g2 - g1)
        else n.v[h] = a;
      }
    return n;
  }
  const uint64_t idx = (uint64_t)(row * ((d + 3) / 4) + jq);
  uint32_t r[4];
  philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)offset, (uint32_t)(offset >> 32),
                (uint32_t)seed, (uint32_t)(seed >> 32), r);
  if (gate_mode == 1) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float V = u01_open(r[h]);
      n.v[h] = V;
      n.w[h] = 1.f - V;
    }
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float rad = sqrtf(-2.f * logf(u01_open(r[2 * h]))), ang = 6.283185307179586f * u01_open(r[2 * h + 1]);
      n.v[2 * h] = rad * cosf(ang);
      n.v[2 * h + 1] = rad * sinf(ang);
    }
  }
  return n;
}

// four consecutive floats: one 16-B load when the quad is whole and 16-B aligned (vec), scalar loads otherwise
__device__ __forceinline__ void load4(const float* __restrict__ p, int nv, bool vec, float (&o)[4]) {
  if (vec && nv == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
  } else {
#pragma unroll
    for (int h = 0; h < 4; ++h) o[h] = h < nv ? p[h] : 0.f;
  }
}
__device__ __forceinline__ bool quad_aligned(const float* base, int64_t ld) {
  return ((ld | (int64_t)(reinterpret_cast<uintptr_t>(base) >> 2)) & 3) == 0;
}

// Gumbel-softmax gate z1/(z1+z2) = 1/(1 + ET * v^(1/T)) from the table entry ET and the ratio v = num / den.  For the two
// temperatures the drivers use the ratio is never formed: 1/(1 + ET num/den) = den / (den + ET num) -- ONE IEEE division per
// gate instead of two (a division is ~10 VALU instructions, the two of them cost as much as the gate's share of the Philox
// call: 0.094 -> 0.08 ms forward, 0.111 -> 0.09 ms backward at cfg4).  num, den have the same sign (both logs <= 0, or V and
// 1 - V in (0, 1)); num = 0 -> 1, den = 0 -> 0, as the two-division form.
// Round 5: the one division is v_rcp_f32 (1 ulp) + a multiply instead of the IEEE sequence (~10 VALU instructions with its
// scaling and fix-up): the gate is a random draw compared at 1e-5 relative in the injected-noise parity tests, ~2 ulp is far
// inside that.  Same limits: denominator inf -> gate 0; num = den = 0 -> NaN either way.
__device__ __forceinline__ float gumbel_gate(float ET, float num, float den, float T) {
  if (T == 1.f) return den * __builtin_amdgcn_rcpf(fmaf(ET, num, den));
  if (T == .5f) { const float n2 = num * num, d2 = den * den; return d2 * __builtin_amdgcn_rcpf(fmaf(ET, n2, d2)); }
  const float vT = exp2f(log2f(num / den) / T);
  return 1.f / (1.f + ET * vT);                       // u1 -> 0: gate 0;  u2 -> 0: gate 1 (as the reference)
}

struct GateArgs {
  const float* x; const float* tab; float* e; const float* ge; int64_t ld_e;
  int gate_mode; float T; const float* u1; const float* u2; uint64_t seed, offset; int hot;
  float* ws; float* gA; float* gx; int64_t B, d, chunk;
  int32_t* plan; const float* gec; float* part_sp; int64_t chunk_sp;
};

// One thread per (i, column quad), looping over the samples of its chunk (blockIdx.y): the table entries of the quad are
// read once, every iteration is one Philox call + four gates + one 16-B store.  (One tiny thread per (b, i, quad)
// -- 78 400 workgroups at cfg4 -- spent its time on the dependent table loads: 0.17 ms; this form 0.06.)
__global__ void dag_gate_fwd_k(GateArgs a) {
  if (a.plan && blockIdx.x == 0 && blockIdx.y == 0) plan_write_flag(a.plan, a.d);    // (uniform per workgroup)
  const int64_t d = a.d, dd = d * d, dq = (d + 3) / 4;
  const int64_t ip = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ip >= d * dq) return;
  const int64_t i = ip / dq, jq = ip - i * dq, j0 = 4 * jq;
  const int nv = d - j0 < 4 ? (int)(d - j0) : 4;
  const bool vt = quad_aligned(a.tab, d), vx = quad_aligned(a.x, d), ve = quad_aligned(a.e, a.ld_e);
  float p4[4], et4[4] = {0.f, 0.f, 0.f, 0.f};
  load4(a.tab + i * d + j0, nv, vt, p4);
  if (a.gate_mode == 1) load4(a.tab + 2 * dd + i * d + j0, nv, vt, et4);
  const int64_t b0 = (int64_t)blockIdx.y * a.chunk;
  const int64_t b1 = b0 + a.chunk < a.B ? b0 + a.chunk : a.B;
  for (int64_t b = b0; b < b1; ++b) {
    const int64_t bi = b * d + i;
    const Draw4 n = draw4(a.gate_mode, a.u1, a.u2, a.seed, a.offset, bi, jq, d);
    float out[4], x4[4];
    load4(a.x + b * d + j0, nv, vx, x4);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float p = p4[h], xv = x4[h];
      if (a.gate_mode == 0) out[h] = xv * p;
      else out[h] = a.gate_mode == 1 ? xv * gumbel_gate(et4[h], n.v[h], n.w[h], a.T) : p * (xv + n.v[h] * fabsf(1.f - p));
    }
    float* erow = a.e + bi * a.ld_e + j0;
    if (nv == 4 && ve) {
      *reinterpret_cast<float4*>(erow) = make_float4(out[0], out[1], out[2], out[3]);
    } else {
#pragma unroll
      for (int h = 0; h < 4; ++h)
        if (h < nv) erow[h] = out[h];
    }
    if (a.hot) {
#pragma unroll
      for (int h = 0; h < 4; ++h)
        if (h < nv) erow[d + h] = (j0 + h == i) ? 1.f : 0.f;
    }
  }
}

// dL/dp partial sums over a chunk of b:  ws[chunk][i*d+j] = sum_b ge[b,i,j] * de/dp[b,i,j];  one thread per (i, column quad);
// (bx, by) = the workgroup's position in the (column-quad blocks, sample chunks) grid
__device__ __forceinline__ void dp_dense_unit(const GateArgs& a, int64_t bx, int64_t by) {
  const int64_t d = a.d, dd = d * d, dq = (d + 3) / 4;
  const int64_t ip = bx * blockDim.x + threadIdx.x;
  if (ip >= d * dq) return;
  const int64_t i = ip / dq, jq = ip - i * dq, j0 = 4 * jq;
  const int nv = d - j0 < 4 ? (int)(d - j0) : 4;
  float p[4], ET[4], Q[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const int64_t ij = i * d + j0 + (h < nv ? h : 0);
    p[h] = a.tab[ij]; ET[h] = a.tab[2 * dd + ij]; Q[h] = a.tab[3 * dd + ij];
  }
  const int64_t b0 = by * a.chunk;
  const int64_t b1 = b0 + a.chunk < a.B ? b0 + a.chunk : a.B;
  const bool vg = quad_aligned(a.ge, a.ld_e), vx = quad_aligned(a.x, d);
  for (int64_t b = b0; b < b1; ++b) {
    const Draw4 n = draw4(a.gate_mode, a.u1, a.u2, a.seed, a.offset, b * d + i, jq, d);
    float g4[4], x4[4];
    load4(a.ge + (b * d + i) * a.ld_e + j0, nv, vg, g4);
    load4(a.x + b * d + j0, nv, vx, x4);
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      if (h >= nv) break;
      const float g = g4[h];
      const float xv = x4[h];
      if (a.gate_mode == 0) {
        acc[h] = fmaf(g, xv, acc[h]);
      } else if (a.gate_mode == 1) {
        const float s = gumbel_gate(ET[h], n.v[h], n.w[h], a.T);
        acc[h] = fmaf(g * xv, s * (1.f - s) * Q[h], acc[h]);
      } else {
        const float om = 1.f - p[h];
        const float sgn = om > 0.f ? 1.f : (om < 0.f ? -1.f : 0.f);
        acc[h] = fmaf(g, xv + n.v[h] * fabsf(om) - p[h] * n.v[h] * sgn, acc[h]);
      }
    }
  }
#pragma unroll
  for (int h = 0; h < 4; ++h)
    if (h < nv) a.ws[by * dd + i * d + j0 + h] = acc[h];
}

__global__ void dag_gate_bwd_dp_k(GateArgs a) { dp_dense_unit(a, blockIdx.x, blockIdx.y); }

// The same sums from the COMPACT cotangent gec[(b*d+i)][k] = dL/de[b,i,cols[i][k]] that the conv backward of the fused
// masked-image front leaves (gnf_mnistcnn_conv_bwd_cols): one thread per (i, slot k < plan[i]) and chunk of samples,
//   part_sp[chunk][i*KC + k] = sum_b gec[b,i,k] * de/dp[b,i,j],  j = cols[i][k]
// -- 17 172 (i,j) pairs instead of 614 656 at the MNIST prior.  The noise of (b,i,j) is element j & 3 of the Philox call
// of its column quad, exactly as the forward drew it.  1-D grid of gx_sp * nc_sp workgroups; if a row of the plan
// overflows, the same workgroups walk the units of the dense kernel over ge instead (the conv backward took the same
// decision from the same word and wrote the dense ge).
__global__ void dag_gate_bwd_dp_plan_k(GateArgs a, unsigned gx_dense, unsigned ny_dense, unsigned gx_sp) {
  const int64_t d = a.d, dd = d * d;
  if (a.plan[plan_flag_index(d)] != 0) {                       // (the same word for every thread)
    for (unsigned u = blockIdx.x; u < gx_dense * ny_dense; u += gridDim.x) dp_dense_unit(a, u % gx_dense, u / gx_dense);
    return;
  }
  const int64_t bx = blockIdx.x % gx_sp, by = blockIdx.x / gx_sp;
  const int64_t it = bx * blockDim.x + threadIdx.x;
  if (it >= d * KC) return;
  const int64_t i = it / KC;
  const int k = (int)(it - i * KC);
  const int cnt = a.plan[i];
  const int64_t j = reinterpret_cast<const int16_t*>(a.plan + d)[it];
  if (k >= cnt) return;
  const int64_t ij = i * d + j, jq = j >> 2;
  const int h = (int)(j & 3);
  const float p = a.tab[ij], ET = a.tab[2 * dd + ij], Q = a.tab[3 * dd + ij];
  const int64_t b0 = by * a.chunk_sp;
  const int64_t b1 = b0 + a.chunk_sp < a.B ? b0 + a.chunk_sp : a.B;
  float acc = 0.f;
  constexpr int UB = 4;                                        // samples per round: their loads are requested together
  for (int64_t bb = b0; bb < b1; bb += UB) {
    float g[UB], xv[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int64_t b = bb + u < b1 ? bb + u : b1 - 1;         // (clamped: a repeated sample is weighted 0 below)
      g[u] = a.gec[(b * d + i) * KC + k];
      xv[u] = a.x[b * d + j];
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (bb + u >= b1) break;
      const Draw4 n = draw4(a.gate_mode, a.u1, a.u2, a.seed, a.offset, (bb + u) * d + i, jq, d);
      float nv = n.v[0], nw = n.w[0];                          // element h of the quad (h is not a compile-time index)
      nv = h == 1 ? n.v[1] : nv; nw = h == 1 ? n.w[1] : nw;
      nv = h == 2 ? n.v[2] : nv; nw = h == 2 ? n.w[2] : nw;
      nv = h == 3 ? n.v[3] : nv; nw = h == 3 ? n.w[3] : nw;
      if (a.gate_mode == 0) {
        acc = fmaf(g[u], xv[u], acc);
      } else if (a.gate_mode == 1) {
        const float s = gumbel_gate(ET, nv, nw, a.T);
        acc = fmaf(g[u] * xv[u], s * (1.f - s) * Q, acc);
      } else {
        const float om = 1.f - p;
        const float sgn = om > 0.f ? 1.f : (om < 0.f ? -1.f : 0.f);
        acc = fmaf(g[u], xv[u] + nv * fabsf(om) - p * nv * sgn, acc);
      }
    }
  }
  a.part_sp[by * d * KC + it] = acc;
}

// gA = (sum over the nc batch chunks, in chunk order: deterministic) * dP/dA.  The chunks are few (<= 16) and long (d*d):
// one pass with four elements per thread; as a launch of the shared row-sum kernel (16 wavefronts per 64 columns, made for
// MANY short partial rows) plus this product the tail of the cfg4 gate backward was 18 + 5 us.
__global__ void dag_gate_bwd_dA_k(const float* __restrict__ tab, const float* __restrict__ part, int nc,
                                  float* __restrict__ gA, int64_t dd) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= dd) return;
  if (i4 + 3 < dd && (dd & 3) == 0) {
    float4 s = *reinterpret_cast<const float4*>(part + i4);
    for (int c = 1; c < nc; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)c * dd + i4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float4 t = *reinterpret_cast<const float4*>(tab + dd + i4);
    *reinterpret_cast<float4*>(gA + i4) = make_float4(s.x * t.x, s.y * t.y, s.z * t.z, s.w * t.w);
    return;
  }
  for (int64_t ij = i4; ij < dd && ij < i4 + 4; ++ij) {
    float s = part[ij];
    for (int c = 1; c < nc; ++c) s += part[(int64_t)c * dd + ij];
    gA[ij] = s * tab[dd + ij];
  }
}

// gA for the plan variant, one workgroup per row i: the dense chunk sums (a row of the plan overflowed) or zeros + the chunk
// sums of the row's <= KC slots scattered to their columns; entries with dP/dA == 0 get the exact 0 the dense product gives
__global__ __launch_bounds__(kBlock) void dag_gate_bwd_dA_plan_k(const float* __restrict__ tab, const float* __restrict__ part,
                                                                 int nc, const float* __restrict__ part_sp, int nc_sp,
                                                                 const int32_t* __restrict__ plan, float* __restrict__ gA,
                                                                 int accumulate, int64_t d) {
  const int64_t dd = d * d, i = blockIdx.x;
  const int tid = threadIdx.x;
  if (plan[plan_flag_index(d)] != 0) {
    for (int64_t j = tid; j < d; j += kBlock) {
      const int64_t ij = i * d + j;
      float s = part[ij];
      for (int c = 1; c < nc; ++c) s += part[(int64_t)c * dd + ij];
      gA[ij] = accumulate ? fmaf(s, tab[dd + ij], gA[ij]) : s * tab[dd + ij];
    }
    return;
  }
  // eight threads per slot split the chunk partials (a serial walk over ~20 chunks was 20 dependent round trips: 9.7 us for
  // this launch), their sums meet in a fixed butterfly order
  const int cnt = plan[i];
  const int k = tid >> 3, cl = tid & 7;
  static_assert(kBlock == 8 * KC, "one 8-thread group per slot");
  const int64_t jk = reinterpret_cast<const int16_t*>(plan + d)[i * KC + k];
  float s = 0.f;
  if (k < cnt)
    for (int c = cl; c < nc_sp; c += 8) s += part_sp[(int64_t)c * d * KC + i * KC + k];
  s += __shfl_xor(s, 1, GNF_WAVE); s += __shfl_xor(s, 2, GNF_WAVE); s += __shfl_xor(s, 4, GNF_WAVE);
  const bool writer = cl == 0 && k < cnt;
  const float dp = writer ? tab[dd + i * d + jk] : 0.f;
  if (accumulate) {                                            // gA already holds another contribution (the acyclicity term's):
    if (writer) gA[i * d + jk] = fmaf(s, dp, gA[i * d + jk]);  // only the row's listed columns change
    return;
  }
  for (int64_t j = tid; j < d; j += kBlock) gA[i * d + j] = 0.f;
  __syncthreads();
  if (writer) gA[i * d + jk] = s * dp;
}

// gx partial sums over a chunk of i (blockIdx.y):  out[chunk][b,j] = sum_{i in chunk} ge[b,i,j] * de/dx[b,i,j];  one thread
// per (b, column quad).  (B * d / 4 threads alone are 77 workgroups at cfg4: the i loop is split to fill the chip; the
// chunk partials are added by the deterministic row-sum kernel.)
__global__ void dag_gate_bwd_dx_k(GateArgs a) {
  const int64_t d = a.d, dd = d * d, dq = (d + 3) / 4, n = a.B * dq;
  const int64_t i0 = (int64_t)blockIdx.y * a.chunk, i1 = i0 + a.chunk < d ? i0 + a.chunk : d;
  float* out = a.gx + (int64_t)blockIdx.y * a.B * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / dq, jq = e - b * dq, j0 = 4 * jq;
    const int nv = d - j0 < 4 ? (int)(d - j0) : 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const bool vg = quad_aligned(a.ge, a.ld_e), vt = quad_aligned(a.tab, d);
    for (int64_t i = i0; i < i1; ++i) {
      Draw4 nz;
      if (a.gate_mode == 1) nz = draw4(1, a.u1, a.u2, a.seed, a.offset, b * d + i, jq, d);
      float g4[4], t4[4];
      load4(a.ge + (b * d + i) * a.ld_e + j0, nv, vg, g4);
      load4(a.tab + (a.gate_mode == 1 ? 2 * dd : 0) + i * d + j0, nv, vt, t4);   // ET (Gumbel) or the importance itself
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        if (h >= nv) break;
        const float gate = a.gate_mode == 1 ? gumbel_gate(t4[h], nz.v[h], nz.w[h], a.T) : t4[h];
        acc[h] = fmaf(g4[h], gate, acc[h]);
      }
    }
#pragma unroll
    for (int h = 0; h < 4; ++h)
      if (h < nv) out[b * d + j0 + h] = acc[h];
  }
}

// chunks of the sample loop of the (i, quad)-threaded kernels: enough workgroups to fill the chip (16 per CU)
inline int64_t bwd_chunks(int64_t B, int64_t d) {
  const int64_t nblk = (d * ((d + 3) / 4) + kBlock - 1) / kBlock;
  int64_t nc = (4096 + nblk - 1) / nblk;
  if (nc > 2048) nc = 2048;
  if (nc > B) nc = B;
  if (nc < 1) nc = 1;
  return nc;
}

// chunks of the i loop of the (b, quad)-threaded dx kernel
inline int64_t dx_chunks(int64_t B, int64_t d) {
  int64_t nblk = (B * ((d + 3) / 4) + kBlock - 1) / kBlock;
  if (nblk < 1) nblk = 1;                    // empty batch
  int64_t nc = (4096 + nblk - 1) / nblk;
  if (nc > d) nc = d;
  if (nc > 64) nc = 64;
  if (nc < 1) nc = 1;
  return nc;
}

inline unsigned grid_1d(int64_t n) {
  int64_t g = (n + kBlock - 1) / kBlock;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

inline int launch_tab(const float* A, float* tab, int imp_mode, float h_thresh, float T, int64_t d, hipStream_t s) {
  hipLaunchKernelGGL(dag_gate_tab_k, dim3((unsigned)((d * d + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, A, tab,
                     imp_mode, h_thresh, T, d * d);
  GNF_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Acyclicity + l1 term of the DAG conditioner (DAGConditioner.py:176-194, 268-271) around the library matrix power:
//   Bm = I + min(1, alpha) * alpha_factor * A o A                                     (dag_loss_prep_k)
//   h = sum_ij P_ij Bm_ji - d  with P = Bm^(k-1);  loss = dag_const (lambd h + c/2 h^2) + l1 mean|A|;
//   coef = dag_const (lambd + c h) k 2 alpha: what d loss / dA needs besides P            (dag_loss_part_k + _final_k)
//   dA = g (coef A o P^T + l1 sign(A) / d^2)                                           (dag_loss_bwd_k)
// All scalars (alpha, lambd, c, dag_const, l1) are read from device memory: no host synchronisation.
// ---------------------------------------------------------------------------------------------
constexpr int kLossBlocks = 256;

__device__ __forceinline__ float alpha_eff(const float* alpha, float factor) { return fminf(*alpha, 1.f) * factor; }

__global__ void dag_loss_prep_k(const float* __restrict__ A, const float* __restrict__ alpha, float factor,
                                float* __restrict__ Bm, int64_t d) {
  const float al = alpha_eff(alpha, factor);
  for (int64_t ij = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ij < d * d; ij += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = ij / d, j = ij - i * d;
    const float a = A[ij];
    Bm[ij] = (i == j ? 1.f : 0.f) + al * a * a;
  }
}

// partial sums per workgroup: part[2 b] = sum P_ij Bm_ji, part[2 b + 1] = sum |A_ij|   (P == nullptr: k = 0, trace = d)
__global__ void dag_loss_part_k(const float* __restrict__ A, const float* __restrict__ Bm, const float* __restrict__ P,
                                float* __restrict__ part, int64_t d) {
  __shared__ float s0[kBlock / 64], s1[kBlock / 64];
  float t = 0.f, l = 0.f;
  for (int64_t ij = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ij < d * d; ij += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = ij / d, j = ij - i * d;
    if (P) t = fmaf(P[ij], Bm[j * d + i], t);
    l += fabsf(A[ij]);
  }
  t = group_sum<64>(t);
  l = group_sum<64>(l);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { s0[w] = t; s1[w] = l; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < kBlock / 64; ++k) { a += s0[k]; b += s1[k]; }
    part[2 * blockIdx.x] = a;
    part[2 * blockIdx.x + 1] = b;
  }
}

// out[0] = loss, out[1] = h (the trace term), out[2] = coef, out[3] = l1 / d^2
__global__ void dag_loss_final_k(const float* __restrict__ part, int nparts, int has_P, const float* __restrict__ alpha,
                                 float factor, const float* __restrict__ lambd, const float* __restrict__ c,
                                 const float* __restrict__ dag_const, const float* __restrict__ l1, int k, int64_t d,
                                 float* __restrict__ out) {
  const int lane = threadIdx.x;
  float t = 0.f, l = 0.f;
  for (int b = lane; b < nparts; b += 64) { t += part[2 * b]; l += part[2 * b + 1]; }
  t = group_sum<64>(t);
  l = group_sum<64>(l);
  if (lane == 0) {
    const float h = has_P ? t - (float)d : 0.f;         // k = 0: tr(I) - d
    const float dd = (float)d * (float)d;
    out[0] = *dag_const * (*lambd * h + *c / 2.f * h * h) + *l1 * (l / dd);
    out[1] = h;
    out[2] = has_P ? *dag_const * (*lambd + *c * h) * (float)k * 2.f * alpha_eff(alpha, factor) : 0.f;
    out[3] = *l1 / dd;
  }
}

__global__ void dag_loss_bwd_k(const float* __restrict__ A, const float* __restrict__ P, const float* __restrict__ sc,
                               const float* __restrict__ g, float* __restrict__ gA, int64_t d) {
  const float go = *g, coef = sc[2], l1 = sc[3];
  for (int64_t ij = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ij < d * d; ij += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = ij / d, j = ij - i * d;
    const float a = A[ij];
    const float sgn = a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f);
    gA[ij] = go * ((P ? coef * a * P[j * d + i] : 0.f) + l1 * sgn);
  }
}

}  // namespace

extern "C" {

int gnf_dag_loss_prep(const float* A, const float* alpha, float alpha_factor, float* Bm, int64_t d,
                      gnf_stream_t stream) {
  if (!A || !alpha || !Bm || d <= 0) return GNF_EINVAL;
  hipLaunchKernelGGL(dag_loss_prep_k, dim3(grid_1d(d * d)), dim3(kBlock), 0, (hipStream_t)stream, A, alpha, alpha_factor,
                     Bm, d);
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_dag_loss_value(const float* A, const float* Bm, const float* P, const float* alpha, float alpha_factor,
                       const float* lambd, const float* c, const float* dag_const, const float* l1_weight, int k,
                       float* out4, float* ws, int64_t d, gnf_stream_t stream) {
  if (!A || !Bm || !alpha || !lambd || !c || !dag_const || !l1_weight || !out4 || !ws || d <= 0 || k < 0) return GNF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dag_loss_part_k, dim3(kLossBlocks), dim3(kBlock), 0, s, A, Bm, P, ws, d);
  GNF_LAUNCH_CHECK();
  hipLaunchKernelGGL(dag_loss_final_k, dim3(1), dim3(64), 0, s, ws, kLossBlocks, P ? 1 : 0, alpha, alpha_factor, lambd, c,
                     dag_const, l1_weight, k, d, out4);
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_dag_loss_bwd(const float* A, const float* P, const float* out4, const float* g, float* gA, int64_t d,
                     gnf_stream_t stream) {
  if (!A || !out4 || !g || !gA || d <= 0) return GNF_EINVAL;
  hipLaunchKernelGGL(dag_loss_bwd_k, dim3(grid_1d(d * d)), dim3(kBlock), 0, (hipStream_t)stream, A, P, out4, g, gA, d);
  GNF_LAUNCH_CHECK();
  return 0;
}

int64_t gnf_dag_gate_fwd_ws_bytes(int64_t d) { return 4 * d * d * (int64_t)sizeof(float); }

int gnf_dag_gate_fwd(const float* x, const float* A, float* e, int64_t ld_e, int imp_mode, int gate_mode,
                     float h_thresh, float temperature, const float* u1, const float* u2, uint64_t seed,
                     uint64_t offset, int hot, float* ws, int64_t B, int64_t d, gnf_stream_t stream) {
  return gnf_dag_gate_fwd_plan(x, A, e, ld_e, imp_mode, gate_mode, h_thresh, temperature, u1, u2, seed, offset, hot, ws,
                               nullptr, 0, B, d, stream);
}

int gnf_dag_gate_fwd_plan(const float* x, const float* A, float* e, int64_t ld_e, int imp_mode, int gate_mode,
                          float h_thresh, float temperature, const float* u1, const float* u2, uint64_t seed,
                          uint64_t offset, int hot, float* ws, int32_t* plan, int64_t plan_bytes, int64_t B, int64_t d,
                          gnf_stream_t stream) {
  if (((!x || !e) && B > 0) || !A || !ws || B < 0 || d <= 0 || imp_mode < 0 || imp_mode > 3 || gate_mode < 0 ||
      gate_mode > 2)
    return GNF_EINVAL;                // batch-sized arrays may be NULL for an empty batch
  if (ld_e < (hot ? 2 * d : d)) return GNF_EINVAL;
  if (gate_mode == 1 && u1 && !u2) return GNF_EINVAL;
  if (imp_mode == 0) gate_mode = 0;   // DAG:151-153: raw A, no gate
  if (plan && d > 32767) return GNF_ESHAPE;               // 16-bit column indices
  if (plan && plan_bytes < gnf_dag_gate_plan_bytes(d)) return GNF_EWS;
  if (B == 0 && !plan) return 0;                          // (with a plan: table, plan and its flag are still written)
  hipStream_t s = (hipStream_t)stream;
  int rc = 0;
  if (plan) {
    hipLaunchKernelGGL(dag_gate_tab_plan_k, dim3((unsigned)d), dim3(kBlock), 0, s, A, ws, imp_mode, h_thresh, temperature,
                       d, plan);
    GNF_LAUNCH_CHECK();
  } else {
    rc = launch_tab(A, ws, imp_mode, h_thresh, temperature, d, s);
  }
  if (rc) return rc;
  GateArgs a{};
  a.x = x; a.tab = ws; a.e = e; a.ld_e = ld_e; a.gate_mode = gate_mode; a.T = temperature; a.u1 = u1; a.u2 = u2;
  a.seed = seed; a.offset = offset; a.hot = hot; a.B = B; a.d = d; a.plan = plan;
  const int64_t nc = bwd_chunks(B, d);
  a.chunk = B > 0 ? (B + nc - 1) / nc : 1;
  const unsigned gxp = (unsigned)((d * ((d + 3) / 4) + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(dag_gate_fwd_k, dim3(gxp, (unsigned)nc), dim3(kBlock), 0, s, a);
  GNF_LAUNCH_CHECK();
  return 0;
}

int64_t gnf_dag_gate_plan_bytes(int64_t d) { return (d + (d * KC + 1) / 2 + 1) * (int64_t)sizeof(int32_t); }

// chunks of the sample loop of the compact (i, slot)-threaded kernel: ~2048 workgroups
inline int64_t sp_chunk(int64_t B, int64_t d) {
  const int64_t gxs = (d * KC + kBlock - 1) / kBlock;
  int64_t nc = (2048 + gxs - 1) / gxs;
  if (nc > 64) nc = 64;
  if (nc > B) nc = B;
  if (nc < 1) nc = 1;
  return B > 0 ? (B + nc - 1) / nc : 1;                   // samples per chunk
}
inline int64_t sp_chunks(int64_t B, int64_t d) {
  const int64_t ch = sp_chunk(B, d);
  const int64_t nc = (B + ch - 1) / ch;
  return nc < 1 ? 1 : nc;
}

int64_t gnf_dag_gate_bwd_cols_ws_bytes(int64_t B, int64_t d) {
  // the dense layout (the fallback runs in it) | compact chunk partials [nc_sp d KC]
  return gnf_dag_gate_bwd_ws_bytes(B, d) + (sp_chunks(B, d) * d * KC + 4) * (int64_t)sizeof(float);
}

int gnf_dag_gate_bwd_cols(const float* x, const float* ge, const float* ge_cols, const int32_t* plan, int imp_mode,
                          int gate_mode, float temperature, const float* u1, const float* u2, uint64_t seed,
                          uint64_t offset, const float* tab_fwd, float* gA, int accumulate, float* ws, int64_t B,
                          int64_t d, gnf_stream_t stream) {
  if (((!x || !ge || !ge_cols) && B > 0) || !plan || !tab_fwd || !gA || !ws || B < 0 || d <= 0 || imp_mode < 0 ||
      imp_mode > 3 || gate_mode < 0 || gate_mode > 2)
    return GNF_EINVAL;
  if (gate_mode == 1 && u1 && !u2) return GNF_EINVAL;
  if (d > 32767) return GNF_ESHAPE;
  if (imp_mode == 0) gate_mode = 0;
  hipStream_t s = (hipStream_t)stream;
  GateArgs a{};
  a.x = x; a.tab = tab_fwd; a.ge = ge; a.ld_e = d; a.gate_mode = gate_mode; a.T = temperature; a.u1 = u1; a.u2 = u2;
  a.seed = seed; a.offset = offset; a.B = B; a.d = d; a.gA = gA; a.ws = ws + 4 * d * d;
  a.plan = const_cast<int32_t*>(plan); a.gec = ge_cols;
  const int64_t nc = bwd_chunks(B, d);
  a.chunk = B > 0 ? (B + nc - 1) / nc : 1;
  a.chunk_sp = sp_chunk(B, d);
  const int64_t ncs = sp_chunks(B, d);
  float* tail = ws + gnf_dag_gate_bwd_ws_bytes(B, d) / (int64_t)sizeof(float);
  a.part_sp = tail;
  const unsigned gxp = (unsigned)((d * ((d + 3) / 4) + kBlock - 1) / kBlock);
  const unsigned gxs = (unsigned)((d * KC + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(dag_gate_bwd_dp_plan_k, dim3(gxs * (unsigned)ncs), dim3(kBlock), 0, s, a, gxp, (unsigned)nc, gxs);
  GNF_LAUNCH_CHECK();
  hipLaunchKernelGGL(dag_gate_bwd_dA_plan_k, dim3((unsigned)d), dim3(kBlock), 0, s, tab_fwd, (const float*)a.ws, (int)nc,
                     (const float*)a.part_sp, (int)ncs, plan, gA, accumulate, d);
  GNF_LAUNCH_CHECK();
  return 0;
}

int64_t gnf_dag_gate_bwd_ws_bytes(int64_t B, int64_t d) {
  // table [4 d^2] | dp chunk partials [nc d^2] | their sum [d^2] | dx chunk partials [nci B d]
  return ((bwd_chunks(B, d) + 5) * d * d + dx_chunks(B, d) * B * d) * (int64_t)sizeof(float);
}

int gnf_dag_gate_bwd(const float* x, const float* A, const float* ge, int64_t ld_e, int imp_mode, int gate_mode,
                     float h_thresh, float temperature, const float* u1, const float* u2, uint64_t seed,
                     uint64_t offset, const float* tab_fwd, float* gA, float* gx, float* ws, int64_t B, int64_t d,
                     gnf_stream_t stream) {
  if (((!x || !ge) && B > 0) || !A || !ws || B < 0 || d <= 0 || imp_mode < 0 || imp_mode > 3 || gate_mode < 0 ||
      gate_mode > 2)
    return GNF_EINVAL;                // empty batch: gA = 0 through the same kernels
  if (gate_mode == 1 && u1 && !u2) return GNF_EINVAL;
  if (imp_mode == 0) gate_mode = 0;
  hipStream_t s = (hipStream_t)stream;
  // [4][d*d] table, then the chunk partials.  tab_fwd: the table the forward call left at the start of ITS workspace
  // (same A / imp_mode / h_thresh / temperature), if the caller kept that buffer: one launch less
  const float* tab = tab_fwd ? tab_fwd : ws;
  int rc = tab_fwd ? 0 : launch_tab(A, ws, imp_mode, h_thresh, temperature, d, s);
  if (rc) return rc;
  GateArgs a{};
  a.x = x; a.tab = tab; a.ge = ge; a.ld_e = ld_e; a.gate_mode = gate_mode; a.T = temperature; a.u1 = u1; a.u2 = u2;
  a.seed = seed; a.offset = offset; a.B = B; a.d = d; a.gA = gA; a.gx = gx; a.ws = ws + 4 * d * d;
  if (gA) {
    const int64_t nc = bwd_chunks(B, d);
    a.chunk = B > 0 ? (B + nc - 1) / nc : 1;
    const unsigned gxp = (unsigned)((d * ((d + 3) / 4) + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(dag_gate_bwd_dp_k, dim3(gxp, (unsigned)nc), dim3(kBlock), 0, s, a);
    GNF_LAUNCH_CHECK();
    // second stage.  A small-d / large-B call (POWER: d = 6, B = 10000) has 2048 chunk rows of only 36 columns: the
    // row-sum kernel spreads the rows over 16 wavefronts (one thread per column walking them serially took 0.45 ms);
    // a few long chunks (MNIST: d*d = 614 656 columns) are summed by the product kernel itself
    const float* sums = a.ws;
    int nsum = (int)nc;
    if (nc > 16) {
      float* tot = a.ws + nc * d * d;
      rc = gnf_rowsum_launch(a.ws, tot, nc, d * d, 0, s);
      if (rc) return rc;
      sums = tot;
      nsum = 1;
    }
    const unsigned gx4 = (unsigned)(((d * d + 3) / 4 + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(dag_gate_bwd_dA_k, dim3(gx4), dim3(kBlock), 0, s, tab, sums, nsum, gA, d * d);
    GNF_LAUNCH_CHECK();
  }
  if (gx && B > 0) {
    const int64_t nci = dx_chunks(B, d);
    a.chunk = (d + nci - 1) / nci;
    float* part = ws + (bwd_chunks(B, d) + 5) * d * d;
    a.gx = nci > 1 ? part : gx;                            // a single chunk writes gx directly
    hipLaunchKernelGGL(dag_gate_bwd_dx_k, dim3(grid_1d(B * ((d + 3) / 4)), (unsigned)nci), dim3(kBlock), 0, s, a);
    GNF_LAUNCH_CHECK();
    if (nci > 1) {
      rc = gnf_rowsum_launch(part, gx, nci, B * d, 0, s);
      if (rc) return rc;
    }
  }
  return 0;
}

}  // extern "C"
