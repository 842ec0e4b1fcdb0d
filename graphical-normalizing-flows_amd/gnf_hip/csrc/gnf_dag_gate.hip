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

struct Noise { float a, b; };

// gate_mode 1: two uniforms; gate_mode 2: one standard normal (Box-Muller on the Philox pair).
__device__ __forceinline__ Noise draw(int gate_mode, const float* u1, const float* u2, uint64_t seed, uint64_t offset,
                                      int64_t idx) {
  Noise n{0.f, 0.f};
  if (gate_mode == 0) return n;
  if (u1) {
    n.a = u1[idx];
    n.b = (gate_mode == 1) ? u2[idx] : 0.f;
    return n;
  }
  uint32_t r[4];
  philox4x32_10((uint32_t)idx, (uint32_t)((uint64_t)idx >> 32), (uint32_t)offset, (uint32_t)(offset >> 32),
                (uint32_t)seed, (uint32_t)(seed >> 32), r);
  if (gate_mode == 1) {
    n.a = u01_24(r[0]);
    n.b = u01_24(r[1]);
  } else {
    const float ua = (float)((r[0] >> 8) + 1u) * (1.0f / 16777216.0f);  // (0,1]
    n.a = sqrtf(-2.f * logf(ua)) * cosf(6.283185307179586f * u01_24(r[1]));
  }
  return n;
}

// gate value s and ds/dp for the Gumbel relaxation (DAG:95-103):
//   z1/(z1+z2) with z1 = exp((log(p+eps)+g1)/T), z2 = exp((log(1-p+eps)+g2)/T)
//   == sigmoid((log(p+eps) - log(1-p+eps) + g1 - g2)/T)   (same value, no overflow)
__device__ __forceinline__ float gumbel_gate(float p, Noise n, float T, float* ds_dp) {
  const float eps = 1e-6f;
  const float g1 = -logf(-logf(n.a));
  const float g2 = -logf(-logf(n.b));
  const float pa = p + eps, pb = 1.f - p + eps;
  const float t = ((logf(pa) + g1) - (logf(pb) + g2)) / T;
  const float s = sigmoidf(t);
  *ds_dp = s * (1.f - s) / T * (1.f / pa + 1.f / pb);
  return s;
}

__global__ void dag_gate_fwd_k(const float* __restrict__ x, const float* __restrict__ A, float* __restrict__ e,
                               int64_t ld_e, int imp_mode, int gate_mode, float h_thresh, float T,
                               const float* __restrict__ u1, const float* __restrict__ u2, uint64_t seed,
                               uint64_t offset, int hot, int64_t B, int64_t d) {
  const int64_t total = B * d * d;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = idx % d;
    const int64_t bi = idx / d;       // b*d + i
    const int64_t i = bi % d;
    const int64_t b = bi / d;
    float dpda;
    const float p = importance(A[i * d + j], imp_mode, h_thresh, &dpda);
    const float xv = x[b * d + j];
    float out;
    if (gate_mode == 0) {
      out = xv * p;
    } else {
      const Noise n = draw(gate_mode, u1, u2, seed, offset, idx);
      if (gate_mode == 1) {
        float ds;
        out = xv * gumbel_gate(p, n, T, &ds);
      } else {
        out = p * (xv + n.a * fabsf(1.f - p));
      }
    }
    e[bi * ld_e + j] = out;
    if (hot) e[bi * ld_e + d + j] = (j == i) ? 1.f : 0.f;
  }
}

// dL/dp partial sums over a chunk of b:  ws[chunk][i*d+j] = sum_b ge[b,i,j] * de/dp[b,i,j]
__global__ void dag_gate_bwd_dp_k(const float* __restrict__ x, const float* __restrict__ A,
                                  const float* __restrict__ ge, int64_t ld_e, int imp_mode, int gate_mode,
                                  float h_thresh, float T, const float* __restrict__ u1,
                                  const float* __restrict__ u2, uint64_t seed, uint64_t offset,
                                  float* __restrict__ ws, int64_t B, int64_t d, int64_t chunk) {
  const int64_t ij = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ij >= d * d) return;
  const int64_t i = ij / d, j = ij - i * d;
  float dpda;
  const float p = importance(A[ij], imp_mode, h_thresh, &dpda);
  const int64_t b0 = (int64_t)blockIdx.y * chunk;
  const int64_t b1 = b0 + chunk < B ? b0 + chunk : B;
  float acc = 0.f;
  for (int64_t b = b0; b < b1; ++b) {
    const float g = ge[(b * d + i) * ld_e + j];
    const float xv = x[b * d + j];
    if (gate_mode == 0) {
      acc = fmaf(g, xv, acc);
    } else {
      const Noise n = draw(gate_mode, u1, u2, seed, offset, (b * d + i) * d + j);
      if (gate_mode == 1) {
        float ds;
        gumbel_gate(p, n, T, &ds);
        acc = fmaf(g * xv, ds, acc);
      } else {
        const float om = 1.f - p;
        const float sgn = om > 0.f ? 1.f : (om < 0.f ? -1.f : 0.f);
        acc = fmaf(g, xv + n.a * fabsf(om) - p * n.a * sgn, acc);
      }
    }
  }
  ws[(int64_t)blockIdx.y * d * d + ij] = acc;
}

__global__ void dag_gate_bwd_dA_k(const float* __restrict__ A, const float* __restrict__ ws, float* __restrict__ gA,
                                  int imp_mode, float h_thresh, int64_t d, int64_t nchunk) {
  const int64_t ij = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ij >= d * d) return;
  float s = 0.f;
  for (int64_t c = 0; c < nchunk; ++c) s += ws[c * d * d + ij];
  float dpda;
  importance(A[ij], imp_mode, h_thresh, &dpda);
  gA[ij] = s * dpda;
}

// gx[b,j] = sum_i ge[b,i,j] * de/dx[b,i,j]
__global__ void dag_gate_bwd_dx_k(const float* __restrict__ A, const float* __restrict__ ge, int64_t ld_e,
                                  int imp_mode, int gate_mode, float h_thresh, float T,
                                  const float* __restrict__ u1, const float* __restrict__ u2, uint64_t seed,
                                  uint64_t offset, float* __restrict__ gx, int64_t B, int64_t d) {
  const int64_t n = B * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / d, j = e - b * d;
    float acc = 0.f;
    for (int64_t i = 0; i < d; ++i) {
      float dpda;
      const float p = importance(A[i * d + j], imp_mode, h_thresh, &dpda);
      float gate = p;
      if (gate_mode == 1) {
        const Noise nz = draw(gate_mode, u1, u2, seed, offset, (b * d + i) * d + j);
        float ds;
        gate = gumbel_gate(p, nz, T, &ds);
      }
      acc = fmaf(ge[(b * d + i) * ld_e + j], gate, acc);
    }
    gx[e] = acc;
  }
}

inline int64_t bwd_chunks(int64_t B, int64_t d) {
  const int64_t nblk = (d * d + kBlock - 1) / kBlock;
  int64_t nc = 2048 / nblk;
  if (nc < 1) nc = 1;
  if (nc > B) nc = B;
  if (nc < 1) nc = 1;
  return nc;
}

inline unsigned grid_1d(int64_t n) {
  int64_t g = (n + kBlock - 1) / kBlock;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" {

int gnf_dag_gate_fwd(const float* x, const float* A, float* e, int64_t ld_e, int imp_mode, int gate_mode,
                     float h_thresh, float temperature, const float* u1, const float* u2, uint64_t seed,
                     uint64_t offset, int hot, int64_t B, int64_t d, gnf_stream_t stream) {
  if (!x || !A || !e || B < 0 || d <= 0 || imp_mode < 0 || imp_mode > 3 || gate_mode < 0 || gate_mode > 2)
    return GNF_EINVAL;
  if (ld_e < (hot ? 2 * d : d)) return GNF_EINVAL;
  if (gate_mode == 1 && u1 && !u2) return GNF_EINVAL;
  if (imp_mode == 0) gate_mode = 0;   // DAG:151-153: raw A, no gate
  if (B == 0) return 0;
  hipLaunchKernelGGL(dag_gate_fwd_k, dim3(grid_1d(B * d * d)), dim3(kBlock), 0, (hipStream_t)stream, x, A, e, ld_e,
                     imp_mode, gate_mode, h_thresh, temperature, u1, u2, seed, offset, hot, B, d);
  GNF_LAUNCH_CHECK();
  return 0;
}

int64_t gnf_dag_gate_bwd_ws_bytes(int64_t B, int64_t d) { return bwd_chunks(B, d) * d * d * (int64_t)sizeof(float); }

int gnf_dag_gate_bwd(const float* x, const float* A, const float* ge, int64_t ld_e, int imp_mode, int gate_mode,
                     float h_thresh, float temperature, const float* u1, const float* u2, uint64_t seed,
                     uint64_t offset, float* gA, float* gx, float* ws, int64_t B, int64_t d, gnf_stream_t stream) {
  if (!x || !A || !ge || B < 0 || d <= 0 || imp_mode < 0 || imp_mode > 3 || gate_mode < 0 || gate_mode > 2)
    return GNF_EINVAL;
  if (gA && !ws) return GNF_EINVAL;
  if (gate_mode == 1 && u1 && !u2) return GNF_EINVAL;
  if (imp_mode == 0) gate_mode = 0;
  hipStream_t s = (hipStream_t)stream;
  if (gA) {
    const int64_t nc = bwd_chunks(B, d);
    const int64_t chunk = B > 0 ? (B + nc - 1) / nc : 1;
    const unsigned gxd = (unsigned)((d * d + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(dag_gate_bwd_dp_k, dim3(gxd, (unsigned)nc), dim3(kBlock), 0, s, x, A, ge, ld_e, imp_mode,
                       gate_mode, h_thresh, temperature, u1, u2, seed, offset, ws, B, d, chunk);
    GNF_LAUNCH_CHECK();
    hipLaunchKernelGGL(dag_gate_bwd_dA_k, dim3(gxd), dim3(kBlock), 0, s, A, ws, gA, imp_mode, h_thresh, d, nc);
    GNF_LAUNCH_CHECK();
  }
  if (gx && B > 0) {
    hipLaunchKernelGGL(dag_gate_bwd_dx_k, dim3(grid_1d(B * d)), dim3(kBlock), 0, s, A, ge, ld_e, imp_mode, gate_mode,
                       h_thresh, temperature, u1, u2, seed, offset, gx, B, d);
    GNF_LAUNCH_CHECK();
  }
  return 0;
}

}  // extern "C"
