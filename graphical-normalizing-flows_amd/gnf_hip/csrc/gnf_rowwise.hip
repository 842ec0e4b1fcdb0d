// HBM-bound row-wise kernels: Affine normalizer (fwd/bwd/inverse) with the log-det
// row reduction fused, log|det J| and Normal log-density row reductions, column sums
// (bias gradients) and the flat Adam step.
//
// Layout: a "row" is one sample b of [B,d].  G = min(64, pow2 >= d) consecutive lanes
// own one row (64/G rows per wavefront), lanes stride over the d columns so global
// loads are coalesced, and the per-row reduction is a __shfl_xor butterfly inside the
// lane group -- no LDS, no atomics, deterministic.
#include "gnf_common.h"

namespace {

constexpr int kBlock = 256;

#define GNF_DISPATCH_GR(G_, R_, KERNEL, grid_rows, ...)                                          \
  do {                                                                                          \
    const int rpb__ = (kBlock / (G_)) * (R_);                                                   \
    const unsigned grid__ = (unsigned)(((grid_rows) + rpb__ - 1) / rpb__);                      \
    hipStream_t s__ = (hipStream_t)stream;                                                      \
    switch (G_) {                                                                               \
      case 1: hipLaunchKernelGGL((KERNEL<1, R_>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break;   \
      case 2: hipLaunchKernelGGL((KERNEL<2, R_>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break;   \
      case 4: hipLaunchKernelGGL((KERNEL<4, R_>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break;   \
      case 8: hipLaunchKernelGGL((KERNEL<8, R_>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break;   \
      case 16: hipLaunchKernelGGL((KERNEL<16, R_>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break; \
      case 32: hipLaunchKernelGGL((KERNEL<32, R_>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break; \
      default: hipLaunchKernelGGL((KERNEL<64, R_>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break; \
    }                                                                                           \
  } while (0)

#define GNF_DISPATCH_G(G_, KERNEL, grid_rows, ...)                                              \
  do {                                                                                          \
    const int rpb__ = kBlock / (G_);                                                            \
    const unsigned grid__ = (unsigned)(((grid_rows) + rpb__ - 1) / rpb__);                      \
    hipStream_t s__ = (hipStream_t)stream;                                                      \
    switch (G_) {                                                                               \
      case 1: hipLaunchKernelGGL((KERNEL<1>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break;   \
      case 2: hipLaunchKernelGGL((KERNEL<2>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break;   \
      case 4: hipLaunchKernelGGL((KERNEL<4>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break;   \
      case 8: hipLaunchKernelGGL((KERNEL<8>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break;   \
      case 16: hipLaunchKernelGGL((KERNEL<16>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break; \
      case 32: hipLaunchKernelGGL((KERNEL<32>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break; \
      default: hipLaunchKernelGGL((KERNEL<64>), dim3(grid__), dim3(kBlock), 0, s__, __VA_ARGS__); break; \
    }                                                                                           \
  } while (0)

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// ------------------------------------------------------------------ Affine normalizer
// R rows per lane group per pass: R independent load streams per lane hide the HBM latency at large B; when h is the
// contiguous [B,d,2] layout its two components are one 8-byte load / store.
template <int G, int R>
__global__ void affine_fwd_k(const float* __restrict__ x, float* __restrict__ h, int64_t h_sb, int64_t h_sd,
                             int64_t h_sc, float* __restrict__ z, float* __restrict__ jac,
                             float* __restrict__ logdet, int clamp_inplace, int64_t B, int64_t d) {
  const int64_t row0 = ((int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G) * R;
  const int g = threadIdx.x % G;
  const bool pair = (h_sc == 1 && h_sd == 2 && (h_sb & 1) == 0);
  float ld[R];
#pragma unroll
  for (int k = 0; k < R; ++k) ld[k] = 0.f;
  for (int64_t i = g; i < d; i += G) {
    float h0[R], h1[R], xv[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int64_t row = row0 + k < B ? row0 + k : B - 1;
      const int64_t hi = row * h_sb + i * h_sd;
      if (pair) {
        const float2 hv = *reinterpret_cast<const float2*>(h + hi);
        h0[k] = hv.x; h1[k] = hv.y;
      } else {
        h0[k] = h[hi]; h1[k] = h[hi + h_sc];
      }
      xv[k] = x[row * d + i];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (row0 + k >= B) continue;
      const int64_t row = row0 + k;
      const float mu = clampf(h0[k], -5.f, 5.f);
      const float ls = clampf(h1[k], -5.f, 2.f);
      const float sg = expf(ls);
      const int64_t e = row * d + i;
      z[e] = fmaf(xv[k], sg, mu);
      if (jac) jac[e] = sg;
      if (clamp_inplace) { const int64_t hi = row * h_sb + i * h_sd; h[hi] = mu; h[hi + h_sc] = ls; }
      ld[k] += ls;
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const float s = group_sum<G>(ld[k]);
    if (logdet && row0 + k < B && g == 0) logdet[row0 + k] = s;
  }
}

template <int G, int R>
__global__ void affine_bwd_k(const float* __restrict__ x, const float* __restrict__ h, int64_t h_sb, int64_t h_sd,
                             int64_t h_sc, const float* __restrict__ gz, const float* __restrict__ gjac,
                             const float* __restrict__ glogdet, float* __restrict__ gx, float* __restrict__ gh,
                             int64_t g_sb, int64_t g_sd, int64_t g_sc, int64_t B, int64_t d) {
  const int64_t row0 = ((int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G) * R;
  const int g = threadIdx.x % G;
  const bool pair = (h_sc == 1 && h_sd == 2 && (h_sb & 1) == 0);
  const bool gpair = (g_sc == 1 && g_sd == 2 && (g_sb & 1) == 0);
  for (int64_t i = g; i < d; i += G) {
    float h0[R], h1[R], xv[R], gzv[R], gjv[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int64_t row = row0 + k < B ? row0 + k : B - 1;
      const int64_t hi = row * h_sb + i * h_sd;
      if (pair) {
        const float2 hv = *reinterpret_cast<const float2*>(h + hi);
        h0[k] = hv.x; h1[k] = hv.y;
      } else {
        h0[k] = h[hi]; h1[k] = h[hi + h_sc];
      }
      const int64_t e = row * d + i;
      xv[k] = x[e];
      gzv[k] = gz ? gz[e] : 0.f;
      gjv[k] = gjac ? gjac[e] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (row0 + k >= B) continue;
      const int64_t row = row0 + k;
      const float gl = glogdet ? glogdet[row] : 0.f;
      // torch clamp backward passes the gradient where min <= v <= max (boundaries included)
      const float m0 = (h0[k] >= -5.f && h0[k] <= 5.f) ? 1.f : 0.f;
      const float m1 = (h1[k] >= -5.f && h1[k] <= 2.f) ? 1.f : 0.f;
      const float sg = expf(clampf(h1[k], -5.f, 2.f));
      const int64_t e = row * d + i;
      if (gx) gx[e] = gzv[k] * sg;
      const int64_t gi = row * g_sb + i * g_sd;
      const float o0 = gzv[k] * m0, o1 = (fmaf(gzv[k] * xv[k], sg, gjv[k] * sg) + gl) * m1;
      if (gpair) *reinterpret_cast<float2*>(gh + gi) = make_float2(o0, o1);
      else { gh[gi] = o0; gh[gi + g_sc] = o1; }
    }
  }
}

// ---- "flat" variants for short rows (d <= 64) in the contiguous [B,d,2] layout, the shape of the tabular configurations
// (cfg5: d = 63).  A wavefront owns RW consecutive rows with (RW d) % 4 == 0 (RW = 4 / gcd(d, 4)), i.e. a 16-B aligned
// span of RW d <= 256 floats: every lane moves ONE float4 of x / z and TWO of h (the row-per-lane-group kernels above
// move 4 B and 8 B per lane and reach 46-48 % of the HBM peak where a float4 copy reaches 70 %); U spans are in flight
// per wavefront.  The row sums of the log-Jacobian are RW masked wave reductions in a fixed order (deterministic).
typedef float f32x4r __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, GNF_WAVE);
  return v;
}

template <int RW, int U>
__global__ __launch_bounds__(kBlock) void affine_fwd_flat_k(const float* __restrict__ x, const float* __restrict__ h,
                                                            float* __restrict__ z, float* __restrict__ jac,
                                                            float* __restrict__ logdet, int64_t B, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (kBlock / 64);
  const int64_t nspan = (B + RW - 1) / RW;
  int rid[4];                                        // row (inside the span) of this lane's four elements
#pragma unroll
  for (int c = 0; c < 4; ++c) rid[c] = (4 * lane + c) / d;
  for (int64_t sp0 = gw * U; sp0 < nspan; sp0 += nw * U) {
    f32x4r xv[U], ha[U], hb[U];
    int ne[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t sp = sp0 + u, row0 = sp * RW;
      const int nrows = sp < nspan ? (int)(B - row0 < RW ? B - row0 : RW) : 0;
      ne[u] = nrows * d;
      const int64_t base = row0 * d + 4 * lane;
      if (4 * lane + 3 < ne[u]) {
        xv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(x + base));
        ha[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(h + 2 * base));
        hb[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(h + 2 * base + 4));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool ok = 4 * lane + c < ne[u];
          xv[u][c] = ok ? x[base + c] : 0.f;
          const float a0 = ok ? h[2 * (base + c)] : 0.f, a1 = ok ? h[2 * (base + c) + 1] : 0.f;
          if (c < 2) { ha[u][2 * c] = a0; ha[u][2 * c + 1] = a1; } else { hb[u][2 * c - 4] = a0; hb[u][2 * c - 3] = a1; }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ne[u] == 0) continue;
      const int64_t row0 = (sp0 + u) * RW, base = row0 * d + 4 * lane;
      const float h0[4] = {ha[u][0], ha[u][2], hb[u][0], hb[u][2]}, h1[4] = {ha[u][1], ha[u][3], hb[u][1], hb[u][3]};
      f32x4r zv, jv;
      float ls[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float mu = clampf(h0[c], -5.f, 5.f);
        ls[c] = clampf(h1[c], -5.f, 2.f);
        const float sg = expf(ls[c]);
        zv[c] = fmaf(xv[u][c], sg, mu);
        jv[c] = sg;
        if (4 * lane + c >= ne[u]) ls[c] = 0.f;
      }
      if (4 * lane + 3 < ne[u]) {
        __builtin_nontemporal_store(zv, reinterpret_cast<f32x4r*>(z + base));
        if (jac) __builtin_nontemporal_store(jv, reinterpret_cast<f32x4r*>(jac + base));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (4 * lane + c < ne[u]) { z[base + c] = zv[c]; if (jac) jac[base + c] = jv[c]; }
      }
      if (logdet) {
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
          float sv = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) sv += rid[c] == rr ? ls[c] : 0.f;
          sv = wave_sum(sv);
          if (lane == 0 && row0 + rr < B) logdet[row0 + rr] = sv;
        }
      }
    }
  }
}

template <int RW, int U>
__global__ __launch_bounds__(kBlock) void affine_bwd_flat_k(const float* __restrict__ x, const float* __restrict__ h,
                                                            const float* __restrict__ gz, const float* __restrict__ gjac,
                                                            const float* __restrict__ glogdet, float* __restrict__ gx,
                                                            float* __restrict__ gh, int64_t B, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (kBlock / 64);
  const int64_t nspan = (B + RW - 1) / RW;
  int rid[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) rid[c] = (4 * lane + c) / d;
  for (int64_t sp0 = gw * U; sp0 < nspan; sp0 += nw * U) {
    f32x4r xv[U], ha[U], hb[U], gv[U], jv[U];
    int ne[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t sp = sp0 + u, row0 = sp * RW;
      const int nrows = sp < nspan ? (int)(B - row0 < RW ? B - row0 : RW) : 0;
      ne[u] = nrows * d;
      const int64_t base = row0 * d + 4 * lane;
      gv[u] = f32x4r{0.f, 0.f, 0.f, 0.f};
      jv[u] = gv[u];
      if (4 * lane + 3 < ne[u]) {
        xv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(x + base));
        ha[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(h + 2 * base));
        hb[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(h + 2 * base + 4));
        if (gz) gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(gz + base));
        if (gjac) jv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(gjac + base));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool ok = 4 * lane + c < ne[u];
          xv[u][c] = ok ? x[base + c] : 0.f;
          const float a0 = ok ? h[2 * (base + c)] : 0.f, a1 = ok ? h[2 * (base + c) + 1] : 0.f;
          if (c < 2) { ha[u][2 * c] = a0; ha[u][2 * c + 1] = a1; } else { hb[u][2 * c - 4] = a0; hb[u][2 * c - 3] = a1; }
          if (ok && gz) gv[u][c] = gz[base + c];
          if (ok && gjac) jv[u][c] = gjac[base + c];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ne[u] == 0) continue;
      const int64_t row0 = (sp0 + u) * RW, base = row0 * d + 4 * lane;
      const float h0[4] = {ha[u][0], ha[u][2], hb[u][0], hb[u][2]}, h1[4] = {ha[u][1], ha[u][3], hb[u][1], hb[u][3]};
      f32x4r ox, oa, ob;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool ok = 4 * lane + c < ne[u];
        const float gl = (glogdet && ok) ? glogdet[row0 + rid[c]] : 0.f;
        // torch clamp backward passes the gradient where min <= v <= max (boundaries included)
        const float m0 = (h0[c] >= -5.f && h0[c] <= 5.f) ? 1.f : 0.f;
        const float m1 = (h1[c] >= -5.f && h1[c] <= 2.f) ? 1.f : 0.f;
        const float sg = expf(clampf(h1[c], -5.f, 2.f));
        ox[c] = gv[u][c] * sg;
        const float o0 = gv[u][c] * m0, o1 = (fmaf(gv[u][c] * xv[u][c], sg, jv[u][c] * sg) + gl) * m1;
        if (c < 2) { oa[2 * c] = o0; oa[2 * c + 1] = o1; } else { ob[2 * c - 4] = o0; ob[2 * c - 3] = o1; }
      }
      if (4 * lane + 3 < ne[u]) {
        if (gx) __builtin_nontemporal_store(ox, reinterpret_cast<f32x4r*>(gx + base));
        __builtin_nontemporal_store(oa, reinterpret_cast<f32x4r*>(gh + 2 * base));
        __builtin_nontemporal_store(ob, reinterpret_cast<f32x4r*>(gh + 2 * base + 4));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (4 * lane + c < ne[u]) {
            if (gx) gx[base + c] = ox[c];
            gh[2 * (base + c)] = c < 2 ? oa[2 * c] : ob[2 * c - 4];
            gh[2 * (base + c) + 1] = c < 2 ? oa[2 * c + 1] : ob[2 * c - 3];
          }
      }
    }
  }
}

// rows per wavefront span of the flat variants, 0 when they do not apply
inline int affine_flat_rw(int64_t h_sb, int64_t h_sd, int64_t h_sc, int64_t B, int64_t d, const void* a, const void* b,
                          const void* c) {
  if (!(h_sc == 1 && h_sd == 2 && h_sb == 2 * d) || d > 64 || B * d < (1 << 18)) return 0;
  if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) != 0) return 0;
  const int rw = (d % 4 == 0) ? 1 : ((d % 2 == 0) ? 2 : 4);
  return rw * d <= 256 ? rw : 0;
}

__global__ void affine_inv_k(const float* __restrict__ z, const float* __restrict__ h, int64_t h_sb, int64_t h_sd,
                             int64_t h_sc, float* __restrict__ x, int64_t B, int64_t d) {
  const int64_t n = B * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / d, i = e - b * d;
    const int64_t hi = b * h_sb + i * h_sd;
    const float mu = clampf(h[hi], -5.f, 5.f);
    const float sg = expf(clampf(h[hi + h_sc], -5.f, 2.f));
    x[e] = (z[e] - mu) / sg;
  }
}

// ------------------------------------------------------------------ row reductions
template <int G>
__global__ void logsum_rows_k(const float* __restrict__ jac, float* __restrict__ out, int64_t B, int64_t d) {
  const int64_t row = (int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G;
  const int g = threadIdx.x % G;
  float s = 0.f;
  if (row < B)
    for (int64_t i = g; i < d; i += G) s += logf(jac[row * d + i]);
  s = group_sum<G>(s);
  if (row < B && g == 0) out[row] = s;
}

template <int G>
__global__ void normal_ld_rows_k(const float* __restrict__ z, float* __restrict__ out, int64_t B, int64_t d) {
  const int64_t row = (int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G;
  const int g = threadIdx.x % G;
  float s = 0.f;
  if (row < B)
    for (int64_t i = g; i < d; i += G) {
      const float v = z[row * d + i];
      s += 1.8378770664093453f + v * v;  // log(2 pi) + z^2, summed like the reference does
    }
  s = group_sum<G>(s);
  if (row < B && g == 0) out[row] = -0.5f * s;
}

__global__ void logsum_rows_bwd_k(const float* __restrict__ jac, const float* __restrict__ g, float* __restrict__ gj,
                                  int64_t B, int64_t d) {
  const int64_t n = B * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    gj[e] = g[e / d] / jac[e];
}

__global__ void normal_ld_bwd_k(const float* __restrict__ z, const float* __restrict__ g, float* __restrict__ gz,
                                int64_t B, int64_t d) {
  const int64_t n = B * d;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    gz[e] = -z[e] * g[e / d];
}

// ------------------------------------------------------------------ column sums
// stage 1: block (bx, by) sums rows [by*R, by*R+R) of column tile bx into ws[by][n];
// stage 2: sums the gridDim.y partials.  Fixed order -> bit-reproducible.
constexpr int kColRows = 512;
__global__ void colsum_stage1_k(const float* __restrict__ a, int64_t lda, float* __restrict__ ws, int64_t M,
                                int64_t N) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int64_t m0 = (int64_t)blockIdx.y * kColRows;
  const int64_t m1 = m0 + kColRows < M ? m0 + kColRows : M;
  float s = 0.f;
  for (int64_t m = m0; m < m1; ++m) s += a[m * lda + n];
  ws[(int64_t)blockIdx.y * N + n] = s;
}
__global__ void colsum_stage2_k(const float* __restrict__ ws, float* __restrict__ out, int64_t P, int64_t N) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int64_t p = 0; p < P; ++p) s += ws[p * N + n];
  out[n] = s;
}

// ------------------------------------------------------------------ Adam (L2 decay, bias-corrected)
// Hyper-parameters arrive as doubles and every derived constant (1-beta, lr / bias-correction, sqrt of the second
// bias-correction) is formed in double and rounded ONCE, as torch.optim.Adam does on the host: 1.f - 0.999f is
// 1.3e-5 away from (float)(1 - 0.999), which shows in exp_avg_sq after a single step.
struct AdamC { float b1, b2, omb1, omb2, eps, wd, gscale; };

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamC& c, float step_size,
                                          float bc2_sqrt) {
  const float gi = fmaf(c.wd, p, g * c.gscale);
  m = fmaf(c.b1, m, c.omb1 * gi);
  v = fmaf(c.b2, v, c.omb2 * gi * gi);
  // torch.optim.Adam: denom = sqrt(v)/sqrt(bc2) + eps; p -= lr/bc1 * m/denom
  p = p - step_size * (m / (sqrtf(v) / bc2_sqrt + c.eps));
}

// n4 float4 groups + a scalar tail; 16-B alignment of the four buffers is checked by the launcher
__device__ __forceinline__ void adam_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                          float* __restrict__ v, int64_t n, bool vec, const AdamC& c, float step_size,
                                          float bc2_sqrt) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = vec ? n >> 2 : 0;
  for (int64_t i = tid; i < n4; i += nth) {
    float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    adam_elem(pp.x, gg.x, mm.x, vv.x, c, step_size, bc2_sqrt);
    adam_elem(pp.y, gg.y, mm.y, vv.y, c, step_size, bc2_sqrt);
    adam_elem(pp.z, gg.z, mm.z, vv.z, c, step_size, bc2_sqrt);
    adam_elem(pp.w, gg.w, mm.w, vv.w, c, step_size, bc2_sqrt);
    reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (int64_t i = 4 * n4 + tid; i < n; i += nth) adam_elem(p[i], g[i], m[i], v[i], c, step_size, bc2_sqrt);
}

__global__ void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, int64_t n, int vec, AdamC c, float step_size, float bc2_sqrt) {
  adam_body(p, g, m, v, n, vec != 0, c, step_size, bc2_sqrt);
}

// Same update with the step count in device memory, so that a captured hipGraph of the whole training step can be
// replayed: the bias corrections are recomputed on the device from *step_dev + 1; adam_bump_k then advances it.
__global__ void adam_dev_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                           float* __restrict__ v, int64_t n, int vec, AdamC c, double lr, double b1, double b2,
                           const int* __restrict__ step_dev) {
  const double t = (double)(*step_dev + 1);
  const float step_size = (float)(lr / (1.0 - pow(b1, t))), bc2_sqrt = (float)sqrt(1.0 - pow(b2, t));
  adam_body(p, g, m, v, n, vec != 0, c, step_size, bc2_sqrt);
}
__global__ void adam_bump_k(int* step_dev) { *step_dev += 1; }

// out[n] (+)= sum_p src[p*N + n]: 64 columns per block, 16 wavefronts stride over the rows
// (coalesced 256-B row segments), fixed-order LDS tree across the wavefronts -> deterministic.
// blockIdx.y selects a chunk of `rpc` rows (tall inputs: first level of a two-level reduction); the chunk's
// result goes to out + blockIdx.y*N.
__global__ __launch_bounds__(1024) void rowsum_k(const float* __restrict__ src, float* __restrict__ out, int64_t P,
                                                 int64_t N, int accumulate, int64_t rpc) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * 64 + lane;
  src += (int64_t)blockIdx.y * rpc * N;
  out += (int64_t)blockIdx.y * N;
  {
    const int64_t left = P - (int64_t)blockIdx.y * rpc;
    P = left < rpc ? left : rpc;
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (n < N) {
    int64_t p = wave;
    for (; p + 48 < P; p += 64) {
      s0 += src[p * N + n]; s1 += src[(p + 16) * N + n]; s2 += src[(p + 32) * N + n]; s3 += src[(p + 48) * N + n];
    }
    for (; p < P; p += 16) s0 += src[p * N + n];
  }
  red[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && n < N) {
    float s = accumulate ? out[n] : 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w][lane];
    out[n] = s;
  }
}

inline unsigned grid_1d(int64_t n) {
  int64_t g = (n + kBlock - 1) / kBlock;
  if (g > 256 * 8) g = 256 * 8;   // 8 blocks per CU, grid-stride the rest
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

int gnf_rowsum_launch(const float* src, float* out, int64_t P, int64_t N, int accumulate, hipStream_t s) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(rowsum_k, dim3((unsigned)((N + 63) / 64), 1), dim3(1024), 0, s, src, out, P, N, accumulate,
                     P > 0 ? P : 1);
  GNF_LAUNCH_CHECK();
  return 0;
}

// tall inputs (P >> N): kRowsumChunks partial rows in ws (kRowsumChunks*N floats), then the plain row-sum
int gnf_rowsum_tall_launch(const float* src, float* out, int64_t P, int64_t N, int accumulate, float* ws,
                           hipStream_t s) {
  if (N <= 0) return 0;
  if (P <= 8192 || !ws) return gnf_rowsum_launch(src, out, P, N, accumulate, s);
  const int64_t rpc = (P + kRowsumChunks - 1) / kRowsumChunks;
  const int64_t nch = (P + rpc - 1) / rpc;
  hipLaunchKernelGGL(rowsum_k, dim3((unsigned)((N + 63) / 64), (unsigned)nch), dim3(1024), 0, s, src, ws, P, N, 0, rpc);
  GNF_LAUNCH_CHECK();
  return gnf_rowsum_launch(ws, out, nch, N, accumulate, s);
}

extern "C" {

int gnf_abi_version(void) { return GNF_ABI_VERSION; }

int gnf_affine_fwd(const float* x, float* h, int64_t h_sb, int64_t h_sd, int64_t h_sc, float* z, float* jac,
                   float* logdet, int clamp_inplace, int64_t B, int64_t d, gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;                    // batch-sized arrays may be NULL for an empty batch
  if (!x || !h || !z) return GNF_EINVAL;
  const int G = gnf_pow2_ge(d, 64);
  if (const int rw = clamp_inplace ? 0 : affine_flat_rw(h_sb, h_sd, h_sc, B, d, x, h, z)) {
    if (!jac || ((uintptr_t)jac & 15) == 0) {
      const int64_t nspan = (B + rw - 1) / rw;
      int64_t grid = (nspan + 2 * (kBlock / 64) - 1) / (2 * (kBlock / 64));
      if (grid > 256 * 16) grid = 256 * 16;
      hipStream_t s = (hipStream_t)stream;
      if (rw == 4) hipLaunchKernelGGL((affine_fwd_flat_k<4, 2>), dim3((unsigned)grid), dim3(kBlock), 0, s, x, h, z, jac, logdet, B, (int)d);
      else if (rw == 2) hipLaunchKernelGGL((affine_fwd_flat_k<2, 2>), dim3((unsigned)grid), dim3(kBlock), 0, s, x, h, z, jac, logdet, B, (int)d);
      else hipLaunchKernelGGL((affine_fwd_flat_k<1, 2>), dim3((unsigned)grid), dim3(kBlock), 0, s, x, h, z, jac, logdet, B, (int)d);
      GNF_LAUNCH_CHECK();
      return 0;
    }
  }
  if (B * d >= (1 << 20)) GNF_DISPATCH_GR(G, 4, affine_fwd_k, B, x, h, h_sb, h_sd, h_sc, z, jac, logdet, clamp_inplace, B, d);
  else GNF_DISPATCH_GR(G, 1, affine_fwd_k, B, x, h, h_sb, h_sd, h_sc, z, jac, logdet, clamp_inplace, B, d);
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_affine_bwd(const float* x, const float* h, int64_t h_sb, int64_t h_sd, int64_t h_sc, const float* gz,
                   const float* gjac, const float* glogdet, float* gx, float* gh, int64_t g_sb, int64_t g_sd,
                   int64_t g_sc, int64_t B, int64_t d, gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;
  if (!x || !h || !gh) return GNF_EINVAL;
  const int G = gnf_pow2_ge(d, 64);
  if (const int rw = (g_sc == 1 && g_sd == 2 && g_sb == 2 * d) ? affine_flat_rw(h_sb, h_sd, h_sc, B, d, x, h, gh) : 0) {
    if ((((uintptr_t)gz | (uintptr_t)gjac | (uintptr_t)gx) & 15) == 0) {
      const int64_t nspan = (B + rw - 1) / rw;
      int64_t grid = (nspan + 2 * (kBlock / 64) - 1) / (2 * (kBlock / 64));
      if (grid > 256 * 16) grid = 256 * 16;
      hipStream_t s = (hipStream_t)stream;
      if (rw == 4) hipLaunchKernelGGL((affine_bwd_flat_k<4, 2>), dim3((unsigned)grid), dim3(kBlock), 0, s, x, h, gz, gjac, glogdet, gx, gh, B, (int)d);
      else if (rw == 2) hipLaunchKernelGGL((affine_bwd_flat_k<2, 2>), dim3((unsigned)grid), dim3(kBlock), 0, s, x, h, gz, gjac, glogdet, gx, gh, B, (int)d);
      else hipLaunchKernelGGL((affine_bwd_flat_k<1, 2>), dim3((unsigned)grid), dim3(kBlock), 0, s, x, h, gz, gjac, glogdet, gx, gh, B, (int)d);
      GNF_LAUNCH_CHECK();
      return 0;
    }
  }
  if (B * d >= (1 << 20)) GNF_DISPATCH_GR(G, 4, affine_bwd_k, B, x, h, h_sb, h_sd, h_sc, gz, gjac, glogdet, gx, gh, g_sb, g_sd, g_sc, B, d);
  else GNF_DISPATCH_GR(G, 1, affine_bwd_k, B, x, h, h_sb, h_sd, h_sc, gz, gjac, glogdet, gx, gh, g_sb, g_sd, g_sc, B, d);
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_affine_inv(const float* z, const float* h, int64_t h_sb, int64_t h_sd, int64_t h_sc, float* x, int64_t B,
                   int64_t d, gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;
  if (!z || !h || !x) return GNF_EINVAL;
  hipLaunchKernelGGL(affine_inv_k, dim3(grid_1d(B * d)), dim3(kBlock), 0, (hipStream_t)stream, z, h, h_sb, h_sd,
                     h_sc, x, B, d);
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_logsum_rows_fwd(const float* jac, float* out, int64_t B, int64_t d, gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;
  if (!jac || !out) return GNF_EINVAL;
  const int G = gnf_pow2_ge(d, 64);
  GNF_DISPATCH_G(G, logsum_rows_k, B, jac, out, B, d);
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_logsum_rows_bwd(const float* jac, const float* g, float* gjac, int64_t B, int64_t d, gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;
  if (!jac || !g || !gjac) return GNF_EINVAL;
  hipLaunchKernelGGL(logsum_rows_bwd_k, dim3(grid_1d(B * d)), dim3(kBlock), 0, (hipStream_t)stream, jac, g, gjac, B,
                     d);
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_normal_logdensity_fwd(const float* z, float* out, int64_t B, int64_t d, gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;
  if (!z || !out) return GNF_EINVAL;
  const int G = gnf_pow2_ge(d, 64);
  GNF_DISPATCH_G(G, normal_ld_rows_k, B, z, out, B, d);
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_normal_logdensity_bwd(const float* z, const float* g, float* gz, int64_t B, int64_t d, gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;
  if (!z || !g || !gz) return GNF_EINVAL;
  hipLaunchKernelGGL(normal_ld_bwd_k, dim3(grid_1d(B * d)), dim3(kBlock), 0, (hipStream_t)stream, z, g, gz, B, d);
  GNF_LAUNCH_CHECK();
  return 0;
}

int64_t gnf_colsum_ws_bytes(int64_t M, int64_t N) {
  (void)M;
  return (int64_t)kRowsumChunks * N * (int64_t)sizeof(float);
}

int gnf_colsum(const float* a, int64_t lda, float* out, int64_t M, int64_t N, float* ws, gnf_stream_t stream) {
  if ((!a && M > 0) || !out || !ws || M < 0 || N <= 0) return GNF_EINVAL;   // M == 0: out = 0
  if (lda == N) return gnf_rowsum_tall_launch(a, out, M, N, 0, ws, (hipStream_t)stream);
  // strided rows: generic two-stage kernels
  const int64_t P = (M + kColRows - 1) / kColRows;
  if (P > kRowsumChunks) return GNF_ESHAPE;
  const unsigned gx = (unsigned)((N + kBlock - 1) / kBlock);
  if (P > 0) {
    hipLaunchKernelGGL(colsum_stage1_k, dim3(gx, (unsigned)P), dim3(kBlock), 0, (hipStream_t)stream, a, lda, ws, M, N);
    GNF_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(colsum_stage2_k, dim3(gx), dim3(kBlock), 0, (hipStream_t)stream, ws, out, P, N);
  GNF_LAUNCH_CHECK();
  return 0;
}

static AdamC adam_consts(double beta1, double beta2, double eps, double weight_decay, double grad_scale) {
  AdamC c;
  c.b1 = (float)beta1; c.b2 = (float)beta2; c.omb1 = (float)(1.0 - beta1); c.omb2 = (float)(1.0 - beta2);
  c.eps = (float)eps; c.wd = (float)weight_decay; c.gscale = (float)grad_scale;
  return c;
}
static int adam_vec_ok(const float* p, const float* g, const float* m, const float* v) {
  return ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15u) == 0) ? 1 : 0;
}

int gnf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                  double eps, double weight_decay, double grad_scale, int step, gnf_stream_t stream) {
  if (n < 0 || step < 1) return GNF_EINVAL;
  if (n == 0) return 0;
  if (!p || !g || !m || !v) return GNF_EINVAL;
  const float step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
  const float bc2s = (float)sqrt(1.0 - pow(beta2, (double)step));
  const int vec = adam_vec_ok(p, g, m, v);
  hipLaunchKernelGGL(adam_k, dim3(grid_1d(vec ? (n + 3) / 4 : n)), dim3(kBlock), 0, (hipStream_t)stream, p, g, m, v, n,
                     vec, adam_consts(beta1, beta2, eps, weight_decay, grad_scale), step_size, bc2s);
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                      double eps, double weight_decay, double grad_scale, int* step_dev, int advance,
                      gnf_stream_t stream) {
  if (!step_dev || n < 0) return GNF_EINVAL;
  if (n == 0) return 0;
  if (!p || !g || !m || !v) return GNF_EINVAL;
  const int vec = adam_vec_ok(p, g, m, v);
  hipLaunchKernelGGL(adam_dev_k, dim3(grid_1d(vec ? (n + 3) / 4 : n)), dim3(kBlock), 0, (hipStream_t)stream, p, g, m, v,
                     n, vec, adam_consts(beta1, beta2, eps, weight_decay, grad_scale), lr, beta1, beta2,
                     (const int*)step_dev);
  if (advance) hipLaunchKernelGGL(adam_bump_k, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
  GNF_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
