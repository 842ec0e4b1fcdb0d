// HBM-bound row-wise kernels: Affine normalizer (fwd/bwd/inverse) with the log-det
// row reduction fused, log|det J| and Normal log-density row reductions, column sums
// (bias gradients) and the flat Adam step.
//
// Layout: a "row" is one sample b of [B,d].  G = min(64, pow2 >= d) consecutive lanes
// own one row (64/G rows per wavefront), lanes stride over the d columns so global
// loads are coalesced, and the per-row reduction is a __shfl_xor butterfly inside the
// lane group -- no LDS, no atomics, deterministic.
#include "gnf_common.h"
#include <stdlib.h>

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
constexpr float kLog2Pi = 1.8378770664093453f;     // log(2 pi): summed per element with z^2, like the reference does

// ------------------------------------------------------------------ Affine normalizer
// R rows per lane group per pass: R independent load streams per lane hide the HBM latency at large B; when h is the
// contiguous [B,d,2] layout its two components are one 8-byte load / store.
// (CLAMP: the reference's in-place clamp_ of h is a separate instantiation, so that the common one never stores to h and the
// loads of the unrolled column loop can all be issued ahead of the arithmetic -- at B = 100, d = 784 the kernel is a chain of
// d / G = 13 dependent load rounds otherwise)
template <int G, int R, bool CLAMP>
__device__ __forceinline__ void affine_fwd_body(const float* __restrict__ x, float* __restrict__ h, int64_t h_sb, int64_t h_sd,
                             int64_t h_sc, float* __restrict__ z, float* __restrict__ jac,
                             float* __restrict__ logdet, float* __restrict__ logn, int clamp_inplace, int64_t B,
                             int64_t d) {
  const int64_t row0 = ((int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G) * R;
  const int g = threadIdx.x % G;
  const bool pair = (h_sc == 1 && h_sd == 2 && (h_sb & 1) == 0);
  float ld[R], ln[R];
#pragma unroll
  for (int k = 0; k < R; ++k) { ld[k] = 0.f; ln[k] = 0.f; }
#pragma unroll 4
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
      const float zv = fmaf(xv[k], sg, mu);
      z[e] = zv;
      if (jac) jac[e] = sg;
      if (CLAMP && clamp_inplace) { const int64_t hi = row * h_sb + i * h_sd; h[hi] = mu; h[hi + h_sc] = ls; }
      ld[k] += ls;
      ln[k] += kLog2Pi + zv * zv;
    }
  }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const float s = group_sum<G>(ld[k]);
    if (logdet && row0 + k < B && g == 0) logdet[row0 + k] = s;
    if (logn) {                                      // Normal log-density of the row (NormalizingFlowFactories.py:15-16)
      const float t = group_sum<G>(ln[k]);
      if (row0 + k < B && g == 0) logn[row0 + k] = -0.5f * t;
    }
  }
}

template <int G, int R>
__global__ void affine_fwd_plain_k(const float* __restrict__ x, float* __restrict__ h, int64_t h_sb, int64_t h_sd, int64_t h_sc,
                                   float* __restrict__ z, float* __restrict__ jac, float* __restrict__ logdet,
                                   float* __restrict__ logn, int clamp_inplace, int64_t B, int64_t d) {
  affine_fwd_body<G, R, false>(x, h, h_sb, h_sd, h_sc, z, jac, logdet, logn, clamp_inplace, B, d);
}
template <int G, int R>
__global__ void affine_fwd_clamp_k(const float* __restrict__ x, float* __restrict__ h, int64_t h_sb, int64_t h_sd, int64_t h_sc,
                                   float* __restrict__ z, float* __restrict__ jac, float* __restrict__ logdet,
                                   float* __restrict__ logn, int clamp_inplace, int64_t B, int64_t d) {
  affine_fwd_body<G, R, true>(x, h, h_sb, h_sd, h_sc, z, jac, logdet, logn, clamp_inplace, B, d);
}

template <int G, int R>
__global__ void affine_bwd_k(const float* __restrict__ x, const float* __restrict__ h, int64_t h_sb, int64_t h_sd,
                             int64_t h_sc, const float* __restrict__ gz, const float* __restrict__ gjac,
                             const float* __restrict__ glogdet, const float* __restrict__ glogn,
                             float* __restrict__ gx, float* __restrict__ gh, int64_t g_sb, int64_t g_sd, int64_t g_sc,
                             int64_t B, int64_t d) {
  const int64_t row0 = ((int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G) * R;
  const int g = threadIdx.x % G;
  const bool pair = (h_sc == 1 && h_sd == 2 && (h_sb & 1) == 0);
  const bool gpair = (g_sc == 1 && g_sd == 2 && (g_sb & 1) == 0);
#pragma unroll 4
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
      // cotangent of z: what came in, plus the Normal log-density's -z * g when that reduction was fused into the forward
      const float gze = glogn ? fmaf(-fmaf(xv[k], sg, clampf(h0[k], -5.f, 5.f)), glogn[row], gzv[k]) : gzv[k];
      if (gx) gx[e] = gze * sg;
      const int64_t gi = row * g_sb + i * g_sd;
      const float o0 = gze * m0, o1 = (fmaf(gze * xv[k], sg, gjv[k] * sg) + gl) * m1;
      if (gpair) *reinterpret_cast<float2*>(gh + gi) = make_float2(o0, o1);
      else { gh[gi] = o0; gh[gi + g_sc] = o1; }
    }
  }
}

// ---- FEW LONG rows (the image configurations: B = 100, d = 784): the row-per-lane-group kernels above put four rows on a
// workgroup, i.e. 25 workgroups walking 13 dependent column rounds each (9-11 us for 1.2 MB).  Here a row gets a whole
// workgroup in the forward (4 column rounds; the two row sums meet in LDS) and the backward, which has no reduction, is one
// thread per element.  Any strides of h / gh.
__global__ __launch_bounds__(kBlock) void affine_fwd_rowblock_k(const float* __restrict__ x, const float* __restrict__ h,
                                                                int64_t h_sb, int64_t h_sd, int64_t h_sc,
                                                                float* __restrict__ z, float* __restrict__ jac,
                                                                float* __restrict__ logdet, float* __restrict__ logn, int d) {
  __shared__ float red[2][kBlock / 64];
  const int row = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sl = 0.f, sn = 0.f;
#pragma unroll 4
  for (int i = threadIdx.x; i < d; i += kBlock) {
    const int64_t hi = row * h_sb + i * h_sd, e = (int64_t)row * d + i;
    const float mu = clampf(h[hi], -5.f, 5.f), ls = clampf(h[hi + h_sc], -5.f, 2.f);
    const float sg = expf(ls), zv = fmaf(x[e], sg, mu);
    z[e] = zv;
    if (jac) jac[e] = sg;
    sl += ls;
    sn += kLog2Pi + zv * zv;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sl += __shfl_xor(sl, off, GNF_WAVE); sn += __shfl_xor(sn, off, GNF_WAVE); }
  if (lane == 0) { red[0][wave] = sl; red[1][wave] = sn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) { a += red[0][w]; b += red[1][w]; }
    if (logdet) logdet[row] = a;
    if (logn) logn[row] = -0.5f * b;
  }
}

__global__ __launch_bounds__(kBlock) void affine_bwd_elem_k(const float* __restrict__ x, const float* __restrict__ h, int64_t h_sb,
                                                            int64_t h_sd, int64_t h_sc, const float* __restrict__ gz,
                                                            const float* __restrict__ gjac, const float* __restrict__ glogdet,
                                                            const float* __restrict__ glogn, float* __restrict__ gx,
                                                            float* __restrict__ gh, int64_t g_sb, int64_t g_sd, int64_t g_sc,
                                                            int n, int d) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= n) return;
  const int row = e / d, i = e - row * d;
  const int64_t hi = row * h_sb + i * h_sd;
  const float h0 = h[hi], h1 = h[hi + h_sc];
  // torch clamp backward passes the gradient where min <= v <= max (boundaries included)
  const float m0 = (h0 >= -5.f && h0 <= 5.f) ? 1.f : 0.f, m1 = (h1 >= -5.f && h1 <= 2.f) ? 1.f : 0.f;
  const float sg = expf(clampf(h1, -5.f, 2.f)), xv = x[e];
  const float gzv = gz ? gz[e] : 0.f, gjv = gjac ? gjac[e] : 0.f;
  const float gze = glogn ? fmaf(-fmaf(xv, sg, clampf(h0, -5.f, 5.f)), glogn[row], gzv) : gzv;
  if (gx) gx[e] = gze * sg;
  const int64_t gi = row * g_sb + i * g_sd;
  gh[gi] = gze * m0;
  gh[gi + g_sc] = (fmaf(gze * xv, sg, gjv * sg) + (glogdet ? glogdet[row] : 0.f)) * m1;
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
                                                            float* __restrict__ logdet, float* __restrict__ logn,
                                                            int64_t B, int d) {
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
      float ls[4], lq[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float mu = clampf(h0[c], -5.f, 5.f);
        ls[c] = clampf(h1[c], -5.f, 2.f);
        const float sg = expf(ls[c]);
        zv[c] = fmaf(xv[u][c], sg, mu);
        jv[c] = sg;
        lq[c] = kLog2Pi + zv[c] * zv[c];
        if (4 * lane + c >= ne[u]) { ls[c] = 0.f; lq[c] = 0.f; }
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
      if (logn) {
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
          float sv = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) sv += rid[c] == rr ? lq[c] : 0.f;
          sv = wave_sum(sv);
          if (lane == 0 && row0 + rr < B) logn[row0 + rr] = -0.5f * sv;
        }
      }
    }
  }
}

template <int RW, int U>
__global__ __launch_bounds__(kBlock) void affine_bwd_flat_k(const float* __restrict__ x, const float* __restrict__ h,
                                                            const float* __restrict__ gz, const float* __restrict__ gjac,
                                                            const float* __restrict__ glogdet,
                                                            const float* __restrict__ glogn, float* __restrict__ gx,
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
        const float gn = (glogn && ok) ? glogn[row0 + rid[c]] : 0.f;
        const float gze = fmaf(-fmaf(xv[u][c], sg, clampf(h0[c], -5.f, 5.f)), gn, gv[u][c]);   // + d logN / dz = -z
        ox[c] = gze * sg;
        const float o0 = gze * m0, o1 = (fmaf(gze * xv[u][c], sg, jv[u][c] * sg) + gl) * m1;
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
// log|det J| = sum_d log jac (NormalizingFlow.py:70) and the Normal log-density -1/2 sum_d (log 2 pi + z^2)
// (NormalizingFlowFactories.py:15-16) of a row in ONE pass over z and jac; either output may be absent (jac == NULL: only
// the density; logn == NULL: only the log-determinant).  G lanes per row.
template <int G>
__global__ void nll_rows_k(const float* __restrict__ z, const float* __restrict__ jac, float* __restrict__ logdet,
                           float* __restrict__ logn, int64_t B, int64_t d) {
  const int64_t row = (int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G;
  const int g = threadIdx.x % G;
  float sl = 0.f, sn = 0.f;
  if (row < B)
#pragma unroll 4
    for (int64_t i = g; i < d; i += G) {
      if (jac) sl += logf(jac[row * d + i]);
      if (logn) { const float v = z[row * d + i]; sn += kLog2Pi + v * v; }
    }
  if (jac) { sl = group_sum<G>(sl); if (row < B && g == 0) logdet[row] = sl; }
  if (logn) { sn = group_sum<G>(sn); if (row < B && g == 0) logn[row] = -0.5f * sn; }
}

// backward of both: gz = gz_in - z * glogn[row], gjac = glogdet[row] / jac (no integer division per element: rows are
// dealt to lane groups as in the forward)
template <int G>
__global__ void nll_rows_bwd_k(const float* __restrict__ z, const float* __restrict__ jac, const float* __restrict__ glogdet,
                               const float* __restrict__ glogn, const float* __restrict__ gz_in, float* __restrict__ gz,
                               float* __restrict__ gjac, int64_t B, int64_t d) {
  const int64_t row = (int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G;
  const int g = threadIdx.x % G;
  if (row >= B) return;
  const float gl = glogdet ? glogdet[row] : 0.f, gn = glogn ? glogn[row] : 0.f;
#pragma unroll 4
  for (int64_t i = g; i < d; i += G) {
    const int64_t e = row * d + i;
    if (gz) gz[e] = fmaf(-z[e], gn, gz_in ? gz_in[e] : 0.f);
    if (gjac) gjac[e] = gl / jac[e];
  }
}

// Few long rows (B = 100, d = 784): a workgroup per row in the forward, a thread per element in the backward (see the Affine
// kernels of the same names)
__global__ __launch_bounds__(kBlock) void nll_rows_rowblock_k(const float* __restrict__ z, const float* __restrict__ jac,
                                                              float* __restrict__ logdet, float* __restrict__ logn, int d) {
  __shared__ float red[2][kBlock / 64];
  const int row = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sl = 0.f, sn = 0.f;
#pragma unroll 4
  for (int i = threadIdx.x; i < d; i += kBlock) {
    const int64_t e = (int64_t)row * d + i;
    if (jac) sl += logf(jac[e]);
    if (logn) { const float v = z[e]; sn += kLog2Pi + v * v; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sl += __shfl_xor(sl, off, GNF_WAVE); sn += __shfl_xor(sn, off, GNF_WAVE); }
  if (lane == 0) { red[0][wave] = sl; red[1][wave] = sn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) { a += red[0][w]; b += red[1][w]; }
    if (jac) logdet[row] = a;
    if (logn) logn[row] = -0.5f * b;
  }
}

__global__ __launch_bounds__(kBlock) void nll_rows_bwd_elem_k(const float* __restrict__ z, const float* __restrict__ jac,
                                                              const float* __restrict__ glogdet, const float* __restrict__ glogn,
                                                              const float* __restrict__ gz_in, float* __restrict__ gz,
                                                              float* __restrict__ gjac, int n, int d) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= n) return;
  const int row = e / d;
  if (gz) gz[e] = fmaf(-z[e], glogn ? glogn[row] : 0.f, gz_in ? gz_in[e] : 0.f);
  if (gjac) gjac[e] = (glogdet ? glogdet[row] : 0.f) / jac[e];
}

// Short rows (d <= 64, contiguous, 16-B aligned): the span-per-wavefront scheme of the flat Affine kernels -- one float4 of
// z and of jac per lane and span, U spans in flight, masked wave reductions per row.
template <int RW, int U>
__global__ __launch_bounds__(kBlock) void nll_rows_flat_k(const float* __restrict__ z, const float* __restrict__ jac,
                                                          float* __restrict__ logdet, float* __restrict__ logn,
                                                          int64_t B, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (kBlock / 64);
  const int64_t nspan = (B + RW - 1) / RW;
  int rid[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) rid[c] = (4 * lane + c) / d;
  for (int64_t sp0 = gw * U; sp0 < nspan; sp0 += nw * U) {
    f32x4r zv[U], jv[U];
    int ne[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t sp = sp0 + u, row0 = sp * RW;
      const int nrows = sp < nspan ? (int)(B - row0 < RW ? B - row0 : RW) : 0;
      ne[u] = nrows * d;
      const int64_t base = row0 * d + 4 * lane;
      zv[u] = f32x4r{0.f, 0.f, 0.f, 0.f};
      jv[u] = f32x4r{1.f, 1.f, 1.f, 1.f};
      if (4 * lane + 3 < ne[u]) {
        if (logn) zv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(z + base));
        if (jac) jv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(jac + base));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (4 * lane + c < ne[u]) {
            if (logn) zv[u][c] = z[base + c];
            if (jac) jv[u][c] = jac[base + c];
          }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ne[u] == 0) continue;
      const int64_t row0 = (sp0 + u) * RW;
      float lj[4], lq[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool ok = 4 * lane + c < ne[u];
        lj[c] = (jac && ok) ? logf(jv[u][c]) : 0.f;
        lq[c] = ok ? kLog2Pi + zv[u][c] * zv[u][c] : 0.f;
      }
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        if (jac) {
          float sv = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) sv += rid[c] == rr ? lj[c] : 0.f;
          sv = wave_sum(sv);
          if (lane == 0 && row0 + rr < B) logdet[row0 + rr] = sv;
        }
        if (logn) {
          float sv = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) sv += rid[c] == rr ? lq[c] : 0.f;
          sv = wave_sum(sv);
          if (lane == 0 && row0 + rr < B) logn[row0 + rr] = -0.5f * sv;
        }
      }
    }
  }
}

template <int RW, int U>
__global__ __launch_bounds__(kBlock) void nll_rows_bwd_flat_k(const float* __restrict__ z, const float* __restrict__ jac,
                                                              const float* __restrict__ glogdet,
                                                              const float* __restrict__ glogn,
                                                              const float* __restrict__ gz_in, float* __restrict__ gz,
                                                              float* __restrict__ gjac, int64_t B, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (kBlock / 64);
  const int64_t nspan = (B + RW - 1) / RW;
  int rid[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) rid[c] = (4 * lane + c) / d;
  for (int64_t sp0 = gw * U; sp0 < nspan; sp0 += nw * U) {
    f32x4r zv[U], jv[U], gv[U];
    int ne[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t sp = sp0 + u, row0 = sp * RW;
      const int nrows = sp < nspan ? (int)(B - row0 < RW ? B - row0 : RW) : 0;
      ne[u] = nrows * d;
      const int64_t base = row0 * d + 4 * lane;
      zv[u] = f32x4r{0.f, 0.f, 0.f, 0.f};
      gv[u] = zv[u];
      jv[u] = f32x4r{1.f, 1.f, 1.f, 1.f};
      if (4 * lane + 3 < ne[u]) {
        if (gz) zv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(z + base));
        if (gz_in) gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(gz_in + base));
        if (gjac) jv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4r*>(jac + base));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (4 * lane + c < ne[u]) {
            if (gz) zv[u][c] = z[base + c];
            if (gz_in) gv[u][c] = gz_in[base + c];
            if (gjac) jv[u][c] = jac[base + c];
          }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ne[u] == 0) continue;
      const int64_t row0 = (sp0 + u) * RW, base = row0 * d + 4 * lane;
      f32x4r oz, oj;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool ok = 4 * lane + c < ne[u];
        const float gn = (glogn && ok) ? glogn[row0 + rid[c]] : 0.f, gl = (glogdet && ok) ? glogdet[row0 + rid[c]] : 0.f;
        oz[c] = fmaf(-zv[u][c], gn, gv[u][c]);
        oj[c] = gl / jv[u][c];
      }
      if (4 * lane + 3 < ne[u]) {
        if (gz) __builtin_nontemporal_store(oz, reinterpret_cast<f32x4r*>(gz + base));
        if (gjac) __builtin_nontemporal_store(oj, reinterpret_cast<f32x4r*>(gjac + base));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (4 * lane + c < ne[u]) { if (gz) gz[base + c] = oz[c]; if (gjac) gjac[base + c] = oj[c]; }
      }
    }
  }
}

// Short rows, forward: ONE ROW PER 16-LANE GROUP.  The span kernels above pay RW full-wave reductions (ds_bpermute
// butterflies) per 252 elements, which is what bounds a kernel that only READS 4-8 B per element (22 % of the HBM peak for
// the density alone).  Here lane g of a group loads elements 4 g .. 4 g + 3 of its row with ONE dwordx4 (gfx950 executes
// global_load_dwordx4 at any dword address; rows of 63 floats are only dword aligned) and the row sum is four DPP adds
// inside the hardware's 16-lane row: quad_perm xor 1, xor 2, row_half_mirror, row_mirror -- VALU only, no LDS crossbar.
// A wavefront covers 4 rows per load instruction and keeps U of them in flight.
typedef float f32x4ru __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
  return v;
}

template <int U>
__global__ __launch_bounds__(kBlock) void nll_rows_g16_k(const float* __restrict__ z, const float* __restrict__ jac,
                                                         float* __restrict__ logdet, float* __restrict__ logn,
                                                         int64_t B, int d) {
  const int lane = threadIdx.x & 63, g = lane & 15, grp = lane >> 4;
  const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (kBlock / 64);
  const int nel = d - 4 * g;                           // elements of the row this lane holds: min(4, nel), <= 0: none
  for (int64_t r0 = gw * (4 * U); r0 < B; r0 += nw * (4 * U)) {
    f32x4ru zv[U], jv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = r0 + 4 * u + grp;
      const int64_t base = row * d + 4 * g;
      zv[u] = f32x4ru{0.f, 0.f, 0.f, 0.f};
      jv[u] = f32x4ru{1.f, 1.f, 1.f, 1.f};
      // the last lanes of the LAST row must not read past the array; everywhere else the float4 may run into the next row
      // (masked below), which keeps every lane's load a single dwordx4
      const bool full = row < B && nel > 0 && (nel >= 4 || row + 1 < B);
      if (full) {
        if (logn) zv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4ru*>(z + base));
        if (jac) jv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4ru*>(jac + base));
      } else if (row < B && nel > 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < nel) { if (logn) zv[u][c] = z[base + c]; if (jac) jv[u][c] = jac[base + c]; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = r0 + 4 * u + grp;
      float sl = 0.f, sn = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bool ok = c < nel;
        if (jac) sl += ok ? logf(jv[u][c]) : 0.f;
        sn += ok ? kLog2Pi + zv[u][c] * zv[u][c] : 0.f;
      }
      if (jac) { sl = row16_sum(sl); if (g == 0 && row < B) logdet[row] = sl; }
      if (logn) { sn = row16_sum(sn); if (g == 0 && row < B) logn[row] = -0.5f * sn; }
    }
  }
}

// The Affine forward on the same row-per-16-lane-group scheme (contiguous [B,d,2] conditioner output, d <= 64, dword
// alignment only): x / z one dwordx4 per lane, h two; log|det J| and the Normal log-density of the row are two DPP sums.
template <int U>
__global__ __launch_bounds__(kBlock) void affine_fwd_g16_k(const float* __restrict__ x, const float* __restrict__ h,
                                                           float* __restrict__ z, float* __restrict__ jac,
                                                           float* __restrict__ logdet, float* __restrict__ logn,
                                                           int64_t B, int d) {
  const int lane = threadIdx.x & 63, g = lane & 15, grp = lane >> 4;
  const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (kBlock / 64);
  const int nel = d - 4 * g;
  for (int64_t r0 = gw * (4 * U); r0 < B; r0 += nw * (4 * U)) {
    f32x4ru xv[U], ha[U], hb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = r0 + 4 * u + grp;
      const int64_t base = row * d + 4 * g;
      xv[u] = f32x4ru{0.f, 0.f, 0.f, 0.f};
      ha[u] = xv[u]; hb[u] = xv[u];
      const bool full = row < B && nel > 0 && (nel >= 4 || row + 1 < B);
      if (full) {
        xv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4ru*>(x + base));
        ha[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4ru*>(h + 2 * base));
        hb[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4ru*>(h + 2 * base + 4));
      } else if (row < B && nel > 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < nel) {
            xv[u][c] = x[base + c];
            const float a0 = h[2 * (base + c)], a1 = h[2 * (base + c) + 1];
            if (c < 2) { ha[u][2 * c] = a0; ha[u][2 * c + 1] = a1; } else { hb[u][2 * c - 4] = a0; hb[u][2 * c - 3] = a1; }
          }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = r0 + 4 * u + grp;
      const int64_t base = row * d + 4 * g;
      const float h0[4] = {ha[u][0], ha[u][2], hb[u][0], hb[u][2]}, h1[4] = {ha[u][1], ha[u][3], hb[u][1], hb[u][3]};
      f32x4ru zv, jv;
      float sl = 0.f, sn = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float mu = clampf(h0[c], -5.f, 5.f), ls = clampf(h1[c], -5.f, 2.f);
        const float sg = expf(ls);
        zv[c] = fmaf(xv[u][c], sg, mu);
        jv[c] = sg;
        if (c < nel) { sl += ls; sn += kLog2Pi + zv[c] * zv[c]; }
      }
      if (row < B && nel >= 4) {
        __builtin_nontemporal_store(zv, reinterpret_cast<f32x4ru*>(z + base));
        if (jac) __builtin_nontemporal_store(jv, reinterpret_cast<f32x4ru*>(jac + base));
      } else if (row < B) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < nel) { z[base + c] = zv[c]; if (jac) jac[base + c] = jv[c]; }
      }
      if (logdet) { sl = row16_sum(sl); if (g == 0 && row < B) logdet[row] = sl; }
      if (logn) { sn = row16_sum(sn); if (g == 0 && row < B) logn[row] = -0.5f * sn; }
    }
  }
}

// ... and its backward: a lane's four elements belong to ONE row, so the per-row cotangents glogdet[row] / glogn[row] are
// one load per lane (the span kernel gathers them per element through its element -> row table).
template <int U>
__global__ __launch_bounds__(kBlock) void affine_bwd_g16_k(const float* __restrict__ x, const float* __restrict__ h,
                                                           const float* __restrict__ gz, const float* __restrict__ gjac,
                                                           const float* __restrict__ glogdet,
                                                           const float* __restrict__ glogn, float* __restrict__ gx,
                                                           float* __restrict__ gh, int64_t B, int d) {
  const int lane = threadIdx.x & 63, g = lane & 15, grp = lane >> 4;
  const int64_t gw = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (kBlock / 64);
  const int nel = d - 4 * g;
  for (int64_t r0 = gw * (4 * U); r0 < B; r0 += nw * (4 * U)) {
    f32x4ru xv[U], ha[U], hb[U], gv[U], jv[U];
    float gl[U], gn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = r0 + 4 * u + grp;
      const int64_t base = row * d + 4 * g;
      xv[u] = f32x4ru{0.f, 0.f, 0.f, 0.f};
      ha[u] = xv[u]; hb[u] = xv[u]; gv[u] = xv[u]; jv[u] = xv[u];
      gl[u] = (glogdet && row < B) ? glogdet[row] : 0.f;
      gn[u] = (glogn && row < B) ? glogn[row] : 0.f;
      const bool full = row < B && nel > 0 && (nel >= 4 || row + 1 < B);
      if (full) {
        xv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4ru*>(x + base));
        ha[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4ru*>(h + 2 * base));
        hb[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4ru*>(h + 2 * base + 4));
        if (gz) gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4ru*>(gz + base));
        if (gjac) jv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4ru*>(gjac + base));
      } else if (row < B && nel > 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < nel) {
            xv[u][c] = x[base + c];
            const float a0 = h[2 * (base + c)], a1 = h[2 * (base + c) + 1];
            if (c < 2) { ha[u][2 * c] = a0; ha[u][2 * c + 1] = a1; } else { hb[u][2 * c - 4] = a0; hb[u][2 * c - 3] = a1; }
            if (gz) gv[u][c] = gz[base + c];
            if (gjac) jv[u][c] = gjac[base + c];
          }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = r0 + 4 * u + grp;
      if (row >= B || nel <= 0) continue;
      const int64_t base = row * d + 4 * g;
      const float h0[4] = {ha[u][0], ha[u][2], hb[u][0], hb[u][2]}, h1[4] = {ha[u][1], ha[u][3], hb[u][1], hb[u][3]};
      f32x4ru ox, oa, ob;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // torch clamp backward passes the gradient where min <= v <= max (boundaries included)
        const float m0 = (h0[c] >= -5.f && h0[c] <= 5.f) ? 1.f : 0.f;
        const float m1 = (h1[c] >= -5.f && h1[c] <= 2.f) ? 1.f : 0.f;
        const float sg = expf(clampf(h1[c], -5.f, 2.f));
        const float gze = fmaf(-fmaf(xv[u][c], sg, clampf(h0[c], -5.f, 5.f)), gn[u], gv[u][c]);   // + d logN / dz = -z
        ox[c] = gze * sg;
        const float o0 = gze * m0, o1 = (fmaf(gze * xv[u][c], sg, jv[u][c] * sg) + gl[u]) * m1;
        if (c < 2) { oa[2 * c] = o0; oa[2 * c + 1] = o1; } else { ob[2 * c - 4] = o0; ob[2 * c - 3] = o1; }
      }
      if (nel >= 4) {
        if (gx) __builtin_nontemporal_store(ox, reinterpret_cast<f32x4ru*>(gx + base));
        __builtin_nontemporal_store(oa, reinterpret_cast<f32x4ru*>(gh + 2 * base));
        __builtin_nontemporal_store(ob, reinterpret_cast<f32x4ru*>(gh + 2 * base + 4));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < nel) {
            if (gx) gx[base + c] = ox[c];
            gh[2 * (base + c)] = c < 2 ? oa[2 * c] : ob[2 * c - 4];
            gh[2 * (base + c) + 1] = c < 2 ? oa[2 * c + 1] : ob[2 * c - 3];
          }
      }
    }
  }
}

// rows per wavefront span of the flat row kernels (contiguous [B,d] arrays, d <= 64, all pointers 16-B aligned), else 0
inline int rows_flat_rw(int64_t B, int64_t d, uintptr_t ptr_bits) {
  if (d > 64 || B * d < (1 << 18) || (ptr_bits & 15) != 0) return 0;
  const int rw = (d % 4 == 0) ? 1 : ((d % 2 == 0) ? 2 : 4);
  return rw * d <= 256 ? rw : 0;
}

int nll_rows_launch(const float* z, const float* jac, float* logdet, float* logn, int64_t B, int64_t d, hipStream_t s) {
  void* stream = (void*)s;
  if (d >= 256 && B <= 4096) {                          // few long rows: a workgroup per row
    hipLaunchKernelGGL(nll_rows_rowblock_k, dim3((unsigned)B), dim3(kBlock), 0, s, z, jac, logdet, logn, (int)d);
  } else if (d <= 64 && B * d >= (1 << 16)) {          // short rows: a row per 16-lane group
    constexpr int U = 4;
    const int64_t per_block = (int64_t)(kBlock / 64) * 4 * U;
    int64_t grid = (B + per_block - 1) / per_block;
    if (grid > 256 * 8) grid = 256 * 8;
    hipLaunchKernelGGL((nll_rows_g16_k<U>), dim3((unsigned)grid), dim3(kBlock), 0, s, z, jac, logdet, logn, B, (int)d);
  } else if (const int rw = rows_flat_rw(B, d, (uintptr_t)z | (uintptr_t)jac)) {
    constexpr int U = 4;
    const int64_t nspan = (B + rw - 1) / rw;
    int64_t grid = (nspan + U * (kBlock / 64) - 1) / (U * (kBlock / 64));
    if (grid > 256 * 8) grid = 256 * 8;
    if (rw == 4) hipLaunchKernelGGL((nll_rows_flat_k<4, U>), dim3((unsigned)grid), dim3(kBlock), 0, s, z, jac, logdet, logn, B, (int)d);
    else if (rw == 2) hipLaunchKernelGGL((nll_rows_flat_k<2, U>), dim3((unsigned)grid), dim3(kBlock), 0, s, z, jac, logdet, logn, B, (int)d);
    else hipLaunchKernelGGL((nll_rows_flat_k<1, U>), dim3((unsigned)grid), dim3(kBlock), 0, s, z, jac, logdet, logn, B, (int)d);
  } else {
    const int G = gnf_pow2_ge(d, 64);
    GNF_DISPATCH_G(G, nll_rows_k, B, z, jac, logdet, logn, B, d);
  }
  GNF_LAUNCH_CHECK();
  return 0;
}

int nll_rows_bwd_launch(const float* z, const float* jac, const float* glogdet, const float* glogn, const float* gz_in,
                        float* gz, float* gjac, int64_t B, int64_t d, hipStream_t s) {
  void* stream = (void*)s;
  if (d >= 256 && B * d <= (1 << 22)) {                 // few long rows: a thread per element
    const int n = (int)(B * d);
    hipLaunchKernelGGL(nll_rows_bwd_elem_k, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, z, jac, glogdet, glogn,
                       gz_in, gz, gjac, n, (int)d);
  } else if (const int rw = rows_flat_rw(B, d, (uintptr_t)z | (uintptr_t)jac | (uintptr_t)gz_in | (uintptr_t)gz | (uintptr_t)gjac)) {
    constexpr int U = 2;
    const int64_t nspan = (B + rw - 1) / rw;
    int64_t grid = (nspan + U * (kBlock / 64) - 1) / (U * (kBlock / 64));
    if (grid > 256 * 16) grid = 256 * 16;
    if (rw == 4) hipLaunchKernelGGL((nll_rows_bwd_flat_k<4, U>), dim3((unsigned)grid), dim3(kBlock), 0, s, z, jac, glogdet, glogn, gz_in, gz, gjac, B, (int)d);
    else if (rw == 2) hipLaunchKernelGGL((nll_rows_bwd_flat_k<2, U>), dim3((unsigned)grid), dim3(kBlock), 0, s, z, jac, glogdet, glogn, gz_in, gz, gjac, B, (int)d);
    else hipLaunchKernelGGL((nll_rows_bwd_flat_k<1, U>), dim3((unsigned)grid), dim3(kBlock), 0, s, z, jac, glogdet, glogn, gz_in, gz, gjac, B, (int)d);
  } else {
    const int G = gnf_pow2_ge(d, 64);
    GNF_DISPATCH_G(G, nll_rows_bwd_k, B, z, jac, glogdet, glogn, gz_in, gz, gjac, B, d);
  }
  GNF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ batch mean of the log-likelihood
// out = -mean_b(logdet[b] + logn[b]): the data term of FCNormalizingFlow.loss (models/NormalizingFlow.py:144-146) in one
// launch (torch: add, mean, neg + their backward = 8 launches of a step that has ~40).  One workgroup, fixed summation
// order: per-thread strided partials, butterfly per wavefront, the 16 wavefront sums in order.
__global__ __launch_bounds__(1024) void nll_mean_k(const float* __restrict__ logdet, const float* __restrict__ logn,
                                                   const float* __restrict__ addend, float* __restrict__ out, int64_t B) {
  __shared__ float red[16];
  float s = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += 1024) s += logdet[b] + logn[b];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, GNF_WAVE);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    const float nll = -t / (float)B;
    out[0] = addend ? addend[0] + nll : nll;          // c + (-mean) == c - mean: the reference's constraints - mean(...)
  }
}
// both cotangents of the above: glogdet[b] = glogn[b] = -g / B (g: device scalar)
__global__ void nll_mean_bwd_k(const float* __restrict__ g, float* __restrict__ glogdet, float* __restrict__ glogn, int64_t B) {
  const float v = -g[0] / (float)B;
  for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
    glogdet[b] = v;
    glogn[b] = v;
  }
}

// The same term computed from z ITSELF: out = addend - mean_b(logdet[b] + logN(z[b,:])), logN(z) = -1/2 sum_i (log 2 pi + z_i^2)
// (NormalizingFlowFactories.py:15-16).  The loss reads the z it is handed -- not a density some earlier kernel reduced from
// what z held at that time -- so a z rewritten in any way between forward and loss (in place, through .data, by a raw
// pointer) is simply the z that is scored, as in the reference.  One workgroup: sum_b logdet and the FLAT sum of z^2 over
// all B*d elements (no row structure is needed for the mean), fixed summation order.  For B*d <= kLossFlatMax; larger
// batches take the row kernel + nll_mean_k.
constexpr int64_t kLossFlatMax = (int64_t)1 << 20;
__global__ __launch_bounds__(1024) void nll_loss_k(const float* __restrict__ z, const float* __restrict__ logdet,
                                                   const float* __restrict__ addend, float* __restrict__ out, int64_t B,
                                                   int64_t d) {
  __shared__ float red[2][16];
  const int64_t n = B * d;
  float s = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += 1024) s += logdet[b];
  const bool vec = (reinterpret_cast<uintptr_t>(z) & 15) == 0;
  const int64_t n4 = vec ? n / 4 : 0;
  int64_t i = threadIdx.x;
  for (; i + 7 * 1024 < n4; i += 8 * 1024) {                   // eight 16-B loads in flight per thread (one workgroup reads it all)
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4*>(z)[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      q0 = fmaf(v[u].x, v[u].x, q0); q1 = fmaf(v[u].y, v[u].y, q1); q2 = fmaf(v[u].z, v[u].z, q2); q3 = fmaf(v[u].w, v[u].w, q3);
    }
  }
  for (; i < n4; i += 1024) {
    const float4 v = reinterpret_cast<const float4*>(z)[i];
    q0 = fmaf(v.x, v.x, q0); q1 = fmaf(v.y, v.y, q1); q2 = fmaf(v.z, v.z, q2); q3 = fmaf(v.w, v.w, q3);
  }
  for (int64_t t = 4 * n4 + threadIdx.x; t < n; t += 1024) q0 = fmaf(z[t], z[t], q0);
  float q = (q0 + q1) + (q2 + q3);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off, GNF_WAVE); q += __shfl_xor(q, off, GNF_WAVE); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ts = 0.f, tq = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) { ts += red[0][w]; tq += red[1][w]; }
    const float logn = -.5f * ((float)d * (float)B * 1.8378770664093453f + tq);      // log(2 pi)
    const float nll = -(ts + logn) / (float)B;
    out[0] = addend ? addend[0] + nll : nll;
  }
}
// its cotangents: gz[b,i] = g z[b,i] / B, glogdet[b] = -g / B (g: device scalar)
__global__ void nll_loss_bwd_k(const float* __restrict__ g, const float* __restrict__ z, float* __restrict__ gz,
                               float* __restrict__ glogdet, int64_t B, int64_t d) {
  const float v = g[0] / (float)B;
  const int64_t n = B * d, stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t b = t0; b < B; b += stride) glogdet[b] = -v;
  const bool vec = ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(gz)) & 15) == 0;
  const int64_t n4 = vec ? n / 4 : 0;
  for (int64_t i = t0; i < n4; i += stride) {
    const float4 a = reinterpret_cast<const float4*>(z)[i];
    reinterpret_cast<float4*>(gz)[i] = make_float4(v * a.x, v * a.y, v * a.z, v * a.w);
  }
  for (int64_t i = 4 * n4 + t0; i < n; i += stride) gz[i] = v * z[i];
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
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  const float4* g4 = reinterpret_cast<const float4*>(g);
  int64_t i = tid;
  for (; i + nth < n4; i += 2 * nth) {               // two independent groups per thread: eight 16-B requests in flight
    const int64_t k = i + nth;
    float4 pa = p4[i], ma = m4[i], va = v4[i], pb = p4[k], mb = m4[k], vb = v4[k];
    const float4 ga = g4[i], gb = g4[k];
    adam_elem(pa.x, ga.x, ma.x, va.x, c, step_size, bc2_sqrt);
    adam_elem(pa.y, ga.y, ma.y, va.y, c, step_size, bc2_sqrt);
    adam_elem(pa.z, ga.z, ma.z, va.z, c, step_size, bc2_sqrt);
    adam_elem(pa.w, ga.w, ma.w, va.w, c, step_size, bc2_sqrt);
    adam_elem(pb.x, gb.x, mb.x, vb.x, c, step_size, bc2_sqrt);
    adam_elem(pb.y, gb.y, mb.y, vb.y, c, step_size, bc2_sqrt);
    adam_elem(pb.z, gb.z, mb.z, vb.z, c, step_size, bc2_sqrt);
    adam_elem(pb.w, gb.w, mb.w, vb.w, c, step_size, bc2_sqrt);
    p4[i] = pa; m4[i] = ma; v4[i] = va; p4[k] = pb; m4[k] = mb; v4[k] = vb;
  }
  for (; i < n4; i += nth) {
    float4 pp = p4[i], mm = m4[i], vv = v4[i];
    const float4 gg = g4[i];
    adam_elem(pp.x, gg.x, mm.x, vv.x, c, step_size, bc2_sqrt);
    adam_elem(pp.y, gg.y, mm.y, vv.y, c, step_size, bc2_sqrt);
    adam_elem(pp.z, gg.z, mm.z, vv.z, c, step_size, bc2_sqrt);
    adam_elem(pp.w, gg.w, mm.w, vv.w, c, step_size, bc2_sqrt);
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
  }
  for (int64_t i = 4 * n4 + tid; i < n; i += nth) adam_elem(p[i], g[i], m[i], v[i], c, step_size, bc2_sqrt);
}

__global__ void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, int64_t n, int vec, AdamC c, float step_size, float bc2_sqrt) {
  adam_body(p, g, m, v, n, vec != 0, c, step_size, bc2_sqrt);
}

// Same update with the step count in device memory, so that a captured hipGraph of the whole training step can be
// replayed: the bias corrections are recomputed on the device from step_dev[0] + 1.  advance: every workgroup takes a
// ticket (step_dev[1]) once all its wavefronts have read the count, and whoever holds the LAST ticket stores the new
// count at its end -- every other workgroup has read the old one by then -- and resets the tickets.  Same-address
// device-scope atomics serialise at ~27 ns each: the grid is at most 256 workgroups of 1024 threads and the tickets are
// drawn BEFORE the streaming loop, whose ~20 us hide them (drawn after it by 1568 workgroups they cost 43 us; a release
// fence in front of them, one L2 write-back per workgroup: 100 us).  A separate one-thread launch for the increment cost
// 4 us of a 170-us MADE step.  tools/bench_adam.py at 3.2 M parameters (HIP events): adam_k 17.3 us, this kernel 19.3
// without / 20.5 with the increment; the 2 us are occupancy (one 16-wavefront workgroup per CU against seven 4-wavefront
// ones) -- more, smaller workgroups cost more in tickets than they win (512 x 1024: 27 us, 768 x 512: 25 us, also with
// two-level tickets), and neither cached bias corrections instead of two double pow() per thread nor non-temporal
// gradient loads moved it.
constexpr int kAdamDevBlock = 1024;
__global__ __launch_bounds__(kAdamDevBlock) void adam_dev_k(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v, int64_t n, int vec,
                                                            AdamC c, double lr, double b1, double b2, int* step_dev, int advance) {
  const int taken = *reinterpret_cast<volatile int*>(step_dev);
  unsigned ticket = 0;
  if (advance) {
    __syncthreads();
    if (threadIdx.x == 0)
      ticket = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(step_dev + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const double t = (double)(taken + 1);
  const float step_size = (float)(lr / (1.0 - pow(b1, t))), bc2_sqrt = (float)sqrt(1.0 - pow(b2, t));
  adam_body(p, g, m, v, n, vec != 0, c, step_size, bc2_sqrt);
  if (advance && threadIdx.x == 0 && ticket == gridDim.x - 1) {
    step_dev[1] = 0;
    step_dev[0] = taken + 1;
  }
}

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

// Two independent row sums in ONE launch (the Monotonic backward's accumulator rows and its partial vector rows: two 6-us
// launches behind the chain kernel before): workgroups [0, ga) take segment a, the rest segment b; same summation order
// per segment as rowsum_k.
struct RowsumSeg { const float* src; float* out; int64_t P, N; int accumulate; };
__global__ __launch_bounds__(1024) void rowsum2_k(RowsumSeg a, RowsumSeg b, unsigned ga) {
  __shared__ float red[16][64];
  const bool first = blockIdx.x < ga;
  const RowsumSeg& g = first ? a : b;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)(first ? blockIdx.x : blockIdx.x - ga) * 64 + lane;
  const float* src = g.src;
  const int64_t P = g.P, N = g.N;
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
    float s = g.accumulate ? g.out[n] : 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += red[w][lane];
    g.out[n] = s;
  }
}

int gnf_rowsum2_launch(const float* src_a, float* out_a, int64_t Pa, int64_t Na, int acc_a, const float* src_b, float* out_b,
                       int64_t Pb, int64_t Nb, int acc_b, hipStream_t s) {
  if (Na <= 0) return gnf_rowsum_launch(src_b, out_b, Pb, Nb, acc_b, s);
  if (Nb <= 0) return gnf_rowsum_launch(src_a, out_a, Pa, Na, acc_a, s);
  const unsigned ga = (unsigned)((Na + 63) / 64), gb = (unsigned)((Nb + 63) / 64);
  hipLaunchKernelGGL(rowsum2_k, dim3(ga + gb), dim3(1024), 0, s, RowsumSeg{src_a, out_a, Pa, Na, acc_a},
                     RowsumSeg{src_b, out_b, Pb, Nb, acc_b}, ga);
  GNF_LAUNCH_CHECK();
  return 0;
}

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
                   float* logdet, float* logn, int clamp_inplace, int64_t B, int64_t d, gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;                    // batch-sized arrays may be NULL for an empty batch
  if (!x || !h || !z) return GNF_EINVAL;
  const int G = gnf_pow2_ge(d, 64);
  if (!clamp_inplace && h_sc == 1 && h_sd == 2 && h_sb == 2 * d && d <= 64 && B * d >= (1 << 16)) {
    // short rows in the contiguous [B,d,2] layout: a row per 16-lane group.  Rows per group and trip (round 6, same-box sweep
    // profiles/r06_affine_floor.txt): ONE below 2^24 elements -- at the BASELINE shape [50 000, 63] the launch is a handful of
    // trips per CU and more workgroups in flight beat more bytes per wavefront (10.9 us back to back against 11.5 / 12.8 for
    // two / four) --, FOUR above (10^6 x 63: 0.66-0.68 of 8 TB/s against 0.63)
    const int U = B * d >= ((int64_t)1 << 24) ? 4 : 1;
    const int64_t per_block = (int64_t)(kBlock / 64) * 4 * U;
    int64_t grid = (B + per_block - 1) / per_block;
    if (grid > 256 * 16) grid = 256 * 16;
    if (U == 1) hipLaunchKernelGGL((affine_fwd_g16_k<1>), dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, x, h, z, jac, logdet, logn, B, (int)d);
    else hipLaunchKernelGGL((affine_fwd_g16_k<4>), dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, x, h, z, jac, logdet, logn, B, (int)d);
    GNF_LAUNCH_CHECK();
    return 0;
  }
  if (const int rw = clamp_inplace ? 0 : affine_flat_rw(h_sb, h_sd, h_sc, B, d, x, h, z)) {
    if (!jac || ((uintptr_t)jac & 15) == 0) {
      const int64_t nspan = (B + rw - 1) / rw;
      int64_t grid = (nspan + 2 * (kBlock / 64) - 1) / (2 * (kBlock / 64));
      if (grid > 256 * 16) grid = 256 * 16;
      hipStream_t s = (hipStream_t)stream;
      if (rw == 4) hipLaunchKernelGGL((affine_fwd_flat_k<4, 2>), dim3((unsigned)grid), dim3(kBlock), 0, s, x, h, z, jac, logdet, logn, B, (int)d);
      else if (rw == 2) hipLaunchKernelGGL((affine_fwd_flat_k<2, 2>), dim3((unsigned)grid), dim3(kBlock), 0, s, x, h, z, jac, logdet, logn, B, (int)d);
      else hipLaunchKernelGGL((affine_fwd_flat_k<1, 2>), dim3((unsigned)grid), dim3(kBlock), 0, s, x, h, z, jac, logdet, logn, B, (int)d);
      GNF_LAUNCH_CHECK();
      return 0;
    }
  }
  if (!clamp_inplace && d >= 256 && B <= 4096) {         // few long rows: a workgroup per row
    hipLaunchKernelGGL(affine_fwd_rowblock_k, dim3((unsigned)B), dim3(kBlock), 0, (hipStream_t)stream, x, h, h_sb, h_sd, h_sc,
                       z, jac, logdet, logn, (int)d);
    GNF_LAUNCH_CHECK();
    return 0;
  }
  if (clamp_inplace) {
    if (B * d >= (1 << 20)) GNF_DISPATCH_GR(G, 4, affine_fwd_clamp_k, B, x, h, h_sb, h_sd, h_sc, z, jac, logdet, logn, clamp_inplace, B, d);
    else GNF_DISPATCH_GR(G, 1, affine_fwd_clamp_k, B, x, h, h_sb, h_sd, h_sc, z, jac, logdet, logn, clamp_inplace, B, d);
  } else {
    if (B * d >= (1 << 20)) GNF_DISPATCH_GR(G, 4, affine_fwd_plain_k, B, x, h, h_sb, h_sd, h_sc, z, jac, logdet, logn, clamp_inplace, B, d);
    else GNF_DISPATCH_GR(G, 1, affine_fwd_plain_k, B, x, h, h_sb, h_sd, h_sc, z, jac, logdet, logn, clamp_inplace, B, d);
  }
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_affine_bwd(const float* x, const float* h, int64_t h_sb, int64_t h_sd, int64_t h_sc, const float* gz,
                   const float* gjac, const float* glogdet, const float* glogn, float* gx, float* gh, int64_t g_sb,
                   int64_t g_sd, int64_t g_sc, int64_t B, int64_t d, gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;
  if (!x || !h || !gh) return GNF_EINVAL;
  const int G = gnf_pow2_ge(d, 64);
  if (g_sc == 1 && g_sd == 2 && g_sb == 2 * d && h_sc == 1 && h_sd == 2 && h_sb == 2 * d && d <= 64 && B * d >= (1 << 16)) {
    // short rows, contiguous [B,d,2] h and gh: a row per 16-lane group; rows per group and trip as in the forward
    const int U = B * d >= ((int64_t)1 << 24) ? 4 : 1;
    const int64_t per_block = (int64_t)(kBlock / 64) * 4 * U;
    int64_t grid = (B + per_block - 1) / per_block;
    if (grid > 256 * 16) grid = 256 * 16;
    if (U == 1) hipLaunchKernelGGL((affine_bwd_g16_k<1>), dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, x, h, gz, gjac, glogdet, glogn, gx, gh, B, (int)d);
    else hipLaunchKernelGGL((affine_bwd_g16_k<4>), dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, x, h, gz, gjac, glogdet, glogn, gx, gh, B, (int)d);
    GNF_LAUNCH_CHECK();
    return 0;
  }
  if (const int rw = (g_sc == 1 && g_sd == 2 && g_sb == 2 * d) ? affine_flat_rw(h_sb, h_sd, h_sc, B, d, x, h, gh) : 0) {
    if ((((uintptr_t)gz | (uintptr_t)gjac | (uintptr_t)gx) & 15) == 0) {
      const int64_t nspan = (B + rw - 1) / rw;
      int64_t grid = (nspan + 2 * (kBlock / 64) - 1) / (2 * (kBlock / 64));
      if (grid > 256 * 16) grid = 256 * 16;
      hipStream_t s = (hipStream_t)stream;
      if (rw == 4) hipLaunchKernelGGL((affine_bwd_flat_k<4, 2>), dim3((unsigned)grid), dim3(kBlock), 0, s, x, h, gz, gjac, glogdet, glogn, gx, gh, B, (int)d);
      else if (rw == 2) hipLaunchKernelGGL((affine_bwd_flat_k<2, 2>), dim3((unsigned)grid), dim3(kBlock), 0, s, x, h, gz, gjac, glogdet, glogn, gx, gh, B, (int)d);
      else hipLaunchKernelGGL((affine_bwd_flat_k<1, 2>), dim3((unsigned)grid), dim3(kBlock), 0, s, x, h, gz, gjac, glogdet, glogn, gx, gh, B, (int)d);
      GNF_LAUNCH_CHECK();
      return 0;
    }
  }
  if (d >= 256 && B * d <= (1 << 22)) {                  // few long rows: a thread per element
    const int n = (int)(B * d);
    hipLaunchKernelGGL(affine_bwd_elem_k, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, x, h,
                       h_sb, h_sd, h_sc, gz, gjac, glogdet, glogn, gx, gh, g_sb, g_sd, g_sc, n, (int)d);
    GNF_LAUNCH_CHECK();
    return 0;
  }
  if (B * d >= (1 << 20)) GNF_DISPATCH_GR(G, 4, affine_bwd_k, B, x, h, h_sb, h_sd, h_sc, gz, gjac, glogdet, glogn, gx, gh, g_sb, g_sd, g_sc, B, d);
  else GNF_DISPATCH_GR(G, 1, affine_bwd_k, B, x, h, h_sb, h_sd, h_sc, gz, gjac, glogdet, glogn, gx, gh, g_sb, g_sd, g_sc, B, d);
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
  return nll_rows_launch(nullptr, jac, out, nullptr, B, d, (hipStream_t)stream);
}

int gnf_logsum_rows_bwd(const float* jac, const float* g, float* gjac, int64_t B, int64_t d, gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;
  if (!jac || !g || !gjac) return GNF_EINVAL;
  return nll_rows_bwd_launch(nullptr, jac, g, nullptr, nullptr, nullptr, gjac, B, d, (hipStream_t)stream);
}

int gnf_normal_logdensity_fwd(const float* z, float* out, int64_t B, int64_t d, gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;
  if (!z || !out) return GNF_EINVAL;
  return nll_rows_launch(z, nullptr, nullptr, out, B, d, (hipStream_t)stream);
}

int gnf_normal_logdensity_bwd(const float* z, const float* g, float* gz, int64_t B, int64_t d, gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;
  if (!z || !g || !gz) return GNF_EINVAL;
  return nll_rows_bwd_launch(z, nullptr, nullptr, g, nullptr, gz, nullptr, B, d, (hipStream_t)stream);
}

int gnf_nll_reduce_fwd(const float* z, const float* jac, float* logdet, float* logn, int64_t B, int64_t d,
                       gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;
  if (!z || !jac || !logdet || !logn) return GNF_EINVAL;
  return nll_rows_launch(z, jac, logdet, logn, B, d, (hipStream_t)stream);
}

int gnf_nll_reduce_bwd(const float* z, const float* jac, const float* glogdet, const float* glogn, const float* gz_in,
                       float* gz, float* gjac, int64_t B, int64_t d, gnf_stream_t stream) {
  if (B < 0 || d <= 0) return GNF_EINVAL;
  if (B == 0) return 0;
  if (!z || !jac || !gz || !gjac) return GNF_EINVAL;
  return nll_rows_bwd_launch(z, jac, glogdet, glogn, gz_in, gz, gjac, B, d, (hipStream_t)stream);
}

int gnf_nll_mean_fwd(const float* logdet, const float* logn, const float* addend, float* out, int64_t B,
                     gnf_stream_t stream) {
  if (B <= 0 || !logdet || !logn || !out) return GNF_EINVAL;
  hipLaunchKernelGGL(nll_mean_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, logdet, logn, addend, out, B);
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_nll_mean_bwd(const float* g, float* glogdet, float* glogn, int64_t B, gnf_stream_t stream) {
  if (B <= 0 || !g || !glogdet || !glogn) return GNF_EINVAL;
  int64_t grid = (B + kBlock - 1) / kBlock;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(nll_mean_bwd_k, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, g, glogdet, glogn, B);
  GNF_LAUNCH_CHECK();
  return 0;
}

int64_t gnf_nll_loss_max_elems(void) { return kLossFlatMax; }

int gnf_nll_loss_fwd(const float* z, const float* logdet, const float* addend, float* out, int64_t B, int64_t d,
                     gnf_stream_t stream) {
  if (B <= 0 || d <= 0 || !z || !logdet || !out) return GNF_EINVAL;
  if (B * d > kLossFlatMax) return GNF_ESHAPE;
  hipLaunchKernelGGL(nll_loss_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, z, logdet, addend, out, B, d);
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_nll_loss_bwd(const float* g, const float* z, float* gz, float* glogdet, int64_t B, int64_t d, gnf_stream_t stream) {
  if (B <= 0 || d <= 0 || !g || !z || !gz || !glogdet) return GNF_EINVAL;
  int64_t grid = (B * d / 4 + kBlock - 1) / kBlock;
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(nll_loss_bwd_k, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, g, z, gz, glogdet, B, d);
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
  hipLaunchKernelGGL(adam_k, dim3(grid_1d(vec ? (n + 7) / 8 : n)), dim3(kBlock), 0, (hipStream_t)stream, p, g, m, v, n,
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
  int64_t grid = ((vec ? (n + 7) / 8 : n) + kAdamDevBlock - 1) / kAdamDevBlock;
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(adam_dev_k, dim3((unsigned)grid), dim3(kAdamDevBlock), 0, (hipStream_t)stream, p, g, m, v,
                     n, vec, adam_consts(beta1, beta2, eps, weight_decay, grad_scale), lr, beta1, beta2, step_dev,
                     advance);
  GNF_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
