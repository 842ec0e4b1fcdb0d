// Shared helpers for the gfx950 kernels of libgnf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gnf_hip.h"

#define GNF_WAVE 64

#define GNF_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

static inline int gnf_pow2_ge(int64_t v, int cap) {
  int g = 1;
  while (g < v && g < cap) g <<= 1;
  return g;
}

// Sum over the G (power of two <= 64) consecutive lanes of an aligned lane group.
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, GNF_WAVE);
  return v;
}

// Philox4x32-10 (Salmon et al. 2011): counter-based, so forward and backward of the DAG
// gate regenerate identical noise from (seed, offset, element index).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a high and a low one: the quarter-rate integer
    // multiplies are the dominant cost of a call
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 24-bit uniform in [0,1), the resolution torch.rand has for fp32.
__device__ __forceinline__ float u01_24(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// out[n] (+)= sum_{p<P} src[p*N + n]   (gnf_rowwise.hip; deterministic order)
int gnf_rowsum_launch(const float* src, float* out, int64_t P, int64_t N, int accumulate, hipStream_t s);
// two independent row sums in one launch
int gnf_rowsum2_launch(const float* src_a, float* out_a, int64_t Pa, int64_t Na, int acc_a, const float* src_b, float* out_b,
                       int64_t Pb, int64_t Nb, int acc_b, hipStream_t s);
// same for tall inputs: two-level, ws >= kRowsumChunks*N floats
constexpr int kRowsumChunks = 1024;
int gnf_rowsum_tall_launch(const float* src, float* out, int64_t P, int64_t N, int accumulate, float* ws,
                           hipStream_t s);
