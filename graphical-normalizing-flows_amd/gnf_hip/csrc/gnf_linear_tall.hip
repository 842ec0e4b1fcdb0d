// Linear layers with a TALL batch and a NARROW output: MNISTCNN.fc2 of the headline model (models/MLP.py:47 behind
// DAGConditioner.py:169): 78 400 masked copies x 128 -> 30, forward and autograd.
//
// The tiled GEMM pads N = 30 to a 64-wide tile and runs the layer's three products as three launches plus split-K
// reductions and column sums (fwd 19 us; bwd 48.6 + 32.6 + 22 us of reductions per cfg4 step).  The bytes say
// 40 MB + 9.4 MB forward and 49 MB in + 40 MB out backward -- ~10 and ~18 us of HBM time -- and the weight (15 KB) fits
// the registers of ONE wavefront as MFMA fragments.  So:
//   forward   lin_fwd_tall_k   a wavefront keeps W as A-operand fragments (out unit = M index), walks 16-row tiles of x
//             (each a contiguous 16 K floats), row j of the tile in lanes (., j): one dwordx4 per 16-wide chunk of k feeds
//             four v_mfma_f32_16x16x4_f32 steps; y rows leave as dwordx4 (bias / ReLU fused).
//   backward  lin_bwd_tall_k   ONE launch for both gradients: wavefronts 0..3 of a workgroup compute
//             gx = (g W) o [a > 0] for the tile (W^T fragments resident), wavefronts 4..7 accumulate gW += g^T a and
//             gb += colsum g over the same tiles in the same order (16 accumulator tiles at K = 128: interleaved
//             columns, unit 64 blk + 4 j + c in lane j of tile c, so that a lane's dwordx4 of an `a` row serves four
//             tiles) -- the second reader of a tile finds it in L2.  Partials per workgroup, summed in fixed order by
//             lin_tall_reduce_k -- which also returns the column sums of gx when the caller asks (the bias gradient of the
//             layer below: fc1's, a 40 MB column-sum pass otherwise).  g rows are 120 B: dwordx4 at dword alignment, through a bounds-checked buffer
//             descriptor (the last row's overhang reads zeros).
#include "gnf_common.h"
#include "gnf_linear_tall.h"
#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));


__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// row loads: a dwordx4 of row `r` at column c0 (a multiple of 4) through a bounds-checked descriptor -- dword alignment
// is enough on gfx950, a request past the end of the buffer returns zeros -- with the columns >= ncols (the next row's
// first floats when ncols is not a multiple of 4, or a whole chunk of padding) cleared
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ f32x4 ldrow(rsrc_t rs, int r, int ncols, int c0) {
  // always issued (no branch around a load: the requests of a tile must go out back to back); what lies outside the
  // row is cleared afterwards
  const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, (r * ncols + c0) * 4, 0, 0);
  f32x4 v = __builtin_bit_cast(f32x4, u);
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = c0 + k < ncols ? v[k] : 0.f;
  return v;
}
// a row whose length is a whole number of 16-wide chunks (KX instantiations): a plain 16-B aligned global load
template <bool KX>
__device__ __forceinline__ f32x4 ldrowx(const float* base, rsrc_t rs, int r, int ncols, int c0) {
  if constexpr (KX) return *reinterpret_cast<const f32x4*>(base + (int64_t)r * ncols + c0);
  else return ldrow(rs, r, ncols, c0);
}
__device__ __forceinline__ rsrc_t mkrsrc(const float* p, int64_t floats) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, (int)(floats * 4), 0x00020000);
}

// ------------------------------------------------------------------------------------------------------------- forward
template <int NT, int KC, bool KX>  // N <= 16 NT out units, K <= 16 KC inputs (KX: K == 16 KC exactly)
__global__ __launch_bounds__(256) void lin_fwd_tall_k(const float* __restrict__ x, const float* __restrict__ W,
                                                      const float* __restrict__ bias, int relu, float* __restrict__ y,
                                                      int M, int N, int Krt) {
  const int K = KX ? 16 * KC : Krt;                   // a compile-time row pitch in the exact instantiations
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, j = lane & 15;
  const rsrc_t rx = mkrsrc(x, (int64_t)M * K), rw = mkrsrc(W, (int64_t)N * K);
  f32x4 wf[NT][KC], bv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int o = 16 * nt + j;
#pragma unroll
    for (int t = 0; t < KC; ++t) wf[nt][t] = o < N ? ldrow(rw, o, K, 16 * t + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[nt][r] = (bias && 16 * nt + 4 * q + r < N) ? bias[16 * nt + 4 * q + r] : 0.f;
  }
  const int ntiles = (M + 15) / 16;
  const int stride = gridDim.x * 4;
  int tile = blockIdx.x * 4 + wave;
  f32x4 xv[KC];
  auto load = [&](int tl, f32x4* dst) {
    const int row = 16 * tl + j;
    const int rc = row < M ? row : M - 1;
#pragma unroll
    for (int t = 0; t < KC; ++t) dst[t] = ldrowx<KX>(x, rx, rc, K, 16 * t + 4 * q);
  };
  if (tile < ntiles) load(tile, xv);
  for (; tile < ntiles; tile += stride) {
    f32x4 xn[KC];
    const bool more = tile + stride < ntiles;
    if (more) load(tile + stride, xn);
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = bv[nt];
#pragma unroll
    for (int t = 0; t < KC; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma(wf[nt][t][r], xv[t][r], acc[nt]);
    const int row = 16 * tile + j;
    if (row < M) {
      float* yr = y + (int64_t)row * N;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 v = acc[nt];
        if (relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        const int c0 = 16 * nt + 4 * q;
        if (c0 + 3 < N) *reinterpret_cast<f32x4u*>(yr + c0) = v;
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (c0 + r < N) yr[c0 + r] = v[r];
        }
      }
    }
    if (more) {
#pragma unroll
      for (int t = 0; t < KC; ++t) xv[t] = xn[t];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ backward
template <int NT, int KC, bool KX>
__global__ __launch_bounds__(512, 1) void lin_bwd_tall_k(const float* __restrict__ g, const float* __restrict__ W,
                                                         const float* __restrict__ a, int gated, float* __restrict__ gx,
                                                         float* __restrict__ part, int M, int N, int Krt, int want_xsum) {
  const int K = KX ? 16 * KC : Krt;                   // a compile-time row pitch in the exact instantiations
  constexpr int KB = (KC + 3) / 4;                    // 64-column blocks of the weight-gradient role
  __shared__ f32x4 red[4][NT * KB * 4][64];           // the four weight-gradient wavefronts' accumulators
  __shared__ float redb[4][NT][16];
  __shared__ float redx[4][16 * KC];                  // column sums of gx (the bias gradient of the layer below)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int ntiles = (M + 15) / 16;
  const int stride = gridDim.x * 4;
  const rsrc_t rg = mkrsrc(g, (int64_t)M * N), ra = mkrsrc(a, (int64_t)M * K);

  if (wave < 4) {
    // ---- data gradient: D[m = in unit][n = row] = sum_o W[o][i] g[row][o]
    f32x4 wa[KC][NT];
#pragma unroll
    for (int it = 0; it < KC; ++it)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * t + 4 * q + r, i = 16 * it + j;
          wa[it][t][r] = (o < N && i < K) ? W[(int64_t)o * K + i] : 0.f;
        }
    f32x4 gv[NT], av[KC], xs[KC];
#pragma unroll
    for (int it = 0; it < KC; ++it) xs[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load = [&](int tl, f32x4* gd, f32x4* ad) {
      const int row = 16 * tl + j;
      const int rc = row < M ? row : M - 1;
#pragma unroll
      for (int t = 0; t < NT; ++t) gd[t] = ldrow(rg, rc, N, 16 * t + 4 * q);
      if (gated) {
#pragma unroll
        for (int it = 0; it < KC; ++it) ad[it] = ldrowx<KX>(a, ra, rc, K, 16 * it + 4 * q);
      }
    };
    int tile = blockIdx.x * 4 + wave;
    if (tile < ntiles) load(tile, gv, av);
    for (; tile < ntiles; tile += stride) {
      f32x4 gn[NT], an[KC];
      const bool more = tile + stride < ntiles;
      if (more) load(tile + stride, gn, an);
      const int row = 16 * tile + j;
      float* xr = gx + (int64_t)row * K;
#pragma unroll
      for (int it = 0; it < KC; ++it) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = mfma(wa[it][t][r], gv[t][r], acc);
        if (gated) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = av[it][r] > 0.f ? acc[r] : 0.f;
        }
        const int c0 = 16 * it + 4 * q;               // K is a multiple of 4: a quad is inside the row or outside
        if (row < M && c0 < K) {
          *reinterpret_cast<f32x4u*>(xr + c0) = acc;
          xs[it] += acc;
        }
      }
      if (more) {
#pragma unroll
        for (int t = 0; t < NT; ++t) gv[t] = gn[t];
#pragma unroll
        for (int it = 0; it < KC; ++it) av[it] = an[it];
      }
    }
    if (want_xsum) {                                  // rows live in the lanes j: sum over them, lane (q, 0) holds units 4 q + r
#pragma unroll
      for (int it = 0; it < KC; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = xs[it][r];
          v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
          if (j == 0) redx[wave][16 * it + 4 * q + r] = v;
        }
    }
    __syncthreads();
  } else {
    // ---- weight gradient: D[m = out unit][n -> in unit 64 blk + 4 j + c] += sum_rows g[row][o] a[row][i]
    const int w = wave - 4;
    f32x4 acc[NT][KB][4];
    float bsum[NT];
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) {
      bsum[ot] = 0.f;
#pragma unroll
      for (int b = 0; b < KB; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[ot][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float gv[NT][4];
    f32x4 av[KB][4];
    auto load = [&](int tl, float (*gd)[4], f32x4 (*ad)[4]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * tl + 4 * q + r;
        const int rc = row < M ? row : M - 1;
#pragma unroll
        for (int ot = 0; ot < NT; ++ot) gd[ot][r] = (16 * ot + j < N) ? g[(int64_t)rc * N + 16 * ot + j] : 0.f;
#pragma unroll
        for (int b = 0; b < KB; ++b) ad[b][r] = ldrowx<(KX && KC % 4 == 0)>(a, ra, rc, K, 64 * b + 4 * j);
      }
    };
    int tile = blockIdx.x * 4 + w;
    if (tile < ntiles) load(tile, gv, av);
    for (; tile < ntiles; tile += stride) {
      float gn[NT][4];
      f32x4 an[KB][4];
      const bool more = tile + stride < ntiles;
      if (more) load(tile + stride, gn, an);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = 16 * tile + 4 * q + r < M;
#pragma unroll
        for (int ot = 0; ot < NT; ++ot) {
          const float gg = ok ? gv[ot][r] : 0.f;
          bsum[ot] += gg;
#pragma unroll
          for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[ot][b][c] = mfma(gg, av[b][r][c], acc[ot][b][c]);
        }
      }
      if (more) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int ot = 0; ot < NT; ++ot) gv[ot][r] = gn[ot][r];
#pragma unroll
          for (int b = 0; b < KB; ++b) av[b][r] = an[b][r];
        }
      }
    }
#pragma unroll
    for (int ot = 0; ot < NT; ++ot) {
#pragma unroll
      for (int b = 0; b < KB; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) red[w][(ot * KB + b) * 4 + c][lane] = acc[ot][b][c];
      float s = bsum[ot];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      if (q == 0) redb[w][ot][j] = s;
    }
    __syncthreads();
  }
  // ---- the workgroup's partial: [N][K] weights then [N] biases, the four wavefronts summed in fixed order
  float* prow = part + (int64_t)blockIdx.x * ((int64_t)N * K + N + K);
  if (want_xsum)
    for (int i = threadIdx.x; i < K; i += blockDim.x)
      prow[(int64_t)N * K + N + i] = ((redx[0][i] + redx[1][i]) + redx[2][i]) + redx[3][i];
  for (int idx = threadIdx.x; idx < NT * KB * 4 * 64; idx += blockDim.x) {
    // item (ot, blk, r, lane): the lane's four interleaved tiles c hold four consecutive columns
    const int ln = idx & 63, r = (idx >> 6) & 3, b = (idx >> 8) % KB, ot = (idx >> 8) / KB;
    const int o = 16 * ot + 4 * (ln >> 4) + r, i0 = 64 * b + 4 * (ln & 15);
    if (o >= N || i0 >= K) continue;
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int tl = (ot * KB + b) * 4 + c;
      v[c] = ((red[0][tl][ln][r] + red[1][tl][ln][r]) + red[2][tl][ln][r]) + red[3][tl][ln][r];
    }
    *reinterpret_cast<f32x4u*>(prow + (int64_t)o * K + i0) = v;
  }
  if (threadIdx.x < NT * 16) {
    const int ot = threadIdx.x >> 4, jj = threadIdx.x & 15;
    if (16 * ot + jj < N)
      prow[(int64_t)N * K + 16 * ot + jj] = ((redb[0][ot][jj] + redb[1][ot][jj]) + redb[2][ot][jj]) + redb[3][ot][jj];
  }
}

// gW / gb <- the workgroups' partials, summed in a fixed order (deterministic): 64 columns per workgroup, wavefront w of 16
// sums the partial rows w, w + 16, ... (four independent chains), the 16 sums meet in LDS
__global__ __launch_bounds__(1024) void lin_tall_reduce_k(const float* __restrict__ part, int nparts, int64_t nw, int64_t nb,
                                                          int64_t nx, int64_t ncols, float* __restrict__ gW,
                                                          float* __restrict__ gb, float* __restrict__ gxs) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  const int64_t ld = nw + nb + nx;                      // row pitch; the first ncols columns are summed
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < ncols) {
    const float* p = part + i;
    int b = wave;
    for (; b + 48 < nparts; b += 64) {
      s0 += p[(int64_t)b * ld]; s1 += p[(int64_t)(b + 16) * ld]; s2 += p[(int64_t)(b + 32) * ld]; s3 += p[(int64_t)(b + 48) * ld];
    }
    for (; b < nparts; b += 16) s0 += p[(int64_t)b * ld];
  }
  red[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && i < ncols) {
    float s = red[0][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) s += red[w][lane];
    if (i < nw) gW[i] = s;
    else if (i < nw + nb) { if (gb) gb[i - nw] = s; }
    else gxs[i - nw - nb] = s;
  }
}

// ------------------------------------------------------------------------------------------ weight gradient alone
// gW[o][i] = sum_rows g[row][o] a[row][i], gb[o] (+)= sum_rows g[row][o] for a tall g [M x N <= 64] (row pitch ldg) and a
// tall a [M x K <= 64] (row pitch lda, any K): the first-layer products of the Monotonic backward -- d W1h = Dsum^T h and
// d b1 = colsum Dsum (MonotonicNormalizer.py:21-38 autograd; 78 400 x 64 against 78 400 x 30 at cfg4), which ran as a
// split-K tiled GEMM + its reduction + a two-stage column sum (17 + 5 + 13 us).  All eight wavefronts take the
// weight-gradient role of lin_bwd_tall_k above; partials per workgroup, summed in fixed order.
template <int NT>
__global__ __launch_bounds__(512, 1) void lin_wgrad_tall_k(const float* __restrict__ g, int ldg, const float* __restrict__ a,
                                                           int lda, float* __restrict__ partW, float* __restrict__ partB,
                                                           int M, int N, int K) {
  __shared__ f32x4 red[8][NT * 4][64];
  __shared__ float redb[8][NT][16];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  const int ntiles = (M + 15) / 16;
  const int stride = gridDim.x * 8;
  const rsrc_t ra = mkrsrc(a, (int64_t)(M - 1) * lda + K);
  f32x4 acc[NT][4];
  float bsum[NT];
#pragma unroll
  for (int ot = 0; ot < NT; ++ot) {
    bsum[ot] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[ot][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float gv[NT][4];
  f32x4 av[4];
  auto load = [&](int tl, float (*gd)[4], f32x4* ad) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * tl + 4 * q + r;
      const int rc = row < M ? row : M - 1;
#pragma unroll
      for (int ot = 0; ot < NT; ++ot) gd[ot][r] = (16 * ot + j < N) ? g[(int64_t)rc * ldg + 16 * ot + j] : 0.f;
      const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(ra, (rc * lda + 4 * j) * 4, 0, 0);
      f32x4 v = __builtin_bit_cast(f32x4, u);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = 4 * j + k < K ? v[k] : 0.f;
      ad[r] = v;
    }
  };
  int tile = blockIdx.x * 8 + wave;
  if (tile < ntiles) load(tile, gv, av);
  for (; tile < ntiles; tile += stride) {
    float gn[NT][4];
    f32x4 an[4];
    const bool more = tile + stride < ntiles;
    if (more) load(tile + stride, gn, an);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = 16 * tile + 4 * q + r < M;
#pragma unroll
      for (int ot = 0; ot < NT; ++ot) {
        const float gg = ok ? gv[ot][r] : 0.f;
        bsum[ot] += gg;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[ot][c] = mfma(gg, av[r][c], acc[ot][c]);
      }
    }
    if (more) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int ot = 0; ot < NT; ++ot) gv[ot][r] = gn[ot][r];
        av[r] = an[r];
      }
    }
  }
#pragma unroll
  for (int ot = 0; ot < NT; ++ot) {
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wave][ot * 4 + c][lane] = acc[ot][c];
    float s = bsum[ot];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (q == 0) redb[wave][ot][j] = s;
  }
  __syncthreads();
  float* pw = partW + (int64_t)blockIdx.x * ((int64_t)N * K);
  for (int idx = threadIdx.x; idx < NT * 4 * 64 * 4; idx += blockDim.x) {
    // item (ot, r, lane, c): out unit 16 ot + 4 (lane >> 4) + r, column 4 (lane & 15) + c
    const int c = idx & 3, ln = (idx >> 2) & 63, r = (idx >> 8) & 3, ot = idx >> 10;
    const int o = 16 * ot + 4 * (ln >> 4) + r, i = 4 * (ln & 15) + c;
    if (o >= N || i >= K) continue;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) v += red[w][ot * 4 + c][ln][r];
    pw[(int64_t)o * K + i] = v;
  }
  if (threadIdx.x < NT * 16) {
    const int ot = threadIdx.x >> 4, jj = threadIdx.x & 15;
    if (16 * ot + jj < N) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += redb[w][ot][jj];
      partB[(int64_t)blockIdx.x * N + 16 * ot + jj] = v;
    }
  }
}

// out[i] (+)= sum over the workgroups' partials, fixed order
__global__ __launch_bounds__(1024) void lin_wgrad_reduce_k(const float* __restrict__ partW, const float* __restrict__ partB,
                                                           int nparts, int64_t nw, int64_t nb, float* __restrict__ gW,
                                                           float* __restrict__ gb, int acc_b) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f;
  if (i < nw + nb) {
    const float* p = i < nw ? partW + i : partB + (i - nw);
    const int64_t ld = i < nw ? nw : nb;
    int b = wave;
    for (; b + 16 < nparts; b += 32) { s0 += p[(int64_t)b * ld]; s1 += p[(int64_t)(b + 16) * ld]; }
    for (; b < nparts; b += 16) s0 += p[(int64_t)b * ld];
  }
  red[wave][lane] = s0 + s1;
  __syncthreads();
  if (wave == 0 && i < nw + nb) {
    float s = red[0][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) s += red[w][lane];
    if (i < nw) gW[i] = s;
    else if (gb) gb[i - nw] = acc_b ? gb[i - nw] + s : s;
  }
}

int tall_grid(int64_t M) {
  const int64_t ntiles = (M + 15) / 16;
  const int64_t g = (ntiles + 3) / 4;
  return (int)(g < 256 ? g : 256);
}

}  // namespace

bool gnf_linear_tall_ok(int64_t M, int64_t N, int64_t K) {
  return M >= 2048 && N >= 1 && N <= 64 && K >= 4 && K <= 128 && K % 4 == 0 && M * (K > N ? K : N) * 4 < (1ll << 31);
}

// the FORWARD alone also below 2048 rows (above the small-batch kernels' 128): a level of a sampling pass is a few hundred
// rows through fc2 (128 -> 30); the tiled GEMM splits K there and pays a reduction launch (12 + 5 us against ~7)
bool gnf_linear_tall_fwd_ok(int64_t M, int64_t N, int64_t K) {
  return M > 128 && N >= 1 && N <= 64 && K >= 4 && K <= 128 && K % 4 == 0 && M * (K > N ? K : N) * 4 < (1ll << 31);
}

int64_t gnf_linear_tall_ws_floats(int64_t M, int64_t N, int64_t K) { return (int64_t)tall_grid(M) * (N * K + N + K); }

namespace {
template <int NT, int KC>
void launch_tall_fwd(dim3 g, hipStream_t s, const float* x, const float* W, const float* b, int relu, float* y, int M, int N, int K) {
  if (K == 16 * KC) hipLaunchKernelGGL((lin_fwd_tall_k<NT, KC, true>), g, dim3(256), 0, s, x, W, b, relu, y, M, N, K);
  else hipLaunchKernelGGL((lin_fwd_tall_k<NT, KC, false>), g, dim3(256), 0, s, x, W, b, relu, y, M, N, K);
}
template <int NT, int KC>
void launch_tall_bwd(dim3 g, hipStream_t s, const float* gr, const float* W, const float* a, int gated, float* gx, float* part,
                     int M, int N, int K, int want_xsum) {
  if (K == 16 * KC) hipLaunchKernelGGL((lin_bwd_tall_k<NT, KC, true>), g, dim3(512), 0, s, gr, W, a, gated, gx, part, M, N, K, want_xsum);
  else hipLaunchKernelGGL((lin_bwd_tall_k<NT, KC, false>), g, dim3(512), 0, s, gr, W, a, gated, gx, part, M, N, K, want_xsum);
}
// instantiation (out tiles NT in {2, 4}) x (k chunks KC in {1, 4, 8})
#define GNF_TALL_DISPATCH(fn, N, K, ...)                                        \
  do {                                                                          \
    if ((N) <= 32) {                                                            \
      if ((K) <= 16) fn<2, 1>(__VA_ARGS__);                                     \
      else if ((K) <= 64) fn<2, 4>(__VA_ARGS__);                                \
      else fn<2, 8>(__VA_ARGS__);                                               \
    } else {                                                                    \
      if ((K) <= 16) fn<4, 1>(__VA_ARGS__);                                     \
      else if ((K) <= 64) fn<4, 4>(__VA_ARGS__);                                \
      else fn<4, 8>(__VA_ARGS__);                                               \
    }                                                                           \
  } while (0)
}  // namespace

int gnf_linear_tall_fwd(const float* x, const float* W, const float* b, int relu, float* y, int64_t M, int64_t N, int64_t K,
                        hipStream_t s) {
  const int64_t ntiles = (M + 15) / 16;
  int64_t grid = (ntiles + 3) / 4;
  // one 4-wavefront workgroup per CU: every wavefront first pulls the whole weight into registers (15 KB at 30 x 128), so
  // more, shorter-lived wavefronts re-read it more often than they add loads in flight (78 400 x 128 -> 30, HIP events
  // around the entry point: 128 workgroups 23.8 us, 256: 16.6, 384: 22.3, 512: 19.5, 768: 20.7, 1225: 23.4)
  if (grid > 256) grid = 256;
  const dim3 g((unsigned)grid);
  GNF_TALL_DISPATCH(launch_tall_fwd, N, K, g, s, x, W, b, relu, y, (int)M, (int)N, (int)K);
  GNF_LAUNCH_CHECK();
  return 0;
}

int gnf_linear_tall_bwd(const float* g, const float* W, const float* a, const float* gate, float* gx, float* gW, float* gb,
                        float* gxsum, int64_t M, int64_t N, int64_t K, float* ws, hipStream_t s) {
  if (gate && gate != a) return GNF_EINVAL;          // the ReLU gate of a layer IS its input
  const int grid = tall_grid(M);
  const dim3 gd((unsigned)grid);
  GNF_TALL_DISPATCH(launch_tall_bwd, N, K, gd, s, g, W, a, gate ? 1 : 0, gx, ws, (int)M, (int)N, (int)K, gxsum ? 1 : 0);
  GNF_LAUNCH_CHECK();
  const int64_t nw = N * K, ncols = nw + N + (gxsum ? K : 0);
  hipLaunchKernelGGL(lin_tall_reduce_k, dim3((unsigned)((ncols + 63) / 64)), dim3(1024), 0, s, ws, grid, nw, (int64_t)N, K, ncols,
                     gW, gb, gxsum);
  GNF_LAUNCH_CHECK();
  return 0;
}

bool gnf_linear_tall_wgrad_ok(int64_t M, int64_t N, int64_t K, int64_t ldg, int64_t lda) {
  return M >= 2048 && N >= 1 && N <= 64 && K >= 1 && K <= 64 && ldg >= N && lda >= K && M * (lda > ldg ? lda : ldg) * 4 < (1ll << 31);
}

int gnf_linear_tall_wgrad_parts(int64_t M) { return tall_grid(M); }

int gnf_linear_tall_wgrad(const float* g, int64_t ldg, const float* a, int64_t lda, float* gW, float* gb, int accumulate_b,
                          int64_t M, int64_t N, int64_t K, float* partW, float* partB, hipStream_t s) {
  const int grid = tall_grid(M);
  const dim3 gd((unsigned)grid), blk(512);
  if (N <= 32) hipLaunchKernelGGL((lin_wgrad_tall_k<2>), gd, blk, 0, s, g, (int)ldg, a, (int)lda, partW, partB, (int)M, (int)N, (int)K);
  else hipLaunchKernelGGL((lin_wgrad_tall_k<4>), gd, blk, 0, s, g, (int)ldg, a, (int)lda, partW, partB, (int)M, (int)N, (int)K);
  GNF_LAUNCH_CHECK();
  const int64_t nw = N * K;
  hipLaunchKernelGGL(lin_wgrad_reduce_k, dim3((unsigned)((nw + N + 63) / 64)), dim3(1024), 0, s, partW, partB, grid, nw, N, gW,
                     gb, accumulate_b);
  GNF_LAUNCH_CHECK();
  return 0;
}
