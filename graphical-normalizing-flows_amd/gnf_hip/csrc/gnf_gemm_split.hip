// fp32 GEMM on the bf16 matrix pipe: every fp32 operand is split EXACTLY into three bf16 numbers
//     x = hi + mid + lo,   hi = rne_bf16(x), mid = rne_bf16(x - hi), lo = x - hi - mid   (24 significand bits = 3 x 8)
// and a product a*b is formed from its six leading cross terms, each EXACT in the fp32 accumulator of
// v_mfma_f32_16x16x32_bf16 (8 x 8 significand bits):
//     hi*hi  |  hi*mid + mid*hi  |  hi*lo + lo*hi + mid*mid          (dropped: mid*lo, lo*mid, lo*lo  <=  2^-25 |a b|)
// The three magnitude classes are accumulated in SEPARATE accumulators and added once at the end, so the small terms are
// never rounded against the large running sum.  bf16 MFMA runs at 16x the rate of v_mfma_f32_16x16x4_f32 and, unlike it,
// not on the vector ALU: six of them per K = 32 replace eight fp32 MFMAs of K = 4.
//
// Round 6 experiment asked for by the review (item 2): `tools/split_bf16_error.py` measures the error of this form and of
// the fp32-MFMA kernels against an fp64 product on the cfg4 fc1 operands and on random shapes
// (profiles/r06_split_bf16_error.txt).  Reference: models/MLP.py:44 (MNISTCNN.fc1) and its autograd.
//
// gemm_split_k: the GENERAL kernel (any element strides, any M, N, K): 64 x 64 tile per 4-wavefront workgroup, K-slabs of 32
// split on the way into LDS.  Non-finite inputs are outside its contract (inf * 0 cross terms give NaN where fp32 gives inf).
#include "gnf_common.h"
#include "gnf_gemm.h"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

// packed pair of RNE bf16: lo half = bf16(a), hi half = bf16(b)
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// x -> (hi, mid, lo) as bf16 bit patterns
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const unsigned ph = cvt_pk_bf16(x, 0.f) & 0xffffu;
  const float r1 = x - __uint_as_float(ph << 16);                   // exact (Sterbenz-like: hi is x rounded to 8 bits)
  const unsigned pm = cvt_pk_bf16(r1, 0.f) & 0xffffu;
  const float r2 = r1 - __uint_as_float(pm << 16);                  // exact; <= 8 significant bits are left
  const unsigned pl = cvt_pk_bf16(r2, 0.f) & 0xffffu;
  h = (unsigned short)ph; m = (unsigned short)pm; l = (unsigned short)pl;
}

constexpr int TS = 64;          // tile edge
constexpr int KS = 32;          // K-slab = one bf16 MFMA
constexpr int LP = KS + 8;      // row pitch in bf16 (80 B)

struct SplitArgs {
  const float* A; int64_t sam, sak;
  const float* B; int64_t sbk, sbn;
  float* C; int64_t scm, scn;
  const float* bias; int relu;
  int64_t M, N, K;
  int64_t k_per_split, c_split_stride;
};

template <int CLASSES>
__global__ __launch_bounds__(256) void gemm_split_k(SplitArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned short sA[3][TS][LP];
  __shared__ __attribute__((aligned(16))) unsigned short sB[3][TS][LP];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  const int64_t m0 = (int64_t)blockIdx.y * TS, n0 = (int64_t)blockIdx.x * TS;
  const int64_t kb = (int64_t)blockIdx.z * g.k_per_split;
  int64_t ke = kb + g.k_per_split;
  if (ke > g.K) ke = g.K;
  f32x4 acc[CLASSES][2][2];
#pragma unroll
  for (int c = 0; c < CLASSES; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool a_kc = g.sak == 1, b_kc = g.sbk == 1;
  for (int64_t k0 = kb; k0 < ke; k0 += KS) {
#pragma unroll
    for (int i = 0; i < TS * KS / 256; ++i) {
      const int e = t + 256 * i;
      {
        const int r = a_kc ? e >> 5 : e & (TS - 1), k = a_kc ? e & (KS - 1) : e >> 6;
        const int64_t m = m0 + r, kk = k0 + k;
        const float x = (m < g.M && kk < ke) ? g.A[m * g.sam + kk * g.sak] : 0.f;
        split3(x, sA[0][r][k], sA[1][r][k], sA[2][r][k]);
      }
      {
        const int r = b_kc ? e >> 5 : e & (TS - 1), k = b_kc ? e & (KS - 1) : e >> 6;
        const int64_t n = n0 + r, kk = k0 + k;
        const float x = (n < g.N && kk < ke) ? g.B[kk * g.sbk + n * g.sbn] : 0.f;
        split3(x, sB[0][r][k], sB[1][r][k], sB[2][r][k]);
      }
    }
    __syncthreads();
    bf16x8 a[3][2], b[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[p][i] = *reinterpret_cast<const bf16x8*>(&sA[p][32 * wm + 16 * i + (lane & 15)][8 * (lane >> 4)]);
        b[p][i] = *reinterpret_cast<const bf16x8*>(&sB[p][32 * wn + 16 * i + (lane & 15)][8 * (lane >> 4)]);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4& big = acc[0][i][j];
        f32x4& mid = acc[CLASSES > 1 ? 1 : 0][i][j];
        f32x4& sml = acc[CLASSES > 2 ? 2 : (CLASSES > 1 ? 1 : 0)][i][j];
        sml = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2][i], b[0][j], sml, 0, 0, 0);     // lo * hi
        sml = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[2][j], sml, 0, 0, 0);     // hi * lo
        sml = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][i], b[1][j], sml, 0, 0, 0);     // mid * mid
        mid = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][i], b[0][j], mid, 0, 0, 0);     // mid * hi
        mid = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[1][j], mid, 0, 0, 0);     // hi * mid
        big = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[0][j], big, 0, 0, 0);     // hi * hi
      }
    __syncthreads();
  }
  float* C = g.C + (int64_t)blockIdx.z * g.c_split_stride;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + 32 * wn + 16 * j + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + 32 * wm + 16 * i + 4 * (lane >> 4) + r;
        if (m < g.M && n < g.N) {
          float v = acc[0][i][j][r];
          if constexpr (CLASSES == 2) v += acc[1][i][j][r];
          if constexpr (CLASSES == 3) v += acc[1][i][j][r] + acc[2][i][j][r];
          if (g.bias) v += g.bias[n];
          if (g.relu) v = fmaxf(v, 0.f);
          C[m * g.scm + n * g.scn] = v;
        }
      }
    }
}


// ------------------------------------------------------------------------------------------------ dedicated kernels
// x0, x1 -> packed (hi, mid, lo) pairs: 11 VALU instructions per two elements
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
  const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = cvt_pk_bf16(q0, q1);
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// The small operand of a dedicated kernel (the fc1 weight in either orientation), split ONCE per call into FRAGMENT-MAJOR
// bf16 planes: planes[p][slab][ntile][lane] = the 16 bytes lane `lane` of a wavefront feeds to v_mfma_f32_16x16x32_bf16 as
// the B operand of (K-slab `slab`, 16-column tile `ntile`): n = 16 ntile + (lane & 15), k = 32 slab + 8 (lane >> 4) .. + 7.
// A fragment is then ONE fully coalesced 1-KB load per wavefront from L2.  Columns n >= N and slabs past K read as zeros.
__global__ __launch_bounds__(256) void split_pack_b_k(const float* __restrict__ B, int64_t sbk, int64_t sbn, int N, int K,
                                                     int ntiles, int nslab, u32x4* __restrict__ out) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)nslab * ntiles * 64;
  if (gid >= total) return;
  const int lane = (int)(gid & 63);
  const int64_t tl = gid >> 6;
  const int nt = (int)(tl % ntiles), slab = (int)(tl / ntiles);
  const int n = 16 * nt + (lane & 15), k = 32 * slab + 8 * (lane >> 4);
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = (n < N && k + e < K) ? B[(int64_t)(k + e) * sbk + (int64_t)n * sbn] : 0.f;
  u32x4 h, m, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned a, b, c;
    split3_pair(x[2 * e], x[2 * e + 1], a, b, c);
    h[e] = a; m[e] = b; l[e] = c;
  }
  out[gid] = h;
  out[total + gid] = m;
  out[2 * total + gid] = l;
}

// ---- tall M, N <= 128, both operands k-contiguous (MNISTCNN.fc1 forward: 78 400 x 128 x 2304, MLP.py:44) -------------------
// One 12-wavefront workgroup per 160-row block (490 blocks at cfg4: two rounds of the CUs, 95.7 % filled), THREE wavefronts
// per SIMD in two roles:
//   * wavefronts 0..7, the MFMA role: wavefront (wm, wn) of a 2 x 4 grid owns 5 x 2 tiles of 16 x 16; per K-slab of 32 it
//     reads 5 x 3 A fragments (ds_read_b128 from the three bf16 planes of the current LDS stage, row pitch 96 B:
//     conflict-free) and issues 60 MFMAs; its B fragments come pre-split and fragment-major from L2 (split_pack_b_k), two
//     slabs ahead in a ring of two register sets.  No VALU work at all in this role.
//   * wavefronts 8..11 (one per SIMD), the LOADER role: A K-slabs arrive as fp32 float4s from HBM (each element ONCE, two to
//     four slabs ahead, in pairs of slabs: an HBM request under load takes several slab times), are split into (hi, mid, lo) and written into the
//     OTHER LDS stage while the MFMA wavefronts of the same SIMD keep the matrix pipe busy -- the split (5.5 VALU
//     instructions per element) runs in the shadow of the bf16 MFMAs, which do not use the vector ALU.
// Two LDS stages, ONE barrier per slab.  Two accumulator classes: hi*hi | the five cross terms.
// (The first version -- eight wavefronts doing everything, one slab ahead -- spent 68 % of its wave cycles in s_waitcnt, MFMA
// pipe busy 0.32: 0.34 ms; deeper rings 0.28 ms; see profiles/r06_split_bf16_ab.txt.)
// Row pitch 96 B = 6 slots of 16 B: ds_read_b128 is serviced in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, ...:
// MI355X_MICROARCH.md, LDS), so a fragment read touches rows {0-3, 12-15} at one k-quarter and rows {4-11} at the next; with
// the 80-B pitch that suits contiguous groups half of all LDS cycles were bank conflicts (SQ_LDS_BANK_CONFLICT / IDX_ACTIVE
// 0.50); 6 r + kq (+1 for rows 4-11) is distinct mod 16 over every group.
constexpr int TL_BM = 160, TL_PITCH = 96, TL_PLANE = TL_BM * TL_PITCH, TL_STAGE = 3 * TL_PLANE;
constexpr int TL_LDS = 2 * TL_STAGE;                                       // 92 160 B
constexpr int TL_THREADS = 768;

struct TallArgs {
  const float* A; int64_t sam;
  const u32x4* Bp;                  // [3][nslab][8][64]
  float* C; int64_t scm;
  const float* bias; int relu;
  int64_t M; int N, nslab;
};

__global__ __launch_bounds__(TL_THREADS) void gemm_split_tall_k(TallArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int64_t m0 = (int64_t)blockIdx.x * TL_BM;
  if (wave >= 8) {
    // ------------------------------------------------------------------------------------------------ loader role
    // 160 rows x 8 float4 chunks = 1280 chunks per slab, five per thread.  No load sits under a branch (hipcc's waitcnt
    // pass answers a predicated load with vmcnt(0) at the join): rows past M are clamped to the last row -- a row of C
    // depends on its own row of A only, and those rows are never stored.
    const int lt = t - 512;
    // (s_setprio(3) here -- the MFMA wavefronts of a SIMD win nearly every VALU issue slot, tools/mfma_bf16_rate.hip -- measured
    // 2 % SLOWER: what the loader gains the matrix pipe loses)
    const float* arow[5];
    int aoff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int q = lt + 256 * i, row = q >> 3, c = q & 7;
      const int64_t m = m0 + row < g.M ? m0 + row : g.M - 1;
      arow[i] = g.A + m * g.sam + 4 * c;
      aoff[i] = row * TL_PITCH + 8 * c;
    }
    f32x4 areg[4][5];
    auto load_a = [&](int slab, f32x4 (&a)[5]) {                  // (prologue; the tail of the loop re-requests the last slab)
      const int sl = slab < g.nslab ? slab : g.nslab - 1;
#pragma unroll
      for (int i = 0; i < 5; ++i) a[i] = *reinterpret_cast<const f32x4*>(arow[i] + 32 * sl);
    };
    auto store_a = [&](int buf, const f32x4 (&a)[5]) {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        unsigned h0, m0_, l0, h1, m1, l1;
        split3_pair(a[i][0], a[i][1], h0, m0_, l0);
        split3_pair(a[i][2], a[i][3], h1, m1, l1);
        unsigned char* base = lds + buf * TL_STAGE + aoff[i];
        *reinterpret_cast<u32x2*>(base) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(base + TL_PLANE) = u32x2{m0_, m1};
        *reinterpret_cast<u32x2*>(base + 2 * TL_PLANE) = u32x2{l0, l1};
      }
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) load_a(u, areg[u]);
    store_a(0, areg[0]);
    __syncthreads();
    for (int s0 = 0; s0 < g.nslab; s0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // Slabs are requested in PAIRS, every odd iteration, into the two slots the last two stores freed: the two requests of a
        // thread are the adjacent 128-B halves of one 256-B run of its row, back to back -- the second one finds the DRAM page open
        // (one slab per iteration, four ahead: 0.305-0.310 ms; pairs: 0.276-0.288, same box, profiles/r06_split_bf16_ab.txt)
        if (u & 1) {
          const int sa = s0 + u + 3 < g.nslab ? s0 + u + 3 : g.nslab - 1, sb = s0 + u + 4 < g.nslab ? s0 + u + 4 : g.nslab - 1;
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            areg[u - 1][i] = *reinterpret_cast<const f32x4*>(arow[i] + 32 * sa);
            areg[u][i] = *reinterpret_cast<const f32x4*>(arow[i] + 32 * sb);
          }
        }
        store_a((u + 1) & 1, areg[(u + 1) & 3]);                // slab s + 1 into the stage the MFMA role is NOT reading
        __syncthreads();
      }
    }
    return;
  }
  // ---------------------------------------------------------------------------------------------------- MFMA role
  const int wm = wave >> 2, wn = wave & 3;
  const int64_t plane_sz = (int64_t)g.nslab * 8 * 64;                       // uint4 entries per plane
  f32x4 acc[2][5][2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 breg[2][2][3];
  auto load_b = [&](int slab, u32x4 (&b)[2][3]) {
    const int sl = slab < g.nslab ? slab : g.nslab - 1;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) b[j][p] = g.Bp[p * plane_sz + ((int64_t)sl * 8 + 2 * wn + j) * 64 + lane];
  };
  load_b(0, breg[0]);
  load_b(1, breg[1]);
  __syncthreads();
  const int frag = (16 * (5 * wm) + (lane & 15)) * TL_PITCH + 16 * (lane >> 4);
  for (int s0 = 0; s0 < g.nslab; s0 += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const unsigned char* stage = lds + u * TL_STAGE + frag;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const bf16x8 ah = as_bf16x8(*reinterpret_cast<const u32x4*>(stage + i * 16 * TL_PITCH));
        const bf16x8 am = as_bf16x8(*reinterpret_cast<const u32x4*>(stage + i * 16 * TL_PITCH + TL_PLANE));
        const bf16x8 al = as_bf16x8(*reinterpret_cast<const u32x4*>(stage + i * 16 * TL_PITCH + 2 * TL_PLANE));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 bh = as_bf16x8(breg[u][j][0]), bm = as_bf16x8(breg[u][j][1]), bl = as_bf16x8(breg[u][j][2]);
          f32x4 r = acc[1][i][j];
          r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, r, 0, 0, 0);
          r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, r, 0, 0, 0);
          r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, r, 0, 0, 0);
          r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, r, 0, 0, 0);
          r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, r, 0, 0, 0);
          acc[1][i][j] = r;
          acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[0][i][j], 0, 0, 0);
        }
      }
      load_b(s0 + u + 2, breg[u]);
      __syncthreads();
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = 16 * (2 * wn + j) + (lane & 15);
    const float bv = (g.bias && n < g.N) ? g.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + 16 * (5 * wm + i) + 4 * (lane >> 4) + r;
        if (m < g.M && n < g.N) {
          float v = acc[0][i][j][r] + acc[1][i][j][r] + bv;
          if (g.relu) v = fmaxf(v, 0.f);
          g.C[m * g.scm + n] = v;
        }
      }
  }
}

// ---- tall M, K = 128, wide N (MNISTCNN.fc1 data gradient: 78 400 x 2304 x 128, autograd of MLP.py:44) ------------------------
// C[M x N] = A[M x 128] B[128 x N].  One 8-wavefront workgroup per 160-row block: the block's A rows (80 KB of fp32) are
// split ONCE into three LDS planes [160][128] (row pitch 288 B = 18 slots: conflict-free for the lane groups of ds_read_b128), then the workgroup walks the
// N / 128 column tiles; B fragments (the small operand, pre-split and fragment-major: split_pack_b_k) stream from L2 two
// K-slabs ahead.  The MFMA operands are SWAPPED (B fragment as the row operand): a lane then holds four CONSECUTIVE columns
// of one output row -- one global_store_dwordx4 per tile and lane instead of four dword stores.
constexpr int WD_BM = 160, WD_K = 128, WD_PITCH = 2 * WD_K + 32, WD_PLANE = WD_BM * WD_PITCH, WD_LDS = 3 * WD_PLANE;   // 138 240 B

struct WideArgs {
  const float* A; int64_t sam;
  const u32x4* Bp;                  // [3][4][N / 16][64]
  float* C; int64_t scm;
  float* dummy;                     // [16][N]: where the lanes of rows >= M store (no store sits under a branch, see below)
  int64_t M; int N;
};

__global__ __launch_bounds__(512) void gemm_split_wide_k(WideArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wm = wave >> 2, wn = wave & 3;
  const int64_t m0 = (int64_t)blockIdx.x * WD_BM;
  const int ntiles = g.N / 16, nct = g.N / 128;
  const int64_t plane_sz = (int64_t)4 * ntiles * 64;
  // ---- prologue: the block's A rows -> (hi, mid, lo) planes.  160 rows x 32 float4 chunks = 5120 chunks, ten per thread
  {
    f32x4 a[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int q = t + 512 * i, row = q >> 5, c = q & 31;
      const int64_t m = m0 + row < g.M ? m0 + row : g.M - 1;
      a[i] = *reinterpret_cast<const f32x4*>(g.A + m * g.sam + 4 * c);
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int q = t + 512 * i, row = q >> 5, c = q & 31;
      unsigned h0, m0_, l0, h1, m1, l1;
      split3_pair(a[i][0], a[i][1], h0, m0_, l0);
      split3_pair(a[i][2], a[i][3], h1, m1, l1);
      unsigned char* base = lds + row * WD_PITCH + 8 * c;
      *reinterpret_cast<u32x2*>(base) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(base + WD_PLANE) = u32x2{m0_, m1};
      *reinterpret_cast<u32x2*>(base + 2 * WD_PLANE) = u32x2{l0, l1};
    }
  }
  u32x4 breg[2][2][3];
  auto load_b = [&](int step, u32x4 (&b)[2][3]) {              // step = 4 * column tile + K-slab
    const int st = step < 4 * nct ? step : 4 * nct - 1;
    const int ct = st >> 2, slab = st & 3;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 3; ++p) b[j][p] = g.Bp[p * plane_sz + ((int64_t)slab * ntiles + 8 * ct + 2 * wn + j) * 64 + lane];
  };
  load_b(0, breg[0]);
  load_b(1, breg[1]);
  __syncthreads();
  int afrag_off = (16 * (5 * wm) + (lane & 15)) * WD_PITCH + 16 * (lane >> 4);
  // Stores under a branch (rows past M in the last block) made hipcc's waitcnt pass answer with vmcnt(0) in EVERY K-slab of
  // the tile loop -- i.e. each slab waited for the stores of the previous tile and for the B fragments it had just requested.
  // Lanes of rows >= M store into a dummy strip of the workspace instead: the loop body is branch-free.
  float* crow[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int64_t m = m0 + 16 * (5 * wm + i) + (lane & 15);
    crow[i] = (m < g.M ? g.C + m * g.scm : g.dummy + (int64_t)(lane & 15) * g.N) + 32 * wn + 4 * (lane >> 4);
  }
  for (int ct = 0; ct < nct; ++ct) {
    f32x4 acc[2][5][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the A planes do not change between column tiles, and hipcc knows it: without this it hoists all 60 fragment reads of
    // a tile out of the loop (240 registers, 190 of them spilled)
    asm volatile("" : "+v"(afrag_off));
    const unsigned char* afrag = lds + afrag_off;
#pragma unroll
    for (int slab = 0; slab < 4; ++slab) {
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const unsigned char* ap = afrag + i * 16 * WD_PITCH + 64 * slab;
        const bf16x8 ah = as_bf16x8(*reinterpret_cast<const u32x4*>(ap));
        const bf16x8 am = as_bf16x8(*reinterpret_cast<const u32x4*>(ap + WD_PLANE));
        const bf16x8 al = as_bf16x8(*reinterpret_cast<const u32x4*>(ap + 2 * WD_PLANE));
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 bh = as_bf16x8(breg[slab & 1][j][0]), bm = as_bf16x8(breg[slab & 1][j][1]), bl = as_bf16x8(breg[slab & 1][j][2]);
          f32x4 r = acc[1][i][j];                            // (B fragment first: the tile comes out transposed, see above)
          r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, r, 0, 0, 0);
          r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, r, 0, 0, 0);
          r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am, r, 0, 0, 0);
          r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am, r, 0, 0, 0);
          r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah, r, 0, 0, 0);
          acc[1][i][j] = r;
          acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc[0][i][j], 0, 0, 0);
        }
      }
      // (pinned: left to itself the machine scheduler sinks these requests to just in front of their first use two slabs
      // later and waits vmcnt(0) there -- an L2 round trip per fragment in the dependent chain of every slab)
      __builtin_amdgcn_sched_barrier(0);
      load_b(4 * ct + slab + 2, breg[slab & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#ifdef GNF_WD_NOSTORE
        if (acc[0][i][j][0] == 123.f)
#endif
        *reinterpret_cast<f32x4*>(crow[i] + 128 * ct + 16 * j) = acc[0][i][j] + acc[1][i][j];
  }
}

// ---- M <= 128, wide N, very long K, both operands k-major (MNISTCNN.fc1 weight gradient: 128 x 2304 x 78 400) ------------------
// C[M x N] = A[K x M]^T B[K x N].  The SMALL operand A (the gated cotangent of fc1's output: 40 MB against 722 MB) is pre-split
// into fragment-major planes by split_pack_b_k -- which also does its transposition: a fragment holds 8 consecutive k of one
// m -- and streams from L2 as the other two kernels' B fragments do.  One 8-wavefront workgroup per (128-column tile, K range):
// stages of 64 k-rows of B arrive as four fp32 float4s per thread (rows 4 kg .. +3 at columns 4 ng .. +3: a wave-level request
// is eight rows x 128 B), and the (hi, mid, lo) split TRANSPOSES for free -- v_cvt_pk_bf16_f32 packs rows (k, k+1) of one
// column -- into planes [n][64 k] (row pitch 160 B: conflict-free ds_read_b128 fragments; the ds_write_b64 of a 16-lane
// group are 2-way, hidden behind their 6-cycle issue).  Two LDS stages, one barrier per stage; wavefront (wm, wn) of a 4 x 2
// grid owns 2 x 4 tiles; the operands are swapped as in the wide kernel (a lane holds four consecutive columns: float4
// stores); per-range partials are summed in a fixed order by split_reduce_k.
// Rows k >= K read as zeros from A's planes (the pack pads) while B's row index is clamped: B must be finite.
constexpr int KM_BN = 128, KM_SK = 64, KM_PITCH = 160, KM_PLANE = KM_BN * KM_PITCH, KM_STAGE = 3 * KM_PLANE;
constexpr int KM_LDS = 2 * KM_STAGE;                                       // 122 880 B

struct KmArgs {
  const u32x4* Ap;                  // [3][nslab][8][64]
  const float* B; int64_t sbk;
  float* part;                      // [splits][M][N]
  int64_t M, N, K;
  int nslab;                        // K-slabs of 32 in A's planes (padded to whole K ranges)
  int stages_per_split;             // stages of 64 k per K range (even)
};

__global__ __launch_bounds__(512) void gemm_split_kmajor_k(KmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wm = wave >> 1, wn = wave & 1;
  const int ntn = (int)(g.N / KM_BN);
  const int nt0 = blockIdx.x % ntn, z = blockIdx.x / ntn;
  const int64_t n0 = (int64_t)nt0 * KM_BN;
  const int64_t st0 = (int64_t)z * g.stages_per_split;                     // first stage (of 64 k) of this K range
  const int64_t plane_sz = (int64_t)g.nslab * 8 * 64;
  // B loader: lane (kg, ng) of wavefront w: k rows 4 (8 (w & 1) + (lane & 7)) .. + 3, columns 4 (8 (w >> 1) + (lane >> 3)) .. + 3
  const int kg = 8 * (wave & 1) + (lane & 7), ng = 8 * (wave >> 1) + (lane >> 3);
  const float* bcol = g.B + n0 + 4 * ng;
  const int woff = (4 * ng) * KM_PITCH + 8 * kg;                          // byte offset of (n = 4 ng, k = 4 kg) in a plane
  f32x4 breg[2][4];
  auto load_b = [&](int64_t stage, f32x4 (&b)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t k = stage * KM_SK + 4 * kg + i;
      k = k < g.K ? k : g.K - 1;                                           // (A's planes are zero there)
      b[i] = *reinterpret_cast<const f32x4*>(bcol + k * g.sbk);
    }
  };
  auto store_b = [&](int buf, const f32x4 (&b)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned h0, m0, l0, h1, m1, l1;
      split3_pair(b[0][e], b[1][e], h0, m0, l0);
      split3_pair(b[2][e], b[3][e], h1, m1, l1);
      unsigned char* base = lds + buf * KM_STAGE + woff + e * KM_PITCH;
      *reinterpret_cast<u32x2*>(base) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(base + KM_PLANE) = u32x2{m0, m1};
      *reinterpret_cast<u32x2*>(base + 2 * KM_PLANE) = u32x2{l0, l1};
    }
  };
  u32x4 areg[2][2][3];
  auto load_a = [&](int64_t slab, u32x4 (&a)[2][3]) {
    const int64_t sl = slab < g.nslab ? slab : g.nslab - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[i][p] = g.Ap[p * plane_sz + (sl * 8 + 2 * wm + i) * 64 + lane];
  };
  f32x4 acc[2][2][4];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  load_b(st0, breg[0]);
  load_b(st0 + 1, breg[1]);
  load_a(2 * st0, areg[0]);
  load_a(2 * st0 + 1, areg[1]);
  store_b(0, breg[0]);
  __syncthreads();
  const int frag = (16 * (4 * wn) + (lane & 15)) * KM_PITCH + 16 * (lane >> 4);
  for (int s0 = 0; s0 < g.stages_per_split; s0 += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int64_t st = st0 + s0 + u;
      __builtin_amdgcn_sched_barrier(0);
      load_b(st + 2, breg[u]);                                   // slot u held stage st: in LDS since the previous iteration
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* stage = lds + u * KM_STAGE + frag;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned char* bp = stage + j * 16 * KM_PITCH + 64 * ks;
          const bf16x8 bh = as_bf16x8(*reinterpret_cast<const u32x4*>(bp));
          const bf16x8 bm = as_bf16x8(*reinterpret_cast<const u32x4*>(bp + KM_PLANE));
          const bf16x8 bl = as_bf16x8(*reinterpret_cast<const u32x4*>(bp + 2 * KM_PLANE));
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const bf16x8 ah = as_bf16x8(areg[ks][i][0]), am = as_bf16x8(areg[ks][i][1]), al = as_bf16x8(areg[ks][i][2]);
            f32x4 r = acc[1][i][j];                              // (B fragment first: a lane holds four consecutive columns)
            r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, al, r, 0, 0, 0);
            r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, ah, r, 0, 0, 0);
            r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, am, r, 0, 0, 0);
            r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, am, r, 0, 0, 0);
            r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm, ah, r, 0, 0, 0);
            acc[1][i][j] = r;
            acc[0][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah, acc[0][i][j], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_a(2 * (st + 1) + ks, areg[ks]);                     // the same K-slab of the NEXT stage
        __builtin_amdgcn_sched_barrier(0);
      }
      store_b((u + 1) & 1, breg[(u + 1) & 1]);                   // stage st + 1 into the other LDS stage
      __syncthreads();
    }
  }
  float* part = g.part + (int64_t)z * g.M * g.N;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int64_t m = 16 * (2 * wm + i) + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (m < g.M) *reinterpret_cast<f32x4*>(part + m * g.N + n0 + 16 * (4 * wn + j) + 4 * (lane >> 4)) = acc[0][i][j] + acc[1][i][j];
  }
}

// C[i] = sum_z part[z][i] in ascending z (deterministic), float4
__global__ __launch_bounds__(256) void split_reduce_k(const float* __restrict__ part, int splits, int64_t n4, int64_t stride4,
                                                      f32x4* __restrict__ C, int64_t scm4, int64_t n4_per_row) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(part) + i;
  f32x4 s = p[0];
  for (int z = 1; z < splits; ++z) s += p[(int64_t)z * stride4];
  C[(i / n4_per_row) * scm4 + i % n4_per_row] = s;
}

}  // namespace

static thread_local const char* g_split_last = "";
extern "C" const char* gnf_gemm_split_last_kernel(void) { return g_split_last; }

static bool tall_eligible(int64_t sam, int64_t sak, int64_t sbk, int64_t scn, const float* A, int64_t M, int64_t N, int64_t K) {
  return sak == 1 && sbk == 1 && scn == 1 && N <= 128 && N > 64 && K % (4 * KS) == 0 && K >= 8 * KS && M >= 64 * TL_BM &&
         sam % 4 == 0 && ((uintptr_t)A & 15) == 0;
}

static bool wide_eligible(int64_t sam, int64_t sak, int64_t scm, int64_t scn, const float* A, const float* C, int64_t M, int64_t N,
                          int64_t K) {
  return sak == 1 && scn == 1 && K == WD_K && N % 128 == 0 && N >= 512 && M >= 64 * WD_BM && sam % 4 == 0 && scm % 4 == 0 &&
         (((uintptr_t)A | (uintptr_t)C) & 15) == 0 && N < (1 << 20);
}

static bool kmajor_shape(int64_t M, int64_t N, int64_t K) { return M <= 128 && M >= 16 && N % KM_BN == 0 && N >= 512 && K >= 16384; }
static int kmajor_splits(int64_t N, int64_t K) {
  int64_t s = 256 / (N / KM_BN);
  if (s > K / 1024) s = K / 1024;
  return (int)(s < 1 ? 1 : s);
}
static int64_t kmajor_stages(int64_t N, int64_t K) {                        // stages of 64 k per K range, even
  const int splits = kmajor_splits(N, K);
  return ((K + splits - 1) / splits + 127) / 128 * 2;
}
static bool kmajor_eligible(int64_t sam, int64_t sbk, int64_t sbn, int64_t scm, int64_t scn, const float* B, const float* C, int64_t M,
                            int64_t N, int64_t K) {
  return kmajor_shape(M, N, K) && sam == 1 && sbn == 1 && scn == 1 && sbk % 4 == 0 && scm % 4 == 0 &&
         (((uintptr_t)B | (uintptr_t)C) & 15) == 0;
}

// bytes of workspace the dedicated kernels of this shape want (0: only the general kernel applies, no workspace)
extern "C" int64_t gnf_gemm_split_ws_bytes(int64_t M, int64_t N, int64_t K) {
  int64_t w = 0;
  if (kmajor_shape(M, N, K)) {                                              // k-major: A planes + the partials of the K ranges
    const int64_t splits = kmajor_splits(N, K), nslab = splits * kmajor_stages(N, K) * 2;
    w = 3 * nslab * 8 * 64 * 16 + splits * M * N * 4;
  }
  if (K == WD_K && N % 128 == 0 && N >= 512 && M >= 64 * WD_BM) w = 3 * 4 * (N / 16) * 64 * 16 + 16 * N * 4;         // wide: B planes + dummy rows
  if (N <= 128 && N > 64 && K % (4 * KS) == 0 && K >= 8 * KS && M >= 64 * TL_BM) w = 3 * (K / KS) * 8 * 64 * 16;      // tall: B planes
  return w;
}

// 1 unless GNF_TRUE_F32=1 is set in the environment (build / run switch: every contraction then stays on v_mfma_f32_*)
extern "C" int gnf_gemm_split_enabled(void) {
  static const int on = !(getenv("GNF_TRUE_F32") && getenv("GNF_TRUE_F32")[0] == '1');
  return on;
}

// classes 1..3: the general kernel with that many accumulator classes (measurement); classes 0: the product's choice --
// a dedicated kernel when the shape has one and `ws` holds gnf_gemm_split_ws_bytes, else the general kernel (3 classes)
extern "C" int gnf_gemm_split_bf16(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                                   float* C, int64_t scm, int64_t scn, const float* bias, int relu,
                                   int64_t M, int64_t N, int64_t K, int classes, int splits, int64_t c_split_stride,
                                   void* ws, int64_t ws_bytes, gnf_stream_t stream) {
  if (M < 0 || N < 0 || K < 0 || classes < 0 || classes > 3 || splits < 1) return GNF_EINVAL;
  if (M == 0 || N == 0) return 0;
  if (!A || !B || !C) return GNF_EINVAL;
  if (splits > 1 && (bias || relu)) return GNF_EINVAL;          // partials: the caller sums C + z * c_split_stride, z < splits
  hipStream_t s = (hipStream_t)stream;
  if (classes == 0 && splits == 1 && ws && ws_bytes >= gnf_gemm_split_ws_bytes(M, N, K) && gnf_gemm_split_ws_bytes(M, N, K) > 0) {
    if (tall_eligible(sam, sak, sbk, scn, A, M, N, K)) {
      const int nslab = (int)(K / KS);
      const int64_t frags = (int64_t)nslab * 8 * 64;
      hipLaunchKernelGGL(split_pack_b_k, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, s, B, sbk, sbn, (int)N, (int)K, 8,
                         nslab, (u32x4*)ws);
      GNF_LAUNCH_CHECK();
      TallArgs t{A, sam, (const u32x4*)ws, C, scm, bias, relu, M, (int)N, nslab};
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_tall_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                TL_LDS);
      hipLaunchKernelGGL(gemm_split_tall_k, dim3((unsigned)((M + TL_BM - 1) / TL_BM)), dim3(TL_THREADS), TL_LDS, s, t);
      GNF_LAUNCH_CHECK();
      g_split_last = "gemm_split_tall_k";
      return 0;
    }
    if (!bias && !relu && kmajor_eligible(sam, sbk, sbn, scm, scn, B, C, M, N, K)) {
      const int splits = kmajor_splits(N, K);
      const int stages = (int)kmajor_stages(N, K), nslab = splits * stages * 2;
      const int64_t frags = (int64_t)nslab * 8 * 64;
      // A[m][k] at m * sam + k * sak is the pack kernel's "B[k][n]" with n = m
      hipLaunchKernelGGL(split_pack_b_k, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, s, A, sak, sam, (int)M, (int)K, 8,
                         nslab, (u32x4*)ws);
      GNF_LAUNCH_CHECK();
      float* part = (float*)((char*)ws + 3 * frags * 16);
      KmArgs ka{(const u32x4*)ws, B, sbk, part, M, N, K, nslab, stages};
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_kmajor_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                KM_LDS);
      hipLaunchKernelGGL(gemm_split_kmajor_k, dim3((unsigned)((N / KM_BN) * splits)), dim3(512), KM_LDS, s, ka);
      GNF_LAUNCH_CHECK();
      const int64_t n4 = M * N / 4;
      hipLaunchKernelGGL(split_reduce_k, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, part, splits, n4, n4, (f32x4*)C,
                         scm / 4, N / 4);
      GNF_LAUNCH_CHECK();
      g_split_last = "gemm_split_kmajor_k";
      return 0;
    }
    if (!bias && !relu && wide_eligible(sam, sak, scm, scn, A, C, M, N, K)) {
      const int ntiles = (int)(N / 16);
      const int64_t frags = (int64_t)4 * ntiles * 64;
      hipLaunchKernelGGL(split_pack_b_k, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, s, B, sbk, sbn, (int)N, (int)K,
                         ntiles, 4, (u32x4*)ws);
      GNF_LAUNCH_CHECK();
      WideArgs wa{A, sam, (const u32x4*)ws, C, scm, (float*)((char*)ws + 3 * frags * 16), M, (int)N};
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_split_wide_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                WD_LDS);
      hipLaunchKernelGGL(gemm_split_wide_k, dim3((unsigned)((M + WD_BM - 1) / WD_BM)), dim3(512), WD_LDS, s, wa);
      GNF_LAUNCH_CHECK();
      g_split_last = "gemm_split_wide_k";
      return 0;
    }
  }
  if (classes == 0) classes = 3;
  int64_t kps = (K + splits - 1) / splits;
  kps = (kps + KS - 1) / KS * KS;
  if (kps < KS) kps = KS;
  SplitArgs g{A, sam, sak, B, sbk, sbn, C, scm, scn, bias, relu, M, N, K, kps, c_split_stride};
  const dim3 grid((unsigned)((N + TS - 1) / TS), (unsigned)((M + TS - 1) / TS), (unsigned)splits);
  if (grid.y > 65535u) return GNF_ESHAPE;
  if (classes == 1) hipLaunchKernelGGL(gemm_split_k<1>, grid, dim3(256), 0, s, g);
  else if (classes == 2) hipLaunchKernelGGL(gemm_split_k<2>, grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL(gemm_split_k<3>, grid, dim3(256), 0, s, g);
  GNF_LAUNCH_CHECK();
  g_split_last = "gemm_split_k";
  return 0;
}

// gnf_gemm's hook: runs the dedicated split-bf16 kernel of the shape if there is one, the switch is on and `ws` is large
// enough.  Returns 0 when it ran, 1 when the call is not its business (the caller goes on to the fp32-MFMA kernels), else
// an error code.
int gnf_gemm_split_try(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C, int64_t scm,
                       int64_t scn, const float* bias, int relu, int64_t M, int64_t N, int64_t K, void* ws, int64_t ws_bytes,
                       hipStream_t s) {
  if (!gnf_gemm_split_enabled() || !ws) return 1;
  const int64_t need = gnf_gemm_split_ws_bytes(M, N, K);
  if (need <= 0 || ws_bytes < need) return 1;
  const bool tall = tall_eligible(sam, sak, sbk, scn, A, M, N, K);
  const bool wide = !bias && !relu && wide_eligible(sam, sak, scm, scn, A, C, M, N, K);
  const bool kmaj = !bias && !relu && kmajor_eligible(sam, sbk, sbn, scm, scn, B, C, M, N, K);
  if (!tall && !wide && !kmaj) return 1;
  return gnf_gemm_split_bf16(A, sam, sak, B, sbk, sbn, C, scm, scn, bias, relu, M, N, K, 0, 1, 0, ws, ws_bytes, (gnf_stream_t)s);
}
