// Shared pieces of the MNISTCNN convolutional front kernels (gnf_mnistcnn_fwd.hip / gnf_mnistcnn.hip): geometry of the
// LDS images, the MFMA / packed-VALU helpers, the Winograd input transform and the conv1 tile routine.
#pragma once
#include "gnf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int IMG = 28, C1 = 26, C2 = 24, PO = 12, NCH = 16;
constexpr int ROWE = 36, ESZ = IMG * ROWE;            // padded input image
constexpr int ROW = 40, CH = C1 * ROW;                // conv1 activations: [16][26][40], CH = 1040 == 16 mod 32
constexpr int ROWD = 40, CHD = 28 * ROWD + 16;        // dY2 with a 2-wide zero border: [16][28][40], CHD = 1136 == 16 mod 32
constexpr int CS = 676;                               // per-tap planes T: [9][26 x 26] flat positions
constexpr int NPOOL = NCH * PO * PO;                  // 2304
constexpr int FWD_WAVES = 8, BWD_WAVES = 8;
constexpr int PROW = NCH * 144 + NCH * 16 + NCH;      // per-wave gradient partial row: dW2 | dW1+db1 | db2

static_assert(CH % 32 == 16 && CHD % 32 == 16, "channel strides must sit 16 banks apart");

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Packed fp32 VALU (2 flops per lane per op) for the Winograd transforms.  One row (t0,t1,t2,t3) of B^T d held as
// A = (t0,t1), B = (t2,t3) gives the four outputs of (B^T d) B in two instructions:
//   A - B                       = (t0 - t2, t1 - t3) = (v0, v3)
//   (A.hi + B.lo, -A.hi + B.lo) = (t1 + t2, t2 - t1) = (v1, v2)      [op_sel picks the halves, neg_hi negates src0]
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_v12(f32x2 a, f32x2 b) {
  f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// B^T d B of a 4x4 patch given as 4 rows x 2 column pairs; out[xi = 4*xi_y + xi_x]
__device__ __forceinline__ void wino_in(const f32x2 (&lo)[4], const f32x2 (&hi)[4], float (&v)[16]) {
  f32x2 tl[4], th[4];                                 // B^T d: rows d0-d2, d1+d2, d2-d1, d1-d3
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

// lane^1 (quad_perm [1,0,3,2]) and lane^8 (row_ror:8 inside a 16-lane row) without touching LDS
__device__ __forceinline__ float dpp_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_xor8(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, false));
}

struct CnnArgs {
  const float* e; const float* W1; const float* b1; const float* W2; const float* b2;
  float* pooled; unsigned char* arg;                                // forward outputs
  const float* gp; const unsigned char* argin; float* ge; float* part;   // backward
  int64_t n;
  const int32_t* plan; float* gec; int dplan;                             // backward, column plan (gnf_hip.h) or NULL
};

// max(x, 0) as ONE v_max_f32: fmaxf() compiles into a canonicalising v_max(x, x) plus the maximum
__device__ __forceinline__ float relu1(float x) { float y; asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x)); return y; }

typedef float f32x16 __attribute__((ext_vector_type(16)));
// conv1 + ReLU of NU units starting at unit u0, on v_mfma_f32_16x16x1_4b_f32 (four independent 16 x 16 x 1 products per
// instruction): block b = the 16-position tile 4 u + b, ONE tap per instruction -- 9 instructions per 64 positions where
// the 16x16x4 form takes 4 x 3 K-steps (a quarter of them multiplying the zero taps 9..11).  B operand = the lane's OWN
// position 64 u + lane, so a tap is an immediate offset of its read (one address per unit instead of three per tile);
// A operand = W1[channel j][tap] for every block; D: register 4 b + r = channel 4 q + r at position 64 u + 16 b + j.
// All reads first, then NU independent chains of 9 MFMAs, then the stores.
template <int NU, int CHS>
__device__ __forceinline__ void conv1_units(const float* e_rd, float* a1_wr, int u0, const float* w1p, const f32x4& b1v,
                                            int q, int j, int lane) {
  // the position -> LDS offsets do not depend on the image: left alone, hipcc hoists them out of the image loop for every
  // unit (registers that spill) -- the opaque copy of u0 keeps the few VALU ops in place.
  // y = pos / 26 = (pos * 2521) >> 16 for pos < 1024; y * IMG + x = pos + 2 y (e and a1 share the row pitch IMG = 28; CHS = channel stride of a1)
  u0 = __builtin_amdgcn_readfirstlane(u0);
  asm volatile("" : "+s"(u0));
  f32x16 acc[NU];
  float ev[NU][9], w1a[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) w1a[tap] = w1p[16 * tap];
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const int pos = 64 * (u0 + k) + lane;
    const int pc = pos < C1 * C1 ? pos : 0;
    const float* pe = e_rd + pc + 2 * (int)(__umul24((unsigned)pc, 2521u) >> 16);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) ev[k][tap] = pe[(tap / 3) * IMG + tap % 3];
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[k][v] = b1v[v & 3];
  }
  __builtin_amdgcn_sched_barrier(0);                   // all reads in flight before the first MFMA (the scheduler otherwise
#pragma unroll                                          // sinks every read next to its MFMA: one LDS round trip each)
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int k = 0; k < NU; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x1f32(w1a[tap], ev[k][tap], acc[k], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < NU; ++k)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int pos = 64 * (u0 + k) + 16 * b + j;
      if (pos < C1 * C1) {
        float* pa = a1_wr + 4 * q * CHS + pos + 2 * (int)(__umul24((unsigned)pos, 2521u) >> 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) pa[r * CHS] = relu1(acc[k][4 * b + r]);
      }
    }
}
// The same with the position -> LDS offsets of the units handed in (a wavefront runs the SAME units for every image of its
// persistent loop): po[k] = {read offset | write offset of block 0 << 16, blocks 1 | 2 << 16, block 3}, 16 bits each, built
// once by conv1_offsets.  Positions past the 26 x 26 grid (the tail of unit 10) read offset 0 and write into the first 8 floats of
// the row BEHIND the image's 26 rows (offsets 26 IMG .. 26 IMG + 7 of their channel: CHS >= 26 IMG + 8), so there is no predicate either: ~8 VALU instructions per
// unit instead of ~45 of address arithmetic.  Costs 3 registers per unit: for kernels that have them (the forward: 230).
template <int CHS>
__device__ __forceinline__ void conv1_offsets(int u, int q, int j, int lane, unsigned (&po)[3]) {
  static_assert(CHS >= 26 * IMG + 8, "the dummy stores of off-grid positions land at 26 IMG + (0..7) of the channel");
  const int pos = 64 * u + lane, pc = pos < C1 * C1 ? pos : 0;
  unsigned w[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int p = 64 * u + 16 * b + j;
    w[b] = (unsigned)(4 * q * CHS + (p < C1 * C1 ? p + 2 * (p / C1) : 26 * IMG + (j & 7)));
  }
  po[0] = (unsigned)(pc + 2 * (pc / C1)) | (w[0] << 16);
  po[1] = w[1] | (w[2] << 16);
  po[2] = w[3];
}
template <int NU, int CHS>
__device__ __forceinline__ void conv1_units_pre(const float* e_rd, float* a1_wr, const unsigned (&po)[3][3], const float* w1p,
                                                const f32x4& b1v) {
  f32x16 acc[NU];
  float ev[NU][9], w1a[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) w1a[tap] = w1p[16 * tap];
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const float* pe = e_rd + (po[k][0] & 0xffffu);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) ev[k][tap] = pe[(tap / 3) * IMG + tap % 3];
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[k][v] = b1v[v & 3];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int k = 0; k < NU; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x1f32(w1a[tap], ev[k][tap], acc[k], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < NU; ++k) {
    const unsigned off[4] = {po[k][0] >> 16, po[k][1] & 0xffffu, po[k][1] >> 16, po[k][2]};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float* pa = a1_wr + off[b];
#pragma unroll
      for (int r = 0; r < 4; ++r) pa[r * CHS] = relu1(acc[k][4 * b + r]);
    }
  }
}
// conv1 + ReLU of the image in e_s into a1_s; 43 tiles of 16 consecutive positions of the 26x26 grid.
// All operand reads of a wave's (up to 6) tiles are issued first, then 6 independent 3-step MFMA chains,
// then the stores: the phase is latency-bound, so nothing may serialise behind a single chain.
template <int NW>
__device__ __forceinline__ void conv1_tiles(const float* e_s, float* a1_s, const float (&w1f)[3], const int (&off1)[3],
                                            const f32x4& b1v, int wave, int q, int j) {
  constexpr int NTL = (43 + NW - 1) / NW, NB = NTL;           // tiles per wave, processed NB at a time
#pragma unroll
  for (int k0 = 0; k0 < NTL; k0 += NB) {
    int po[NB];
    f32x4 acc[NB];
    float ev[NB][3];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int pos = 16 * (wave + NW * (k0 + k)) + j;
      const int pc = pos < C1 * C1 ? pos : 0;
      const int y = pc / C1, x = pc - y * C1;
      po[k] = pos < C1 * C1 ? y * ROW + x : -1;
#pragma unroll
      for (int s = 0; s < 3; ++s) ev[k][s] = e_s[y * ROWE + x + off1[s]];
      acc[k] = b1v;
    }
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int k = 0; k < NB; ++k) acc[k] = mfma(w1f[s], ev[k][s], acc[k]);
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if (po[k] >= 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a1_s[(4 * q + r) * CH + po[k]] = fmaxf(acc[k][r], 0.f);
      }
  }
}

}  // namespace
