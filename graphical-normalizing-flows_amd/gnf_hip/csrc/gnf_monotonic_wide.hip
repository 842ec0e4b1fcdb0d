// Monotonic (UMNN) normalizer, backward of WIDE integrand nets (hidden widths 113..160, e.g. BASELINE cfg5's
// [150,150,150]): models/Normalizers/MonotonicNormalizer.py:21-66 + the UMNN 1.0 backward (Leibniz rule for x, quadrature
// of df/dh and df/dtheta), restated in oracle/gnf_oracle.py.
//
// Why a second formulation.  gnf_monotonic.hip gives every wavefront 16 elements and chains the hidden state of ONE
// quadrature node through the MFMA C/D registers.  At H = 150 that costs 484 registers, the two 105 KB hidden matrices
// must be swapped through LDS twice per node (four workgroup barriers), and the hidden->hidden weight gradients do not
// fit anywhere: layer inputs and dpre were staged to HBM (2 560 B per (element, node), 177 GB per cfg5 step) and
// contracted by a second kernel -- 354 GB of traffic for 0.8 GB of algorithmic bytes.
//
// Here the roles are turned around:
//  * A workgroup (4 wavefronts, one per SIMD, one workgroup per CU) walks groups of 32 elements and, per step, a BATCH
//    of 64 (element, node) pairs = 32 elements x 2 quadrature nodes.  The hidden state of the batch lives in LDS,
//    pair-major ([64][HP + 4] floats per layer): exactly the B operand a layer needs (one ds_read_b128 = 4 K-steps).
//  * A layer is out[HP x 64] = W[HP x HP] act[HP x 64].  The OUTPUT units are split over the wavefronts: wavefront
//    (mh, nh) owns HT/2 of the HT out tiles and the 32 pairs of node nh, so every weight fragment it fetches feeds 8
//    MFMAs and every activation read 20.  Weights are never LDS-resident: they stream from L2 as ready-made A fragments
//    (1 KB of consecutive bytes per wave-instruction, MonoLayout::o_Wf / o_WTf), double-buffered in registers under the
//    40 MFMAs of the previous k-tile -- with the state in LDS there are registers to spare for that.
//  * dW_l = sum over pairs of dpre_l (x) input_l is an MFMA contraction with K = the 64 pairs of the batch, both operands
//    read back from the same LDS buffers (ds_read_b32, transposed view).  The HT x HT accumulator tiles of a layer are
//    dealt over the four wavefronts ((HT/2)^2 tiles = 100 registers per layer and wavefront at HT = 10) and stay in
//    registers for the whole persistent loop.  NOTHING per (element, node) goes to HBM.
//  * Bias / first / last layer gradients: per-lane partials; the first layer's dpre summed over the nodes (Dsum, per
//    element) is written once per element for the d W1h GEMM of the caller, as in gnf_monotonic.hip.
// Work per batch and wavefront at H = 150, 3 hidden layers: 2 400 MFMAs (recompute 800, data gradient 800, weight
// gradient 800), 7 workgroup barriers, ~200 global fragment loads, ~500 LDS accesses.
#include "gnf_common.h"
#include "gnf_monotonic.h"
#include <cstdlib>

namespace {

using namespace gnfmono;

constexpr int kGE = 32;            // elements per group
constexpr int kNP = 64;            // (element, node) pairs per batch: node slot nh = pair / 32, element = pair % 32

// LDS plan (floats)
template <int HT, int NH>
struct WidePlan {
#ifndef GNF_WIDE_PITCH_PAD
#define GNF_WIDE_PITCH_PAD 4
#endif
  static constexpr int HP = 16 * HT, P = HP + GNF_WIDE_PITCH_PAD;           // P: row pitch of the pair-major buffers
  static constexpr int o_w1x = 0, o_wL = HP, o_b = 2 * HP;  // b_l at o_b + (l-1) HP, l = 1..NH-1
  static constexpr int o_bL = o_b + (NH - 1) * HP;
  static constexpr int o_c1 = o_bL + 4;                     // [32][P]  W1h h + b1 of the group's elements
  static constexpr int o_act = o_c1 + kGE * P;              // input of hidden layer l at o_act + (l-1) 64 P
  static constexpr int o_dp = o_act + (NH - 1) * kNP * P;   // [64][P]  dpre of the layer being back-propagated
  static constexpr int o_sred = o_dp + kNP * P;             // [2][64]  last-layer partial dot products per out half
  static constexpr int o_sx = o_sred + 2 * kNP;             // [2][32]  df/dx partials of the Jacobian node
  static constexpr int total = o_sx + 2 * kGE;
};

// A fragment of the pack through a buffer descriptor: descriptor and fragment offset (soff, bytes) in SGPRs, 16 lane
// bytes in ONE VGPR -- no 64-bit per-lane addresses (flat loads reach +-4 KB by immediate, a matrix is 100 KB: hipcc
// hoisted two dozen address pairs per matrix out of the node loop and spilled them)
// Values the compiler must not recognise as loop-invariant: everything derived from them (fragment offsets, the small
// vectors w1x / wL / b_l in LDS) would otherwise be hoisted out of the node loop and held in registers -- 80 VGPRs of
// hoisted LDS reads and 200 SGPR offsets in the first build of this kernel, spilled in turn.
__device__ __forceinline__ int opaque_v(int x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ int opaque_s(int x) { asm volatile("" : "+s"(x)); return x; }
// max(x, 0) as ONE v_max_f32: fmaxf() is compiled into a canonicalising v_max(x, x) plus the maximum (NaN quieting the
// kernels do not need: a NaN pre-activation stays a NaN either way)
__device__ __forceinline__ float relu1(float x) { float y; asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x)); return y; }
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ f32x4 ldfrag(rsrc_t rs, int voff, int soff) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  return __builtin_bit_cast(f32x4, v);
}

// One pass over a hidden->hidden matrix for a chain wavefront: MF full out tiles (both node slots) and, for an odd tile
// count, the shared middle tile for ONE node slot (the other out half takes the other slot, so both halves carry
// HT/2 tile products per k-tile):
//   acc[mi][sl] += sum_k A(full tile mi, k) B(k, the wavefront's 16 pairs of node slot sl),   acc[MF][0] likewise for
//   (shared tile, own slot).
//   rs, voff, soff : fragment (first full tile, 0) of the matrix (global, L2); fragment (mi, t) sits (mi HT + t) KB further;
//   soffx : fragment (shared tile, 0);  bsrc : LDS address of (pair (slot 0, element j), unit 4 q), slot 1 adds 32 P,
//   k-tile t adds 16;  bsrcx : the same for the shared tile's slot.
// Fragments of k-tile t+1 are requested in front of the MFMAs of k-tile t; those of k-tile 0 (A0) were requested by the
// caller (frag_prefetch) in front of whatever serial section and barrier precede the pass, so that their L2 latency
// is not exposed at the head of every pass.
template <int HT, int MF, int XT>
__device__ __forceinline__ void frag_prefetch(rsrc_t rs, int voff, int soff, int soffx, f32x4 (&A0)[MF + XT]) {
#pragma unroll
  for (int mi = 0; mi < MF; ++mi) A0[mi] = ldfrag(rs, voff, soff + (mi * HT) * 1024);
  if constexpr (XT) A0[MF] = ldfrag(rs, voff, soffx);
}
template <int HT, int MF, int XT, int P>
__device__ __forceinline__ void layer_pass(rsrc_t rs, int voff, int soff, int soffx, const float* bsrc, const float* bsrcx,
                                           const f32x4 (&A0)[MF + XT], f32x4 (&acc)[MF + XT][2], int ksv) {
  f32x4 A[2][MF + XT], B[2][2 + XT];
  constexpr int MT = MF + XT, AH = (MT + 1) / 2;
  const int rlast = __builtin_amdgcn_readfirstlane(ksv - 4 * (HT - 1));   // valid k-steps of the last tile (1..4)
  // the requests for k-tile t+1 ride in the shadow of the MFMAs of k-tile t, a few at a time: issued as one block between
  // two k-tiles they take ~100 issue cycles during which the matrix pipe runs dry (one wavefront per SIMD feeds it in the
  // forward passes): weight fragments behind the MFMAs of r = 0 and r = 1 (longest latency first), LDS reads behind r = 2
  auto fetch_part = [&](int part, int t, f32x4 (&Ad)[MF + XT], f32x4 (&Bd)[2 + XT]) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      if ((part == 0 && mi < AH) || (part == 1 && mi >= AH)) {
#ifdef GNF_WIDE_EXP_L1       // measurement only: every fragment from the same KB (L1 hits; wrong results)
        Ad[mi] = ldfrag(rs, voff, opaque_s(mi < MF ? soff : soffx));                    // opaque: no CSE of the MFMAs
#else
        Ad[mi] = mi < MF ? ldfrag(rs, voff, soff + (mi * HT + t) * 1024) : ldfrag(rs, voff, soffx + t * 1024);
#endif
      }
    }
    if (part == 2) {
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) Bd[sl] = ld4(bsrc + sl * kGE * P + 16 * t);
      if constexpr (XT) Bd[2] = ld4(bsrcx + 16 * t);
    }
  };
#pragma unroll
  for (int mi = 0; mi < MF + XT; ++mi) A[0][mi] = A0[mi];
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) B[0][sl] = ld4(bsrc + sl * kGE * P);
  if constexpr (XT) B[0][2] = ld4(bsrcx);
#pragma unroll
  for (int t = 0; t < HT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      __builtin_amdgcn_sched_barrier(0);
      // (the k-steps past the layer's last unit multiply zeros, MonoLayout::perm; only the last tile can hold such steps, and
      // rlast is wave-uniform BY CONSTRUCTION: left as a plain comparison against a kernel argument, hipcc precomputed all 4 HT
      // conditions as 64-bit lane masks outside the batch loop and reloaded them from spilled SGPRs, two v_readlane each)
      if (t < HT - 1 || r < rlast) {
#pragma unroll
        for (int mi = 0; mi < MF; ++mi)
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) acc[mi][sl] = mfma(A[t & 1][mi][r], B[t & 1][sl][r], acc[mi][sl]);
        if constexpr (XT) acc[MF][0] = mfma(A[t & 1][MF][r], B[t & 1][2][r], acc[MF][0]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < HT) fetch_part(r, t + 1, A[(t + 1) & 1], B[(t + 1) & 1]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 matrix pipe with exact 3 x bf16 operand splits (round 6; see mono_fwd_wide_split_k below for the method)
// ---------------------------------------------------------------------------------------------------------------------
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8w;

__device__ __forceinline__ unsigned cvt_pk_bf16w(float a, float b) {
  unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
typedef float f32x2w __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pairw(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  // the two remainders of a level as ONE packed subtraction (v_pk_add_f32): 9 instead of 11 instructions per pair
  h = cvt_pk_bf16w(x0, x1);
  const f32x2w r = f32x2w{x0, x1} - f32x2w{__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};   // exact
  m = cvt_pk_bf16w(r[0], r[1]);
  const f32x2w q = r - f32x2w{__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};                // exact
  l = cvt_pk_bf16w(q[0], q[1]);
}
__device__ __forceinline__ u32x4w ldfragu(rsrc_t rs, int voff, int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
}
__device__ __forceinline__ f32x4 mfma_bf(const u32x4w& a, const u32x4w& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8w, a), __builtin_bit_cast(bf16x8w, b), c, 0, 0, 0);
}

// fragments (plane p, the wavefront's tiles, k tile t) of one matrix: full tile mi at soff + ((p HT + mi) KT + t) KB, the shared
// tile at soffx + (p HT KT + t) KB
template <int HT, int MF, int XT, int KT>
__device__ __forceinline__ void fragp_load(rsrc_t rs, int voff, int soff, int soffx, int p, int t, u32x4w (&A)[MF + XT]) {
#pragma unroll
  for (int mi = 0; mi < MF; ++mi) A[mi] = ldfragu(rs, voff, soff + ((p * HT + mi) * KT + t) * 1024);
  if constexpr (XT) A[MF] = ldfragu(rs, voff, soffx + (p * HT * KT + t) * 1024);
}


// The same products for the BACKWARD's chain wavefronts, whose layer inputs and dpre stay fp32 in LDS (the weight-gradient
// wavefronts contract them as fp32, the ReLU gates read them): a B fragment is 8 consecutive floats of the pair's row
// (two ds_read_b128), split in registers (44 VALU instructions) and used by every tile of the wavefront -- 6 (MF + shared)
// MFMAs of 16 cycles each per split.  One accumulator class (registers: the chain wavefronts carry 60 registers of
// per-lane gradient partials through the loop): against fp64 the same error as the fp32 MFMA's (tools/wide_split_probe.hip).
//   bsrc: LDS address (floats) of (pair (slot 0, element j), position 8 q); slot 1 adds 32 P; k tile t adds 32
template <int HT, int MF, int XT, int KT, int P>
__device__ __forceinline__ void layer_pass_split32(rsrc_t rs, int voff, int soff, int soffx, const float* bsrc, const float* bsrcx,
                                                   u32x4w (&A)[3][MF + XT], f32x4 (&acc)[MF + XT][2]) {
  constexpr int HP = 16 * HT;
  u32x4w B[3][2 + XT];
  const int qv = (int)(threadIdx.x & 63) >> 4;
  auto loadB = [&](int t) {
#pragma unroll
    for (int sl = 0; sl < 2 + XT; ++sl) {
      const float* src = (sl < 2 ? bsrc + sl * kGE * P : bsrcx) + 32 * t;
      f32x4 lo = ld4(src), hi = ld4(src + 4);
      if (32 * t + 32 > HP) {                        // the last k tile of an odd tile count: positions >= HP belong to the next row
        if (32 * t + 8 * qv >= HP) { lo = f32x4{0.f, 0.f, 0.f, 0.f}; hi = lo; }
      }
      unsigned h[4], m[4], l[4];
      split3_pairw(lo[0], lo[1], h[0], m[0], l[0]);
      split3_pairw(lo[2], lo[3], h[1], m[1], l[1]);
      split3_pairw(hi[0], hi[1], h[2], m[2], l[2]);
      split3_pairw(hi[2], hi[3], h[3], m[3], l[3]);
      B[0][sl] = u32x4w{h[0], h[1], h[2], h[3]};
      B[1][sl] = u32x4w{m[0], m[1], m[2], m[3]};
      B[2][sl] = u32x4w{l[0], l[1], l[2], l[3]};
    }
  };
  auto prod = [&](int pa, int pb) {
#pragma unroll
    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) acc[mi][sl] = mfma_bf(A[pa][mi], B[pb][sl], acc[mi][sl]);
    if constexpr (XT) acc[MF][0] = mfma_bf(A[pa][MF], B[pb][2], acc[MF][0]);
  };
  // Only the lo plane of k tile 0 was requested ahead of the serial section (A[2]; 20 registers live across it instead of
  // 60: with all three the chain role spilled 100 registers and wrote 12 GB of scratch per cfg5 launch); mid and hi follow
  // here, behind them the first B fragments' reads and splits
  fragp_load<HT, MF, XT, KT>(rs, voff, soff, soffx, 1, 0, A[1]);
  fragp_load<HT, MF, XT, KT>(rs, voff, soff, soffx, 0, 0, A[0]);
  loadB(0);
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    __builtin_amdgcn_sched_barrier(0);
    prod(2, 0);                                                           // small terms first, hi hi last
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < KT) fragp_load<HT, MF, XT, KT>(rs, voff, soff, soffx, 2, t + 1, A[2]);
    __builtin_amdgcn_sched_barrier(0);
    prod(1, 1); prod(1, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < KT) fragp_load<HT, MF, XT, KT>(rs, voff, soff, soffx, 1, t + 1, A[1]);
    __builtin_amdgcn_sched_barrier(0);
    prod(0, 2); prod(0, 1); prod(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < KT) { fragp_load<HT, MF, XT, KT>(rs, voff, soff, soffx, 0, t + 1, A[0]); loadB(t + 1); }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// In-place MFMA for the long-lived weight-gradient accumulators.  The builtin leaves vdst and srcC untied; with 200 of the
// 256 registers holding accumulators the allocator then splits their live ranges (v_mov chains through the whole tile
// set and scratch spills in the first build of the role).  Tied through the "+v" constraint every tile stays where it is.
// Nothing reads a tile between its MFMAs except other MFMAs on it (the hardware interlocks those); the stores at the
// end of the kernel sit behind workgroup barriers.
__device__ __forceinline__ void mfma_acc(float a, float b, f32x4& c) {
  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// How the HT x HT accumulator tiles of a layer are dealt over the four weight-gradient wavefronts.
//   even HT: wavefront w owns the (HT/2) x (HT/2) block (w & 1, w >> 1): one code path, block origin at run time
//   odd  HT: wavefront W owns the row-major tile range [HT^2 W / 4, HT^2 (W+1) / 4): 12-13 tiles of 2-3 rows at HT = 7,
//            one code path per wavefront (W at compile time)
template <int HT, int W>
struct DwDeal {
  static constexpr int lo = HT * HT * W / 4, hi = HT * HT * (W + 1) / 4, NT = hi - lo;
  static constexpr int r0 = lo / HT, NR = (hi - 1) / HT - r0 + 1, NC = HT;
  static constexpr int lrow(int i) { return (lo + i) / HT - r0; }
  static constexpr int lcol(int i) { return (lo + i) % HT; }
  static constexpr bool bias_row(int r) { return (r0 + r) * HT >= lo && (r0 + r) * HT < hi; }   // owns tile (row, 0)
  __device__ static int ti0(int) { return r0; }
  __device__ static int tn0(int) { return 0; }
  __device__ static bool bias_rt(int) { return true; }
};
template <int HT>
struct DwDeal<HT, -1> {
  static constexpr int NR = HT / 2, NC = HT / 2, NT = NR * NC;
  static constexpr int lrow(int i) { return i / NC; }
  static constexpr int lcol(int i) { return i % NC; }
  static constexpr bool bias_row(int) { return true; }
  __device__ static int ti0(int w) { return (w & 1) * NR; }
  __device__ static int tn0(int w) { return (w >> 1) * NC; }
  __device__ static bool bias_rt(int w) { return (w >> 1) == 0; }    // both column blocks of a row block read the same dpre
};

// Weight-gradient tiles of one layer: accW[i] += sum over the 64 pairs of dpre[pair][16 row(i) + .] (x) in[pair][16 col(i) + .]
//   dsrc : dp buffer  + q P + 16 ti0 + j   (lane (q, j) reads pair 4 s + q of K-step s)
//   asrc : act buffer + q P + 16 tn0 + j
// pb[r] += the dpre operands themselves: summed over the K-steps here and over q by the caller they are the column sums of
// dpre, i.e. the bias gradient of the rows' out units.
template <class D, int P>
__device__ __forceinline__ void dw_pass(const float* dsrc, const float* asrc, f32x4 (&accW)[D::NT], float (&pb)[D::NR]) {
  float fa[2][D::NR], fb[2][D::NC];
#pragma unroll
  for (int a = 0; a < D::NR; ++a) fa[0][a] = dsrc[16 * a];
#pragma unroll
  for (int b = 0; b < D::NC; ++b) fb[0][b] = asrc[16 * b];
  // the tiles of a K-step in chunks; behind each chunk a share of the next K-step's operand reads and of this one's bias
  // additions (as one block between two K-steps they leave the matrix pipe idle whenever the chain wavefront of the SIMD
  // is not issuing)
  constexpr int NCH = 4, CH = (D::NT + NCH - 1) / NCH, NRD = D::NR + D::NC, RCH = (NRD + NCH - 1) / NCH;
#pragma unroll
  for (int s = 0; s < kNP / 4; ++s) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = ch * CH; i < (ch + 1) * CH && i < D::NT; ++i)
        mfma_acc(fa[s & 1][D::lrow(i)], fb[s & 1][D::lcol(i)], accW[i]);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < kNP / 4) {
#pragma unroll
        for (int i = ch * RCH; i < (ch + 1) * RCH && i < NRD; ++i) {
          if (i < D::NR) fa[(s + 1) & 1][i] = dsrc[4 * (s + 1) * P + 16 * i];
          else fb[(s + 1) & 1][i - D::NR] = asrc[4 * (s + 1) * P + 16 * (i - D::NR)];
        }
      }
#pragma unroll
      for (int a = ch; a < D::NR; a += NCH)           // volatile: left to itself the compiler sinks all additions of a pass
        if (D::bias_row(a))                           // to its end and spills their operands
          asm volatile("v_add_f32 %0, %0, %1" : "+v"(pb[a]) : "v"(fa[s & 1][a]));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

#ifdef GNF_WIDE_TIMING        // measurement only (tools/time_wide_phases.py): s_memtime stamps of ONE batch of workgroup 0
__device__ unsigned long long gnf_wide_stamps[8 * 32];
__device__ int gnf_wide_stamp_on;
#define STAMP(i) do { if (gnf_wide_stamp_on_l) gnf_wide_stamps[wave * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#define STAMP_SEL(grp, k0) const bool gnf_wide_stamp_on_l = gnf_wide_stamp_on && blockIdx.x == 0 && (grp) == (int64_t)gridDim.x * 2 && (k0) == 6 && lane == 0
#else
#define STAMP(i) do {} while (0)
#define STAMP_SEL(grp, k0) do {} while (0)
#endif

constexpr int kWideWaves = 8;      // 4 chain wavefronts + 4 weight-gradient wavefronts: two per SIMD, 256 registers each

// Every wavefront of the workgroup passes the same sequence of workgroup barriers (the two roles execute different
// s_barrier instructions; the hardware counts arrivals):
//   group:   [c1 written]  batches ...  [Ds written] ([shared tile's Ds halves added], odd HT)
//   batch:   [input of layer 1 written] ([input of layer l+1 written]) x (NH-2)  [last-layer partial dots written]
//            then for l = NH-1 .. 1:  [dpre_l written]  [dpre_l and input_l no longer read]
#ifdef GNF_WIDE_EXP_NOBAR     // measurement only: no workgroup barriers (wrong results)
__device__ __forceinline__ void wg_barrier() {}
#else
__device__ __forceinline__ void wg_barrier() { __syncthreads(); }
#endif

template <int HT, int NH>
struct WideCtx {                    // what both roles need
  using PL = WidePlan<HT, NH>;
  static constexpr int HP = PL::HP, P = PL::P;
};

// Work items of the persistent backward loop.  A group is 32 elements x 2 quadrature nodes per batch.  When the groups
// do not fill the last round of the grid (POWER: 1875 groups on 256 CUs = 7.3 rounds, the last one 8 % of the kernel),
// the elements of that round are dealt as HALF groups instead: 16 elements x 4 nodes per batch -- the element-half bit nh
// of a pair becomes a node bit -- so a half group needs half the batches, and the partial first-layer sums of its two
// node halves meet in the group epilogue.
struct WideSched {
  int64_t nfull;      // full groups (a multiple of the grid when there are half groups)
  int64_t nhalf;      // half groups behind them (<= grid), element base 32 nfull + 16 h
};
__host__ __device__ inline WideSched wide_sched(int64_t ecount, int64_t grid) {
  const int64_t g32 = (ecount + kGE - 1) / kGE;
  const int64_t tail = g32 % grid;
  WideSched w{g32, 0};
  if (tail != 0 && g32 > grid) {
    const int64_t nfull = g32 - tail;
    const int64_t nhalf = (ecount - nfull * kGE + 15) / 16;
    if (nhalf <= grid) { w.nfull = nfull; w.nhalf = nhalf; }
  }
  return w;
}

// Per-group epilogue, all 8 wavefronts, behind the [Ds written] barrier: rows 0..31 of the input-1 buffer hold Ds (the
// first layer's dpre summed over the nodes) of the group's elements -> Dsum rows (for d W1h, d b1) and dh
template <int HT, int NH>
__device__ __forceinline__ void group_epilogue(const MonoArgs& a, float* smem, int64_t ebase, bool half, int64_t erows, int wave,
                                               int q, int j) {
  using PL = WidePlan<HT, NH>;
  constexpr int HP = PL::HP, P = PL::P, MF = HT / 2, XT = HT & 1;
  const MonoLayout& L = a.L;
  float* d0 = smem + PL::o_act;
  if constexpr (XT) {               // the shared middle tile: each out half summed its own node slot; slot-1 half in rows 32..
    if (threadIdx.x < kGE * 4) {
      const int el = threadIdx.x >> 2, c4 = threadIdx.x & 3;
      float* p0 = d0 + el * P + 16 * MF + 4 * c4;
      *reinterpret_cast<f32x4*>(p0) = ld4(p0) + ld4(p0 + kGE * P);
    }
    wg_barrier();
  }
  constexpr int C4 = HP / 4;
  if (half) {                       // rows 16.. hold the second node half of elements 0..15
    for (int idx = threadIdx.x; idx < 16 * C4; idx += blockDim.x) {
      const int el = idx / C4, c4 = idx - el * C4;
      float* p0 = d0 + el * P + 4 * c4;
      *reinterpret_cast<f32x4*>(p0) = ld4(p0) + ld4(p0 + 16 * P);
    }
    wg_barrier();
  }
  const int nel = half ? 16 : kGE;
  for (int idx = threadIdx.x; idx < nel * C4; idx += blockDim.x) {
    const int el = idx / C4, c4 = idx - el * C4;
    const int64_t row = ebase + el;
    if (row < erows) *reinterpret_cast<f32x4*>(a.Dsum + row * HP + 4 * c4) = ld4(d0 + el * P + 4 * c4);
  }
  // dh[el][cc] = sum_u W1h[u][cc] Ds[el][u]: (c tile, element half) pairs dealt over the wavefronts
  for (int idx = wave; idx < 2 * (L.CP / 16); idx += kWideWaves) {
    const int ct = idx >> 1, eh = idx & 1;
    if (half && eh) continue;
    f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* dr = d0 + (16 * eh + j) * P + 4 * q;
#pragma unroll
    for (int t = 0; t < HT; ++t) {
      const f32x4 A = ld4(a.pack + L.o_W1hT + (16 * ct + j) * L.LDW + 16 * t + 4 * q);
      const f32x4 Bv = ld4(dr + 16 * t);
#pragma unroll
      for (int r = 0; r < 4; ++r) o = mfma(A[r], Bv[r], o);
    }
    const int64_t el = ebase + 16 * eh + j;
    if (el < a.ecount) {
      const int64_t e = a.e0 + el;
      const int64_t b = e / a.d, i = e - b * a.d;
      const int64_t gbase = b * a.g_sb + i * a.g_sd;
      const float gz = a.gz[e];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cc = 16 * ct + 4 * q + r;
        if (cc < L.c) a.gh[gbase + cc * a.g_sc] = o[r] + (cc == 0 ? gz : 0.f);   // + gz: the "+ z0" term
      }
    }
  }
}

// =========================================================================================================================
// weight-gradient role (wavefronts 4..7): accumulator tiles per DwDeal, all layers, for the whole persistent loop
// =========================================================================================================================
template <int HT, int NH, int W>
__device__ __forceinline__ void dw_role(const MonoArgs& a, float* smem, int w, WideSched ws, int64_t erows, int wave,
                                        int q, int j, float* prow_g) {
  using PL = WidePlan<HT, NH>;
  using D = DwDeal<HT, W>;
  constexpr int HP = PL::HP, P = PL::P;
  const int ti0 = D::ti0(w), tn0 = D::tn0(w);
  f32x4 accW[NH - 1][D::NT];
  float p_b[NH - 1][D::NR];         // lane (q, j): share of d b_l[16 (ti0 + r) + j] (pairs = q mod 4)
#pragma unroll
  for (int l = 0; l < NH - 1; ++l) {
#pragma unroll
    for (int r = 0; r < D::NR; ++r) p_b[l][r] = 0.f;
#pragma unroll
    for (int i = 0; i < D::NT; ++i) accW[l][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* dsrc = smem + PL::o_dp + q * P + 16 * ti0 + j;
  const int aroff = q * P + 16 * tn0 + j;
  for (int64_t grp = blockIdx.x; grp < ws.nfull + ws.nhalf; grp += gridDim.x) {
    const bool half = grp >= ws.nfull;                // (one round of half groups: workgroup b takes half group b)
    const int64_t ebase = half ? ws.nfull * kGE + 16 * (grp - ws.nfull) : grp * kGE;
    const int nbat = half ? (a.NK + 3) / 4 : a.NK / 2;
    wg_barrier();                                     // c1 written
    for (int bt = 0; bt < nbat; ++bt) {
      const int lane = threadIdx.x & 63;
      const int k0 = 2 * bt;
      STAMP_SEL(grp, k0);
      STAMP(0);
#pragma unroll
      for (int l = 0; l < NH; ++l) { wg_barrier(); STAMP(1 + l); }   // layer inputs 1..NH-1, last-layer partial dots
#pragma unroll
      for (int l = NH - 1; l >= 1; --l) {
        wg_barrier();                                 // dpre_l written
        STAMP(8 + 3 * (NH - 1 - l));
        dw_pass<D, P>(dsrc, smem + PL::o_act + (l - 1) * kNP * P + aroff, accW[l - 1], p_b[l - 1]);
        STAMP(9 + 3 * (NH - 1 - l));
        wg_barrier();                                 // dpre_l, input_l no longer read
        STAMP(10 + 3 * (NH - 1 - l));
      }
    }
    wg_barrier();                                     // Ds written
    group_epilogue<HT, NH>(a, smem, ebase, half, erows, wave, q, j);
  }
  if (D::bias_rt(w)) {
#pragma unroll
    for (int l = 1; l < NH; ++l)
#pragma unroll
      for (int r = 0; r < D::NR; ++r)
        if (D::bias_row(r)) {
          const float v = qsum(p_b[l - 1][r]);
          if (q == 0) prow_g[(2 + l) * HP + 16 * (ti0 + r) + j] = v;
        }
  }
  float* wrow = a.wpart + (int64_t)blockIdx.x * ((NH - 1) * HP * HP);
#pragma unroll
  for (int l = 0; l < NH - 1; ++l)
#pragma unroll
    for (int i = 0; i < D::NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        wrow[l * HP * HP + (16 * (ti0 + D::lrow(i)) + 4 * q + r) * HP + 16 * (tn0 + D::lcol(i)) + j] = accW[l][i][r];
}

// SPLIT: the chain wavefronts' hidden->hidden products (recompute and data gradient) on the bf16 matrix pipe
// (layer_pass_split32); the weight-gradient wavefronts are the same in both forms.
template <int HT, int NH, bool SPLIT>
__global__ __launch_bounds__(64 * kWideWaves, 1) void mono_bwd_wide_k(MonoArgs a) {
  static_assert(NH >= 2, "at least one hidden->hidden layer");
  using PL = WidePlan<HT, NH>;
  constexpr int HP = PL::HP, P = PL::P;
  constexpr int MF = HT / 2, XT = HT & 1, MT = MF + XT;     // full out tiles per chain wavefront, shared middle tile, local tiles
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const MonoLayout& L = a.L;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;

  float* const c1buf = smem + PL::o_c1;
  float* const dpbuf = smem + PL::o_dp;
  float* const sred = smem + PL::o_sred;
  float* const sxbuf = smem + PL::o_sx;
  auto actbuf = [&](int l) -> float* { return smem + PL::o_act + (l - 1) * kNP * P; };   // input of hidden layer l

  for (int i = threadIdx.x; i < HP; i += blockDim.x) {
    smem[PL::o_w1x + i] = a.pack[L.o_w1x + i];
    smem[PL::o_wL + i] = a.pack[L.o_wL + i];
#pragma unroll
    for (int l = 1; l < NH; ++l) smem[PL::o_b + (l - 1) * HP + i] = a.pack[L.o_b[l] + i];
  }
  if (threadIdx.x == 0) smem[PL::o_bL] = a.pack[L.o_bL];
  // (the first barrier of the group loop orders these stores before their readers)

  const WideSched ws = wide_sched(a.ecount, gridDim.x);
  const int64_t erows = (a.ecount + 15) / 16 * 16;    // Dsum rows the caller's column sums read
  const int64_t vecw = (NH + 2) * HP + 4;
  float* const prow_g = a.part + ((int64_t)blockIdx.x * kWaves + (wave & 3)) * vecw;   // one row per pair of wavefronts

  if (wave >= 4) {
    const int w = wave - 4;
    if constexpr (XT == 0) {
      dw_role<HT, NH, -1>(a, smem, w, ws, erows, wave, q, j, prow_g);
    } else {
      switch (w) {
        case 0: dw_role<HT, NH, 0>(a, smem, w, ws, erows, wave, q, j, prow_g); break;
        case 1: dw_role<HT, NH, 1>(a, smem, w, ws, erows, wave, q, j, prow_g); break;
        case 2: dw_role<HT, NH, 2>(a, smem, w, ws, erows, wave, q, j, prow_g); break;
        default: dw_role<HT, NH, 3>(a, smem, w, ws, erows, wave, q, j, prow_g); break;
      }
    }
    return;
  }

  // =======================================================================================================================
  // chain role: wavefront (mh, nh) owns MF full out tiles of every layer (mh = 0: tiles 0.., mh = 1: the last MF) for both
  // node slots of a batch and, for odd HT, the middle tile for node slot mh; its elements are 16 nh + j of the group, i.e.
  // rows 32 s + 16 nh + j (s = node slot) of the batch buffers.  Local tile mi < MF: full; mi = MF: the shared one.
  // =======================================================================================================================
  const int mh = wave & 1, nh = wave >> 1;
  const int m0 = mh * (MF + XT);
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pack), 0, L.pack_floats * 4, 0x00020000);
  const int prow = (16 * nh + j) * P;                 // slot 0 row of this lane's element (slot 1: + 32 P)
  const int xrow = prow + mh * kGE * P;               // row of the shared tile's slot
  const int ucol_c = 16 * m0 + 4 * q;                 // own unit column of the first full tile (tile mi adds 16)
  const int xcol_c = 16 * MF + 4 * q;                 // ... of the shared tile
  // column of local tile mi relative to the laundered bases (uc, xc)
  auto col = [&](int mi, int uc, int xc) { return mi < MF ? uc + 16 * mi : xc; };

  f32x4 p_wL[MT], p_w1x[MT];
  float p_bL = 0.f;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    p_wL[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    p_w1x[mi] = p_wL[mi];
  }
  const float fS = (float)a.S;

  for (int64_t grp = blockIdx.x; grp < ws.nfull + ws.nhalf; grp += gridDim.x) {
    // ---- the lane's element.  Half group: both element halves nh hold elements 0..15 and nh is a node bit instead
    const bool half = grp >= ws.nfull;
    const int64_t ebase = half ? ws.nfull * kGE + 16 * (grp - ws.nfull) : grp * kGE;
    const int nbat = half ? (a.NK + 3) / 4 : a.NK / 2;
    const int knh = half ? 2 * nh : 0;                // node offset of this element half inside a batch
    const int kstep = half ? 4 : 2;
    const int64_t el = ebase + (half ? j : 16 * nh + j);
    const bool valid = el < a.ecount;
    const int64_t e = a.e0 + (valid ? el : a.ecount - 1);
    const float xv = a.x[e];
    const float xT = fS * (xv / fS);                  // xT = x0 + nb_steps * step, x0 = 0
    const float gz = valid ? a.gz[e] : 0.f;
    const float gj = (valid && a.gjac) ? a.gjac[e] : 0.f;
    const float cotq = gz * xT * .5f;                 // grad_out (xT - x0) / 2
    // ---- c1 = b1 + W1h h of the element, own out tiles (MFMA, K = c); the shared tile is computed by both out halves
    {
      const int64_t b = e / a.d, i = e - b * a.d;
      const int64_t hbase = b * a.h_sb + i * a.h_sd;
      f32x4 c[MT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) c[mi] = ld4(a.pack + L.o_b1 + col(mi, ucol_c, xcol_c));
      for (int s = 0; s < L.CP / 16; ++s) {
        float hv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cc = 16 * s + 4 * q + r;
          hv[r] = cc < L.c ? a.h[hbase + cc * a.h_sc] : 0.f;
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const int tile = mi < MF ? m0 + mi : MF;
          const f32x4 A = ld4(a.pack + L.o_W1h + (16 * tile + j) * L.LDH + 16 * s + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) c[mi] = mfma(A[r], hv[r], c[mi]);
        }
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) *reinterpret_cast<f32x4*>(c1buf + prow + col(mi, ucol_c, xcol_c)) = c[mi];
    }
    wg_barrier();                                     // c1 written

    f32x4 Ds[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) Ds[mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    float fjac = 0.f;
    bool sawj = false;                                // this wavefront's pairs include the Jacobian node of its element

    // The node loop is software-pipelined by one serial section: the input of hidden layer 1 (rank-1 in x_k) of batch
    // k0 + 2 is computed at the END of batch k0 -- in front of the barrier at which the chain wavefronts wait for the
    // weight-gradient wavefronts anyway -- and only stored behind it.
    float xk[2], cot[2];
    bool isj[2];
    f32x4 acc[MT][2];                                 // layer-1 input -> pre-activation -> activation -> dpre of (own units, own pairs)
    f32x4 Apre[SPLIT ? 1 : MT];                       // first weight fragments of the next pass
    u32x4w Asp[SPLIT ? 3 : 1][MT];                    // ... and, split form, the fragment registers themselves
    constexpr int KT = (HP + 31) / 32;
    // request the first fragments of a pass over matrix `om` (fp32: o_Wf / o_WTf, split: o_Wp / o_WTp of the same layer)
    auto prefetch = [&](int of32, int osp) {
      if constexpr (SPLIT) {                        // (the lo plane only: see layer_pass_split32)
        fragp_load<HT, MF, XT, KT>(rs, 16 * lane, opaque_s(4 * (osp + m0 * KT * 256)), opaque_s(4 * (osp + MF * KT * 256)), 2, 0, Asp[2]);
      } else {
        frag_prefetch<HT, MF, XT>(rs, 16 * lane, opaque_s(4 * (of32 + m0 * HT * 256)), opaque_s(4 * (of32 + MF * HT * 256)), Apre);
      }
    };
    auto pass = [&](int of32, int osp, const float* buf, int ksv) {
      if constexpr (SPLIT) {
        layer_pass_split32<HT, MF, XT, KT, P>(rs, 16 * lane, opaque_s(4 * (osp + m0 * KT * 256)), opaque_s(4 * (osp + MF * KT * 256)),
                                              buf + prow + 8 * q, buf + xrow + 8 * q, Asp, acc);
      } else {
        layer_pass<HT, MF, XT, P>(rs, 16 * lane, opaque_s(4 * (of32 + m0 * HT * 256)), opaque_s(4 * (of32 + MF * HT * 256)),
                                  buf + prow + 4 * q, buf + xrow + 4 * q, Apre, acc, ksv);
      }
    };
    auto node_params = [&](int k0) {
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int k = k0 + knh + sl;
        const bool isq = k <= a.S;
        isj[sl] = k == a.S + 1;
        const float wk = isq ? a.ccw[k] : 0.f, tk = isq ? a.cct[k] : 0.f;
        xk[sl] = isq ? xT * (tk + 1.f) * .5f : xv;
        cot[sl] = isq ? wk * cotq : (isj[sl] ? gj : 0.f);
      }
    };
    auto layer0 = [&](int ucol, int xcol) {           // acc = relu(c1 + w1x x_k) for the own (units, pairs)
      const float xkx = mh ? xk[1] : xk[0];           // the shared tile's node slot
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int cm = col(mi, ucol, xcol);
        const f32x4 wx = ld4(smem + PL::o_w1x + cm);
        const f32x4 c = ld4(c1buf + prow + cm);
#pragma unroll
        for (int sl = 0; sl < (mi < MF ? 2 : 1); ++sl) {
          const float xs = mi < MF ? xk[sl] : xkx;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mi][sl][r] = relu1(fmaf(wx[r], xs, c[r]));
        }
      }
    };
    node_params(0);
    layer0(opaque_v(ucol_c), opaque_v(xcol_c));
    prefetch(L.o_Wf[1], L.o_Wp[1]);

    for (int bt = 0; bt < nbat; ++bt) {
      const int k0 = kstep * bt;
      const int ucol = opaque_v(ucol_c), xcol = opaque_v(xcol_c);   // (see opaque_v)
      STAMP_SEL(grp, k0);
      STAMP(0);
      const float xkx = mh ? xk[1] : xk[0];           // the shared tile's node slot
      // ---- input of hidden layer 1 (computed ahead, see above)
      {
        float* a1 = actbuf(1);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int sl = 0; sl < (mi < MF ? 2 : 1); ++sl)
            *reinterpret_cast<f32x4*>(a1 + (mi < MF ? prow + sl * kGE * P : xrow) + col(mi, ucol, xcol)) = acc[mi][sl];
      }
      STAMP(1);
      wg_barrier();                                   // input of layer 1 written
      STAMP(2);
      // ---- hidden layers 1..NH-1, forward
#pragma unroll
      for (int l = 1; l < NH; ++l) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          acc[mi][0] = ld4(smem + PL::o_b + (l - 1) * HP + col(mi, ucol, xcol));
          acc[mi][1] = acc[mi][0];
        }
        pass(L.o_Wf[l], L.o_Wp[l], actbuf(l), L.ksv[l]);
        // next pass: the next layer, or the top layer's transpose
        prefetch(l < NH - 1 ? L.o_Wf[l < NH - 1 ? l + 1 : l] : L.o_WTf[NH - 1], l < NH - 1 ? L.o_Wp[l < NH - 1 ? l + 1 : l] : L.o_WTp[NH - 1]);
        STAMP(3 + 3 * (l - 1));
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int sl = 0; sl < (mi < MF ? 2 : 1); ++sl)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][sl][r] = relu1(acc[mi][sl][r]);
        if (l < NH - 1) {
          float* an = actbuf(l + 1);
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int sl = 0; sl < (mi < MF ? 2 : 1); ++sl)
              *reinterpret_cast<f32x4*>(an + (mi < MF ? prow + sl * kGE * P : xrow) + col(mi, ucol, xcol)) = acc[mi][sl];
          STAMP(4 + 3 * (l - 1));
          wg_barrier();                               // input of layer l+1 written
          STAMP(5 + 3 * (l - 1));
        }
      }
      // ---- last layer (H -> 1): partial dot over this wavefront's units, the two out halves meet in LDS
      {
        float sp[2] = {0.f, 0.f}, spx = 0.f;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const f32x4 wl = ld4(smem + PL::o_wL + col(mi, ucol, xcol));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (mi < MF) {
              sp[0] = fmaf(wl[r], acc[mi][0][r], sp[0]);
              sp[1] = fmaf(wl[r], acc[mi][1][r], sp[1]);
            } else {
              spx = fmaf(wl[r], acc[mi][0][r], spx);
            }
          }
        }
        if constexpr (XT) { sp[0] += mh ? 0.f : spx; sp[1] += mh ? spx : 0.f; }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          sp[sl] = qsum(sp[sl]);
          if (q == 0) sred[mh * kNP + kGE * sl + 16 * nh + j] = sp[sl];
        }
      }
      STAMP(12);
      wg_barrier();                                   // last-layer partial dots written
      STAMP(13);
      float dpl[2];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const float s = (sred[kGE * sl + 16 * nh + j] + sred[kNP + kGE * sl + 16 * nh + j]) + smem[PL::o_bL];
        if (isj[sl]) { fjac = elu_plus(s); sawj = true; }
        dpl[sl] = cot[sl] * (s > 0.f ? 1.f : expf(s));
      }
      const float dplx = mh ? dpl[1] : dpl[0];
      if (mh == 0 && q == 0) p_bL += dpl[0] + dpl[1];
      // ---- backward through the last layer: d wL partials, dpre of hidden layer NH-1
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const f32x4 wl = ld4(smem + PL::o_wL + col(mi, ucol, xcol));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (mi < MF) {
            p_wL[mi][r] = fmaf(dpl[0], acc[mi][0][r], fmaf(dpl[1], acc[mi][1][r], p_wL[mi][r]));
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) acc[mi][sl][r] = acc[mi][sl][r] > 0.f ? wl[r] * dpl[sl] : 0.f;
          } else {
            p_wL[mi][r] = fmaf(dplx, acc[mi][0][r], p_wL[mi][r]);
            acc[mi][0][r] = acc[mi][0][r] > 0.f ? wl[r] * dplx : 0.f;
          }
        }
      }
      // ---- hidden layers, top down
#pragma unroll
      for (int l = NH - 1; l >= 1; --l) {
        // acc = dpre of layer l for this wavefront's (units, pairs): into the dp buffer for everybody
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int sl = 0; sl < (mi < MF ? 2 : 1); ++sl)
            *reinterpret_cast<f32x4*>(dpbuf + (mi < MF ? prow + sl * kGE * P : xrow) + col(mi, ucol, xcol)) = acc[mi][sl];
        STAMP(14 + 5 * (NH - 1 - l));
        wg_barrier();                                 // dpre_l written
        STAMP(15 + 5 * (NH - 1 - l));
        // d input_l = W_l^T dpre_l, gated by input_l > 0 (for l = 1 that is the first layer's dpre)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) { acc[mi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mi][1] = acc[mi][0]; }
        pass(L.o_WTf[l], L.o_WTp[l], dpbuf, L.ksv[l + 1]);
        // next pass: the layer below, or layer 1 of the next batch
        prefetch(l > 1 ? L.o_WTf[l > 1 ? l - 1 : l] : L.o_Wf[1], l > 1 ? L.o_WTp[l > 1 ? l - 1 : l] : L.o_Wp[1]);
        STAMP(16 + 5 * (NH - 1 - l));
        const float* ag = actbuf(l);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int sl = 0; sl < (mi < MF ? 2 : 1); ++sl) {
            const f32x4 g = ld4(ag + (mi < MF ? prow + sl * kGE * P : xrow) + col(mi, ucol, xcol));
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][sl][r] = g[r] > 0.f ? acc[mi][sl][r] : 0.f;
          }
        STAMP(17 + 5 * (NH - 1 - l));
        if (l > 1) {
          wg_barrier();                               // every reader of the dp buffer (and of input_l) is done
          STAMP(18 + 5 * (NH - 1 - l));
        }                                             // (l = 1: behind the first-layer section below)
      }
      // ---- first layer: rank-1 in x_k, node-independent in h
      float sx[2] = {0.f, 0.f}, sxx = 0.f;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const f32x4 wx = ld4(smem + PL::o_w1x + col(mi, ucol, xcol));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (mi < MF) {
            p_w1x[mi][r] = fmaf(acc[mi][0][r], xk[0], fmaf(acc[mi][1][r], xk[1], p_w1x[mi][r]));
            Ds[mi][r] += acc[mi][0][r] + acc[mi][1][r];
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) sx[sl] = fmaf(wx[r], acc[mi][sl][r], sx[sl]);
          } else {
            p_w1x[mi][r] = fmaf(acc[mi][0][r], xkx, p_w1x[mi][r]);
            Ds[mi][r] += acc[mi][0][r];
            sxx = fmaf(wx[r], acc[mi][0][r], sxx);
          }
        }
      }
      if (isj[0] || isj[1]) {                         // Jacobian node: this out half's share of df/dx
        float v = isj[0] ? sx[0] : sx[1];
        if (XT && (mh ? isj[1] : isj[0])) v += sxx;
        v = qsum(v);
        if (q == 0) sxbuf[mh * kGE + 16 * nh + j] = v;
      }
      if (bt + 1 < nbat) {                            // layer-1 input of the next batch, stored at the top of the loop
        node_params(k0 + kstep);
        layer0(ucol, xcol);
      }
      STAMP(30);
      wg_barrier();                                   // every reader of the dp buffer and of the layer inputs is done
      STAMP(31);
    }

    // ---- per-group epilogue: Ds rows of the group's elements into the (free) input-1 buffer, then Dsum / dh / dx
    {
      float* d1 = actbuf(1);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        *reinterpret_cast<f32x4*>(d1 + (mi < MF ? prow : xrow) + col(mi, ucol_c, xcol_c)) = Ds[mi];
    }
    wg_barrier();                                     // Ds written
    group_epilogue<HT, NH>(a, smem, ebase, half, erows, wave, q, j);
    if (mh == 0 && q == 0 && valid && sawj && a.gx)   // Leibniz rule: dz/dx = f(x; h);  + gjac df/dx(x; h) (its cotangent was gjac)
      a.gx[e] = gz * fjac + (sxbuf[16 * nh + j] + sxbuf[kGE + 16 * nh + j]);
    // (the [c1 written] barrier of the next group orders these reads before the next writes of the buffers)
  }

  // ---- per-lane partials -> the wavefront's share of its partial row (the caller zeroes the rows; the shared tile's
  //      entries are written by both out halves, each into its own row)
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int hid = (mi < MF ? 16 * (m0 + mi) : 16 * MF) + 4 * q + r;
      float v = jsum(p_wL[mi][r]);
      if (j == 0) prow_g[hid] = v;
      v = jsum(p_w1x[mi][r]);
      if (j == 0) prow_g[HP + hid] = v;
    }
  const float vbl = jsum(p_bL);
  if (lane == 0 && mh == 0) prow_g[(NH + 2) * HP] = vbl;
}

// =========================================================================================================================
// Forward (z, jac) of wide nets in the same formulation: chain wavefronts only.  Nothing has to be kept for a backward,
// so ONE pair-major buffer is enough (a layer's outputs wait in registers until every wavefront has read its inputs):
// 66 KB of LDS at HT = 10 and 4 wavefronts of <= 256 registers per workgroup, i.e. TWO independent workgroups per CU --
// the serial sections of one (layer-0 inputs, ReLU + stores, the last-layer exchange, barriers) sit under the MFMAs of
// the other.  z = sum_k w_k f(x_k) * xT/2 + h0 accumulates per lane over the batches of the element's group.
// =========================================================================================================================
template <int HT, int NH>
struct WidePlanF {
  static constexpr int HP = 16 * HT, P = HP + GNF_WIDE_PITCH_PAD;
  static constexpr int o_w1x = 0, o_wL = HP, o_b = 2 * HP;
  static constexpr int o_bL = o_b + (NH - 1) * HP;
  static constexpr int o_c1 = o_bL + 4;                     // [32][P]
  static constexpr int o_act = o_c1 + kGE * P;              // [64][P]  input of the layer being evaluated
  static constexpr int o_sred = o_act + kNP * P;            // [2][64]
  static constexpr int total = o_sred + 2 * kNP;
};

template <int HT, int NH>
__global__ __launch_bounds__(64 * kWaves, 2) void mono_fwd_wide_k(MonoArgs a) {
  using PL = WidePlanF<HT, NH>;
  constexpr int HP = PL::HP, P = PL::P;
  constexpr int MF = HT / 2, XT = HT & 1, MT = MF + XT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const MonoLayout& L = a.L;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  float* const c1buf = smem + PL::o_c1;
  float* const act = smem + PL::o_act;
  float* const sred = smem + PL::o_sred;
  for (int i = threadIdx.x; i < HP; i += blockDim.x) {
    smem[PL::o_w1x + i] = a.pack[L.o_w1x + i];
    smem[PL::o_wL + i] = a.pack[L.o_wL + i];
#pragma unroll
    for (int l = 1; l < NH; ++l) smem[PL::o_b + (l - 1) * HP + i] = a.pack[L.o_b[l] + i];
  }
  if (threadIdx.x == 0) smem[PL::o_bL] = a.pack[L.o_bL];

  const int mh = wave & 1, nh = wave >> 1;            // as in the backward's chain role
  const int m0 = mh * (MF + XT);
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pack), 0, L.pack_floats * 4, 0x00020000);
  const int prow = (16 * nh + j) * P, xrow = prow + mh * kGE * P;
  const int ucol_c = 16 * m0 + 4 * q, xcol_c = 16 * MF + 4 * q;
  auto col = [&](int mi, int uc, int xc) { return mi < MF ? uc + 16 * mi : xc; };
  const WideSched ws = wide_sched(a.n, gridDim.x);      // (an unfilled last round is dealt as half groups, see WideSched)
  const float fS = (float)a.S;
  const int NK = (a.S + 2 + 1) / 2 * 2;

  for (int64_t grp = blockIdx.x; grp < ws.nfull + ws.nhalf; grp += gridDim.x) {
    const bool half = grp >= ws.nfull;
    const int64_t ebase = half ? ws.nfull * kGE + 16 * (grp - ws.nfull) : grp * kGE;
    const int nbat = half ? (NK + 3) / 4 : NK / 2;
    const int knh = half ? 2 * nh : 0, kstep = half ? 4 : 2;
    const int64_t el = ebase + (half ? j : 16 * nh + j);
    const bool valid = el < a.n;
    const int64_t e = valid ? el : a.n - 1;
    const int64_t b = e / a.d, i = e - b * a.d;
    const int64_t hbase = b * a.h_sb + i * a.h_sd;
    const float xv = a.x[e];
    const float xT = fS * (xv / fS);                  // xT = x0 + nb_steps * step, x0 = 0
    const float h0 = a.h[hbase];
    {
      f32x4 c[MT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) c[mi] = ld4(a.pack + L.o_b1 + col(mi, ucol_c, xcol_c));
      for (int s = 0; s < L.CP / 16; ++s) {
        float hv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cc = 16 * s + 4 * q + r;
          hv[r] = cc < L.c ? a.h[hbase + cc * a.h_sc] : 0.f;
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const int tile = mi < MF ? m0 + mi : MF;
          const f32x4 A = ld4(a.pack + L.o_W1h + (16 * tile + j) * L.LDH + 16 * s + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) c[mi] = mfma(A[r], hv[r], c[mi]);
        }
      }
      __syncthreads();                                // the previous group's c1 / act / sred are no longer read
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) *reinterpret_cast<f32x4*>(c1buf + prow + col(mi, ucol_c, xcol_c)) = c[mi];
    }
    __syncthreads();                                  // c1 written

    float xk[2], wq[2];
    bool isj[2];
    f32x4 acc[MT][2], Apre[MT];
    auto node_params = [&](int k0) {
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int k = k0 + knh + sl;
        const bool isq = k <= a.S;
        isj[sl] = k == a.S + 1;
        wq[sl] = isq ? a.ccw[k] : 0.f;
        xk[sl] = isq ? xT * (a.cct[isq ? k : 0] + 1.f) * .5f : xv;
      }
    };
    auto layer0 = [&](int ucol, int xcol) {
      const float xkx = mh ? xk[1] : xk[0];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int cm = col(mi, ucol, xcol);
        const f32x4 wx = ld4(smem + PL::o_w1x + cm);
        const f32x4 c = ld4(c1buf + prow + cm);
#pragma unroll
        for (int sl = 0; sl < (mi < MF ? 2 : 1); ++sl) {
          const float xs = mi < MF ? xk[sl] : xkx;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[mi][sl][r] = relu1(fmaf(wx[r], xs, c[r]));
        }
      }
    };
    auto store_act = [&](int ucol, int xcol) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int sl = 0; sl < (mi < MF ? 2 : 1); ++sl)
          *reinterpret_cast<f32x4*>(act + (mi < MF ? prow + sl * kGE * P : xrow) + col(mi, ucol, xcol)) = acc[mi][sl];
    };
    node_params(0);
    layer0(opaque_v(ucol_c), opaque_v(xcol_c));
    frag_prefetch<HT, MF, XT>(rs, 16 * lane, opaque_s(4 * (L.o_Wf[1] + m0 * HT * 256)), opaque_s(4 * (L.o_Wf[1] + MF * HT * 256)), Apre);
    float zacc = 0.f, fjac = 0.f;

    for (int bt = 0; bt < nbat; ++bt) {
      const int k0 = kstep * bt;
      const int ucol = opaque_v(ucol_c), xcol = opaque_v(xcol_c);
      store_act(ucol, xcol);                          // input of hidden layer 1 (computed ahead)
      __syncthreads();
#pragma unroll
      for (int l = 1; l < NH; ++l) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          acc[mi][0] = ld4(smem + PL::o_b + (l - 1) * HP + col(mi, ucol, xcol));
          acc[mi][1] = acc[mi][0];
        }
        layer_pass<HT, MF, XT, P>(rs, 16 * lane, opaque_s(4 * (L.o_Wf[l] + m0 * HT * 256)),
                                  opaque_s(4 * (L.o_Wf[l] + MF * HT * 256)), act + prow + 4 * q, act + xrow + 4 * q, Apre, acc, L.ksv[l]);
        {
          const int on = L.o_Wf[l < NH - 1 ? l + 1 : 1];
          frag_prefetch<HT, MF, XT>(rs, 16 * lane, opaque_s(4 * (on + m0 * HT * 256)), opaque_s(4 * (on + MF * HT * 256)), Apre);
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int sl = 0; sl < (mi < MF ? 2 : 1); ++sl)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][sl][r] = relu1(acc[mi][sl][r]);
        if (l < NH - 1) {
          __syncthreads();                            // every wavefront has read the layer's inputs
          store_act(ucol, xcol);
          __syncthreads();
        }
      }
      {
        float sp[2] = {0.f, 0.f}, spx = 0.f;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const f32x4 wl = ld4(smem + PL::o_wL + col(mi, ucol, xcol));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (mi < MF) {
              sp[0] = fmaf(wl[r], acc[mi][0][r], sp[0]);
              sp[1] = fmaf(wl[r], acc[mi][1][r], sp[1]);
            } else {
              spx = fmaf(wl[r], acc[mi][0][r], spx);
            }
          }
        }
        if constexpr (XT) { sp[0] += mh ? 0.f : spx; sp[1] += mh ? spx : 0.f; }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          sp[sl] = qsum(sp[sl]);
          if (q == 0) sred[mh * kNP + kGE * sl + 16 * nh + j] = sp[sl];
        }
      }
      __syncthreads();                                // partial dots written (and the layer inputs no longer read)
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const float s = (sred[kGE * sl + 16 * nh + j] + sred[kNP + kGE * sl + 16 * nh + j]) + smem[PL::o_bL];
        const float f = elu_plus(s);
        zacc = fmaf(wq[sl], f, zacc);
        if (isj[sl]) fjac = f;
      }
      if (bt + 1 < nbat) {
        node_params(k0 + kstep);
        layer0(ucol, xcol);
      }
    }
    if (half) {                                       // the two node halves of an element meet (nh = 1 -> nh = 0)
      __syncthreads();                                // the last batch's partial dots are read
      if (nh == 1 && mh == 0 && q == 0) { sred[j] = zacc; sred[16 + j] = fjac; }
      __syncthreads();
      if (nh == 0) { zacc += sred[j]; fjac += sred[16 + j]; }   // (the Jacobian node was in one half: the other holds 0)
    }
    if (valid && mh == 0 && q == 0 && (!half || nh == 0)) {
      a.z[el] = zacc * xT * .5f + h0;
      a.jac[el] = fjac;
    }
  }
}

// =========================================================================================================================
// Forward on the bf16 matrix pipe with fp32 accuracy (round 6; the fc1 kernels' method, gnf_gemm_split.hip): every fp32
// operand of a hidden->hidden product is split exactly into three bf16 numbers (x = hi + mid + lo, round to nearest at each
// level), a product is the sum of its six leading cross terms on v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate), hi*hi
// in one fp32 accumulator and the five small terms in a second one.  Against an fp64 chain of three 160 x 160 layers the
// result is 3x CLOSER than the fp32-MFMA chain's (tools/wide_split_probe.hip: 1.0e-7 against 3.3e-7 relative L2).
//  * Weights: pre-split by mono_pack_k into fragment-major planes (MonoLayout::o_Wp), 1 KB per (plane, out tile, 32-wide k
//    tile), streamed from L2 through the buffer descriptor like the fp32 fragments.
//  * Activations: the layer inputs of the batch live in LDS as THREE bf16 planes [3][64 pairs][32 KT32 + 16] -- written by
//    the lane that holds the values after the ReLU (split: 5.5 VALU instructions per value, three ds_write_b64 per tile), read
//    back as B operands with one ds_read_b128 per plane and k tile (row pitch = 2 (mod 4) 16-byte slots: conflict-free for
//    the hardware's 4 x 16 lane groups).  70.7 KB at H = 160: two workgroups per CU as before.
//  * The first layer's conditioner part c1 stays in registers (the fp32 kernel parks it in LDS to leave room for a third
//    workgroup; here the planes decide the occupancy).
//  * Single-buffered weight fragments: a plane's registers are re-requested for the next k tile right behind its last
//    product (lo: 1 of 6 products, mid: 2, hi: 3), so each request has at least a third of a k tile's MFMAs to land.
// Layer 1 (rank-1 in x on top of c1), the last layer (a dot product per pair) and the quadrature stay fp32 VALU work.
// GNF_TRUE_F32=1 keeps mono_fwd_wide_k.
// =========================================================================================================================
template <int HT, int NH>
struct WidePlanS {
  static constexpr int HP = 16 * HT, KT = (HP + 31) / 32, KP = 32 * KT;
  static constexpr int PB = 2 * KP + 32;                    // bytes; PB / 16 = KP / 8 + 2 = 2 (mod 4)
  static constexpr int PLANE = kNP * PB;                    // bytes
  static constexpr int o_w1x = 0, o_wL = HP, o_b = 2 * HP;  // floats
  static constexpr int o_bL = o_b + (NH - 1) * HP;
  static constexpr int o_sred = o_bL + 4;                   // [2][64]
  static constexpr int o_planes = o_sred + 2 * kNP;         // floats: a multiple of 4 (16-byte aligned)
  static constexpr int total_bytes = 4 * o_planes + 3 * PLANE;
};

// acc[.][.][0] += hi hi, acc[.][.][1] += the five small terms, over all k tiles.  A: fragments of k tile 0 (requested by the
// caller ahead of its serial section).  bsrc: LDS byte address of (plane 0, pair (slot 0, element j), position 8 q); slot 1 adds
// 32 PB, plane p adds PLANE, k tile t adds 64;  bsrcx: the same for the shared tile's slot.
template <int HT, int MF, int XT, int KT, int PB, int PLANE>
__device__ __forceinline__ void layer_pass_split(rsrc_t rs, int voff, int soff, int soffx, const unsigned char* bsrc,
                                                 const unsigned char* bsrcx, u32x4w (&A)[3][MF + XT],
                                                 f32x4 (&acc)[MF + XT][2][2]) {
  u32x4w B[3][2 + XT];
  auto loadB = [&](int t) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) B[p][sl] = *reinterpret_cast<const u32x4w*>(bsrc + p * PLANE + sl * kGE * PB + 64 * t);
      if constexpr (XT) B[p][2] = *reinterpret_cast<const u32x4w*>(bsrcx + p * PLANE + 64 * t);
    }
  };
  // products of (A plane pa) x (B plane pb) for every tile of the wavefront, into class cl
  auto prod = [&](int pa, int pb, int cl) {
#pragma unroll
    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) acc[mi][sl][cl] = mfma_bf(A[pa][mi], B[pb][sl], acc[mi][sl][cl]);
    if constexpr (XT) acc[MF][0][cl] = mfma_bf(A[pa][MF], B[pb][2], acc[MF][0][cl]);
  };
  loadB(0);
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    __builtin_amdgcn_sched_barrier(0);
    prod(2, 0, 1);                                                        // lo hi
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < KT) fragp_load<HT, MF, XT, KT>(rs, voff, soff, soffx, 2, t + 1, A[2]);
    __builtin_amdgcn_sched_barrier(0);
    prod(1, 1, 1); prod(1, 0, 1);                                         // mid mid, mid hi
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < KT) fragp_load<HT, MF, XT, KT>(rs, voff, soff, soffx, 1, t + 1, A[1]);
    __builtin_amdgcn_sched_barrier(0);
    prod(0, 2, 1); prod(0, 1, 1); prod(0, 0, 0);                          // hi lo, hi mid | hi hi
    __builtin_amdgcn_sched_barrier(0);
    // the activation planes are single-buffered too (registers): their LDS latency at the head of the next k tile is covered
    // by the CU's second workgroup
    if (t + 1 < KT) { fragp_load<HT, MF, XT, KT>(rs, voff, soff, soffx, 0, t + 1, A[0]); loadB(t + 1); }
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int HT, int NH>
__global__ __launch_bounds__(64 * kWaves, 2) void mono_fwd_wide_split_k(MonoArgs a) {
  using PL = WidePlanS<HT, NH>;
  constexpr int HP = PL::HP, KT = PL::KT, PB = PL::PB, PLANE = PL::PLANE;
  constexpr int MF = HT / 2, XT = HT & 1, MT = MF + XT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const MonoLayout& L = a.L;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15;
  float* const sred = smem + PL::o_sred;
  unsigned char* const planes = reinterpret_cast<unsigned char*>(smem + PL::o_planes);
  for (int i = threadIdx.x; i < HP; i += blockDim.x) {
    smem[PL::o_w1x + i] = a.pack[L.o_w1x + i];
    smem[PL::o_wL + i] = a.pack[L.o_wL + i];
#pragma unroll
    for (int l = 1; l < NH; ++l) smem[PL::o_b + (l - 1) * HP + i] = a.pack[L.o_b[l] + i];
  }
  if (threadIdx.x == 0) smem[PL::o_bL] = a.pack[L.o_bL];
  // positions HP .. KP-1 of every row (and the pad) meet zero weights, but must not hold NaN patterns: cleared once, never written
  for (int i = threadIdx.x; i < 3 * kNP * (PB - 2 * HP) / 4; i += blockDim.x) {
    const int row = i / ((PB - 2 * HP) / 4), w = i - row * ((PB - 2 * HP) / 4);
    *reinterpret_cast<unsigned*>(planes + row * PB + 2 * HP + 4 * w) = 0u;
  }

  const int mh = wave & 1, nh = wave >> 1;
  const int m0 = mh * (MF + XT);
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.pack), 0, L.pack_floats * 4, 0x00020000);
  const int prow = (16 * nh + j) * PB, xrow = prow + mh * kGE * PB;       // bytes
  // 16-byte slots of a row are swapped in pairs for rows 4..7, 12..15 (slot ^= (row >> 2) & 1): the plane stores are
  // ds_write_b64 of 16 rows at one column -- 4-way bank conflicts at any 16-byte-aligned pitch, 2-way (their floor: the 16
  // lanes of a store group share one 8-byte half) with the swap; the fragment reads stay conflict-free.  Both offsets are
  // lane constants.
  const int sws = (j >> 2) & 1;
  const int rd16 = 16 * (q ^ sws);                                        // read: slot 4 t + q
  const int wr8 = 16 * ((q >> 1) ^ sws) + 8 * (q & 1);                    // write: slot 2 tile + (q >> 1), half q & 1
  const int ucol_c = 16 * m0 + 4 * q, xcol_c = 16 * MF + 4 * q;
  auto col = [&](int mi, int uc, int xc) { return mi < MF ? uc + 16 * mi : xc; };
  const WideSched ws = wide_sched(a.n, gridDim.x);
  const float fS = (float)a.S;
  const int NK = (a.S + 2 + 1) / 2 * 2;
  auto wfrag = [&](int l) { return opaque_s(4 * (L.o_Wp[l] + m0 * KT * 256)); };
  auto wfragx = [&](int l) { return opaque_s(4 * (L.o_Wp[l] + MF * KT * 256)); };

  for (int64_t grp = blockIdx.x; grp < ws.nfull + ws.nhalf; grp += gridDim.x) {
    const bool half = grp >= ws.nfull;
    const int64_t ebase = half ? ws.nfull * kGE + 16 * (grp - ws.nfull) : grp * kGE;
    const int nbat = half ? (NK + 3) / 4 : NK / 2;
    const int knh = half ? 2 * nh : 0, kstep = half ? 4 : 2;
    const int64_t el = ebase + (half ? j : 16 * nh + j);
    const bool valid = el < a.n;
    const int64_t e = valid ? el : a.n - 1;
    const int64_t b = e / a.d, i = e - b * a.d;
    const int64_t hbase = b * a.h_sb + i * a.h_sd;
    const float xv = a.x[e];
    const float xT = fS * (xv / fS);
    const float h0 = a.h[hbase];
    f32x4 c1[MT];                                       // W1h h + b1 of the lane's element and units
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) c1[mi] = ld4(a.pack + L.o_b1 + col(mi, ucol_c, xcol_c));
    for (int s = 0; s < L.CP / 16; ++s) {
      float hv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cc = 16 * s + 4 * q + r;
        hv[r] = cc < L.c ? a.h[hbase + cc * a.h_sc] : 0.f;
      }
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int tile = mi < MF ? m0 + mi : MF;
        const f32x4 A = ld4(a.pack + L.o_W1h + (16 * tile + j) * L.LDH + 16 * s + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) c1[mi] = mfma(A[r], hv[r], c1[mi]);
      }
    }

    float xk[2], wq[2];
    bool isj[2];
    f32x4 acc[MT][2][2];                                // [tile][node slot][class: hi hi | small terms]
    f32x4 val[MT][2];                                   // the layer's outputs after the ReLU
    u32x4w Apre[3][MT];
    auto node_params = [&](int k0) {
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int k = k0 + knh + sl;
        const bool isq = k <= a.S;
        isj[sl] = k == a.S + 1;
        wq[sl] = isq ? a.ccw[k] : 0.f;
        xk[sl] = isq ? xT * (a.cct[isq ? k : 0] + 1.f) * .5f : xv;
      }
    };
    auto layer0 = [&](int ucol, int xcol) {
      const float xkx = mh ? xk[1] : xk[0];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const f32x4 wx = ld4(smem + PL::o_w1x + col(mi, ucol, xcol));
#pragma unroll
        for (int sl = 0; sl < (mi < MF ? 2 : 1); ++sl) {
          const float xs = mi < MF ? xk[sl] : xkx;
#pragma unroll
          for (int r = 0; r < 4; ++r) val[mi][sl][r] = relu1(fmaf(wx[r], xs, c1[mi][r]));
        }
      }
    };
    auto store_split = [&](int ucol, int xcol) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int sl = 0; sl < (mi < MF ? 2 : 1); ++sl) {
          unsigned h0_, m0_, l0_, h1_, m1_, l1_;
          split3_pairw(val[mi][sl][0], val[mi][sl][1], h0_, m0_, l0_);
          split3_pairw(val[mi][sl][2], val[mi][sl][3], h1_, m1_, l1_);
          unsigned char* dst = planes + (mi < MF ? prow + sl * kGE * PB : xrow) + 32 * (mi < MF ? m0 + mi : MF) + opaque_v(wr8);
          *reinterpret_cast<u32x2w*>(dst) = u32x2w{h0_, h1_};
          *reinterpret_cast<u32x2w*>(dst + PLANE) = u32x2w{m0_, m1_};
          *reinterpret_cast<u32x2w*>(dst + 2 * PLANE) = u32x2w{l0_, l1_};
        }
    };
    auto prefetch = [&](int l) {
#pragma unroll
      for (int p = 0; p < 3; ++p) fragp_load<HT, MF, XT, KT>(rs, 16 * lane, wfrag(l), wfragx(l), p, 0, Apre[p]);
    };
    node_params(0);
    layer0(opaque_v(ucol_c), opaque_v(xcol_c));
    prefetch(1);
    float zacc = 0.f, fjac = 0.f;

    for (int bt = 0; bt < nbat; ++bt) {
      const int k0 = kstep * bt;
      const int ucol = opaque_v(ucol_c), xcol = opaque_v(xcol_c);
      __syncthreads();                                  // the previous batch's (group's) planes and partial dots are read
      store_split(ucol, xcol);                          // input of hidden layer 1 (computed ahead)
      __syncthreads();
#pragma unroll
      for (int l = 1; l < NH; ++l) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          acc[mi][0][0] = ld4(smem + PL::o_b + (l - 1) * HP + col(mi, ucol, xcol));
          acc[mi][1][0] = acc[mi][0][0];
          acc[mi][0][1] = acc[mi][1][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        layer_pass_split<HT, MF, XT, KT, PB, PLANE>(rs, 16 * lane, wfrag(l), wfragx(l), planes + prow + rd16, planes + xrow + rd16,
                                                    Apre, acc);
        prefetch(l < NH - 1 ? l + 1 : 1);
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
          for (int sl = 0; sl < (mi < MF ? 2 : 1); ++sl)
#pragma unroll
            for (int r = 0; r < 4; ++r) val[mi][sl][r] = relu1(acc[mi][sl][0][r] + acc[mi][sl][1][r]);
        if (l < NH - 1) {
          __syncthreads();                              // every wavefront has read the layer's inputs
          store_split(ucol, xcol);
          __syncthreads();
        }
      }
      {
        float sp[2] = {0.f, 0.f}, spx = 0.f;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          const f32x4 wl = ld4(smem + PL::o_wL + col(mi, ucol, xcol));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (mi < MF) {
              sp[0] = fmaf(wl[r], val[mi][0][r], sp[0]);
              sp[1] = fmaf(wl[r], val[mi][1][r], sp[1]);
            } else {
              spx = fmaf(wl[r], val[mi][0][r], spx);
            }
          }
        }
        if constexpr (XT) { sp[0] += mh ? 0.f : spx; sp[1] += mh ? spx : 0.f; }
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          sp[sl] = qsum(sp[sl]);
          if (q == 0) sred[mh * kNP + kGE * sl + 16 * nh + j] = sp[sl];
        }
      }
      __syncthreads();                                  // partial dots written (and the layer inputs no longer read)
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const float s = (sred[kGE * sl + 16 * nh + j] + sred[kNP + kGE * sl + 16 * nh + j]) + smem[PL::o_bL];
        const float f = elu_plus(s);
        zacc = fmaf(wq[sl], f, zacc);
        if (isj[sl]) fjac = f;
      }
      if (bt + 1 < nbat) {
        node_params(k0 + kstep);
        layer0(ucol, xcol);
      }
    }
    if (half) {                                         // the two node halves of an element meet (nh = 1 -> nh = 0)
      __syncthreads();
      if (nh == 1 && mh == 0 && q == 0) { sred[j] = zacc; sred[16 + j] = fjac; }
      __syncthreads();
      if (nh == 0) { zacc += sred[j]; fjac += sred[16 + j]; }
    }
    if (valid && mh == 0 && q == 0 && (!half || nh == 0)) {
      a.z[el] = zacc * xT * .5f + h0;
      a.jac[el] = fjac;
    }
  }
}

template <int HT, int NH>
int launch_wide_fwd_split(const MonoArgs& a, hipStream_t s) {
  const size_t lds = (size_t)WidePlanS<HT, NH>::total_bytes;
  const int64_t groups = (a.n + kGE - 1) / kGE;
  const unsigned grid = (unsigned)(groups < 512 ? groups : 512);        // persistent, two workgroups per CU
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_fwd_wide_split_k<HT, NH>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((mono_fwd_wide_split_k<HT, NH>), dim3(grid), dim3(64 * kWaves), lds, s, a);
  GNF_LAUNCH_CHECK();
  return 0;
}

template <int HT, int NH>
int launch_wide_fwd(const MonoArgs& a, hipStream_t s) {
  const size_t lds = (size_t)WidePlanF<HT, NH>::total * sizeof(float);
  const int64_t groups = (a.n + kGE - 1) / kGE;
  const int per_cu = lds * 3 <= (size_t)160 * 1024 ? 3 : 2;          // workgroups per CU (140 registers: up to 3 per SIMD)
  const unsigned grid = (unsigned)(groups < 256 * per_cu ? groups : 256 * per_cu);   // persistent
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_fwd_wide_k<HT, NH>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((mono_fwd_wide_k<HT, NH>), dim3(grid), dim3(64 * kWaves), lds, s, a);
  GNF_LAUNCH_CHECK();
  return 0;
}

template <int HT, int NH, bool SPLIT>
int launch_wide(const MonoArgs& a, unsigned grid, hipStream_t s) {
  const size_t lds = (size_t)WidePlan<HT, NH>::total * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mono_bwd_wide_k<HT, NH, SPLIT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((mono_bwd_wide_k<HT, NH, SPLIT>), dim3(grid), dim3(64 * kWideWaves), lds, s, a);
  GNF_LAUNCH_CHECK();
  return 0;
}

constexpr size_t kLds = 160 * 1024;

}  // namespace

bool gnf_mono_bwd_wide_ok(const gnfmono::MonoLayout& L) {
  if (L.c > 32) return false;
#define GNF_WIDE_CASE(HT_, NH_) \
  if (L.HT == HT_ && L.NH == NH_) return WidePlan<HT_, NH_>::total * sizeof(float) <= kLds;
  GNF_WIDE_CASE(7, 2) GNF_WIDE_CASE(7, 3) GNF_WIDE_CASE(7, 4) GNF_WIDE_CASE(10, 2) GNF_WIDE_CASE(10, 3)
#undef GNF_WIDE_CASE
  return false;
}

unsigned gnf_mono_bwd_wide_grid(const gnfmono::MonoLayout&, int64_t n) {
  const int64_t groups = (n + kGE - 1) / kGE;
  return (unsigned)(groups < 256 ? groups : 256);     // one workgroup per CU, persistent
}

#ifdef GNF_WIDE_TIMING
extern "C" int gnf_debug_wide_stamps(unsigned long long* host, int enable) {
  if (enable >= 0) return (int)hipMemcpyToSymbol(HIP_SYMBOL(gnf_wide_stamp_on), &enable, sizeof(int));
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(gnf_wide_stamps), sizeof(unsigned long long) * 8 * 32);
}
#endif

extern "C" int gnf_gemm_split_enabled(void);

int gnf_mono_bwd_wide_launch(const gnfmono::MonoArgs& a, unsigned grid, hipStream_t s, bool true_f32, const char* volatile* kernel) {
  *kernel = "mono_bwd_wide_k<f32>";
  if (!true_f32 && gnf_gemm_split_enabled()) {
    *kernel = "mono_bwd_wide_k<split>";
#define GNF_WIDE_CASE(HT_, NH_) \
  if (a.L.HT == HT_ && a.L.NH == NH_) return launch_wide<HT_, NH_, true>(a, grid, s);
    // H = 97..112 stays fp32: its contraction pads 112 to 128 and the in-register splits cost what the matrix pipe saves
    // (tools/bench_mono_split.py, cfg2: 1.74 ms against 1.59)
    GNF_WIDE_CASE(10, 2) GNF_WIDE_CASE(10, 3)
#undef GNF_WIDE_CASE
    *kernel = "mono_bwd_wide_k<f32>";
  }
#define GNF_WIDE_CASE(HT_, NH_) \
  if (a.L.HT == HT_ && a.L.NH == NH_) return launch_wide<HT_, NH_, false>(a, grid, s);
  GNF_WIDE_CASE(7, 2) GNF_WIDE_CASE(7, 3) GNF_WIDE_CASE(7, 4) GNF_WIDE_CASE(10, 2) GNF_WIDE_CASE(10, 3)
#undef GNF_WIDE_CASE
  return GNF_ESHAPE;
}

bool gnf_mono_fwd_wide_ok(const gnfmono::MonoLayout& L) {
  if (L.c > 32 || L.NH < 2 || L.NH > 4) return false;
  return L.HT == 7 || L.HT == 10;
}

extern "C" int gnf_gemm_split_enabled(void);

int gnf_mono_fwd_wide_launch(const gnfmono::MonoArgs& a, hipStream_t s, bool true_f32, const char* volatile* kernel) {
  *kernel = "mono_fwd_wide_k";
  if (!true_f32 && gnf_gemm_split_enabled()) {
    *kernel = "mono_fwd_wide_split_k";                     // bf16 matrix pipe, exact 3 x bf16 splits (GNF_TRUE_F32=1: fp32 MFMA)
#define GNF_WIDE_CASE(HT_, NH_) \
  if (a.L.HT == HT_ && a.L.NH == NH_) return launch_wide_fwd_split<HT_, NH_>(a, s);
    GNF_WIDE_CASE(7, 2) GNF_WIDE_CASE(7, 3) GNF_WIDE_CASE(7, 4) GNF_WIDE_CASE(10, 2) GNF_WIDE_CASE(10, 3) GNF_WIDE_CASE(10, 4)
#undef GNF_WIDE_CASE
    *kernel = "mono_fwd_wide_k";
  }
#define GNF_WIDE_CASE(HT_, NH_) \
  if (a.L.HT == HT_ && a.L.NH == NH_) return launch_wide_fwd<HT_, NH_>(a, s);
  GNF_WIDE_CASE(7, 2) GNF_WIDE_CASE(7, 3) GNF_WIDE_CASE(7, 4) GNF_WIDE_CASE(10, 2) GNF_WIDE_CASE(10, 3) GNF_WIDE_CASE(10, 4)
#undef GNF_WIDE_CASE
  return GNF_ESHAPE;
}
