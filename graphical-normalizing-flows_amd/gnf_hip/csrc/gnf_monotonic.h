// Internal (not part of the C ABI): layout of the padded weight image, argument block and device helpers shared by
// the Monotonic-normalizer translation units (gnf_monotonic.hip, gnf_monotonic_wide.hip).
#pragma once
#include "gnf_common.h"

namespace gnfmono {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 4;          // wavefronts per workgroup
constexpr int kMaxNH = GNF_MONO_MAX_LAYERS - 1;
constexpr int kNarrowQ = 3 * 3 * (256 + 128);   // words of one bf16-split 48 x 48 block (MonoLayout::o_Wq)

// ---------------------------------------------------------------------------------------
// Padded weight image ("pack"): every matrix row-major with leading dimension LD = pad+4
// floats (keeps float4 fragment reads 16-B aligned and staggers LDS banks).
// ---------------------------------------------------------------------------------------
struct MonoLayout {
  int HT, HP, NH, c, CP, LDH, LDW;
  int HM, EX;                         // "peeled" narrow nets (all hidden widths H, H mod 16 in {1,2,3}): HM = H / 16 full
                                      // tiles on the MFMA, EX = H mod 16 units on the VALU (0, 0 otherwise)
  int o_w1x, o_b1, o_wL, o_bL, o_W1h;
  int o_W[kMaxNH], o_b[kMaxNH];       // hidden->hidden layers l = 1..NH-1
  int fwd_floats;                     // prefix used by forward / inverse
  int o_WT[kMaxNH], o_W1hT;           // transposes, backward only
  int total_floats;                   // end of the row-major image (what the LDS-resident kernels copy)
  // Fragment-major copies of the hidden->hidden matrices for the kernels that stream weights from L2 as MFMA A operands
  // (gnf_monotonic_wide.hip; HT >= 7 only): fragment (mt, t) = 256 consecutive floats, lane (q, j) owns floats 4 lane .. +3
  //   Wf [l][mt][t][lane][r] = W_l[16 mt + j][16 t + 4 q + r]       (forward:   out tile mt, k tile t)
  //   WTf[l][mt][t][lane][r] = W_l[16 t + 4 q + r][16 mt + j]       (backward:  in tile mt,  k tile t over the out units)
  // so that one global_load_dwordx4 per lane fetches a whole fragment as 1 KB of consecutive bytes.
  int o_Wf[kMaxNH], o_WTf[kMaxNH];
  // Round 6: the same matrices as exact 3 x bf16 splits (W = hi + mid + lo, round to nearest at each level) for the forward
  // on the bf16 matrix pipe (mono_fwd_wide_split_k), fragment-major for v_mfma_f32_16x16x32_bf16:
  //   Wp[l][plane][mt][t][lane] = 16 bytes = plane of W_l[16 mt + j][32 t + 8 q .. + 7]   (t < KT32 = ceil(HP / 32); zeros past HP)
  // offsets in 4-byte words like everything else in the pack; 3 * HT * KT32 * 256 words per layer.
  //   WTp[l][plane][mt][t][lane] = plane of W_l[32 t + 8 q .. + 7][16 mt + j]  (the data gradient's transposed products)
  int o_Wp[kMaxNH], o_WTp[kMaxNH], KT32;
  // ... and for the peeled narrow nets (HT = 4, H = 49..51: the 48 x 48 main block on the MFMA): per matrix 3 planes x 3 out
  // tiles x [a K = 32 fragment (1 KB: lane (q, j) = W[16 mt + j][4 q + i] for i < 4, W[16 mt + j][16 + 4 q + i - 4] for i >= 4 --
  // the order in which a lane holds the activations of tiles 0 and 1 in its MFMA C/D registers) + a K = 16 fragment (512 B:
  // W[16 mt + j][32 + 4 q + i])] = kNarrowQ words.  Wq: forward; WTq: the transposed products of the data gradient.
  int o_Wq[kMaxNH], o_WTq[kMaxNH];
  int pack_floats;                    // size of the whole pack
  // K order of the last unit tile (HT >= 7 only, round 5).  An MFMA k-step r of k-tile t contracts the padded positions
  // 16 t + 4 q + r, q = 0..3: with the units of a width-H layer at positions 0 .. H-1 the H mod 16 units of the last tile are
  // spread over all four of its k-steps (H = 100: one real unit and three zeros in each).  With perm = 1 the pack puts
  // unit 16 T + i of the LAST tile T = (H-1)/16 at position 16 T + 4 (i mod 4) + i / 4, so that they fill k-step 0 first,
  // then k-step 1, ...: the passes whose K runs over that layer stop after ksv[l] = 4 T + ceil((H - 16 T) / 4) k-steps
  // (25 instead of 28 at H = 100, 38 instead of 40 at 150).  Everything else addresses padded positions and is unaffected;
  // mono_pack_k / mono_unpack_k are the only places that translate (mono_pos_of / mono_unit_at).
  int perm;
  int ksv[kMaxNH + 1];                // k-steps of a contraction over the units of hidden layer l = 1..NH
};

// padded position of unit u of a width-H layer, and the unit at padded position p (-1: padding)
__host__ __device__ inline int mono_pos_of(int u, int H, int perm) {
  const int T = (H - 1) / 16;
  if (!perm || u < 16 * T) return u;
  const int i = u - 16 * T;
  return 16 * T + 4 * (i & 3) + (i >> 2);
}
__host__ __device__ inline int mono_unit_at(int p, int H, int perm) {
  const int T = (H - 1) / 16;
  if (!perm || p < 16 * T) return p < H ? p : -1;
  if (p >= 16 * (T + 1)) return -1;
  const int i = p - 16 * T, u = 16 * T + (i >> 2) + 4 * (i & 3);
  return u < H ? u : -1;
}

__host__ __device__ inline MonoLayout make_layout(int HT, int NH, int c) {
  MonoLayout L;
  L.HT = HT; L.HP = 16 * HT; L.NH = NH; L.c = c; L.HM = 0; L.EX = 0;
  L.perm = 0;
  for (int l = 0; l <= kMaxNH; ++l) L.ksv[l] = 4 * HT;
  // Row pitches of the row-major weight images.  A weight fragment is one ds_read_b128 per lane at (row 16 mt + j, floats
  // 16 t + 4 q ..): ds_read_b128 is serviced in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31},
  // ...: MI355X_MICROARCH.md, LDS), i.e. rows {0-3, 12-15} at one q together with rows {4-11} at the next -- the 16-byte
  // slots (pitch / 4) j + q (+1) are distinct mod 16 over such a group iff pitch / 4 = 2 (mod 4).  Until round 5 the pitch
  // was HP + 4 (chosen for contiguous groups): every fragment read of the LDS-resident kernels had a 2-way conflict
  // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.30 in mono_bwd_pair_x_k, 0.20 in mono_fwd_x_k).  HP + 8 for the narrow nets
  // (HT <= 4: every kernel that keeps the image in LDS reads it with b128); wider nets keep HP + 4, which their transposed
  // 4 x ds_read_b32 reads (two 32-lane halves, 16 banks apart) need.
#ifdef GNF_MONO_OLD_LDS                 /* A/B build of the narrow kernels only (tools/build_variant.sh, profiles/r06_mono_lds_ab.txt) */
  L.CP = (c + 15) / 16 * 16; L.LDH = L.CP + 4; L.LDW = L.HP + 4;
#else
  L.CP = (c + 15) / 16 * 16; L.LDH = L.CP + 8; L.LDW = L.HP + (HT <= 4 ? 8 : 4);
#endif
  int o = 0;
  L.o_w1x = o; o += L.HP;
  L.o_b1 = o; o += L.HP;
  L.o_wL = o; o += L.HP;
  L.o_bL = o; o += 4;
  L.o_W1h = o; o += L.HP * L.LDH;
  for (int l = 1; l < NH; ++l) { L.o_W[l] = o; o += L.HP * L.LDW; L.o_b[l] = o; o += L.HP; }
  L.fwd_floats = o;
  for (int l = 1; l < NH; ++l) { L.o_WT[l] = o; o += L.HP * L.LDW; }
  L.o_W1hT = o; o += L.CP * L.LDW;
  L.total_floats = o;
  L.KT32 = (L.HP + 31) / 32;
  for (int l = 1; l < NH; ++l) { L.o_Wf[l] = 0; L.o_WTf[l] = 0; L.o_Wp[l] = 0; L.o_WTp[l] = 0; }
  if (HT >= 7) {
    for (int l = 1; l < NH; ++l) { L.o_Wf[l] = o; o += L.HP * L.HP; L.o_WTf[l] = o; o += L.HP * L.HP; }
    for (int l = 1; l < NH; ++l) { L.o_Wp[l] = o; o += 3 * HT * L.KT32 * 256; }
    for (int l = 1; l < NH; ++l) { L.o_WTp[l] = o; o += 3 * HT * L.KT32 * 256; }
  }
  for (int l = 1; l < NH; ++l) { L.o_Wq[l] = 0; L.o_WTq[l] = 0; }
  if (HT == 4) {
    for (int l = 1; l < NH; ++l) { L.o_Wq[l] = o; o += kNarrowQ; }
    for (int l = 1; l < NH; ++l) { L.o_WTq[l] = o; o += kNarrowQ; }
  }
  L.pack_floats = o;
  return L;
}

struct MonoArgs {
  const float* pack; MonoLayout L;
  const float* x; const float* h; int64_t h_sb, h_sd, h_sc;
  const float* ccw; const float* cct; int S;
  float* z; float* jac;                 // forward outputs
  const float* zt; float* xo;           // inverse: target z, output x
  const int32_t* xo_row; int64_t xo_sd; // inverse, scattered result (gnf_monotonic_inv_scatter): x[xo_row[e / d] + (e % d) * xo_sd]
  int64_t n, d;                         // n = B*d elements
  // backward
  const float* gz; const float* gjac; float* gx; float* gh; int64_t g_sb, g_sd, g_sc;
  float* SA[kMaxNH]; float* SD[kMaxNH]; float* Dsum; float* part;
  int64_t e0, ecount;                   // element chunk [e0, e0+ecount)
  int NK;                               // node slots per group (S+2 rounded up to even)
  int ones;                             // backward: bias gradients of the hidden layers via a ones column (see mono_bwd_k)
  int indw;                             // backward: weight gradients accumulated in the chain kernel (1, 2: narrow nets;
                                        // 3: wide nets, gnf_monotonic_wide.hip)
  float* wpart;                         // [workgroups * kWaves][(NH-1) * HP * HP] accumulator rows of that variant
                                        // (indw = 3, or wcomb: one row per workgroup)
  int wcomb;                            // mono_bwd_pair_x_k: the wavefronts' accumulator rows are added in LDS at the end
  int f32only;                          // host side: keep the fp32-MFMA kernels (gnf_monotonic_fwd_f32 / _bwd_f32)
};

__device__ __forceinline__ void store_inverse(const MonoArgs& a, int64_t e, float v) {
  if (a.xo_row) {
    const int64_t b = e / a.d;
    a.xo[a.xo_row[b] + (e - b * a.d) * a.xo_sd] = v;
  } else {
    a.xo[e] = v;
  }
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// 16 B per lane global -> LDS without passing through registers (global_load_lds_dwordx4): the LDS destination of a
// wave-instruction is lane-linear, which a contiguous copy is.  Completion: s_waitcnt vmcnt(0) before the barrier.
__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// sum over the 4 lane-slots q (lanes j, j+16, j+32, j+48), result in all of them.  gfx950's v_permlane16_swap /
// v_permlane32_swap exchange 16- / 32-lane rows between two registers on the VALU: swapping a value with its own copy
// leaves (row, neighbour row) side by side, one add finishes the level -- no trip through the LDS crossbar
// (ds_bpermute, ~100 cycles of latency per level in the dependent chain of every node evaluation).
__device__ __forceinline__ float qsum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float jsum(float v) {   // sum over the 16 elements of a lane-slot
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
  return v;
}
__device__ __forceinline__ float elu_plus(float s) { return (s > 0.f ? s : expm1f(s)) + 1.05f; }

}  // namespace gnfmono

// gnf_monotonic_wide.hip: backward of wide integrand nets (H = 97..160) with the hidden state of a batch of (element, node)
// pairs in LDS, the output units split over the wavefronts and every weight gradient accumulated in registers.
// ok(): the net's shape has an instantiation and its LDS plan fits.  grid(): persistent workgroups (= rows of a.wpart).
bool gnf_mono_bwd_wide_ok(const gnfmono::MonoLayout& L);
unsigned gnf_mono_bwd_wide_grid(const gnfmono::MonoLayout& L, int64_t n);
int gnf_mono_bwd_wide_launch(const gnfmono::MonoArgs& a, unsigned grid, hipStream_t s, bool true_f32, const char* volatile* kernel);
// forward (z, jac) of the same nets in the same formulation, two workgroups per CU
bool gnf_mono_fwd_wide_ok(const gnfmono::MonoLayout& L);
// true_f32: the fp32-MFMA kernel even when the split-bf16 one is enabled.  *kernel: the family launched.
int gnf_mono_fwd_wide_launch(const gnfmono::MonoArgs& a, hipStream_t s, bool true_f32, const char* volatile* kernel);
