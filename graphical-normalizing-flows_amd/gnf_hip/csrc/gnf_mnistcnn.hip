// Convolutional front of the MNISTCNN embedding net (models/MLP.py:36-41), BACKWARD: see gnf_mnistcnn_fwd.hip for the
// forward and the overall design, gnf_mnistcnn.h for the shared geometry.
#include "gnf_mnistcnn.h"

namespace {

#ifdef GNF_CNN_TIMING
#define TSTAMP(k) do { const long long t__ = __builtin_readcyclecounter(); tacc[k] += t__ - tlast; tlast = t__; } while (0)
#else
#define TSTAMP(k)
#endif

// ---------------------------------------------------------------------------------------------
// Backward with both conv2-sized contractions in the Winograd F(2x2,3x3) domain (2.25x fewer MFMAs):
//   forward   Y = A^T [ sum_c U[o][c] (.) V[c] ] A,  U = G w G^T,  V = B^T d B   (d = 4x4 patch of a1)
//   dW2:      dU[o][c] = sum_tiles (A dY A^T)[o] (.) V[c],  dw = G^T dU G once at the end of the kernel.
//             The 2x2 output tile is the pool window, so dY has ONE non-zero g at the saved argmax (py,px) and
//             Z = A dY A^T = g * alpha_py alpha_px^T with alpha_0 = (1,1,1,0), alpha_1 = (0,1,-1,-1): a few adds on the
//             window of dY2 in LDS.  GEMM per xi: M = o, N = c, K = tiles (144/image).
//   da1:      a 3x3 valid correlation of the zero-bordered dY2 (28x28) with the flipped kernel
//             w'[c][o][a][b] = W2[o][c][2-a][2-b] -> 13x13 tiles of 2x2; U' = G w' G^T lives in LDS (A operand),
//             the lane transforms its own dY2 patch (B operand); the output transform, ReLU gate, dW1/db1
//             partials and the per-tap planes T are lane-local.
//
// Round 4: the next image's conv1 runs UNDER the current image's MFMA work.  Until round 3 an image was {dY2 scatter, de
// gather, conv1 recompute} | barrier | {dW2, da1} | barrier, and the first interval -- 22 % of the time with 8 % of the
// MFMAs, all latency -- could not be overlapped because one image took 159.6 of the 160 KB.  What made room:
//  * dY2 is stored WINDOW-MAJOR without its border: [16 ch][144 pool windows][2x2] = one ds_write_b128 per window, one
//    ds_read_b128 per window for both consumers (the 4x4 patch of a da1 tile is exactly four pool windows; windows off the
//    image come from a zero pad).  41 KB instead of the 70 KB bordered tile-row image.
//  * the per-tap planes T = W1^T dpre1 are written IN PLACE over channels 0..8 of the a1 image: the lane that reads the ReLU
//    gates of (channels 4q..4q+3, tile) is the lane that stores (taps 4q..4q+3, tile), same addresses.  This needs dW2 (which
//    reads all of a1) finished before da1 starts: a barrier between them.  24 KB.
//  * that pays for a SECOND a1 image (2 x 46.8 KB).  Per image (p = parity):
//      Xa  stage e(i+1), dW2(i) [a1[p], dY2], de(i-1) [T in a1[1-p]]                         | barrier
//      Xb  da1(i) [dY2, U', gates a1[p] -> T in place], conv1(i+1) -> a1[1-p], scatter(i+1)    | barrier
//    Both intervals are MFMA work dealt evenly over the SIMDs.  The scatter of the next image's dY2 may start once every
//    wavefront has issued its last dY2 read of this image: an LDS counter, bumped behind each wavefront's last da1 MFMA
//    batch, read (normally already complete) at the end of Xb -- a third barrier interval for it cost 7 % of the image.
// ---------------------------------------------------------------------------------------------
constexpr int NG4 = 11;                              // groups of 16 da1 tiles (13 x 13 = 169 tiles of 2x2)
constexpr int USZ = 16 * 4 * 64;                     // U' as [g][xi_y][lane][xi_x]
// LDS layouts (bank rules of MI355X_MICROARCH.md: ds_read_b64 = 2 groups of 32 lanes over 64 banks, ds_read_b128 = 4
// groups of 16 lanes {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...; checked with a bank model of every gather, 1.00x):
//  a1 [2][16][26][ROWB]: the dW2 gather reads 16 CHANNELS (lane j) x one column quad; hipcc fuses the two b64 halves of a
//     quad into ONE ds_read2_b64 (2 accesses x 4 groups of 16 contiguous lanes over 32 banks): CHB * j mod 32 must be 16
//     distinct even banks (CHB = 2 * odd).  (732 = 4 * odd, the conflict-free stride for plain ds_read_b64, is 2-way
//     conflicted here.)
//  dY2 [16][DCS]: 144 windows x 4 floats + 16 zero chunks.  Channel o sits in slot sigma(o) = (o&3) + 4(o>>3) + 8((o>>2)&1),
//     slots DCS == 8 (mod 64) dwords apart:
//     - da1 (lane (q,j): channel 4q+g, 16 consecutive tiles): a 16-lane group holds 8 lanes of channel o and 8 of o^4
//       (q = 0,1 or 2,3), j in {0-3,12-15} / {4-11} -- the two channels must sit 0 (mod 64) apart (slots s, s+8) so that
//       the 16 consecutive windows tile the 64 banks.  Hence K slot q of step g is channel 4q+g, not 4g+q.
//     - dW2 (lane (q,j): channel j, window 4s+q): 8 channels {0-3,12-15} at one window + 8 channels {4-11} at the next:
//       the 8 slot pairs are 2 chunks apart, the partner channel takes the odd chunk.
//     - a window off the image (tx-1 < 0, ty > 11, ...) reads the zero pad of slot g at chunk rho = (last valid window of
//       the group + 1) mod 16 (+8 for q >= 2): the one residue its 16-lane group leaves free; all such lanes share ONE
//       address (broadcast).
constexpr int ROWB = 28, CHB = 730, A1B = NCH * CHB;
constexpr int ESZB = IMG * IMG;                      // unpadded input image, two buffers
constexpr int DCS = 648, DPAD = 576, DSZW = NCH * DCS;
static_assert(ROWB == IMG, "the de gather indexes the T planes with the pixel index");
static_assert(CHB >= 25 * ROWB + C1 && CHB % 4 == 2, "a1 channel stride");
static_assert(DCS % 64 == 8 && DCS >= DPAD + 64, "dY2 channel slots");
__device__ __forceinline__ int dslot(int o) { return (o & 3) + 4 * (o >> 3) + 8 * ((o >> 2) & 1); }

// conv1 is recomputed in UNITS of 64 consecutive positions of the flat 26 x 26 grid (11 per image, see conv1_units).  Units
// each wavefront takes for the NEXT image inside interval Xb: the da1 groups are 2/2/2/1/1/1/1/1 over the wavefronts
// (3/3/3/2 per SIMD), so SIMD 3 (wavefronts 3, 7) takes most of conv1.  (4/1/1/1/4 measured: SIMD 3 becomes the last
// one, 2.906 -> 2.919 ms; a unit is ~1 000 cycles, the balance cannot get finer than that.  Round 4, with the barrier
// waits per wavefront in hand: eleven other deals incl. 4-unit wavefronts, -DGNF_BWD_C1U=..., all within +-0.3 % of this one.)
// Round 5 (compact de, lighter da1 groups): {0,0,0,4,1,1,2,3} -- one unit each off the second wavefronts of SIMD 0 / 1, which
// reached the end-of-image barrier last (tools/time_cnn_phases.py) -- 2.723 / 2.719 / 2.729 -> 2.714 / 2.711 / 2.708 ms, alternating on
// one box (dense de: 2.890 / 2.886 / 2.882 -> 2.886 / 2.863 / 2.874); fourteen other deals within +-0.5 % or worse
// (profiles/r05_c1u_sweep.txt).
#ifndef GNF_BWD_C1U
#define GNF_BWD_C1U {0, 0, 0, 4, 1, 1, 2, 3}
#endif
__device__ constexpr int C1U[8] = GNF_BWD_C1U;
__device__ constexpr int C1PRO[8] = {2, 2, 2, 1, 1, 1, 1, 1};   // the first image: dealt evenly (nothing to overlap with)
constexpr int c1sum(const int (&a)[8]) { int s = 0; for (int i = 0; i < 8; ++i) s += a[i]; return s; }
constexpr int NU1 = (C1 * C1 + 63) / 64;
static_assert(c1sum(C1U) == NU1 && c1sum(C1PRO) == NU1, "conv1 units");

__device__ __forceinline__ void conv1_run(const float* e_rd, float* a1_wr, int u0, int nu, const float* w1a,
                                          const f32x4& b1v, int q, int j, int lane) {
  if (nu == 4) {                                                                     // (two calls: 4 units at once spill)
    conv1_units<2, CHB>(e_rd, a1_wr, u0, w1a, b1v, q, j, lane);
    conv1_units<2, CHB>(e_rd, a1_wr, u0 + 2, w1a, b1v, q, j, lane);
  } else if (nu == 3) conv1_units<3, CHB>(e_rd, a1_wr, u0, w1a, b1v, q, j, lane);   // nu is wave-uniform
  else if (nu == 2) conv1_units<2, CHB>(e_rd, a1_wr, u0, w1a, b1v, q, j, lane);
  else if (nu == 1) conv1_units<1, CHB>(e_rd, a1_wr, u0, w1a, b1v, q, j, lane);
}

// ---- dW2: dU_xi[o][c] += sum_tiles Z_xi[o][tile] V_xi[c][tile], this wavefront's xi_y half (HY) and its 3 tile rows:
//      9 K-steps of 4 tiles (tile row 3 wq + st / 3, columns q + 4 (st % 3)), every address an immediate offset from two
//      per-lane bases.  Z = A dY A^T from the window (A = [[1,0],[1,1],[1,-1],[0,-1]]); V = B^T d B needs the patch rows
//      HY..HY+2 only: xi_y 0,1 = d0-d2, d1+d2;  xi_y 2,3 = d2-d1, d1-d3.  The operands of step st+1 are requested before
//      the MFMAs of step st; all operands of a step are formed before its 8 back-to-back MFMAs (VALU and MFMA of ONE
//      wavefront do not overlap, tools/mfma_feed.hip).
//      Signs: the accumulators of xi_x = 3 (and, for HY = 1, of the whole -y1 row) collect the NEGATED products; the
//      epilogue flips them once (exact), which saves 2-4 VALU negations per step.
template <int HY>
__device__ __forceinline__ void dw2_phase(const float* av, const float* dz, f32x4 (&dU)[8]) {
  f32x4 zw[2];
  f32x2 dl[2][3], dh[2][3];
  auto load = [&](int st, int b) {
    const int tyl = st / 3, k = st - 3 * tyl;
    zw[b] = *reinterpret_cast<const f32x4*>(dz + 48 * tyl + 16 * k);
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      dl[b][rr] = *reinterpret_cast<const f32x2*>(av + (2 * tyl + rr) * ROWB + 8 * k);
      dh[b][rr] = *reinterpret_cast<const f32x2*>(av + (2 * tyl + rr) * ROWB + 8 * k + 2);
    }
  };
  load(0, 0);
#pragma unroll
  for (int st = 0; st < 9; ++st) {
    const int b = st & 1;
    f32x2 tal, tah, tbl, tbh;                                  // the two rows of B^T d of this half
    float za[2], zb[2];
    if (HY == 0) {
      tal = dl[b][0] - dl[b][2]; tah = dh[b][0] - dh[b][2]; tbl = dl[b][1] + dl[b][2]; tbh = dh[b][1] + dh[b][2];
      za[0] = zw[b][0]; za[1] = zw[b][1]; zb[0] = zw[b][0] + zw[b][2]; zb[1] = zw[b][1] + zw[b][3];
    } else {
      tal = dl[b][1] - dl[b][0]; tah = dh[b][1] - dh[b][0]; tbl = dl[b][0] - dl[b][2]; tbh = dh[b][0] - dh[b][2];
      za[0] = zw[b][0] - zw[b][2]; za[1] = zw[b][1] - zw[b][3]; zb[0] = zw[b][2]; zb[1] = zw[b][3];   // zb = +y1 (negated row)
    }
    float vv[8], zz[8];
    {
      const f32x2 a03 = tal - tah, a12 = pk_v12(tal, tah), b03 = tbl - tbh, b12 = pk_v12(tbl, tbh);
      vv[0] = a03.x; vv[1] = a12.x; vv[2] = a12.y; vv[3] = a03.y;
      vv[4] = b03.x; vv[5] = b12.x; vv[6] = b12.y; vv[7] = b03.y;
    }
    zz[0] = za[0]; zz[1] = za[0] + za[1]; zz[2] = za[0] - za[1]; zz[3] = za[1];      // [3]: -(-za1)
    zz[4] = zb[0]; zz[5] = zb[0] + zb[1]; zz[6] = zb[0] - zb[1]; zz[7] = zb[1];
    if (st + 1 < 9) load(st + 1, b ^ 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int xi = 0; xi < 8; ++xi) dU[xi] = mfma(zz[xi], vv[xi], dU[xi]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// SP: the cotangent of the input image is wanted at the <= KC columns of the plan's row i = image % dplan only (gnf_hip.h,
// "structural zeros of the gate backward"): the de gather shrinks to one half wavefront and a dword store per column, and
// a da1 group whose 16 tiles no such column reaches (bit of the per-row group mask gmt[i], built once per kernel in LDS)
// skips its T planes -- 16 of its 80 MFMAs and the stores.
constexpr int KC = GNF_DAG_PLAN_KC;
template <bool SP>
__device__ __forceinline__ void cnn_bwd_body(const CnnArgs& a, float* smem) {
#ifdef GNF_CNN_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = __builtin_readcyclecounter();
#endif
  float* e_s = smem;                        // [2][28 x 28] input images
  float* a1_s = smem + 2 * ESZB;            // [2][16][CHB] conv1 activations; channels 0..8 become the T planes of the image
  float* d_s = a1_s + 2 * A1B;              // dY2, window-major
  float* u_s = d_s + DSZW;                  // U' as [g][xi_y][lane][xi_x]
  unsigned* cnt_s = reinterpret_cast<unsigned*>(u_s + USZ);      // "wavefronts done reading dY2", counts up over the images
  float* w1_s = u_s + USZ + 4;                                   // W1 as [tap][channel]: conv1's A operands are re-read per call
  unsigned* gmt_s = reinterpret_cast<unsigned*>(w1_s + 9 * NCH); // SP: per plan row, the da1 groups whose T planes a column reads
  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = BWD_WAVES, NT = 64 * BWD_WAVES;
  static_assert(((2 * ESZB) % 4 == 0) && ((2 * A1B) % 4 == 0) && (DSZW % 4 == 0), "d_s / u_s must be 16-B aligned");

  if (tid < 9 * NCH) w1_s[tid] = a.W1[(tid & 15) * 9 + (tid >> 4)];   // (nine registers per lane otherwise: the kernel is at 256)
  const float* w1a = w1_s + j;               // W1 as A operand of conv1: row = channel j (every block), tap t at w1a[16 t]
  f32x4 b1v;
#pragma unroll
  for (int r = 0; r < 4; ++r) b1v[r] = a.b1[4 * q + r];
  float w1t[4];                              // W1^T as A operand of the per-tap planes: row = tap j, K slot q, step r
#pragma unroll
  for (int r = 0; r < 4; ++r) w1t[r] = j < 9 ? a.W1[(4 * q + r) * 9 + j] : 0.f;

  // U'[c = j][o = 4q+g] = G w' G^T, w'[a][b] = W2[o][c][2-a][2-b]; fp64 once, stored for one ds_read_b128 per (g, xi_y)
  if (wave < 4) {
    const int g = wave;
    const float* w = a.W2 + ((4 * q + g) * NCH + j) * 9;
    double gw[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double w0 = w[8 - c], w1 = w[5 - c], w2 = w[2 - c];         // rows a = 0,1,2 of the flipped kernel, column b = c
      gw[0][c] = w0; gw[1][c] = 0.5 * (w0 + w1 + w2); gw[2][c] = 0.5 * (w0 - w1 + w2); gw[3][c] = w2;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x4 u;
      u[0] = (float)gw[r][0];
      u[1] = (float)(0.5 * (gw[r][0] + gw[r][1] + gw[r][2]));
      u[2] = (float)(0.5 * (gw[r][0] - gw[r][1] + gw[r][2]));
      u[3] = (float)gw[r][2];
      *reinterpret_cast<f32x4*>(u_s + ((g * 4 + r) * 64 + lane) * 4) = u;
    }
  }

  // dW2 in the Winograd domain, split by xi_y over the two wavefronts of a SIMD: wavefronts 0-3 own xi_y in {0,1},
  // wavefronts 4-7 xi_y in {2,3}; wavefront (hy, wq) covers the tile rows 3 wq .. 3 wq + 2 (9 of the image's 36 K-steps).
  // dU[4*(xi_y & 1) + xi_x][r] = dU_xi[o = 4q+r][c = j]
  const int hy = wave >> 2, wq = wave & 3;   // wave-uniform
  f32x4 dU[8];
#pragma unroll
  for (int xi = 0; xi < 8; ++xi) dU[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x2 gW1p[2][10];                         // dW1 / db1 per-lane partials: channels 4q+2h, 4q+2h+1 (packed), tap k (9: bias)
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 10; ++k) gW1p[h][k] = f32x2{0.f, 0.f};
  float gb2 = 0.f;                           // thread tid accumulates channel tid/32
  unsigned aprev = 0;                        // argmax of this thread's windows in the previous image, 2 bits each

  for (int i = tid; i < DSZW; i += NT) d_s[i] = 0.f;              // the zero pads stay zero, the windows are rewritten
  for (int i = tid; i < 2 * A1B; i += NT) a1_s[i] = 0.f;          // the pad columns of the a1 / T planes (see de_gather)
  if (tid == 0) *cnt_s = 0u;
  // SP: gmt[i] = the da1 groups (16 consecutive tiles of the 13 x 13 grid) holding a conv1 position (y - ky, x - kx) that the
  // de of one of row i's columns (y, x) sums over -- exactly the T entries de_cols reads.  Built once per workgroup.
  const int16_t* const pcols = SP ? reinterpret_cast<const int16_t*>(a.plan + a.dplan) : nullptr;
  if (SP) {
    for (int i = tid; i < a.dplan; i += NT) gmt_s[i] = 0u;
    __syncthreads();
    for (int t = tid; t < a.dplan * KC; t += NT) {
      const int jj = pcols[t];
      if (jj >= 0) {
        const int y = jj / IMG, x = jj - y * IMG;
        unsigned m = 0u;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int py = y - ky, px = x - kx;
            if ((unsigned)py < (unsigned)C1 && (unsigned)px < (unsigned)C1) m |= 1u << ((13 * (py >> 1) + (px >> 1)) >> 4);
          }
        atomicOr(&gmt_s[t / KC], m);
      }
    }
  }

  constexpr int EPT = (IMG * IMG + NT - 1) / NT, WPT = (PO * PO + 31) / 32;
  float epre[EPT], gpre[WPT];
  unsigned apre[WPT];                        // raw loads only: any arithmetic here would wait for the data and make
  // the per-image streams go through buffer descriptors built per image in SGPRs (base = the image's first byte, extent
  // = one image or 0 behind the last one: out-of-range lanes read 0 / store nothing): ONE 32-bit lane offset per stream
  // instead of a 64-bit address computation per load -- the scatter interval S is VALU-bound
  typedef __amdgpu_buffer_rsrc_t rsrc_t;
  auto rsrc_of = [&](const void* base, int64_t im, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(static_cast<const char*>(base)) + im * bytes, 0,
                                             im < a.n ? bytes : 0, 0x00020000);
  };
  const int vo_e = tid * 4, vo_g = ((tid >> 5) * (PO * PO) + (tid & 31)) * 4, vo_a = (tid >> 5) * (PO * PO) + (tid & 31);
  auto prefetch_e = [&](int64_t im) {        // the prefetch synchronous
    const rsrc_t rs = rsrc_of(a.e, im, IMG * IMG * 4);
#pragma unroll
    for (int k = 0; k < EPT; ++k) epre[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo_e + k * NT * 4, 0, 0));
  };
  auto prefetch_g = [&](int64_t im) {        // (windows >= 144 of a thread read its neighbour's or 0: never used)
    const rsrc_t rg = rsrc_of(a.gp, im, NPOOL * 4), ra = rsrc_of(a.argin, im, NPOOL);
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
      gpre[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rg, vo_g + 128 * k, 0, 0));
      apre[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(ra, vo_a + 32 * k, 0, 0);
    }
  };
  auto stage_e = [&](float* dst) {                               // the prefetched image into an input buffer
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int i = tid + k * NT;
      if (i < IMG * IMG) dst[i] = epre[k];
    }
  };
  // this thread's pool windows (channel tid/32, window (tid&31) + 32k): fixed LDS offsets
  float* const dwin = d_s + dslot(tid >> 5) * DCS + 4 * (tid & 31);

  // per-lane bases of the dW2 gathers (see dw2_phase) and of the da1 tiles of this wavefront's (up to) two groups
  const int zb = dslot(j) * DCS + 4 * (36 * wq + q);
  const int vb = j * CHB + (6 * wq + hy) * ROWB + 2 * q;
  // window offsets [group][TL | TR << 16, BL | BR << 16] in 16-B units from d_s (step g adds g * DCS floats as an immediate),
  // tile offsets [group 0 | group 1 << 16] (0xFFFF: no tile) -- two 16-bit values per register: the kernel sits at the
  // 256-register limit and everything kept per lane across the image loop counts
  unsigned wpk[2][2], tpk = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int grp = wave + NW * k;           // wave-uniform
    const int t = 16 * grp + j;
    const bool ok = grp < NG4 && t < 169;
    const int tc = ok ? t : 0, ty = tc / 13, tx = tc - 13 * ty;
    tpk |= (unsigned)(ok ? 2 * ty * ROWB + 2 * tx : 0xFFFF) << (16 * k);
    wpk[k][0] = wpk[k][1] = 0;
#pragma unroll
    for (int wk = 0; wk < 4; ++wk) {
      const int ay = wk >> 1, ax = wk & 1;
      int vlast = -1;                        // the last valid window of the group (scalar loop: grp is wave-uniform)
      for (int jj = 15; jj >= 0 && vlast < 0; --jj) {
        const int tt = 16 * grp + jj;
        if (grp < NG4 && tt < 169) {
          const int wy = tt / 13 - 1 + ay, wx = tt % 13 - 1 + ax;
          if ((unsigned)wy < 12u && (unsigned)wx < 12u) vlast = 12 * wy + wx;
        }
      }
      const int rho = (vlast + 1 + 8 * (q >> 1)) & 15;
      const int wy = ty - 1 + ay, wx = tx - 1 + ax;
      const bool valid = ok && (unsigned)wy < 12u && (unsigned)wx < 12u;
      const int off = valid ? dslot(4 * q) * DCS + 4 * (12 * wy + wx) : DPAD + 4 * rho;     // floats, a multiple of 4
      wpk[k][wk >> 1] |= (unsigned)(off >> 2) << (16 * (wk & 1));
    }
  }
  static_assert(DSZW / 4 < 65536 && 25 * ROWB + 24 < 0xFFFF, "16-bit packing");
  int ulo = lane * 4;                        // U' fragments: ONE base register, (g, xi_y) as immediate offsets (left to
  asm volatile("" : "+v"(ulo));              // itself hipcc materialises 16 absolute addresses per group)
  const float* ul = u_s + ulo;

  // ---- de[y][x] = sum_tap T[tap][y-ky][x-kx] of a finished image, T in the a1 planes 0..8 (row pitch = IMG, so the
  //      pixel index IS the plane offset).  Nine reads at immediate offsets from ONE base.  Columns need no check: a tap
  //      that falls off the 26 columns lands on the two pad columns of its own or the previous row (or the two pad entries
  //      728, 729 behind the previous plane), which nobody ever writes -- zeroed once at kernel start; tap 0 never moves
  //      left.  Rows off the plane are dropped by a select after the read, three taps at a time.
  static_assert(783 + 8 * CHB - 2 * IMG - 2 < A1B && CHB - 2 >= 0 && CHB == 26 * ROWB + 2, "de gather stays inside the a1 buffer");
  auto de_gather = [&](int64_t im, const float* Tp) {
    const rsrc_t rs = rsrc_of(a.ge, im, IMG * IMG * 4);
#pragma unroll
    for (int k = 0; k < (IMG * IMG + NT - 1) / NT; ++k) {
      if (k * NT + 64 * wave < IMG * IMG) {                      // wave-uniform
        const int i = tid + k * NT;
        const int ic = i < IMG * IMG ? i : 0;
        const int y = (int)(__umul24((unsigned)ic, 2341u) >> 16);   // ic / 28 for ic < 784 (checked exhaustively)
        const float* tp = Tp + ic;
        float t[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) t[tap] = tp[tap * CHB - (tap / 3) * IMG - tap % 3];
        float s = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float r3 = (t[3 * ky] + t[3 * ky + 1]) + t[3 * ky + 2];
          s += (unsigned)(y - ky) < (unsigned)C1 ? r3 : 0.f;
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s), rs, i * 4, 0, 0);   // i >= 784: out of range, dropped
      }
    }
  };

  // SP: de at the plan's columns only -- lane k < KC of ONE wavefront takes column cj = cols[row][k] (-1: none), same nine
  // reads and row selects as above, one dword into the compact slab ge_cols[image][k]
  auto de_cols = [&](int64_t im, int cj, const float* Tp) {
    const rsrc_t rs = rsrc_of(a.gec, im, KC * 4);
    const int ic = cj >= 0 ? cj : 0;
    const int y = (int)(__umul24((unsigned)ic, 2341u) >> 16);
    const float* tp = Tp + ic;
    float t[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) t[tap] = tp[tap * CHB - (tap / 3) * IMG - tap % 3];
    float sum = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const float r3 = (t[3 * ky] + t[3 * ky + 1]) + t[3 * ky + 2];
      sum += (unsigned)(y - ky) < (unsigned)C1 ? r3 : 0.f;
    }
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), rs, cj >= 0 && lane < KC ? lane * 4 : KC * 4, 0, 0);
  };

  // ---- prologue: the first image's input, its conv1 (dealt evenly), the prefetches of what the first Xa consumes
  const int64_t img0 = blockIdx.x, gstride = gridDim.x;
  // SP: plan row (= image % dplan) of the previous / this image, advanced without a 64-bit division per image
  const int rstep = SP ? (int)(gstride % a.dplan) : 0;
  int rowc = SP ? (int)(img0 % a.dplan) : 0, rowp = 0;
  prefetch_e(img0);
  prefetch_g(img0);
  stage_e(e_s);
  prefetch_e(img0 + gstride);
  __syncthreads();                                               // e(0) staged, U' and the dY2 zeros written
  if (img0 < a.n) {
    int u0 = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) u0 += w < wave ? C1PRO[w] : 0;
    conv1_run(e_s, a1_s, u0, C1PRO[wave], w1a, b1v, q, j, lane);
  }

  // ---- dY2 = pool-backward scatter of g_pooled: every window holds ONE non-zero, so the previous image's entry is
  //      cleared and the new one written (two dword stores instead of four compares + four selects + a 16-B store)
  auto scatter = [&]() {
    unsigned anew = 0;
#pragma unroll
    for (int k = 0; k < WPT; ++k)
      if ((tid & 31) + 32 * k < PO * PO) {
        const float g = gpre[k];
        const unsigned am = apre[k];
        gb2 += g;
        float* w = dwin + 128 * k;
        w[(aprev >> (2 * k)) & 3u] = 0.f;
        w[am] = g;                                               // same lane, program order: wins over the clear
        anew |= am << (2 * k);
      }
    aprev = anew;
  };
  if (img0 < a.n) scatter();                                     // the first image's (dY2 is all zeros behind the barrier above)

  int par = 0;                                                   // buffers of this image (wave-uniform)
  unsigned rd_target = 0;                                        // cnt_s once every wavefront is done with dY2 of this image
  for (int64_t img = img0; img < a.n; img += gstride, par ^= 1) {
    float* a1p = a1_s + par * A1B;                               // a1 of this image -> its T planes
    float* a1n = a1_s + (par ^ 1) * A1B;                         // T planes of the previous image -> a1 of the next one
    const float* e_rd = e_s + par * ESZB;
    float* e_nx = e_s + (par ^ 1) * ESZB;
    rd_target += NW;
    __syncthreads();                                             // a1p (conv1), dY2 (scatter) of this image and the previous
    TSTAMP(0);                                                   // image's T planes complete; its input buffer is free
    // ---- Xa: stage the next image (requested one image ago), request what the image after needs, dW2 (9 K-steps per
    //      wavefront), the previous image's de (last: nothing waits on vmcnt right behind its global stores)
    stage_e(e_nx);
    prefetch_e(img + 2 * gstride);
    prefetch_g(img + gstride);
    int cjp = -1;                                                // SP: the previous image's column of this lane (wavefront 0),
    if (SP && wave == 0 && img != img0 && lane < KC) cjp = pcols[rowp * KC + lane];   // requested ahead of dW2
    unsigned gm = 0x7FFu;                                        // da1 groups whose T planes are wanted
    if (SP) gm = __builtin_amdgcn_readfirstlane(gmt_s[rowc]);
    TSTAMP(1);
    if (hy == 0) dw2_phase<0>(a1p + vb, d_s + zb, dU);
    else dw2_phase<1>(a1p + vb, d_s + zb, dU);
    TSTAMP(2);
    if (SP) {
      if (wave == 0 && img != img0) de_cols(img - gstride, cjp, a1n);
    } else if (img != img0) de_gather(img - gstride, a1n);
#ifdef GNF_CNN_EXP_NOBMID          // measurement only (wrong results): what the barrier between the two intervals costs
    if (!SP)
#endif
    __syncthreads();
    TSTAMP(3);
    // ---- Xb: da1 of this image and conv1 of the next one.  Of the two wavefronts of a SIMD one starts with its conv1
    //      tiles (latency-bound) while the other starts with a da1 group (MFMA-bound)
    const bool has_next = img + gstride < a.n;
    int c1t0 = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) c1t0 += w < wave ? C1U[w] : 0;
    const bool c1_first = wave != 7;                             // SIMD 3 holds two conv1-heavy wavefronts: one starts with conv1,
#ifdef GNF_CNN_EXP_NOCONV1
    const bool c1_on = false;
#else
    const bool c1_on = has_next;
#endif
    if (c1_on && c1_first) {                                     // the other with its da1 group
      conv1_run(e_nx, a1n, c1t0, C1U[wave], w1a, b1v, q, j, lane);
    }
    TSTAMP(4);
    // ---- da1: dpre1 = conv2^T(dY2) * gate on 2x2 tiles: per 4 dY2 channels (step g: channels 4q+g) 16 operands, then 16 MFMAs
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int grp = wave + NW * k;                            // wave-uniform
      if (grp < NG4) {
        const unsigned tk = (tpk >> (16 * k)) & 0xFFFFu;
        const bool ok = tk != 0xFFFFu;
        const int to = ok ? (int)tk : 0;
        f32x4 m[16];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) m[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* pwin[4];                                   // the four windows of the patch: TL, TR, BL, BR
#pragma unroll
        for (int wk = 0; wk < 4; ++wk) pwin[wk] = d_s + 4 * ((wpk[k][wk >> 1] >> (16 * (wk & 1))) & 0xFFFFu);
        f32x4 pw[4];
#pragma unroll
        for (int wk = 0; wk < 4; ++wk) pw[wk] = *reinterpret_cast<const f32x4*>(pwin[wk]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 uf[4];
#pragma unroll
          for (int xy = 0; xy < 4; ++xy) uf[xy] = *reinterpret_cast<const f32x4*>(ul + (g * 4 + xy) * 256);
          f32x2 plo[4], phi[4];                                 // patch rows as two column pairs
          plo[0] = f32x2{pw[0][0], pw[0][1]}; phi[0] = f32x2{pw[1][0], pw[1][1]};
          plo[1] = f32x2{pw[0][2], pw[0][3]}; phi[1] = f32x2{pw[1][2], pw[1][3]};
          plo[2] = f32x2{pw[2][0], pw[2][1]}; phi[2] = f32x2{pw[3][0], pw[3][1]};
          plo[3] = f32x2{pw[2][2], pw[2][3]}; phi[3] = f32x2{pw[3][2], pw[3][3]};
          float vv[16];
          wino_in(plo, phi, vv);
          if (g < 3) {                                          // the next channel step's windows: in flight under the MFMAs
#pragma unroll
            for (int wk = 0; wk < 4; ++wk) pw[wk] = *reinterpret_cast<const f32x4*>(pwin[wk] + (g + 1) * DCS);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int xi = 0; xi < 16; ++xi) m[xi] = mfma(uf[xi >> 2][xi & 3], vv[xi], m[xi]);
          __builtin_amdgcn_sched_barrier(0);
        }
        // this wavefront's last read of dY2 is behind it: tell the others (LDS executes a wavefront's operations in order)
        if (grp + NW >= NG4 && lane == 0)
          __hip_atomic_fetch_add(cnt_s, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        f32x4 dp[4];                                            // dpre1 at sub-position p, channels 4q+r
        float* const pa = a1p + 4 * q * CHB + to;               // gates of (channel 4q+r, tile) -> T of (tap 4q+r, tile)
#pragma unroll
        for (int r = 0; r < 4; ++r) {                           // A^T M A, A^T = [[1,1,1,0],[0,1,-1,-1]]
          float s0[4], s1[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            s0[c] = m[c][r] + m[4 + c][r] + m[8 + c][r];
            s1[c] = m[4 + c][r] - m[8 + c][r] - m[12 + c][r];
          }
          const float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
          const float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
          // ReLU gate = (a1 > 0), read back from the a1 image (two ds_read_b64 per channel: the tile's two rows)
          const float2 g0 = *reinterpret_cast<const float2*>(pa + r * CHB), g1 = *reinterpret_cast<const float2*>(pa + r * CHB + ROWB);
          dp[0][r] = (ok && g0.x > 0.f) ? y00 : 0.f;
          dp[1][r] = (ok && g0.y > 0.f) ? y01 : 0.f;
          dp[2][r] = (ok && g1.x > 0.f) ? y10 : 0.f;
          dp[3][r] = (ok && g1.y > 0.f) ? y11 : 0.f;
        }
        // dW1 / db1 partials against the 4x4 image patch of this tile (2 ty * IMG + 2 tx = the tile's a1 offset: ROWB == IMG)
        float ep[4][4];
        {
          const float* pe = e_rd + to;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const float2 lo = *reinterpret_cast<const float2*>(pe + rr * IMG);
            const float2 hi = *reinterpret_cast<const float2*>(pe + rr * IMG + 2);
            ep[rr][0] = lo.x; ep[rr][1] = lo.y; ep[rr][2] = hi.x; ep[rr][3] = hi.y;
          }
        }
#ifndef GNF_CNN_EXP_NODW1
#pragma unroll
        for (int h = 0; h < 2; ++h) {                           // packed over the channel pair (2 flops per lane per op)
          f32x2 d2[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) d2[p] = f32x2{dp[p][2 * h], dp[p][2 * h + 1]};
          gW1p[h][9] += (d2[0] + d2[1]) + (d2[2] + d2[3]);
          // position outermost: nine INDEPENDENT accumulators per pass (tap outermost is a dependent chain of four
          // v_pk_fma_f32 per tap, each behind an s_nop)
#pragma unroll
          for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
              const float ev = ep[(p >> 1) + tap / 3][(p & 1) + tap % 3];
              gW1p[h][tap] = __builtin_elementwise_fma(d2[p], f32x2{ev, ev}, gW1p[h][tap]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#endif
        // T[tap][pos] = sum_oc W1[oc][tap] dpre1[oc][pos]: dpre1 in the C/D layout IS the B operand
        if (!SP || ((gm >> grp) & 1u)) {                        // (wave-uniform) SP: only where a plan column reads it
          f32x4 tq[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) tq[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int p = 0; p < 4; ++p) tq[p] = mfma(w1t[r], dp[p][r], tq[p]);
          // plane of tap 4q+r = channel plane 4q+r of this a1 buffer, at the addresses the gates came from
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ok && 4 * q + r < 9) {                           // taps 9..15 of the MFMA tile and idle lanes: nothing to store
              *reinterpret_cast<float2*>(pa + r * CHB) = make_float2(tq[0][r], tq[1][r]);
              *reinterpret_cast<float2*>(pa + r * CHB + ROWB) = make_float2(tq[2][r], tq[3][r]);
            }
        }
      }
      TSTAMP(5 + k);
    }
    if (c1_on && !c1_first) {
      conv1_run(e_nx, a1n, c1t0, C1U[wave], w1a, b1v, q, j, lane);
    }
    // the next image's dY2 as soon as EVERY wavefront is done reading this one's -- normally long before this point (the
    // last group's output transform, dW1 partials and T planes lie in between): no separate barrier interval for the scatter
    if (has_next) {
      while (__hip_atomic_load(cnt_s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < rd_target) __builtin_amdgcn_s_sleep(1);
      scatter();
    }
    if (SP) { rowp = rowc; rowc += rstep; rowc -= rowc >= a.dplan ? a.dplan : 0; }
    TSTAMP(7);
  }
  __syncthreads();
  if (img0 < a.n) {                                              // the last image's de (par was flipped once more)
    const int64_t last = img0 + (a.n - 1 - img0) / gstride * gstride;
    if (SP) {
      if (wave == 0) de_cols(last, lane < KC ? (int)pcols[rowp * KC + lane] : -1, a1_s + (par ^ 1) * A1B);
    } else de_gather(last, a1_s + (par ^ 1) * A1B);
  }
#ifdef GNF_CNN_TIMING
  if (blockIdx.x == 7 && (tid & 63) == 0)
    for (int k = 0; k < 8; ++k) a.part[((int64_t)gridDim.x * NW + 1) * PROW + wave * 8 + k] = (float)tacc[k];
#endif

  // ---- per-wave partial row: dW2 [16][144] | dW1+db1 [16][16] | db2 [16] -- into LDS (the images are done), the eight
  //      rows of the workgroup are then added in wavefront order: 256 partial rows for the row-sum launch, not 2048
  __syncthreads();                                     // the last de gather has read its T planes
  float* prow = smem + wave * PROW;
  // the accumulators that collected negated products (dw2_phase): xi_x = 3 of both rows; for hy = 1 the whole second row
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    dU[3][r] = -dU[3][r];
    if (hy == 0) dU[7][r] = -dU[7][r];
    else { dU[4][r] = -dU[4][r]; dU[5][r] = -dU[5][r]; dU[6][r] = -dU[6][r]; }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {          // this half's share of dw = G^T dU G for (o = 4q+r, c = j); the halves add up
    float ar[3][4];                      // G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const float ua = dU[x][r], ub = dU[4 + x][r];            // xi_y = 2hy, 2hy+1
      if (hy == 0) { ar[0][x] = ua + 0.5f * ub; ar[1][x] = 0.5f * ub; ar[2][x] = 0.5f * ub; }
      else { ar[0][x] = 0.5f * ua; ar[1][x] = -0.5f * ua; ar[2][x] = 0.5f * ua + ub; }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float* o = prow + (4 * q + r) * 144 + j * 9 + 3 * i;
      o[0] = ar[i][0] + 0.5f * (ar[i][1] + ar[i][2]);
      o[1] = 0.5f * (ar[i][1] - ar[i][2]);
      o[2] = 0.5f * (ar[i][1] + ar[i][2]) + ar[i][3];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      float v = gW1p[r >> 1][k][r & 1];                        // sum over the 16 tile lanes
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
      if (j == 0) prow[NCH * 144 + (4 * q + r) * 16 + k] = v;
    }
  if (j == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int k = 10; k < 16; ++k) prow[NCH * 144 + (4 * q + r) * 16 + k] = 0.f;
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) gb2 += __shfl_xor(gb2, off, 64);    // over the 32 threads of a channel
  if ((lane & 31) == 0) prow[NCH * 144 + NCH * 16 + 2 * wave + (lane >> 5)] = gb2;
  if (lane < NCH && (lane >> 1) != wave) prow[NCH * 144 + NCH * 16 + lane] = 0.f;
  __syncthreads();
  float* grow = a.part + (int64_t)blockIdx.x * PROW;
  for (int i = tid; i < PROW; i += 64 * NW) {
    float v = smem[i];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += smem[w * PROW + i];
    grow[i] = v;
  }
}

// With a plan: compact de unless one of its rows holds more than KC columns -- decided on the device from the plan's
// overflow word (the gate backward takes the same decision from the same word).
__global__ __launch_bounds__(64 * BWD_WAVES) void cnn_bwd_wino_k(CnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (a.plan && a.plan[a.dplan + (a.dplan * KC + 1) / 2] == 0) {   // the plan's overflow word (gnf_hip.h)
    cnn_bwd_body<true>(a, smem);
    return;
  }
  cnn_bwd_body<false>(a, smem);
}

// the workgroups' partial rows summed in row order (64 columns per workgroup, wavefront w of 16 takes rows w, w + 16, ...,
// the 16 sums meet in LDS: deterministic) and written straight into the parameter-shaped gradients
__global__ __launch_bounds__(1024) void cnn_reduce_unpack_k(const float* __restrict__ part, int rows, float* gW1, float* gb1,
                                                            float* gW2, float* gb2) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f;
  if (n < PROW) {
    const float* p = part + n;
    int r = wave;
    for (; r + 16 < rows; r += 32) { s0 += p[(int64_t)r * PROW]; s1 += p[(int64_t)(r + 16) * PROW]; }
    for (; r < rows; r += 16) s0 += p[(int64_t)r * PROW];
  }
  red[wave][lane] = s0 + s1;
  __syncthreads();
  if (wave != 0 || n >= PROW) return;
  float s = red[0][lane];
#pragma unroll
  for (int w = 1; w < 16; ++w) s += red[w][lane];
  if (n < NCH * 144) gW2[n] = s;
  else if (n < NCH * 144 + NCH * 16) {
    const int k = n - NCH * 144, oc = k >> 4, c = k & 15;
    if (c < 9) gW1[oc * 9 + c] = s;
    else if (c == 9) gb1[oc] = s;
  } else gb2[n - NCH * 144 - NCH * 16] = s;
}

constexpr int kPlanRows = IMG * IMG;                // the plan's rows are the masked copies of ONE sample: d = 784
constexpr size_t kBwdWinoLds = (size_t)(2 * ESZB + 2 * A1B + DSZW + USZ + 4 + 9 * NCH + kPlanRows) * sizeof(float);
static_assert(kBwdWinoLds <= 160 * 1024, "conv backward LDS image");
static_assert((size_t)BWD_WAVES * PROW * sizeof(float) <= kBwdWinoLds, "the partial rows of the epilogue reuse the image LDS");
// one 8-wave workgroup per CU: at its 256 VGPRs a second one is not admitted
constexpr unsigned kBwdGrid = 256;

}  // namespace

extern "C" {

int64_t gnf_mnistcnn_conv_bwd_ws_bytes(int64_t n_img) {
  (void)n_img;
  return ((int64_t)kBwdGrid * BWD_WAVES + 2) * PROW * (int64_t)sizeof(float);
}

int gnf_mnistcnn_conv_bwd(const float* e, const float* W1, const float* b1, const float* W2, const float* g_pooled,
                          const unsigned char* argmax, float* ge, float* gW1, float* gb1, float* gW2, float* gb2,
                          void* ws, int64_t ws_bytes, int64_t n_img, gnf_stream_t stream) {
  return gnf_mnistcnn_conv_bwd_cols(e, W1, b1, W2, g_pooled, argmax, ge, nullptr, 0, nullptr, gW1, gb1, gW2, gb2, ws,
                                    ws_bytes, n_img, stream);
}

int gnf_mnistcnn_conv_bwd_cols(const float* e, const float* W1, const float* b1, const float* W2, const float* g_pooled,
                               const unsigned char* argmax, float* ge, const int32_t* plan, int64_t d_plan,
                               float* ge_cols, float* gW1, float* gb1, float* gW2, float* gb2, void* ws,
                               int64_t ws_bytes, int64_t n_img, gnf_stream_t stream) {
  if (((!e || !g_pooled || !argmax || !ge) && n_img > 0) || !W1 || !b1 || !W2 || !gW1 || !gb1 || !gW2 || !gb2 || !ws ||
      n_img < 0)
    return GNF_EINVAL;                       // empty batch: zero weight gradients through the same kernels
  if (plan && (!ge_cols && n_img > 0)) return GNF_EINVAL;
  if (plan && d_plan != kPlanRows) return GNF_ESHAPE;              // a masked copy per pixel of the 28 x 28 image
  if (ws_bytes < gnf_mnistcnn_conv_bwd_ws_bytes(n_img)) return GNF_EWS;
  CnnArgs a{};
  a.e = e; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.gp = g_pooled; a.argin = argmax; a.ge = ge; a.part = (float*)ws;
  a.n = n_img;
  a.plan = plan; a.gec = ge_cols; a.dplan = (int)d_plan;
  // fixed grid: every workgroup (also one without images) writes its partial rows
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cnn_bwd_wino_k),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdWinoLds);
  hipLaunchKernelGGL(cnn_bwd_wino_k, dim3(kBwdGrid), dim3(64 * BWD_WAVES), kBwdWinoLds, (hipStream_t)stream, a);
  GNF_LAUNCH_CHECK();
  // one partial row per workgroup -> the four gradients, one launch
  hipLaunchKernelGGL(cnn_reduce_unpack_k, dim3((PROW + 63) / 64), dim3(1024), 0, (hipStream_t)stream, (const float*)ws,
                     (int)kBwdGrid, gW1, gb1, gW2, gb2);
  GNF_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
