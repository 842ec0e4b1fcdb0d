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
//             window of the dY2 image in LDS.  GEMM per xi: M = o, N = c, K = tiles (144/image).
//   da1:      a 3x3 valid correlation of the zero-bordered dY2 (28x28) with the flipped kernel
//             w'[c][o][a][b] = W2[o][c][2-a][2-b] -> 13x13 tiles of 2x2; U' = G w' G^T lives in LDS (A operand),
//             the lane transforms its own dY2 patch (B operand); the output transform, ReLU gate, dW1/db1
//             partials and the per-tap planes T are lane-local as in the direct kernel.
// conv1 is recomputed per image as 43 flat tiles of 16 positions dealt over the wavefronts; da1 reads its ReLU gates
// back from the a1 image.  Per image: {dY2 scatter, de gather of the previous image, conv1} | barrier | {dW2, da1,
// staging of the next input image} | barrier.
// ---------------------------------------------------------------------------------------------
constexpr int NG4 = 11;                              // groups of 16 da1 tiles (13 x 13 = 169 tiles of 2x2)
constexpr int USZ = 16 * 4 * 64;                     // U' as [xi_y][g][lane][xi_x]
// LDS layouts of this kernel.  Every gather is a ds_read_b64 whose 16-lane (ds_read2) / 32-lane groups must spread
// over the 64 banks; with the direct kernel's strides the dW2 gather was 4-way and the da1 gather 2-way conflicted
// and both phases LDS-bound (measured: 4.4k of 11.5k and ~5k of 17k cycles per image).
//  a1 [16][26][ROWB]: the dW2 gather reads 16 CHANNELS x one column pair per 16-lane group -> CHB*j mod 64 must be
//     16 distinct even banks (CHB = 2 * odd).
//  dY2 (zero-bordered 28x28): the two channels of a PAIR share [14 tile rows][TRD = 154]: image rows 2t / 2t+1 of the even
//     channel at +0 / +ROD, of the odd channel at +CHDW / +CHDW+ROD (40 floats of each tile row stay unused).  The da1
//     gather reads 16 consecutive 2x2 TILES (13 per tile row) of 2 channels per 32-lane group -> consecutive tiles are
//     +2 dwords, the tile-row wrap TRD - 24 == 2 (mod 64), the two channels of a pair CHDW == 32 (mod 64) apart;
//     channel pairs PSD == 2 (mod 64) apart keep the 16-channel Z read of dW2 conflict-free too.  70 KB; round 1 gave
//     every channel its own [14][90] block (84 KB, same timings) -- the 14 KB are what lets the T planes leave the a1 region.
constexpr int ROWB = 28, CHB = 730;
// the input image, unpadded rows (no read leaves a row: conv1 x+kx <= 27, the dW1 patches 2tx+3 <= 27), TWO buffers:
// image n+1 is staged while image n is still being read, which removes the staging barrier interval
constexpr int ROWEB = IMG, ESZB = IMG * IMG;
constexpr int TRD = 154, ROD = 28, CHDW = 96, PSD = 14 * TRD + 22, DSZW = (NCH / 2) * PSD;
static_assert(CHB >= C1 * ROWB && (CHB % 4) == 2, "a1 channel stride");
static_assert((TRD - 24) % 64 == 2 && CHDW % 64 == 32 && PSD % 64 == 2 && 14 * TRD <= PSD + 2 && ROD >= 28, "dY2 layout");
__device__ __forceinline__ int dofs(int c) { return (c >> 1) * PSD + (c & 1) * CHDW; }

__global__ __launch_bounds__(64 * BWD_WAVES) void cnn_bwd_wino_k(CnnArgs a) {
#ifdef GNF_CNN_TIMING
  long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = __builtin_readcyclecounter();
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* a1_s = smem + 2 * ESZB;            // conv1 activations (the two input-image buffers sit in front)
  float* d_s = a1_s + NCH * CHB;            // dY2 with a 2-wide zero border, tile-row layout
  float* T_s = d_s + DSZW;                  // per-tap planes T [9][26 x 26], their own region: dW2 and da1 share one phase
  float* u_s = T_s + 9 * CS;                // U' as [g][xi_y][lane][xi_x]
  const int tid = threadIdx.x, lane = tid & 63, q = lane >> 4, j = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = BWD_WAVES, NT = 64 * BWD_WAVES;
  static_assert(((2 * ESZB) % 4 == 0) && ((NCH * CHB) % 4 == 0) && (DSZW % 4 == 0), "u_s must be 16-B aligned");
  static_assert((9 * CS) % 4 == 0 && 9 * CS <= NCH * CHB, "T planes");

  float w1f[3];
  int off1[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = 4 * s + q;
    w1f[s] = tap < 9 ? a.W1[j * 9 + tap] : 0.f;
    const int tt = tap < 9 ? tap : 0;
    off1[s] = (tt / 3) * ROWEB + tt % 3;
  }
  f32x4 b1v;
#pragma unroll
  for (int r = 0; r < 4; ++r) b1v[r] = a.b1[4 * q + r];
  float w1t[4];                              // W1^T as A operand of the per-tap planes: row = tap j, K slot q, step r
#pragma unroll
  for (int r = 0; r < 4; ++r) w1t[r] = j < 9 ? a.W1[(4 * q + r) * 9 + j] : 0.f;

  // U'[c = j][o = 4g+q] = G w' G^T, w'[a][b] = W2[o][c][2-a][2-b]; fp64 once, stored for one ds_read_b128 per (g, xi_y)
  if (wave < 4) {
    const int g = wave;
    const float* w = a.W2 + ((4 * g + q) * NCH + j) * 9;
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
  // wavefronts 4-7 xi_y in {2,3}; each covers all 36 K-steps of an image with its three partners (s = wave&3 mod 4).
  // dU[4*(xi_y & 1) + xi_x][r] = dU_xi[o = 4q+r][c = j]
  const int hy = wave >> 2;                  // wave-uniform
  f32x4 dU[8];
#pragma unroll
  for (int xi = 0; xi < 8; ++xi) dU[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x2 gW1p[2][10];                         // dW1 / db1 per-lane partials: channels 4q+2h, 4q+2h+1 (packed), tap k (9: bias)
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 10; ++k) gW1p[h][k] = f32x2{0.f, 0.f};
  float gb2 = 0.f;                           // thread tid accumulates channel tid/32

  for (int i = tid; i < DSZW; i += NT) d_s[i] = 0.f;

  constexpr int EPT = (IMG * IMG + NT - 1) / NT, WPT = (PO * PO + 31) / 32;
  float epre[EPT], gpre[WPT];
  unsigned apre[WPT];                        // raw loads only: any arithmetic here would wait for the data and make
  auto prefetch = [&](int64_t im) {          // the prefetch synchronous
    const bool on = im < a.n;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int i = tid + k * NT;
      epre[k] = (on && i < IMG * IMG) ? a.e[im * (IMG * IMG) + i] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
      const int w = (tid & 31) + 32 * k;
      const bool ok = on && w < PO * PO;
      const int64_t o = im * NPOOL + (tid >> 5) * (PO * PO) + w;
      gpre[k] = ok ? a.gp[o] : 0.f;
      apre[k] = ok ? (unsigned)a.argin[o] : 0u;
    }
  };
  prefetch(blockIdx.x);
  // this thread's pool windows (channel tid/32, window (tid&31) + 32k) and pixels: fixed LDS offsets
  const int dwin = dofs(tid >> 5) + TRD + 2;
  int woff[WPT];
#pragma unroll
  for (int k = 0; k < WPT; ++k) {
    const int w = (tid & 31) + 32 * k;
    woff[k] = w < PO * PO ? dwin + (w / PO) * TRD + 2 * (w % PO) : -1;
  }

  // ---- de[y][x] = sum_tap T[tap][y-ky][x-kx] of a finished image.  Branch-free: the nine reads are immediate offsets
  //      from ONE base and always land inside the T region (largest: 27*26+27 + 8*CS - 2*C1 - 2 = 9*CS - 1), taps that
  //      fall off the 26 x 26 plane are dropped by a select AFTER the read -- the bounds-checked form compiled into nine
  //      dependent branch / ds_read / s_waitcnt rounds per pixel (2.3 k of 27 k cycles per image)
  static_assert(27 * C1 + 27 + 8 * CS - 2 * C1 - 2 < 9 * CS, "de gather stays inside the T planes");
  auto de_gather = [&](int64_t im) {
#pragma unroll
    for (int k = 0; k < (IMG * IMG + NT - 1) / NT; ++k) {
      if (k * NT + 64 * wave < IMG * IMG) {                      // wave-uniform
        const int i = tid + k * NT;
        const bool on = i < IMG * IMG;
        const int ic = on ? i : 0, y = ic / IMG, x = ic - IMG * y;
        const float* tp = T_s + y * C1 + x;
        float t[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) t[tap] = tp[tap * CS - (tap / 3) * C1 - tap % 3];
        float s = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const bool v = (unsigned)(y - tap / 3) < (unsigned)C1 && (unsigned)(x - tap % 3) < (unsigned)C1;
          s += v ? t[tap] : 0.f;
        }
        if (on) a.ge[im * (IMG * IMG) + i] = s;
      }
    }
  };

  auto stage_e = [&](float* dst) {                               // the prefetched image into an input buffer
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int i = tid + k * NT;
      if (i < IMG * IMG) dst[i] = epre[k];
    }
  };
  stage_e(smem);

  // TWO barrier intervals per image: {dY2 scatter, the PREVIOUS image's de gather, conv1 recompute} | {dW2, da1, staging
  // of the NEXT image into the other input buffer}.  de sits in the first interval so that its LDS latency overlaps the
  // conv1 MFMAs and nothing waits on vmcnt right behind its global stores (at the end of the loop body the compiler's
  // s_waitcnt vmcnt(0) for the prefetched bytes of the next image also waited for the store acknowledgements).
  int par = 0;                                                   // input buffer of this image (wave-uniform)
  for (int64_t img = blockIdx.x; img < a.n; img += gridDim.x, par ^= 1) {
    const float* e_rd = smem + par * ESZB;
    __syncthreads();                                             // previous image done: dY2, a1 free, T complete; e staged
    // ---- dY2 = pool-backward scatter of g_pooled (one write per conv2 position)
#pragma unroll
    for (int k = 0; k < WPT; ++k)
      if (woff[k] >= 0) {
        const float g = gpre[k];
        const int am = (int)apre[k];
        gb2 += g;
        float* p = d_s + woff[k];                                 // even offset: the two rows of the window as 8-B stores
        *reinterpret_cast<float2*>(p) = make_float2(am == 0 ? g : 0.f, am == 1 ? g : 0.f);
        *reinterpret_cast<float2*>(p + ROD) = make_float2(am == 2 ? g : 0.f, am == 3 ? g : 0.f);
      }
    prefetch(img + gridDim.x);
    TSTAMP(0);
    if (img != (int64_t)blockIdx.x) de_gather(img - gridDim.x);
    TSTAMP(4);
    // ---- P1: conv1 + ReLU, 43 tiles of 16 consecutive positions of the 26 x 26 grid dealt over the 8 wavefronts
    //      (6 / 5 each).  a1 stays intact until the end of the image (the T planes have their own region), so da1 reads
    //      its ReLU gates back from a1 itself and this phase no longer has to follow da1's 11-groups-over-8 layout
    {
      constexpr int NTL = (43 + NW - 1) / NW;
      int po[NTL];
      f32x4 acc[NTL];
      float ev[NTL][3];
#pragma unroll
      for (int k = 0; k < NTL; ++k) {
        const int pos = 16 * (wave + NW * k) + j;
        const int pc = pos < C1 * C1 ? pos : 0;
        const int y = pc / C1, x = pc - y * C1;
        po[k] = pos < C1 * C1 ? y * ROWB + x : -1;
#pragma unroll
        for (int s = 0; s < 3; ++s) ev[k][s] = e_rd[y * ROWEB + x + off1[s]];
        acc[k] = b1v;
      }
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int k = 0; k < NTL; ++k) acc[k] = mfma(w1f[s], ev[k][s], acc[k]);
#pragma unroll
      for (int k = 0; k < NTL; ++k)
        if (po[k] >= 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) a1_s[(4 * q + r) * CHB + po[k]] = fmaxf(acc[k][r], 0.f);
        }
    }
    __syncthreads();
    TSTAMP(1);
    // ---- P3: dU_xi[o][c] += sum_tiles Z_xi[o][tile] V_xi[c][tile]; K-step s = 4 tiles; this wavefront's xi_y half.
    //      Z = A dY A^T from the 2x2 window of dY2 in LDS (A = [[1,0],[1,1],[1,-1],[0,-1]]); V = B^T d B needs the
    //      patch rows hy..hy+2 only: xi_y 0,1 = d0-d2, d1+d2;  xi_y 2,3 = d2-d1, d1-d3
    // K-steps of this wavefront.  With dW2 and da1 in ONE barrier interval the wavefronts that own two da1 groups
    // (0-2) take fewer dW2 steps
    const int s_step = 1;
    // K-steps of the last wavefront of each half (3 and 7: one da1 group and SIMD 3 to themselves).  Measured sweep
    // (tools/bench_cnn.py, same box): (18,9) 3.32 ms, (17,9) 3.22, (16,9) 3.20, (15,9) 3.23, (14,9) 3.25, (18,6) 3.24,
    // (16,8) 3.23, (15,6) 3.28 -- equal MFMA counts per SIMD (18,9) is not equal time: a dW2 step carries more VALU
    // and LDS latency per MFMA than a da1 group
    constexpr int KL = 16, KH = 9;
    const int wq = wave & 3, kl = hy == 0 ? KL : KH, kb = (36 - kl) / 3, kr = (36 - kl) % 3;
    const int s_first = wq < 3 ? wq * kb + (wq < kr ? wq : kr) : 36 - kl;
    const int s_last = wq < 3 ? s_first + kb + (wq < kr ? 1 : 0) : 36;
#pragma nounroll
    for (int s = s_first; s < s_last; s += s_step) {
      const int T = 4 * s + q, ty = T / 12, tx = T - 12 * ty;
      const float* p = a1_s + j * CHB + (2 * ty + hy) * ROWB + 2 * tx;
      const float* pz = d_s + dofs(j) + (ty + 1) * TRD + 2 * tx + 2;
      const float2 y0 = *reinterpret_cast<const float2*>(pz);
      const float2 y1 = *reinterpret_cast<const float2*>(pz + ROD);
      f32x2 dl[3], dh[3];                                        // patch rows hy..hy+2 as two column pairs
#pragma unroll
      for (int rr = 0; rr < 3; ++rr) {
        dl[rr] = *reinterpret_cast<const f32x2*>(p + rr * ROWB);
        dh[rr] = *reinterpret_cast<const f32x2*>(p + rr * ROWB + 2);
      }
      f32x2 tal, tah, tbl, tbh;                                  // the two rows of B^T d of this half
      float za[2], zb[2];
      if (hy == 0) {
        tal = dl[0] - dl[2]; tah = dh[0] - dh[2]; tbl = dl[1] + dl[2]; tbh = dh[1] + dh[2];
        za[0] = y0.x; za[1] = y0.y; zb[0] = y0.x + y1.x; zb[1] = y0.y + y1.y;
      } else {
        tal = dl[1] - dl[0]; tah = dh[1] - dh[0]; tbl = dl[0] - dl[2]; tbh = dh[0] - dh[2];
        za[0] = y0.x - y1.x; za[1] = y0.y - y1.y; zb[0] = -y1.x; zb[1] = -y1.y;
      }
      // all operands first, then 8 back-to-back MFMAs: VALU and MFMA of ONE wavefront do not overlap
      // (tools/mfma_feed.hip); a VALU op in front of every MFMA would stall the pipe for both wavefronts
      float vv[8], zz[8];
      {
        const f32x2 a03 = tal - tah, a12 = pk_v12(tal, tah), b03 = tbl - tbh, b12 = pk_v12(tbl, tbh);
        vv[0] = a03.x; vv[1] = a12.x; vv[2] = a12.y; vv[3] = a03.y;
        vv[4] = b03.x; vv[5] = b12.x; vv[6] = b12.y; vv[7] = b03.y;
      }
      zz[0] = za[0]; zz[1] = za[0] + za[1]; zz[2] = za[0] - za[1]; zz[3] = -za[1];
      zz[4] = zb[0]; zz[5] = zb[0] + zb[1]; zz[6] = zb[0] - zb[1]; zz[7] = -zb[1];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int xi = 0; xi < 8; ++xi) dU[xi] = mfma(zz[xi], vv[xi], dU[xi]);
      __builtin_amdgcn_sched_barrier(0);
    }
    TSTAMP(2);
    // ---- P4: dpre1 = conv2^T(dY2) * gate on 2x2 tiles: per 4 input channels (g) 16 operands, then 16 MFMAs
#pragma nounroll
    for (int k = 0; k < 2; ++k) {
      const int grp = wave + NW * k;                            // wave-uniform
      if (grp < NG4) {
        const int t = 16 * grp + j;
        const bool ok = t < 169;
        const int tc = ok ? t : 0, ty = tc / 13, tx = tc - 13 * ty;
        const float* pd = d_s + ty * TRD + 2 * tx;
        f32x4 m[16];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) m[xi] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int RO4[4] = {0, ROD, TRD, TRD + ROD};           // patch row a at (a>>1)*TRD + (a&1)*ROD
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float* pp = pd + dofs(4 * g + q);
          f32x2 plo[4], phi[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            plo[rr] = *reinterpret_cast<const f32x2*>(pp + RO4[rr]);
            phi[rr] = *reinterpret_cast<const f32x2*>(pp + RO4[rr] + 2);
          }
          f32x4 uf[4];
#pragma unroll
          for (int xy = 0; xy < 4; ++xy) uf[xy] = *reinterpret_cast<const f32x4*>(u_s + ((g * 4 + xy) * 64 + lane) * 4);
          float vv[16];
          wino_in(plo, phi, vv);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int xi = 0; xi < 16; ++xi) m[xi] = mfma(uf[xi >> 2][xi & 3], vv[xi], m[xi]);
          __builtin_amdgcn_sched_barrier(0);
        }
        TSTAMP(3);                                              // (timing build) P4a: the 64 Winograd MFMAs + transforms
        f32x4 dp[4];                                            // dpre1 at sub-position p, channels 4q+r
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
          const float* pa = a1_s + (4 * q + r) * CHB + 2 * ty * ROWB + 2 * tx;
          const float2 g0 = *reinterpret_cast<const float2*>(pa), g1 = *reinterpret_cast<const float2*>(pa + ROWB);
          dp[0][r] = (ok && g0.x > 0.f) ? y00 : 0.f;
          dp[1][r] = (ok && g0.y > 0.f) ? y01 : 0.f;
          dp[2][r] = (ok && g1.x > 0.f) ? y10 : 0.f;
          dp[3][r] = (ok && g1.y > 0.f) ? y11 : 0.f;
        }
        TSTAMP(5);                                              // P4b: output transform + gate
        // dW1 / db1 partials against the 4x4 image patch of this tile
        float ep[4][4];
        {
          const float* pe = e_rd + 2 * ty * ROWEB + 2 * tx;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const float2 lo = *reinterpret_cast<const float2*>(pe + rr * ROWEB);
            const float2 hi = *reinterpret_cast<const float2*>(pe + rr * ROWEB + 2);
            ep[rr][0] = lo.x; ep[rr][1] = lo.y; ep[rr][2] = hi.x; ep[rr][3] = hi.y;
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {                           // packed over the channel pair (2 flops per lane per op)
          f32x2 d2[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) d2[p] = f32x2{dp[p][2 * h], dp[p][2 * h + 1]};
          gW1p[h][9] += (d2[0] + d2[1]) + (d2[2] + d2[3]);
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            f32x2 acc = gW1p[h][tap];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const float ev = ep[(p >> 1) + ky][(p & 1) + kx];
              acc = __builtin_elementwise_fma(d2[p], f32x2{ev, ev}, acc);
            }
            gW1p[h][tap] = acc;
          }
        }
        TSTAMP(6);                                              // P4c: dW1 / db1 partials
        // T[tap][pos] = sum_oc W1[oc][tap] dpre1[oc][pos]: dpre1 in the C/D layout IS the B operand
        f32x4 tq[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) tq[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int p = 0; p < 4; ++p) tq[p] = mfma(w1t[r], dp[p][r], tq[p]);
        // plane of tap 4q+r, or the dump plane 9 (taps >= 9, lanes without a tile): one base per r, the four
        // sub-positions are immediate offsets of the stores
        {
          const int pos0 = 2 * ty * C1 + 2 * tx;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ok && 4 * q + r < 9) {                           // taps 9..15 of the MFMA tile and idle lanes: nothing to store
              float* tp = T_s + (4 * q + r) * CS + pos0;
              tp[0] = tq[0][r]; tp[1] = tq[1][r]; tp[C1] = tq[2][r]; tp[C1 + 1] = tq[3][r];
            }
        }
      }
    }
    TSTAMP(7);                                                  // P4d: T planes
    stage_e(smem + (par ^ 1) * ESZB);                           // the next image (zeros behind the last one)
  }
  __syncthreads();
  if ((int64_t)blockIdx.x < a.n) de_gather(blockIdx.x + (a.n - 1 - blockIdx.x) / gridDim.x * gridDim.x);   // the last image
#ifdef GNF_CNN_TIMING
  if (blockIdx.x == 7 && (tid & 63) == 0)
    for (int k = 0; k < 8; ++k) a.part[((int64_t)gridDim.x * NW + 1) * PROW + wave * 8 + k] = (float)tacc[k];
#endif

  // ---- per-wave partial row: dW2 [16][144] | dW1+db1 [16][16] | db2 [16] -- into LDS (the images are done), the eight
  //      rows of the workgroup are then added in wavefront order: 256 partial rows for the row-sum launch, not 2048
  __syncthreads();                                     // the last de gather has read its T planes
  float* prow = smem + wave * PROW;
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

constexpr size_t kBwdWinoLds = (size_t)(2 * ESZB + NCH * CHB + DSZW + 9 * CS + USZ) * sizeof(float);
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
  if (((!e || !g_pooled || !argmax || !ge) && n_img > 0) || !W1 || !b1 || !W2 || !gW1 || !gb1 || !gW2 || !gb2 || !ws ||
      n_img < 0)
    return GNF_EINVAL;                       // empty batch: zero weight gradients through the same kernels
  if (ws_bytes < gnf_mnistcnn_conv_bwd_ws_bytes(n_img)) return GNF_EWS;
  CnnArgs a{};
  a.e = e; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.gp = g_pooled; a.argin = argmax; a.ge = ge; a.part = (float*)ws;
  a.n = n_img;
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
