// Convolutional front of the MNISTCNN embedding net (models/MLP.py:36-41) over the B*d masked
// images of the DAG conditioner: conv3x3(1->16) + ReLU + conv3x3(16->16) + maxpool2, forward
// and backward, as implicit GEMMs on v_mfma_f32_16x16x4_f32 (exact fp32).  This is where
// the MNIST d=784 Monotonic+DAG step spends its flops (1.42 MMAC per image, 78 400 images
// per 100 samples); the fc layers behind it run on the GEMM of gnf_gemm.hip.
//
// One workgroup processes one 28x28 image at a time, entirely out of LDS:
//   M = 16 output channels (all of them), N = 16 output positions (a 2-row x 8-column patch,
//   so 2x2 pool windows never straddle tiles), K = taps / (input channel, tap) pairs.
//   The weights are the A operand and live in registers for the whole kernel; the B operand
//   is gathered from the LDS image with per-K-step immediate offsets.  Row / channel strides
//   (40 and == 16 mod 32 dwords) make the 64-lane gather bank-conflict free:
//   lane (q, j) -> channel 4g+q (bank +16q), row j>>3 (bank +8), column j&7.
// Backward recomputes conv1 (3 % of the flops) instead of storing 43 KB of activations per
// image, takes the pool argmax saved by the forward (1 byte per pooled value), and keeps the
// weight-gradient accumulators in registers across all images of a workgroup.
#include "gnf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int IMG = 28, C1 = 26, C2 = 24, PO = 12, NCH = 16;
constexpr int ROWE = 36, ESZ = IMG * ROWE;            // padded input image
constexpr int ROW = 40, CH = C1 * ROW;                // conv1 activations: [16][26][40], CH = 1040 == 16 mod 32
constexpr int ROWD = 40, CHD = 28 * ROWD + 16;        // dY2 with a 2-wide zero border: [16][28][40], CHD = 1136 == 16 mod 32
constexpr int CS = 688;                               // per-tap planes T: [9][688] flat 26x26 positions
constexpr int NPOOL = NCH * PO * PO;                  // 2304
constexpr int FWD_WAVES = 8, BWD_WAVES = 8;
constexpr int PROW = NCH * 144 + NCH * 16 + NCH;      // per-wave gradient partial row: dW2 | dW1+db1 | db2

static_assert(CH % 32 == 16 && CHD % 32 == 16, "channel strides must sit 16 banks apart");

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
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
};

// conv1 + ReLU of the image in e_s into a1_s; 43 tiles of 16 consecutive positions of the 26x26 grid
template <int NW>
__device__ __forceinline__ void conv1_tiles(const float* e_s, float* a1_s, const float (&w1f)[3], const int (&off1)[3],
                                            const f32x4& b1v, int wave, int q, int j) {
#pragma nounroll
  for (int t = wave; t < 43; t += NW) {
    const int pos = 16 * t + j;
    const int pc = pos < C1 * C1 ? pos : 0;
    const int y = pc / C1, x = pc - y * C1;
    const int base = y * ROWE + x;
    f32x4 acc = b1v;
#pragma unroll
    for (int s = 0; s < 3; ++s) acc = mfma(w1f[s], e_s[base + off1[s]], acc);
    if (pos < C1 * C1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a1_s[(4 * q + r) * CH + y * ROW + x] = fmaxf(acc[r], 0.f);
    }
  }
}

__global__ __launch_bounds__(64 * FWD_WAVES) void cnn_fwd_k(CnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* e_s = smem;
  float* a1_s = smem + ESZ;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, j = lane & 15;

  // weights as MFMA A operands (row i = j = output channel, K slot q), resident in registers
  float w1f[3];
  int off1[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = 4 * s + q;
    w1f[s] = tap < 9 ? a.W1[j * 9 + tap] : 0.f;
    const int tt = tap < 9 ? tap : 0;
    off1[s] = (tt / 3) * ROWE + tt % 3;
  }
  float w2f[36];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < 9; ++k) w2f[g * 9 + k] = a.W2[(j * NCH + 4 * g + q) * 9 + k];
  f32x4 b1v, b2v;
#pragma unroll
  for (int r = 0; r < 4; ++r) { b1v[r] = a.b1[4 * q + r]; b2v[r] = a.b2[4 * q + r]; }

  for (int i = tid; i < ESZ; i += blockDim.x) e_s[i] = 0.f;

  constexpr int NT = 64 * FWD_WAVES, EPT = (IMG * IMG + NT - 1) / NT;   // image elements per thread
  float pre[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int i = tid + k * NT;
    pre[k] = (blockIdx.x < a.n && i < IMG * IMG) ? a.e[(int64_t)blockIdx.x * (IMG * IMG) + i] : 0.f;
  }
  for (int64_t img = blockIdx.x; img < a.n; img += gridDim.x) {
    __syncthreads();                                   // previous image fully consumed
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int i = tid + k * NT;
      if (i < IMG * IMG) e_s[(i / IMG) * ROWE + i % IMG] = pre[k];
    }
    __syncthreads();
    {                                                  // next image's pixels: in flight under the MFMAs
      const int64_t nx = img + gridDim.x;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int i = tid + k * NT;
        pre[k] = (nx < a.n && i < IMG * IMG) ? a.e[nx * (IMG * IMG) + i] : 0.f;
      }
    }
    conv1_tiles<FWD_WAVES>(e_s, a1_s, w1f, off1, b1v, wave, q, j);
    __syncthreads();
    // conv2 (implicit GEMM, K = 16 channels x 9 taps) + 2x2 max pool: 36 tiles, two in flight per wave
#pragma nounroll
    for (int t = wave; t < 36; t += 2 * FWD_WAVES) {
      const int tB = t + FWD_WAVES < 36 ? t + FWD_WAVES : t;
      const int yA = 2 * (t / 3) + (j >> 3), xA = 8 * (t % 3) + (j & 7);
      const int yB = 2 * (tB / 3) + (j >> 3), xB = 8 * (tB % 3) + (j & 7);
      const float* pA = a1_s + q * CH + yA * ROW + xA;
      const float* pB = a1_s + q * CH + yB * ROW + xB;
      f32x4 accA = b2v, accB = b2v;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int o = 4 * g * CH + ky * ROW + kx;
            accA = mfma(w2f[g * 9 + ky * 3 + kx], pA[o], accA);
            accB = mfma(w2f[g * 9 + ky * 3 + kx], pB[o], accB);
          }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const f32x4 acc = half ? accB : accA;
        const int tt = half ? tB : t;
        if (half && tB == t) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v00 = acc[r];
          const float v01 = dpp_xor1(v00), v10 = dpp_xor8(v00), v11 = dpp_xor8(v01);
          if ((j & 9) == 0) {                          // top-left lane of a 2x2 window; first max wins ties
            float best = v00; int bi = 0;
            if (v01 > best) { best = v01; bi = 1; }
            if (v10 > best) { best = v10; bi = 2; }
            if (v11 > best) { best = v11; bi = 3; }
            const int64_t o = img * NPOOL + (4 * q + r) * (PO * PO) + (tt / 3) * PO + 4 * (tt % 3) + ((j & 7) >> 1);
            a.pooled[o] = best;
            a.arg[o] = (unsigned char)bi;
          }
        }
      }
    }
  }
}

__global__ __launch_bounds__(64 * BWD_WAVES) void cnn_bwd_k(CnnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* e_s = smem;
  float* a1_s = smem + ESZ;                 // conv1 activations, overwritten in place by dpre1
  float* d_s = a1_s + NCH * CH;             // dY2 with zero border; later reused for the per-tap planes T
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, j = lane & 15;
  constexpr int NW = BWD_WAVES, NT = 64 * BWD_WAVES;

  float w1f[3];
  int off1[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int tap = 4 * s + q;
    w1f[s] = tap < 9 ? a.W1[j * 9 + tap] : 0.f;
    const int tt = tap < 9 ? tap : 0;
    off1[s] = (tt / 3) * ROWE + tt % 3;
  }
  f32x4 b1v;
#pragma unroll
  for (int r = 0; r < 4; ++r) b1v[r] = a.b1[4 * q + r];
  // W2^T as A operand of the data gradient: row i = j = input channel, K slot q -> output channel 4g+q
  float w2t[36];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int k = 0; k < 9; ++k) w2t[g * 9 + k] = a.W2[((4 * g + q) * NCH + j) * 9 + k];
  // W1^T as A operand of the per-tap planes: row i = j = tap, K slot q -> channel 4s+q
  float w1t[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) w1t[s] = j < 9 ? a.W1[(4 * s + q) * 9 + j] : 0.f;
  // per-lane column offsets of the im2col B operand of dW2: column c = 16 nt + j = ic*9 + ky*3 + kx
  int colo[9];
#pragma unroll
  for (int nt = 0; nt < 9; ++nt) {
    const int c = 16 * nt + j;
    colo[nt] = (c / 9) * CH + ((c % 9) / 3) * ROW + c % 3;
  }
  const int tapo = j < 9 ? (j / 3) * ROWE + j % 3 : 0;       // B operand of dW1: e patch for tap j

  f32x4 gW2[9], gW1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < 9; ++nt) gW2[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float gb2 = 0.f;                                           // thread tid accumulates channel tid/32

  for (int i = tid; i < ESZ; i += NT) e_s[i] = 0.f;
  for (int i = tid; i < NCH * CHD; i += NT) d_s[i] = 0.f;

  // software prefetch: the next image's pixels, pooled gradients and argmax bytes are loaded into
  // registers while the MFMA phases of the current image run
  constexpr int EPT = (IMG * IMG + NT - 1) / NT, WPT = (PO * PO + 31) / 32;
  float epre[EPT], gpre[WPT];
  int apre[WPT];
  auto prefetch = [&](int64_t im) {
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
      apre[k] = ok ? (int)a.argin[o] : 0;
    }
  };
  prefetch(blockIdx.x);

  for (int64_t img = blockIdx.x; img < a.n; img += gridDim.x) {
    __syncthreads();
    // ---- P0: image, and dY2 = pool-backward scatter of g_pooled (one write per conv2 position)
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int i = tid + k * NT;
      if (i < IMG * IMG) e_s[(i / IMG) * ROWE + i % IMG] = epre[k];
    }
    {
      const int oc = tid >> 5;                                // 32 threads per channel
#pragma unroll
      for (int k = 0; k < WPT; ++k) {
        const int w = (tid & 31) + 32 * k;
        if (w < PO * PO) {
          const float g = gpre[k];
          const int am = apre[k];
          gb2 += g;
          float* p = d_s + oc * CHD + (2 * (w / PO) + 2) * ROWD + 2 * (w % PO) + 2;
          p[0] = am == 0 ? g : 0.f;
          p[1] = am == 1 ? g : 0.f;
          p[ROWD] = am == 2 ? g : 0.f;
          p[ROWD + 1] = am == 3 ? g : 0.f;
        }
      }
    }
    __syncthreads();
    prefetch(img + gridDim.x);
    // ---- P1: recompute conv1 + ReLU
    conv1_tiles<NW>(e_s, a1_s, w1f, off1, b1v, wave, q, j);
    __syncthreads();
    // ---- P3: dW2[oc][c] += sum_pos dY2[oc][pos] * a1[ic(c)][pos + tap(c)];  K = 576 positions, 72 per wave
#pragma nounroll
    for (int s = 0; s < 18; ++s) {
      const int pos = 72 * wave + 4 * s + q;
      const int y = pos / C2, x = pos - y * C2;
      const float av = d_s[j * CHD + (y + 2) * ROWD + x + 2];
      const float* bp = a1_s + y * ROW + x;
#pragma unroll
      for (int nt = 0; nt < 9; ++nt) gW2[nt] = mfma(av, bp[colo[nt]], gW2[nt]);
    }
    __syncthreads();                                           // a1 as im2col operand is done
    // ---- P4: dpre1 = conv2^T(dY2) * (a1 > 0) on the 26x26 grid, 43 tiles of 16 consecutive positions (two in
    //      flight), written IN PLACE over a1: element (ic,y,x) is read (ReLU gate) and overwritten by one lane only
#pragma nounroll
    for (int t = wave; t < 43; t += 2 * NW) {
      const int tB = t + NW < 43 ? t + NW : t;
      const int posA = 16 * t + j, posB = 16 * tB + j;
      const int pcA = posA < C1 * C1 ? posA : 0, pcB = posB < C1 * C1 ? posB : 0;
      const int yA = pcA / C1, xA = pcA - yA * C1, yB = pcB / C1, xB = pcB - yB * C1;
      const float* pA = d_s + q * CHD + (yA + 2) * ROWD + xA + 2;
      const float* pB = d_s + q * CHD + (yB + 2) * ROWD + xB + 2;
      f32x4 accA = f32x4{0.f, 0.f, 0.f, 0.f}, accB = accA;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int o = 4 * g * CHD - ky * ROWD - kx;
            accA = mfma(w2t[g * 9 + ky * 3 + kx], pA[o], accA);
            accB = mfma(w2t[g * 9 + ky * 3 + kx], pB[o], accB);
          }
      if (posA < C1 * C1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* pa = a1_s + (4 * q + r) * CH + yA * ROW + xA;
          *pa = *pa > 0.f ? accA[r] : 0.f;
        }
      }
      if (tB != t && posB < C1 * C1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* pa = a1_s + (4 * q + r) * CH + yB * ROW + xB;
          *pa = *pa > 0.f ? accB[r] : 0.f;
        }
      }
    }
    __syncthreads();                                           // dpre1 complete; dY2 no longer needed
    // ---- P5a: dW1[oc][tap] += sum_pos dpre1[oc][pos] * e[pos + tap]; column 9 = ones -> db1.  704 = 8 x 88 positions
    f32x4 gW1b = f32x4{0.f, 0.f, 0.f, 0.f};                       // second chain: hides the 40-cycle MFMA latency
#pragma nounroll
    for (int s = 0; s < 22; s += 2) {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int pos = 88 * wave + 4 * (s + h2) + q;
        const bool ok = pos < C1 * C1;
        const int pc = ok ? pos : 0;
        const int y = pc / C1, x = pc - y * C1;
        const float av = ok ? a1_s[j * CH + y * ROW + x] : 0.f;
        const float bv = j < 9 ? e_s[y * ROWE + x + tapo] : (j == 9 ? 1.f : 0.f);
        if (h2) gW1b = mfma(av, bv, gW1b); else gW1 = mfma(av, bv, gW1);
      }
    }
    gW1 += gW1b;
    // ---- P5b: T[tap][pos] = sum_oc W1[oc][tap] * dpre1[oc][pos]  (43 position tiles), into the dY2 region
    float* T_s = d_s;
#pragma nounroll
    for (int t = wave; t < 43; t += NW) {
      const int pos = 16 * t + j;
      const int pc = pos < C1 * C1 ? pos : 0;
      const int y = pc / C1, x = pc - y * C1;
      const float* bp = a1_s + q * CH + y * ROW + x;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = mfma(w1t[s], bp[4 * s * CH], acc);
      if (pos < C1 * C1) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * q + r < 9) T_s[(4 * q + r) * CS + pos] = acc[r];
      }
    }
    __syncthreads();
    // de[y][x] = sum_tap T[tap][y-ky][x-kx]
    for (int i = tid; i < IMG * IMG; i += NT) {
      const int y = i / IMG, x = i - y * IMG;
      float s = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yy = y - ky, xx = x - kx;
          if (yy >= 0 && yy < C1 && xx >= 0 && xx < C1) s += T_s[(ky * 3 + kx) * CS + yy * C1 + xx];
        }
      a.ge[img * (IMG * IMG) + i] = s;
    }
    // restore the zero border of dY2 that T overwrote (the interior is rewritten by the next P0)
    __syncthreads();
    for (int i = tid; i < 9 * CS; i += NT) d_s[i] = 0.f;
  }

  // ---- per-wave partial row: dW2 [16][144] | dW1+db1 [16][16] | db2 [16]
  float* prow = a.part + ((int64_t)blockIdx.x * NW + wave) * PROW;
#pragma unroll
  for (int nt = 0; nt < 9; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[(4 * q + r) * 144 + 16 * nt + j] = gW2[nt][r];
#pragma unroll
  for (int r = 0; r < 4; ++r) prow[NCH * 144 + (4 * q + r) * 16 + j] = gW1[r];
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) gb2 += __shfl_xor(gb2, off, 64);    // over the 32 threads of a channel
  // wave w owns channels 2w, 2w+1 (lanes 0 and 32); the other 14 db2 slots of its row are zero
  if ((lane & 31) == 0) prow[NCH * 144 + NCH * 16 + 2 * wave + (lane >> 5)] = gb2;
  if (lane < NCH && (lane >> 1) != wave) prow[NCH * 144 + NCH * 16 + lane] = 0.f;
}

// unpack the summed partial row into the parameter-shaped gradients
__global__ void cnn_unpack_k(const float* __restrict__ vec, float* gW1, float* gb1, float* gW2, float* gb2) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= PROW) return;
  const float s = vec[n];
  if (n < NCH * 144) gW2[n] = s;
  else if (n < NCH * 144 + NCH * 16) {
    const int k = n - NCH * 144, oc = k >> 4, c = k & 15;
    if (c < 9) gW1[oc * 9 + c] = s;
    else if (c == 9) gb1[oc] = s;
  } else gb2[n - NCH * 144 - NCH * 16] = s;
}

constexpr size_t kFwdLds = (size_t)(ESZ + NCH * CH) * sizeof(float);
constexpr size_t kBwdLds = (size_t)(ESZ + NCH * CH + NCH * CHD) * sizeof(float);
constexpr unsigned kFwdGrid = 512, kBwdGrid = 256;

}  // namespace

extern "C" {

int gnf_mnistcnn_conv_fwd(const float* e, const float* W1, const float* b1, const float* W2, const float* b2,
                          float* pooled, unsigned char* argmax, int64_t n_img, gnf_stream_t stream) {
  if (!e || !W1 || !b1 || !W2 || !b2 || !pooled || !argmax || n_img < 0) return GNF_EINVAL;
  if (n_img == 0) return 0;
  CnnArgs a{};
  a.e = e; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.pooled = pooled; a.arg = argmax; a.n = n_img;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cnn_fwd_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kFwdLds);
  const unsigned grid = n_img < kFwdGrid ? (unsigned)n_img : kFwdGrid;
  hipLaunchKernelGGL(cnn_fwd_k, dim3(grid), dim3(64 * FWD_WAVES), kFwdLds, (hipStream_t)stream, a);
  GNF_LAUNCH_CHECK();
  return 0;
}

int64_t gnf_mnistcnn_conv_bwd_ws_bytes(int64_t n_img) {
  (void)n_img;
  return ((int64_t)kBwdGrid * BWD_WAVES + 1) * PROW * (int64_t)sizeof(float);
}

int gnf_mnistcnn_conv_bwd(const float* e, const float* W1, const float* b1, const float* W2, const float* g_pooled,
                          const unsigned char* argmax, float* ge, float* gW1, float* gb1, float* gW2, float* gb2,
                          void* ws, int64_t ws_bytes, int64_t n_img, gnf_stream_t stream) {
  if (!e || !W1 || !b1 || !W2 || !g_pooled || !argmax || !ge || !gW1 || !gb1 || !gW2 || !gb2 || !ws || n_img < 0)
    return GNF_EINVAL;
  if (ws_bytes < gnf_mnistcnn_conv_bwd_ws_bytes(n_img)) return GNF_EWS;
  CnnArgs a{};
  a.e = e; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.gp = g_pooled; a.argin = argmax; a.ge = ge; a.part = (float*)ws;
  a.n = n_img;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cnn_bwd_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kBwdLds);
  // fixed grid: every workgroup (also one without images) writes its partial rows
  hipLaunchKernelGGL(cnn_bwd_k, dim3(kBwdGrid), dim3(64 * BWD_WAVES), kBwdLds, (hipStream_t)stream, a);
  GNF_LAUNCH_CHECK();
  const int64_t rows = (int64_t)kBwdGrid * BWD_WAVES;
  float* vec = (float*)ws + rows * PROW;
  const int rc = gnf_rowsum_launch((const float*)ws, vec, rows, PROW, 0, (hipStream_t)stream);
  if (rc) return rc;
  hipLaunchKernelGGL(cnn_unpack_k, dim3((PROW + 255) / 256), dim3(256), 0, (hipStream_t)stream, vec, gW1, gb1, gW2,
                     gb2);
  GNF_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
