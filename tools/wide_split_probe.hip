// Sizing probe (round 6): the hidden->hidden layers of the WIDE Monotonic forward kernel (mono_fwd_wide_k, H = 160 padded,
// a batch of 64 (element, node) pairs in LDS, weights streamed from L2 as ready-made MFMA fragments, 4 wavefronts (mh, nh),
// two workgroups per CU) in two forms with the SAME structure:
//   f32  : v_mfma_f32_16x16x4_f32 on fp32 activations [64][164] (what the product runs)
//   split: v_mfma_f32_16x16x32_bf16 on exact 3 x bf16 splits -- activations as three bf16 planes [3][64][176] written by the
//          producing layer's epilogue (ReLU + split + three ds_write_b64 per tile), weights pre-split into fragment-major planes,
//          six cross terms per product, fp32 accumulate in two classes (as gnf_gemm_split.hip)
// Timed over the whole launch (512 persistent workgroups, `iters` batches of 3 layers each); one workgroup's final activations
// are compared between the two forms and against an fp64 host chain.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wide_split_probe tools/wide_split_probe.hip && /tmp/wide_split_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int HT = 10, HP = 160, NL = 3, NP = 64, MT = 5;
constexpr int PF = HP + 4;            // fp32 row pitch (floats)
constexpr int PB = 352;               // bf16 plane row pitch (bytes): 352 / 16 = 22 = 2 (mod 4): conflict-free b128 fragment reads
constexpr int KT32 = HP / 32, KT16 = HP / 16;

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
  const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = cvt_pk_bf16(q0, q1);
}
__device__ __forceinline__ float relu1(float x) { float y; asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x)); return y; }

// ------------------------------------------------------------------------------------------------------------------- fp32 form
// wf: [NL][HT tiles][KT16][64 lanes] f32x4: lane (j, q) of fragment (m, t) = W[16 m + j][16 t + 4 q .. + 3]
__global__ __launch_bounds__(256, 2) void chain_f32_k(const f32x4* __restrict__ wf, const float* __restrict__ x0, float* __restrict__ out,
                                                      int iters) {
  extern __shared__ __attribute__((aligned(16))) float act[];          // [64][PF]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15, mh = wave & 1, nh = wave >> 1, m0 = MT * mh;
  const int prow = (16 * nh + j) * PF;
  f32x4 acc[MT][2];
  for (int it = 0; it < iters; ++it) {
    // "layer 0": the batch's first activations, computed by the lane that owns them (as the product's layer0 + store_act)
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x0 + (sl * 32 + 16 * nh + j) * HP + 16 * (m0 + mi) + 4 * q);
        *reinterpret_cast<f32x4*>(act + prow + sl * 32 * PF + 16 * (m0 + mi) + 4 * q) = v;
      }
    __syncthreads();
    for (int l = 0; l < NL; ++l) {
      const f32x4* w = wf + ((size_t)l * HT + m0) * KT16 * 64 + lane;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) acc[mi][0] = acc[mi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 A[2][MT], B[2][2];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) A[0][mi] = w[(mi * KT16) * 64];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) B[0][sl] = *reinterpret_cast<const f32x4*>(act + prow + sl * 32 * PF + 4 * q);
#pragma unroll
      for (int t = 0; t < KT16; ++t) {
        if (t + 1 < KT16) {
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) A[(t + 1) & 1][mi] = w[(mi * KT16 + t + 1) * 64];
#pragma unroll
          for (int sl = 0; sl < 2; ++sl) B[(t + 1) & 1][sl] = *reinterpret_cast<const f32x4*>(act + prow + sl * 32 * PF + 16 * (t + 1) + 4 * q);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
              acc[mi][sl] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[t & 1][mi][r], B[t & 1][sl][r], acc[mi][sl], 0, 0, 0);
      }
      __syncthreads();
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = relu1(acc[mi][sl][r]);
          *reinterpret_cast<f32x4*>(act + prow + sl * 32 * PF + 16 * (m0 + mi) + 4 * q) = v;
        }
      __syncthreads();
    }
  }
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < NP * HP; i += 256) out[i] = act[(i / HP) * PF + i % HP];
}

// ------------------------------------------------------------------------------------------------------------------ split form
// wp: [NL][3 planes][HT tiles][KT32][64 lanes] u32x4: lane (j, q) of fragment (p, m, t) = plane p of W[16 m + j][32 t + 8 q .. + 7]
template <int CLASSES, int PIPE>
__global__ __launch_bounds__(256, 2) void chain_split_k(const u32x4* __restrict__ wp, const float* __restrict__ x0, float* __restrict__ out,
                                                        int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char planes[];   // [3][64][PB]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, j = lane & 15, mh = wave & 1, nh = wave >> 1, m0 = MT * mh;
  constexpr int PLANE = NP * PB;
  const int prow = (16 * nh + j) * PB;
  f32x4 big[MT][2], sml[MT][2];
  float keep[MT][2][4];
  auto store_split = [&](int mi, int sl, const float (&v)[4]) {
    unsigned h0, m0_, l0, h1, m1, l1;
    split3_pair(v[0], v[1], h0, m0_, l0);
    split3_pair(v[2], v[3], h1, m1, l1);
    unsigned char* dst = planes + prow + sl * 32 * PB + 2 * (16 * (m0 + mi) + 4 * q);
    *reinterpret_cast<u32x2*>(dst) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(dst + PLANE) = u32x2{m0_, m1};
    *reinterpret_cast<u32x2*>(dst + 2 * PLANE) = u32x2{l0, l1};
  };
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(x0 + (sl * 32 + 16 * nh + j) * HP + 16 * (m0 + mi) + 4 * q);
        const float v[4] = {v4[0], v4[1], v4[2], v4[3]};
        store_split(mi, sl, v);
      }
    __syncthreads();
    for (int l = 0; l < NL; ++l) {
      const u32x4* w = wp + ((size_t)l * 3 * HT + m0) * KT32 * 64 + lane;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) { big[mi][sl] = f32x4{0.f, 0.f, 0.f, 0.f}; if (CLASSES > 1) sml[mi][sl] = big[mi][sl]; }
      auto loadA = [&](int t, u32x4 (&A)[3][MT]) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi) A[p][mi] = w[((p * HT + mi) * KT32 + t) * 64];
      };
      auto loadB = [&](int t, u32x4 (&B)[3][2]) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int sl = 0; sl < 2; ++sl)
            B[p][sl] = *reinterpret_cast<const u32x4*>(planes + p * PLANE + prow + sl * 32 * PB + 2 * (32 * t + 8 * q));
      };
      auto mm = [&](const u32x4& a, const u32x4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
      };
      if constexpr (PIPE) {
        u32x4 A[2][3][MT], B[2][3][2];
        loadA(0, A[0]); loadB(0, B[0]);
#pragma unroll
        for (int t = 0; t < KT32; ++t) {
          // products in three groups of two; the next k-tile's fragments are requested behind the first group
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
              for (int sl = 0; sl < 2; ++sl) {
                f32x4& cs = (CLASSES > 1 ? sml : big)[mi][sl];
                if (g == 0) { mm(A[t & 1][2][mi], B[t & 1][0][sl], cs); mm(A[t & 1][0][mi], B[t & 1][2][sl], cs); }
                if (g == 1) { mm(A[t & 1][1][mi], B[t & 1][1][sl], cs); mm(A[t & 1][1][mi], B[t & 1][0][sl], cs); }
                if (g == 2) { mm(A[t & 1][0][mi], B[t & 1][1][sl], cs); mm(A[t & 1][0][mi], B[t & 1][0][sl], big[mi][sl]); }
              }
            __builtin_amdgcn_sched_barrier(0);
            if (g == 0 && t + 1 < KT32) loadA(t + 1, A[(t + 1) & 1]);
            if (g == 1 && t + 1 < KT32) loadB(t + 1, B[(t + 1) & 1]);
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < KT32; ++t) {
          u32x4 A[3][MT], B[3][2];
          loadA(t, A); loadB(t, B);
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
              f32x4& cs = (CLASSES > 1 ? sml : big)[mi][sl];
              mm(A[2][mi], B[0][sl], cs); mm(A[0][mi], B[2][sl], cs); mm(A[1][mi], B[1][sl], cs); mm(A[1][mi], B[0][sl], cs);
              mm(A[0][mi], B[1][sl], cs); mm(A[0][mi], B[0][sl], big[mi][sl]);
            }
        }
      }
      __syncthreads();
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = keep[mi][sl][r] = relu1(CLASSES > 1 ? big[mi][sl][r] + sml[mi][sl][r] : big[mi][sl][r]);
          store_split(mi, sl, v);
        }
      __syncthreads();
    }
  }
  if (blockIdx.x == 0) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(sl * 32 + 16 * nh + j) * HP + 16 * (m0 + mi) + 4 * q + r] = keep[mi][sl][r];
  }
}

static unsigned short bf16_rne(float x) {
  unsigned u; memcpy(&u, &x, 4);
  const unsigned r = u + 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(r >> 16);
}
static float bf16_f(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
  srand(5);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  std::vector<float> W((size_t)NL * HP * HP), X((size_t)NP * HP);
  for (auto& w : W) w = rnd() * 0.19f;                 // ~ sqrt(6 / 160): activations keep their scale through the ReLUs
  for (auto& x : X) x = fabsf(rnd());
  std::vector<f32x4> wf((size_t)NL * HT * KT16 * 64);
  std::vector<u32x4> wp((size_t)NL * 3 * HT * KT32 * 64);
  for (int l = 0; l < NL; ++l)
    for (int m = 0; m < HT; ++m) {
      for (int t = 0; t < KT16; ++t)
        for (int ln = 0; ln < 64; ++ln)
          for (int r = 0; r < 4; ++r) wf[(((size_t)l * HT + m) * KT16 + t) * 64 + ln][r] = W[((size_t)l * HP + 16 * m + (ln & 15)) * HP + 16 * t + 4 * (ln >> 4) + r];
      for (int t = 0; t < KT32; ++t)
        for (int ln = 0; ln < 64; ++ln) {
          unsigned short pl[3][8];
          for (int i = 0; i < 8; ++i) {
            float x = W[((size_t)l * HP + 16 * m + (ln & 15)) * HP + 32 * t + 8 * (ln >> 4) + i];
            for (int p = 0; p < 3; ++p) { pl[p][i] = bf16_rne(x); x -= bf16_f(pl[p][i]); }
          }
          for (int p = 0; p < 3; ++p)
            for (int i = 0; i < 4; ++i)
              wp[((((size_t)l * 3 + p) * HT + m) * KT32 + t) * 64 + ln][i] = (unsigned)pl[p][2 * i] | ((unsigned)pl[p][2 * i + 1] << 16);
        }
    }
  // fp64 chain of ONE batch pass (iters = 1)
  std::vector<double> a(X.begin(), X.end()), b((size_t)NP * HP);
  for (int l = 0; l < NL; ++l) {
    for (int p = 0; p < NP; ++p)
      for (int u = 0; u < HP; ++u) {
        double s = 0.;
        for (int k = 0; k < HP; ++k) s += (double)W[((size_t)l * HP + u) * HP + k] * a[(size_t)p * HP + k];
        b[(size_t)p * HP + u] = s > 0. ? s : 0.;
      }
    a = b;
  }
  f32x4* dwf; u32x4* dwp; float *dx, *dout;
  CK(hipMalloc(&dwf, wf.size() * 16)); CK(hipMalloc(&dwp, wp.size() * 16)); CK(hipMalloc(&dx, X.size() * 4)); CK(hipMalloc(&dout, X.size() * 4));
  CK(hipMemcpy(dwf, wf.data(), wf.size() * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(dwp, wp.data(), wp.size() * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  const int lds_f = NP * PF * 4, lds_s = 3 * NP * PB;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_f32_k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_f));
  typedef void (*split_fn)(const u32x4*, const float*, float*, int);
  const split_fn fns[4] = {chain_split_k<2, 0>, chain_split_k<1, 0>, chain_split_k<2, 1>, chain_split_k<1, 1>};
  const char* names[4] = {"split 2 classes        ", "split 1 class          ", "split 2 classes, piped ", "split 1 class, piped   "};
  for (auto f : fns) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, lds_s));
  std::vector<float> o(X.size());
  auto err = [&](const char* tag) {
    CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
    double num = 0., den = 0., mx = 0.;
    for (size_t i = 0; i < o.size(); ++i) { const double d = o[i] - a[i]; num += d * d; den += a[i] * a[i]; mx = fmax(mx, fabs(d)); }
    printf("%s: one batch pass against the fp64 chain: rel. L2 error %.2e, max abs %.2e (|a| rms %.3f)\n", tag, sqrt(num / den), mx, sqrt(den / o.size()));
  };
  hipLaunchKernelGGL(chain_f32_k, dim3(1), dim3(256), lds_f, 0, dwf, dx, dout, 1); CK(hipDeviceSynchronize()); err("f32 MFMA               ");
  for (int v = 0; v < 4; ++v) {
    CK(hipMemset(dout, 0, X.size() * 4));
    hipLaunchKernelGGL(fns[v], dim3(1), dim3(256), lds_s, 0, dwp, dx, dout, 1); CK(hipDeviceSynchronize()); err(names[v]);
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000, grid = 512;
  const double flop = 2. * grid * (double)iters * NL * NP * HP * HP;
  for (int rep = 0; rep < 2; ++rep)
    for (int form = 0; form < 5; ++form) {
      CK(hipEventRecord(e0));
      if (form == 0) hipLaunchKernelGGL(chain_f32_k, dim3(grid), dim3(256), lds_f, 0, dwf, dx, dout, iters);
      else hipLaunchKernelGGL(fns[form - 1], dim3(grid), dim3(256), lds_s, 0, dwp, dx, dout, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s: %.3f ms for %d batches x %d layers on %d workgroups: %.2f us per (batch, layer) of a workgroup, %.1f TFLOP/s fp32-equivalent\n",
             form ? names[form - 1] : "f32 MFMA               ", ms, iters, NL, grid, ms * 1e3 / (iters * NL), flop / (ms * 1e-3) / 1e12);
    }
  return 0;
}
