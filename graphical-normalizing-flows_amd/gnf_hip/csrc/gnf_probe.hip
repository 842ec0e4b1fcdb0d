// Device-ceiling probes used by bench.py to print MEASURED peaks beside the nominal ones
// ("use those as the denominator alongside the nominal figure").
// They are measurement aids, not part of the flow path.
#include "gnf_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// 8 independent accumulator chains of v_mfma_f32_16x16x4_f32 per wavefront, nothing else in the loop:
// the sustained rate of this kernel is the practical fp32 matrix ceiling of the chip at its running clock.
__global__ __launch_bounds__(512) void probe_mfma_k(float* out, int iters) {
  f32x4 acc[8];
  const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;   // keep the chain alive; practically never true
}

// STREAM copy: dst[i] = src[i].  Every workgroup owns ONE CONTIGUOUS chunk (DRAM page locality), 128-bit accesses,
// EIGHT independent loads in flight per lane per trip, non-temporal loads and stores (each byte is touched once),
// 16 workgroups per CU: the best of the 35 launch shapes of tools/copy_probe.hip on this part (5.6 TB/s; the first
// version -- grid-strided, one load in flight -- reached 4.6; MI355X_MICROARCH.md quotes 6.29 for its float4 copy).
__global__ __launch_bounds__(256) void probe_copy_k(f32x4* __restrict__ dst, const f32x4* __restrict__ src, int64_t n4) {
  const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = per * blockIdx.x, b1 = b0 + per < n4 ? b0 + per : n4;
  int64_t i = b0 + threadIdx.x;
  for (; i + 7 * 256 < b1; i += 8 * 256) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
    for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(v[u], dst + i + u * 256);
  }
  for (; i < b1; i += 256) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

__global__ void probe_empty_k() {}

}  // namespace

// an EMPTY grid of a given launch shape: the launch + ramp + drain floor a kernel of that shape cannot go below
extern "C" int gnf_probe_empty(int64_t grid, int block, gnf_stream_t stream) {
  if (grid < 1 || block < 1 || block > 1024) return GNF_EINVAL;
  hipLaunchKernelGGL(probe_empty_k, dim3((unsigned)grid), dim3(block), 0, (hipStream_t)stream);
  GNF_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t gnf_probe_mfma_f32(float* out, int iters, int blocks, gnf_stream_t stream) {
  if (!out || iters < 1 || blocks < 1) return GNF_EINVAL;
  hipLaunchKernelGGL(probe_mfma_k, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, out, iters);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return -(int64_t)e - 1000;
  // flops issued by this launch: blocks * 8 waves * iters * 32 MFMAs * (16*16*4*2)
  return (int64_t)blocks * 8 * (int64_t)iters * 32 * 2048;
}

extern "C" int gnf_probe_copy(float* dst, const float* src, int64_t n, gnf_stream_t stream) {
  if (!dst || !src || n < 0 || (n & 3)) return GNF_EINVAL;
  if (n == 0) return 0;
  hipLaunchKernelGGL(probe_copy_k, dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, (f32x4*)dst, (const f32x4*)src,
                     n / 4);
  GNF_LAUNCH_CHECK();
  return 0;
}
