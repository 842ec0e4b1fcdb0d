// STREAM-copy variants: which launch shape reaches the guide's 6.29 TB/s (float4 copy) on this box?
// unroll U in {1,2,4,8} independent 16-B loads per lane per trip, plain vs non-temporal, grid = CUs x {4,8,16,32,64}.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_k(f32x4* __restrict__ dst, const f32x4* __restrict__ src, long long n4) {
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
  }
  for (; i < n4; i += stride) dst[i] = src[i];
}
// contiguous-per-block variant: each block owns a contiguous chunk (better DRAM page locality)
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_chunk_k(f32x4* __restrict__ dst, const f32x4* __restrict__ src, long long n4) {
  const long long per = (n4 + gridDim.x - 1) / gridDim.x;
  const long long b0 = per * blockIdx.x, b1 = b0 + per < n4 ? b0 + per : n4;
  long long i = b0 + threadIdx.x;
  for (; i + (U - 1) * 256 < b1; i += U * 256) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * 256); else dst[i + u * 256] = v[u]; }
  }
  for (; i < b1; i += 256) dst[i] = src[i];
}
template <typename K>
void run(const char* name, K kern, int grid, f32x4* a, f32x4* b, long long n4) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a, b, n4);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, a, b, n4);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s grid %6d : %.0f GB/s\n", name, grid, 10.0 * 2 * 16 * n4 / (ms * 1e-3) / 1e9);
}
int main() {
  const long long n4 = 1LL << 26;   // 1 GiB per buffer
  f32x4 *a, *b; (void)hipMalloc(&a, n4 * 16); (void)hipMalloc(&b, n4 * 16);
  (void)hipMemset(b, 1, n4 * 16);
  for (int g : {1024, 2048, 4096, 8192, 16384}) {
    run("stride U=1 plain", copy_k<1, false>, g, a, b, n4);
    run("stride U=4 plain", copy_k<4, false>, g, a, b, n4);
    run("stride U=4 nt", copy_k<4, true>, g, a, b, n4);
    run("stride U=8 nt", copy_k<8, true>, g, a, b, n4);
    run("chunk  U=4 plain", copy_chunk_k<4, false>, g, a, b, n4);
    run("chunk  U=4 nt", copy_chunk_k<4, true>, g, a, b, n4);
    run("chunk  U=8 nt", copy_chunk_k<8, true>, g, a, b, n4);
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipMemcpyAsync(a, b, n4 * 16, hipMemcpyDeviceToDevice, 0);
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < 10; ++r) (void)hipMemcpyAsync(a, b, n4 * 16, hipMemcpyDeviceToDevice, 0);
  (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("hipMemcpyAsync D2D: %.0f GB/s\n", 10.0 * 2 * 16 * n4 / (ms * 1e-3) / 1e9);
  return 0;
}
