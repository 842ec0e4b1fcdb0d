// Residency census: how many workgroups of (threads, dynamic LDS, VGPR budget) does a gfx950 CU really hold?
// Each workgroup spins for a fixed wall-clock time; kernel time / spin time = number of rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int T, int W>
__global__ __launch_bounds__(T, W) void spin_k(long long ticks, float* out) {
  extern __shared__ float sm[];
  const long long t0 = wall_clock64();
  float acc = 0.f;
  while (wall_clock64() - t0 < ticks) acc += sm[threadIdx.x % 64];
  if (acc == 123.f) out[0] = acc;
}
template <int T, int W>
float run(int grid, size_t lds) {
  float* d; hipMalloc(&d, 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_k<T, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const long long ticks = 100000;   // 1 ms at 100 MHz
  hipLaunchKernelGGL((spin_k<T, W>), dim3(grid), dim3(T), lds, 0, ticks, d);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((spin_k<T, W>), dim3(grid), dim3(T), lds, 0, ticks, d);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  int nb = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin_k<T, W>, T, lds);
  printf("threads %4d minwaves/EU %d lds %6zu grid %5d : %.2f ms (rounds ~%.1f)  occupancy API %d\n", T, W, lds, grid, ms, ms / 1.0, nb);
  hipFree(d);
  return ms;
}
int main() {
  int v; hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0); printf("MaxSharedMemoryPerMultiprocessor %d\n", v);
  for (size_t lds : {16 * 1024, 60 * 1024, 70 * 1024, 80 * 1024, 100 * 1024}) {
    run<512, 1>(512, lds);
    run<256, 1>(512, lds);
    run<256, 1>(1024, lds);
  }
  run<512, 1>(1024, 16 * 1024);
  run<1024, 1>(512, 16 * 1024);
  return 0;
}
