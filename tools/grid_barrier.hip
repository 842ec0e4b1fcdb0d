// Cost of a grid barrier among 256 resident 1024-thread workgroups on MI355X, by variant (tools/grid_barrier.hip):
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/grid_barrier tools/grid_barrier.hip && gpurun_out/grid_barrier
// mode bit 0: agent-scope release fence before the arrival      bit 1: agent-scope acquire fence behind the wait
//      bit 2: two-level (8 groups) instead of flat               bit 3: s_sleep in the poll loop
//      bit 5: the data stores are agent-scope atomic stores (write-through)   bit 6: the data loads are agent-scope atomic loads
//      bit 4: every workgroup stores 1 KB per round before the barrier and reads another workgroup's 1 KB behind it (checked)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void bar_k(unsigned* bar, float* buf, int rounds, int mode, unsigned* bad) {
  const int nwg = gridDim.x, g = blockIdx.x & 7;
  const unsigned in_group = (unsigned)((nwg - g + 7) / 8), ngroups = nwg < 8 ? nwg : 8;
  unsigned errs = 0;
  for (int p = 0; p < rounds; ++p) {
    if (mode & 16) {
      float* q = buf + (size_t)(p & 1) * nwg * 256 + blockIdx.x * 256 + threadIdx.x;
      if (threadIdx.x < 256) {
        if (mode & 32) __hip_atomic_store(q, (float)(p * 1000 + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through (sc1)
        else *q = (float)(p * 1000 + blockIdx.x);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (mode & 1) __atomic_thread_fence(__ATOMIC_RELEASE); else __builtin_amdgcn_s_waitcnt(0);
      unsigned want;
      if (mode & 4) {
        const unsigned old = __hip_atomic_fetch_add(bar + 16 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == in_group * (unsigned)(p + 1) - 1u) __hip_atomic_fetch_add(bar + 16 * 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        want = ngroups * (unsigned)(p + 1);
      } else {
        __hip_atomic_fetch_add(bar + 16 * 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        want = (unsigned)nwg * (unsigned)(p + 1);
      }
      while (__hip_atomic_load(bar + 16 * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (mode & 8) __builtin_amdgcn_s_sleep(1);
      }
      if (mode & 2) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    if (mode & 16) {
      const int other = (blockIdx.x * 37 + 11 + p) % nwg;
      if (threadIdx.x < 256) {
        const float* q = buf + (size_t)(p & 1) * nwg * 256 + other * 256 + threadIdx.x;
        const float v = (mode & 64) ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;                 // sc1 load: past the local L2
        if (v != (float)(p * 1000 + other)) ++errs;
      }
    }
  }
  if (errs) atomicAdd(bad, errs);
}

int main() {
  unsigned *bar, *bad; float* buf;
  CK(hipMalloc(&bar, 4096)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&buf, 2 * 256 * 256 * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int modes[] = {0, 8, 4, 12, 1, 2, 3, 7, 15, 16, 16 + 4, 16 + 4 + 8, 16 + 7, 16 + 15, 16 + 3, 16 + 4 + 32 + 64, 16 + 4 + 32 + 2, 16 + 4 + 1 + 64};
  for (int nwg : {256, 64}) for (int mode : modes) {
    float best[2] = {1e9f, 1e9f};
    unsigned hb = 0;
    for (int rep = 0; rep < 5; ++rep) for (int k = 0; k < 2; ++k) {
      const int rounds = k ? 210 : 10;
      CK(hipMemset(bar, 0, 4096)); CK(hipMemset(bad, 0, 4));
      CK(hipEventRecord(a));
      hipLaunchKernelGGL(bar_k, dim3(nwg), dim3(1024), 0, 0, bar, buf, rounds, mode, bad);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (ms < best[k]) best[k] = ms;
      unsigned h; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost)); hb += h;
    }
    printf("nwg %3d mode %3d (%s%s%s%s%s%s%s): %.2f us per barrier   stale reads %u\n", nwg, mode, mode & 1 ? "rel " : "", mode & 2 ? "acq " : "",
           mode & 4 ? "2lvl " : "flat ", mode & 8 ? "sleep " : "", mode & 16 ? "data" : "", mode & 32 ? " st-sc1" : "", mode & 64 ? " ld-sc1" : "", (best[1] - best[0]) * 1e3f / 200.f, hb);
  }
  return 0;
}
