// Internal (not part of the C ABI): argument block + launcher of the fp32 MFMA GEMM,
// shared between gnf_gemm.hip and gnf_monotonic.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct GemmArgs {
  const float* A; int64_t sam, sak;
  const float* B; const float* Bmask; int64_t sbk, sbn;
  float* C; int64_t scm, scn;
  const float* bias;
  const float* Cmask; int64_t scmm, scmn;
  const float* gate; int64_t sgm, sgn;
  int flags; int64_t M, N, K;
  int64_t k_per_split, c_split_stride;   // split-K: blockIdx.z owns [z*kps, (z+1)*kps), writes C + z*stride
  // grouped launch (gnf_gemm_grouped_launch): blockIdx.z = group z owns rows [grp[2z], grp[2z] + grp[2z+1]) of A and C
  // and multiplies them with its own B = B + z * b_grp_stride; M is the largest group's row count; no split-K
  const int32_t* grp; int64_t b_grp_stride;
  // grp_k != 0: the group table partitions K instead (group z contracts k in [grp[2z], grp[2z] + grp[2z+1]) of the
  // shared A and B and writes its own C + z * c_split_stride): per-group weight gradients
  int grp_k;
};

// internal epilogue flag: C += result (chunked accumulation of split-K partials)
#define GNF_GEMM_ACCUM (1 << 30)

// splits > 1: split-K; partial z is written to C + z*c_split_stride (caller sets the
// stride and reduces the partials); epilogue options other than ACCUM must be off.
int gnf_gemm_launch(GemmArgs g, int splits, hipStream_t s);
// C[rows of group z] = epi(A[rows of group z] * B_z): ngroups row ranges (device table g.grp, see GemmArgs), one launch
int gnf_gemm_grouped_launch(GemmArgs g, int ngroups, hipStream_t s);
// number of partials gnf_gemm_launch(K, splits) writes (without ACCUM)
int64_t gnf_gemm_num_splits(int64_t K, int splits);

// gnf_gemm_split.hip: fp32-accurate products on the bf16 matrix pipe (three-way exact operand splits).  gnf_gemm_split_try
// returns 0 when a dedicated split kernel ran, 1 when the call is not its business, else an error code.
extern "C" int64_t gnf_gemm_split_ws_bytes(int64_t M, int64_t N, int64_t K);
extern "C" int gnf_gemm_split_enabled(void);
extern "C" const char* gnf_gemm_split_last_kernel(void);
int gnf_gemm_split_try(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn, float* C, int64_t scm,
                       int64_t scn, const float* bias, int relu, int64_t M, int64_t N, int64_t K, void* ws, int64_t ws_bytes,
                       hipStream_t s);
