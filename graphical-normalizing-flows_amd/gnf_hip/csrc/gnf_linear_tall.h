// Internal (not part of the C ABI): the tall-batch / narrow-output Linear kernels (gnf_linear_tall.hip) behind
// gnf_linear_fwd / gnf_linear_bwd.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// M >= 2048 rows, N <= 64 out units, K <= 128 a multiple of 4, no mask
bool gnf_linear_tall_ok(int64_t M, int64_t N, int64_t K);
bool gnf_linear_tall_fwd_ok(int64_t M, int64_t N, int64_t K);      // forward only: any M > 128
// floats of workspace gnf_linear_tall_bwd needs (per-workgroup partial weight / bias gradients)
int64_t gnf_linear_tall_ws_floats(int64_t M, int64_t N, int64_t K);
int gnf_linear_tall_fwd(const float* x, const float* W, const float* b, int relu, float* y, int64_t M, int64_t N, int64_t K,
                        hipStream_t s);
// gx = (g W) o [gate > 0] (gate: the layer's input `a` or NULL), gW = g^T a, gb = colsum g (gb may be NULL),
// gxsum = colsum gx ([K], may be NULL)
int gnf_linear_tall_bwd(const float* g, const float* W, const float* a, const float* gate, float* gx, float* gW, float* gb,
                        float* gxsum, int64_t M, int64_t N, int64_t K, float* ws, hipStream_t s);

// weight gradient alone: gW[N][K] = g^T a, gb[N] (+)= colsum g for tall g [M x N <= 64] (pitch ldg) and a [M x K <= 64]
// (pitch lda); partW: >= parts * N * K floats, partB: >= parts * N floats with parts = gnf_linear_tall_wgrad_parts(M)
bool gnf_linear_tall_wgrad_ok(int64_t M, int64_t N, int64_t K, int64_t ldg, int64_t lda);
int gnf_linear_tall_wgrad_parts(int64_t M);
int gnf_linear_tall_wgrad(const float* g, int64_t ldg, const float* a, int64_t lda, float* gW, float* gb, int accumulate_b,
                          int64_t M, int64_t N, int64_t K, float* partW, float* partB, hipStream_t s);
