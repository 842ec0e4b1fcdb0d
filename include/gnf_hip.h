/* gnf_hip.h -- C ABI of libgnf_hip.so: the MI355X (gfx950) kernels under the
 * Conditioner / Normalizer plug-in API of Graphical-Normalizing-Flows.
 *
 * The reference has no FFI of its own (it is pure Python on torch ops); the drop-in
 * boundary is the Python class protocol of models/ (SURVEY.md 8b).  Each entry point
 * below replaces the torch-op sequence of the cited reference lines and is what the
 * host-side mirror (graphical-normalizing-flows_amd/models) binds with ctypes.
 *
 * Conventions (all entry points):
 *   - plain C types only; device pointers are raw (tensor.data_ptr()); fp32 everywhere;
 *   - `stream` is a hipStream_t passed as void* (torch's current stream);
 *   - returns 0 on success, a negative GNF_E* code for bad arguments, or a positive
 *     hipError_t from the launch; never throws, never allocates, never synchronises;
 *   - the caller owns and sizes every buffer, including workspaces (gnf_*_ws_bytes);
 *   - no global mutable state: safe for one process per GPU and for several streams;
 *   - strides are in ELEMENTS, not bytes;
 *   - an EMPTY batch (B, n_img, M or the row count = 0) is a valid call, as it is in the
 *     reference (torch ops on [0, d] tensors): arrays sized by the batch may then be NULL
 *     (torch hands out a null data_ptr for them), nothing is launched over them, and the
 *     backward entry points still write ZERO parameter gradients.
 */
#ifndef GNF_HIP_H
#define GNF_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GNF_ABI_VERSION 8
#define GNF_EINVAL (-1)   /* bad argument (null pointer, negative size, ...)          */
#define GNF_ESHAPE (-2)   /* shape not supported by any compiled kernel instantiation */
#define GNF_EWS    (-3)   /* workspace too small                                      */

typedef void* gnf_stream_t;

int gnf_abi_version(void);

/* ---- Affine normalizer: models/Normalizers/AffineNormalizer.py:9-17 -------------------
 * mu = clamp(h[..,0],-5,5); sigma = exp(clamp(h[..,1],-5,2)); z = x*sigma + mu.
 * x,z,jac: [B,d] contiguous.  h[b,i,c] at b*h_sb + i*h_sd + c*h_sc (MADE hands over a
 * permuted view).  jac (= sigma), logdet (= sum_i log sigma = sum_i clamp(h1)) and logn
 * (= -0.5 sum_i (log 2pi + z^2), the NormalLogDensity of NormalizingFlowFactories.py:15-16
 * fused into the pass that produces z) may be NULL.  clamp_inplace != 0 writes the clamped
 * values back into h like the reference's clamp_ does. */
int gnf_affine_fwd(const float* x, float* h, int64_t h_sb, int64_t h_sd, int64_t h_sc,
                   float* z, float* jac, float* logdet, float* logn, int clamp_inplace,
                   int64_t B, int64_t d, gnf_stream_t stream);
/* Backward of the above.  gz: [B,d] or NULL; gjac: [B,d] or NULL; glogdet, glogn: [B] or
 * NULL (glogn: z's cotangent gains -z*glogn[b], z recomputed in the kernel).
 * gx: [B,d] or NULL.  gh[b,i,c] at b*g_sb + i*g_sd + c*g_sc receives components 0,1
 * (components >= 2 are never read by the normalizer: caller zero-fills them). */
int gnf_affine_bwd(const float* x, const float* h, int64_t h_sb, int64_t h_sd, int64_t h_sc,
                   const float* gz, const float* gjac, const float* glogdet, const float* glogn,
                   float* gx, float* gh, int64_t g_sb, int64_t g_sd, int64_t g_sc,
                   int64_t B, int64_t d, gnf_stream_t stream);
/* x = (z - mu)/sigma  (AffineNormalizer.py:14-17). */
int gnf_affine_inv(const float* z, const float* h, int64_t h_sb, int64_t h_sd, int64_t h_sc,
                   float* x, int64_t B, int64_t d, gnf_stream_t stream);

/* ---- row reductions: models/NormalizingFlow.py:70, NormalizingFlowFactories.py:15-16 --
 * logsum:  out[b] = sum_i log(jac[b,i]);        bwd: gjac[b,i] = g[b]/jac[b,i]
 * normal:  out[b] = -0.5*sum_i(log(2pi)+z^2);   bwd: gz[b,i]  = -z[b,i]*g[b]        */
int gnf_logsum_rows_fwd(const float* jac, float* out, int64_t B, int64_t d, gnf_stream_t stream);
int gnf_logsum_rows_bwd(const float* jac, const float* g, float* gjac, int64_t B, int64_t d, gnf_stream_t stream);
int gnf_normal_logdensity_fwd(const float* z, float* out, int64_t B, int64_t d, gnf_stream_t stream);
int gnf_normal_logdensity_bwd(const float* z, const float* g, float* gz, int64_t B, int64_t d, gnf_stream_t stream);
/* Both reductions of a flow step's tail in ONE pass over z and jac (the "gnf_nll_reduce" of SURVEY.md 8b; used behind
 * normalizers whose kernel cannot reduce over a row itself, i.e. the Monotonic one):
 *   logdet[b] = sum_i log(jac[b,i])   logn[b] = -0.5*sum_i(log(2pi)+z[b,i]^2)
 * bwd: gz[b,i] = (gz_in ? gz_in[b,i] : 0) - z[b,i]*glogn[b],  gjac[b,i] = glogdet[b]/jac[b,i]
 *      (glogdet / glogn / gz_in may be NULL = zero). */
int gnf_nll_reduce_fwd(const float* z, const float* jac, float* logdet, float* logn, int64_t B, int64_t d,
                       gnf_stream_t stream);
int gnf_nll_reduce_bwd(const float* z, const float* jac, const float* glogdet, const float* glogn, const float* gz_in,
                       float* gz, float* gjac, int64_t B, int64_t d, gnf_stream_t stream);
/* FCNormalizingFlow.loss (models/NormalizingFlow.py:144-146): out[0] = addend[0] - mean_b(logdet[b] + logn[b]) (addend: the
 * constraints term as a device scalar, or NULL = 0), one workgroup, fixed summation order;
 * bwd: glogdet[b] = glogn[b] = -g[0]/B (g: device scalar; the addend's cotangent is g itself). */
int gnf_nll_mean_fwd(const float* logdet, const float* logn, const float* addend, float* out, int64_t B,
                     gnf_stream_t stream);
int gnf_nll_mean_bwd(const float* g, float* glogdet, float* glogn, int64_t B, gnf_stream_t stream);
/* The same loss from z itself (round 5): out[0] = addend[0] - mean_b(logdet[b] + logN(z[b,:])), logN as in
 * NormalizingFlowFactories.py:15-16 -- the loss scores the z it is handed, whatever wrote it (no density remembered from
 * the forward pass).  One workgroup, B*d <= gnf_nll_loss_max_elems() (GNF_ESHAPE above: use gnf_normal_logdensity_fwd +
 * gnf_nll_mean_fwd).  bwd: gz[b,i] = g[0] z[b,i] / B, glogdet[b] = -g[0] / B. */
int64_t gnf_nll_loss_max_elems(void);
int gnf_nll_loss_fwd(const float* z, const float* logdet, const float* addend, float* out, int64_t B, int64_t d,
                     gnf_stream_t stream);
int gnf_nll_loss_bwd(const float* g, const float* z, float* gz, float* glogdet, int64_t B, int64_t d, gnf_stream_t stream);
/* out[n] = sum_m a[m*lda + n]  (bias gradients; deterministic two-level reduction).
 * ws: >= gnf_colsum_ws_bytes(M,N) bytes. */
int64_t gnf_colsum_ws_bytes(int64_t M, int64_t N);
int gnf_colsum(const float* a, int64_t lda, float* out, int64_t M, int64_t N, float* ws, gnf_stream_t stream);

/* ---- Linear / masked-Linear layers of the conditioner MLPs ----------------------------------------------------------
 * MADE's MaskedLinear (models/Conditionners/AutoregressiveConditioner.py:14-25: F.linear(x, mask * W, b)) and the plain
 * Linear + ReLU chains of CouplingMLP / DAGMLP, forward and autograd.  x: [M,K], W: [N,K] (nn.Linear layout), y: [M,N],
 * all contiguous.  The mask is either `mask` ([N,K], 0/1, may be NULL = none) or, when deg_out / deg_in are given, the
 * degree rule mask[o][i] = deg_in[i] <= deg_out[o] (strict != 0: <) of AutoregressiveConditioner.py:85-96 evaluated in
 * the kernel -- the caller guarantees that `mask` (still needed by the large-batch path) equals it.  M <= 128 rows run on
 * the weight-streaming kernels of gnf_linear.hip (no mask stream, no split-K partials), anything else on the tiled GEMM.
 *   fwd:    y  = act(x (W o mask)^T + b)                       relu != 0: act = ReLU
 *   bwd_x:  gx = (g (W o mask)) o [gate > 0]                   g: [M,N]; gate: [M,K] (the layer's input) or NULL
 *   bwd_w:  gW = (g^T a) o mask, gb = column sums of g         a: [M,K]; gb: [N] or NULL
 *   bwd:    bwd_w and bwd_x of one layer (what autograd runs for F.linear when both the weight and the input need a
 *           gradient); at M <= 128, and for tall batches with a narrow output (M >= 2048, N <= 64, K <= 128, no mask:
 *           MNISTCNN.fc2, DAGMLP), the two run in ONE launch.  gxsum ([K] or NULL): the column sums of gx = the bias
 *           gradient of the layer that produced `a`; gnf_linear_gxsum_fused() tells whether it comes out of the same
 *           launch (else: one more pass over gx)
 * ws: >= gnf_linear_ws_bytes(M,N,K) bytes. */
int64_t gnf_linear_ws_bytes(int64_t M, int64_t N, int64_t K);
int gnf_linear_fwd(const float* x, const float* W, const float* b, const float* mask, const float* deg_out,
                   const float* deg_in, int strict, int relu, float* y, int64_t M, int64_t N, int64_t K,
                   float* ws, int64_t ws_bytes, gnf_stream_t stream);
int gnf_linear_bwd_x(const float* g, const float* W, const float* mask, const float* deg_out, const float* deg_in,
                     int strict, const float* gate, float* gx, int64_t M, int64_t N, int64_t K,
                     float* ws, int64_t ws_bytes, gnf_stream_t stream);
int gnf_linear_bwd_w(const float* g, const float* a, const float* mask, const float* deg_out, const float* deg_in,
                     int strict, float* gW, float* gb, int64_t M, int64_t N, int64_t K,
                     float* ws, int64_t ws_bytes, gnf_stream_t stream);
int gnf_linear_bwd(const float* g, const float* W, const float* a, const float* mask, const float* deg_out,
                   const float* deg_in, int strict, const float* gate, float* gx, float* gW, float* gb, float* gxsum,
                   int64_t M, int64_t N, int64_t K, float* ws, int64_t ws_bytes, gnf_stream_t stream);
int gnf_linear_gxsum_fused(int64_t M, int64_t N, int64_t K, int masked);

/* ---- fp32 MFMA GEMM with fused masks / bias / ReLU ------------------------------------
 * Replaces F.linear(input, mask*weight, bias) (AutoregressiveConditioner.py:24-25), the
 * nn.Linear+ReLU chains of CouplingMLP / DAGMLP / MNISTCNN.fc* and their autograd
 * backward.  C[m,n] = epi( sum_k A[m,k] * (B[k,n] * Bmask[k,n]) ):
 *   A[m,k] at m*sam + k*sak;  B and Bmask (NULL = none) at k*sbk + n*sbn;
 *   epi: + bias[n] (NULL = none);  * Cmask[m*scmm + n*scmn] (NULL = none);
 *        relu if flags&GNF_GEMM_RELU;  * (gate[m*sgm + n*sgn] > 0) (NULL = none);
 *   C[m,n] at m*scm + n*scn.  v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain. */
#define GNF_GEMM_RELU 1
/* Optional split-K workspace: when gnf_gemm_ws_bytes(M,N,K) > 0 and `ws` holds at least
 * that many bytes, K is spread over several workgroups (few output tiles, long K: the
 * weight-gradient shapes) and a second kernel reduces the partials and applies the
 * epilogue.  ws == NULL always selects the single-pass kernel. */
/* name of the kernel family the most recent gnf_gemm (or Linear entry point routed through it) of the calling thread
 * dispatched to -- "gemm_tall_k", "gemm_wide_k", "gemm_kmajor_k", "gemm_vec_k<128,128>", ...: measurement and tests only. */
const char* gnf_gemm_last_kernel(void);
int64_t gnf_gemm_ws_bytes(int64_t M, int64_t N, int64_t K);
/* the share of gnf_gemm_ws_bytes the fp32-MFMA kernels' split-K partials need: a workspace of exactly this size keeps a call on
 * the fp32 kernels (measurement / tests) where the full size also admits the split-bf16 kernels below */
int64_t gnf_gemm_f32_ws_bytes(int64_t M, int64_t N, int64_t K);
int gnf_gemm(const float* A, int64_t sam, int64_t sak,
             const float* B, const float* Bmask, int64_t sbk, int64_t sbn,
             float* C, int64_t scm, int64_t scn,
             const float* bias,
             const float* Cmask, int64_t scmm, int64_t scmn,
             const float* gate, int64_t sgm, int64_t sgn,
             int flags, int64_t M, int64_t N, int64_t K, float* ws, int64_t ws_bytes, gnf_stream_t stream);

/* ---- the same product on the bf16 matrix pipe with fp32 accuracy (round 6; models/MLP.py:44 and its autograd) ---------
 * Every fp32 operand is split exactly into three bf16 numbers (hi + mid + lo, round-to-nearest at each level) on its way
 * into LDS; a product is the sum of its six leading cross terms (hi hi | hi mid, mid hi | hi lo, lo hi, mid mid), each exact
 * in the fp32 accumulator of v_mfma_f32_16x16x32_bf16; `classes` = 1 / 2 / 3 accumulators per output tile of the GENERAL
 * kernel (3: the magnitude classes are summed separately and added once at the end; measurement), `classes` = 0: the
 * product's choice -- a dedicated kernel (gnf_gemm_split_last_kernel names it) when the shape has one and `ws` holds
 * gnf_gemm_split_ws_bytes(M, N, K) bytes (pre-split fragment-major planes of the small operand / split-K partials).  C[m,n] = epi(sum_k A[m,k] B[k,n]), epi = + bias[n], relu.
 * splits > 1: K is cut into `splits` ranges, partial z goes to C + z * c_split_stride (no epilogue; the caller adds them).
 * Operands must be finite (inf * 0 cross terms would give NaN where an fp32 product gives inf).
 * Measured against the fp32-MFMA kernels and an fp64 product: profiles/r06_split_bf16_error.txt. */
int64_t gnf_gemm_split_ws_bytes(int64_t M, int64_t N, int64_t K);
const char* gnf_gemm_split_last_kernel(void);
/* 1 unless the environment holds GNF_TRUE_F32=1: gnf_gemm (and the Linear entry points routed through it) then never leave the
 * v_mfma_f32_* kernels. */
int gnf_gemm_split_enabled(void);
int gnf_gemm_split_bf16(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                        float* C, int64_t scm, int64_t scn, const float* bias, int relu,
                        int64_t M, int64_t N, int64_t K, int classes, int splits, int64_t c_split_stride,
                        void* ws, int64_t ws_bytes, gnf_stream_t stream);

/* ---- DAG conditioner gate: models/Conditionners/DAGConditioner.py:94-166 --------------
 * e[(b*d+i)*ld_e + j] = x[b,j] * gate(importance(A[i,j])) (+ one-hot of i in columns
 * d..2d-1 when hot != 0, ld_e >= 2d).
 *   imp_mode 0: raw A (DAG:151-153)  1: soft threshold 2(sigmoid(2A^2)-.5) (:118-119)
 *            2: soft * (soft > h_thresh)  3: A^2 * (A^2 > h_thresh)   (:121-124)
 *   gate_mode 0: deterministic x*imp   1: Gumbel-softmax gate (:95-103)
 *             2: noise gate imp*(x + n*|1-imp|) (:114-116)
 * Randomness: if u1 != NULL the uniforms (gate 1: u1,u2; gate 2: u1 holds N(0,1)
 * samples) are read from [B,d,d] arrays (parity tests); otherwise a Philox4x32-10
 * stream keyed by (seed, offset) is used -- counter = (b*d+i)*ceil(d/4) + j/4, one call
 * serving the four adjacent columns 4(j/4) .. 4(j/4)+3 -- and the backward regenerates the
 * same numbers from the same (seed, offset). */
/* ws: >= gnf_dag_gate_fwd_ws_bytes(d) bytes (per-(i,j) table of importance / gate constants). */
int64_t gnf_dag_gate_fwd_ws_bytes(int64_t d);
int gnf_dag_gate_fwd(const float* x, const float* A, float* e, int64_t ld_e,
                     int imp_mode, int gate_mode, float h_thresh, float temperature,
                     const float* u1, const float* u2, uint64_t seed, uint64_t offset,
                     int hot, float* ws, int64_t B, int64_t d, gnf_stream_t stream);
/* ge: [(B*d), ld_e].  gA: [d,d] (written, not accumulated) or NULL; gx: [B,d] or NULL.
 * tab_fwd: the workspace the matching gnf_dag_gate_fwd call was given, if the caller kept it intact (its first
 * gnf_dag_gate_fwd_ws_bytes(d) bytes hold the per-(i,j) table of the same A and gate settings), else NULL = recomputed.
 * ws: >= gnf_dag_gate_bwd_ws_bytes(B,d). */
int64_t gnf_dag_gate_bwd_ws_bytes(int64_t B, int64_t d);
int gnf_dag_gate_bwd(const float* x, const float* A, const float* ge, int64_t ld_e,
                     int imp_mode, int gate_mode, float h_thresh, float temperature,
                     const float* u1, const float* u2, uint64_t seed, uint64_t offset, const float* tab_fwd,
                     float* gA, float* gx, float* ws, int64_t B, int64_t d, gnf_stream_t stream);

/* ---- structural zeros of the gate backward (round 5) --------------------------------------------------------------
 * dL/dA[i,j] = dP/dA[i,j] * sum_b dL/de[b,i,j] x[b,j] dgate/dP (DAGConditioner.py:118-119: dP/dA = 8 A s(1-s) is
 * EXACTLY zero wherever A is zero -- 97.2 % of MNIST_A_prior(28, 2), NormalizingFlowFactories.py:35-46), so with no
 * gradient wanted for x the cotangent of e is only needed at the columns j of row i with dP/dA[i,j] != 0.
 *   plan (gnf_dag_gate_plan_bytes(d) bytes, written by gnf_dag_gate_fwd_plan from the very table its gate uses):
 *     int32 count[d] (columns with dP/dA != 0 in row i, also when more than GNF_DAG_PLAN_KC),
 *     int16 cols[d][GNF_DAG_PLAN_KC] (those columns in ascending order, -1 padded),
 *     int32 overflow (1 when some count exceeds GNF_DAG_PLAN_KC).
 *   Consumers (gnf_mnistcnn_conv_bwd_cols, gnf_dag_gate_bwd_cols) use the lists only if overflow == 0 and run
 *   their dense code otherwise; the decision is taken on the device, nothing is cached on the host.
 *   ge_cols [(B*d), GNF_DAG_PLAN_KC]: ge_cols[(b*d+i)*KC + k] = dL/de[b,i,cols[i][k]] for k < count[i].
 * gnf_dag_gate_fwd_plan = gnf_dag_gate_fwd that also writes the plan (plan == NULL: exactly gnf_dag_gate_fwd).
 * gnf_dag_gate_bwd_cols: gA only (x frozen), no one-hot columns (ld_e = d); tab_fwd (required) = the forward's ws;
 *   ge [(B*d), d] is read when the plan overflows, ge_cols otherwise; accumulate != 0: gA += (A's other contribution, the
 *   acyclicity term's, is already there: no separate add launch). */
#define GNF_DAG_PLAN_KC 32
int64_t gnf_dag_gate_plan_bytes(int64_t d);
int gnf_dag_gate_fwd_plan(const float* x, const float* A, float* e, int64_t ld_e,
                          int imp_mode, int gate_mode, float h_thresh, float temperature,
                          const float* u1, const float* u2, uint64_t seed, uint64_t offset,
                          int hot, float* ws, int32_t* plan, int64_t plan_bytes, int64_t B, int64_t d,
                          gnf_stream_t stream);
int64_t gnf_dag_gate_bwd_cols_ws_bytes(int64_t B, int64_t d);
int gnf_dag_gate_bwd_cols(const float* x, const float* ge, const float* ge_cols, const int32_t* plan,
                          int imp_mode, int gate_mode, float temperature,
                          const float* u1, const float* u2, uint64_t seed, uint64_t offset, const float* tab_fwd,
                          float* gA, int accumulate, float* ws, int64_t B, int64_t d, gnf_stream_t stream);

/* ---- DAG acyclicity + l1 term: DAGConditioner.get_power_trace / loss (DAGConditioner.py:176-194, 268-271) --------
 * The d x d matrix power stays on the GEMM library; these entries fuse the ~40 elementwise / reduction
 * launches around it.  alpha, lambd, c, dag_const, l1_weight: device scalars (the conditioner's buffers).
 *   prep:  Bm = I + min(1, alpha) * alpha_factor * A o A                                 (:184-190)
 *   value: out4 = [loss, h, coef, l1/d^2] with h = tr(Bm^k) - d = sum_ij P_ij Bm_ji - d, P = Bm^(k-1) (NULL when k = 0),
 *          loss = dag_const (lambd h + c/2 h^2) + l1 mean|A|, coef = dag_const (lambd + c h) k 2 alpha_eff.
 *          ws: >= 2 * 256 floats.
 *   bwd:   gA = g (coef A o P^T + l1/d^2 sign(A))  with g the device scalar d/d loss.       (written, not accumulated) */
int gnf_dag_loss_prep(const float* A, const float* alpha, float alpha_factor, float* Bm, int64_t d, gnf_stream_t stream);
int gnf_dag_loss_value(const float* A, const float* Bm, const float* P, const float* alpha, float alpha_factor,
                       const float* lambd, const float* c, const float* dag_const, const float* l1_weight, int k,
                       float* out4, float* ws, int64_t d, gnf_stream_t stream);
int gnf_dag_loss_bwd(const float* A, const float* P, const float* out4, const float* g, float* gA, int64_t d,
                     gnf_stream_t stream);

/* ---- Monotonic (UMNN) normalizer: models/Normalizers/MonotonicNormalizer.py:21-83 -----
 * Integrand net: Linear(1+c,H1) ReLU ... Linear(H_last,1) ELU+1.05 evaluated on rows
 * (x[b,i], h[b,i,:]).  `nl` = number of Linear layers (>= 2); W[l]: [out_l,in_l]
 * row-major, b[l]: [out_l]; dims[0]=1+c, dims[l+1]=out_l, dims[nl]=1.
 *   z   = h[..,0] + (xT/2) * sum_k cc_w[k] f(xT (cc_t[k]+1)/2 ; h),  xT = S*(x/S)
 *   jac = f(x; h)
 * cc_w, cc_t: S+1 device floats (host-built Clenshaw-Curtis rule; UMNN 1.0's
 * construction, see oracle/gnf_oracle.py: parity unpinned).  n = B*d elements; element
 * e=(b,i): x[e], h at b*h_sb + i*h_sd + c*h_sc with b=e/d, i=e%d.
 * pack: >= gnf_monotonic_pack_floats(...) floats of workspace holding the padded weight
 * image (written by gnf_monotonic_pack, read by fwd/bwd/inv of the same step). */
#define GNF_MONO_MAX_LAYERS 8
typedef struct {
  int nl;                                   /* number of Linear layers */
  int dims[GNF_MONO_MAX_LAYERS + 1];
  const float* W[GNF_MONO_MAX_LAYERS];
  const float* b[GNF_MONO_MAX_LAYERS];
} gnf_mono_net;

int64_t gnf_monotonic_pack_floats(const gnf_mono_net* net);
int gnf_monotonic_pack(const gnf_mono_net* net, float* pack, gnf_stream_t stream);
int gnf_monotonic_fwd(const float* pack, const gnf_mono_net* net,
                      const float* x, const float* h, int64_t h_sb, int64_t h_sd, int64_t h_sc,
                      const float* cc_w, const float* cc_t, int S,
                      float* z, float* jac, int64_t B, int64_t d, gnf_stream_t stream);
/* Wide integrand nets (hidden widths 97..112, 145..160: the [100]^3 / [150]^3 nets of UCIExperiments.yml) and the peeled
 * narrow nets (all hidden widths equal, 49..51: the reference's default [50, 50, 50]) run their hidden->hidden products on
 * the bf16 matrix pipe with exact 3 x bf16 operand splits, six cross terms and fp32 accumulation (as close to an fp64
 * evaluation as the fp32-MFMA kernels or closer, tests/test_gpu_mono_split.py).  GNF_TRUE_F32=1 (read once per process)
 * selects the fp32-MFMA kernels for gnf_monotonic_fwd; gnf_monotonic_fwd_f32 always does (same arguments: the A/B handle
 * of the tests and tools).  gnf_monotonic_fwd_kernel(): name of the kernel family the last forward call of this PROCESS
 * launched ("mono_fwd_wide_split_k", "mono_fwd_wide_k", "mono_fwd_x_k<split>", "mono_fwd_k"); reporting only, not
 * synchronised. */
int gnf_monotonic_fwd_f32(const float* pack, const gnf_mono_net* net,
                          const float* x, const float* h, int64_t h_sb, int64_t h_sd, int64_t h_sc,
                          const float* cc_w, const float* cc_t, int S,
                          float* z, float* jac, int64_t B, int64_t d, gnf_stream_t stream);
const char* gnf_monotonic_fwd_kernel(void);
/* 20-step bisection on [-20,20] with the quadrature inside (MonotonicNormalizer.py:69-83). */
int gnf_monotonic_inv(const float* pack, const gnf_mono_net* net,
                      const float* z, const float* h, int64_t h_sb, int64_t h_sd, int64_t h_sc,
                      const float* cc_w, const float* cc_t, int S,
                      float* x, int64_t B, int64_t d, gnf_stream_t stream);
/* The same inverse with a strided / scattered result: element (b, j) of the [B, d] problem is written to
 * x[x_row_off[b] + j * x_sd] (x_row_off [B] int32 on the device, element offsets).  The level-scheduled inversion of a DAG flow
 * (NormalizingFlow.py:98-107 restated per topological level) solves a [level rows, batch] problem whose result belongs in
 * columns rows[.] of the [batch, d] sample: x_row_off = rows, x_sd = d -- no transposing copy, no index_put launch. */
int gnf_monotonic_inv_scatter(const float* pack, const gnf_mono_net* net,
                              const float* z, const float* h, int64_t h_sb, int64_t h_sd, int64_t h_sc,
                              const float* cc_w, const float* cc_t, int S,
                              float* x, const int32_t* x_row_off, int64_t x_sd, int64_t B, int64_t d, gnf_stream_t stream);
/* Backward with UMNN's conventions: gx = gz*f(x;h) + gjac*df/dx(x;h) (Leibniz rule);
 * gh, gW, gb = quadrature of df/dh, df/dtheta weighted by gz*xT/2, plus the gjac path,
 * plus gz on h[..,0].  gW[l]/gb[l] are WRITTEN (same shapes as W[l]/b[l]).
 * gh[e,c] at b*g_sb + i*g_sd + c*g_sc.  ws: >= gnf_monotonic_bwd_ws_bytes(...) bytes. */
int64_t gnf_monotonic_bwd_ws_bytes(const gnf_mono_net* net, int S, int64_t B, int64_t d);
int gnf_monotonic_bwd(const float* pack, const gnf_mono_net* net,
                      const float* x, const float* h, int64_t h_sb, int64_t h_sd, int64_t h_sc,
                      const float* cc_w, const float* cc_t, int S,
                      const float* gz, const float* gjac,
                      float* gx, float* gh, int64_t g_sb, int64_t g_sd, int64_t g_sc,
                      float* const* gW, float* const* gb,
                      void* ws, int64_t ws_bytes, int64_t B, int64_t d, gnf_stream_t stream);
/* Wide integrand nets (H = 145..160): the chain wavefronts of the backward (recompute and data gradient through the
 * hidden->hidden layers) run on the bf16 matrix pipe with exact 3 x bf16 splits, like gnf_monotonic_fwd; peeled narrow nets
 * (H = 49..51, at most three hidden layers): the same for the 48 x 48 main blocks.  The weight-gradient contraction stays
 * fp32 MFMA.  GNF_TRUE_F32=1 / gnf_monotonic_bwd_f32 (same arguments): all fp32 MFMA.  gnf_monotonic_bwd_kernel(): the
 * chain kernel the last backward call of this PROCESS launched (autograd calls from its own thread): "mono_bwd_wide_k<split>",
 * "mono_bwd_wide_k<f32>", "mono_bwd_pair_x_k<split>", "mono_bwd_k". */
int gnf_monotonic_bwd_f32(const float* pack, const gnf_mono_net* net,
                          const float* x, const float* h, int64_t h_sb, int64_t h_sd, int64_t h_sc,
                          const float* cc_w, const float* cc_t, int S,
                          const float* gz, const float* gjac,
                          float* gx, float* gh, int64_t g_sb, int64_t g_sd, int64_t g_sc,
                          float* const* gW, float* const* gb,
                          void* ws, int64_t ws_bytes, int64_t B, int64_t d, gnf_stream_t stream);
const char* gnf_monotonic_bwd_kernel(void);

/* ---- MNISTCNN convolutional front: models/MLP.py:36-41 as the DAG embedding net --------
 * (ImageExperiments / NormalizingFlowFactories.py:83-86: size_img = [1,28,28]).
 * e: [n_img, 784] masked images; W1 [16,1,3,3], b1 [16], W2 [16,16,3,3], b2 [16].
 * pooled: [n_img, 2304] = flatten(maxpool2(conv2(relu(conv1(e))))) in [16,12,12] order;
 * argmax: [n_img, 2304] bytes, index 0..3 of the first maximum of each 2x2 window in scan
 * order (what torch's max_pool2d records), consumed by the backward.
 * exact_ties = 0: conv2 in the Winograd F(2x2,3x3) domain (2.25x fewer MFMAs); pool windows whose four values are
 *   EXACTLY equal in exact arithmetic (constant image regions: a deterministic gate on a windowed A) are then decided
 *   by rounding noise of the transforms -- an equally valid subgradient, not torch's.
 * exact_ties = 1: conv2 as the direct implicit GEMM: equal patches give bit-equal outputs, the first maximum wins as
 *   in torch.  models.DAGConditioner selects it for deterministic gates that cannot use the sparse front. */
int gnf_mnistcnn_conv_fwd(const float* e, const float* W1, const float* b1, const float* W2, const float* b2,
                          float* pooled, unsigned char* argmax, int64_t n_img, int exact_ties, gnf_stream_t stream);
/* Backward: recomputes conv1, writes ge [n_img,784] and the parameter gradients (written,
 * not accumulated).  ws: >= gnf_mnistcnn_conv_bwd_ws_bytes(n_img) bytes. */
int64_t gnf_mnistcnn_conv_bwd_ws_bytes(int64_t n_img);
int gnf_mnistcnn_conv_bwd(const float* e, const float* W1, const float* b1, const float* W2,
                          const float* g_pooled, const unsigned char* argmax,
                          float* ge, float* gW1, float* gb1, float* gW2, float* gb2,
                          void* ws, int64_t ws_bytes, int64_t n_img, gnf_stream_t stream);
/* The same backward for the masked copies of a DAG conditioner whose gate backward needs the cotangent of e at the plan's
 * columns only (above): image n is the masked copy (b, i) = (n / d_plan, n % d_plan), d_plan = 784.  Unless a row of the
 * plan overflows, ge is NOT written: ge_cols [n_img, GNF_DAG_PLAN_KC] receives dL/de at the row's columns, and the
 * kernel skips the W1^T dpre1 products of the conv1 positions no such column reaches (7 of 11 position groups at the
 * MNIST prior).  The parameter gradients are those of gnf_mnistcnn_conv_bwd, bit for bit.  plan == NULL: that call. */
int gnf_mnistcnn_conv_bwd_cols(const float* e, const float* W1, const float* b1, const float* W2,
                               const float* g_pooled, const unsigned char* argmax,
                               float* ge, const int32_t* plan, int64_t d_plan, float* ge_cols,
                               float* gW1, float* gb1, float* gW2, float* gb2,
                               void* ws, int64_t ws_bytes, int64_t n_img, gnf_stream_t stream);

/* ---- sparse masked-image front for a DETERMINISTIC DAG gate ---------------
 * Replaces, for evaluation / sampling, the chain  e = x * P[i]  (DAGConditioner.py:142-153, deterministic branches)
 * -> conv1/ReLU/conv2/maxpool (MLP.py:36-41) -> fc1 + ReLU (MLP.py:43-44)  when every row i of the importance matrix
 * P [784,784] is zero outside the 5x5 window around pixel i (MNIST_A_prior with kernel 2,
 * NormalizingFlowFactories.py:35-46): only the 14x14 crop around the window is convolved and fc1 contracts the
 * 5x5x16 pooled block that can differ from the constant background (exact, not an approximation).
 *   x [B,784];  pix [R] (device): the masked copies (pixel indices i) to evaluate, SORTED by crop origin
 *   g(i) = 8*o(i/28) + o(i%28), o(p) = clamp(floor((p-6)/2), 0, 7);
 *   groups [2*64] (device): for each origin g the first output row and the number of output rows (= B * number of
 *   pix entries with that origin);  max_group_rows: the largest of those counts (host value, sizes the grid);
 *   W1 [16,1,3,3], b1, W2 [16,16,3,3], b2: conv parameters;  Wfc1 [F,2304], bfc1 [F]  (F % 4 == 0);
 *   h1 [R*B, F] (out): relu(fc1(...)) of masked copy pix[r] of sample b at row r*B + b;
 *   pd_save [R*B, 400], argmax_save [R*B, 400] bytes (both or neither; NULL for inference): what the backward
 *   needs -- the pooled block minus the background in [cell][channel] order and the max-pool winners.
 * ws: >= gnf_mnistcnn_sparse_ws_bytes(R*B, F) bytes. */
int64_t gnf_mnistcnn_sparse_ws_bytes(int64_t n_rows, int64_t F);
int gnf_mnistcnn_sparse_fwd(const float* x, int64_t B, const float* P, const int32_t* pix, int64_t R,
                            const int32_t* groups, int64_t max_group_rows,
                            const float* W1, const float* b1, const float* W2, const float* b2,
                            const float* Wfc1, const float* bfc1, int64_t F,
                            float* h1, float* pd_save, unsigned char* argmax_save,
                            void* ws, int64_t ws_bytes, gnf_stream_t stream);
/* The training form of the same call (a backward follows): pd_save / argmax_save are required, and instead of a
 * workspace whose first R*B*400 floats the call would never touch (pd goes to pd_save), the caller hands in ONLY the buffer
 * of the parameter-only tables (>= gnf_mnistcnn_sparse_prep_bytes(F) bytes), which it keeps alive and passes to
 * gnf_mnistcnn_sparse_bwd_tables.  Same h1 / pd_save / argmax_save as gnf_mnistcnn_sparse_fwd, bit for bit.
 * (Round 6: an in-flight forward pinned 125 MB of dead workspace per conditioner at B = 100.) */
int gnf_mnistcnn_sparse_fwd_train(const float* x, int64_t B, const float* P, const int32_t* pix, int64_t R,
                                  const int32_t* groups, int64_t max_group_rows,
                                  const float* W1, const float* b1, const float* W2, const float* b2,
                                  const float* Wfc1, const float* bfc1, int64_t F,
                                  float* h1, float* pd_save, unsigned char* argmax_save,
                                  void* tables, int64_t tables_bytes, gnf_stream_t stream);
/* The same front for a caller that evaluates it MANY times with unchanged parameters (the 109 levels of one sampling
 * pass, NormalizingFlow.py:98-107 under ImageExperiments.py:341-350): the parameter-only tables -- the fc1 weight columns
 * of each crop origin, conv2's response to the all-zero image and fc1 of that background -- are built once by
 * gnf_mnistcnn_sparse_prepare into `prep` (>= gnf_mnistcnn_sparse_prep_bytes(F) bytes) and read by every
 * gnf_mnistcnn_sparse_fwd_prepared call (inference only: nothing is saved for a backward; ws >= R*B*400*4 bytes).
 * Same h1 as gnf_mnistcnn_sparse_fwd, bit for bit. */
int64_t gnf_mnistcnn_sparse_prep_bytes(int64_t F);
int gnf_mnistcnn_sparse_prepare(const float* b1, const float* W2, const float* b2, const float* Wfc1, const float* bfc1,
                                int64_t F, void* prep, int64_t prep_bytes, gnf_stream_t stream);
int gnf_mnistcnn_sparse_fwd_prepared(const float* x, int64_t B, const float* P, const int32_t* pix, int64_t R,
                                     const int32_t* groups, int64_t max_group_rows,
                                     const float* W1, const float* b1, const float* W2, const float* b2, int64_t F,
                                     const void* prep, float* h1, void* ws, int64_t ws_bytes, gnf_stream_t stream);
/* The prepared front followed by fc2 (MLP.py:47: out = fc2(relu(fc1(.)))) in the same call: the fc1 + ReLU + fc2 chain of
 * the masked copies is ONE launch (a workgroup per 16 rows of a crop origin, relu(fc1) never leaves LDS) -- two launches per
 * level of a sampling pass instead of three, ~7 us instead of ~23 after the crop kernel.
 *   Wfc2 [out_d, F], bfc2 [out_d];  h2 [R*B, out_d] (out), row r*B + b as h1 above.
 * GNF_ESHAPE unless F == 128 and out_d <= 32 (the reference's MNISTCNN: fc1 2304 -> 128, fc2 128 -> out_d = 30 for the
 * Monotonic normalizer's conditioner, ImageExperiments.py:60-66); other widths take gnf_mnistcnn_sparse_fwd_prepared and
 * gnf_linear_fwd.  Equal to that pair up to fp32 summation order.  ws >= R*B*400*4 bytes. */
int gnf_mnistcnn_sparse_fwd_prepared_fc2(const float* x, int64_t B, const float* P, const int32_t* pix, int64_t R,
                                         const int32_t* groups, int64_t max_group_rows,
                                         const float* W1, const float* b1, const float* W2, const float* b2, int64_t F,
                                         const void* prep, const float* Wfc2, const float* bfc2, int64_t out_d,
                                         float* h2, void* ws, int64_t ws_bytes, gnf_stream_t stream);
/* Backward w.r.t. the network parameters (training with a frozen deterministic gate: P and x get no gradient).
 * g_h1 [R*B, F]: cotangent of h1 with the ReLU already applied (zero where h1 == 0).  Gradients are written, not
 * accumulated: gW1 [16,1,3,3], gb1 [16], gW2 [16,16,3,3], gb2 [16], gWfc1 [F,2304], gbfc1 [F].
 * The fc1 weight gradient is contracted per CHUNK of output rows so that the few large origins do not serialise:
 *   kgroups [2*n_kgroups] (device): (first output row, row count) of each chunk, a chunk lying inside one origin's
 *   rows, the chunks of an origin consecutive;  origin_chunks [2*64] (device): (first chunk, number of chunks). */
int64_t gnf_mnistcnn_sparse_bwd_ws_bytes(int64_t n_rows, int64_t F, int64_t n_kgroups);
/* gnf_mnistcnn_sparse_bwd against the parameter-only tables the FORWARD of the same step built (tables = the forward's
 * `(float*)ws + R*B*400`, or a gnf_mnistcnn_sparse_prepare buffer: fc1 column blocks per crop origin | background response): the
 * two table launches (~30 us of a 2.6 ms frozen-gate step) run once per step instead of twice.  tables == NULL: builds them. */
int gnf_mnistcnn_sparse_bwd_tables(const float* x, int64_t B, const float* P, const int32_t* pix, int64_t R,
                                   const int32_t* groups, int64_t max_group_rows,
                                   const int32_t* kgroups, int64_t n_kgroups, const int32_t* origin_chunks,
                                   const float* W1, const float* b1, const float* W2, const float* b2,
                                   const float* Wfc1, int64_t F, const void* tables,
                                   const float* pd, const unsigned char* argmax, const float* g_h1,
                                   float* gW1, float* gb1, float* gW2, float* gb2, float* gWfc1, float* gbfc1,
                                   void* ws, int64_t ws_bytes, gnf_stream_t stream);
int gnf_mnistcnn_sparse_bwd(const float* x, int64_t B, const float* P, const int32_t* pix, int64_t R,
                            const int32_t* groups, int64_t max_group_rows,
                            const int32_t* kgroups, int64_t n_kgroups, const int32_t* origin_chunks,
                            const float* W1, const float* b1, const float* W2, const float* b2,
                            const float* Wfc1, int64_t F,
                            const float* pd, const unsigned char* argmax, const float* g_h1,
                            float* gW1, float* gb1, float* gW2, float* gb2, float* gWfc1, float* gbfc1,
                            void* ws, int64_t ws_bytes, gnf_stream_t stream);

/* ---- Adam on one flat fp32 buffer (torch.optim.Adam semantics, L2 weight decay) --------
 * ImageExperiments.py:173 / UCIExperiments.py:97; used by the data-parallel harness after
 * the single RCCL all-reduce.  grad_scale multiplies the (summed) gradient first.  Hyper-parameters are doubles:
 * 1-beta, lr/(1-beta1^t) and sqrt(1-beta2^t) are formed in double and rounded once, as torch.optim.Adam forms them
 * (tested against it on the device, 1e-6). */
int gnf_adam_step(float* p, const float* g, float* m, float* v, int64_t n,
                  double lr, double beta1, double beta2, double eps, double weight_decay,
                  double grad_scale, int step, gnf_stream_t stream);

/* Same update (torch.optim.Adam at ImageExperiments.py:173 / UCIExperiments.py:97), with the step count in device
 * memory: step_dev points to TWO ints, {number of steps already taken, 0} -- the second is the ticket counter by which
 * the last workgroup of the launch increments the first when advance != 0, and reads 0 again afterwards (a step over
 * several disjoint runs of the flat buffer, e.g. around a frozen parameter, advances on its last launch only):
 * nothing step-dependent is passed by value, so a captured hipGraph of a whole training step can be replayed
 * (gnf_hip.dp.GraphedStep -- the launch-bound configurations). */
int gnf_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n,
                      double lr, double beta1, double beta2, double eps, double weight_decay,
                      double grad_scale, int* step_dev, int advance, gnf_stream_t stream);

/* ---- device-ceiling probes (measurement aids for bench.py) ---------------
 * gnf_probe_mfma_f32 launches `blocks` workgroups of 8 wavefronts that do nothing but
 * independent v_mfma_f32_16x16x4_f32 chains and returns the number of flops the launch issues
 * (negative on error); gnf_probe_copy is a STREAM copy of n floats (n % 4 == 0). */
int64_t gnf_probe_mfma_f32(float* out, int iters, int blocks, gnf_stream_t stream);
int gnf_probe_copy(float* dst, const float* src, int64_t n, gnf_stream_t stream);
/* an empty grid of `grid` workgroups of `block` threads: the launch-ramp floor of a kernel of that launch shape (measurement) */
int gnf_probe_empty(int64_t grid, int block, gnf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
