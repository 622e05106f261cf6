/*
 * dpgp.h — C ABI of libdpgp_hip.so: the DP-GP-LVM variational-ELBO inner loop on AMD MI355X (gfx950), hand-written HIP.
 *
 * The reference (AndrewRLawrence/dp_gp_lvm) has no FFI: its boundary for this path is the Python operator API
 *   k_ard_rbf / Kernel.covariance_matrix | covariance_diag | psi_0 | psi_1 | psi_2   (src/kernels/rbf_kernel.py:26-203,
 *                                                                                     src/kernels/interfaces/kernel.py:204-280)
 *   dp_gp_lvm(...).objective                                                          (src/models/dp_gp_lvm.py:100-154)
 * which dispatches to TensorFlow 1.15 ops (tf.matmul / tf.exp / tf.cholesky / tf.matrix_triangular_solve ...).
 * These entry points are what a binding of that API would call instead of TensorFlow; dp_gp_lvm_amd/ binds them with
 * ctypes (dp_gp_lvm_amd/_lib.py), INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (e.g. torch's allocator); row-major, contiguous, leading batch
 *    dimension; the library keeps no global state and allocates nothing: scratch is a caller-provided workspace whose
 *    size comes from the matching *_workspace_bytes() query (a pure host function, callable without a GPU);
 *  - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work on it and returns (graph-capturable);
 *  - return value: DPGP_OK (0), or -(1-based index of the first bad argument), or DPGP_ERR_LAUNCH (-100) if the HIP
 *    launch itself failed.  A non-positive-definite matrix is never an abort: it is reported LAPACK-style in the
 *    caller-provided device array info[B] (0 = ok, j>0 = leading minor j not positive; for the fused ELBO, M + j
 *    refers to the second factorisation);
 *  - suffix _f32 / _f64 is the arithmetic AND storage type of the call;  `s` is always the DIAGONAL [N,Q] of q(X)'s
 *    covariance (the reference API takes [N,Q,Q] and reads only its diagonal: rbf_kernel.py:151,182);
 *  - B = kernel batch (= D output dims in dp_gp_lvm), N observations, M inducing points, Q latent dims (Q <= 30).
 */
#ifndef DPGP_H
#define DPGP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DPGP_OK 0
#define DPGP_ERR_LAUNCH (-100)
#define DPGP_FLAG_NOISE 1  /* include_noise  (only honoured when x1 == NULL: rbf_kernel.py:80)  */
#define DPGP_FLAG_JITTER 2 /* include_jitter (only honoured when x1 == NULL: rbf_kernel.py:86)  */
#define DPGP_MAX_Q 30

/* algorithm selectors for the calls that have a matrix-core and a plain-VALU implementation */
#define DPGP_ALGO_AUTO 0     /* matrix-core kernels (the product path); fp32 psi2 = pair-tile GEMM, f16 hi/lo operands */
#define DPGP_ALGO_PLAIN 1    /* straightforward one-thread-per-element HIP kernels (cross-check)                  */
#define DPGP_ALGO_MFMA_F32 2 /* as AUTO, but fp32 psi2 on v_mfma_f32_16x16x4_f32 (exact fp32 products; slower)    */
#define DPGP_ALGO_PATCH_F16 3 /* as AUTO, but fp32 psi2 by the per-observation 64x64 patch kernel (round-1 form)    */

/* precision modes of the fused ELBO */
#define DPGP_PREC_F32 0   /* psi-statistics fp32, Cholesky chain fp32                          */
#define DPGP_PREC_MIXED 1 /* psi-statistics fp32 (MFMA), Cholesky chain + reductions fp64      */
#define DPGP_PREC_F64 2   /* everything fp64                                                   */
#define DPGP_PREC_MIXED_FAST 4  /* dpgp_elbo_grad_psi[_ex], dpgp_elbo_step only: as DPGP_PREC_MIXED, but the second products of the pair-tile
                                 * stage B take the exponentials as their f16 roundings alone (11 significant bits instead of 22; the pass
                                 * that also yields Psi2 keeps both halves).  The rounding errors are independent from element to element
                                 * and average over the observations / pairs a sum runs over: measured <= 3e-6 of the largest gradient
                                 * entry at N = 2000 (the default's 6e-7), beyond the 5e-4 gradient tolerance only for N of a few dozen
                                 * (tests/test_gpu_grad.py).  Opt-in; never the default. */
#define DPGP_PREC_MIXED_PATCH 3 /* dpgp_elbo_grad_psi[_ex] only: as DPGP_PREC_MIXED, the Psi2 term in the per-observation patch
                                 * form (psi2_grad_kernel) instead of the pair-tile form (psi2_pairs_grad.hip).  The pair-tile form
                                 * is faster (config 3: 6.7 vs 8.0 ms) but sums a' and b weighted exponentials SEPARATELY before
                                 * they cancel in 2 a s + b; where the adjoints themselves cancel heavily (K_uu nearly singular:
                                 * an fp64 forward pass with this stage in mixed precision) the patch form keeps 2e-3 of the
                                 * largest gradient entry, the pair form 7e-3 (tests/test_gpu_illcond.py) */

/* info[] codes besides the LAPACK-style positive ones (fused ELBO only; 0 = fine):
 *   DPGP_INFO_ILL_CONDITIONED: with an fp32 Psi2 (DPGP_PREC_MIXED / DPGP_PREC_F32) the rounding of Psi2, amplified by
 *   K_uu^-1, may move this output dim's terms by more than DPGP_GUARD_REL * N (bound written to the workspace's guard[d],
 *   see dpgp_elbo_workspace_layout): the terms are written but are not covered by the mixed-precision tolerance any more.
 *   Evaluate with DPGP_PREC_F64 (the reference's arithmetic, src/utils/types.py:13-14).
 *   Two paths use a LOOSER form of the bound (they flag earlier, never later): M > 128 with D >= 128 (chain_big.hip) takes
 *   tr K_uu^-1 for |K_uu^-1|_F (at most sqrt(M) larger); the over-T evaluation (dpgp_elbo_fhat_t) evaluates the guard of atom t with
 *   the data-fit factor of a stand-in column (column t of y) instead of the D columns the atom is solved against — there the flag
 *   says "this atom's K_uu / B is ill-conditioned for an fp32 Psi2", not a bound on a particular output dim's terms.
 * Operands outside the f16 range of the default fp32 psi kernels (|z - mean z| or |mu - mean z| beyond ~90 length scales)
 * are detected in the kernels: the affected Psi2 patch / Psi1^T y slab comes out as NaN, which the fused ELBO reports as a
 * failed factorisation (info > 0, NaN terms).  DPGP_ALGO_MFMA_F32 and fp64 have no such limit.                              */
#define DPGP_INFO_ILL_CONDITIONED (-2)
#define DPGP_GUARD_REL 2.0e-3

int dpgp_version(void);
/* text of the HIP error behind the last DPGP_ERR_LAUNCH returned to the calling thread (diagnostics) */
const char *dpgp_last_hip_error(void);

/* ---- Kernel.covariance_matrix (rbf_kernel.py:58-93): out[B,N0,N1] = alpha_b exp(-1/2 sum_q gamma_bq (x0_iq-x1_jq)^2)
 *      x1 == NULL means input_1 is None: N1 is ignored (= N0) and noise/jitter flags apply on the diagonal.          */
int dpgp_ard_rbf_gram_f32(int B, int N0, int N1, int Q, const float *x0, const float *x1, const float *gamma,
                          const float *alpha, const float *beta, int flags, double jitter, float *out, void *stream);
int dpgp_ard_rbf_gram_f64(int B, int N0, int N1, int Q, const double *x0, const double *x1, const double *gamma,
                          const double *alpha, const double *beta, int flags, double jitter, double *out, void *stream);

/* ---- Kernel.covariance_diag (rbf_kernel.py:96-116): out[B,N] = alpha_b (+1/beta_b) (+jitter) */
int dpgp_ard_rbf_diag_f32(int B, int N, const float *alpha, const float *beta, int flags, double jitter, float *out,
                          void *stream);
int dpgp_ard_rbf_diag_f64(int B, int N, const double *alpha, const double *beta, int flags, double jitter, double *out,
                          void *stream);

/* ---- Kernel.psi_0 (rbf_kernel.py:119-132): out[B] = alpha_b * N */
int dpgp_psi0_f32(int B, int N, const float *alpha, float *out, void *stream);
int dpgp_psi0_f64(int B, int N, const double *alpha, double *out, void *stream);

/* ---- Kernel.psi_1 (rbf_kernel.py:135-161): out[B,N,M] */
int dpgp_psi1_f32(int B, int N, int M, int Q, const float *z, const float *mu, const float *s, const float *gamma,
                  const float *alpha, float *out, void *stream);
int dpgp_psi1_f64(int B, int N, int M, int Q, const double *z, const double *mu, const double *s, const double *gamma,
                  const double *alpha, double *out, void *stream);

/* ---- Psi1_b^T y_b without materialising Psi1 (replaces the two [D,M,N] solves + [D,N,N] product of
 *      dp_gp_lvm.py:132-136,145): y is [N, ldy] with column b used for batch entry b; out[B,M].
 *      ws: dpgp_psi1T_y_workspace_bytes(B,N,M).                                                                   */
size_t dpgp_psi1T_y_workspace_bytes(int B, int N, int M);
int dpgp_psi1T_y_f32(int B, int N, int M, int Q, const float *z, const float *mu, const float *s, const float *gamma,
                     const float *alpha, const float *y, int ldy, float *out, void *ws, size_t ws_bytes, void *stream);
int dpgp_psi1T_y_f64(int B, int N, int M, int Q, const double *z, const double *mu, const double *s,
                     const double *gamma, const double *alpha, const double *y, int ldy, double *out, void *ws,
                     size_t ws_bytes, void *stream);

/* ---- Kernel.psi_2 (rbf_kernel.py:164-199): out[B,M,M], streamed over n (no [B,N,M,M,Q] temporary).
 *      ws: dpgp_psi2_workspace_bytes(B,N,M,Q,elem_size).                                                           */
size_t dpgp_psi2_workspace_bytes(int B, int N, int M, int Q, int elem_size);
int dpgp_psi2_f32(int B, int N, int M, int Q, const float *z, const float *mu, const float *s, const float *gamma,
                  const float *alpha, float *out, void *ws, size_t ws_bytes, int algo, void *stream);
int dpgp_psi2_f64(int B, int N, int M, int Q, const double *z, const double *mu, const double *s, const double *gamma,
                  const double *alpha, double *out, void *ws, size_t ws_bytes, int algo, void *stream);

/* ---- tf.cholesky (dp_gp_lvm.py:116,127): in-place lower Cholesky of a[B,M,M] (upper triangle zeroed), info[B].
 *      ws: dpgp_potrf_workspace_bytes(B,M,elem_size).                                                              */
size_t dpgp_potrf_workspace_bytes(int B, int M, int elem_size);
int dpgp_potrf_batched_f32(int B, int M, float *a, int *info, void *ws, size_t ws_bytes, int algo, void *stream);
int dpgp_potrf_batched_f64(int B, int M, double *a, int *info, void *ws, size_t ws_bytes, int algo, void *stream);

/* ---- tf.matrix_triangular_solve(lower=True) (dp_gp_lvm.py:118-121,132-133): rhs[B,M,K] <- l[B,M,M]^-1 rhs.
 *      ws: dpgp_trsm_workspace_bytes(B,M,K,elem_size).                                                             */
size_t dpgp_trsm_workspace_bytes(int B, int M, int K, int elem_size);
int dpgp_trsm_batched_f32(int B, int M, int K, const float *l, float *rhs, void *ws, size_t ws_bytes, int algo,
                          void *stream);
int dpgp_trsm_batched_f64(int B, int M, int K, const double *l, double *rhs, void *ws, size_t ws_bytes, int algo,
                          void *stream);
/* L^-1 of B lower-triangular factors l[B][M][M] (M a multiple of 128; -2 otherwise): out[B][M][M] = L^-1, lower, zeros above the
 * diagonal — tf.matrix_triangular_solve(l, eye) (the composed stage A of the backward pass for M > 128 forms K_uu^-1 = W^T W and
 * B^-1 from it) as one persistent-workgroup launch per batch (potrf_persist.hip).  ws: B doubles. */
int dpgp_trtri_lower_batched_f64(int B, int M, const double *l, double *out, void *ws, size_t ws_bytes, void *stream);

/* ---- strided batched matrix product, fp64, on the matrix cores:  C[b] = alpha A[b] B[b] + beta C[b],
 *      A[b][i][k] = a[b a_sb + i a_si + k a_sk],  B[b][k][j] = b[b b_sb + k b_sk + j b_sj],  C likewise (element strides:
 *      a transposed, sliced or batch-broadcast (stride 0) operand is a choice of strides).  m x k times k x n, batch >= 1.
 *      Replaces the tf.matmul call sites of the composed models: Psi1^T Y of the over-T objective
 *      (/root/reference/src/models/dp_gp_lvm.py:657-658), the prediction bound (dp_gp_lvm.py:300-420) and the adjoint
 *      algebra of their backward passes.  beta == 0: C is not read.  Returns -(argument index) on a bad argument. */
int dpgp_gemm_strided_f64(int batch, int m, int n, int k, double alpha, const double *a, long long a_sb, long long a_si,
                          long long a_sk, const double *b, long long b_sb, long long b_sk, long long b_sj, double beta,
                          double *c, long long c_sb, long long c_si, long long c_sj, void *stream);

/* ---- calculate_kl_divergence_standard_prior (gp_expressions.py:10-24): out[1] (fp64) */
int dpgp_kl_qx_f32(int N, int Q, const float *mu, const float *s, double *out, void *stream);
int dpgp_kl_qx_f64(int N, int Q, const double *mu, const double *s, double *out, void *stream);

/* ---- the fused per-output ELBO reduction, dp_gp_lvm.py:108-145, for D output dims resident on this GPU.
 *   inputs (all fp64 device arrays; they are rounded to fp32 on the fly where `prec` says so):
 *     y[N,ldy] (column d = output d), z[M,Q], mu[N,Q], s[N,Q], gamma[D,Q], alpha[D], beta[D]
 *   outputs (fp64): terms[D,5] = { N/2 (log beta - log 2pi), -sum log diag L_A, beta/2 (tr(L^-1 Psi2 L^-T) - alpha N),
 *                                  -beta/2 y'y, beta^2/2 |L_A^-1 L^-1 Psi1' y|^2 }   (f_hat = sum of all entries)
 *                   sums[2]    = { f_hat, KL(q(X)||p(X)) }
 *                   info[D]    (see top of file)
 *   ws: dpgp_elbo_workspace_bytes(D,N,M,Q,prec).  algo: DPGP_ALGO_*.                                               */
size_t dpgp_elbo_workspace_bytes(int D, int N, int M, int Q, int prec);
/* where a finished dpgp_elbo_fhat call left its streaming results inside ws: out[10] = { byte offset of the Psi2 partial slabs
 * [ns2][D][Mp][Mp] (lower 64x64 patches; their sum over the slabs is Psi2), ns2, element size of the slabs (4 or 8), Mp,
 * byte offset of the Psi1^T y partial slabs [ns1][D][M] (fp64), ns1, byte offset of the y^T y partial slabs [nyy][D] (fp64), nyy,
 * byte offset of guard[D] (fp64: bound on the effect of an fp32 Psi2's rounding on each output dim's terms), D }.
 * Used by the host-side composition of the backward pass's stage A for M > 128 (ops.elbo_grad_chain). */
int dpgp_elbo_workspace_layout(int D, int N, int M, int Q, int prec, size_t *out);
int dpgp_elbo_fhat(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                   const double *s, const double *gamma, const double *alpha, const double *beta, double jitter,
                   int prec, int algo, double *terms, double *sums, int *info, void *ws, size_t ws_bytes,
                   void *stream);
/* same, with optional extras (exec may be NULL = dpgp_elbo_fhat):
 *   ev_psi2_begin / ev_psi2_end : caller-created hipEvent_t recorded on `stream` immediately before / after the psi2
 *       launch so that a harness can time the dominant kernel inside its timed region; either may be NULL.
 *   model_scal / model_pack / model_out : fold the model-level tail into the last launch.  model_scal = the scal array
 *       written by dpgp_model_prepare for the same D; then model_pack[2] (if given) receives what dpgp_model_pack would
 *       write and model_out[5] (if given; single-GPU case) what dpgp_model_finalize would write.  NULL: not done.     */
typedef struct dpgp_exec {
    void *ev_psi2_begin;
    void *ev_psi2_end;
    const double *model_scal;
    double *model_pack;
    double *model_out;
    /* parallel branch (all three set, or none): the Psi1^T y launch — independent of the psi2 launch, both feed the chain —
     * goes to `stream_aux` between `ev_fork` (recorded on `stream` after the front launch) and `ev_join` (waited for by
     * `stream` before the chain launch).  Caller-created: dpgp_stream_create / dpgp_event_create.  Graph-capturable. */
    void *stream_aux;
    void *ev_fork;
    void *ev_join;
} dpgp_exec_t;
int dpgp_elbo_fhat_ex(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                      const double *s, const double *gamma, const double *alpha, const double *beta, double jitter,
                      int prec, int algo, double *terms, double *sums, int *info, void *ws, size_t ws_bytes,
                      void *stream, const dpgp_exec_t *exec);
/* Backward pass of the fused ELBO, stage A (first version; reference: tf.gradients(objective) as used by every training
 * script, test/synthetic_data_hard_test.py:143-155; the forward it differentiates: src/models/dp_gp_lvm.py:108-145).
 * Adjoints of the per-output dense algebra, computed from the workspace `ws` of a FINISHED dpgp_elbo_fhat[_ex] call with
 * the same D, N, M, Q, prec (prec = DPGP_PREC_MIXED or DPGP_PREC_F64; M <= 128: B_d is kept in LDS; -30 otherwise — for larger
 * M the host side composes this stage from dpgp_potrf_batched / dpgp_trsm_batched and plain GEMMs, ops.py).  Mp = 16 * ceil(M / 16).
 *   g_psi2[D][Mp][Mp]  d f_hat / d Psi2_d   (lower triangle j <= i valid, symmetric)
 *   w_kuu [D][Mp][Mp]  (d f_hat / d K_uu,d) .* (K_uu,d - jitter I)   (lower triangle)
 *   g_v   [D][Mp]      d f_hat / d (Psi1_d^T y_d)
 *   d_alpha_beta[D][2] d f_hat / d alpha_d (complete: direct + through K_uu, Psi1, Psi2) and d f_hat / d beta_d
 *   info[D]            0 or the failing minor of B_d = K_uu,d + beta_d Psi2_d                                              */
int dpgp_elbo_grad_chain(int D, int N, int M, int Q, const double *alpha, const double *beta, double jitter, int prec,
                         void *ws, size_t ws_bytes, double *g_psi2, double *w_kuu, double *g_v, double *d_alpha_beta,
                         int *info, void *stream);
/* Stage A for M > 128, M a multiple of 128 (-3 otherwise: the host side composes it from dpgp_potrf_batched / dpgp_trsm_batched /
 * dpgp_gemm_strided_f64 then): the same adjoints from the forward evaluation's workspace, on [D][M][M] fp64 matrices in memory — the
 * persistent-workgroup Cholesky and solve (L^-1 with every block row stored), five strided MFMA products and three streaming kernels
 * (csrc/chain_grad_big.hip).  K_uu is rebuilt from z, gamma, alpha (rbf_kernel.py:58-93).  psi2_slabs: 0 = the forward's own count of
 * Psi2 slabs, 1 after dpgp_elbo_fhat_step.  ws: dpgp_elbo_grad_chain_big_workspace_bytes(D, M) (six [D][M][M] matrices).  Outputs as
 * dpgp_elbo_grad_chain with Mp = M, both triangles written. */
size_t dpgp_elbo_grad_chain_big_workspace_bytes(int D, int M);
int dpgp_elbo_grad_chain_big(int D, int N, int M, int Q, const double *z, const double *gamma, const double *alpha,
                             const double *beta, double jitter, int prec, void *fwd_ws, size_t fwd_ws_bytes, int psi2_slabs, void *ws,
                             size_t ws_bytes, double *g_psi2, double *w_kuu, double *g_v, double *d_alpha_beta, int *info,
                             void *stream);

/* Backward pass, stage B: second streaming pass over the observations — the derivatives of
 * <g_psi2, Psi2> + <g_v, Psi1^T y> + <d f_hat / d K_uu, K_uu> (reference forward: src/kernels/rbf_kernel.py:58-199) with
 * respect to mu[N,Q], the diagonal q(X) variances s[N,Q], z[M,Q] and gamma[D,Q], given the stage-A adjoints.  alpha is a
 * constant factor here (its derivative is complete in d_alpha_beta of stage A).
 *   DPGP_PREC_MIXED: fp32 arithmetic, fp64 sums of the per-workgroup partial results, any M: the Psi2 term on the matrix pipe
 *     (psi2_grad_kernel: the forward's f16-split exponent tiles, a second MFMA product for the z-weighted column sums), the
 *     Psi1 term by two reduction-free kernels, the K_uu term without a pass over the observations;
 *   DPGP_PREC_F64: one plain kernel, M <= 128 (-30 otherwise).
 * M may exceed N (prediction evaluates few test points).  ws: dpgp_elbo_grad_psi_workspace_bytes_ex(D,N,M,Q,prec) — the
 * images and results of the pair-tile form (the bulk: ~1.6 GB at N=2000, D=512, M=128, Q=10) are only part of it for
 * DPGP_PREC_MIXED; dpgp_elbo_grad_psi_workspace_bytes(D,N,M,Q) = the largest of the three (works for every prec).            */
size_t dpgp_elbo_grad_psi_workspace_bytes(int D, int N, int M, int Q);
size_t dpgp_elbo_grad_psi_workspace_bytes_ex(int D, int N, int M, int Q, int prec);
int dpgp_elbo_grad_psi(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                       const double *s, const double *gamma, const double *alpha, const double *g_psi2,
                       const double *w_kuu, const double *g_v, int prec, void *ws, size_t ws_bytes, double *d_mu,
                       double *d_s, double *d_z, double *d_gamma, void *stream);
/* as dpgp_elbo_grad_psi with a full adjoint g_psi1[D][N][Mp] of Psi1 (may be NULL) in place of the rank-1 form
 * g_v[d][a] y[n][d]: then y and g_v may be NULL.  Mixed precision only.  (The over-T model, reference dp_gp_lvm.py:513-676,
 * couples every atom with all columns of y.) */
int dpgp_elbo_grad_psi_ex(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                          const double *s, const double *gamma, const double *alpha, const double *g_psi2,
                          const double *w_kuu, const double *g_v, const double *g_psi1, int prec, void *ws,
                          size_t ws_bytes, double *d_mu, double *d_s, double *d_z, double *d_gamma, void *stream);

/* ---- Training step: the f_hat terms AND their gradients with respect to mu, s, z, gamma (stage B) and alpha, beta (stage A) in
 * one call — what one Adam iteration of every training script of the reference evaluates (objective + tf.gradients,
 * test/synthetic_data_hard_test.py:143-155; forward src/models/dp_gp_lvm.py:108-145, src/kernels/rbf_kernel.py:135-199).
 * Mixed precision (prec = DPGP_PREC_MIXED, as of the three calls it replaces: dpgp_elbo_fhat_ex + dpgp_elbo_grad_chain +
 * dpgp_elbo_grad_psi; or DPGP_PREC_MIXED_FAST for stage B), M <= 128 (-3), Q <= 20 (-4).  Same results as the three calls within the mixed-precision tolerance; the
 * Psi2 statistic comes out of the first pass of stage B (same exponentials, constant feature), so its exponentials are
 * evaluated twice per step instead of three times — the minimum: Psi2 is needed before the adjoints exist, the observation-side sums
 * after; Psi1^T y likewise comes out of the Psi1 term's adjoint-free pass (its y-weighted constant feature).  exec (may be NULL): with
 * stream_aux / ev_fork / ev_join set and D <= 256 the K_uu branch and stage A run on the second stream beside the image build and
 * chain_b (few output dims leave most of the chip idle); the streams are joined again on every way out.
 *   terms / sums / info / ws: as dpgp_elbo_fhat (ws: dpgp_elbo_workspace_bytes(D,N,M,Q,DPGP_PREC_MIXED));
 *   g_psi2 / w_kuu / g_v / d_alpha_beta / info_grad: as dpgp_elbo_grad_chain (outputs, caller-allocated);
 *   gws: dpgp_elbo_grad_psi_workspace_bytes_ex(D,N,M,Q,DPGP_PREC_MIXED);  d_mu / d_s / d_z / d_gamma: as dpgp_elbo_grad_psi.   */
int dpgp_elbo_step(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu, const double *s,
                   const double *gamma, const double *alpha, const double *beta, double jitter, int prec, double *terms, double *sums,
                   int *info, void *ws, size_t ws_bytes, double *g_psi2, double *w_kuu, double *g_v, double *d_alpha_beta,
                   int *info_grad, void *gws, size_t gws_bytes, double *d_mu, double *d_s, double *d_z, double *d_gamma,
                   void *stream, const dpgp_exec_t *exec);

/* The two halves of dpgp_elbo_step as calls of their own, for M > 128 (stage A is then composed by the caller from dpgp_potrf_batched /
 * dpgp_trsm_batched / dpgp_gemm_strided_f64, as after dpgp_elbo_fhat_ex): dpgp_elbo_fhat_step = the forward evaluation with Psi2 out of
 * the first pass of stage B (any M; afterwards the forward workspace holds ONE Psi2 slab: slab count 1 instead of
 * dpgp_elbo_workspace_layout's), dpgp_elbo_grad_psi_step = the rest of stage B on the same two workspaces. */
int dpgp_elbo_fhat_step(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu, const double *s,
                        const double *gamma, const double *alpha, const double *beta, double jitter, double *terms, double *sums,
                        int *info, void *ws, size_t ws_bytes, void *gws, size_t gws_bytes, void *stream, const dpgp_exec_t *exec);
int dpgp_elbo_grad_psi_step(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu, const double *s,
                            const double *gamma, const double *alpha, const double *g_psi2, const double *w_kuu, const double *g_v,
                            int prec, void *ws, size_t ws_bytes, void *gws, size_t gws_bytes, double *d_mu, double *d_s, double *d_z,
                            double *d_gamma, void *stream);

/* ---- f_hat of the over-T model dp_gp_lvm_t (reference: src/models/dp_gp_lvm.py:608-676; the [T,M,N] x [N,D] contraction at
 * :657-658): T atoms with their own kernel hyper-parameters, every atom coupled to all D columns of y through phit[T,D].
 *   inputs (fp64 device arrays): y[N,ldy], yy[D] = column sums of y^2, z[M,Q], mu[N,Q], s[N,Q], gamma[T,Q], alpha[T], beta[T],
 *     phit: assignment probabilities phi_dt of the D output dims on this GPU, element strides below
 *   outputs: per_t[T] = N/2 log beta_t + beta_t/2 (tr(L^-1 Psi2 L^-T) - alpha_t N) - sum log diag L_A,t,
 *            quad[T,D] = beta_t^2 |L_A,t^-1 L^-1 Psi1_t^T y_d|^2,   sums[2] = { f_hat (of these D output dims), KL(q(X)||p(X)) },
 *            info[T] (as for dpgp_elbo_fhat; a failed atom makes f_hat NaN)
 *   prec: DPGP_PREC_MIXED (Psi2 on the fp32 matrix path) or DPGP_PREC_F64; Psi1, the contraction and the chain are fp64.
 *   M <= 128 and T <= D (-30 / -2 otherwise: the host composes the same value from the operators above).
 *   ws: dpgp_elbo_fhat_t_workspace_bytes(T,D,N,M,Q,prec).  Eleven launches, no host synchronisation.                     */
size_t dpgp_elbo_fhat_t_workspace_bytes(int T, int D, int N, int M, int Q, int prec);
int dpgp_elbo_fhat_t(int T, int D, int N, int M, int Q, const double *y, int ldy, const double *yy, const double *z,
                     const double *mu, const double *s, const double *gamma, const double *alpha, const double *beta,
                     const double *phit, long long phi_st, long long phi_sd, double jitter, int prec, double *per_t,
                     double *quad, double *sums, int *info, void *ws, size_t ws_bytes, void *stream,
                     const double *model_scal, double *model_pack, double *model_out, void *stream_aux, void *ev_fork,
                     void *ev_join);
/*   phit is read as phit[t * phi_st + d * phi_sd] (phi[D,T] of dpgp_model_prepare_t: phi_st = 1, phi_sd = T).
 *   model_scal / model_pack / model_out (each may be NULL): the model-level tail in the last launch, as dpgp_exec_t's fields
 *   of the same names — model_scal = scal of dpgp_model_prepare_t for the same D output dims.
 *   stream_aux / ev_fork / ev_join (all three or none; dpgp_stream_create / dpgp_event_create): Psi1 and the product Psi1^T Y run
 *   on stream_aux beside the fused reduction on the atoms — both are short latency-bound chains at small T.  Ignored while
 *   `stream` is being captured into a graph. */

/* hipEvent helpers for hosts without their own HIP binding */
void *dpgp_event_create(void);
void dpgp_event_destroy(void *event);
void *dpgp_stream_create(void);          /* a non-blocking hipStream_t (dpgp_exec_t.stream_aux) */
void dpgp_stream_destroy(void *stream);
float dpgp_event_elapsed_ms(void *begin, void *end);

/* ---- model-level glue of dp_gp_lvm(...).objective that is O(D T + N Q) (all fp64):
 *   dpgp_model_prepare : from the RAW variational parameters (softplus / softmax parameterisation of utils/types.py:40-57,
 *     dirichlet_process.py:39-59) of the D output dims resident on this GPU (global index d_offset + d; logits row
 *     (d_offset+d)/mask_size) compute  phi[D,T] (optional), gamma[D,Q] = phi gamma_atoms, alpha[D], beta[D]
 *     (dp_gp_lvm.py:100-102), s[N,Q] = softplus(s_raw), and scal[dpgp_model_scal_count(D)]:
 *       scal[0] = D-independent terms of the DP ELBO (dirichlet_process.py:68-77; 0 unless add_constants != 0, i.e. on
 *                 exactly one rank), scal[1] = hyper-prior log-likelihood of the atoms (dp_gp_lvm.py:96-98),
 *       scal[2..] = per-row-block partial sums of E[log p(Z|V)] + H[q(Z)] (dirichlet_process.py:64-66,75).
 *   dpgp_model_pack    : pack[0] = f_hat (from the fused ELBO's sums[0]), pack[1] = this GPU's share of the DP objective
 *                        (= -(sum of scal[0], scal[2..])): the 2-vector that is sum-all-reduced when D is sharded.
 *   dpgp_model_finalize: pack[2] (summed over GPUs), kl[1], hyper[1] ->
 *       out[5] = {objective (dp_gp_lvm.py:154), f_hat, KL, DP objective, hyper-prior}.                               */
int dpgp_model_scal_count(int D);
int dpgp_model_prepare(int D, int T, int Q, int N, int d_offset, int mask_size, const double *logits,
                       const double *gamma_atoms_raw, const double *alpha_atoms_raw, const double *beta_atoms_raw,
                       const double *s_raw, const double *g1_raw, const double *g2_raw, const double *w_raw, double s1,
                       double s2, int add_constants, double *gamma, double *alpha, double *beta, double *s, double *phi,
                       double *scal, void *stream);
/* dpgp_model_prepare for the over-T model: no mixing; outputs s[N,Q], phi[D,T], atoms[T*Q + 2 T] = softplus of the atoms
 * (gamma [T,Q] | alpha [T] | beta [T]) and scal as above. */
int dpgp_model_prepare_t(int D, int T, int Q, int N, int d_offset, int mask_size, const double *logits,
                         const double *gamma_atoms_raw, const double *alpha_atoms_raw, const double *beta_atoms_raw,
                         const double *s_raw, const double *g1_raw, const double *g2_raw, const double *w_raw, double s1,
                         double s2, int add_constants, double *s, double *phi, double *atoms, double *scal, void *stream);
/* The trouble flag of a gradient evaluation: out[0] = 1.0 if any info[0..d) != 0 (failed factorisation / conditioning guard of the
 * forward evaluation) or any of flat[0..n) is not finite, else 0.0 — the value the host-side optimiser loop branches on where the
 * reference's tf.cholesky raises (dp_gp_lvm.py:116,127), reduced over the ranks with the packed gradients.  One launch. */
int dpgp_trouble_flag(size_t n, const double *flat, int d, const int *info, double *out, void *stream);
/* Model-level backward pass (first version): d objective / d (the reference's eleven raw trainable variables,
 * dp_gp_lvm.py:63-94, dirichlet_process.py:40-59) from d f_hat / d (mu, S, z, gamma, alpha, beta) of dpgp_elbo_grad_chain +
 * dpgp_elbo_grad_psi, for the D output dims resident on this GPU.  phi[D,T]: as written by dpgp_model_prepare.  Outputs are
 * PARTIAL over this GPU's output dims, the D-independent terms being added iff add_constants (exactly one rank): a
 * sum-all-reduce of the outputs is the gradient.  d_logits[logits_rows][T] is zero outside the rows of the local dims.  */
int dpgp_model_backward(int D, int T, int Q, int N, int M, int d_offset, int mask_size, int logits_rows,
                        const double *logits, const double *gamma_atoms_raw, const double *alpha_atoms_raw,
                        const double *beta_atoms_raw, const double *s_raw, const double *g1_raw, const double *g2_raw,
                        const double *w_raw, const double *x_mean, const double *phi, double s1, double s2,
                        int add_constants, const double *df_dmu, const double *df_ds, const double *df_dz,
                        const double *df_dgamma, const double *df_dalpha_beta, double *d_x_mean, double *d_s_raw,
                        double *d_x_u, double *d_logits, double *d_g1_raw, double *d_g2_raw, double *d_w_raw,
                        double *d_gamma_atoms_raw, double *d_alpha_atoms_raw, double *d_beta_atoms_raw, void *stream);
/* the same for the over-T model dp_gp_lvm_t (reference src/models/dp_gp_lvm.py:513-676: the kernel batch is the T atoms, phi enters f_hat
 * directly): df_dgamma_atoms [T][Q] and df_dalpha_beta_atoms [T][2] are d f_hat / d (softplus'd atoms) themselves, df_dphi [D][T] is
 * d f_hat / d phi of the D output dims on this GPU.  Everything else as dpgp_model_backward. */
int dpgp_model_backward_t(int D, int T, int Q, int N, int M, int d_offset, int mask_size, int logits_rows, const double *logits,
                          const double *gamma_atoms_raw, const double *alpha_atoms_raw, const double *beta_atoms_raw,
                          const double *s_raw, const double *g1_raw, const double *g2_raw, const double *w_raw,
                          const double *x_mean, const double *phi, double s1, double s2, int add_constants, const double *df_dmu,
                          const double *df_ds, const double *df_dz, const double *df_dgamma_atoms,
                          const double *df_dalpha_beta_atoms, const double *df_dphi, double *d_x_mean, double *d_s_raw,
                          double *d_x_u, double *d_logits, double *d_g1_raw, double *d_g2_raw, double *d_w_raw,
                          double *d_gamma_atoms_raw, double *d_alpha_atoms_raw, double *d_beta_atoms_raw, void *stream);
int dpgp_model_pack(int D, const double *fhat, const double *scal, double *pack, void *stream);
int dpgp_model_finalize(const double *pack, const double *kl, const double *hyper, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DPGP_H */
