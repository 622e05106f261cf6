// Internal launch functions shared between the translation units of libdpgp_hip (not part of the C ABI).
#pragma once
#include "common.h"

// ---- elementwise.hip -------------------------------------------------------------------------------------------
// gram with explicit output leading dimension / batch stride so the fused ELBO can write straight into its padded
// Cholesky workspace.  x1 == nullptr: symmetric case, noise/jitter flags honoured.
template <typename TIN, typename T>
int launch_gram(int B, int N0, int N1, int Q, const TIN *x0, const TIN *x1, const TIN *gamma, const TIN *alpha,
                const TIN *beta, int flags, double jitter, T *out, int ld_out, size_t batch_stride, hipStream_t st);

// partial Psi1^T y slabs: part[ns][B][M] (fp64), ns = psi1T_y_nsplit(B, N, M)
int psi1T_y_nsplit(int B, int N, int M);
template <typename TIN, typename T>
int launch_psi1T_y_partial(int B, int N, int M, int Q, const TIN *z, const TIN *mu, const TIN *s, const TIN *gamma,
                           const TIN *alpha, const TIN *y, int ldy, double *part, int ns, unsigned char *psi2_consts,
                           int consts_ready, hipStream_t st);   // consts: as for launch_psi2_partial (f16 kernel only)

// yy_out: DPGP_YY_NCH partial slabs [DPGP_YY_NCH][D];  kl_out: DPGP_KL_NBLK partial sums (their total is 2 KL + N Q)
#define DPGP_YY_NCH 16
#define DPGP_KL_NBLK 16
template <typename TIN>
int launch_kl_yy(int N, int Q, const TIN *mu, const TIN *s, double *kl_out, int D, const TIN *y, int ldy,
                 double *yy_out, const TIN *z, int M, unsigned char *psi2_consts, hipStream_t st);
// the same roles plus the K_uu + jitter I tiles of all D output dims in ONE launch (the front launch of the fused ELBO)
template <typename TL>
int launch_elbo_front(int N, int Q, const double *mu, const double *s, double *kl_out, int D, const double *y, int ldy,
                      double *yy_out, const double *z, int M, unsigned char *psi2_consts, const double *gamma,
                      const double *alpha, const double *beta, double jitter, TL *kuu, int ld_kuu, size_t kuu_stride,
                      float *pair_scale, hipStream_t st);   // pair_scale != nullptr: also the pair-scale table (psi2_consts.h)
// psi2_consts != nullptr: the same launch also builds the z-only constants of the f16 psi2 kernel (psi2_consts.h) from
// z[M,Q] into psi2_consts (psi2_consts_bytes(M, Q) bytes)

// ---- psi2.hip --------------------------------------------------------------------------------------------------
// partial Psi2 slabs: part[ns][B][Mp][Mp] (type T), Mp = round_up(M,16); only the lower block-triangle (16x16 tiles,
// J <= I) is written; entries with row/col >= M are zero.  ns = psi2_nsplit(B, N, M).
int psi2_nsplit(int B, int N, int M);
// chain_ws != nullptr (matrix-core algos only): the launch carries an extra slice of B workgroups in front of the psi2
// workgroups that run the K_uu branch (chain_k_body, linalg_dev.h) on the per-output workspaces chain_ws (elements of
// chain_elem = 4 or 8 bytes) -> logdet_k[B], info_k[B].
// consts: psi2_consts_bytes(M, Q) bytes of workspace for the z-only constants of the f16 kernel; consts_ready != 0: they
// were already built on this stream (launch_kl_yy), otherwise the launch builds them first.
size_t psi2_consts_bytes(int M, int Q);
template <typename TIN> int launch_psi2_consts(const TIN *z, int M, int Q, unsigned char *consts, hipStream_t st);
template <typename TIN, typename T>
int launch_psi2_partial(int B, int N, int M, int Q, const TIN *z, const TIN *mu, const TIN *s, const TIN *gamma,
                        const TIN *alpha, T *part, int ns, int algo, hipStream_t st, void *chain_ws, int chain_elem,
                        double *logdet_k, int *info_k, unsigned char *consts, int consts_ready, float *pair_scale);
// pair_scale: psi2_pairs_scale_bytes(B, M) bytes of scratch for the pair-tile kernel (fp32 results, psi2_pairs.hip)
size_t psi2_pairs_scale_bytes(int B, int M);

// ---- linalg.hip ------------------------------------------------------------------------------------------------
// per-d workspace of the fused Cholesky chain, in elements of TL (layout: linalg.hip)
size_t la_chain_ws_elems(int M);
// true if the K_uu branch of an M x M problem with elem-byte elements runs LDS-resident (and may ride in the f16 psi2 dispatch)
bool la_chain_k_resident(int M, int elem);
// everything that depends on K_uu only (may overlap the psi2 kernel on another stream)
template <typename TL>
int launch_chain_k(int D, int M, TL *ws, double *logdet_k, int *info_k, int algo, hipStream_t st);
// everything after Psi2: B = K + beta Psi2, bordered Cholesky, the five f_hat terms per output dim
template <typename TP, typename TL>
int launch_chain_b(int D, int N, int M, const TP *psi2_part, int ns2, const double *v_part, int ns1,
                   const double *alpha, const double *beta, const double *yy_part, const double *logdet_k,
                   const int *info_k, double *terms, int *info, double *guard, TL *ws, int algo, hipStream_t st,
                   const double *kl_part = nullptr, double *sums = nullptr, const double *model_scal = nullptr,
                   double *model_pack = nullptr, double *model_out = nullptr, TL *lb_out = nullptr);
// lb_out != nullptr (LDS-resident sizes only, -30 otherwise): also the factors L_B as the LDS image of their lower tiles,
// lb_out[D][nb (nb + 1) / 2][16][17] (linalg_dev.h: TSZ, LDT)
// sums != nullptr: the workgroup that finishes last also runs the reduction of launch_sum_terms (same arguments; the arrival
// counter is the int behind kl_part[DPGP_KL_NBLK], zeroed by launch_elbo_front / launch_kl_yy on the same stream)
// sums[0] = sum of terms (f_hat); sums[1] = KL from the DPGP_KL_NBLK partials (kl_part may be null: sums[1] untouched)
//   model_scal (optional, see dpgp_model_prepare): also pack[0..1] = {f_hat, this GPU's DP-objective share} and, if out is
//   given, out[0..4] = {objective, f_hat, KL, DP objective, hyper-prior} (single-GPU finalisation) in the same launch
int launch_sum_terms(int D, const double *terms, const double *kl_part, double *sums, const double *model_scal,
                     double *model_pack, double *model_out, hipStream_t st);
// ---- potrf_big.hip: batched Cholesky spread over the whole GPU for matrices that do not fit one workgroup's LDS -------
// a[B][M][M] in/out (lower, zeros above), info[B], ws: potrf_big_ws_elems(B, M) elements of T
size_t potrf_big_ws_elems(int B, int M);
template <typename T> int launch_potrf_big(int B, int M, T *a, int *info, T *ws, hipStream_t st);
// ---- potrf_persist.hip: fp64, one persistent workgroup per matrix (Mw a multiple of 128, w[B][Mw][Mw] in place, lower + zeros)
bool potrf_persist_applicable(int B, int M, int elem_size);
int launch_potrf_persist(int B, int Mw, double *w, int *info, hipStream_t st, size_t wstride = 0);   // 0: Mw * Mw
// X = L^-1 R in place for lower-triangular R (zeros above the diagonal), same scheme; only nrm2[b * nstride] = |X_b|_F^2 is a
// result (X is scratch afterwards: its last 128 rows are not stored)
int launch_ptrsm_persist(int B, int Mw, const double *l, size_t lstride, double *x, size_t xstride, double *nrm2,
                         size_t nstride, hipStream_t st, int identity = 0);   // identity: R = I, x need not be initialised
// quad[t][d] = scale[t]^2 |L_t^-1 v_td|^2 for the D columns of V_t = sum of nsl slabs vp[ks v_ss + (t M + r) D + d]; M <= 128,
// lb: the lb_out image of launch_chain_b
int launch_tcols_quad(int T, int M, int D, const double *lb, const double *vp, int nsl, long long v_ss, const double *scale,
                      double *quad, hipStream_t st);
// gemm.hip: split-k form of dpgp_gemm_strided_f64 (slab ks of the product into c + ks c_ss, alpha = 1, beta = 0)
int launch_gemm_splitk_f64(int batch, int m, int n, int k, const double *a, long long a_sb, long long a_si, long long a_sk,
                           const double *b, long long b_sb, long long b_sk, long long b_sj, double *c, long long c_sb,
                           long long c_si, long long c_sj, int ksplit, long long c_ss, hipStream_t st);
// ---- chain_big.hip: the dense chain of the fused ELBO for M > 128 in fp64 (one persistent workgroup per output dim and step)
bool chain_big_applicable(int D, int M, int elem);
size_t chain_big_ws_elems(int M);                        // per output dim, doubles (layout: chain_big.hip)
int launch_chain_big_k(int D, int M, double *ws, int *info_k, hipStream_t st);     // after the front launch: everything on K_uu
template <typename TP>
int launch_chain_big_b(int D, int N, int M, const TP *psi2_part, int ns2, const double *v_part, int ns1, const double *alpha,
                       const double *beta, const double *yy_part, const int *info_k, double *terms, int *info, double *guard,
                       double *ws, hipStream_t st, const double *kl_part, double *sums, const double *model_scal,
                       double *model_pack, double *model_out);
// ---- grad.hip: backward pass (first version) ---------------------------------------------------------------------
// adjoints of the per-output dense algebra, after a forward evaluation on the same chain workspace ws:
//   GP[D][Mp][Mp] = df/dPsi2 (lower), WK[D][Mp][Mp] = (df/dK_uu) * (K_uu - jitter I) (lower), Gv[D][Mp] = df/d(Psi1^T y),
//   dab[D][2] = (df/dalpha_d, df/dbeta_d) complete, info[D]
template <typename TP>
int launch_chain_grad(int D, int N, int M, const TP *psi2_part, int ns2, const double *v_part, int ns1, const double *alpha,
                      const double *beta, const double *yy_part, double jitter, double *ws, double *GP, double *WK,
                      double *Gv, double *dab, int *info, hipStream_t st);
// second streaming pass: d f_hat / d mu [N,Q], d S [N,Q], d z [M,Q], d gamma [D,Q] from the stage-A adjoints (alpha is a
// constant factor here: its derivative is complete in stage A).  ws: psi_grad_ws_bytes(D, N, M, Q, nullptr) bytes.
size_t psi_grad_ws_bytes(int D, int N, int M, int Q, int *nsplit_out);
size_t psi_grad_ws_bytes_kuu(int D, int M, int Q);     // (K_uu term only: do_psi2 = 0 / launch_kuu_grad)
template <typename TC>
int launch_psi_grad(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu, const double *s,
                    const double *gamma, const double *alpha, const double *GP, const double *WK, const double *Gv,
                    double *ws, double *dmu, double *ds, double *dz, double *dgamma, int do_psi2, hipStream_t st);
// do_psi2 = 0: only the K_uu term (dz, dgamma written; dmu, ds untouched) -- the other two terms then come from:
// grad.hip: the Psi1 term (mixed precision): dmu, ds overwritten; dz, dgamma added to.  ws: psi1_grad_ws_elems doubles,
// stage: reduce_rows_stage_elems(max(N Q, M Q, D Q)) doubles, consts as below
size_t psi1_grad_ws_elems(int D, int N, int M, int Q);
int launch_psi1_grad(int D, int N, int M, int Q, const double *y, int ldy, const unsigned char *consts, const double *mu,
                     const double *s, const double *gamma, const double *alpha, const double *Gv, const double *G1,
                     double *ws, double *stage, double *dmu, double *ds, double *dz, double *dgamma, hipStream_t st);
// (G1 != nullptr: full adjoint [D][N][Mp] of Psi1 instead of the rank-1 form Gv[d][a] y[n][d])
// grad.hip: the K_uu term for any M (mixed precision): dz, dgamma overwritten.  ws: D M Q + ceil(M / 64) D Q doubles
int launch_kuu_grad(int D, int M, int Q, const unsigned char *consts, const double *gamma, const double *WK, double *ws,
                    double *stage, double *dz, double *dgamma, hipStream_t st);
size_t reduce_rows_stage_elems(size_t n);
template <typename TP>
int launch_reduce_rows(size_t n, size_t pitch, int nk, const TP *part, double *out, int accumulate, double *stage, hipStream_t st);
// psi2.hip: the Psi2 term of the same pass on the matrix pipe (mixed precision, Q <= 12), ADDED to the outputs of
// launch_psi_grad(..., do_psi2 = 0, ...); consts: psi2 constants of z (launch_psi2_consts), part: psi2_grad_part_elems doubles
bool psi2_grad_supported(int M, int Q);
size_t psi2_grad_part_elems(int B, int N, int M, int Q);
int launch_psi2_grad(int B, int N, int M, int Q, const unsigned char *consts, const double *mu, const double *s,
                     const double *gamma, const double *alpha, const double *GP, double *part, double *stage, double *dmu,
                     double *ds, double *dz, double *dgamma, hipStream_t st);
#define DPGP_LB_TILE_ELEMS (16 * 17)   // elements of one tile of the lb_out image (= TSZ of linalg_dev.h)
#define DPGP_PREP_ROWS 16   // output dims per row-block of dpgp_model_prepare (scal has 2 + ceil(D / 16) entries)

// ---- psi2_pairs_grad.hip: the Psi2 term of stage B in the pair-tile form (Q <= 10) ---------------------------------------
bool psi2_pgrad_supported(int M, int Q);
size_t psi2_pgrad_ws_bytes(int D, int N, int M, int Q);
// which: 1 = the part that does not depend on the adjoints (observation images, pass 1; psi2_part / scale != nullptr: Psi2 as a
// by-product into slab 0 of the forward's partial slabs), 2 = the rest (after part 1 on the same ws), 3 = both
// stage A for M > 128 (M a multiple of 128) on [D][M][M] matrices in memory (chain_grad_big.hip); ws: chain_grad_big_ws_bytes
bool chain_grad_big_supported(int M);
size_t chain_grad_big_ws_bytes(int D, int M);
template <typename TP>
int launch_chain_grad_big(int D, int N, int M, int Q, const double *z, const double *gamma, const double *alpha, const double *beta,
                          double jitter, const TP *psi2_part, int ns2, const double *v_part, int ns1, const double *yy_part,
                          unsigned char *ws, double *GP, double *WK, double *Gv, double *dab, int *info, hipStream_t st);
int launch_psi2_pgrad(int D, int N, int M, int Q, const unsigned char *consts, const double *z, const double *mu,
                      const double *s, const double *gamma, const double *alpha, const double *GP, unsigned char *ws,
                      double *stage, double *dmu, double *ds, double *dz, double *dgamma, hipStream_t st, int which = 3,
                      float *psi2_part = nullptr, const float *scale = nullptr, const double *y = nullptr, int ldy = 0,
                      const double *Gv = nullptr, int fast = 0, double *psi1v = nullptr);
// (y, Gv != nullptr: the Psi1 term with the rank-1 adjoint g_v[d][a] y[n][d] through the same passes: dmu, ds overwritten)
