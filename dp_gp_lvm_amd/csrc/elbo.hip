// Fused per-output ELBO reduction (the hot path of dp_gp_lvm.py:108-148): no host synchronisation, no allocation —
// graph-capturable.  Launch order on the main stream:
//   1. K_uu + jitter I                                        (gram_kernel)
//   2. KL(q(X)||p(X)) and y_d^T y_d partials                  (kl_yy_kernel)
//   3. Psi1_d^T y_d partial slabs                             (psi1T_y_f16_kernel)
//   4. Psi2_d partial slabs on the matrix cores, with the K_uu branch (chol(K_uu), log-det, K_uu^-1: D latency-bound
//      workgroups) as an extra task slice of the SAME dispatch  (psi2_f16_kernel + chain_k_body)
//   5. B = K + beta Psi2, bordered Cholesky, f_hat terms; in the workgroup that finishes last: f_hat = sum of terms,
//      KL = sum of partials [+ model-level tail]                (chain_b_kernel)
#include "internal.h"
#include "psi2_consts.h"

struct ElboLayout {
    size_t off_yy, off_v, off_p2, off_la, off_ld, off_ik, off_kl, off_pc, off_guard, off_sc, total;
    int ns1, ns2, Mp;
};

static ElboLayout elbo_layout(int D, int N, int M, int Q, int prec) {
    ElboLayout L;
    L.Mp = dpgp_round_up(M, 16);
    L.ns1 = psi1T_y_nsplit(D, N, M);
    L.ns2 = psi2_nsplit(D, N, M);
    const size_t sp = (prec == DPGP_PREC_F64) ? 8 : 4, sl = (prec == DPGP_PREC_F32) ? 4 : 8;
    size_t o = 0;
    L.off_yy = o; o += dpgp_align256(sizeof(double) * DPGP_YY_NCH * D);
    L.off_ld = o; o += dpgp_align256(sizeof(double) * D);
    L.off_ik = o; o += dpgp_align256(sizeof(int) * D);
    L.off_kl = o; o += dpgp_align256(sizeof(double) * DPGP_KL_NBLK + sizeof(int));   // + chain_b's arrival counter
    L.off_guard = o; o += dpgp_align256(sizeof(double) * D);
    L.off_pc = o; o += dpgp_align256(psi2_consts_bytes(M, Q));
    L.off_sc = o; o += (prec == DPGP_PREC_F64) ? 0 : psi2_pairs_scale_bytes(D, M);
    L.off_v = o;  o += dpgp_align256(sizeof(double) * (size_t)L.ns1 * D * M);
    L.off_p2 = o; o += dpgp_align256(sp * (size_t)L.ns2 * D * L.Mp * L.Mp);
    L.off_la = o; o += dpgp_align256(sl * (size_t)D * la_chain_ws_elems(M));
    L.total = o;
    return L;
}

extern "C" size_t dpgp_elbo_workspace_bytes(int D, int N, int M, int Q, int prec) {
    if (D <= 0 || N <= 0 || M <= 0 || Q <= 0 || prec < 0 || prec > 2) return 0;
    return elbo_layout(D, N, M, Q, prec).total;
}

// Where a finished dpgp_elbo_fhat call left its streaming results inside `ws` (for stage A of the backward pass composed on
// the host side when M > 128): out[10] = { byte offset of the Psi2 partial slabs [ns2][D][Mp][Mp] (lower patches), ns2, their
// element size (4 or 8), Mp, byte offset of the Psi1^T y partial slabs [ns1][D][M] (fp64), ns1, byte offset of the y^T y
// partial slabs [nyy][D] (fp64), nyy, byte offset of the conditioning guard values guard[D] (fp64; see dpgp.h), D }
extern "C" int dpgp_elbo_workspace_layout(int D, int N, int M, int Q, int prec, size_t *out) {
    if (D <= 0 || N <= 0 || M <= 0 || Q <= 0 || prec < 0 || prec > 2) return -1;
    if (!out) return -6;
    const ElboLayout L = elbo_layout(D, N, M, Q, prec);
    out[0] = L.off_p2; out[1] = (size_t)L.ns2; out[2] = (prec == DPGP_PREC_F64) ? 8 : 4; out[3] = (size_t)L.Mp;
    out[4] = L.off_v;  out[5] = (size_t)L.ns1; out[6] = L.off_yy; out[7] = DPGP_YY_NCH;
    out[8] = L.off_guard; out[9] = (size_t)D;
    return DPGP_OK;
}

// A forked second stream is joined on EVERY way out: an error return between fork and join would otherwise leave work in flight on
// it that still writes the caller's workspace (and, inside a stream capture, an unjoined stream invalidates the capture).
struct AuxJoin {
    hipStream_t st = nullptr, aux = nullptr;
    hipEvent_t ev = nullptr;
    bool armed = false;
    void arm(hipStream_t st_, hipStream_t aux_, hipEvent_t ev_) { st = st_; aux = aux_; ev = ev_; armed = true; }
    int join() {                                               // the regular join (the event has been recorded on aux)
        armed = false;
        return hipStreamWaitEvent(st, ev, 0) == hipSuccess ? DPGP_OK : DPGP_ERR_LAUNCH;
    }
    ~AuxJoin() {
        if (armed && hipEventRecord(ev, aux) == hipSuccess) (void)hipStreamWaitEvent(st, ev, 0);
    }
};

// A launch that needs what the last kernel of elbo_run (chain_b) needs and nothing of chain_b itself — stage A of a training step: it
// goes to the second stream beside chain_b where both fit on the chip at once (2 D workgroups on 2 x 256 slots).
struct SideLaunch {
    int (*fn)(void *ctx, hipStream_t st);
    void *ctx;
};

template <typename TP, typename TL>
static int elbo_run(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                    const double *s, const double *gamma, const double *alpha, const double *beta, double jitter,
                    int algo, double *terms, double *sums, int *info, unsigned char *ws, const ElboLayout &L,
                    hipStream_t st, const dpgp_exec_t *ex, TL *lb_out = nullptr, unsigned char *pgws = nullptr,
                    bool psi1_from_pass = false, const SideLaunch *side = nullptr, bool *side_done = nullptr) {
    double *yy = reinterpret_cast<double *>(ws + L.off_yy);
    double *ldk = reinterpret_cast<double *>(ws + L.off_ld);
    int *ik = reinterpret_cast<int *>(ws + L.off_ik);
    double *vpart = reinterpret_cast<double *>(ws + L.off_v);
    TP *p2 = reinterpret_cast<TP *>(ws + L.off_p2);
    TL *la = reinterpret_cast<TL *>(ws + L.off_la);
    double *klp = reinterpret_cast<double *>(ws + L.off_kl);
    unsigned char *pconst = ws + L.off_pc;
    hipEvent_t ev0 = ex ? (hipEvent_t)ex->ev_psi2_begin : nullptr, ev1 = ex ? (hipEvent_t)ex->ev_psi2_end : nullptr;
    int rc;
    // one front launch: KL, y'y, the z-only constants of the psi kernels (incl. the pair image), K_uu + jitter I and — for
    // the pair-tile psi2 kernel — its per-(output dim, pair) scale table
    const bool pairs_psi2 = (sizeof(TP) == 4 && algo == DPGP_ALGO_AUTO && psi2_pairs_ksteps(Q) <= 8);
    float *pscale = reinterpret_cast<float *>(ws + L.off_sc);
    if ((rc = launch_elbo_front<TL>(N, Q, mu, s, klp, D, y, ldy, yy, z, M, pconst, gamma, alpha, beta, jitter, la, L.Mp,
                                    la_chain_ws_elems(M), pairs_psi2 ? pscale : nullptr, st)))
        return rc;
    // Psi1^T y needs only the front launch and feeds only the chain: with a second stream in `exec` it runs beside the psi2
    // launch instead of in front of it (12 us of the per-GPU share at D = 64)
    hipStream_t aux = (ex && ex->stream_aux && ex->ev_fork && ex->ev_join) ? (hipStream_t)ex->stream_aux : nullptr;
    // A fork / join pair costs tens of microseconds of cross-stream hand-over: in a training step the second stream is used only with few
    // output dims (D <= 256: the K_uu branch and stage A are then launches that leave most of the chip idle — config 2: 1.142 -> 1.12 ms);
    // at configs 3 / 5 it cost more than it hid (5.53 -> 5.58 / 3.71 -> 3.82 ms)
    if (pgws != nullptr && psi1_from_pass && D > 256) aux = nullptr;
    AuxJoin aj;
    if (aux) {
        if (hipEventRecord((hipEvent_t)ex->ev_fork, st) != hipSuccess || hipStreamWaitEvent(aux, (hipEvent_t)ex->ev_fork, 0) != hipSuccess)
            return DPGP_ERR_LAUNCH;
        aj.arm(st, aux, (hipEvent_t)ex->ev_join);
    }
    // psi1_from_pass (dpgp_elbo_step): Psi1^T y comes out of the Psi1 term's adjoint-free pass of stage B (rows = observations, columns
    // = inducing points, y-weighted features: psi2_pairs_grad.hip, launch_psi1_front) as ONE slab — no Psi1^T y launch here
    const int ns1 = psi1_from_pass ? 1 : L.ns1;
    if (!psi1_from_pass &&
        (rc = launch_psi1T_y_partial<double, TP>(D, N, M, Q, z, mu, s, gamma, alpha, y, ldy, vpart, L.ns1, pconst, 1, aux ? aux : st)))
        return rc;
    // training step with a second stream: the K_uu branch (a launch of its own there, 0.12 ms of latency-bound Cholesky work at config
    // 3) follows Psi1^T y on it, beside the image build and the head of pass 1
    const bool k_on_aux = aux && pgws != nullptr && la_chain_k_resident(M, (int)sizeof(TL)) && algo != DPGP_ALGO_PLAIN;
    if (k_on_aux && (rc = launch_chain_k<TL>(D, M, la, ldk, ik, algo, aux))) return rc;
    if (aux && hipEventRecord((hipEvent_t)ex->ev_join, aux) != hipSuccess) return DPGP_ERR_LAUNCH;
    // the K_uu branch rides in the psi2 dispatch when it is LDS-resident (or the exact-MFMA psi2 kernel runs, which carries
    // both forms); otherwise it is a launch of its own ahead of psi2
    const bool f16_psi2 = (sizeof(TP) == 4 && algo != DPGP_ALGO_MFMA_F32);
    // pgws != nullptr (training step, dpgp_elbo_step): Psi2 comes out of pass 1 of the backward pass's stage B, which evaluates
    // the same exponentials and does not depend on the adjoints (psi2_pairs_grad.hip) — no psi2 dispatch here, the K_uu branch is
    // a launch of its own, and the chain reads ONE slab
    const bool step = pgws != nullptr;
    if (step && !(pairs_psi2 && psi2_pgrad_supported(M, Q))) return -30;
    const bool fused_k = !step && (algo != DPGP_ALGO_PLAIN) && (!f16_psi2 || la_chain_k_resident(M, (int)sizeof(TL))) &&
                         !getenv("DPGP_UNFUSED_K");      // (experiments only)
    // M > 128 in fp64 with a matrix per compute unit: the persistent-workgroup chain (chain_big.hip); its K_uu side runs here
    bool big = false;
    if constexpr (sizeof(TL) == 8) {
        big = (algo != DPGP_ALGO_PLAIN) && chain_big_applicable(D, M, 8);
        if (big && (rc = launch_chain_big_k(D, M, reinterpret_cast<double *>(la), ik, st))) return rc;
    }
    if (!big && !fused_k && !k_on_aux && (rc = launch_chain_k<TL>(D, M, la, ldk, ik, algo, st))) return rc;
    if (ev0 && hipEventRecord(ev0, st) != hipSuccess) return DPGP_ERR_LAUNCH;
    // psi2 on the matrix cores; the same dispatch carries, ahead of the psi2 workgroups, the D workgroups of the K_uu
    // branch (Cholesky, log-det, inverse of K_uu), which are latency-bound and overlap the psi2 work completely
    if (step) {
        if constexpr (sizeof(TP) == 4) {
            if ((rc = launch_psi2_pgrad(D, N, M, Q, pconst, z, mu, s, gamma, alpha, nullptr, pgws, nullptr, nullptr, nullptr, nullptr,
                                        nullptr, st, 1, reinterpret_cast<float *>(p2), pscale, psi1_from_pass ? y : nullptr, ldy, nullptr, 0,
                                        psi1_from_pass ? vpart : nullptr)))
                return rc;
        }
    } else if ((rc = launch_psi2_partial<double, TP>(D, N, M, Q, z, mu, s, gamma, alpha, p2, L.ns2, algo, st,
                                                     (fused_k && !big) ? (void *)la : nullptr, (int)sizeof(TL), ldk, ik, pconst,
                                                     pairs_psi2 ? 2 : 1, pscale)))
        return rc;
    if (ev1 && hipEventRecord(ev1, st) != hipSuccess) return DPGP_ERR_LAUNCH;
    if (aux && aj.join() != DPGP_OK) return DPGP_ERR_LAUNCH;
    // ... and, in the workgroup that finishes last, f_hat and KL; with the model-level pointers of exec also the packed pair /
    // the finished objective (round 2: a launch of its own, sum_terms_kernel)
    if (big && lb_out) return -30;
    if (side && aux && !big && 2 * D <= 512) {
        // fork again: the side launch on the second stream, chain_b on this one; this stream waits for the side launch behind chain_b
        if (hipEventRecord((hipEvent_t)ex->ev_fork, st) != hipSuccess || hipStreamWaitEvent(aux, (hipEvent_t)ex->ev_fork, 0) != hipSuccess)
            return DPGP_ERR_LAUNCH;
        aj.arm(st, aux, (hipEvent_t)ex->ev_join);
        if ((rc = side->fn(side->ctx, aux))) return rc;
        if (hipEventRecord((hipEvent_t)ex->ev_join, aux) != hipSuccess) return DPGP_ERR_LAUNCH;
        rc = launch_chain_b<TP, TL>(D, N, M, p2, step ? 1 : L.ns2, vpart, ns1, alpha, beta, yy, ldk, ik, terms, info,
                                    reinterpret_cast<double *>(ws + L.off_guard), la, algo, st, klp, sums,
                                    ex ? (const double *)ex->model_scal : nullptr, ex ? (double *)ex->model_pack : nullptr,
                                    ex ? (double *)ex->model_out : nullptr, lb_out);
        if (rc) return rc;
        if (side_done) *side_done = true;
        return aj.join();
    }
    if constexpr (sizeof(TL) == 8) {
        if (big)
            return launch_chain_big_b<TP>(D, N, M, p2, step ? 1 : L.ns2, vpart, ns1, alpha, beta, yy, ik, terms, info,
                                          reinterpret_cast<double *>(ws + L.off_guard), reinterpret_cast<double *>(la), st, klp,
                                          sums, ex ? (const double *)ex->model_scal : nullptr,
                                          ex ? (double *)ex->model_pack : nullptr, ex ? (double *)ex->model_out : nullptr);
    }
    return launch_chain_b<TP, TL>(D, N, M, p2, step ? 1 : L.ns2, vpart, ns1, alpha, beta, yy, ldk, ik, terms, info,
                                  reinterpret_cast<double *>(ws + L.off_guard), la, algo, st, klp, sums,
                                  ex ? (const double *)ex->model_scal : nullptr, ex ? (double *)ex->model_pack : nullptr,
                                  ex ? (double *)ex->model_out : nullptr, lb_out);
}

// Backward pass, stage A (grad.hip): adjoints of the per-output dense algebra from the workspace of a finished forward
// evaluation (same D, N, M, Q, prec; prec must be mixed or f64: the chain is fp64).
extern "C" int dpgp_elbo_grad_chain(int D, int N, int M, int Q, const double *alpha, const double *beta, double jitter,
                                    int prec, void *ws, size_t ws_bytes, double *g_psi2, double *w_kuu, double *g_v,
                                    double *d_alpha_beta, int *info, void *stream) {
    if (D <= 0) return -1;
    if (N <= 0) return -2;
    if (M <= 0) return -3;
    if (Q <= 0 || Q > DPGP_MAX_Q) return -4;
    if (!alpha) return -5;
    if (!beta) return -6;
    if (!(jitter >= 0.0)) return -7;
    if (prec != DPGP_PREC_MIXED && prec != DPGP_PREC_F64) return -8;
    if (!ws) return -9;
    const ElboLayout L = elbo_layout(D, N, M, Q, prec);
    if (ws_bytes < L.total) return -10;
    if (!g_psi2) return -11;
    if (!w_kuu) return -12;
    if (!g_v) return -13;
    if (!d_alpha_beta) return -14;
    if (!info) return -15;
    unsigned char *w = (unsigned char *)ws;
    const double *yy = reinterpret_cast<const double *>(w + L.off_yy), *vpart = reinterpret_cast<const double *>(w + L.off_v);
    double *la = reinterpret_cast<double *>(w + L.off_la);
    // the f16 psi2 kernel leaves ns2 slabs; (slab count as in elbo_run)
    if (prec == DPGP_PREC_MIXED)
        return launch_chain_grad<float>(D, N, M, reinterpret_cast<const float *>(w + L.off_p2), L.ns2, vpart, L.ns1, alpha,
                                        beta, yy, jitter, la, g_psi2, w_kuu, g_v, d_alpha_beta, info, (hipStream_t)stream);
    return launch_chain_grad<double>(D, N, M, reinterpret_cast<const double *>(w + L.off_p2), L.ns2, vpart, L.ns1, alpha,
                                     beta, yy, jitter, la, g_psi2, w_kuu, g_v, d_alpha_beta, info, (hipStream_t)stream);
}

// Stage A for M > 128 (M a multiple of 128; other M: the host composes it from the batched operators) from the workspace of a finished
// forward evaluation, as dpgp_elbo_grad_chain for M <= 128.  psi2_slabs: 0 = the forward's own slab count, 1 after dpgp_elbo_fhat_step.
extern "C" size_t dpgp_elbo_grad_chain_big_workspace_bytes(int D, int M) {
    return (D > 0 && M > 0) ? chain_grad_big_ws_bytes(D, M) : 0;
}
extern "C" int dpgp_elbo_grad_chain_big(int D, int N, int M, int Q, const double *z, const double *gamma, const double *alpha,
                                        const double *beta, double jitter, int prec, void *fwd_ws, size_t fwd_ws_bytes, int psi2_slabs,
                                        void *ws, size_t ws_bytes, double *g_psi2, double *w_kuu, double *g_v, double *d_alpha_beta,
                                        int *info, void *stream) {
    if (D <= 0) return -1;
    if (N <= 0) return -2;
    if (M <= 0 || !chain_grad_big_supported(M)) return -3;
    if (Q <= 0 || Q > DPGP_MAX_Q) return -4;
    if (!z) return -5;
    if (!gamma) return -6;
    if (!alpha) return -7;
    if (!beta) return -8;
    if (!(jitter >= 0.0)) return -9;
    if (prec != DPGP_PREC_MIXED && prec != DPGP_PREC_F64) return -10;
    if (!fwd_ws) return -11;
    const ElboLayout L = elbo_layout(D, N, M, Q, prec);
    if (fwd_ws_bytes < L.total) return -12;
    if (psi2_slabs < 0 || psi2_slabs > L.ns2) return -13;
    if (!ws) return -14;
    if (ws_bytes < chain_grad_big_ws_bytes(D, M)) return -15;
    if (!g_psi2) return -16;
    if (!w_kuu) return -17;
    if (!g_v) return -18;
    if (!d_alpha_beta) return -19;
    if (!info) return -20;
    const unsigned char *w = (const unsigned char *)fwd_ws;
    const double *yy = reinterpret_cast<const double *>(w + L.off_yy), *vpart = reinterpret_cast<const double *>(w + L.off_v);
    const int ns2 = psi2_slabs ? psi2_slabs : L.ns2;
    if (prec == DPGP_PREC_MIXED)
        return launch_chain_grad_big<float>(D, N, M, Q, z, gamma, alpha, beta, jitter, reinterpret_cast<const float *>(w + L.off_p2), ns2,
                                            vpart, L.ns1, yy, (unsigned char *)ws, g_psi2, w_kuu, g_v, d_alpha_beta, info, (hipStream_t)stream);
    return launch_chain_grad_big<double>(D, N, M, Q, z, gamma, alpha, beta, jitter, reinterpret_cast<const double *>(w + L.off_p2), ns2,
                                         vpart, L.ns1, yy, (unsigned char *)ws, g_psi2, w_kuu, g_v, d_alpha_beta, info, (hipStream_t)stream);
}

// Stage B in mixed precision.  Workspace: [plain kernel's partials | psi2 constants | patch-form partials | Psi1 term | reduction
// stage | pair-tile form].  fwd_consts / fwd_scale != nullptr (training step): the forward evaluation's constants and per-pair
// factors are used as they are, and part 1 of the pair-tile form (images of the observations, pass 1) has already run on this
// workspace (elbo_run with pgws).
struct GradPsiWs { unsigned char *consts; double *part, *ws1, *stage; unsigned char *pgws; };
// patch-form partials of the Psi2 term: only where that form can run (asked for, or no pair-tile form: Q > 20)
static bool grad_psi_needs_patch(int M, int Q, bool patch_form) {
    return patch_form || !psi2_pgrad_supported(M, Q) || getenv("DPGP_GRAD_PATCH");
}
static GradPsiWs grad_psi_ws(int D, int N, int M, int Q, unsigned char *ws, bool patch_form) {
    GradPsiWs W;
    const size_t mx = (size_t)Q * (N > D ? (N > M ? N : M) : (D > M ? D : M));
    W.consts = ws + dpgp_align256(psi_grad_ws_bytes_kuu(D, M, Q));
    W.part = reinterpret_cast<double *>(W.consts + dpgp_align256(psi2_consts_bytes(M, Q)));
    W.ws1 = reinterpret_cast<double *>((unsigned char *)W.part +
                                       (grad_psi_needs_patch(M, Q, patch_form) ? dpgp_align256(sizeof(double) * psi2_grad_part_elems(D, N, M, Q)) : 0));
    W.stage = reinterpret_cast<double *>((unsigned char *)W.ws1 + dpgp_align256(sizeof(double) * psi1_grad_ws_elems(D, N, M, Q)));
    W.pgws = reinterpret_cast<unsigned char *>(W.stage) + dpgp_align256(sizeof(double) * reduce_rows_stage_elems(mx));
    return W;
}
static int grad_psi_mixed(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu, const double *s,
                          const double *gamma, const double *alpha, const double *g_psi2, const double *w_kuu, const double *g_v,
                          const double *g_psi1, bool patch_form, unsigned char *ws, double *d_mu, double *d_s, double *d_z,
                          double *d_gamma, hipStream_t st, const unsigned char *fwd_consts, const float *fwd_scale,
                          const float *fwd_psi2 = nullptr, int w11 = 0, double *psi1_front_done = nullptr) {
    // K_uu term by the plain kernel (no pass over the observations), Psi1 by the reduction-free kernels, Psi2 (nearly all
    // of the work) on the matrix pipe -- where those apply; otherwise everything by the plain kernel
    const bool fast = psi2_grad_supported(M, Q) && !getenv("DPGP_GRAD_PLAIN");
    const bool big = dpgp_round_up(M, 16) > 128;             // the plain kernel holds one row of the M x M statistics per thread
    if (big && !fast) return -30;
    int rc = DPGP_OK;
    if (!big && !fast) {                                     // everything by the plain kernel (Q > DPGP_MAX_Q-type shapes, experiments)
        if (g_psi1) return -15;
        return launch_psi_grad<float>(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, g_psi2, w_kuu, g_v ? g_v : w_kuu, (double *)ws,
                                      d_mu, d_s, d_z, d_gamma, 1, st);
    }
    const GradPsiWs W = grad_psi_ws(D, N, M, Q, ws, patch_form);
    const unsigned char *consts = fwd_consts ? fwd_consts : W.consts;
    if (!fwd_consts && (rc = launch_psi2_consts<double>(z, M, Q, W.consts, st)) != DPGP_OK) return rc;
    // K_uu term, any M: kuu_grad_kernel, (64 rows, output dim) per workgroup (partials in the plain kernel's workspace).  For M <= 128 the
    // plain kernel's K_uu-only mode did this with one workgroup per output dim: 51 us at config 3
    rc = launch_kuu_grad(D, M, Q, consts, gamma, w_kuu, (double *)ws, W.stage, d_z, d_gamma, st);
    if (rc != DPGP_OK) return rc;
    // the Psi2 term: pair-tile form (psi2_pairs_grad.hip) where it exists, else the per-observation patch form; the Psi1 term: the
    // pair-tile form's passes on the diagonal pairs (rank-1 adjoint), else the reduction-free kernels of grad.hip
    const bool pair_form = psi2_pgrad_supported(M, Q) && !patch_form && !getenv("DPGP_GRAD_PATCH");      // (DPGP_GRAD_PATCH: experiments)
    const bool psi1_pairs = pair_form && !g_psi1 && y && g_v && !getenv("DPGP_PSI1_PLAIN");
    if (!psi1_pairs) {
        rc = launch_psi1_grad(D, N, M, Q, y, ldy, consts, mu, s, gamma, alpha, g_v, g_psi1, W.ws1, W.stage, d_mu, d_s, d_z, d_gamma, st);
        if (rc != DPGP_OK) return rc;
    }
    if (pair_form)
        return launch_psi2_pgrad(D, N, M, Q, consts, z, mu, s, gamma, alpha, g_psi2, W.pgws, W.stage, d_mu, d_s, d_z, d_gamma, st,
                                 fwd_consts ? 2 : 3, const_cast<float *>(fwd_psi2), fwd_scale, psi1_pairs ? y : nullptr, ldy,
                                 psi1_pairs ? g_v : nullptr, w11, (fwd_consts && psi1_pairs) ? psi1_front_done : nullptr);
    if (fwd_consts) return -30;
    return launch_psi2_grad(D, N, M, Q, consts, mu, s, gamma, alpha, g_psi2, W.part, W.stage, d_mu, d_s, d_z, d_gamma, st);
}

// Backward pass, stage B (grad.hip): the second streaming pass over the observations.
extern "C" size_t dpgp_elbo_grad_psi_workspace_bytes_ex(int D, int N, int M, int Q, int prec) {
    if (D <= 0 || N <= 0 || M <= 0 || Q <= 0) return 0;
    if (prec == DPGP_PREC_F64 || !psi2_grad_supported(M, Q) || getenv("DPGP_GRAD_PLAIN"))
        return dpgp_align256(psi_grad_ws_bytes(D, N, M, Q, nullptr));        // the plain kernel only (with its [D][N][Q] partials)
    // mixed precisions: the plain kernel carries the K_uu term only; then the psi2 constants, the patch-form partials (where that form
    // can run), the Psi1 term's partials, the reduction stage and the pair-tile form (Q <= 20: its images and the two passes' results —
    // not when the patch form is asked for)
    const bool patch = (prec == DPGP_PREC_MIXED_PATCH);
    const size_t mx = (size_t)Q * (N > D ? (N > M ? N : M) : (D > M ? D : M));
    size_t b = dpgp_align256(psi_grad_ws_bytes_kuu(D, M, Q)) + dpgp_align256(psi2_consts_bytes(M, Q)) +
               (grad_psi_needs_patch(M, Q, patch) ? dpgp_align256(sizeof(double) * psi2_grad_part_elems(D, N, M, Q)) : 0) +
               dpgp_align256(sizeof(double) * psi1_grad_ws_elems(D, N, M, Q)) + dpgp_align256(sizeof(double) * reduce_rows_stage_elems(mx));
    if (!patch) b += dpgp_align256(psi2_pgrad_ws_bytes(D, N, M, Q));
    return b;
}
extern "C" size_t dpgp_elbo_grad_psi_workspace_bytes(int D, int N, int M, int Q) {
    return dpgp_elbo_grad_psi_workspace_bytes_ex(D, N, M, Q, DPGP_PREC_MIXED);      // (the largest)
}
extern "C" int dpgp_elbo_grad_psi_ex(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                                     const double *s, const double *gamma, const double *alpha, const double *g_psi2,
                                     const double *w_kuu, const double *g_v, const double *g_psi1, int prec, void *ws,
                                     size_t ws_bytes, double *d_mu, double *d_s, double *d_z, double *d_gamma, void *stream);
extern "C" int dpgp_elbo_grad_psi(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                                  const double *s, const double *gamma, const double *alpha, const double *g_psi2,
                                  const double *w_kuu, const double *g_v, int prec, void *ws, size_t ws_bytes,
                                  double *d_mu, double *d_s, double *d_z, double *d_gamma, void *stream) {
    return dpgp_elbo_grad_psi_ex(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, g_psi2, w_kuu, g_v, nullptr, prec, ws, ws_bytes,
                                 d_mu, d_s, d_z, d_gamma, stream);
}
// g_psi1 != NULL: a full adjoint [D][N][Mp] of Psi1 replaces the rank-1 form g_v[d][a] y[n][d] (then y may be NULL and g_v is
// not read): the over-T model's data-fit term couples every atom with all columns of y.  Mixed precision only.
extern "C" int dpgp_elbo_grad_psi_ex(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                                     const double *s, const double *gamma, const double *alpha, const double *g_psi2,
                                     const double *w_kuu, const double *g_v, const double *g_psi1, int prec, void *ws,
                                     size_t ws_bytes, double *d_mu, double *d_s, double *d_z, double *d_gamma, void *stream) {
    if (D <= 0) return -1;
    if (N <= 0) return -2;
    if (M <= 0) return -3;
    if (Q <= 0 || Q > DPGP_MAX_Q) return -4;
    if (!y && !g_psi1) return -5;
    if (y && ldy < D) return -6;
    if (!z) return -7;
    if (!mu) return -8;
    if (!s) return -9;
    if (!gamma) return -10;
    if (!alpha) return -11;
    if (!g_psi2) return -12;
    if (!w_kuu) return -13;
    if (!g_v && !g_psi1) return -14;
    const bool patch_form = (prec == DPGP_PREC_MIXED_PATCH), fast = (prec == DPGP_PREC_MIXED_FAST);
    if (patch_form || fast) prec = DPGP_PREC_MIXED;
    if (prec != DPGP_PREC_MIXED && prec != DPGP_PREC_F64) return -15;
    if (g_psi1 && (prec != DPGP_PREC_MIXED || !psi2_grad_supported(M, Q))) return -15;
    if (!ws) return -16;
    if (ws_bytes < dpgp_elbo_grad_psi_workspace_bytes_ex(D, N, M, Q, patch_form ? DPGP_PREC_MIXED_PATCH : prec)) return -17;
    if (!d_mu) return -18;
    if (!d_s) return -19;
    if (!d_z) return -20;
    if (!d_gamma) return -21;
    if (prec == DPGP_PREC_MIXED)
        return grad_psi_mixed(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, g_psi2, w_kuu, g_v, g_psi1, patch_form, (unsigned char *)ws,
                              d_mu, d_s, d_z, d_gamma, (hipStream_t)stream, nullptr, nullptr, nullptr, fast ? 1 : 0);
    if (dpgp_round_up(M, 16) > 128) return -30;
    return launch_psi_grad<double>(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, g_psi2, w_kuu, g_v, (double *)ws, d_mu, d_s,
                                   d_z, d_gamma, 1, (hipStream_t)stream);
}

extern "C" int dpgp_elbo_fhat_ex(int D, int N, int M, int Q, const double *y, int ldy, const double *z,
                                 const double *mu, const double *s, const double *gamma, const double *alpha,
                                 const double *beta, double jitter, int prec, int algo, double *terms, double *sums,
                                 int *info, void *ws, size_t ws_bytes, void *stream, const dpgp_exec_t *exec) {
    if (D <= 0) return -1;
    if (N <= 0) return -2;
    if (M <= 0) return -3;                                  // (M > N is legal: prediction evaluates f_hat on few test points)
    if (Q <= 0 || Q > DPGP_MAX_Q) return -4;
    if (!y) return -5;
    if (ldy < D) return -6;
    if (!z) return -7;
    if (!mu) return -8;
    if (!s) return -9;
    if (!gamma) return -10;
    if (!alpha) return -11;
    if (!beta) return -12;
    if (!(jitter >= 0.0)) return -13;
    if (prec < 0 || prec > 2) return -14;
    if (algo < 0 || algo > DPGP_ALGO_PATCH_F16) return -15;
    if (!terms) return -16;
    if (!sums) return -17;
    if (!info) return -18;
    if (!ws) return -19;
    const ElboLayout L = elbo_layout(D, N, M, Q, prec);
    if (ws_bytes < L.total) return -20;
    hipStream_t st = (hipStream_t)stream;
    unsigned char *w = (unsigned char *)ws;
    switch (prec) {
    case DPGP_PREC_F32:
        return elbo_run<float, float>(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, beta, jitter, algo, terms, sums, info,
                                      w, L, st, exec);
    case DPGP_PREC_MIXED:
        return elbo_run<float, double>(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, beta, jitter, algo, terms, sums,
                                       info, w, L, st, exec);
    default:
        return elbo_run<double, double>(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, beta, jitter, algo, terms, sums,
                                        info, w, L, st, exec);
    }
}

extern "C" int dpgp_elbo_fhat(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                              const double *s, const double *gamma, const double *alpha, const double *beta,
                              double jitter, int prec, int algo, double *terms, double *sums, int *info, void *ws,
                              size_t ws_bytes, void *stream) {
    return dpgp_elbo_fhat_ex(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, beta, jitter, prec, algo, terms, sums, info, ws,
                             ws_bytes, stream, nullptr);
}

// ---- training step (mixed precision, M <= 128, Q <= 20): f_hat terms AND the stage-A / stage-B gradients of one evaluation.
// What one Adam iteration of the reference needs (forward + tf.gradients, test/synthetic_data_hard_test.py:143-155).  Compared
// with dpgp_elbo_fhat + dpgp_elbo_grad_chain + dpgp_elbo_grad_psi the exponentials of the Psi2 statistic are evaluated twice
// instead of three times: pass 1 of stage B (rows = observations, columns = pairs of inducing points) does not depend on the
// adjoints, holds the constant 1 among its features and therefore yields Psi2 itself — the forward's psi2 dispatch is dropped.
extern "C" int dpgp_elbo_step(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                              const double *s, const double *gamma, const double *alpha, const double *beta, double jitter,
                              int prec, double *terms, double *sums, int *info, void *ws, size_t ws_bytes, double *g_psi2, double *w_kuu,
                              double *g_v, double *d_alpha_beta, int *info_grad, void *gws, size_t gws_bytes, double *d_mu,
                              double *d_s, double *d_z, double *d_gamma, void *stream, const dpgp_exec_t *exec) {
    if (D <= 0) return -1;
    if (N <= 0) return -2;
    if (M <= 0 || dpgp_round_up(M, 16) > 128) return -3;
    if (Q <= 0 || Q > DPGP_MAX_Q || !psi2_pgrad_supported(M, Q)) return -4;
    if (!y) return -5;
    if (ldy < D) return -6;
    if (!z) return -7;
    if (!mu) return -8;
    if (!s) return -9;
    if (!gamma) return -10;
    if (!alpha) return -11;
    if (!beta) return -12;
    if (!(jitter >= 0.0)) return -13;
    if (prec != DPGP_PREC_MIXED && prec != DPGP_PREC_MIXED_FAST) return -14;
    if (!terms) return -15;
    if (!sums) return -16;
    if (!info) return -17;
    if (!ws) return -18;
    const ElboLayout L = elbo_layout(D, N, M, Q, DPGP_PREC_MIXED);
    if (ws_bytes < L.total) return -19;
    if (!g_psi2) return -20;
    if (!w_kuu) return -21;
    if (!g_v) return -22;
    if (!d_alpha_beta) return -23;
    if (!info_grad) return -24;
    if (!gws) return -25;
    if (gws_bytes < dpgp_elbo_grad_psi_workspace_bytes(D, N, M, Q)) return -26;
    if (!d_mu) return -27;
    if (!d_s) return -28;
    if (!d_z) return -29;
    if (!d_gamma) return -30;
    hipStream_t st = (hipStream_t)stream;
    unsigned char *w = (unsigned char *)ws;
    const GradPsiWs W = grad_psi_ws(D, N, M, Q, (unsigned char *)gws, false);
    // (DPGP_PSI1_PLAIN, an experiment switch of grad_psi_mixed, keeps the Psi1 term off the passes: then the forward's own Psi1^T y launch)
    const bool v_from_pass = !getenv("DPGP_PSI1_PLAIN") && !getenv("DPGP_GRAD_PATCH") && !getenv("DPGP_GRAD_PLAIN");
    // stage A on the ONE slab pass 1 left (and the one slab of Psi1^T y): it needs what chain_b needs and nothing of chain_b, so with a
    // second stream and few output dims (per-GPU shares of a sharded run: both launches fit on the chip at once) it runs beside chain_b
    struct StageA {
        int D, N, M, ns1;
        const unsigned char *w;
        const ElboLayout *L;
        const double *alpha, *beta;
        double jitter, *g_psi2, *w_kuu, *g_v, *dab;
        int *info;
    } sa = {D, N, M, v_from_pass ? 1 : L.ns1, w, &L, alpha, beta, jitter, g_psi2, w_kuu, g_v, d_alpha_beta, info_grad};
    auto stage_a = [](void *c, hipStream_t s_) -> int {
        const StageA &a = *static_cast<const StageA *>(c);
        return launch_chain_grad<float>(a.D, a.N, a.M, reinterpret_cast<const float *>(a.w + a.L->off_p2), 1,
                                        reinterpret_cast<const double *>(a.w + a.L->off_v), a.ns1, a.alpha, a.beta,
                                        reinterpret_cast<const double *>(a.w + a.L->off_yy), a.jitter,
                                        reinterpret_cast<double *>(const_cast<unsigned char *>(a.w) + a.L->off_la), a.g_psi2, a.w_kuu, a.g_v,
                                        a.dab, a.info, s_);
    };
    const SideLaunch side = {stage_a, &sa};
    bool side_done = false;
    int rc = elbo_run<float, double>(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, beta, jitter, DPGP_ALGO_AUTO, terms, sums, info, w,
                                     L, st, exec, nullptr, W.pgws, v_from_pass, &side, &side_done);
    if (rc != DPGP_OK) return rc;
    if (!side_done && (rc = stage_a(&sa, st)) != DPGP_OK) return rc;
    return grad_psi_mixed(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, g_psi2, w_kuu, g_v, nullptr, false, (unsigned char *)gws, d_mu,
                          d_s, d_z, d_gamma, st, w + L.off_pc, reinterpret_cast<const float *>(w + L.off_sc),
                          reinterpret_cast<const float *>(w + L.off_p2), prec == DPGP_PREC_MIXED_FAST ? 1 : 0,
                          v_from_pass ? reinterpret_cast<double *>(w + L.off_v) : nullptr);
}

// The two halves of the step as entry points of their own (M > 128: stage A between them is composed on the host side):
// dpgp_elbo_fhat_step = dpgp_elbo_fhat_ex in mixed precision with Psi2 out of pass 1 of stage B (gws: the stage-B workspace, which
// keeps that pass's results); afterwards the forward workspace holds ONE Psi2 slab (dpgp_elbo_workspace_layout's count does not apply).
extern "C" int dpgp_elbo_fhat_step(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                                   const double *s, const double *gamma, const double *alpha, const double *beta, double jitter,
                                   double *terms, double *sums, int *info, void *ws, size_t ws_bytes, void *gws, size_t gws_bytes,
                                   void *stream, const dpgp_exec_t *exec) {
    if (D <= 0) return -1;
    if (N <= 0) return -2;
    if (M <= 0) return -3;
    if (Q <= 0 || Q > DPGP_MAX_Q || !psi2_pgrad_supported(M, Q)) return -4;
    if (!y) return -5;
    if (ldy < D) return -6;
    if (!z) return -7;
    if (!mu) return -8;
    if (!s) return -9;
    if (!gamma) return -10;
    if (!alpha) return -11;
    if (!beta) return -12;
    if (!(jitter >= 0.0)) return -13;
    if (!terms) return -14;
    if (!sums) return -15;
    if (!info) return -16;
    if (!ws) return -17;
    const ElboLayout L = elbo_layout(D, N, M, Q, DPGP_PREC_MIXED);
    if (ws_bytes < L.total) return -18;
    if (!gws) return -19;
    if (gws_bytes < dpgp_elbo_grad_psi_workspace_bytes(D, N, M, Q)) return -20;
    const GradPsiWs W = grad_psi_ws(D, N, M, Q, (unsigned char *)gws, false);
    return elbo_run<float, double>(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, beta, jitter, DPGP_ALGO_AUTO, terms, sums, info,
                                   (unsigned char *)ws, L, (hipStream_t)stream, exec, nullptr, W.pgws);
}
// ... and the rest of stage B after dpgp_elbo_fhat_step on the same two workspaces (prec: DPGP_PREC_MIXED or DPGP_PREC_MIXED_FAST)
extern "C" int dpgp_elbo_grad_psi_step(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                                       const double *s, const double *gamma, const double *alpha, const double *g_psi2,
                                       const double *w_kuu, const double *g_v, int prec, void *ws, size_t ws_bytes, void *gws,
                                       size_t gws_bytes, double *d_mu, double *d_s, double *d_z, double *d_gamma, void *stream) {
    if (D <= 0) return -1;
    if (N <= 0) return -2;
    if (M <= 0) return -3;
    if (Q <= 0 || Q > DPGP_MAX_Q || !psi2_pgrad_supported(M, Q)) return -4;
    if (!y) return -5;
    if (ldy < D) return -6;
    if (!z) return -7;
    if (!mu) return -8;
    if (!s) return -9;
    if (!gamma) return -10;
    if (!alpha) return -11;
    if (!g_psi2) return -12;
    if (!w_kuu) return -13;
    if (!g_v) return -14;
    if (prec != DPGP_PREC_MIXED && prec != DPGP_PREC_MIXED_FAST) return -15;
    if (!ws) return -16;
    const ElboLayout L = elbo_layout(D, N, M, Q, DPGP_PREC_MIXED);
    if (ws_bytes < L.total) return -17;
    if (!gws) return -18;
    if (gws_bytes < dpgp_elbo_grad_psi_workspace_bytes(D, N, M, Q)) return -19;
    if (!d_mu) return -20;
    if (!d_s) return -21;
    if (!d_z) return -22;
    if (!d_gamma) return -23;
    unsigned char *w = (unsigned char *)ws;
    return grad_psi_mixed(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, g_psi2, w_kuu, g_v, nullptr, false, (unsigned char *)gws, d_mu,
                          d_s, d_z, d_gamma, (hipStream_t)stream, w + L.off_pc, reinterpret_cast<const float *>(w + L.off_sc),
                          reinterpret_cast<const float *>(w + L.off_p2), prec == DPGP_PREC_MIXED_FAST ? 1 : 0);
}

// ---- f_hat of the over-T model (dp_gp_lvm_t, reference dp_gp_lvm.py:608-676): the T atoms play the part of the output dims in
// the fused reduction above (terms 0-2 of atom t do not depend on y), and every atom is solved against ALL D columns:
//      f_hat = -N D / 2 log 2pi + sum_td phi_td ( per_t + 1/2 beta_t^2 |L_B,t^-1 Psi1_t^T y_d|^2 - 1/2 beta_t y_d^T y_d ),
//      per_t = N/2 log beta_t + beta_t / 2 (tr(L^-1 Psi2 L^-T) - alpha_t N) - sum log diag L_A,t
// Launches: the five of dpgp_elbo_fhat on T "output dims" (chain_b also exports L_B,t), Psi1 [T,N,M] (fp64), the split-k product
// Psi1_t^T Y (:657-658), the D-column solve + squared norms (potrf_persist.hip), one combining workgroup.  M <= 128.
struct ElboTLayout {
    ElboLayout L;
    size_t off_fused, off_terms, off_sums, off_lb, off_p1, off_vp, total;
    int ksplit, nlow;
};
static ElboTLayout elbo_t_layout(int T, int D, int N, int M, int Q, int prec) {
    ElboTLayout E;
    E.L = elbo_layout(T, N, M, Q, prec);
    const int Mp = E.L.Mp, nb = Mp / 16;
    E.nlow = nb * (nb + 1) / 2;
    const int tiles = dpgp_ceil_div(D, 64) * dpgp_ceil_div(M, 64) * T;
    int ks = 512 / (tiles > 0 ? tiles : 1);
    if (ks > 8) ks = 8;
    if (ks > N / 64) ks = N / 64;
    if (ks < 1) ks = 1;
    E.ksplit = ks;
    size_t o = 0;
    E.off_fused = o; o += dpgp_align256(E.L.total);
    E.off_terms = o; o += dpgp_align256(sizeof(double) * 5 * T);
    E.off_sums = o;  o += 256;
    E.off_lb = o;    o += dpgp_align256(sizeof(double) * (size_t)T * E.nlow * DPGP_LB_TILE_ELEMS);
    E.off_p1 = o;    o += dpgp_align256(sizeof(double) * (size_t)T * N * M);
    E.off_vp = o;    o += dpgp_align256(sizeof(double) * (size_t)ks * T * M * D);
    E.total = o;
    return E;
}
extern "C" size_t dpgp_elbo_fhat_t_workspace_bytes(int T, int D, int N, int M, int Q, int prec) {
    if (T <= 0 || D < T || N <= 0 || M <= 0 || Q <= 0 || prec < 1 || prec > 2) return 0;
    return elbo_t_layout(T, D, N, M, Q, prec).total;
}

__global__ __launch_bounds__(256) void fhat_t_combine_kernel(int T, int D, int N, const double *__restrict__ terms,
                                                             const double *__restrict__ phit, const double *__restrict__ quad,
                                                             const double *__restrict__ yy, const double *__restrict__ beta,
                                                             const double *__restrict__ sums_in, double *__restrict__ per_t,
                                                             double *__restrict__ sums, long long phi_st, long long phi_sd,
                                                             const double *__restrict__ model_scal,
                                                             double *__restrict__ model_pack, double *__restrict__ model_out) {
    __shared__ double scratch[8];
    const int t = threadIdx.x;
    double acc = 0.0;
    for (int e = t; e < T * D; e += 256) {
        const int a = e / D, d = e - a * D;
        const double *o = terms + 5 * a;
        const double pt = o[0] + 0.5 * N * DPGP_LOG_2PI + o[1] + o[2];
        acc += phit[a * phi_st + d * phi_sd] * (pt + 0.5 * quad[e] - 0.5 * beta[a] * yy[d]);
    }
    acc = block_sum(acc, scratch);
    if (t < T) {
        const double *o = terms + 5 * t;
        per_t[t] = o[0] + 0.5 * N * DPGP_LOG_2PI + o[1] + o[2];
    }
    if (t == 0) {
        const double fhat = acc - 0.5 * (double)N * (double)D * DPGP_LOG_2PI, kl = sums_in[1];
        sums[0] = fhat;
        sums[1] = kl;
        if (model_scal) {                                    // the model-level tail, as in sum_terms_body (linalg.hip)
            double a = model_scal[0];
            const int nrb = (D + DPGP_PREP_ROWS - 1) / DPGP_PREP_ROWS;
            for (int i = 0; i < nrb; ++i) a += model_scal[2 + i];
            if (model_pack) { model_pack[0] = fhat; model_pack[1] = -a; }
            if (model_out) {
                const double hyper = model_scal[1];
                model_out[0] = -a - (fhat - kl) - hyper;
                model_out[1] = fhat; model_out[2] = kl; model_out[3] = -a; model_out[4] = hyper;
            }
        }
    }
}

extern "C" int dpgp_elbo_fhat_t(int T, int D, int N, int M, int Q, const double *y, int ldy, const double *yy, const double *z,
                                const double *mu, const double *s, const double *gamma, const double *alpha, const double *beta,
                                const double *phit, long long phi_st, long long phi_sd, double jitter, int prec, double *per_t,
                                double *quad, double *sums, int *info, void *ws, size_t ws_bytes, void *stream,
                                const double *model_scal, double *model_pack, double *model_out, void *stream_aux,
                                void *ev_fork, void *ev_join) {
    if (T <= 0) return -1;
    if (D < T) return -2;                                   // (the model asserts truncation_level <= D, dp_gp_lvm.py:567)
    if (N <= 0) return -3;
    if (M <= 0) return -4;
    if (Q <= 0 || Q > DPGP_MAX_Q) return -5;
    if (!y) return -6;
    if (ldy < D) return -7;
    if (!yy) return -8;
    if (!z || !mu || !s) return -9;
    if (!gamma || !alpha || !beta) return -12;
    if (!phit) return -15;
    if (!(jitter >= 0.0)) return -16;
    if (prec != DPGP_PREC_MIXED && prec != DPGP_PREC_F64) return -17;
    if (!per_t || !quad || !sums || !info) return -18;
    if (!ws) return -22;
    if (dpgp_round_up(M, 16) > 128) return -30;            // first version: L_B,t comes out of the LDS-resident chain
    const ElboTLayout E = elbo_t_layout(T, D, N, M, Q, prec);
    if (ws_bytes < E.total) return -23;
    hipStream_t st = (hipStream_t)stream;
    unsigned char *w = (unsigned char *)ws;
    double *terms = reinterpret_cast<double *>(w + E.off_terms), *sums2 = reinterpret_cast<double *>(w + E.off_sums);
    double *lb = reinterpret_cast<double *>(w + E.off_lb), *p1 = reinterpret_cast<double *>(w + E.off_p1);
    double *vp = reinterpret_cast<double *>(w + E.off_vp);
    int rc;
    // Psi1 and the product Psi1^T Y need only the inputs; the fused reduction on the atoms needs neither: with a second stream
    // (and two events) they run side by side — both are latency-bound at T = 8 (84 us against 111 us at config 3).  Not while
    // `stream` is being captured into a graph (the replay keeps the serial order).
    hipStream_t aux = (stream_aux && ev_fork && ev_join) ? (hipStream_t)stream_aux : nullptr;
    if (aux) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) aux = nullptr;
    }
    if (aux && (hipEventRecord((hipEvent_t)ev_fork, st) != hipSuccess || hipStreamWaitEvent(aux, (hipEvent_t)ev_fork, 0) != hipSuccess))
        return DPGP_ERR_LAUNCH;
    AuxJoin aj;
    if (aux) aj.arm(st, aux, (hipEvent_t)ev_join);
    hipStream_t sb = aux ? aux : st;
    if (aux) {                                                 // the branch first, so that both start at once
        if ((rc = dpgp_psi1_f64(T, N, M, Q, z, mu, s, gamma, alpha, p1, (void *)sb))) return rc - 100;
        if ((rc = launch_gemm_splitk_f64(T, M, D, N, p1, (long long)N * M, 1, M, y, 0, ldy, 1, vp, (long long)M * D, D, 1, E.ksplit,
                                         (long long)T * M * D, sb)))
            return rc - 200;
        if (hipEventRecord((hipEvent_t)ev_join, aux) != hipSuccess) return DPGP_ERR_LAUNCH;
    }
    // (the first T columns of y stand in for the per-output columns of the fused reduction: its terms 3 and 4 are not used)
    if (prec == DPGP_PREC_MIXED)
        rc = elbo_run<float, double>(T, N, M, Q, y, ldy, z, mu, s, gamma, alpha, beta, jitter, DPGP_ALGO_AUTO, terms, sums2, info,
                                     w + E.off_fused, E.L, st, nullptr, lb);
    else
        rc = elbo_run<double, double>(T, N, M, Q, y, ldy, z, mu, s, gamma, alpha, beta, jitter, DPGP_ALGO_AUTO, terms, sums2, info,
                                      w + E.off_fused, E.L, st, nullptr, lb);
    if (rc) return rc;
    if (!aux) {
        if ((rc = dpgp_psi1_f64(T, N, M, Q, z, mu, s, gamma, alpha, p1, stream))) return rc - 100;
        // V_t[m][d] = sum_n Psi1_t[n][m] y[n][d]
        if ((rc = launch_gemm_splitk_f64(T, M, D, N, p1, (long long)N * M, 1, M, y, 0, ldy, 1, vp, (long long)M * D, D, 1, E.ksplit,
                                         (long long)T * M * D, st)))
            return rc - 200;
    } else if (aj.join() != DPGP_OK) {
        return DPGP_ERR_LAUNCH;
    }
    if ((rc = launch_tcols_quad(T, M, D, lb, vp, E.ksplit, (long long)T * M * D, beta, quad, st))) return rc;
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(fhat_t_combine_kernel, dim3(1), dim3(256), 0, st, T, D, N, terms, phit, quad, yy, beta, sums2, per_t, sums,
                       phi_st, phi_sd, model_scal, model_pack, model_out);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

// HIP events for callers that have no HIP runtime binding of their own (bench.py times the psi2 kernel with these).
extern "C" void *dpgp_event_create(void) {
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? (void *)e : nullptr;
}
extern "C" void dpgp_event_destroy(void *e) {
    if (e) (void)hipEventDestroy((hipEvent_t)e);
}
extern "C" void *dpgp_stream_create(void) {
    hipStream_t s = nullptr;
    return hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess ? (void *)s : nullptr;
}
extern "C" void dpgp_stream_destroy(void *s) {
    if (s) (void)hipStreamDestroy((hipStream_t)s);
}
extern "C" float dpgp_event_elapsed_ms(void *a, void *b) {
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b) != hipSuccess) return -1.0f;
    return ms;
}

// ---------------------------------------------------------------------------------------------------------------
// Model-level glue of dp_gp_lvm(...).objective that is O(D T + N Q): variational-parameter transforms, soft-assignment
// mixing (dp_gp_lvm.py:100-102), the DP objective (dirichlet_process.py:39-88) and the hyper-prior on the atoms
// (dp_gp_lvm.py:96-98), so that one objective evaluation is a fixed handful of launches with no host arithmetic.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double softplus_d(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))); }

// digamma for x > 0: recurrence up to x >= 10, then the asymptotic series (error < 1e-15 there)
__device__ double digamma_d(double x) {
    double r = 0.0;
    while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
    const double f = 1.0 / (x * x);
    const double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 + f * (1.0 / 240.0 + f * (-1.0 / 132.0 +
                     f * (691.0 / 32760.0 + f * (-1.0 / 12.0)))))));
    return r + log(x) - 0.5 / x + t;
}

#define PREP_MAX_T 64
#define PREP_ROWS DPGP_PREP_ROWS   // output dims per row-block

// grid: block 0 = atoms, hyper-prior and the D-independent DP terms; blocks 1..nrb = PREP_ROWS output dims each (thread per
// (d, column)); remaining blocks = softplus of the q(X) variances.
//   scal[0] = D-independent DP terms (0 unless add_constants), scal[1] = hyper-prior, scal[2 + rb] = row-block partials.
__global__ __launch_bounds__(256) void model_prepare_kernel(
    int D, int T, int Q, int N, int d_offset, int mask_size, int nrb, const double *__restrict__ logits,
    const double *__restrict__ gat_raw, const double *__restrict__ aat_raw, const double *__restrict__ bat_raw,
    const double *__restrict__ s_raw, const double *__restrict__ g1_raw, const double *__restrict__ g2_raw,
    const double *__restrict__ w_raw, double s1, double s2, int add_constants, double *__restrict__ gamma,
    double *__restrict__ alpha, double *__restrict__ beta, double *__restrict__ s_out, double *__restrict__ phi_out,
    double *__restrict__ scal, double *__restrict__ atoms_out) {
    // (atoms_out, optional: softplus of the atoms, [T*Q gamma | T alpha | T beta] — the over-T model works on the atoms themselves
    //  and passes gamma = alpha = beta = nullptr)
    const int t = threadIdx.x;
    __shared__ double scratch[8];
    if ((int)blockIdx.x > nrb) {   // q(X) variances: s = softplus(raw)   (dp_gp_lvm.py:67-69, utils/types.py:40-57)
        const size_t tot = (size_t)N * Q;
        const int nsb = gridDim.x - 1 - nrb;
        for (size_t i = (size_t)(blockIdx.x - 1 - nrb) * 256 + t; i < tot; i += (size_t)nsb * 256)
            s_out[i] = softplus_d(s_raw[i]);
        return;
    }
    if (blockIdx.x == 0) {
        double hyper = 0.0, consts = 0.0;
        for (int i = t; i < T * Q + 2 * T; i += 256) {
            const double raw = i < T * Q ? gat_raw[i] : (i < T * Q + T ? aat_raw[i - T * Q] : bat_raw[i - T * Q - T]);
            const double sp = softplus_d(raw), lx = log(sp);
            if (atoms_out) atoms_out[i] = sp;
            hyper += -lx - 0.5 * (DPGP_LOG_2PI + lx * lx);        // log_normal.log_pdf (log_normal.py:34-39)
        }
        const double w1 = softplus_d(w_raw[0]), w2 = softplus_d(w_raw[1]);
        // The special functions of the stick parameters, ONE per thread (thread = (stick k, function f)): the fp64 lgamma /
        // digamma of the device library are hundreds of instructions each; six of them back to back on the threads k < T - 1
        // (plus three more on thread 0) were 10 of this launch's 16 us — the whole evaluation waits for this block.
        __shared__ double sf[6][PREP_MAX_T], sx[3];
        for (int i = t; i < 6 * (T - 1) + 3; i += 256) {
            if (i < 6 * (T - 1)) {
                const int k = i / 6, f = i - 6 * k;
                const double g1 = softplus_d(g1_raw[k]), g2 = softplus_d(g2_raw[k]);
                const double x = (f % 3 == 0) ? g1 : ((f % 3 == 1) ? g2 : g1 + g2);
                sf[f][k] = f < 3 ? lgamma(x) : digamma_d(x);
            } else {
                const int f = i - 6 * (T - 1);
                sx[f] = f == 0 ? digamma_d(w1) : (f == 1 ? lgamma(s1) : lgamma(w1));
            }
        }
        __syncthreads();
        if (t < T - 1) {
            const double g1 = softplus_d(g1_raw[t]), g2 = softplus_d(g2_raw[t]);
            const double p1 = sf[3][t], p2 = sf[4][t], p12 = sf[5][t];
            // per stick t: part of E[log p(V|alpha)] and the Beta entropy
            consts += (w1 / w2 - 1.0) * (p2 - p12) +
                      (sf[0][t] + sf[1][t] - sf[2][t] - (g1 - 1.0) * p1 - (g2 - 1.0) * p2 + (g1 + g2 - 2.0) * p12);
        }
        if (t == 0) {
            const double pw = sx[0], lw2 = log(w2);
            consts += (T - 1.0) * (pw - lw2)                                                   // rest of E[log p(V|alpha)]
                      + s1 * log(s2) - sx[1] + (s1 - 1.0) * (pw - lw2) - s2 * (w1 / w2)        // E[log p(alpha)]
                      + w1 - lw2 + sx[2] + (1.0 - w1) * pw;                                     // Gamma entropy
        }
        hyper = block_sum(hyper, scratch);
        consts = block_sum(consts, scratch);
        if (t == 0) {
            scal[0] = add_constants ? consts : 0.0;
            scal[1] = hyper;
        }
        return;
    }
    // ---- row blocks: thread = (output dim d, column j); j < Q: gamma_dj, j == Q: alpha, beta, phi row and the DP terms ----
    __shared__ double gat[PREP_MAX_T * DPGP_MAX_Q], aat[PREP_MAX_T], bat[PREP_MAX_T];
    __shared__ double c1[PREP_MAX_T], c2[PREP_MAX_T];   // psi(g1)-psi(g1+g2), psi(g2)-psi(g1+g2)
    for (int i = t; i < T * Q; i += 256) gat[i] = softplus_d(gat_raw[i]);
    if (t < T) {
        aat[t] = softplus_d(aat_raw[t]);
        bat[t] = softplus_d(bat_raw[t]);
    }
    // (one digamma per thread — thread = (stick k, argument f) — instead of three in a row on the threads k < T - 1: every row
    //  block waits for these before its first output dim)
    __shared__ double dg[3][PREP_MAX_T];
    if (t >= 64 && t - 64 < 3 * (T - 1)) {
        const int k = (t - 64) / 3, f = (t - 64) - 3 * k;
        const double g1 = softplus_d(g1_raw[k]), g2 = softplus_d(g2_raw[k]);
        dg[f][k] = digamma_d(f == 0 ? g1 : (f == 1 ? g2 : g1 + g2));
    }
    __syncthreads();
    if (t < T - 1) {
        c1[t] = dg[0][t] - dg[2][t];
        c2[t] = dg[1][t] - dg[2][t];
    }
    __syncthreads();
    const int rb = blockIdx.x - 1, cols = Q + 1;
    double dsum = 0.0;
    for (int e = t; e < PREP_ROWS * cols; e += 256) {
        const int rr = e / cols, j = e - rr * cols, d = rb * PREP_ROWS + rr;
        if (d >= D) continue;
        const double *lr = logits + (size_t)((d_offset + d) / mask_size) * T;
        double mx = lr[0];
        for (int k = 1; k < T; ++k) mx = fmax(mx, lr[k]);
        double zsum = 0.0;
        for (int k = 0; k < T; ++k) zsum += exp(lr[k] - mx);
        const double lz = log(zsum);
        if (j < Q) {                                                     // mixing (dp_gp_lvm.py:100)
            double g = 0.0;
            if (!gamma) continue;
            for (int k = 0; k < T; ++k) g += exp(lr[k] - mx - lz) * gat[k * Q + j];
            gamma[(size_t)d * Q + j] = g;
        } else {
            double al = 0.0, be = 0.0, ent = 0.0, ev = 0.0, tail = 0.0;
            for (int k = T - 1; k >= 0; --k) {
                const double lp = lr[k] - mx - lz, p = exp(lp);        // phi_dt = softmax (dirichlet_process.py:40-42)
                if (phi_out) phi_out[(size_t)d * T + k] = p;
                al += p * aat[k];
                be += p * bat[k];
                ent -= p * lp;                                           // entropy of q(Z) (:75)
                if (k < T - 1) ev += p * c1[k] + tail * c2[k];           // E[log p(Z|V)] (:64-66); tail = sum_{j>k} phi_dj
                tail += p;
            }
            if (alpha) alpha[d] = al;                                    // (dp_gp_lvm.py:101-102)
            if (beta) beta[d] = be;
            dsum += ev + ent;
        }
    }
    dsum = block_sum(dsum, scratch);
    if (t == 0) scal[2 + rb] = dsum;
}

// objective = DP objective - (f_hat - KL) - hyper-prior   (dp_gp_lvm.py:151-154)
//   fhat / dp: device scalars (already summed over GPUs when D is sharded)
__global__ void model_finalize_kernel(const double *__restrict__ fhat, const double *__restrict__ dp,
                                      const double *__restrict__ kl, const double *__restrict__ hyper,
                                      double *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = dp[0] - (fhat[0] - kl[0]) - hyper[0];
        out[1] = fhat[0];
        out[2] = kl[0];
        out[3] = dp[0];
        out[4] = hyper[0];
    }
}

// pack[0] = f_hat, pack[1] = -(constants + sum of the row-block partials) = this GPU's share of the DP objective
__global__ void model_pack_kernel(const double *__restrict__ fhat, const double *__restrict__ scal, int nrb,
                                  double *__restrict__ pack) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double a = scal[0];
        for (int i = 0; i < nrb; ++i) a += scal[2 + i];
        pack[0] = fhat[0];
        pack[1] = -a;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Model-level backward pass (stage C, first version): d objective / d (the reference's eleven raw trainable variables)
// from d f_hat / d (mu, S, z, gamma, alpha, beta) of stages A + B.  objective = DP objective - (f_hat - KL) - hyper-prior
// (dp_gp_lvm.py:148-154) with gamma = phi gamma_atoms etc. (:100-102), phi = softmax(logits) (dirichlet_process.py:40-51),
// S, atoms, q(V), q(alpha) parameters = softplus(raw) (utils/types.py:40-72).  Outputs are this GPU's PARTIAL sums over its
// output dims; the D-independent terms (KL, hyper-prior, E[log p(V|alpha)], E[log p(alpha)], entropies of q(V), q(alpha))
// are added iff add_constants (exactly one rank), so that a sum-all-reduce of the outputs gives the gradient.
//   block 0            : atoms, q(V), q(alpha)           (thread = (stick t, column j), loops over the local output dims)
//   blocks 1..nlb      : logits rows (thread = one mask group of output dims)
//   remaining blocks   : x_mean, x_var_raw, x_u element-wise
// ---------------------------------------------------------------------------------------------------------------
__device__ double trigamma_d(double x) {
    double r = 0.0;
    while (x < 6.0) { r += 1.0 / (x * x); x += 1.0; }
    const double f = 1.0 / (x * x);
    return r + 1.0 / x + 0.5 * f + (1.0 / x) * f * (1.0 / 6.0 - f * (1.0 / 30.0 - f * (1.0 / 42.0 - f * (1.0 / 30.0))));
}
__device__ __forceinline__ double sigmoid_d(double x) { return 1.0 / (1.0 + exp(-x)); }

__global__ __launch_bounds__(256) void model_backward_kernel(
    int D, int T, int Q, int N, int M, int d_offset, int mask_size, int nlb, const double *__restrict__ logits,
    const double *__restrict__ gat_raw, const double *__restrict__ aat_raw, const double *__restrict__ bat_raw,
    const double *__restrict__ s_raw, const double *__restrict__ g1_raw, const double *__restrict__ g2_raw,
    const double *__restrict__ w_raw, const double *__restrict__ x_mean, const double *__restrict__ phi, double s1,
    double s2, int add_constants, const double *__restrict__ df_dmu, const double *__restrict__ df_ds,
    const double *__restrict__ df_dz, const double *__restrict__ df_dgamma, const double *__restrict__ df_dab,
    const double *__restrict__ df_dphi, double *__restrict__ d_x_mean, double *__restrict__ d_s_raw, double *__restrict__ d_x_u,
    double *__restrict__ d_logits, double *__restrict__ d_g1_raw, double *__restrict__ d_g2_raw,
    double *__restrict__ d_w_raw, double *__restrict__ d_gat_raw, double *__restrict__ d_aat_raw,
    double *__restrict__ d_bat_raw) {
    const int t = threadIdx.x;
    __shared__ double gat[PREP_MAX_T * DPGP_MAX_Q], aat[PREP_MAX_T], bat[PREP_MAX_T];
    __shared__ double c1[PREP_MAX_T], c2cum[PREP_MAX_T + 1], scratch[8];
    if ((int)blockIdx.x > nlb) {                       // element-wise part
        const size_t nq = (size_t)N * Q, mq = (size_t)M * Q;
        const int nb = gridDim.x - 1 - nlb;
        for (size_t i = (size_t)(blockIdx.x - 1 - nlb) * 256 + t; i < nq + mq; i += (size_t)nb * 256) {
            if (i < nq) {
                const double sv = softplus_d(s_raw[i]);
                d_x_mean[i] = -df_dmu[i] + (add_constants ? x_mean[i] : 0.0);                   // KL: gp_expressions.py:10-24
                d_s_raw[i] = (-df_ds[i] + (add_constants ? 0.5 * (1.0 - 1.0 / sv) : 0.0)) * sigmoid_d(s_raw[i]);
            } else {
                d_x_u[i - nq] = -df_dz[i - nq];
            }
        }
        return;
    }
    for (int i = t; i < T * Q; i += 256) gat[i] = softplus_d(gat_raw[i]);
    if (t < T) {
        aat[t] = softplus_d(aat_raw[t]);
        bat[t] = softplus_d(bat_raw[t]);
    }
    // c1_k = psi(g1)-psi(g1+g2); c2cum_k = sum_{k' < k} (psi(g2)-psi(g1+g2)): the digammas one stick per thread (each is a chain of ~10
    // dependent fp64 divisions: one thread walking all 3 (T - 1) of them was 40 us of this kernel's 70), the prefix sum by thread 0
    __shared__ double c2inc[PREP_MAX_T];
    if (t < T) {
        if (t < T - 1) {
            const double g1 = softplus_d(g1_raw[t]), g2 = softplus_d(g2_raw[t]), p12 = digamma_d(g1 + g2);
            c1[t] = digamma_d(g1) - p12;
            c2inc[t] = digamma_d(g2) - p12;
        } else {
            c1[t] = 0.0;
            c2inc[t] = 0.0;
        }
    }
    __syncthreads();
    if (t == 0) {
        double cum = 0.0;
        for (int k = 0; k < T; ++k) {
            c2cum[k] = cum;
            cum += c2inc[k];
        }
        c2cum[T] = cum;
    }
    __syncthreads();
    // df_dphi != nullptr: the over-T model (dp_gp_lvm_t, reference dp_gp_lvm.py:513-676) — the kernel hyper-parameters ARE the atoms, so
    // df_dgamma [T][Q] and df_dab [T][2] are d f_hat / d (softplus'd atoms) themselves, and phi enters f_hat directly: df_dphi [D][T]
    if (blockIdx.x == 0) {
        // A[k][j] = sum_d phi[d,k] X[d,j], X = [df/dgamma (Q) | df/dalpha | df/dbeta | 1]: output dims staged through LDS in
        // chunks (coalesced loads, fixed summation order) instead of one serial pass over D per thread; as many output dims per chunk as
        // 2048 doubles hold (128 at T = 8, Q = 10: four barriers pairs at D = 512 where chunks of 32 took sixteen: 69 -> ~30 us)
        const int cols = Q + 3;                          // j < Q: gamma atoms; Q: alpha atoms; Q+1: beta atoms; Q+2: sum_d phi
        __shared__ double ph_s[2048], x_s[2048], sphi_s[PREP_MAX_T];
        int DC = 2048 / (T > cols ? T : cols);
        DC = DC > 128 ? 128 : DC;
        for (int e0 = 0; e0 < T * cols; e0 += 256) {
            const int e = e0 + t, k = e / cols, j = e - k * cols;
            double acc = 0.0;
            for (int dc = 0; dc < D; dc += DC) {
                __syncthreads();
                for (int i = t; i < DC * T; i += 256) {
                    const int dd = i / T;
                    ph_s[i] = (dc + dd < D) ? phi[(size_t)dc * T + i] : 0.0;
                }
                for (int i = t; i < DC * cols; i += 256) {
                    const int dd = i / cols, jj = i - dd * cols, d = dc + dd;
                    x_s[i] = (d < D) ? (jj > Q + 1 ? 1.0 : (df_dphi ? 0.0 : (jj < Q ? df_dgamma[(size_t)d * Q + jj] : (jj == Q ? df_dab[2 * d] : df_dab[2 * d + 1])))) : 0.0;
                }
                __syncthreads();
                if (e < T * cols)
                    for (int dd = 0; dd < DC; ++dd) acc += ph_s[dd * T + k] * x_s[dd * cols + j];
            }
            if (e < T * cols) {
                if (j <= Q + 1) {
                    if (df_dphi) acc = j < Q ? df_dgamma[(size_t)k * Q + j] : (j == Q ? df_dab[2 * k] : df_dab[2 * k + 1]);
                    const double raw = j < Q ? gat_raw[k * Q + j] : (j == Q ? aat_raw[k] : bat_raw[k]);
                    const double x = softplus_d(raw);
                    const double hyp = add_constants ? (1.0 / x + log(x) / x) : 0.0;               // -d/dx log_normal.log_pdf(x)
                    const double g = (-acc + hyp) * sigmoid_d(raw);
                    if (j < Q) d_gat_raw[k * Q + j] = g;
                    else if (j == Q) d_aat_raw[k] = g;
                    else d_bat_raw[k] = g;
                } else {
                    sphi_s[k] = acc;
                }
            }
        }
        __syncthreads();
        // q(V): thread k < T-1 needs sum_d phi[d,k] and sum_d tail[d,k], tail[d,k] = sum_{k' > k} phi[d,k']
        if (t < T - 1) {
            const int k = t;
            const double sphi = sphi_s[k];
            double stail = 0.0;
            for (int kk = k + 1; kk < T; ++kk) stail += sphi_s[kk];
            const double g1 = softplus_d(g1_raw[k]), g2 = softplus_d(g2_raw[k]);
            const double t1 = trigamma_d(g1), t2 = trigamma_d(g2), t12 = trigamma_d(g1 + g2);
            double e1 = sphi * (t1 - t12) - stail * t12, e2 = -sphi * t12 + stail * (t2 - t12);
            if (add_constants) {
                const double w1 = softplus_d(w_raw[0]), w2 = softplus_d(w_raw[1]), r = w1 / w2 - 1.0;
                e1 += -r * t12 - (g1 - 1.0) * t1 + (g1 + g2 - 2.0) * t12;
                e2 += r * (t2 - t12) - (g2 - 1.0) * t2 + (g1 + g2 - 2.0) * t12;
            }
            d_g1_raw[k] = -e1 * sigmoid_d(g1_raw[k]);
            d_g2_raw[k] = -e2 * sigmoid_d(g2_raw[k]);
        }
        if (t == 0) {
            double e1 = 0.0, e2 = 0.0;
            if (add_constants) {
                const double w1 = softplus_d(w_raw[0]), w2 = softplus_d(w_raw[1]), tw = trigamma_d(w1), sc2 = c2cum[T];
                e1 = (T - 1.0) * tw + sc2 / w2 + (s1 - 1.0) * tw - s2 / w2 + 1.0 + (1.0 - w1) * tw;
                e2 = -(T - 1.0) / w2 - (w1 / (w2 * w2)) * sc2 - (s1 - 1.0) / w2 + s2 * w1 / (w2 * w2) - 1.0 / w2;
            }
            d_w_raw[0] = -e1 * sigmoid_d(w_raw[0]);
            d_w_raw[1] = -e2 * sigmoid_d(w_raw[1]);
        }
        return;
    }
    // ---- logits rows touched by the local output dims: thread = mask group ----
    const int r0 = d_offset / mask_size, r1 = (d_offset + D - 1) / mask_size;
    const int r = r0 + ((int)blockIdx.x - 1) * 256 + t;
    if (r > r1) return;
    const int dlo = max(r * mask_size, d_offset) - d_offset, dhi = min((r + 1) * mask_size, d_offset + D) - d_offset;
    for (int k = 0; k < T; ++k) d_logits[(size_t)r * T + k] = 0.0;
    for (int d = dlo; d < dhi; ++d) {
        const double *ph = phi + (size_t)d * T;
        const double *dg = df_dgamma + (size_t)(df_dphi ? 0 : d) * Q;
        const double da = df_dphi ? 0.0 : df_dab[2 * d], db = df_dphi ? 0.0 : df_dab[2 * d + 1];
        auto gk = [&](int k) {
            double mix = da * aat[k] + db * bat[k];
            if (df_dphi) mix = df_dphi[(size_t)d * T + k];
            else
                for (int q = 0; q < Q; ++q) mix += dg[q] * gat[k * Q + q];
            // d DP objective / d phi = -( [k < T-1] c1_k + sum_{k' < k} c2_k' - log phi - 1 )
            return -mix - (c1[k] + c2cum[k] - log(ph[k]) - 1.0);
        };
        double sdot = 0.0;
        for (int k = 0; k < T; ++k) sdot += ph[k] * gk(k);
        for (int k = 0; k < T; ++k) d_logits[(size_t)r * T + k] += ph[k] * (gk(k) - sdot);
    }
}

static int model_backward_run(int D, int T, int Q, int N, int M, int d_offset, int mask_size, int logits_rows,
                                   const double *logits, const double *gamma_atoms_raw, const double *alpha_atoms_raw,
                                   const double *beta_atoms_raw, const double *s_raw, const double *g1_raw,
                                   const double *g2_raw, const double *w_raw, const double *x_mean, const double *phi,
                                   double s1, double s2, int add_constants, const double *df_dmu, const double *df_ds,
                                   const double *df_dz, const double *df_dgamma, const double *df_dalpha_beta,
                                   double *d_x_mean, double *d_s_raw, double *d_x_u, double *d_logits, double *d_g1_raw,
                                   double *d_g2_raw, double *d_w_raw, double *d_gamma_atoms_raw,
                                   double *d_alpha_atoms_raw, double *d_beta_atoms_raw, void *stream, const double *df_dphi) {
    if (D <= 0) return -1;
    if (T <= 0 || T > PREP_MAX_T) return -2;
    if (Q <= 0 || Q > DPGP_MAX_Q) return -3;
    if (N <= 0) return -4;
    if (M <= 0) return -5;
    if (d_offset < 0) return -6;
    if (mask_size <= 0) return -7;
    if (logits_rows <= 0 || (d_offset + D - 1) / mask_size >= logits_rows) return -8;
    const void *ins[] = {logits, gamma_atoms_raw, alpha_atoms_raw, beta_atoms_raw, s_raw, g1_raw, g2_raw, w_raw, x_mean, phi};
    for (int i = 0; i < 10; ++i)
        if (!ins[i] && !(T == 1 && (i == 5 || i == 6))) return -(9 + i);
    if (!(s1 > 0.0)) return -19;
    if (!(s2 > 0.0)) return -20;
    const void *gs[] = {df_dmu, df_ds, df_dz, df_dgamma, df_dalpha_beta};
    for (int i = 0; i < 5; ++i)
        if (!gs[i]) return -(22 + i);
    void *outs[] = {d_x_mean, d_s_raw, d_x_u, d_logits, d_g1_raw, d_g2_raw, d_w_raw, d_gamma_atoms_raw, d_alpha_atoms_raw,
                    d_beta_atoms_raw};
    for (int i = 0; i < 10; ++i)
        if (!outs[i] && !(T == 1 && (i == 4 || i == 5))) return -(27 + i);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(d_logits, 0, sizeof(double) * (size_t)logits_rows * T, st) != hipSuccess) return DPGP_ERR_LAUNCH;
    const int rows = (d_offset + D - 1) / mask_size - d_offset / mask_size + 1, nlb = dpgp_ceil_div(rows, 256);
    int neb = dpgp_ceil_div(N * Q + M * Q, 256 * 4);
    if (neb > 1024) neb = 1024;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(model_backward_kernel, dim3(1 + nlb + neb), dim3(256), 0, st, D, T, Q, N, M, d_offset, mask_size, nlb,
                       logits, gamma_atoms_raw, alpha_atoms_raw, beta_atoms_raw, s_raw, g1_raw, g2_raw, w_raw, x_mean, phi, s1, s2,
                       add_constants, df_dmu, df_ds, df_dz, df_dgamma, df_dalpha_beta, df_dphi, d_x_mean, d_s_raw, d_x_u, d_logits,
                       d_g1_raw, d_g2_raw, d_w_raw, d_gamma_atoms_raw, d_alpha_atoms_raw, d_beta_atoms_raw);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

// ---- the trouble flag of a gradient evaluation: out[0] = 1 if any info[0..d) != 0 or any of flat[0..n) is not finite, else 0 (what
// optimise() branches on, summed over the ranks with the packed gradients; eight small torch kernels before) ----
__global__ __launch_bounds__(1024) void trouble_flag_kernel(size_t n, const double *__restrict__ flat, int d, const int *__restrict__ info,
                                                            double *__restrict__ out) {
    // out[0] is zero on entry (dpgp_trouble_flag clears it on the stream); a workgroup that sees trouble stores 1 (every writer the same value)
    const int t = threadIdx.x;
    const size_t i0 = (size_t)blockIdx.x * 1024 + t, stride = (size_t)gridDim.x * 1024;
    int bad = 0;
    for (size_t i = i0; i < n; i += stride) {
        const double v = flat[i];
        bad |= !(fabs(v) <= 1.7976931348623157e308);            // (NaN compares false)
    }
    for (size_t i = i0; i < (size_t)d; i += stride) bad |= info[i] != 0;
    if (__any(bad) && (t & 63) == 0) out[0] = 1.0;
}
extern "C" int dpgp_trouble_flag(size_t n, const double *flat, int d, const int *info, double *out, void *stream) {
    if (n > 0 && !flat) return -2;
    if (d < 0 || (d > 0 && !info)) return -4;
    if (!out) return -5;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(double), st) != hipSuccess) return DPGP_ERR_LAUNCH;
    const size_t work = n > (size_t)d ? n : (size_t)d;
    unsigned grid = (unsigned)((work + 8191) / 8192);
    if (grid < 1) grid = 1;
    if (grid > 256) grid = 256;
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(trouble_flag_kernel, dim3(grid), dim3(1024), 0, st, n, flat, d, info, out);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

extern "C" int dpgp_model_backward(int D, int T, int Q, int N, int M, int d_offset, int mask_size, int logits_rows,
                                   const double *logits, const double *gamma_atoms_raw, const double *alpha_atoms_raw,
                                   const double *beta_atoms_raw, const double *s_raw, const double *g1_raw,
                                   const double *g2_raw, const double *w_raw, const double *x_mean, const double *phi,
                                   double s1, double s2, int add_constants, const double *df_dmu, const double *df_ds,
                                   const double *df_dz, const double *df_dgamma, const double *df_dalpha_beta,
                                   double *d_x_mean, double *d_s_raw, double *d_x_u, double *d_logits, double *d_g1_raw,
                                   double *d_g2_raw, double *d_w_raw, double *d_gamma_atoms_raw,
                                   double *d_alpha_atoms_raw, double *d_beta_atoms_raw, void *stream) {
    return model_backward_run(D, T, Q, N, M, d_offset, mask_size, logits_rows, logits, gamma_atoms_raw, alpha_atoms_raw, beta_atoms_raw,
                              s_raw, g1_raw, g2_raw, w_raw, x_mean, phi, s1, s2, add_constants, df_dmu, df_ds, df_dz, df_dgamma,
                              df_dalpha_beta, d_x_mean, d_s_raw, d_x_u, d_logits, d_g1_raw, d_g2_raw, d_w_raw, d_gamma_atoms_raw,
                              d_alpha_atoms_raw, d_beta_atoms_raw, stream, nullptr);
}
// the same for the over-T model dp_gp_lvm_t: df_dgamma_atoms [T][Q] / df_dalpha_beta_atoms [T][2] = d f_hat / d (softplus'd atoms) and
// df_dphi [D][T] = d f_hat / d phi (both direct: no mixing between them)
extern "C" int dpgp_model_backward_t(int D, int T, int Q, int N, int M, int d_offset, int mask_size, int logits_rows,
                                     const double *logits, const double *gamma_atoms_raw, const double *alpha_atoms_raw,
                                     const double *beta_atoms_raw, const double *s_raw, const double *g1_raw, const double *g2_raw,
                                     const double *w_raw, const double *x_mean, const double *phi, double s1, double s2,
                                     int add_constants, const double *df_dmu, const double *df_ds, const double *df_dz,
                                     const double *df_dgamma_atoms, const double *df_dalpha_beta_atoms, const double *df_dphi,
                                     double *d_x_mean, double *d_s_raw, double *d_x_u, double *d_logits, double *d_g1_raw,
                                     double *d_g2_raw, double *d_w_raw, double *d_gamma_atoms_raw, double *d_alpha_atoms_raw,
                                     double *d_beta_atoms_raw, void *stream) {
    if (!df_dphi) return -27;
    const int rc = model_backward_run(D, T, Q, N, M, d_offset, mask_size, logits_rows, logits, gamma_atoms_raw, alpha_atoms_raw,
                                      beta_atoms_raw, s_raw, g1_raw, g2_raw, w_raw, x_mean, phi, s1, s2, add_constants, df_dmu, df_ds,
                                      df_dz, df_dgamma_atoms, df_dalpha_beta_atoms, d_x_mean, d_s_raw, d_x_u, d_logits, d_g1_raw,
                                      d_g2_raw, d_w_raw, d_gamma_atoms_raw, d_alpha_atoms_raw, d_beta_atoms_raw, stream, df_dphi);
    return rc <= -27 ? rc - 1 : rc;                 // (argument indices behind df_dphi move by one)
}

extern "C" int dpgp_model_scal_count(int D) { return D > 0 ? 2 + dpgp_ceil_div(D, PREP_ROWS) : 0; }

static int model_prepare_run(int D, int T, int Q, int N, int d_offset, int mask_size, const double *logits,
                             const double *gamma_atoms_raw, const double *alpha_atoms_raw, const double *beta_atoms_raw,
                             const double *s_raw, const double *g1_raw, const double *g2_raw, const double *w_raw, double s1,
                             double s2, int add_constants, double *gamma, double *alpha, double *beta, double *s, double *phi,
                             double *scal, double *atoms, void *stream) {
    if (D <= 0) return -1;
    if (T <= 0 || T > PREP_MAX_T) return -2;
    if (Q <= 0 || Q > DPGP_MAX_Q) return -3;
    if (N <= 0) return -4;
    if (d_offset < 0) return -5;
    if (mask_size <= 0) return -6;
    if (!logits) return -7;
    if (!gamma_atoms_raw) return -8;
    if (!alpha_atoms_raw) return -9;
    if (!beta_atoms_raw) return -10;
    if (!s_raw) return -11;
    if (T > 1 && (!g1_raw || !g2_raw)) return -12;
    if (!w_raw) return -14;
    if (!(s1 > 0.0)) return -15;
    if (!(s2 > 0.0)) return -16;
    if (!s) return -21;
    if (!scal) return -23;
    const int nrb = dpgp_ceil_div(D, PREP_ROWS);
    int sblocks = dpgp_ceil_div(N * Q, 256 * 2);
    if (sblocks > 512) sblocks = 512;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(model_prepare_kernel, dim3(1 + nrb + sblocks), dim3(256), 0, (hipStream_t)stream, D, T, Q, N,
                       d_offset, mask_size, nrb, logits, gamma_atoms_raw, alpha_atoms_raw, beta_atoms_raw, s_raw, g1_raw,
                       g2_raw, w_raw, s1, s2, add_constants, gamma, alpha, beta, s, phi, scal, atoms);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
extern "C" int dpgp_model_prepare(int D, int T, int Q, int N, int d_offset, int mask_size, const double *logits,
                                  const double *gamma_atoms_raw, const double *alpha_atoms_raw,
                                  const double *beta_atoms_raw, const double *s_raw, const double *g1_raw,
                                  const double *g2_raw, const double *w_raw, double s1, double s2, int add_constants,
                                  double *gamma, double *alpha, double *beta, double *s, double *phi, double *scal,
                                  void *stream) {
    if (!gamma) return -18;
    if (!alpha) return -19;
    if (!beta) return -20;
    return model_prepare_run(D, T, Q, N, d_offset, mask_size, logits, gamma_atoms_raw, alpha_atoms_raw, beta_atoms_raw, s_raw,
                             g1_raw, g2_raw, w_raw, s1, s2, add_constants, gamma, alpha, beta, s, phi, scal, nullptr, stream);
}
// the same launch for the over-T model (dp_gp_lvm_t): no mixing; phi[D,T] and the softplus of the atoms are the outputs
extern "C" int dpgp_model_prepare_t(int D, int T, int Q, int N, int d_offset, int mask_size, const double *logits,
                                    const double *gamma_atoms_raw, const double *alpha_atoms_raw,
                                    const double *beta_atoms_raw, const double *s_raw, const double *g1_raw,
                                    const double *g2_raw, const double *w_raw, double s1, double s2, int add_constants,
                                    double *s, double *phi, double *atoms, double *scal, void *stream) {
    if (!phi) return -22;
    if (!atoms) return -24;
    return model_prepare_run(D, T, Q, N, d_offset, mask_size, logits, gamma_atoms_raw, alpha_atoms_raw, beta_atoms_raw, s_raw,
                             g1_raw, g2_raw, w_raw, s1, s2, add_constants, nullptr, nullptr, nullptr, s, phi, scal, atoms, stream);
}

extern "C" int dpgp_model_pack(int D, const double *fhat, const double *scal, double *pack, void *stream) {
    if (D <= 0) return -1;
    if (!fhat) return -2;
    if (!scal) return -3;
    if (!pack) return -4;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(model_pack_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, fhat, scal,
                       dpgp_ceil_div(D, PREP_ROWS), pack);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

extern "C" int dpgp_model_finalize(const double *pack, const double *kl, const double *hyper, double *out,
                                   void *stream) {
    if (!pack) return -1;
    if (!kl) return -2;
    if (!hyper) return -3;
    if (!out) return -4;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(model_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pack, pack + 1, kl, hyper, out);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

extern "C" const char *dpgp_last_hip_error(void) { return hipGetErrorString(dpgp_last_error_slot()); }
