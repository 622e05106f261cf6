// Fused per-output ELBO reduction (the hot path of dp_gp_lvm.py:108-148): six kernel launches on one stream, no host
// synchronisation, no allocation — graph-capturable.
//   1. KL(q(X)||p(X)) and y_d^T y_d                        (kl_yy_kernel)
//   2. K_uu + jitter I straight into the padded Cholesky workspace (gram_kernel)
//   3. Psi1_d^T y_d partial slabs                          (psi1T_y_kernel)
//   4. Psi2_d partial slabs on the matrix cores            (psi2_mfma_kernel)
//   5. per-d Cholesky chain + the five f_hat terms         (la_chain_kernel)
//   6. f_hat = sum of terms                                 (sum_terms_kernel)
#include "internal.h"

struct ElboLayout {
    size_t off_yy, off_v, off_p2, off_la, total;
    int ns1, ns2, Mp;
};

static ElboLayout elbo_layout(int D, int N, int M, int prec) {
    ElboLayout L;
    L.Mp = dpgp_round_up(M, 16);
    L.ns1 = psi1T_y_nsplit(D, N, M);
    L.ns2 = psi2_nsplit(D, N, M);
    const size_t sp = (prec == DPGP_PREC_F64) ? 8 : 4, sl = (prec == DPGP_PREC_F32) ? 4 : 8;
    size_t o = 0;
    L.off_yy = o; o += dpgp_align256(sizeof(double) * D);
    L.off_v = o;  o += dpgp_align256(sizeof(double) * (size_t)L.ns1 * D * M);
    L.off_p2 = o; o += dpgp_align256(sp * (size_t)L.ns2 * D * L.Mp * L.Mp);
    L.off_la = o; o += dpgp_align256(sl * (size_t)D * la_chain_ws_elems(M));
    L.total = o;
    return L;
}

extern "C" size_t dpgp_elbo_workspace_bytes(int D, int N, int M, int Q, int prec) {
    if (D <= 0 || N <= 0 || M <= 0 || Q <= 0 || prec < 0 || prec > 2) return 0;
    return elbo_layout(D, N, M, prec).total;
}

template <typename TP, typename TL>
static int elbo_run(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                    const double *s, const double *gamma, const double *alpha, const double *beta, double jitter,
                    int algo, double *terms, double *sums, int *info, unsigned char *ws, const ElboLayout &L,
                    hipStream_t st) {
    double *yy = reinterpret_cast<double *>(ws + L.off_yy);
    double *vpart = reinterpret_cast<double *>(ws + L.off_v);
    TP *p2 = reinterpret_cast<TP *>(ws + L.off_p2);
    TL *la = reinterpret_cast<TL *>(ws + L.off_la);
    int rc;
    if ((rc = launch_kl_yy<double>(N, Q, mu, s, sums + 1, D, y, ldy, yy, st))) return rc;
    if ((rc = launch_gram<double, TL>(D, M, M, Q, z, nullptr, gamma, alpha, beta, DPGP_FLAG_JITTER, jitter, la, L.Mp,
                                      la_chain_ws_elems(M), st)))
        return rc;
    if ((rc = launch_psi1T_y_partial<double, TP>(D, N, M, Q, z, mu, s, gamma, alpha, y, ldy, vpart, L.ns1, st)))
        return rc;
    if ((rc = launch_psi2_partial<double, TP>(D, N, M, Q, z, mu, s, gamma, alpha, p2, L.ns2, algo, st))) return rc;
    if ((rc = launch_la_chain<TP, TL>(D, N, M, la, p2, L.ns2, vpart, L.ns1, alpha, beta, yy, terms, info, la, algo,
                                      st)))
        return rc;
    return launch_sum_terms(D, terms, sums, st);
}

extern "C" int dpgp_elbo_fhat(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                              const double *s, const double *gamma, const double *alpha, const double *beta,
                              double jitter, int prec, int algo, double *terms, double *sums, int *info, void *ws,
                              size_t ws_bytes, void *stream) {
    if (D <= 0) return -1;
    if (N <= 0) return -2;
    if (M <= 0 || M > N) return -3;
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
    if (algo != DPGP_ALGO_AUTO && algo != DPGP_ALGO_PLAIN) return -15;
    if (!terms) return -16;
    if (!sums) return -17;
    if (!info) return -18;
    if (!ws) return -19;
    const ElboLayout L = elbo_layout(D, N, M, prec);
    if (ws_bytes < L.total) return -20;
    hipStream_t st = (hipStream_t)stream;
    unsigned char *w = (unsigned char *)ws;
    switch (prec) {
    case DPGP_PREC_F32:
        return elbo_run<float, float>(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, beta, jitter, algo, terms, sums, info,
                                      w, L, st);
    case DPGP_PREC_MIXED:
        return elbo_run<float, double>(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, beta, jitter, algo, terms, sums,
                                       info, w, L, st);
    default:
        return elbo_run<double, double>(D, N, M, Q, y, ldy, z, mu, s, gamma, alpha, beta, jitter, algo, terms, sums,
                                        info, w, L, st);
    }
}
