/*
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C fp64 restatement of the DP-GP-LVM variational ELBO inner loop of AndrewRLawrence/dp_gp_lvm, memory-lean
 * so that it also runs the BASELINE.json shapes the reference itself cannot hold in memory (its psi2 materialises
 * [D,N,M,M,Q], src/kernels/rbf_kernel.py:194-197, and its data-fit term [D,N,N], src/models/dp_gp_lvm.py:134).
 * Used (through oracle/c_oracle.py) only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — as the
 * checker and as the timed CPU "port" baseline; dp_gp_lvm_amd never links or loads it.
 *
 * Parity status: PINNED — tests/test_oracle_golden.py checks every function here against tests/golden/*.npz, which
 * oracle/gen_golden.py produced by running the reference's own source and its NumPy known-answer functions.
 *
 * Built twice by oracle/Makefile from this one file:
 *   _build/libdpgp_oracle.so       strict IEEE (-O2)                      -> the checker
 *   _build/libdpgp_oracle_fast.so  -O3 -ffast-math -march=x86-64-v3 (libmvec exp) -> cpu_baseline timing; also checked
 * Row-major, leading batch dimension, `s` = DIAGONAL of q(X)'s covariance [N,Q]. Citations are into /root/reference.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DPGP_FLAG_NOISE 1
#define DPGP_FLAG_JITTER 2
static const double LOG_2PI = 1.8378770664093454835606594728112;

int dpgp_ref_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* rbf_kernel.py:58-93.  x1 == NULL means "input_1 is None": only then are noise / jitter added (:80,:86). */
void dpgp_ref_gram(int B, int N0, int N1, int Q, const double *x0, const double *x1, const double *gamma,
                   const double *alpha, const double *beta, int flags, double jitter, double *out, int nthreads) {
    const double *xb = x1 ? x1 : x0;
    if (!x1) N1 = N0;
#pragma omp parallel for collapse(2) num_threads(nthreads) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N0; ++i) {
            const double *g = gamma + (size_t)b * Q;
            double *o = out + ((size_t)b * N0 + i) * N1;
            for (int j = 0; j < N1; ++j) {
                double e = 0.0;
                for (int q = 0; q < Q; ++q) {
                    double df = x0[(size_t)i * Q + q] - xb[(size_t)j * Q + q];
                    e += g[q] * df * df;
                }
                o[j] = alpha[b] * exp(-0.5 * e);
            }
            if (!x1) {
                if (flags & DPGP_FLAG_NOISE) o[i] += 1.0 / beta[b];
                if (flags & DPGP_FLAG_JITTER) o[i] += jitter;
            }
        }
}

/* rbf_kernel.py:135-161 -> out[B,N,M] */
void dpgp_ref_psi1(int B, int N, int M, int Q, const double *z, const double *mu, const double *s,
                   const double *gamma, const double *alpha, double *out, int nthreads) {
#pragma omp parallel for collapse(2) num_threads(nthreads) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            const double *g = gamma + (size_t)b * Q;
            double w[64], ld = 0.0;
            for (int q = 0; q < Q; ++q) {
                double den = g[q] * s[(size_t)n * Q + q] + 1.0;
                w[q] = g[q] / den;
                ld += log(den);
            }
            double *o = out + ((size_t)b * N + n) * M;
            for (int m = 0; m < M; ++m) {
                double e = ld;
                for (int q = 0; q < Q; ++q) {
                    double df = mu[(size_t)n * Q + q] - z[(size_t)m * Q + q];
                    e += w[q] * df * df;
                }
                o[m] = alpha[b] * exp(-0.5 * e);
            }
        }
}

/* Psi1_d^T y_d -> out[D,M]; y is [N,D] with leading dimension ldy (dp_gp_lvm.py:132-145: Psi1 only enters via c y_d). */
static void psi1T_y_one(int N, int M, int Q, const double *z, const double *mu, const double *s, const double *g,
                        double alpha, const double *y, int ldy, double *v) {
    for (int m = 0; m < M; ++m) v[m] = 0.0;
    for (int n = 0; n < N; ++n) {
        double w[64], ld = 0.0;
        for (int q = 0; q < Q; ++q) {
            double den = g[q] * s[(size_t)n * Q + q] + 1.0;
            w[q] = g[q] / den;
            ld += log(den);
        }
        double yn = y[(size_t)n * ldy];
        for (int m = 0; m < M; ++m) {
            double e = ld;
            for (int q = 0; q < Q; ++q) {
                double df = mu[(size_t)n * Q + q] - z[(size_t)m * Q + q];
                e += w[q] * df * df;
            }
            v[m] += alpha * exp(-0.5 * e) * yn;
        }
    }
}

void dpgp_ref_psi1T_y(int D, int N, int M, int Q, const double *z, const double *mu, const double *s,
                      const double *gamma, const double *alpha, const double *y, int ldy, double *out, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(dynamic)
    for (int d = 0; d < D; ++d)
        psi1T_y_one(N, M, Q, z, mu, s, gamma + (size_t)d * Q, alpha[d], y + d, ldy, out + (size_t)d * M);
}

/* rbf_kernel.py:164-199, the LITERAL per-element formula (validation of the streamed form below; small shapes). */
void dpgp_ref_psi2_literal(int B, int N, int M, int Q, const double *z, const double *mu, const double *s,
                           const double *gamma, const double *alpha, double *out, int nthreads) {
#pragma omp parallel for collapse(2) num_threads(nthreads) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int m1 = 0; m1 < M; ++m1) {
            const double *g = gamma + (size_t)b * Q;
            for (int m2 = 0; m2 < M; ++m2) {
                double acc = 0.0;
                for (int n = 0; n < N; ++n) {
                    double lp = 2.0 * log(alpha[b]);
                    for (int q = 0; q < Q; ++q) {
                        double den = 2.0 * g[q] * s[(size_t)n * Q + q] + 1.0;            /* :193 */
                        double zb = 0.5 * (z[(size_t)m1 * Q + q] + z[(size_t)m2 * Q + q]); /* :189 */
                        double dz = z[(size_t)m1 * Q + q] - z[(size_t)m2 * Q + q];
                        double dm = mu[(size_t)n * Q + q] - zb;
                        lp -= 0.5 * log(den) + 0.25 * g[q] * dz * dz + g[q] * dm * dm / den; /* :191-197 */
                    }
                    acc += exp(lp);
                }
                out[((size_t)b * M + m1) * M + m2] = acc;                                 /* :199 */
            }
        }
}

/*
 * psi2 for ONE batch entry, streamed over n (identical to the literal formula, verified by
 * tests/test_oracle_golden.py):  with den = 2 g s + 1, w = g/den, u = w mu,
 *   log psi2[n,m,m'] = 2 log alpha + a_n + b_mm' + sum_q( -w_nq/4 (z_m+z_m')_q^2 + u_nq (z_m+z_m')_q ),
 *   a_n = -1/2 sum log den - sum w mu^2,   b_mm' = -1/4 sum_q g_q (z_m - z_m')_q^2.
 * Lower triangle m' <= m only, mirrored on store.  work: (2Q+1) * NC doubles.
 */
#define NC 256
static void psi2_one(int N, int M, int Q, const double *z, const double *mu, const double *s, const double *g,
                     double alpha, double *out, double *work) {
    double *F = work;                 /* [2Q][NC]: rows 0..Q-1 = -w/4, rows Q..2Q-1 = u */
    double *A = work + (size_t)2 * Q * NC; /* [NC] a_n */
    for (size_t i = 0; i < (size_t)M * M; ++i) out[i] = 0.0;
    for (int n0 = 0; n0 < N; n0 += NC) {
        int nc = N - n0 < NC ? N - n0 : NC;
        for (int i = 0; i < nc; ++i) {
            double a = 0.0;
            for (int q = 0; q < Q; ++q) {
                double den = 2.0 * g[q] * s[(size_t)(n0 + i) * Q + q] + 1.0;
                double w = g[q] / den, m_ = mu[(size_t)(n0 + i) * Q + q];
                a -= 0.5 * log(den) + w * m_ * m_;
                F[(size_t)q * NC + i] = -0.25 * w;
                F[(size_t)(Q + q) * NC + i] = w * m_;
            }
            A[i] = a;
        }
        for (int m1 = 0; m1 < M; ++m1)
            for (int m2 = 0; m2 <= m1; ++m2) {
                double e[NC];
                for (int i = 0; i < nc; ++i) e[i] = A[i];
                for (int q = 0; q < Q; ++q) {
                    double sm = z[(size_t)m1 * Q + q] + z[(size_t)m2 * Q + q], sm2 = sm * sm;
                    const double *f0 = F + (size_t)q * NC, *f1 = F + (size_t)(Q + q) * NC;
                    for (int i = 0; i < nc; ++i) e[i] += f0[i] * sm2 + f1[i] * sm;
                }
                double acc = 0.0;
                for (int i = 0; i < nc; ++i) acc += exp(e[i]);
                out[(size_t)m1 * M + m2] += acc;
            }
    }
    for (int m1 = 0; m1 < M; ++m1)
        for (int m2 = 0; m2 <= m1; ++m2) {
            double b = 0.0;
            for (int q = 0; q < Q; ++q) {
                double dz = z[(size_t)m1 * Q + q] - z[(size_t)m2 * Q + q];
                b += g[q] * dz * dz;
            }
            double v = alpha * alpha * exp(-0.25 * b) * out[(size_t)m1 * M + m2];
            out[(size_t)m1 * M + m2] = v;
            out[(size_t)m2 * M + m1] = v;
        }
}

void dpgp_ref_psi2(int B, int N, int M, int Q, const double *z, const double *mu, const double *s,
                   const double *gamma, const double *alpha, double *out, int nthreads) {
#pragma omp parallel num_threads(nthreads)
    {
        double *work = (double *)malloc(sizeof(double) * (size_t)(2 * Q + 1) * NC);
#pragma omp for schedule(dynamic)
        for (int b = 0; b < B; ++b)
            psi2_one(N, M, Q, z, mu, s, gamma + (size_t)b * Q, alpha[b], out + (size_t)b * M * M, work);
        free(work);
    }
}

/* Lower Cholesky in place (row-major, lower triangle referenced; upper zeroed). Returns 0 or 1-based failing minor. */
int dpgp_ref_potrf(int M, double *a) {
    for (int j = 0; j < M; ++j) {
        double d = a[(size_t)j * M + j];
        for (int k = 0; k < j; ++k) d -= a[(size_t)j * M + k] * a[(size_t)j * M + k];
        if (!(d > 0.0)) return j + 1;
        d = sqrt(d);
        a[(size_t)j * M + j] = d;
        for (int i = j + 1; i < M; ++i) {
            double v = a[(size_t)i * M + j];
            for (int k = 0; k < j; ++k) v -= a[(size_t)i * M + k] * a[(size_t)j * M + k];
            a[(size_t)i * M + j] = v / d;
        }
        for (int k = j + 1; k < M; ++k) a[(size_t)j * M + k] = 0.0;
    }
    return 0;
}

/* rhs[M,K] <- L^-1 rhs (forward substitution, L lower [M,M]). */
void dpgp_ref_trsm(int M, int K, const double *l, double *rhs) {
    for (int i = 0; i < M; ++i) {
        double *ri = rhs + (size_t)i * K;
        for (int k = 0; k < i; ++k) {
            double lik = l[(size_t)i * M + k];
            const double *rk = rhs + (size_t)k * K;
            for (int c = 0; c < K; ++c) ri[c] -= lik * rk[c];
        }
        double inv = 1.0 / l[(size_t)i * M + i];
        for (int c = 0; c < K; ++c) ri[c] *= inv;
    }
}

/*
 * dp_gp_lvm.py:108-145 per output dim d -> terms[D,5] (see oracle/dpgp_oracle.py:fhat_terms for the five terms) and
 * info[D] (0, or 1-based failing minor of chol(K_uu) / M + that of chol(A)).  y is [N,D] (ld = D).
 * Optional outputs (may be NULL): psi2_out[D,M,M], kuu_out[D,M,M], v_out[D,M].
 */
void dpgp_ref_fhat_terms(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu,
                         const double *s, const double *gamma, const double *alpha, const double *beta, double jitter,
                         double *terms, int *info, double *psi2_out, double *kuu_out, double *v_out, int nthreads) {
#pragma omp parallel num_threads(nthreads)
    {
        size_t mm = (size_t)M * M;
        double *work = (double *)malloc(sizeof(double) * ((size_t)(2 * Q + 1) * NC + 3 * mm + 2 * (size_t)M));
        double *K = work + (size_t)(2 * Q + 1) * NC, *P2 = K + mm, *X = P2 + mm, *v = X + mm;
#pragma omp for schedule(dynamic)
        for (int d = 0; d < D; ++d) {
            const double *g = gamma + (size_t)d * Q;
            double al = alpha[d], be = beta[d];
            double one = 1.0;
            dpgp_ref_gram(1, M, M, Q, z, NULL, g, &al, &one, DPGP_FLAG_JITTER, jitter, K, 1);   /* :115 */
            psi2_one(N, M, Q, z, mu, s, g, al, P2, work);                                          /* :110 */
            psi1T_y_one(N, M, Q, z, mu, s, g, al, y + d, ldy, v);
            if (kuu_out) memcpy(kuu_out + (size_t)d * mm, K, sizeof(double) * mm);
            if (psi2_out) memcpy(psi2_out + (size_t)d * mm, P2, sizeof(double) * mm);
            if (v_out) memcpy(v_out + (size_t)d * M, v, sizeof(double) * M);
            double *t = terms + (size_t)d * 5;
            double yy = 0.0;
            for (int n = 0; n < N; ++n) yy += y[(size_t)n * ldy + d] * y[(size_t)n * ldy + d];
            t[0] = 0.5 * N * (log(be) - LOG_2PI);
            t[3] = -0.5 * be * yy;
            t[1] = t[2] = t[4] = NAN;
            info[d] = dpgp_ref_potrf(M, K);                                                        /* :116 */
            if (info[d]) continue;
            memcpy(X, P2, sizeof(double) * mm);
            dpgp_ref_trsm(M, M, K, X);                                                             /* :118 */
            for (int i = 0; i < M; ++i)                                                            /* transpose */
                for (int j = 0; j < i; ++j) { double tmp = X[(size_t)i * M + j]; X[(size_t)i * M + j] = X[(size_t)j * M + i]; X[(size_t)j * M + i] = tmp; }
            dpgp_ref_trsm(M, M, K, X);                                                             /* :119-121 (T2^T = T2) */
            double tr = 0.0;
            for (int i = 0; i < M; ++i) tr += X[(size_t)i * M + i];
            for (size_t i = 0; i < mm; ++i) X[i] *= be;
            for (int i = 0; i < M; ++i) X[(size_t)i * M + i] += 1.0;                              /* :124-126 */
            int ia = dpgp_ref_potrf(M, X);                                                         /* :127 */
            if (ia) { info[d] = M + ia; continue; }
            double ld = 0.0;
            for (int i = 0; i < M; ++i) ld += log(X[(size_t)i * M + i]);                          /* :129 */
            dpgp_ref_trsm(M, 1, K, v);                                                             /* :132 */
            dpgp_ref_trsm(M, 1, X, v);                                                             /* :133 */
            double cc = 0.0;
            for (int i = 0; i < M; ++i) cc += v[i] * v[i];
            t[1] = -ld;
            t[2] = 0.5 * be * (tr - al * N);                                                       /* :140-142 */
            t[4] = 0.5 * be * be * cc;                                                             /* :145 */
        }
        free(work);
    }
}

/* gp_expressions.py:10-24 */
double dpgp_ref_kl_qx(int N, int Q, const double *mu, const double *s) {
    double a = 0.0;
    for (size_t i = 0; i < (size_t)N * Q; ++i) a += mu[i] * mu[i] + s[i] - log(s[i]);
    return 0.5 * (a - (double)N * Q);
}
