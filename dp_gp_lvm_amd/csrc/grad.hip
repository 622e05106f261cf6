// Backward pass of the fused ELBO (the reference trains on tf.gradients(objective),
// test/synthetic_data_hard_test.py:143-155).  First, correctness-oriented version; oracle: oracle/dpgp_oracle_torch.py,
// pinned by tests/golden/grad_ref_*.npz (gradients of the reference's own objective).
//
// Stage A (this file, chain_grad_kernel): adjoints of the per-output dense algebra.  With B = K + beta P (P = Psi2,
// v = Psi1^T y, w = B^-1 v) the per-output term of f_hat (dp_gp_lvm.py:108-145 in the form of DESIGN.md section 2) is
//     f = 1/2 N (log beta - log 2 pi) - log|L_B| + log|L_K| + 1/2 beta (<K^-1, P> - alpha N) - 1/2 beta y^T y + 1/2 beta^2 v^T w
// and its adjoints are
//     G_B = -1/2 B^-1 - 1/2 beta^2 w w^T            G_v = beta^2 w
//     G_K = 1/2 K^-1 - 1/2 beta K^-1 P K^-1 + G_B   G_P = 1/2 beta K^-1 + beta G_B
//     df/dbeta  = N / (2 beta) + 1/2 (<K^-1, P> - alpha N) - 1/2 y^T y + beta v^T w + <G_B, P>
//     df/dalpha = -1/2 beta N + ( <G_K, K - jitter I> + 2 <G_P, P> + <G_v, v> ) / alpha      (K - jitter I, P, v scale with alpha, alpha^2, alpha)
// One 256-thread workgroup per output dim, fp64, LDS-resident (M <= 128): B is assembled, factored (potrf_lds) and
// inverted in place (potri_lds) in LDS; K^-1 P K^-1 is two passes of 16x16 fp64 MFMA tile products over the symmetric
// matrices in global memory.  It runs after a forward evaluation on the same workspace (K, K^-1 and the Psi slabs are there).
#include "internal.h"
#include "linalg_dev.h"
#include "psi2_consts.h"

__device__ __forceinline__ double sym_at(const double *__restrict__ a, int ld, int i, int j) {
    return (i >= j) ? a[(size_t)i * ld + j] : a[(size_t)j * ld + i];
}
__device__ __forceinline__ double binv_at(const double *tiles, int nb, int i, int j) {      // lower tiles of a symmetric matrix
    if (i < j) { const int s = i; i = j; j = s; }
    return tiles[lds_tile_index(i >> 4, j >> 4, nb) * TSZ + (i & 15) * LDT + (j & 15)];
}

template <typename TP>
__global__ __launch_bounds__(256, 2) void chain_grad_kernel(int D, int N, int M, int Mp, const TP *__restrict__ psi2_part,
                                                            int ns2, const double *__restrict__ v_part, int ns1,
                                                            const double *__restrict__ alpha,
                                                            const double *__restrict__ beta,
                                                            const double *__restrict__ yy_part, double jitter,
                                                            double *__restrict__ ws, size_t ws_stride,
                                                            double *__restrict__ GP, double *__restrict__ WK,
                                                            double *__restrict__ Gv, double *__restrict__ dab,
                                                            int *__restrict__ info) {
    typedef Mfma<double>::acc_t acc_t;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double *scratch = reinterpret_cast<double *>(smem_raw);
    int &fail = *reinterpret_cast<int *>(smem_raw + 64);
    double *dinv = reinterpret_cast<double *>(smem_raw + LA_LDS_HDR);
    double *tiles = dinv + TSZ;
    double *vv = dinv, *ww = dinv + 128;                      // (the dinv tile is free once the inverse is done; Mp <= 128)
    const int d = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kk = lane >> 4;
    const int nb = Mp / 16, nlow = nb * (nb + 1) / 2;
    const double *K0 = ws + (size_t)d * ws_stride;
    double *Pd = ws + (size_t)d * ws_stride + (size_t)Mp * Mp;                 // Kb region: the summed Psi2 (lower)
    double *T1 = Pd + (size_t)Mp * Mp;                                          // Wb region: K^-1 P (full)
    const double *KI = T1 + (size_t)(Mp + 16) * Mp;
    double *dinv_g = const_cast<double *>(KI) + (size_t)Mp * Mp;
    const double be = beta[d], al = alpha[d];
    if (t == 0) fail = 0;
    // ---- B = K + beta P -> LDS, P -> Pd ----
    {
        typedef TP tp4 __attribute__((ext_vector_type(4)));
        typedef double d4 __attribute__((ext_vector_type(4)));
        const int u = t >> 6, r = (t & 63) >> 2, c4 = (t & 3) * 4;
        for (int t0 = 0; t0 < nlow; t0 += 4) {
            const int tt = t0 + u;
            if (tt >= nlow) continue;
            int I = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
            while ((I + 1) * (I + 2) / 2 <= tt) ++I;
            while (I * (I + 1) / 2 > tt) --I;
            const int J = tt - I * (I + 1) / 2, i = 16 * I + r, j = 16 * J + c4;
            const size_t off = (size_t)i * Mp + j;
            const d4 k0 = *reinterpret_cast<const d4 *>(K0 + off);
            double p2[4] = {0.0, 0.0, 0.0, 0.0};
            for (int k = 0; k < ns2; ++k) {
                const tp4 v = *reinterpret_cast<const tp4 *>(psi2_part + ((size_t)k * D + d) * (size_t)Mp * Mp + off);
#pragma unroll
                for (int e = 0; e < 4; ++e) p2[e] += (double)v[e];
            }
            double *dst = tiles + lds_tile_index(I, J, nb) * TSZ + r * LDT + c4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int je = j + e;
                const bool in = (je <= i && i < M);
                dst[e] = in ? k0[e] + be * p2[e] : ((i == je) ? 1.0 : 0.0);
                Pd[off + e] = in ? p2[e] : 0.0;
            }
        }
    }
    for (int j = t; j < Mp; j += 256) {
        double a = 0.0;
        if (j < M)
            for (int k = 0; k < ns1; ++k) a += v_part[((size_t)k * D + d) * M + j];
        Gv[(size_t)d * Mp + j] = a;                           // parked in the output buffer until vv/ww exist
    }
    __threadfence_block();
    __syncthreads();
    potrf_lds<double, 2>(tiles, dinv, nb, nb, &fail, dinv_g);
    __threadfence_block();
    __syncthreads();
    potri_lds<double>(tiles, dinv, dinv_g, nb);               // tiles = B^-1 (lower)
    __syncthreads();
    if (t < Mp) vv[t] = Gv[(size_t)d * Mp + t];
    __syncthreads();
    if (t < Mp) {
        double a = 0.0;
        if (t < M)
            for (int j = 0; j < M; ++j) a += binv_at(tiles, nb, t, j) * vv[j];
        ww[t] = a;
    }
    __syncthreads();
    const double vw = block_sum((t < M) ? vv[t] * ww[t] : 0.0, scratch);
    // ---- T1 = K^-1 P (full), 16x16 tile products C += X Y^T on the symmetric operands ----
    // Register-blocked: a wave owns two tile rows and all their column tiles, so one K-step of the operands is 2 + nb loads for
    // 2 nb matrix instructions (one tile per wave and trip: 2 loads per instruction, all of them 16 x 32-byte gathers from L2 — they
    // set the kernel's pace).  nb <= 8 (Mp <= 128): at most one pair of rows per wave.
    constexpr int NBX = 8;
    {
        const int I0 = 2 * wv, I1 = I0 + 1;
        if (I0 < nb) {
            acc_t c[2][NBX];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int J = 0; J < NBX; ++J) c[r][J] = (acc_t){0, 0, 0, 0};
            const bool two = I1 < nb;
            for (int k = 0; k < nb; ++k) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int col = 16 * k + 4 * ks + kk;
                    const double a0 = sym_at(KI, Mp, 16 * I0 + li, col), a1 = two ? sym_at(KI, Mp, 16 * I1 + li, col) : 0.0;
                    double bj[NBX];
#pragma unroll
                    for (int J = 0; J < NBX; ++J) bj[J] = J < nb ? sym_at(Pd, Mp, 16 * J + li, col) : 0.0;
#pragma unroll
                    for (int J = 0; J < NBX; ++J)
                        if (J < nb) {
                            c[0][J] = Mfma<double>::mma(a0, bj[J], c[0][J]);
                            c[1][J] = Mfma<double>::mma(a1, bj[J], c[1][J]);
                        }
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int J = 0; J < NBX; ++J)
                    if (J < nb && (r == 0 || two)) {
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            T1[(size_t)(16 * (I0 + r) + Mfma<double>::row(lane, v)) * Mp + 16 * J + li] = c[r][J][v];
                    }
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- X = T1 K^-1 (lower tiles) and the outputs, element by element in the MFMA result layout ----
    // wave p owns the tile rows p and nb - 1 - p (p + 1 and nb - p lower tiles: nb + 1 per wave)
    double sK = 0.0, sP = 0.0, sGBP = 0.0, tr = 0.0;
    {
        const int Ia = wv, Ib = nb - 1 - wv;
        if (Ia <= Ib) {
            acc_t c[2][NBX];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int J = 0; J < NBX; ++J) c[r][J] = (acc_t){0, 0, 0, 0};
            const bool two = Ia < Ib;
            for (int k = 0; k < nb; ++k) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int col = 16 * k + 4 * ks + kk;
                    const double aa = two ? T1[(size_t)(16 * Ia + li) * Mp + col] : 0.0, ab = T1[(size_t)(16 * Ib + li) * Mp + col];
                    double bj[NBX];
#pragma unroll
                    for (int J = 0; J < NBX; ++J) bj[J] = J <= Ib ? sym_at(KI, Mp, 16 * J + li, col) : 0.0;
#pragma unroll
                    for (int J = 0; J < NBX; ++J)
                        if (J <= Ib) {
                            if (J <= Ia) c[0][J] = Mfma<double>::mma(aa, bj[J], c[0][J]);
                            c[1][J] = Mfma<double>::mma(ab, bj[J], c[1][J]);
                        }
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int J = 0; J < NBX; ++J) {
                    const int I = r ? Ib : Ia;
                    if (J > I || (r == 0 && !two)) continue;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int i = 16 * I + Mfma<double>::row(lane, v), j = 16 * J + li;
                        double gp = 0.0, wk = 0.0;
                        if (j <= i && i < M) {
                            const double ki = KI[(size_t)i * Mp + j], bi = binv_at(tiles, nb, i, j), p = Pd[(size_t)i * Mp + j];
                            const double k0 = K0[(size_t)i * Mp + j] - (i == j ? jitter : 0.0);
                            const double gb = -0.5 * bi - 0.5 * be * be * ww[i] * ww[j];
                            const double gk = 0.5 * ki - 0.5 * be * c[r][J][v] + gb;
                            gp = 0.5 * be * ki + be * gb;
                            wk = gk * k0;
                            const double mult = (i == j) ? 1.0 : 2.0;
                            sK += mult * wk;
                            sP += mult * gp * p;
                            sGBP += mult * gb * p;
                            tr += mult * ki * p;
                        }
                        GP[(size_t)d * Mp * Mp + (size_t)i * Mp + j] = gp;
                        WK[(size_t)d * Mp * Mp + (size_t)i * Mp + j] = wk;
                    }
                }
        }
    }
    sK = block_sum(sK, scratch);
    sP = block_sum(sP, scratch);
    sGBP = block_sum(sGBP, scratch);
    tr = block_sum(tr, scratch);
    double yy = 0.0;
    for (int k = t; k < DPGP_YY_NCH; k += 256) yy += yy_part[(size_t)k * D + d];
    yy = block_sum(yy, scratch);
    __syncthreads();
    if (t < Mp) Gv[(size_t)d * Mp + t] = (t < M) ? be * be * ww[t] : 0.0;
    if (t == 0) {
        dab[2 * d] = -0.5 * be * N + (sK + 2.0 * sP + be * be * vw) / al;
        dab[2 * d + 1] = 0.5 * N / be + 0.5 * (tr - al * N) - 0.5 * yy + be * vw + sGBP;
        info[d] = fail;
    }
}

template <typename TP>
int launch_chain_grad(int D, int N, int M, const TP *psi2_part, int ns2, const double *v_part, int ns1, const double *alpha,
                      const double *beta, const double *yy_part, double jitter, double *ws, double *GP, double *WK,
                      double *Gv, double *dab, int *info, hipStream_t st) {
    const int Mp = dpgp_round_up(M, 16);
    if (!chain_k_resident(Mp, sizeof(double))) return -30;          // first version: LDS-resident sizes only (M <= 128)
    const size_t lds = chain_k_resident_bytes(Mp, sizeof(double));
    auto kern = chain_grad_kernel<TP>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3(D), dim3(256), lds, st, D, N, M, Mp, psi2_part, ns2, v_part, ns1, alpha, beta, yy_part,
                       jitter, ws, la_chain_ws_elems(M), GP, WK, Gv, dab, info);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
template int launch_chain_grad<float>(int, int, int, const float *, int, const double *, int, const double *, const double *,
                                      const double *, double, double *, double *, double *, double *, double *, int *,
                                      hipStream_t);
template int launch_chain_grad<double>(int, int, int, const double *, int, const double *, int, const double *,
                                       const double *, const double *, double, double *, double *, double *, double *,
                                       double *, int *, hipStream_t);

// ---------------------------------------------------------------------------------------------------------------
// Stage B (psi_grad_kernel): second streaming pass over the observations, first version (plain VALU, no symmetry saving).
// With G = d f_hat / d Psi2_d (symmetric), g = d f_hat / d (Psi1_d^T y_d), W = (d f_hat / d K_uu,d) .* (K_uu,d - jitter I):
//   w [n,a,m'] = G[a,m'] psi2(n,a,m'),     R = sum_m' w,  T_q = sum_m' w z_m'q             (rbf_kernel.py:164-199)
//   w1[n,a]    = g[a] y_nd psi1(n,a)                                                        (rbf_kernel.py:135-161)
//   wk[a,m']   = W[a,m'],                 R_K = sum_m' wk, T_K,q = sum_m' wk z_m'q          (rbf_kernel.py:58-93)
// thread = (row a of the M x M statistics, one of two observations); per observation the sums over a of
// (R, R z_a, R z_a^2, z_a T) and (w1, w1 z_a, w1 z_a^2) give d/dmu_n, d/dS_n and the per-observation part of d/dgamma_d;
// d/dz_a accumulates thread-locally over the observations.  Partial results per (output dim, n-split) are summed by
// grad_reduce_kernel in a fixed order (deterministic).  alpha enters only as a factor (its derivative is stage A's).
//   log psi2 = 2 log alpha - sum_q [ 1/2 log den2 + 1/4 gamma (z_a - z_m')^2 + gamma (mu - (z_a + z_m')/2)^2 / den2 ],  den2 = 2 gamma S + 1
//   log psi1 =   log alpha - 1/2 sum_q [ log den1 + gamma (mu - z_a)^2 / den1 ],                                       den1 = gamma S + 1
// ---------------------------------------------------------------------------------------------------------------
#define PG_RED_ELEMS(Q) ((size_t)1024)       // wave sums [4][5 Q + 2] <= 608; 256 doubles of column-mean scratch
__device__ __forceinline__ float pg_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ double pg_exp(double x) { return exp(x); }
// QP: latent dims padded to a multiple of 4 (compile time, so that the per-thread q-arrays stay in registers)
template <typename TC, int QP>
__global__ __launch_bounds__(256, 2) void psi_grad_kernel(int D, int N, int M, int Mp, int Q, const double *__restrict__ y, int ldy,
                                                       const double *__restrict__ z, const double *__restrict__ mu,
                                                       const double *__restrict__ s, const double *__restrict__ gamma,
                                                       const double *__restrict__ alpha, const double *__restrict__ GP,
                                                       const double *__restrict__ WK, const double *__restrict__ Gv,
                                                       int n_per_split, int do_psi2, double *__restrict__ dmu_part,
                                                       double *__restrict__ ds_part, double *__restrict__ dz_part,
                                                       double *__restrict__ dg_part) {
    // do_psi2 == 0: the Psi2 and Psi1 terms are left to psi2_grad_kernel (psi2.hip, matrix pipe) and psi1_grad_*_kernel; this
    // kernel then only does the K_uu term (no pass over the observations, d/dmu and d/dS partials not written)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    TC *gs = reinterpret_cast<TC *>(smem_raw);                 // [Mp][Mp] symmetric G (then W for the K_uu part; [2 Q][128] at the very end)
    TC *zs = gs + (size_t)(Mp * Mp > 2 * Q * 128 ? Mp * Mp : 2 * Q * 128);                       // [Mp][QP], zero padded
    TC *red = zs + (size_t)Mp * QP;                             // scratch: wave sums [4][5 Q + 2]; 256 doubles (column means)
    TC *nq = red + PG_RED_ELEMS(Q);              // [2][6][Q] per-observation per-q factors
    TC *pbuf = nq + (size_t)12 * Q;                            // [2][128] per-observation P[n, m] (see below)
    TC *zcs = pbuf + 256;                                      // [32] column means of z (everything works on centred z, mu)
    TC *rsum = zcs + 32;                                       // [2][7 Q + 2] sums over the rows of `red`
    const int d = blockIdx.x, sp = blockIdx.y, nsplit = gridDim.y, t = threadIdx.x, a = t & 127, nl = t >> 7;
    const int NV = 5 * Q + 2;
    const TC al = (TC)alpha[d];
    if (do_psi2)
        for (int e = t; e < Mp * Mp; e += 256) {
            const int i = e / Mp, j = e - i * Mp;
            const double v = (i < M && j < M) ? GP[(size_t)d * Mp * Mp + (size_t)(i >= j ? i : j) * Mp + (i >= j ? j : i)] : 0.0;
            gs[i * Mp + j] = (TC)v;
        }
    {   // column means (fp64 partial sums through the `red` area, which is free here)
        double *sc = reinterpret_cast<double *>(red);
        const int q = t & 31, rg = t >> 5;
        double acc = 0.0;
        if (q < Q)
            for (int m_ = rg; m_ < M; m_ += 8) acc += z[(size_t)m_ * Q + q];
        sc[rg * 32 + q] = acc;
        __syncthreads();
        if (t < 32) {
            double v = 0.0;
            for (int k = 0; k < 8; ++k) v += sc[k * 32 + t];
            zcs[t] = (t < Q) ? (TC)(v / (double)M) : (TC)0;
        }
        __syncthreads();
    }
    for (int e = t; e < Mp * QP; e += 256) {
        const int i = e / QP, q = e - i * QP;
        zs[e] = (i < M && q < Q) ? (TC)(z[(size_t)i * Q + q] - (double)zcs[q]) : (TC)0;
    }
    __syncthreads();
    TC ga[QP], za[QP], dza[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        ga[q] = (q < Q) ? (TC)gamma[(size_t)d * Q + q] : (TC)0;
        za[q] = (a < M) ? zs[a * QP + q] : (TC)0;
        dza[q] = 0;
    }
    const TC gva = (a < M) ? (TC)Gv[(size_t)d * Mp + a] : (TC)0;
    TC dg_mine = 0;
    const int nbeg = sp * n_per_split, nend = do_psi2 ? min(N, nbeg + n_per_split) : nbeg;
    for (int n0 = nbeg; n0 < nend; n0 += 2) {
        const int n = n0 + nl;
        const bool live = (n < nend);
        // per-observation factors, one thread per (observation, q): a2 = gamma / den2, a1 = gamma / den1, ...
        if (t < 2 * Q) {
            const int l = t / Q, q = t - l * Q, nn = n0 + l;
            TC *o = nq + (l * 6) * Q + q;
            if (nn < nend) {
                const TC g = (TC)gamma[(size_t)d * Q + q], ss = (TC)s[(size_t)nn * Q + q];
                const TC m_ = (TC)(mu[(size_t)nn * Q + q] - (double)zcs[q]);
                const TC den2 = 2 * g * ss + 1, den1 = g * ss + 1;
                o[0] = g / den2; o[Q] = g / den1; o[2 * Q] = m_; o[3 * Q] = ss;
                o[4 * Q] = dpgp_log(den2); o[5 * Q] = dpgp_log(den1);
            } else {
                o[0] = 0; o[Q] = 0; o[2 * Q] = 0; o[3 * Q] = 1; o[4 * Q] = 0; o[5 * Q] = 0;
            }
        }
        __syncthreads();
        const TC *f = nq + (nl * 6) * Q;
        TC a2[QP], mq[QP];
        TC l2 = 0, l1 = 0;
        TC a1v[QP];
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            const bool in = q < Q;
            a2[q] = in ? f[q] : (TC)0; a1v[q] = in ? f[Q + q] : (TC)0; mq[q] = in ? f[2 * Q + q] : (TC)0;
            l2 += in ? f[4 * Q + q] : (TC)0; l1 += in ? f[5 * Q + q] : (TC)0;
        }
        // log psi2(n,a,m') = P[a] + P[m'] + sum_q X_q z_aq z_m'q with (centred z, mu)
        //   X_q = (gamma_q - a2_q) / 2,   P[m] = const_n / 2 - sum_q ( (gamma_q + a2_q) z_mq^2 / 4 - a2_q mu_q z_mq ),
        //   const_n = 2 log alpha - 1/2 sum_q log den2_q - sum_q a2_q mu_q^2
        // (the expansion the forward kernel uses): the inner loop is one dot product per (a, m') instead of ~7 Q flops
        TC cn = 2 * dpgp_log(al) - (TC)0.5 * l2, pa = 0, xz[QP];
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            cn -= a2[q] * mq[q] * mq[q];
            pa -= (TC)0.25 * (ga[q] + a2[q]) * za[q] * za[q] - a2[q] * mq[q] * za[q];
            xz[q] = (TC)0.5 * (ga[q] - a2[q]) * za[q];
        }
        pa += (TC)0.5 * cn;
        if (do_psi2) {
            pbuf[nl * 128 + a] = pa;
            __syncthreads();
        }
        TC R = 0, T[QP];
#pragma unroll
        for (int q = 0; q < QP; ++q) T[q] = 0;
        TC w1 = 0;
        if (live && a < M) {
            if (do_psi2)
              for (int mp = 0; mp < M; ++mp) {
                TC e = pa + pbuf[nl * 128 + mp], zm[QP];
#pragma unroll
                for (int q = 0; q < QP; ++q) {
                    zm[q] = zs[mp * QP + q];
                    e += xz[q] * zm[q];
                }
                const TC w = gs[mp * Mp + a] * pg_exp(e);
                R += w;
#pragma unroll
                for (int q = 0; q < QP; ++q) T[q] += w * zm[q];
            }
            TC e1 = dpgp_log(al) - (TC)0.5 * l1;
#pragma unroll
            for (int q = 0; q < QP; ++q) { const TC c = mq[q] - za[q]; e1 -= (TC)0.5 * a1v[q] * c * c; }
            w1 = gva * (TC)y[(size_t)n * ldy + d] * pg_exp(e1);
#pragma unroll
            for (int q = 0; q < QP; ++q) {
                // gamma (1 +- 1 / den2) = gamma +- a2
                dza[q] += -(ga[q] + a2[q]) * za[q] * R + (ga[q] - a2[q]) * T[q] + 2 * a2[q] * mq[q] * R
                          + w1 * a1v[q] * (mq[q] - za[q]);
            }
        }
        // sums over the rows a of [R | R z | R z^2 | z T | w1 | w1 z | w1 z^2]: wave-level shuffles, then the two waves of
        // an observation through a few LDS words (staging all of it in LDS cost 53 KB and a second workgroup per CU)
        {
            const int wave = t >> 6, lane_ = t & 63;
            auto put = [&](int v, TC x) {
                x = wave_sum(x);
                if (lane_ == 0) red[wave * NV + v] = x;
            };
            if (do_psi2) put(0, R);
            else if (lane_ == 0) red[wave * NV] = 0;
            put(3 * Q + 1, w1);
#pragma unroll
            for (int q = 0; q < QP; ++q)
                if (q < Q) {
                    if (do_psi2) {
                        put(1 + q, R * za[q]);
                        put(1 + Q + q, R * za[q] * za[q]);
                        put(1 + 2 * Q + q, za[q] * T[q]);
                    } else if (lane_ == 0) {
                        red[wave * NV + 1 + q] = 0; red[wave * NV + 1 + Q + q] = 0; red[wave * NV + 1 + 2 * Q + q] = 0;
                    }
                    put(3 * Q + 2 + q, w1 * za[q]);
                    put(4 * Q + 2 + q, w1 * za[q] * za[q]);
                }
        }
        __syncthreads();
        for (int e = t; e < 2 * NV; e += 256) {
            const int l = e / NV, v = e - l * NV;
            rsum[l * NV + v] = red[(2 * l) * NV + v] + red[(2 * l + 1) * NV + v];
        }
        __syncthreads();
        if (t < 2 * Q) {
            const int l = t / Q, q = t - l * Q, nn = n0 + l;
            if (nn < nend) {
                const TC *r0 = rsum + (size_t)l * NV;
                const TC S0 = r0[0], S1 = r0[1 + q], S2 = r0[1 + Q + q], S3 = r0[1 + 2 * Q + q], V0 = r0[3 * Q + 1],
                         V1 = r0[3 * Q + 2 + q], V2 = r0[4 * Q + 2 + q];
                const TC *o = nq + (l * 6) * Q + q;
                const TC a2_ = o[0], a1_ = o[Q], m_ = o[2 * Q], ss = o[3 * Q], g = (TC)gamma[(size_t)d * Q + q];
                const TC id2 = a2_ / g, id1 = a1_ / g;                       // 1 / den2, 1 / den1
                const TC A2 = (TC)0.5 * (S2 + S3);
                const TC q2 = m_ * m_ * S0 - 2 * m_ * S1 + A2;               // sum w (mu - zbar)^2
                const TC q1 = m_ * m_ * V0 - 2 * m_ * V1 + V2;               // sum w1 (mu - z)^2
                const TC dmu = -2 * a2_ * (m_ * S0 - S1) - a1_ * (m_ * V0 - V1);
                const TC dss = -a2_ * S0 + 2 * a2_ * a2_ * q2 + (TC)0.5 * (a1_ * a1_ * q1 - a1_ * V0);
                dmu_part[((size_t)d * N + nn) * Q + q] = (double)dmu;
                ds_part[((size_t)d * N + nn) * Q + q] = (double)dss;
                dg_mine += -ss * id2 * S0 - q2 * id2 * id2 - (TC)0.5 * (S2 - S3) - (TC)0.5 * (q1 * id1 * id1 + ss * id1 * V0);
            }
        }
        __syncthreads();
    }
    // K_uu part, once per output dim (split 0): wk = W (symmetric), no dependence on the observations
    TC dgk = 0;
    if (sp == 0) {
        __syncthreads();
        for (int e = t; e < Mp * Mp; e += 256) {
            const int i = e / Mp, j = e - i * Mp;
            const double v = (i < M && j < M) ? WK[(size_t)d * Mp * Mp + (size_t)(i >= j ? i : j) * Mp + (i >= j ? j : i)] : 0.0;
            gs[i * Mp + j] = (TC)v;
        }
        __syncthreads();
        TC RK = 0, TK[QP];
#pragma unroll
        for (int q = 0; q < QP; ++q) TK[q] = 0;
        if (nl == 0 && a < M) {
            for (int mp = 0; mp < M; ++mp) {
                const TC w = gs[mp * Mp + a];
                RK += w;
#pragma unroll
                for (int q = 0; q < QP; ++q) TK[q] += w * zs[mp * QP + q];
            }
#pragma unroll
            for (int q = 0; q < QP; ++q) dza[q] += -2 * ga[q] * (za[q] * RK - TK[q]);
        }
        {
            const int wave = t >> 6, lane_ = t & 63;
#pragma unroll
            for (int q = 0; q < QP; ++q)
                if (q < Q) {
                    TC x = (nl == 0 && a < M) ? (RK * za[q] * za[q] - za[q] * TK[q]) : (TC)0;
                    x = wave_sum(x);
                    if (lane_ == 0) red[wave * Q + q] = x;
                }
        }
        __syncthreads();
        if (t < Q) {
            const TC v = red[t] + red[Q + t];                                // (waves 0 and 1 hold the rows; 2 and 3 wrote zeros)
            dgk = -v;                                                        // -1/2 sum wk (z_a - z_m')^2 = -(S2 - S3)
        }
        __syncthreads();
    }
    // d/dgamma: threads (l, q) hold their observations' share; combine the two observation lanes through LDS
    if (t < 2 * Q) red[t] = dg_mine;
    __syncthreads();
    if (t < Q) dg_part[((size_t)sp * D + d) * Q + t] = (double)(red[t] + red[Q + t] + dgk);
    // d/dz: the two observation lanes of row a
    __syncthreads();
    if (a < M) {
#pragma unroll
        for (int q = 0; q < QP; ++q)
            if (q < Q) gs[(size_t)(nl * Q + q) * 128 + a] = dza[q];
    }
    __syncthreads();
    if (nl == 0 && a < M)
        for (int q = 0; q < Q; ++q)
            dz_part[(((size_t)d * nsplit + sp) * M + a) * Q + q] = (double)(gs[(size_t)q * 128 + a] + gs[(size_t)(Q + q) * 128 + a]);
}

// out[i] = sum_k part[k * n + i], k < nk, fixed order
__global__ __launch_bounds__(256) void grad_reduce_kernel(size_t n, int nk, const double *__restrict__ part, double *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double a = 0.0;
    for (int k = 0; k < nk; ++k) a += part[(size_t)k * n + i];
    out[i] = a;
}

// out[i] (+)= sum_k part[k * pitch + i], i < n, k < nk, in a fixed order; many partial rows go through `stage`
// (reduce_rows_stage_elems(n) doubles): <= 64 chunks of rows are summed side by side, then the chunk sums.
template <typename TP>
__global__ __launch_bounds__(256) void reduce_rows_kernel(size_t n, size_t pitch, int nk, int kchunk, const TP *__restrict__ part,
                                                          double *__restrict__ out, int accumulate) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = blockIdx.y, k0 = c * kchunk, k1 = min(nk, k0 + kchunk);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int k = k0;
    for (; k + 3 < k1; k += 4) {
        a0 += (double)part[(size_t)k * pitch + i];
        a1 += (double)part[(size_t)(k + 1) * pitch + i];
        a2 += (double)part[(size_t)(k + 2) * pitch + i];
        a3 += (double)part[(size_t)(k + 3) * pitch + i];
    }
    for (; k < k1; ++k) a0 += (double)part[(size_t)k * pitch + i];
    const double a = (a0 + a1) + (a2 + a3);
    double *o = out + (size_t)c * n + i;
    *o = accumulate ? *o + a : a;
}
size_t reduce_rows_stage_elems(size_t n) { return 64 * n; }
template <typename TP>
int launch_reduce_rows(size_t n, size_t pitch, int nk, const TP *part, double *out, int accumulate, double *stage, hipStream_t st) {
    const unsigned gx = (unsigned)((n + 255) / 256);
    if (nk <= 96 || !stage) {
        DPGP_PRELAUNCH(); hipLaunchKernelGGL(reduce_rows_kernel<TP>, dim3(gx, 1), dim3(256), 0, st, n, pitch, nk, nk, part, out, accumulate);
        DPGP_LAUNCH_CHECK();
        return DPGP_OK;
    }
    const int kchunk = nk > 64 * 64 ? dpgp_ceil_div(nk, 64) : 64, nch = dpgp_ceil_div(nk, kchunk);
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(reduce_rows_kernel<TP>, dim3(gx, nch), dim3(256), 0, st, n, pitch, nk, kchunk, part, stage, 0);
    DPGP_LAUNCH_CHECK();
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(reduce_rows_kernel<double>, dim3(gx, 1), dim3(256), 0, st, n, n, nch, nch, (const double *)stage, out, accumulate);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
template int launch_reduce_rows<float>(size_t, size_t, int, const float *, double *, int, double *, hipStream_t);
template int launch_reduce_rows<double>(size_t, size_t, int, const double *, double *, int, double *, hipStream_t);

// ---------------------------------------------------------------------------------------------------------------
// Psi1 term of stage B without any per-observation workgroup reduction (mixed precision; the first version above spends
// most of its time in those).  w1[n,a] = g_d[a] y_nd psi1_d(n,a),  c = mu_n - z_a (both centred),  a1 = gamma / (gamma S + 1):
//   pass n (thread = observation, loops over a chunk of output dims and all a):   V0 = sum_a w1, U1_q = sum_a w1 c_q,
//       U2_q = sum_a w1 c_q^2  ->  d/dmu_nq -= a1 U1,  d/dS_nq += (a1^2 U2 - a1 V0) / 2,  d/dgamma_dq -= (U2 / den1^2 + S V0 / den1) / 2
//   pass z (thread = inducing point a, loops over a chunk of output dims and its observations):  d/dz_aq += w1 a1 c_q
// both recompute psi1 (D N M exponentials each, ~0.1 ms), z rows come centred from the psi2 constants.     (rbf_kernel.py:135-161)
// ---------------------------------------------------------------------------------------------------------------
template <int QP>
__global__ __launch_bounds__(256) void psi1_grad_n_kernel(int D, int N, int M, int Q, const double *__restrict__ y, int ldy,
                                                          const unsigned char *__restrict__ consts, const double *__restrict__ mu,
                                                          const double *__restrict__ s, const double *__restrict__ gamma,
                                                          const double *__restrict__ alpha, const double *__restrict__ Gv,
                                                          const double *__restrict__ G1, int Mp,
                                                          int d_per_wg, float *__restrict__ dmu_part, float *__restrict__ ds_part,
                                                          double *__restrict__ dg_part) {
    // G1 != nullptr: a full adjoint G1[d][n][Mp] of Psi1_d takes the place of the rank-1 form g_d[a] y_nd (over-T model)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float *zs = reinterpret_cast<float *>(smem_raw);            // [M][QP] centred
    float *gv = zs + (size_t)M * QP;                            // [M] g_d
    float *red = gv + Mp;                                       // [4][QP]
    const Psi2Consts C = psi2_consts_layout(M, Q);
    const float *zs_g = reinterpret_cast<const float *>(consts + C.off_zs), *zc = reinterpret_cast<const float *>(consts);
    const int t = threadIdx.x, n = blockIdx.x * 256 + t, lane = t & 63, wave = t >> 6;
    const bool live = n < N;
    for (int e = t; e < M * QP; e += 256) {
        const int a = e / QP, q = e - a * QP;
        zs[e] = (q < C.ZLD) ? zs_g[(size_t)a * C.ZLD + q] : 0.0f;
    }
    float mq[QP], sv[QP], dmu[QP], dss[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        mq[q] = (live && q < Q) ? (float)mu[(size_t)n * Q + q] - zc[q] : 0.0f;
        sv[q] = (live && q < Q) ? (float)s[(size_t)n * Q + q] : 1.0f;
        dmu[q] = 0.0f;
        dss[q] = 0.0f;
    }
    const int d0 = blockIdx.y * d_per_wg, d1 = min(D, d0 + d_per_wg);
    for (int d = d0; d < d1; ++d) {
        __syncthreads();
        for (int a = t; a < M; a += 256) gv[a] = G1 ? 1.0f : (float)Gv[(size_t)d * Mp + a];
        __syncthreads();
        float a1[QP], l1 = 0.0f;
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            const float g = (q < Q) ? (float)gamma[(size_t)d * Q + q] : 0.0f, den = g * sv[q] + 1.0f;
            a1[q] = g / den;
            l1 += __builtin_amdgcn_logf(den);                     // log2
        }
        const float al = (float)alpha[d];
        const float e0 = __builtin_amdgcn_logf(al) - 0.5f * l1;  // log2 units
        float V0 = 0.0f, U1[QP], U2[QP], h[QP];
#pragma unroll
        for (int q = 0; q < QP; ++q) { U1[q] = 0.0f; U2[q] = 0.0f; h[q] = (float)(-0.5 * DPGP_LOG2E) * a1[q]; }
        for (int a = 0; a < M; ++a) {
            float c[QP], cc[QP], e = e0;
#pragma unroll
            for (int q4 = 0; q4 < QP / 4; ++q4) {
                const f32x4 zv = *reinterpret_cast<const f32x4 *>(zs + a * QP + 4 * q4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int q = 4 * q4 + k;
                    c[q] = mq[q] - zv[k];
                    cc[q] = c[q] * c[q];
                    e += h[q] * cc[q];
                }
            }
            const float w = (G1 ? (live ? (float)G1[((size_t)d * N + n) * Mp + a] : 0.0f) : gv[a]) * dpgp_exp2(e);
            V0 += w;
#pragma unroll
            for (int q = 0; q < QP; ++q) { U1[q] += w * c[q]; U2[q] += w * cc[q]; }
        }
        const float yv = G1 ? 1.0f : (live ? (float)y[(size_t)n * ldy + d] : 0.0f);
        V0 *= yv;
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            const float u1 = yv * U1[q], u2 = yv * U2[q];
            dmu[q] -= a1[q] * u1;
            dss[q] += 0.5f * (a1[q] * a1[q] * u2 - a1[q] * V0);
            const float id1 = 1.0f / ((q < Q ? (float)gamma[(size_t)d * Q + q] : 0.0f) * sv[q] + 1.0f);
            float dg = -0.5f * (u2 * id1 * id1 + sv[q] * id1 * V0);
            dg = wave_sum(dg);
            if (lane == 0) red[wave * QP + q] = dg;
        }
        __syncthreads();
        if (t < Q) dg_part[((size_t)blockIdx.x * D + d) * Q + t] = (double)(red[t] + red[QP + t] + red[2 * QP + t] + red[3 * QP + t]);
    }
    if (live)
#pragma unroll
        for (int q = 0; q < QP; ++q)
            if (q < Q) {
                dmu_part[((size_t)blockIdx.y * N + n) * Q + q] = dmu[q];
                ds_part[((size_t)blockIdx.y * N + n) * Q + q] = dss[q];
            }
}

// AL = lanes over the inducing points (64 for M <= 64, else 128); the 256 / AL groups of lanes share the observations
template <int QP, int AL>
__global__ __launch_bounds__(256) void psi1_grad_z_kernel(int D, int N, int M, int Q, const double *__restrict__ y, int ldy,
                                                          const unsigned char *__restrict__ consts, const double *__restrict__ mu,
                                                          const double *__restrict__ s, const double *__restrict__ gamma,
                                                          const double *__restrict__ alpha, const double *__restrict__ Gv,
                                                          const double *__restrict__ G1, int Mp,
                                                          int d_per_wg, int n_per_split, double *__restrict__ dz_part) {
    constexpr int FS = 2 * QP + 4;                              // per observation: a1[QP], mu'[QP], e0, y, pad
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float *fac = reinterpret_cast<float *>(smem_raw);           // [64][FS]
    constexpr int NLN = 256 / AL;
    float *comb = fac + 64 * FS;                                // [NLN - 1][AL][QP] (end: the observation lanes of a row)
    const Psi2Consts C = psi2_consts_layout(M, Q);
    const float *zs_g = reinterpret_cast<const float *>(consts + C.off_zs), *zc = reinterpret_cast<const float *>(consts);
    const int t = threadIdx.x, a = blockIdx.z * AL + t % AL, al_ = t % AL, nl = t / AL;
    float za[QP], dza[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        za[q] = (a < M && q < C.ZLD) ? zs_g[(size_t)a * C.ZLD + q] : 0.0f;
        dza[q] = 0.0f;
    }
    const int d0 = blockIdx.y * d_per_wg, d1 = min(D, d0 + d_per_wg);
    const int nbeg = blockIdx.x * n_per_split, nend = min(N, nbeg + n_per_split);
    const int fn = t >> 2, fp = t & 3;                          // factor computation: observation fn of the chunk, q = fp, fp + 4, ...
    for (int d = d0; d < d1; ++d) {
        const float gva = (a < M) ? (G1 ? 1.0f : (float)Gv[(size_t)d * Mp + a]) : 0.0f;
        const float l2al = __builtin_amdgcn_logf((float)alpha[d]);
        for (int nc = nbeg; nc < nend; nc += 64) {
            __syncthreads();
            {
                const int n = nc + fn;
                float l1 = 0.0f;
                for (int q = fp; q < QP; q += 4) {
                    float a1 = 0.0f, m_ = 0.0f;
                    if (q < Q && n < nend) {
                        const float g = (float)gamma[(size_t)d * Q + q], den = g * (float)s[(size_t)n * Q + q] + 1.0f;
                        a1 = g / den;
                        m_ = (float)mu[(size_t)n * Q + q] - zc[q];
                        l1 += __builtin_amdgcn_logf(den);
                    }
                    fac[fn * FS + q] = a1;
                    fac[fn * FS + QP + q] = m_;
                }
                l1 += __shfl_xor(l1, 1, 64);
                l1 += __shfl_xor(l1, 2, 64);
                if (fp == 0) {
                    fac[fn * FS + 2 * QP] = l2al - 0.5f * l1;
                    fac[fn * FS + 2 * QP + 1] = (n < nend) ? (G1 ? 1.0f : (float)y[(size_t)n * ldy + d]) : 0.0f;
                }
            }
            __syncthreads();
            for (int i = nl; i < 64; i += NLN) {
                const float *f = fac + i * FS;
                float c[QP], a1[QP], e = f[2 * QP];
#pragma unroll
                for (int q4 = 0; q4 < QP / 4; ++q4) {
                    const f32x4 av = *reinterpret_cast<const f32x4 *>(f + 4 * q4), mv = *reinterpret_cast<const f32x4 *>(f + QP + 4 * q4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int q = 4 * q4 + k;
                        a1[q] = av[k];
                        c[q] = mv[k] - za[q];
                        e += (float)(-0.5 * DPGP_LOG2E) * a1[q] * c[q] * c[q];
                    }
                }
                float w = gva * f[2 * QP + 1] * dpgp_exp2(e);
                if (G1) w *= (nc + i < nend && a < M) ? (float)G1[((size_t)d * N + nc + i) * Mp + a] : 0.0f;
#pragma unroll
                for (int q = 0; q < QP; ++q) dza[q] += w * a1[q] * c[q];
            }
        }
    }
    __syncthreads();
    if (nl > 0)
#pragma unroll
        for (int q = 0; q < QP; ++q) comb[((nl - 1) * AL + al_) * QP + q] = dza[q];
    __syncthreads();
    if (nl == 0 && a < M)
#pragma unroll
        for (int q = 0; q < QP; ++q)
            if (q < Q) {
                float v = dza[q];
                for (int k = 0; k < NLN - 1; ++k) v += comb[(k * AL + al_) * QP + q];
                dz_part[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * M + a) * Q + q] = (double)v;
            }
}

// ---------------------------------------------------------------------------------------------------------------
// K_uu term of stage B for any M (the plain kernel above holds one row per thread, M <= 128): wk = (df/dK_uu) .* (K_uu - jitter I)
// symmetric,  R[a] = sum_m' wk[a,m'],  T[a,q] = sum_m' wk[a,m'] z_m'q:   d/dz_aq = -2 gamma_q (z_aq R - T),
// d/dgamma_q = -sum_a (R z_aq^2 - z_aq T).   Workgroup = (64 rows a, output dim), 4 lane groups share the columns m'.
// ---------------------------------------------------------------------------------------------------------------
template <int QP>
__global__ __launch_bounds__(256) void kuu_grad_kernel(int D, int M, int Mp, int Q, const unsigned char *__restrict__ consts,
                                                       const double *__restrict__ gamma, const double *__restrict__ WK,
                                                       double *__restrict__ dz_part, double *__restrict__ dg_part) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float *zs = reinterpret_cast<float *>(smem_raw);            // [M][QP] centred
    float *comb = zs + (size_t)M * QP;                          // [3][64][QP + 1]
    float *red = comb + 3 * 64 * (QP + 1);                      // [QP]
    const Psi2Consts C = psi2_consts_layout(M, Q);
    const float *zs_g = reinterpret_cast<const float *>(consts + C.off_zs);
    const int t = threadIdx.x, al = t & 63, part = t >> 6, a = blockIdx.x * 64 + al, d = blockIdx.y;
    for (int e = t; e < M * QP; e += 256) {
        const int r = e / QP, q = e - r * QP;
        zs[e] = (q < C.ZLD) ? zs_g[(size_t)r * C.ZLD + q] : 0.0f;
    }
    if (t < QP) red[t] = 0.0f;
    __syncthreads();
    const double *W = WK + (size_t)d * Mp * Mp;
    float R = 0.0f, T[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) T[q] = 0.0f;
    if (a < M)
        for (int mp = part; mp < M; mp += 4) {
            const float w = (float)((a >= mp) ? W[(size_t)a * Mp + mp] : W[(size_t)mp * Mp + a]);
            R += w;
#pragma unroll
            for (int q = 0; q < QP; ++q) T[q] += w * zs[mp * QP + q];
        }
    if (part > 0) {
        float *c = comb + ((part - 1) * 64 + al) * (QP + 1);
        c[QP] = R;
#pragma unroll
        for (int q = 0; q < QP; ++q) c[q] = T[q];
    }
    __syncthreads();
    if (part == 0) {
        for (int k = 0; k < 3; ++k) {
            const float *c = comb + (k * 64 + al) * (QP + 1);
            R += c[QP];
#pragma unroll
            for (int q = 0; q < QP; ++q) T[q] += c[q];
        }
#pragma unroll
        for (int q = 0; q < QP; ++q) {
            const float za = (a < M) ? zs[a * QP + q] : 0.0f;
            const float g = (q < Q) ? (float)gamma[(size_t)d * Q + q] : 0.0f;
            if (a < M && q < Q) dz_part[((size_t)d * M + a) * Q + q] = (double)(-2.0f * g * (za * R - T[q]));
            float x = (a < M) ? -(R * za * za - za * T[q]) : 0.0f;
            x = wave_sum(x);
            if (al == 0) red[q] = x;
        }
    }
    __syncthreads();
    if (t < Q) dg_part[((size_t)blockIdx.x * D + d) * Q + t] = (double)red[t];
}
// dz [M,Q], dgamma [D,Q] overwritten.  ws: D M Q + ceil(M / 64) D Q doubles
int launch_kuu_grad(int D, int M, int Q, const unsigned char *consts, const double *gamma, const double *WK, double *ws,
                    double *stage, double *dz, double *dgamma, hipStream_t st) {
    const int Mp = dpgp_round_up(M, 16), nblk = dpgp_ceil_div(M, 64), QPr = 4 * dpgp_ceil_div(Q, 4);
    double *dz_part = ws, *dg_part = ws + (size_t)D * M * Q;
    void (*k)(int, int, int, int, const unsigned char *, const double *, const double *, double *, double *) = nullptr;
    switch (QPr / 4) {
#define CASE(kk) case kk: k = kuu_grad_kernel<4 * kk>; break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
    if (!k) return -4;
    const size_t lds = sizeof(float) * ((size_t)M * QPr + 3 * 64 * (QPr + 1) + QPr);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(k, dim3(nblk, D), dim3(256), lds, st, D, M, Mp, Q, consts, gamma, WK, dz_part, dg_part);
    DPGP_LAUNCH_CHECK();
    const size_t mq = (size_t)M * Q, dq = (size_t)D * Q;
    int rc = launch_reduce_rows<double>(mq, mq, D, dz_part, dz, 0, stage, st);
    if (rc == DPGP_OK) rc = launch_reduce_rows<double>(dq, dq, nblk, dg_part, dgamma, 0, stage, st);
    return rc;
}

// workspace (doubles) of launch_psi1_grad: d/dmu, d/dS partials [DC][N][Q] float; d/dgamma [NB][D][Q]; d/dz [DC ns][M][Q]
static void psi1_grad_shape(int D, int N, int *dpw, int *DC, int *NB, int *ns, int *nper) {
    *NB = dpgp_ceil_div(N, 256);
    // output dims per workgroup: 8, fewer when few output dims (the over-T model: D = T atoms) would leave the GPU to NB workgroups
    *dpw = (int)((long long)D * *NB / 512);
    if (*dpw > 8) *dpw = 8;
    if (*dpw < 1) *dpw = 1;
    *DC = dpgp_ceil_div(D, *dpw);
    int k = dpgp_ceil_div(1024, *DC);
    if (k > dpgp_ceil_div(N, 128)) k = dpgp_ceil_div(N, 128);
    if (k < 1) k = 1;
    *nper = 64 * dpgp_ceil_div(dpgp_ceil_div(N, k), 64);
    *ns = dpgp_ceil_div(N, *nper);
}
size_t psi1_grad_ws_elems(int D, int N, int M, int Q) {
    int dpw, DC, NB, ns, nper;
    psi1_grad_shape(D, N, &dpw, &DC, &NB, &ns, &nper);
    return (size_t)DC * N * Q + 2 + (size_t)NB * D * Q + (size_t)DC * ns * M * Q;
}
// dmu, ds: overwritten; dz, dgamma: added to
int launch_psi1_grad(int D, int N, int M, int Q, const double *y, int ldy, const unsigned char *consts, const double *mu,
                     const double *s, const double *gamma, const double *alpha, const double *Gv, const double *G1,
                     double *ws, double *stage, double *dmu, double *ds, double *dz, double *dgamma, hipStream_t st) {
    const int Mp = dpgp_round_up(M, 16);
    int dpw, DC, NB, ns, nper;
    psi1_grad_shape(D, N, &dpw, &DC, &NB, &ns, &nper);
    const size_t slab = (size_t)DC * N * Q;
    float *dmu_part = reinterpret_cast<float *>(ws), *ds_part = dmu_part + slab;
    double *dg_part = ws + slab + 2, *dz_part = dg_part + (size_t)NB * D * Q;
    const int QPr = 4 * dpgp_ceil_div(Q, 4);
    void (*kn)(int, int, int, int, const double *, int, const unsigned char *, const double *, const double *, const double *,
               const double *, const double *, const double *, int, int, float *, float *, double *) = nullptr;
    void (*kz)(int, int, int, int, const double *, int, const unsigned char *, const double *, const double *, const double *,
               const double *, const double *, const double *, int, int, int, double *) = nullptr;
    switch (QPr / 4) {
#define CASE(k) case k: kn = psi1_grad_n_kernel<4 * k>; kz = Mp <= 64 ? psi1_grad_z_kernel<4 * k, 64> : psi1_grad_z_kernel<4 * k, 128>; break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
    if (!kn) return -4;
    const size_t lds_n = sizeof(float) * ((size_t)M * QPr + Mp + 4 * QPr), lds_z = sizeof(float) * ((size_t)64 * (2 * QPr + 4) + 192 * QPr);
    if (lds_n > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_n) != hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kn, dim3(NB, DC), dim3(256), lds_n, st, D, N, M, Q, y, ldy, consts, mu, s, gamma, alpha, Gv, G1, Mp, dpw,
                       dmu_part, ds_part, dg_part);
    DPGP_LAUNCH_CHECK();
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kz, dim3(ns, DC, Mp <= 64 ? 1 : dpgp_ceil_div(M, 128)), dim3(256), lds_z, st, D, N, M, Q, y, ldy, consts, mu, s, gamma, alpha, Gv, G1, Mp, dpw,
                       nper, dz_part);
    DPGP_LAUNCH_CHECK();
    const size_t nq = (size_t)N * Q, mq = (size_t)M * Q, dq = (size_t)D * Q;
    int rc = launch_reduce_rows<float>(nq, nq, DC, dmu_part, dmu, 0, stage, st);
    if (rc == DPGP_OK) rc = launch_reduce_rows<float>(nq, nq, DC, ds_part, ds, 0, stage, st);
    if (rc == DPGP_OK) rc = launch_reduce_rows<double>(dq, dq, NB, dg_part, dgamma, 1, stage, st);
    if (rc == DPGP_OK) rc = launch_reduce_rows<double>(mq, mq, DC * ns, dz_part, dz, 1, stage, st);
    return rc;
}

size_t psi_grad_ws_bytes(int D, int N, int M, int Q, int *nsplit_out) {
    int ns = dpgp_ceil_div(1024, D);
    if (ns > dpgp_ceil_div(N, 32)) ns = dpgp_ceil_div(N, 32);
    if (ns < 1) ns = 1;
    if (nsplit_out) *nsplit_out = ns;
    const size_t rows = (size_t)2 * D * N * Q + (size_t)D * ns * M * Q + (size_t)ns * D * Q;
    // the K_uu term of the M > 128 path (launch_kuu_grad, elbo.hip) puts D M Q + ceil(M / 64) D Q doubles into the same
    // region; M > N is legal there (prediction evaluates few test points)
    const size_t kuu = (size_t)D * M * Q + (size_t)dpgp_ceil_div(M, 64) * D * Q;
    return sizeof(double) * (rows > kuu ? rows : kuu);
}
// the same region when only the K_uu term goes through launch_psi_grad (do_psi2 = 0: the mixed-precision stage B) or launch_kuu_grad:
// no [D][N][Q] partials of dmu / ds (328 MB of the 545 at N = 2000, D = 512, Q = 10), the two-stage reduction of dz behind the partials
size_t psi_grad_ws_bytes_kuu(int D, int M, int Q) {
    const size_t a = (size_t)D * M * Q + (size_t)D * Q + reduce_rows_stage_elems((size_t)M * Q);
    const size_t kuu = (size_t)D * M * Q + (size_t)dpgp_ceil_div(M, 64) * D * Q;
    return sizeof(double) * (a > kuu ? a : kuu);
}

template <typename TC>
int launch_psi_grad(int D, int N, int M, int Q, const double *y, int ldy, const double *z, const double *mu, const double *s,
                    const double *gamma, const double *alpha, const double *GP, const double *WK, const double *Gv,
                    double *ws, double *dmu, double *ds, double *dz, double *dgamma, int do_psi2, hipStream_t st) {
    const int Mp = dpgp_round_up(M, 16);
    if (Mp > 128) return -30;                                  // first version: one thread per row of the M x M statistics
    int ns = 1;
    psi_grad_ws_bytes(D, N, M, Q, &ns);
    if (!do_psi2) ns = 1;
    const int nper = 2 * dpgp_ceil_div(dpgp_ceil_div(N, ns), 2);
    // (K_uu term only: no dmu / ds partials, see psi_grad_ws_bytes_kuu)
    double *dmu_part = ws, *ds_part = dmu_part + (do_psi2 ? (size_t)D * N * Q : 0), *dz_part = ds_part + (do_psi2 ? (size_t)D * N * Q : 0),
           *dg_part = dz_part + (size_t)D * ns * M * Q, *kstage = dg_part + (size_t)ns * D * Q;
    const int QPr = 4 * dpgp_ceil_div(Q, 4);
    const size_t lds = sizeof(TC) * ((size_t)(Mp * Mp > 2 * Q * 128 ? Mp * Mp : 2 * Q * 128) + (size_t)Mp * QPr + PG_RED_ELEMS(Q) + (size_t)12 * Q + 256 + 32 + (size_t)2 * (7 * Q + 2));
    void (*kern)(int, int, int, int, int, const double *, int, const double *, const double *, const double *, const double *,
                 const double *, const double *, const double *, const double *, int, int, double *, double *, double *, double *) = nullptr;
    switch (QPr / 4) {
#define CASE(k) case k: kern = psi_grad_kernel<TC, 4 * k>; break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
    if (!kern) return -4;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3(D, ns), dim3(256), lds, st, D, N, M, Mp, Q, y, ldy, z, mu, s, gamma, alpha, GP, WK, Gv,
                       nper, do_psi2, dmu_part, ds_part, dz_part, dg_part);
    DPGP_LAUNCH_CHECK();
    const size_t nq = (size_t)N * Q, mq = (size_t)M * Q, dq = (size_t)D * Q;
    if (do_psi2) {
        DPGP_PRELAUNCH(); hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, nq, D, (const double *)dmu_part, dmu);
        DPGP_LAUNCH_CHECK();
        DPGP_PRELAUNCH(); hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, nq, D, (const double *)ds_part, ds);
        DPGP_LAUNCH_CHECK();
    }
    if (!do_psi2) {
        // K_uu term only: M Q sums over D rows — side by side in 64 chunks instead of five workgroups walking 512 rows each
        // (69 -> ~10 us at D = 512)
        const int rc = launch_reduce_rows<double>(mq, mq, D * ns, dz_part, dz, 0, kstage, st);
        if (rc != DPGP_OK) return rc;
    } else {
        DPGP_PRELAUNCH(); hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)((mq + 255) / 256)), dim3(256), 0, st, mq, D * ns, (const double *)dz_part, dz);
        DPGP_LAUNCH_CHECK();
    }
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)((dq + 255) / 256)), dim3(256), 0, st, dq, ns, (const double *)dg_part, dgamma);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
template int launch_psi_grad<float>(int, int, int, int, const double *, int, const double *, const double *, const double *,
                                    const double *, const double *, const double *, const double *, const double *, double *,
                                    double *, double *, double *, double *, int, hipStream_t);
template int launch_psi_grad<double>(int, int, int, int, const double *, int, const double *, const double *, const double *,
                                     const double *, const double *, const double *, const double *, const double *,
                                     double *, double *, double *, double *, double *, int, hipStream_t);
