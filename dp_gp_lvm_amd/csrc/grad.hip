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
    {
        int cnt = 0;
        for (int I = 0; I < nb; ++I)
            for (int J = 0; J < nb; ++J, ++cnt) {
                if ((cnt & 3) != wv) continue;
                acc_t c = {0, 0, 0, 0};
                for (int k = 0; k < nb; ++k) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        c = Mfma<double>::mma(sym_at(KI, Mp, 16 * I + li, 16 * k + 4 * ks + kk),
                                              sym_at(Pd, Mp, 16 * J + li, 16 * k + 4 * ks + kk), c);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) T1[(size_t)(16 * I + Mfma<double>::row(lane, v)) * Mp + 16 * J + li] = c[v];
            }
    }
    __threadfence_block();
    __syncthreads();
    // ---- X = T1 K^-1 (lower tiles) and the outputs, element by element in the MFMA result layout ----
    double sK = 0.0, sP = 0.0, sGBP = 0.0, tr = 0.0;
    {
        int cnt = 0;
        for (int I = 0; I < nb; ++I)
            for (int J = 0; J <= I; ++J, ++cnt) {
                if ((cnt & 3) != wv) continue;
                acc_t c = {0, 0, 0, 0};
                for (int k = 0; k < nb; ++k) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        c = Mfma<double>::mma(T1[(size_t)(16 * I + li) * Mp + 16 * k + 4 * ks + kk],
                                              sym_at(KI, Mp, 16 * J + li, 16 * k + 4 * ks + kk), c);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 16 * I + Mfma<double>::row(lane, v), j = 16 * J + li;
                    double gp = 0.0, wk = 0.0;
                    if (j <= i && i < M) {
                        const double ki = KI[(size_t)i * Mp + j], bi = binv_at(tiles, nb, i, j), p = Pd[(size_t)i * Mp + j];
                        const double k0 = K0[(size_t)i * Mp + j] - (i == j ? jitter : 0.0);
                        const double gb = -0.5 * bi - 0.5 * be * be * ww[i] * ww[j];
                        const double gk = 0.5 * ki - 0.5 * be * c[v] + gb;
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
