// Stage A of the backward pass for M > 128 (M a multiple of 128) as ONE library call: the adjoints of the per-output dense algebra that
// chain_grad_kernel (grad.hip) forms in LDS for M <= 128, here on [D][M][M] fp64 matrices in memory.  Reference: tf.gradients through
// dp_gp_lvm.py:108-145 (test/synthetic_data_hard_test.py:143-155 trains at any M).  With B = K + beta P (P = Psi2, v = Psi1^T y,
// w = B^-1 v):
//     G_B = -1/2 B^-1 - 1/2 beta^2 w w^T            G_v = beta^2 w
//     G_K = 1/2 K^-1 - 1/2 beta K^-1 P K^-1 + G_B   G_P = 1/2 beta K^-1 + beta G_B
//     d/dalpha, d/dbeta as in grad.hip
// Every M^3 step is one of the library's own kernels — the persistent-workgroup Cholesky (potrf_persist.hip), the persistent solve
// X = L^-1 I with every block row stored (dpgp_trtri_lower_batched_f64), the strided fp64 MFMA product (gemm.hip) — and the rest is three
// streaming kernels: gather (Psi2 slabs -> symmetric P, B = K + beta P; v, y'y), w = B^-1 v, and the element-wise adjoints with their
// four sums.  Round 4 composed the same on the host from ~50 launches, 35 of them torch element-wise kernels over 537 MB arrays
// (config 4: 28 ms, 15 with the persistent solve; this: see DESIGN.md 7.1).
#include "internal.h"

// ---- gather: P[d] = symmetric sum of the lower-patch slabs, Bm[d] = K[d] + beta_d P[d] --------------------------------------
template <typename TP>
__global__ __launch_bounds__(256) void sab_gather_kernel(int D, int M, const TP *__restrict__ slabs, int ns2, const double *__restrict__ K,
                                                         const double *__restrict__ beta, double *__restrict__ P, double *__restrict__ Bm) {
    const int d = blockIdx.y;
    const size_t mm = (size_t)M * M, e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= mm) return;
    const int i = (int)(e / M), j = (int)(e - (size_t)i * M);
    const size_t lo = (i >= j) ? e : (size_t)j * M + i;            // (the psi2 kernels write entries m' <= m)
    double a = 0.0;
    for (int k = 0; k < ns2; ++k) a += (double)slabs[((size_t)k * D + d) * mm + lo];
    P[(size_t)d * mm + e] = a;
    Bm[(size_t)d * mm + e] = K[(size_t)d * mm + e] + beta[d] * a;
}
// v[d][m] = sum of the Psi1^T y slabs, yy[d] = sum of the y'y partials
__global__ __launch_bounds__(256) void sab_vec_kernel(int D, int M, const double *__restrict__ vpart, int ns1, const double *__restrict__ yy_part,
                                                      double *__restrict__ v, double *__restrict__ yy) {
    const int d = blockIdx.x;
    for (int m = threadIdx.x; m < M; m += 256) {
        double a = 0.0;
        for (int k = 0; k < ns1; ++k) a += vpart[((size_t)k * D + d) * M + m];
        v[(size_t)d * M + m] = a;
    }
    if (threadIdx.x == 0) {
        double a = 0.0;
        for (int k = 0; k < DPGP_YY_NCH; ++k) a += yy_part[(size_t)k * D + d];
        yy[d] = a;
    }
}
// w[d] = B^-1[d] v[d] (wave per row), vw[d] = v . w
__global__ __launch_bounds__(256) void sab_w_kernel(int M, const double *__restrict__ Binv, const double *__restrict__ v,
                                                    double *__restrict__ w, double *__restrict__ vw) {
    __shared__ double scratch[8];
    const int d = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const double *B = Binv + (size_t)d * M * M, *vd = v + (size_t)d * M;
    double part = 0.0;
    for (int i = wv; i < M; i += 4) {
        double a = 0.0;
        for (int j = lane; j < M; j += 64) a += B[(size_t)i * M + j] * vd[j];
        a = wave_sum(a);
        if (lane == 0) {
            w[(size_t)d * M + i] = a;
            part += a * vd[i];
        }
    }
    part = block_sum(part, scratch);
    if (t == 0) vw[d] = part;
}
// the element-wise adjoints and their four sums: blocks of 2048 elements, partial sums [D][nblk][4] (fixed order: no atomics)
__global__ __launch_bounds__(256) void sab_adjoint_kernel(int M, int nblk, const double *__restrict__ Kinv, const double *__restrict__ Binv,
                                                          const double *__restrict__ X, const double *__restrict__ P,
                                                          const double *__restrict__ K, const double *__restrict__ w,
                                                          const double *__restrict__ beta, double jitter, double *__restrict__ GP,
                                                          double *__restrict__ WK, double *__restrict__ part) {
    __shared__ double scratch[8];
    const int d = blockIdx.y, t = threadIdx.x;
    const size_t mm = (size_t)M * M, base = (size_t)d * mm;
    const double be = beta[d];
    const double *wd = w + (size_t)d * M;
    double sK = 0.0, sP = 0.0, sGBP = 0.0, tr = 0.0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const size_t e = (size_t)blockIdx.x * 2048 + (size_t)r * 256 + t;
        if (e < mm) {
            const int i = (int)(e / M), j = (int)(e - (size_t)i * M);
            const double ki = Kinv[base + e], bi = Binv[base + e], p = P[base + e];
            const double gb = -0.5 * bi - 0.5 * be * be * wd[i] * wd[j];
            const double gk = 0.5 * ki - 0.5 * be * X[base + e] + gb;
            const double gp = 0.5 * be * ki + be * gb;
            const double wk = gk * (K[base + e] - (i == j ? jitter : 0.0));
            GP[base + e] = gp;
            WK[base + e] = wk;
            sK += wk; sP += gp * p; sGBP += gb * p; tr += ki * p;
        }
    }
    sK = block_sum(sK, scratch);
    sP = block_sum(sP, scratch);
    sGBP = block_sum(sGBP, scratch);
    tr = block_sum(tr, scratch);
    if (t == 0) {
        double *o = part + ((size_t)d * nblk + blockIdx.x) * 4;
        o[0] = sK; o[1] = sP; o[2] = sGBP; o[3] = tr;
    }
}
__global__ __launch_bounds__(256) void sab_finish_kernel(int D, int N, int M, int nblk, const double *__restrict__ part,
                                                         const double *__restrict__ w, const double *__restrict__ vw,
                                                         const double *__restrict__ yy, const double *__restrict__ alpha,
                                                         const double *__restrict__ beta, const int *__restrict__ info_k,
                                                         const int *__restrict__ info_b, double *__restrict__ Gv, double *__restrict__ dab,
                                                         int *__restrict__ info) {
    __shared__ double scratch[8];
    const int d = blockIdx.x, t = threadIdx.x;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = t; b < nblk; b += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += part[((size_t)d * nblk + b) * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = block_sum(s[k], scratch);
    const double be = beta[d], al = alpha[d];
    for (int m = t; m < M; m += 256) Gv[(size_t)d * M + m] = be * be * w[(size_t)d * M + m];
    if (t == 0) {
        dab[2 * d] = -0.5 * be * N + (s[0] + 2.0 * s[1] + be * be * vw[d]) / al;
        dab[2 * d + 1] = 0.5 * N / be + 0.5 * (s[3] - al * N) - 0.5 * yy[d] + be * vw[d] + s[2];
        info[d] = info_k[d] > info_b[d] ? info_k[d] : info_b[d];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct SabLayout {
    size_t off_p, off_k, off_kinv, off_binv, off_ta, off_tb, off_potrf, off_v, off_w, off_yy, off_vw, off_part, off_ik, off_ib, off_tr, total;
    size_t potrf_bytes;
    int nblk;
};
static SabLayout sab_layout(int D, int M) {
    SabLayout L;
    const size_t mat = dpgp_align256(sizeof(double) * (size_t)D * M * M);
    size_t o = 0;
    L.off_p = o; o += mat;
    L.off_k = o; o += mat;
    L.off_kinv = o; o += mat;
    L.off_binv = o; o += mat;
    L.off_ta = o; o += mat;
    L.off_tb = o; o += mat;
    L.potrf_bytes = dpgp_potrf_workspace_bytes(D, M, 8);
    L.off_potrf = o; o += dpgp_align256(L.potrf_bytes);
    L.off_v = o; o += dpgp_align256(sizeof(double) * (size_t)D * M);
    L.off_w = o; o += dpgp_align256(sizeof(double) * (size_t)D * M);
    L.off_yy = o; o += dpgp_align256(sizeof(double) * D);
    L.off_vw = o; o += dpgp_align256(sizeof(double) * D);
    L.nblk = (int)(((size_t)M * M + 2047) / 2048);
    L.off_part = o; o += dpgp_align256(sizeof(double) * (size_t)D * L.nblk * 4);
    L.off_ik = o; o += dpgp_align256(sizeof(int) * D);
    L.off_ib = o; o += dpgp_align256(sizeof(int) * D);
    L.off_tr = o; o += dpgp_align256(sizeof(double) * D);
    L.total = o;
    return L;
}
bool chain_grad_big_supported(int M) { return M > 128 && M % 128 == 0; }
size_t chain_grad_big_ws_bytes(int D, int M) { return chain_grad_big_supported(M) ? sab_layout(D, M).total : 0; }

template <typename TP>
int launch_chain_grad_big(int D, int N, int M, int Q, const double *z, const double *gamma, const double *alpha, const double *beta,
                          double jitter, const TP *psi2_part, int ns2, const double *v_part, int ns1, const double *yy_part,
                          unsigned char *ws, double *GP, double *WK, double *Gv, double *dab, int *info, hipStream_t st) {
    if (!chain_grad_big_supported(M)) return -3;
    const SabLayout L = sab_layout(D, M);
    double *P = reinterpret_cast<double *>(ws + L.off_p), *K = reinterpret_cast<double *>(ws + L.off_k),
           *Kinv = reinterpret_cast<double *>(ws + L.off_kinv), *Binv = reinterpret_cast<double *>(ws + L.off_binv),
           *Ta = reinterpret_cast<double *>(ws + L.off_ta), *Tb = reinterpret_cast<double *>(ws + L.off_tb);
    double *v = reinterpret_cast<double *>(ws + L.off_v), *w = reinterpret_cast<double *>(ws + L.off_w),
           *yy = reinterpret_cast<double *>(ws + L.off_yy), *vw = reinterpret_cast<double *>(ws + L.off_vw),
           *part = reinterpret_cast<double *>(ws + L.off_part), *trws = reinterpret_cast<double *>(ws + L.off_tr);
    int *ik = reinterpret_cast<int *>(ws + L.off_ik), *ib = reinterpret_cast<int *>(ws + L.off_ib);
    void *pws = ws + L.off_potrf;
    const long long mm = (long long)M * M;
    const size_t bytes = sizeof(double) * (size_t)D * M * M;
    int rc;
    // K_uu + jitter I (rbf_kernel.py:58-93), its factor and inverse
    if ((rc = dpgp_ard_rbf_gram_f64(D, M, M, Q, z, nullptr, gamma, alpha, beta, DPGP_FLAG_JITTER, jitter, K, st))) return rc;
    if (hipMemcpyAsync(Ta, K, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return DPGP_ERR_LAUNCH;
    if ((rc = dpgp_potrf_batched_f64(D, M, Ta, ik, pws, L.potrf_bytes, DPGP_ALGO_AUTO, st))) return rc;
    if ((rc = dpgp_trtri_lower_batched_f64(D, M, Ta, Tb, trws, sizeof(double) * (size_t)D, st))) return rc;
    if ((rc = dpgp_gemm_strided_f64(D, M, M, M, 1.0, Tb, mm, 1, M, Tb, mm, M, 1, 0.0, Kinv, mm, M, 1, st))) return rc;      // K^-1 = W^T W
    // P, B = K + beta P, its factor and inverse; v, y'y
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL((sab_gather_kernel<TP>), dim3((unsigned)((mm + 255) / 256), D), dim3(256), 0, st, D, M, psi2_part, ns2, (const double *)K, beta, P, Ta);
    DPGP_LAUNCH_CHECK();
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(sab_vec_kernel, dim3(D), dim3(256), 0, st, D, M, v_part, ns1, yy_part, v, yy);
    DPGP_LAUNCH_CHECK();
    if ((rc = dpgp_potrf_batched_f64(D, M, Ta, ib, pws, L.potrf_bytes, DPGP_ALGO_AUTO, st))) return rc;
    if ((rc = dpgp_trtri_lower_batched_f64(D, M, Ta, Tb, trws, sizeof(double) * (size_t)D, st))) return rc;
    if ((rc = dpgp_gemm_strided_f64(D, M, M, M, 1.0, Tb, mm, 1, M, Tb, mm, M, 1, 0.0, Binv, mm, M, 1, st))) return rc;
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(sab_w_kernel, dim3(D), dim3(256), 0, st, M, (const double *)Binv, (const double *)v, w, vw);
    DPGP_LAUNCH_CHECK();
    // X = K^-1 P K^-1
    if ((rc = dpgp_gemm_strided_f64(D, M, M, M, 1.0, Kinv, mm, M, 1, P, mm, M, 1, 0.0, Ta, mm, M, 1, st))) return rc;
    if ((rc = dpgp_gemm_strided_f64(D, M, M, M, 1.0, Ta, mm, M, 1, Kinv, mm, M, 1, 0.0, Tb, mm, M, 1, st))) return rc;
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(sab_adjoint_kernel, dim3(L.nblk, D), dim3(256), 0, st, M, L.nblk, (const double *)Kinv, (const double *)Binv,
                       (const double *)Tb, (const double *)P, (const double *)K, (const double *)w, beta, jitter, GP, WK, part);
    DPGP_LAUNCH_CHECK();
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(sab_finish_kernel, dim3(D), dim3(256), 0, st, D, N, M, L.nblk, (const double *)part, (const double *)w,
                       (const double *)vw, (const double *)yy, alpha, beta, (const int *)ik, (const int *)ib, Gv, dab, info);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
template int launch_chain_grad_big<float>(int, int, int, int, const double *, const double *, const double *, const double *, double,
                                          const float *, int, const double *, int, const double *, unsigned char *, double *, double *,
                                          double *, double *, int *, hipStream_t);
template int launch_chain_grad_big<double>(int, int, int, int, const double *, const double *, const double *, const double *, double,
                                           const double *, int, const double *, int, const double *, unsigned char *, double *, double *,
                                           double *, double *, int *, hipStream_t);
