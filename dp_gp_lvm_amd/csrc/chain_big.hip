// The dense chain of the fused ELBO (dp_gp_lvm.py:115-145) for M > 128 in fp64, when there are enough output dims to give every
// compute unit a matrix of its own (config 4: D = 256, M = 512).  Round 2 ran this case as ONE workgroup per output dim over
// global memory (chain_k_kernel 7.9 ms + chain_b_kernel 2.05 ms per evaluation); here every M^3 step is a persistent-workgroup
// kernel of potrf_persist.hip and the rest is streaming:
//      K side (before Psi2):   Kw = pad(K_uu + jitter I);  L_K = chol(Kw);  W = L_K^-1 (only |W|_F^2 = tr K^-1 is kept)
//      B side (after Psi2):    Bw = pad(K + beta Psi2);     L_B = chol(Bw);  c = L_B^-1 v;  X = L_K^-1 L_B (only |X|_F^2)
// X is the reference's L_A (A = L^-1 B L^-T = X X^T, X lower with positive diagonal), so with the identity padding to Mw rows
//      -sum log diag L_A = -(sum log diag L_B - sum log diag L_K)
//      tr(L^-1 Psi2 L^-T) = (tr A - M) / beta = (|X|_F^2 - Mw) / beta        (fp64: the cancellation costs ~M eps)
//      |L_A^-1 L^-1 v|^2 = |L_B^-1 v|^2 = |c|^2
// and no inverse of K_uu is formed.  The conditioning guard (see chain_b_kernel) needs |K^-1|_F: it uses the bound
// |K^-1|_F <= tr K^-1 = |W|_F^2 (at most sqrt(M) looser, typically ~1-3x: a few small eigenvalues dominate both).
// Per-output workspace (doubles, stride la_chain_ws_elems(M)):
//      K0 [Mp x Mp] (gram output of the front launch) | Kw [Mw x Mw] | Ww [Mw x Mw] | Bw [Mw x Mw] | tail [128]
//      tail: 0 |W|^2, 1 |X|^2, 2 sum log diag L_K, 3 sum log diag L_B, 4 |c|^2, 16.. partial |Psi2|_F^2 per 64 rows
#include "internal.h"
#include "linalg_dev.h"

#define CB_TAIL 128
#define CB_ROWS 64                 // rows per workgroup of the streaming kernels

typedef double cb_f2 __attribute__((ext_vector_type(2)));
typedef double cb_f4 __attribute__((ext_vector_type(4)));

bool chain_big_applicable(int D, int M, int elem) {
    if (const char *e = getenv("DPGP_CHAIN_BIG")) {                // (experiments / cross-checks only)
        if (atoi(e) == 0) return false;
        if (atoi(e) == 1 && elem == 8 && M > 128 && dpgp_round_up(M, 128) <= 4096) return true;
    }
    return potrf_persist_applicable(D, M, elem) && dpgp_round_up(M, 128) <= 4096;
}
size_t chain_big_ws_elems(int M) { return chain_big_elems_inline(M); }

struct CbLayout {
    int Mp, Mw;
    size_t kw, ww, bw, tail, stride;
};
static CbLayout cb_layout(int M) {
    CbLayout c;
    c.Mp = dpgp_round_up(M, 16);
    c.Mw = dpgp_round_up(M, 128);
    c.kw = (size_t)c.Mp * c.Mp;
    c.ww = c.kw + (size_t)c.Mw * c.Mw;
    c.bw = c.ww + (size_t)c.Mw * c.Mw;
    c.tail = c.bw + (size_t)c.Mw * c.Mw;
    c.stride = la_chain_ws_elems_inline(M);
    return c;
}

// Kw = lower(K0) with identity padding; block-lower region only (what the persistent kernels read); grid (Mw / 64, D).
// (Ww, the scratch of the solve W = L_K^-1 I, needs no initialisation: launch_ptrsm_persist(..., identity = 1))
__global__ __launch_bounds__(256) void cbig_pad_kernel(int M, int Mp, int Mw, double *__restrict__ ws, size_t stride, size_t off_kw,
                                                       size_t off_ww) {
    const int d = blockIdx.y, i0 = CB_ROWS * blockIdx.x, t = threadIdx.x;
    const double *K0 = ws + (size_t)d * stride;
    double *Kw = ws + (size_t)d * stride + off_kw;
    const int lim = 128 * (i0 / 128 + 1), npair = lim / 2;         // columns of the block-lower region of these rows
    for (int e = t; e < CB_ROWS * npair; e += 256) {
        const int i = i0 + e / npair, j = 2 * (e % npair);
        cb_f2 kv;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int jj = j + u;
            kv[u] = (jj > i) ? 0.0 : ((i < M) ? K0[(size_t)i * Mp + jj] : (i == jj ? 1.0 : 0.0));
        }
        *reinterpret_cast<cb_f2 *>(Kw + (size_t)i * Mw + j) = kv;
    }
}

// Bw = lower(K0 + beta sum of the Psi2 slabs) with identity padding; |Psi2|_F^2 of these rows -> tail[16 + blockIdx.x]
template <typename TP>
__global__ __launch_bounds__(256) void cbig_assemble_kernel(int D, int M, int Mp, int Mw, const TP *__restrict__ psi2_part, int ns2,
                                                            const double *__restrict__ beta, double *__restrict__ ws,
                                                            size_t stride, size_t off_bw, size_t off_tail) {
    __shared__ double scratch[8];
    typedef TP tp4 __attribute__((ext_vector_type(4)));
    const int d = blockIdx.y, i0 = CB_ROWS * blockIdx.x, t = threadIdx.x;
    const double *K0 = ws + (size_t)d * stride;
    double *Bw = ws + (size_t)d * stride + off_bw;
    const double be = beta[d];
    const int lim = 128 * (i0 / 128 + 1), nq = lim / 4;
    double p2n2 = 0.0;
    for (int e = t; e < CB_ROWS * nq; e += 256) {
        const int i = i0 + e / nq, j = 4 * (e % nq);
        cb_f4 bv;
        if (i < M && j <= i) {                                      // (j < M follows; j + 3 < Mp: Mp is a multiple of 16)
            const size_t off = (size_t)i * Mp + j;
            const cb_f4 k0 = *reinterpret_cast<const cb_f4 *>(K0 + off);
            double p2[4] = {0.0, 0.0, 0.0, 0.0};
            for (int kb = 0; kb < ns2; kb += 8) {
                tp4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (kb + k < ns2) v[k] = *reinterpret_cast<const tp4 *>(psi2_part + ((size_t)(kb + k) * D + d) * (size_t)Mp * Mp + off);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (kb + k < ns2) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) p2[u] += (double)v[k][u];
                    }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int jj = j + u;
                if (jj <= i) {
                    bv[u] = k0[u] + be * p2[u];
                    p2n2 += p2[u] * p2[u] * (i == jj ? 1.0 : 2.0);
                } else {
                    bv[u] = 0.0;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) bv[u] = (i >= M && i == j + u) ? 1.0 : 0.0;
        }
        *reinterpret_cast<cb_f4 *>(Bw + (size_t)i * Mw + j) = bv;
    }
    p2n2 = block_sum(p2n2, scratch);
    if (t == 0) ws[(size_t)d * stride + off_tail + 16 + blockIdx.x] = p2n2;
}

// c = L_B^-1 v (v = sum of the Psi1^T y slabs), |c|^2 and the two log-determinants; one workgroup per output dim.
// Forward substitution in blocks of 64 rows: the part of the rows left of the block as a matrix-vector product over all 256
// threads (thread = (row, quarter of the columns), 32-byte loads), the 64 x 64 diagonal block staged in LDS and solved by one
// wave with one lane per row (64 dependent steps of a shuffle and an LDS read).  (Round 3's first version went 16 rows at a
// time, each step a global-load round trip and two barriers: 285 us at M = 512.)
#define CB_TB 64
__global__ __launch_bounds__(256) void cbig_trsv_kernel(int D, int M, int Mw, double *__restrict__ ws, size_t stride, size_t off_kw,
                                                        size_t off_bw, size_t off_tail, const double *__restrict__ v_part,
                                                        int ns1) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double *scratch = reinterpret_cast<double *>(smem_raw);       // 8 doubles
    double *rs = scratch + 8;                                      // CB_TB doubles
    double *ld = rs + CB_TB;                                       // [CB_TB][CB_TB + 1]: the diagonal block
    double *c = ld + CB_TB * (CB_TB + 1);                          // Mw doubles
    const int d = blockIdx.x, t = threadIdx.x;
    const double *LB = ws + (size_t)d * stride + off_bw, *LK = ws + (size_t)d * stride + off_kw;
    for (int j = t; j < Mw; j += 256) {
        double a = 0.0;
        if (j < M)
            for (int k = 0; k < ns1; ++k) a += v_part[((size_t)k * D + d) * M + j];
        c[j] = a;
    }
    __syncthreads();
    const int nblk = (M + CB_TB - 1) / CB_TB, i = t >> 2, g = t & 3;
    for (int Ib = 0; Ib < nblk; ++Ib) {
        const double *row = LB + (size_t)(CB_TB * Ib + i) * Mw;
        double a = 0.0;
        for (int j = 4 * g; j < CB_TB * Ib; j += 16) {
            const cb_f4 l4 = *reinterpret_cast<const cb_f4 *>(row + j);
            a += (l4[0] * c[j] + l4[1] * c[j + 1]) + (l4[2] * c[j + 2] + l4[3] * c[j + 3]);
        }
        a += __shfl_xor(a, 1, 4);
        a += __shfl_xor(a, 2, 4);
        if (g == 0) rs[i] = c[CB_TB * Ib + i] - a;
        for (int e = t; e < CB_TB * CB_TB / 4; e += 256) {        // the diagonal block -> LDS (16 threads per row)
            const int r = e >> 4, c4 = (e & 15) * 4;
            const cb_f4 l4 = *reinterpret_cast<const cb_f4 *>(LB + (size_t)(CB_TB * Ib + r) * Mw + CB_TB * Ib + c4);
#pragma unroll
            for (int u = 0; u < 4; ++u) ld[r * (CB_TB + 1) + c4 + u] = l4[u];
        }
        __syncthreads();
        if (t < CB_TB) {                                           // one wave, lane = row (rows >= M: identity padding, v = 0)
            const double rd = 1.0 / ld[t * (CB_TB + 1) + t];
            double x = rs[t];
            for (int j = 0; j < CB_TB; ++j) {
                const double xj = __shfl(x * rd, j, 64);
                if (t > j) x -= ld[t * (CB_TB + 1) + j] * xj;
                if (t == j) x = xj;
            }
            c[CB_TB * Ib + t] = x;
        }
        __syncthreads();
    }
    double ldk = 0.0, ldb = 0.0, cc = 0.0;
    for (int j = t; j < M; j += 256) {
        ldk += log(LK[(size_t)j * Mw + j]);
        ldb += log(LB[(size_t)j * Mw + j]);
        cc += c[j] * c[j];
    }
    ldk = block_sum(ldk, scratch);
    ldb = block_sum(ldb, scratch);
    cc = block_sum(cc, scratch);
    if (t == 0) {
        double *tail = ws + (size_t)d * stride + off_tail;
        tail[2] = ldk;
        tail[3] = ldb;
        tail[4] = cc;
    }
}

// the five f_hat terms per output dim from the tail values (thread = output dim); semantics of info / guard: chain_b_kernel
__global__ __launch_bounds__(256) void cbig_terms_kernel(int D, int N, int M, int Mw, const double *__restrict__ ws, size_t stride,
                                                         size_t off_tail, const double *__restrict__ alpha,
                                                         const double *__restrict__ beta, const double *__restrict__ yy_part,
                                                         const int *__restrict__ info_k, double *__restrict__ terms,
                                                         int *__restrict__ info, double *__restrict__ guard, int psi2_f32) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const double *tail = ws + (size_t)d * stride + off_tail;
    const double b_ = beta[d], a_ = alpha[d];
    const double trk = tail[0] - (double)(Mw - M);                 // tr K^-1 >= |K^-1|_F
    const double ip = (tail[1] - (double)Mw) / b_;                 // tr(L^-1 Psi2 L^-T)
    const double ldk = tail[2], ldb = tail[3], cc = tail[4];
    double p2n2 = 0.0, yy = 0.0;
    for (int k = 0; k < Mw / CB_ROWS; ++k) p2n2 += tail[16 + k];
    for (int k = 0; k < DPGP_YY_NCH; ++k) yy += yy_part[(size_t)k * D + d];
    const int fk = info_k[d], fb = info[d];
    int f = fk ? fk : (fb ? M + fb : 0);
    const double errb = 1.1920928955078125e-07 * b_ * trk * sqrt(p2n2) * (1.0 + 0.5 * b_ * b_ * cc);
    if (guard) guard[d] = errb;
    if (psi2_f32 && !f && !(errb <= DPGP_GUARD_REL * (double)N)) f = DPGP_INFO_ILL_CONDITIONED;
    const bool flagged_only = (f == DPGP_INFO_ILL_CONDITIONED);
    info[d] = f;
    if (flagged_only) f = 0;
    const double nan_ = __longlong_as_double(0x7ff8000000000000LL);
    double *o = terms + (size_t)d * 5;
    o[0] = 0.5 * N * (log(b_) - DPGP_LOG_2PI);
    o[1] = f ? nan_ : -(ldb - ldk);
    o[2] = f ? nan_ : 0.5 * b_ * (ip - a_ * N);
    o[3] = -0.5 * b_ * yy;
    o[4] = f ? nan_ : 0.5 * b_ * b_ * cc;
}

int launch_chain_big_k(int D, int M, double *ws, int *info_k, hipStream_t st) {
    const CbLayout c = cb_layout(M);
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(cbig_pad_kernel, dim3(c.Mw / CB_ROWS, D), dim3(256), 0, st, M, c.Mp, c.Mw, ws, c.stride, c.kw, c.ww);
    DPGP_LAUNCH_CHECK();
    int rc;
    if ((rc = launch_potrf_persist(D, c.Mw, ws + c.kw, info_k, st, c.stride))) return rc;
    return launch_ptrsm_persist(D, c.Mw, ws + c.kw, c.stride, ws + c.ww, c.stride, ws + c.tail + 0, c.stride, st, 1);
}

template <typename TP>
int launch_chain_big_b(int D, int N, int M, const TP *psi2_part, int ns2, const double *v_part, int ns1, const double *alpha,
                       const double *beta, const double *yy_part, const int *info_k, double *terms, int *info, double *guard,
                       double *ws, hipStream_t st, const double *kl_part, double *sums, const double *model_scal,
                       double *model_pack, double *model_out) {
    const CbLayout c = cb_layout(M);
    int rc;
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL((cbig_assemble_kernel<TP>), dim3(c.Mw / CB_ROWS, D), dim3(256), 0, st, D, M, c.Mp, c.Mw, psi2_part, ns2, beta,
                       ws, c.stride, c.bw, c.tail);
    DPGP_LAUNCH_CHECK();
    if ((rc = launch_potrf_persist(D, c.Mw, ws + c.bw, info, st, c.stride))) return rc;
    const size_t lds = sizeof(double) * (size_t)(8 + CB_TB + CB_TB * (CB_TB + 1) + c.Mw);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(cbig_trsv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(cbig_trsv_kernel, dim3(D), dim3(256), lds, st, D, M, c.Mw, ws, c.stride, c.kw, c.bw, c.tail, v_part, ns1);
    DPGP_LAUNCH_CHECK();
    if ((rc = launch_ptrsm_persist(D, c.Mw, ws + c.kw, c.stride, ws + c.bw, c.stride, ws + c.tail + 1, c.stride, st))) return rc;
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(cbig_terms_kernel, dim3(dpgp_ceil_div(D, 256)), dim3(256), 0, st, D, N, M, c.Mw, ws, c.stride, c.tail, alpha,
                       beta, yy_part, info_k, terms, info, guard, (int)(sizeof(TP) == 4));
    DPGP_LAUNCH_CHECK();
    if (!sums) return DPGP_OK;
    return launch_sum_terms(D, terms, kl_part, sums, model_scal, model_pack, model_out, st);
}
template int launch_chain_big_b<float>(int, int, int, const float *, int, const double *, int, const double *, const double *,
                                       const double *, const int *, double *, int *, double *, double *, hipStream_t,
                                       const double *, double *, const double *, double *, double *);
template int launch_chain_big_b<double>(int, int, int, const double *, int, const double *, int, const double *, const double *,
                                        const double *, const int *, double *, int *, double *, double *, hipStream_t,
                                        const double *, double *, const double *, double *, double *);
