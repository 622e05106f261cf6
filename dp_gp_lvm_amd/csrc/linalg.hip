// K4/K5/K6  batched blocked Cholesky, triangular solves and the per-output ELBO reduction on the matrix cores.
// Reference ops replaced: tf.cholesky / tf.matrix_triangular_solve / trace / log-det at
// /root/reference/src/models/dp_gp_lvm.py:115-145.
//
// One 256-thread workgroup (4 waves) owns one matrix.  Everything is tiled in 16x16 blocks (v_mfma_*_16x16x4):
//   potrf  : right-looking.  Step k: wave 0 factors the diagonal tile in registers (one row per lane, columns via
//            cross-lane broadcasts) and also inverts it; the panel below becomes a GEMM with that inverse
//            (P_I = A_Ik Linv_kk^T, MFMA); the trailing lower triangle gets the SYRK/GEMM update A_IJ -= P_I P_J^T
//            (MFMA, operands from an LDS copy of the panel).
//            "Border" tile-rows below the SPD part are carried along (panel + trailing steps only): a border row
//            holding v^T comes out as (L^-1 v)^T, which is how the two M-vector solves of the data-fit term are done.
//   trsm   : with the inverted diagonal tiles both triangular solves are pure MFMA GEMM sweeps.
// Matrices live in global memory (L2/MALL resident: <= 2 x 147 KB per output dim at M=128 fp64), padded to a multiple
// of 16 with an identity block so no tile needs bounds checks.
#include "internal.h"

#define LDT 17          // LDS tile row stride (16 + 1 pad)
#define LA_LDS_HDR 128  // bytes at the start of the dynamic LDS region: 8 doubles of reduction scratch + fail flag

template <typename T> __device__ __forceinline__ T lane_bcast(T v, int src) { return __shfl(v, src, 64); }

// ---- diagonal tile: Cholesky (optional) + inverse, by the calling wave; lanes 0..15 hold one row each ------------
// A: tile origin in global memory (row stride ld).  On exit (FACTOR): tile holds L (upper zeroed).  dinv_lds[16][LDT]
// and, if non-null, dinv_glob[16][16] receive L^-1.  *fail (LDS) gets base+j+1 for the first non-positive pivot.
template <typename T, bool FACTOR>
__device__ void diag_tile(T *A, int ld, T *dinv_lds, T *dinv_glob, int *fail, int base) {
    const int lane = threadIdx.x & 63, li = lane & 15;
    T a[16], rinv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = (c <= li) ? A[(size_t)li * ld + c] : (T)0;
    if (FACTOR) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            T d = lane_bcast(a[j], j);
            if (!(d > (T)0)) {
                if (lane == 0 && *fail == 0) *fail = base + j + 1;
                d = (T)1;
            }
            const T piv = sqrt(d);
            rinv[j] = (T)1 / piv;
            a[j] = (li == j) ? piv : a[j] * rinv[j];
#pragma unroll
            for (int c = j + 1; c < 16; ++c) {
                const T lcj = lane_bcast(a[j], c);
                a[c] = (li >= c) ? a[c] - a[j] * lcj : (T)0;
            }
        }
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 16; ++c) A[(size_t)li * ld + c] = a[c];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) rinv[j] = (T)1 / lane_bcast(a[j], j);
    }
    // inverse: lane c owns column c of X = L^-1;  x_i = (delta_ic - sum_{k<i} L_ik x_k) / L_ii
    T x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        T acc = (li == i) ? (T)1 : (T)0;
#pragma unroll
        for (int k = 0; k < i; ++k) acc -= lane_bcast(a[k], i) * x[k];
        x[i] = acc * rinv[i];
    }
    if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            dinv_lds[i * LDT + li] = x[i];
            if (dinv_glob) dinv_glob[i * 16 + li] = x[i];
        }
    }
}

// ---- blocked Cholesky of the leading nbf x nbf tiles of A, carrying nbr - nbf border tile-rows --------------------
// lds: dinv[16*LDT] + panel[(nbr)*16*LDT];  dinv_glob: [nbf][256] or null.
template <typename T>
__device__ void potrf_blocked(T *A, int ld, int nbf, int nbr, T *lds, T *dinv_glob, int *fail, int fail_base) {
    typedef typename Mfma<T>::acc_t acc_t;
    T *dinv = lds, *panel = lds + 16 * LDT;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, kk = lane >> 4;
    for (int k = 0; k < nbf; ++k) {
        if (wv == 0)
            diag_tile<T, true>(A + (size_t)(16 * k) * ld + 16 * k, ld, dinv, dinv_glob ? dinv_glob + k * 256 : nullptr,
                               fail, fail_base + 16 * k);
        __syncthreads();
        // panel: P_I = A_Ik * Linv^T
        for (int I = k + 1 + wv; I < nbr; I += 4) {
            T *tile = A + (size_t)(16 * I) * ld + 16 * k;
            acc_t c = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const T av = tile[(size_t)li * ld + 4 * ks + kk];
                const T bv = dinv[li * LDT + 4 * ks + kk];
                c = Mfma<T>::mma(av, bv, c);
            }
            T *pl = panel + (I - k - 1) * 16 * LDT;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = Mfma<T>::row(lane, v);
                tile[(size_t)r * ld + li] = c[v];
                pl[r * LDT + li] = c[v];
            }
        }
        __syncthreads();
        // trailing update: A_IJ -= P_I P_J^T for k < J < nbf, J <= I < nbr
        int cnt = 0;
        for (int I = k + 1; I < nbr; ++I) {
            const int jmax = min(I, nbf - 1);
            for (int J = k + 1; J <= jmax; ++J, ++cnt) {
                if ((cnt & 3) != wv) continue;
                T *tile = A + (size_t)(16 * I) * ld + 16 * J;
                const T *pI = panel + (I - k - 1) * 16 * LDT, *pJ = panel + (J - k - 1) * 16 * LDT;
                acc_t c;
#pragma unroll
                for (int v = 0; v < 4; ++v) c[v] = tile[(size_t)Mfma<T>::row(lane, v) * ld + li];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    c = Mfma<T>::mma(-pI[li * LDT + 4 * ks + kk], pJ[li * LDT + 4 * ks + kk], c);
#pragma unroll
                for (int v = 0; v < 4; ++v) tile[(size_t)Mfma<T>::row(lane, v) * ld + li] = c[v];
            }
        }
        __syncthreads();
    }
}

// ---- X = L^-1 B (in place in Bm), L lower nb x nb tiles with inverted diagonal tiles dinv_glob, B nb x nbc tiles -----
// lds: dinv[16*LDT] + xrow[nbc*16*LDT]
template <typename T>
__device__ void trsm_left_blocked(const T *L, int ldl, const T *dinv_glob, T *Bm, int ldb, int nb, int nbc, T *lds) {
    typedef typename Mfma<T>::acc_t acc_t;
    T *dinv = lds, *xrow = lds + 16 * LDT;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kk = lane >> 4;
    for (int k = 0; k < nb; ++k) {
        dinv[(t >> 4) * LDT + (t & 15)] = dinv_glob[k * 256 + t];
        __syncthreads();
        for (int J = wv; J < nbc; J += 4) {
            T *tile = Bm + (size_t)(16 * k) * ldb + 16 * J;
            acc_t c = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                c = Mfma<T>::mma(dinv[li * LDT + 4 * ks + kk], tile[(size_t)(4 * ks + kk) * ldb + li], c);
            T *xl = xrow + J * 16 * LDT;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = Mfma<T>::row(lane, v);
                tile[(size_t)r * ldb + li] = c[v];
                xl[r * LDT + li] = c[v];
            }
        }
        __syncthreads();
        int cnt = 0;
        for (int I = k + 1; I < nb; ++I) {
            const T *lt = L + (size_t)(16 * I) * ldl + 16 * k;
            for (int J = 0; J < nbc; ++J, ++cnt) {
                if ((cnt & 3) != wv) continue;
                T *tile = Bm + (size_t)(16 * I) * ldb + 16 * J;
                const T *xl = xrow + J * 16 * LDT;
                acc_t c;
#pragma unroll
                for (int v = 0; v < 4; ++v) c[v] = tile[(size_t)Mfma<T>::row(lane, v) * ldb + li];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    c = Mfma<T>::mma(-lt[(size_t)li * ldl + 4 * ks + kk], xl[(4 * ks + kk) * LDT + li], c);
#pragma unroll
                for (int v = 0; v < 4; ++v) tile[(size_t)Mfma<T>::row(lane, v) * ldb + li] = c[v];
            }
        }
        __syncthreads();
    }
}

// ---- T = X L^-T on the lower block-triangle (I >= J), in place in X ------------------------------------------------
// lds: dinv[16*LDT] + tpan[nb*16*LDT]
template <typename T>
__device__ void trsm_right_lower_blocked(const T *L, int ldl, const T *dinv_glob, T *X, int ldx, int nb, T *lds) {
    typedef typename Mfma<T>::acc_t acc_t;
    T *dinv = lds, *tpan = lds + 16 * LDT;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kk = lane >> 4;
    for (int J = 0; J < nb; ++J) {
        dinv[(t >> 4) * LDT + (t & 15)] = dinv_glob[J * 256 + t];
        __syncthreads();
        for (int I = J + wv; I < nb; I += 4) {
            T *tile = X + (size_t)(16 * I) * ldx + 16 * J;
            acc_t c = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                c = Mfma<T>::mma(tile[(size_t)li * ldx + 4 * ks + kk], dinv[li * LDT + 4 * ks + kk], c);
            T *pl = tpan + (I - J) * 16 * LDT;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = Mfma<T>::row(lane, v);
                tile[(size_t)r * ldx + li] = c[v];
                pl[r * LDT + li] = c[v];
            }
        }
        __syncthreads();
        int cnt = 0;
        for (int Jp = J + 1; Jp < nb; ++Jp) {
            const T *lt = L + (size_t)(16 * Jp) * ldl + 16 * J;
            for (int I = Jp; I < nb; ++I, ++cnt) {
                if ((cnt & 3) != wv) continue;
                T *tile = X + (size_t)(16 * I) * ldx + 16 * Jp;
                const T *pl = tpan + (I - J) * 16 * LDT;
                acc_t c;
#pragma unroll
                for (int v = 0; v < 4; ++v) c[v] = tile[(size_t)Mfma<T>::row(lane, v) * ldx + li];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    c = Mfma<T>::mma(-pl[li * LDT + 4 * ks + kk], lt[(size_t)li * ldl + 4 * ks + kk], c);
#pragma unroll
                for (int v = 0; v < 4; ++v) tile[(size_t)Mfma<T>::row(lane, v) * ldx + li] = c[v];
            }
        }
        __syncthreads();
    }
}

// ---- plain (VALU, unblocked) versions for cross-checking ----------------------------------------------------------
template <typename T> __device__ void potrf_plain(T *A, int ld, int n, int nrows, int *fail, int fail_base) {
    const int t = threadIdx.x;
    for (int j = 0; j < n; ++j) {
        __syncthreads();
        T d = A[(size_t)j * ld + j];
        if (!(d > (T)0)) {
            if (t == 0 && *fail == 0) *fail = fail_base + j + 1;
            d = (T)1;
        }
        const T piv = sqrt(d);
        __syncthreads();
        for (int i = j + t; i < nrows; i += 256) A[(size_t)i * ld + j] = (i == j) ? piv : A[(size_t)i * ld + j] / piv;
        __syncthreads();
        const int rem = nrows - j - 1, remc = n - j - 1;
        for (int e = t; e < rem * remc; e += 256) {
            const int i = j + 1 + e / remc, c = j + 1 + e % remc;
            if (c <= i || i >= n) A[(size_t)i * ld + c] -= A[(size_t)i * ld + j] * A[(size_t)c * ld + j];
        }
    }
    __syncthreads();
}
// X = L^-1 B, column per thread
template <typename T> __device__ void trsm_left_plain(const T *L, int ldl, T *Bm, int ldb, int n, int ncols) {
    for (int c = threadIdx.x; c < ncols; c += 256)
        for (int i = 0; i < n; ++i) {
            T v = Bm[(size_t)i * ldb + c];
            for (int k = 0; k < i; ++k) v -= L[(size_t)i * ldl + k] * Bm[(size_t)k * ldb + c];
            Bm[(size_t)i * ldb + c] = v / L[(size_t)i * ldl + i];
        }
    __syncthreads();
}
// T = X L^-T, row per thread (all columns)
template <typename T> __device__ void trsm_right_plain(const T *L, int ldl, T *X, int ldx, int n, int nrows) {
    for (int r = threadIdx.x; r < nrows; r += 256)
        for (int j = 0; j < n; ++j) {
            T v = X[(size_t)r * ldx + j];
            for (int k = 0; k < j; ++k) v -= X[(size_t)r * ldx + k] * L[(size_t)j * ldl + k];
            X[(size_t)r * ldx + j] = v / L[(size_t)j * ldl + j];
        }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// The fused Cholesky chain of one output dim (dp_gp_lvm.py:115-145).  Workspace per d (elements of TL):
//   Kb [(Mp+16) x Mp] : K_uu (written by the gram kernel into [0,M)x[0,M)) -> L_uu ; border row Mp: v^T -> (L^-1 v)^T
//   Pb [(Mp+16) x Mp] : Psi2 -> L^-1 Psi2 -> T2 = L^-1 Psi2 L^-T -> A = beta T2 + I -> L_A ; border row: w^T -> (L_A^-1 w)^T
//   dinv [nb x 256]   : inverted diagonal tiles of L_uu
// ---------------------------------------------------------------------------------------------------------------
size_t la_chain_ws_elems(int M) {
    const int Mp = dpgp_round_up(M, 16);
    return (size_t)2 * (Mp + 16) * Mp + (size_t)(Mp / 16) * 256;
}

template <typename TP, typename TL>
__global__ __launch_bounds__(256) void la_chain_kernel(int D, int N, int M, int Mp, const TP *__restrict__ psi2_part,
                                                       int ns2, const double *__restrict__ v_part, int ns1,
                                                       const double *__restrict__ alpha,
                                                       const double *__restrict__ beta, const double *__restrict__ yy,
                                                       double *__restrict__ terms, int *__restrict__ info,
                                                       TL *__restrict__ ws, size_t ws_stride, int plain) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // first LA_LDS_HDR bytes: reduction scratch + failure flag (no static __shared__ in front of the dynamic region)
    double *scratch = reinterpret_cast<double *>(smem_raw);
    int &fail = *reinterpret_cast<int *>(smem_raw + 64);
    TL *lds = reinterpret_cast<TL *>(smem_raw + LA_LDS_HDR);
    const int d = blockIdx.x, t = threadIdx.x, nb = Mp / 16;
    TL *Kb = ws + (size_t)d * ws_stride, *Pb = Kb + (size_t)(Mp + 16) * Mp, *dinv = Pb + (size_t)(Mp + 16) * Mp;
    if (t == 0) fail = 0;
    // ---- assemble: identity padding of K, border row v^T, Psi2 summed over slabs and mirrored ----
    for (int e = t; e < (Mp + 16) * Mp; e += 256) {
        const int i = e / Mp, j = e - i * Mp;
        if (i >= M || j >= M) {
            TL kv = (i == j) ? (TL)1 : (TL)0;
            if (i == Mp && j < M) {
                double a = 0.0;
                for (int k = 0; k < ns1; ++k) a += v_part[((size_t)k * D + d) * M + j];
                kv = (TL)a;
            }
            Kb[e] = kv;
        }
        TL pv = 0;
        if (i < Mp) {
            const int a_ = i > j ? i : j, c_ = i > j ? j : i;   // slabs hold the lower triangle; mirror it
            double a = 0.0;
            for (int k = 0; k < ns2; ++k) a += (double)psi2_part[((size_t)k * D + d) * (size_t)Mp * Mp + (size_t)a_ * Mp + c_];
            pv = (TL)a;
        }
        Pb[e] = pv;
    }
    __syncthreads();
    // ---- L = chol(K_uu), border -> w = L^-1 v  (dp_gp_lvm.py:116,132) ----
    if (plain) potrf_plain<TL>(Kb, Mp, Mp, Mp + 1, &fail, 0);
    else potrf_blocked<TL>(Kb, Mp, nb, nb + 1, lds, dinv, &fail, 0);
    __syncthreads();
    const int fail1 = fail;
    __syncthreads();
    if (t == 0) fail = 0;
    // ---- T2 = L^-1 Psi2 L^-T  (:118-121) ----
    if (plain) {
        trsm_left_plain<TL>(Kb, Mp, Pb, Mp, Mp, Mp);
        trsm_right_plain<TL>(Kb, Mp, Pb, Mp, Mp, Mp);
    } else {
        trsm_left_blocked<TL>(Kb, Mp, dinv, Pb, Mp, nb, nb, lds);
        trsm_right_lower_blocked<TL>(Kb, Mp, dinv, Pb, Mp, nb, lds);
    }
    __syncthreads();
    // ---- trace, A = beta T2 + I (lower part; :124-126), border row of A <- w^T ----
    const TL be = (TL)beta[d];
    double tr = 0.0;
    for (int e = t; e < (Mp + 16) * Mp; e += 256) {
        const int i = e / Mp, j = e - i * Mp;
        if (i < Mp) {
            if (j <= i) {
                TL v = Pb[e];
                if (i == j && i < M) tr += (double)v;
                Pb[e] = (i < M && j < M) ? be * v + ((i == j) ? (TL)1 : (TL)0) : ((i == j) ? (TL)1 : (TL)0);
            }
        } else {
            Pb[e] = (i == Mp) ? Kb[e] : (TL)0;
        }
    }
    tr = block_sum(tr, scratch);
    __syncthreads();
    // ---- L_A = chol(A), border -> L_A^-1 w  (:127,133) ----
    if (plain) potrf_plain<TL>(Pb, Mp, Mp, Mp + 1, &fail, 0);
    else potrf_blocked<TL>(Pb, Mp, nb, nb + 1, lds, (TL *)nullptr, &fail, 0);
    __syncthreads();
    double ld = 0.0, cc = 0.0;
    for (int i = t; i < M; i += 256) {
        ld += log((double)Pb[(size_t)i * Mp + i]);
        const double c = (double)Pb[(size_t)Mp * Mp + i];
        cc += c * c;
    }
    ld = block_sum(ld, scratch);
    cc = block_sum(cc, scratch);
    if (t == 0) {
        const double b_ = beta[d], a_ = alpha[d];
        double *o = terms + (size_t)d * 5;
        const int f = fail1 ? fail1 : (fail ? M + fail : 0);
        info[d] = f;
        const double nan_ = __longlong_as_double(0x7ff8000000000000LL);
        o[0] = 0.5 * N * (log(b_) - DPGP_LOG_2PI);
        o[1] = f ? nan_ : -ld;
        o[2] = f ? nan_ : 0.5 * b_ * (tr - a_ * N);
        o[3] = -0.5 * b_ * yy[d];
        o[4] = f ? nan_ : 0.5 * b_ * b_ * cc;
    }
}

static size_t la_lds_bytes(int Mp, size_t elem) {
    return LA_LDS_HDR + elem * (size_t)(16 * LDT) * (size_t)(Mp / 16 + 2);
}

template <typename TP, typename TL>
int launch_la_chain(int D, int N, int M, TL *kuu_ws, const TP *psi2_part, int ns2, const double *v_part, int ns1,
                    const double *alpha, const double *beta, const double *yy, double *terms, int *info, TL *ws,
                    int algo, hipStream_t st) {
    (void)kuu_ws;
    const int Mp = dpgp_round_up(M, 16);
    size_t lds = la_lds_bytes(Mp, sizeof(TL));
    auto kern = la_chain_kernel<TP, TL>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3(D), dim3(256), lds, st, D, N, M, Mp, psi2_part, ns2, v_part, ns1, alpha, beta, yy,
                       terms, info, ws, la_chain_ws_elems(M), algo == DPGP_ALGO_PLAIN ? 1 : 0);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
template int launch_la_chain<float, float>(int, int, int, float *, const float *, int, const double *, int,
                                           const double *, const double *, const double *, double *, int *, float *,
                                           int, hipStream_t);
template int launch_la_chain<float, double>(int, int, int, double *, const float *, int, const double *, int,
                                            const double *, const double *, const double *, double *, int *, double *,
                                            int, hipStream_t);
template int launch_la_chain<double, double>(int, int, int, double *, const double *, int, const double *, int,
                                             const double *, const double *, const double *, double *, int *, double *,
                                             int, hipStream_t);

__global__ __launch_bounds__(256) void sum_terms_kernel(int D, const double *__restrict__ terms,
                                                        double *__restrict__ sums) {
    __shared__ double scratch[8];
    double a = 0.0;
    for (int i = threadIdx.x; i < D * 5; i += 256) a += terms[i];
    a = block_sum(a, scratch);
    if (threadIdx.x == 0) sums[0] = a;
}
int launch_sum_terms(int D, const double *terms, double *sums, hipStream_t st) {
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(sum_terms_kernel, dim3(1), dim3(256), 0, st, D, terms, sums);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Stand-alone batched potrf / trsm (C ABI): copy into an identity-padded workspace, run the blocked routine, copy back.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void potrf_batched_kernel(int M, int Mp, T *__restrict__ a, int *__restrict__ info,
                                                            T *__restrict__ ws, int plain) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    int &fail = *reinterpret_cast<int *>(smem_raw + 64);
    T *lds = reinterpret_cast<T *>(smem_raw + LA_LDS_HDR);
    const int b = blockIdx.x, t = threadIdx.x;
    T *A = a + (size_t)b * M * M, *W = ws + (size_t)b * Mp * Mp;
    if (t == 0) fail = 0;
    for (int e = t; e < Mp * Mp; e += 256) {
        const int i = e / Mp, j = e - i * Mp;
        W[e] = (i < M && j < M) ? A[(size_t)i * M + j] : ((i == j) ? (T)1 : (T)0);
    }
    __syncthreads();
    if (plain) potrf_plain<T>(W, Mp, Mp, Mp, &fail, 0);
    else potrf_blocked<T>(W, Mp, Mp / 16, Mp / 16, lds, (T *)nullptr, &fail, 0);
    __syncthreads();
    for (int e = t; e < M * M; e += 256) {
        const int i = e / M, j = e - i * M;
        A[e] = (j <= i) ? W[(size_t)i * Mp + j] : (T)0;
    }
    if (t == 0) info[b] = fail;
}

template <typename T>
__global__ __launch_bounds__(256) void trsm_batched_kernel(int M, int K, int Mp, int Kp, const T *__restrict__ l,
                                                           T *__restrict__ rhs, T *__restrict__ ws, int plain) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *lds = reinterpret_cast<T *>(smem_raw + LA_LDS_HDR);
    const int b = blockIdx.x, t = threadIdx.x, nb = Mp / 16;
    const T *Lg = l + (size_t)b * M * M;
    T *R = rhs + (size_t)b * M * K;
    T *Lw = ws + (size_t)b * ((size_t)Mp * Mp + (size_t)Mp * Kp + (size_t)nb * 256), *Rw = Lw + (size_t)Mp * Mp,
      *dinv = Rw + (size_t)Mp * Kp;
    for (int e = t; e < Mp * Mp; e += 256) {
        const int i = e / Mp, j = e - i * Mp;
        Lw[e] = (i < M && j < M) ? ((j <= i) ? Lg[(size_t)i * M + j] : (T)0) : ((i == j) ? (T)1 : (T)0);
    }
    for (int e = t; e < Mp * Kp; e += 256) {
        const int i = e / Kp, j = e - i * Kp;
        Rw[e] = (i < M && j < K) ? R[(size_t)i * K + j] : (T)0;
    }
    __syncthreads();
    if (plain) {
        trsm_left_plain<T>(Lw, Mp, Rw, Kp, Mp, Kp);
    } else {
        const int wv = t >> 6;
        for (int k = wv; k < nb; k += 4)
            diag_tile<T, false>(Lw + (size_t)(16 * k) * Mp + 16 * k, Mp, lds + (size_t)wv * 16 * LDT, dinv + k * 256,
                                (int *)nullptr, 0);
        __syncthreads();
        trsm_left_blocked<T>(Lw, Mp, dinv, Rw, Kp, nb, Kp / 16, lds);
    }
    __syncthreads();
    for (int e = t; e < M * K; e += 256) {
        const int i = e / K, j = e - i * K;
        R[e] = Rw[(size_t)i * Kp + j];
    }
}

extern "C" size_t dpgp_potrf_workspace_bytes(int B, int M, int elem_size) {
    if (B <= 0 || M <= 0) return 0;
    const int Mp = dpgp_round_up(M, 16);
    return dpgp_align256((size_t)elem_size * B * Mp * Mp);
}
extern "C" size_t dpgp_trsm_workspace_bytes(int B, int M, int K, int elem_size) {
    if (B <= 0 || M <= 0 || K <= 0) return 0;
    const int Mp = dpgp_round_up(M, 16), Kp = dpgp_round_up(K, 16);
    return dpgp_align256((size_t)elem_size * B * ((size_t)Mp * Mp + (size_t)Mp * Kp + (size_t)(Mp / 16) * 256));
}

template <typename T>
static int potrf_api(int B, int M, T *a, int *info, void *ws, size_t ws_bytes, int algo, void *stream) {
    if (B <= 0) return -1;
    if (M <= 0) return -2;
    if (!a) return -3;
    if (!info) return -4;
    if (!ws) return -5;
    if (ws_bytes < dpgp_potrf_workspace_bytes(B, M, sizeof(T))) return -6;
    if (algo != DPGP_ALGO_AUTO && algo != DPGP_ALGO_PLAIN) return -7;
    const int Mp = dpgp_round_up(M, 16);
    size_t lds = la_lds_bytes(Mp, sizeof(T));
    auto kern = potrf_batched_kernel<T>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, (hipStream_t)stream, M, Mp, a, info, (T *)ws,
                       algo == DPGP_ALGO_PLAIN ? 1 : 0);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
extern "C" int dpgp_potrf_batched_f32(int B, int M, float *a, int *info, void *ws, size_t ws_bytes, int algo,
                                      void *stream) {
    return potrf_api<float>(B, M, a, info, ws, ws_bytes, algo, stream);
}
extern "C" int dpgp_potrf_batched_f64(int B, int M, double *a, int *info, void *ws, size_t ws_bytes, int algo,
                                      void *stream) {
    return potrf_api<double>(B, M, a, info, ws, ws_bytes, algo, stream);
}

template <typename T>
static int trsm_api(int B, int M, int K, const T *l, T *rhs, void *ws, size_t ws_bytes, int algo, void *stream) {
    if (B <= 0) return -1;
    if (M <= 0) return -2;
    if (K <= 0) return -3;
    if (!l) return -4;
    if (!rhs) return -5;
    if (!ws) return -6;
    if (ws_bytes < dpgp_trsm_workspace_bytes(B, M, K, sizeof(T))) return -7;
    if (algo != DPGP_ALGO_AUTO && algo != DPGP_ALGO_PLAIN) return -8;
    const int Mp = dpgp_round_up(M, 16), Kp = dpgp_round_up(K, 16);
    size_t tiles = (size_t)(Kp / 16 + 1);
    if (tiles < 4) tiles = 4;   // the diagonal-tile inversion uses one dinv slot per wave
    size_t lds = LA_LDS_HDR + sizeof(T) * (size_t)(16 * LDT) * tiles;
    auto kern = trsm_batched_kernel<T>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, (hipStream_t)stream, M, K, Mp, Kp, l, rhs, (T *)ws,
                       algo == DPGP_ALGO_PLAIN ? 1 : 0);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
extern "C" int dpgp_trsm_batched_f32(int B, int M, int K, const float *l, float *rhs, void *ws, size_t ws_bytes, int algo,
                                     void *stream) {
    return trsm_api<float>(B, M, K, l, rhs, ws, ws_bytes, algo, stream);
}
extern "C" int dpgp_trsm_batched_f64(int B, int M, int K, const double *l, double *rhs, void *ws, size_t ws_bytes,
                                     int algo, void *stream) {
    return trsm_api<double>(B, M, K, l, rhs, ws, ws_bytes, algo, stream);
}
