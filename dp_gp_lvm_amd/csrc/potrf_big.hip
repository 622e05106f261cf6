// Batched Cholesky for matrices too large for one workgroup's LDS (M > 128 in fp64): right-looking, 128-wide panels,
// every step spread over the whole GPU (reference: tf.cholesky on [D,M,M], dp_gp_lvm.py:116,127, at BASELINE config 4's
// M = 512).  Per panel step k, for all B matrices at once:
//   1. diag   (B workgroups)              : L_kk = chol(A_kk) (128 x 128) and W_kk = L_kk^-1, LDS-resident (potrf_lds +
//                                           trtri_lds); a launch of its own only for k = 0 (see 3.)
//   2. panel  (B x row blocks x 2)        : P_i = A_ik W_kk^T for the 64-row blocks below, into a panel buffer
//   3. update (B x lower pairs of blocks) : A_IJ -= P_I P_J^T (128 x 128 x 128 MFMA block products), the trailing update.
//                                           The workgroup that owns the NEXT diagonal 128 x 128 block updates it and factors
//                                           + inverts it right away (look-ahead: the ~90 us of that latency-bound step
//                                           hide behind the other workgroups' updates).
// Panel width: with 64-wide panels the update does 8 flops per byte of the trailing matrix it reads and writes and was
// HBM-bound at 20-22 TFLOP/s (27-29 % of the fp64 MFMA peak); 128-wide panels double that.  The single-workgroup routine
// (potrf_blocked) uses 1 of 4 CUs at B = 64 (1.5 TFLOP/s).  Matrices are padded with an identity block to a multiple of
// 128 (in place when M is one already); all kernels are bounded to 256 VGPRs (two workgroups per compute unit).
#include <cstdlib>
#include "internal.h"
#include "linalg_dev.h"

#define PW 128                // panel width (K of the update)
#define RB 64                 // row / column block of the update
#define RB_LD (RB + 2)        // LDS row stride of a staged 64 x 64 block (elements): 16-byte aligned rows in fp64 and fp32

template <typename T>
__global__ __launch_bounds__(256, 2) void pbig_copy_in(int M, int Mw, const T *__restrict__ a, T *__restrict__ w) {
    const T *A = a + (size_t)blockIdx.y * M * M;
    T *W = w + (size_t)blockIdx.y * Mw * Mw;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)Mw * Mw; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / Mw), j = (int)(e - (size_t)i * Mw);
        W[e] = (i < M && j < M) ? A[(size_t)i * M + j] : ((i == j) ? (T)1 : (T)0);
    }
}
template <typename T>
__global__ __launch_bounds__(256, 2) void pbig_copy_out(int M, int Mw, T *__restrict__ a, const T *__restrict__ w) {
    T *A = a + (size_t)blockIdx.y * M * M;
    const T *W = w + (size_t)blockIdx.y * Mw * Mw;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)M * M; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / M), j = (int)(e - (size_t)i * M);
        A[e] = (j <= i) ? W[(size_t)i * Mw + j] : (T)0;
    }
}
// M a multiple of 128: the factorisation ran in place on the caller's array; only the zeros above the diagonal remain to write
template <typename T> __global__ __launch_bounds__(256, 2) void pbig_zero_upper(int M, T *__restrict__ a) {
    T *A = a + (size_t)blockIdx.y * M * M;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)M * M; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / M), j = (int)(e - (size_t)i * M);
        if (j > i) A[e] = (T)0;
    }
}

// step 1 for diagonal block k of matrix b: L_kk = chol(A_kk) in place, winv[b] = L_kk^-1 (row-major 128 x 128, zeros above
// the diagonal).  smem_raw: LA_LDS_HDR + 37 LDS tiles (80 KB in fp64).
template <typename T>
__device__ __forceinline__ void pbig_diag_body(int Mw, int k, int b, T *__restrict__ w, T *__restrict__ winv,
                                               T *__restrict__ dinv_g, int *__restrict__ info, unsigned char *smem_raw) {
    int &fail = *reinterpret_cast<int *>(smem_raw + 64);
    T *dinv = reinterpret_cast<T *>(smem_raw + LA_LDS_HDR), *tiles = dinv + TSZ;
    const int t = threadIdx.x;
    constexpr int nb = PW / 16, nlow = nb * (nb + 1) / 2;
    T *Akk = w + (size_t)b * Mw * Mw + (size_t)(PW * k) * Mw + PW * k;
    T *dg = dinv_g + (size_t)b * nb * 256;
    if (t == 0) fail = 0;
    for (int e = t; e < nlow * 256; e += 256) {
        const int tt = e >> 8, r = (e >> 4) & 15, c = e & 15;
        int I = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
        while ((I + 1) * (I + 2) / 2 <= tt) ++I;
        while (I * (I + 1) / 2 > tt) --I;
        const int J = tt - I * (I + 1) / 2;
        tiles[tt * TSZ + r * LDT + c] = Akk[(size_t)(16 * I + r) * Mw + 16 * J + c];
    }
    __syncthreads();
    potrf_lds<T, 2>(tiles, dinv, nb, nb, &fail, dg);
    __syncthreads();
    for (int e = t; e < PW * PW; e += 256) {                     // L_kk back (lower; the upper part of A is never read)
        const int i = e / PW, j = e - i * PW;
        if (j <= i) Akk[(size_t)i * Mw + j] = tiles[lds_tile_index(i >> 4, j >> 4, nb) * TSZ + (i & 15) * LDT + (j & 15)];
    }
    if (t == 0 && fail && info[b] == 0) info[b] = PW * k + fail;
    __threadfence_block();
    __syncthreads();
    trtri_lds<T>(tiles, dinv, dg, nb);
    T *Wi = winv + (size_t)b * PW * PW;
    for (int e = t; e < PW * PW; e += 256) {
        const int i = e / PW, j = e - i * PW;
        Wi[e] = (j <= i) ? tiles[lds_tile_index(i >> 4, j >> 4, nb) * TSZ + (i & 15) * LDT + (j & 15)] : (T)0;
    }
}
template <typename T>
__global__ __launch_bounds__(256, 2) void pbig_diag(int Mw, int k, T *__restrict__ w, T *__restrict__ winv,
                                                    T *__restrict__ dinv_g, int *__restrict__ info) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    pbig_diag_body<T>(Mw, k, (int)blockIdx.x, w, winv, dinv_g, info, smem_raw);
}

// stage a 64 x 64 block (row-major, leading dimension ld) into LDS [64][RB_LD]
template <typename T> __device__ __forceinline__ void pbig_stage(const T *__restrict__ g, int ld, T *__restrict__ s) {
    typedef T t4 __attribute__((ext_vector_type(4)));
    for (int e = threadIdx.x; e < RB * RB / 4; e += 256) {
        const int r = e >> 4, c4 = (e & 15) * 4;
        const t4 v = *reinterpret_cast<const t4 *>(g + (size_t)r * ld + c4);
        T *d = s + r * RB_LD + c4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
}
// this wave's 16 rows x 4 column tiles of a 64 x 64 block: c += sign * X Y^T over one staged K-half (64)
template <typename T>
__device__ __forceinline__ void pbig_mma_half(const T *xs, const T *ys, typename Mfma<T>::acc_t (&c)[4], T sign) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, kk = lane >> 4;
#pragma unroll 4
    for (int ks = 0; ks < RB / 4; ++ks) {
        const T xv = sign * xs[(16 * wv + li) * RB_LD + 4 * ks + kk];
#pragma unroll
        for (int J = 0; J < 4; ++J) c[J] = Mfma<T>::mma(xv, ys[(16 * J + li) * RB_LD + 4 * ks + kk], c[J]);
    }
}
template <typename T>
__device__ __forceinline__ void pbig_store(T *__restrict__ cg, int ldc, const typename Mfma<T>::acc_t (&c)[4]) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15;
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int v = 0; v < 4; ++v) cg[(size_t)(16 * wv + Mfma<T>::row(lane, v)) * ldc + 16 * J + li] = c[J][v];
}

// step 2: P_i[:, 64 c : 64 c + 64] = A_ik W_kk[64 c : 64 c + 64, :]^T  (row block i below the panel, column half c), into pbuf
template <typename T>
__global__ __launch_bounds__(256, 2) void pbig_panel(int Mw, int k, const T *__restrict__ w, const T *__restrict__ winv,
                                                     T *__restrict__ pbuf) {
    typedef typename Mfma<T>::acc_t acc_t;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *xs = reinterpret_cast<T *>(smem_raw), *ys = xs + RB * RB_LD;
    const int b = blockIdx.x, i = blockIdx.y, c = blockIdx.z;          // i: 64-row block index below the panel
    const T *Aik = w + (size_t)b * Mw * Mw + (size_t)(PW * (k + 1) + RB * i) * Mw + PW * k;
    const T *Wk = winv + (size_t)b * PW * PW + (size_t)(RB * c) * PW;
    acc_t acc[4];
#pragma unroll
    for (int J = 0; J < 4; ++J) acc[J] = (acc_t){0, 0, 0, 0};
    for (int h = 0; h < PW / RB; ++h) {
        if (h) __syncthreads();
        pbig_stage<T>(Aik + RB * h, Mw, xs);
        pbig_stage<T>(Wk + RB * h, PW, ys);
        __syncthreads();
        pbig_mma_half<T>(xs, ys, acc, (T)1);
    }
    pbig_store<T>(pbuf + ((size_t)b * Mw + RB * i) * PW + RB * c, PW, acc);
}

// one 128 x 128 trailing block: A_IJ -= P_I P_J^T, K = 128 staged in four quarters of 32 through LDS; every wave owns 32
// rows x 128 columns of the block in registers (16 result tiles).  128 x 128 rather than 64 x 64 per workgroup: the panel
// rows are fetched from memory once per 128 x 128 block (8 flops per byte moved instead of 5.5; the 64 x 64 version ran at
// 22 TFLOP/s whatever the panel width).  I == J also writes P_I into the panel columns of A (the finished L_Ik).
#define UB 128                // row / column block of the update
#define KQ 32                 // K staged per pass
#define KQ_LD (KQ + 2)
template <typename T> __device__ __forceinline__ void pbig_stage_q(const T *__restrict__ g, int ld, T *__restrict__ s) {
    typedef T t4 __attribute__((ext_vector_type(4)));
    for (int e = threadIdx.x; e < UB * KQ / 4; e += 256) {            // 128 rows x 32 columns
        const int r = e >> 3, c4 = (e & 7) * 4;
        const t4 v = *reinterpret_cast<const t4 *>(g + (size_t)r * ld + c4);
        T *d = s + r * KQ_LD + c4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
}
template <typename T>
__device__ __forceinline__ void pbig_update_block(int Mw, int k, int b, int I, int J, T *__restrict__ w,
                                                  const T *__restrict__ pbuf, T *xs, T *ys) {
    typedef typename Mfma<T>::acc_t acc_t;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, kk = lane >> 4;
    T *Wb = w + (size_t)b * Mw * Mw;
    const int r0 = PW * (k + 1);
    T *Cij = Wb + (size_t)(r0 + UB * I) * Mw + r0 + UB * J;
    const T *Pi = pbuf + ((size_t)b * Mw + UB * I) * PW, *Pj = pbuf + ((size_t)b * Mw + UB * J) * PW;
    acc_t acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int Jt = 0; Jt < 8; ++Jt)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                acc[i][Jt][v] = Cij[(size_t)(32 * wv + 16 * i + Mfma<T>::row(lane, v)) * Mw + 16 * Jt + li];
    for (int h = 0; h < PW / KQ; ++h) {
        __syncthreads();
        pbig_stage_q<T>(Pi + KQ * h, PW, xs);
        if (I != J) pbig_stage_q<T>(Pj + KQ * h, PW, ys);
        __syncthreads();
        const T *yy = (I != J) ? ys : xs;
#pragma unroll 2
        for (int ks = 0; ks < KQ / 4; ++ks) {
            const T x0 = -xs[(32 * wv + li) * KQ_LD + 4 * ks + kk], x1 = -xs[(32 * wv + 16 + li) * KQ_LD + 4 * ks + kk];
#pragma unroll
            for (int Jt = 0; Jt < 8; ++Jt) {
                const T yv = yy[(16 * Jt + li) * KQ_LD + 4 * ks + kk];
                acc[0][Jt] = Mfma<T>::mma(x0, yv, acc[0][Jt]);
                acc[1][Jt] = Mfma<T>::mma(x1, yv, acc[1][Jt]);
            }
        }
        if (I == J) {                                            // the finished panel block L_Ik = P_I goes back into A
            T *Lik = Wb + (size_t)(r0 + UB * I) * Mw + PW * k + KQ * h;
            for (int e = threadIdx.x; e < UB * KQ; e += 256) Lik[(size_t)(e >> 5) * Mw + (e & 31)] = xs[(e >> 5) * KQ_LD + (e & 31)];
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int Jt = 0; Jt < 8; ++Jt)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                Cij[(size_t)(32 * wv + 16 * i + Mfma<T>::row(lane, v)) * Mw + 16 * Jt + li] = acc[i][Jt][v];
}

// step 3.  blockIdx.y enumerates the lower pairs (I, J) of the 128-row blocks below the panel; pair (0,0) is the next
// diagonal block: its workgroup factors + inverts it right after updating it (look-ahead).
template <typename T>
__global__ __launch_bounds__(256, 2) void pbig_update(int Mw, int k, T *__restrict__ w, const T *__restrict__ pbuf,
                                                      T *__restrict__ winv, T *__restrict__ dinv_g, int *__restrict__ info,
                                                      int lookahead) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *xs = reinterpret_cast<T *>(smem_raw), *ys = xs + UB * KQ_LD;
    const int b = blockIdx.x, p = blockIdx.y;
    int I = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
    while ((I + 1) * (I + 2) / 2 <= p) ++I;
    while (I * (I + 1) / 2 > p) --I;
    pbig_update_block<T>(Mw, k, b, I, p - I * (I + 1) / 2, w, pbuf, xs, ys);
    if (p == 0 && lookahead) {
        __threadfence_block();
        __syncthreads();
        pbig_diag_body<T>(Mw, k + 1, b, w, winv, dinv_g, info, smem_raw);
    }
}

size_t potrf_big_ws_elems(int B, int M) {
    const size_t Mw = (size_t)dpgp_round_up(M, PW);
    return (size_t)B * (Mw * Mw + PW * PW + (PW / 16) * 256 + Mw * PW);
}

template <typename T>
int launch_potrf_big(int B, int M, T *a, int *info, T *ws, hipStream_t st) {
    const int Mw = dpgp_round_up(M, PW), nblk = Mw / PW;
    const bool in_place = (Mw == M);
    // DPGP_POTRF_NO_LOOKAHEAD=1 (profiling only): diagonal blocks as launches of their own, so that the update kernel's own
    // rate can be read off a kernel trace
    const char *nl_ = getenv("DPGP_POTRF_NO_LOOKAHEAD");
    const int lookahead = (nl_ && nl_[0] == '1') ? 0 : 1;
    T *w = in_place ? a : ws, *winv = ws + (size_t)B * Mw * Mw, *dinv_g = winv + (size_t)B * PW * PW,
      *pbuf = dinv_g + (size_t)B * (PW / 16) * 256;
    // enough fp64 matrices to give every compute unit its own: one persistent workgroup per matrix (potrf_persist.hip).
    // DPGP_POTRF_PERSISTENT=0 / 1 (tests, profiling): never / whenever the element type allows it
    if constexpr (sizeof(T) == 8) {
        const char *pe_ = getenv("DPGP_POTRF_PERSISTENT");
        const bool use = pe_ ? pe_[0] == '1' : potrf_persist_applicable(B, M, (int)sizeof(T));
        if (use) {
            const int cp0 = dpgp_ceil_div(Mw * Mw, 256 * 8);
            if (!in_place) {
                DPGP_PRELAUNCH(); hipLaunchKernelGGL((pbig_copy_in<T>), dim3(cp0, B), dim3(256), 0, st, M, Mw, (const T *)a, w);
                DPGP_LAUNCH_CHECK();
            }
            const int rc = launch_potrf_persist(B, Mw, reinterpret_cast<double *>(w), info, st);
            if (rc != DPGP_OK) return rc;
            if (!in_place) {
                DPGP_PRELAUNCH(); hipLaunchKernelGGL((pbig_copy_out<T>), dim3(cp0, B), dim3(256), 0, st, M, Mw, a, (const T *)w);
                DPGP_LAUNCH_CHECK();
            }
            return DPGP_OK;
        }
    }
    const size_t lds_diag = LA_LDS_HDR + sizeof(T) * (size_t)TSZ * (1 + (PW / 16) * (PW / 16 + 1) / 2);
    const size_t lds_blk = sizeof(T) * (size_t)2 * RB * RB_LD, lds_u = sizeof(T) * (size_t)2 * UB * KQ_LD;
    const size_t lds_upd = lds_u > lds_diag ? lds_u : lds_diag;        // the look-ahead workgroup reuses it for the diagonal block
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(pbig_diag<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_diag) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(pbig_panel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_blk) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(pbig_update<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_upd) != hipSuccess)
        return DPGP_ERR_LAUNCH;
    if (hipMemsetAsync(info, 0, sizeof(int) * (size_t)B, st) != hipSuccess) return DPGP_ERR_LAUNCH;
    const int cp = dpgp_ceil_div(Mw * Mw, 256 * 8);
    if (!in_place) {
        DPGP_PRELAUNCH(); hipLaunchKernelGGL((pbig_copy_in<T>), dim3(cp, B), dim3(256), 0, st, M, Mw, (const T *)a, w);
        DPGP_LAUNCH_CHECK();
    }
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((pbig_diag<T>), dim3(B), dim3(256), lds_diag, st, Mw, 0, w, winv, dinv_g, info);
    DPGP_LAUNCH_CHECK();
    for (int k = 0; k + 1 < nblk; ++k) {
        const int nrb = (Mw - PW * (k + 1)) / RB;                 // 64-row blocks below the panel (even, >= 2)
        DPGP_PRELAUNCH(); hipLaunchKernelGGL((pbig_panel<T>), dim3(B, nrb, PW / RB), dim3(256), lds_blk, st, Mw, k, (const T *)w, (const T *)winv,
                           pbuf);
        DPGP_LAUNCH_CHECK();
        const int nub = nrb / 2;                                   // 128-row blocks below the panel
        DPGP_PRELAUNCH(); hipLaunchKernelGGL((pbig_update<T>), dim3(B, nub * (nub + 1) / 2), dim3(256), lds_upd, st, Mw, k, w, (const T *)pbuf,
                           winv, dinv_g, info, lookahead);
        DPGP_LAUNCH_CHECK();
        if (!lookahead) {
            DPGP_PRELAUNCH(); hipLaunchKernelGGL((pbig_diag<T>), dim3(B), dim3(256), lds_diag, st, Mw, k + 1, w, winv, dinv_g, info);
            DPGP_LAUNCH_CHECK();
        }
    }
    if (in_place) {
        DPGP_PRELAUNCH(); hipLaunchKernelGGL((pbig_zero_upper<T>), dim3(cp, B), dim3(256), 0, st, M, a);
    } else {
        DPGP_PRELAUNCH(); hipLaunchKernelGGL((pbig_copy_out<T>), dim3(cp, B), dim3(256), 0, st, M, Mw, a, (const T *)w);
    }
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
template int launch_potrf_big<float>(int, int, float *, int *, float *, hipStream_t);
template int launch_potrf_big<double>(int, int, double *, int *, double *, hipStream_t);
