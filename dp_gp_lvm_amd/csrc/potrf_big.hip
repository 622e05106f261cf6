// Batched Cholesky for matrices too large for one workgroup's LDS (M > 128 in fp64): right-looking, 64 x 64 blocks,
// every step spread over the whole GPU (reference: tf.cholesky on [D,M,M], dp_gp_lvm.py:116,127, at BASELINE config 4's
// M = 512).  Per step k, for all B matrices at once:
//   1. diag   (B workgroups)            : L_kk = chol(A_kk) and W_kk = L_kk^-1, LDS-resident (potrf_lds + trtri_lds); a launch
//                                         of its own only for k = 0, afterwards done by the syrk workgroup of that block
//   2. panel  (B x blocks below)        : P_i = A_ik W_kk^T                                  (64x64x64 MFMA GEMM)
//   3. syrk   (B x lower block pairs)   : A_ij -= P_i P_j^T, k < j <= i                      (the trailing update)
// The single-workgroup routine (potrf_blocked) runs the same steps inside one workgroup: with B = 64 matrices only 64 of
// the 256 compute units work (1.5 TFLOP/s fp64 at M = 512).  Matrices live in a workspace padded with an identity block
// to a multiple of 64; all kernels are bounded to 256 VGPRs (two workgroups per compute unit).
#include "internal.h"
#include "linalg_dev.h"

#define PB 64                 // block edge
#define PB_LD (PB + 2)        // LDS row stride of a staged block (elements): 16-byte aligned rows in fp64 and fp32

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

// step 1 for diagonal block k of matrix b: L_kk = chol(A_kk) in place, winv[b] = L_kk^-1 (row-major 64 x 64, zeros above
// the diagonal).  smem_raw: LA_LDS_HDR + 11 LDS tiles.  Called by pbig_diag (k = 0) and, for k + 1, by the workgroup of
// pbig_syrk that has just finished the trailing update of that block (look-ahead: the ~40 us of this latency-bound step
// hide behind the other workgroups' updates).
template <typename T>
__device__ __forceinline__ void pbig_diag_body(int Mw, int k, int b, T *__restrict__ w, T *__restrict__ winv,
                                               T *__restrict__ dinv_g, int *__restrict__ info, unsigned char *smem_raw) {
    int &fail = *reinterpret_cast<int *>(smem_raw + 64);
    T *dinv = reinterpret_cast<T *>(smem_raw + LA_LDS_HDR), *tiles = dinv + TSZ;
    const int t = threadIdx.x;
    constexpr int nb = PB / 16, nlow = nb * (nb + 1) / 2;
    T *Akk = w + (size_t)b * Mw * Mw + (size_t)(PB * k) * Mw + PB * k;
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
    for (int e = t; e < PB * PB; e += 256) {                     // L_kk back (lower; the upper part of A is never read)
        const int i = e >> 6, j = e & 63;
        if (j <= i) Akk[(size_t)i * Mw + j] = tiles[lds_tile_index(i >> 4, j >> 4, nb) * TSZ + (i & 15) * LDT + (j & 15)];
    }
    if (t == 0 && fail && info[b] == 0) info[b] = PB * k + fail;
    __threadfence_block();
    __syncthreads();
    trtri_lds<T>(tiles, dinv, dg, nb);
    T *Wi = winv + (size_t)b * PB * PB;
    for (int e = t; e < PB * PB; e += 256) {
        const int i = e >> 6, j = e & 63;
        Wi[e] = (j <= i) ? tiles[lds_tile_index(i >> 4, j >> 4, nb) * TSZ + (i & 15) * LDT + (j & 15)] : (T)0;
    }
}
template <typename T>
__global__ __launch_bounds__(256, 2) void pbig_diag(int Mw, int k, T *__restrict__ w, T *__restrict__ winv,
                                                    T *__restrict__ dinv_g, int *__restrict__ info) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    pbig_diag_body<T>(Mw, k, (int)blockIdx.x, w, winv, dinv_g, info, smem_raw);
}

// stage a 64 x 64 block (row-major, leading dimension ld) into LDS [64][PB_LD]
template <typename T> __device__ __forceinline__ void pbig_stage(const T *__restrict__ g, int ld, T *__restrict__ s) {
    typedef T t4 __attribute__((ext_vector_type(4)));
    for (int e = threadIdx.x; e < PB * PB / 4; e += 256) {
        const int r = e >> 4, c4 = (e & 15) * 4;
        const t4 v = *reinterpret_cast<const t4 *>(g + (size_t)r * ld + c4);
        T *d = s + r * PB_LD + c4;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
}

// C (64 x 64, this wave's 16 rows x 4 column tiles) (+)= sign * X Y^T with X, Y staged [64][PB_LD]
template <typename T, bool LOAD_C>
__device__ __forceinline__ void pbig_block_mma(const T *xs, const T *ys, T *__restrict__ cg, int ldc, T sign) {
    typedef typename Mfma<T>::acc_t acc_t;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, kk = lane >> 4;
    acc_t c[4];
#pragma unroll
    for (int J = 0; J < 4; ++J) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
            c[J][v] = LOAD_C ? cg[(size_t)(16 * wv + Mfma<T>::row(lane, v)) * ldc + 16 * J + li] : (T)0;
    }
#pragma unroll 4
    for (int ks = 0; ks < PB / 4; ++ks) {
        const T xv = sign * xs[(16 * wv + li) * PB_LD + 4 * ks + kk];
#pragma unroll
        for (int J = 0; J < 4; ++J) c[J] = Mfma<T>::mma(xv, ys[(16 * J + li) * PB_LD + 4 * ks + kk], c[J]);
    }
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int v = 0; v < 4; ++v) cg[(size_t)(16 * wv + Mfma<T>::row(lane, v)) * ldc + 16 * J + li] = c[J][v];
}

// step 2: P_i = A_ik W_kk^T in place, i = k + 1 + blockIdx.y
template <typename T>
__global__ __launch_bounds__(256, 2) void pbig_panel(int Mw, int k, T *__restrict__ w, const T *__restrict__ winv) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *xs = reinterpret_cast<T *>(smem_raw), *ys = xs + PB * PB_LD;
    const int b = blockIdx.x, i = k + 1 + blockIdx.y;
    T *Aik = w + (size_t)b * Mw * Mw + (size_t)(PB * i) * Mw + PB * k;
    pbig_stage<T>(Aik, Mw, xs);
    pbig_stage<T>(winv + (size_t)b * PB * PB, PB, ys);
    __syncthreads();
    pbig_block_mma<T, false>(xs, ys, Aik, Mw, (T)1);
}

// step 3: A_ij -= P_i P_j^T for the blockIdx.y-th pair k < j <= i of the trailing lower triangle
template <typename T>
__global__ __launch_bounds__(256, 2) void pbig_syrk(int Mw, int k, T *__restrict__ w, T *__restrict__ winv,
                                                    T *__restrict__ dinv_g, int *__restrict__ info) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *xs = reinterpret_cast<T *>(smem_raw), *ys = xs + PB * PB_LD;
    const int b = blockIdx.x, p = blockIdx.y;
    int ii = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
    while ((ii + 1) * (ii + 2) / 2 <= p) ++ii;
    while (ii * (ii + 1) / 2 > p) --ii;
    const int jj = p - ii * (ii + 1) / 2, i = k + 1 + ii, j = k + 1 + jj;
    T *Wb = w + (size_t)b * Mw * Mw;
    pbig_stage<T>(Wb + (size_t)(PB * i) * Mw + PB * k, Mw, xs);
    if (i != j) pbig_stage<T>(Wb + (size_t)(PB * j) * Mw + PB * k, Mw, ys);
    __syncthreads();
    pbig_block_mma<T, true>(xs, (i != j) ? ys : xs, Wb + (size_t)(PB * i) * Mw + PB * j, Mw, (T)-1);
    if (p == 0) {                      // block (k+1, k+1) is final: factor and invert it now (look-ahead of step k + 1)
        __threadfence_block();
        __syncthreads();
        pbig_diag_body<T>(Mw, k + 1, b, w, winv, dinv_g, info, smem_raw);
    }
}

// M a multiple of 64: the factorisation ran in place on the caller's array; only the zeros above the diagonal remain to write
template <typename T> __global__ __launch_bounds__(256, 2) void pbig_zero_upper(int M, T *__restrict__ a) {
    T *A = a + (size_t)blockIdx.y * M * M;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < (size_t)M * M; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e / M), j = (int)(e - (size_t)i * M);
        if (j > i) A[e] = (T)0;
    }
}

size_t potrf_big_ws_elems(int B, int M) {
    const size_t Mw = (size_t)dpgp_round_up(M, PB);
    return (size_t)B * (Mw * Mw + PB * PB + (PB / 16) * 256);
}

template <typename T>
int launch_potrf_big(int B, int M, T *a, int *info, T *ws, hipStream_t st) {
    const int Mw = dpgp_round_up(M, PB), nblk = Mw / PB;
    const bool in_place = (Mw == M);
    T *w = in_place ? a : ws, *winv = ws + (size_t)B * Mw * Mw, *dinv_g = winv + (size_t)B * PB * PB;
    const size_t lds_diag = LA_LDS_HDR + sizeof(T) * (size_t)TSZ * (1 + (PB / 16) * (PB / 16 + 1) / 2);
    const size_t lds_blk = sizeof(T) * (size_t)2 * PB * PB_LD;        // (>= lds_diag: pbig_syrk reuses it for the look-ahead)
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(pbig_panel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_blk) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void *>(pbig_syrk<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_blk) != hipSuccess)
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
        const int nrem = nblk - k - 1;
        DPGP_PRELAUNCH(); hipLaunchKernelGGL((pbig_panel<T>), dim3(B, nrem), dim3(256), lds_blk, st, Mw, k, w, (const T *)winv);
        DPGP_LAUNCH_CHECK();
        DPGP_PRELAUNCH(); hipLaunchKernelGGL((pbig_syrk<T>), dim3(B, nrem * (nrem + 1) / 2), dim3(256), lds_blk, st, Mw, k, w, winv,
                           dinv_g, info);
        DPGP_LAUNCH_CHECK();
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
