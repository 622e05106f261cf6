// Device-side linear algebra shared by linalg.hip and psi2.hip (16x16-tiled MFMA Cholesky / triangular solves).
#pragma once
#include "internal.h"

#ifndef STAMP          // diagnostic stamps are only compiled into linalg.hip of the scratch build (-DDPGP_PROFILE_CHAIN)
#define STAMP(i)
#define ACC_BEGIN()
#define ACC_END(i)
#endif

#define LDT 17          // LDS tile row stride (16 + 1 pad)
#define TSZ (16 * LDT)  // elements per LDS tile
#define LA_LDS_HDR 128  // bytes at the start of the dynamic LDS region: 8 doubles of reduction scratch + fail flag
#define LA_LDS_LIMIT (150 * 1024)

__device__ __forceinline__ float dpgp_rsqrt(float x) { return rsqrtf(x); }
__device__ __forceinline__ double dpgp_rsqrt(double x) { return rsqrt(x); }

// broadcast of lane `src` (a compile-time constant after unrolling) through an SGPR: v_readlane_b32, no LDS crossbar
__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
__device__ __forceinline__ double lane_bcast(double v, int src) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}


// ---- diagonal tile: Cholesky (optional) + inverse, by the calling wave; lanes 0..15 hold one row each ------------
// A: tile origin (global memory or LDS, row stride ld).  On exit (FACTOR): tile holds L (upper zeroed).  dinv_lds[16][LDT]
// and, if non-null, dinv_glob[16][16] receive L^-1.  *fail (LDS) gets base+j+1 for the first non-positive pivot.
// INV_FROM_LDS (tile lives in LDS): the inverse reads L back from the tile with wave-uniform addresses (LDS broadcast
// reads) instead of 120 more cross-lane broadcasts — the v_readlane form keeps ~240 SGPRs live and the compiler spills
// them through v_writelane (measured 4.6 us per fp64 tile, ~2200 instructions).
template <typename T, bool FACTOR, bool INV_FROM_LDS>
__device__ __forceinline__ void diag_tile(T *A, int ld, T *dinv_lds, T *dinv_glob, int *fail, int base) {
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
            rinv[j] = dpgp_rsqrt(d);                 // one rsq + refinement instead of sqrt followed by a division
            const T piv = d * rinv[j];
            a[j] = (li == j) ? piv : a[j] * rinv[j];  // (rows li < j: don't-care)
#pragma unroll
            for (int c = j + 1; c < 16; ++c) {
                const T lcj = lane_bcast(a[j], c);
                a[c] = fma(-a[j], lcj, a[c]);    // entries above the diagonal (li < c) hold don't-care values, zeroed on store
            }
        }
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 16; ++c) A[(size_t)li * ld + c] = (c <= li) ? a[c] : (T)0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) rinv[j] = (T)1 / lane_bcast(a[j], j);
    }
    // inverse: lane c owns column c of X = L^-1;  x_i = (delta_ic - sum_{k<i} L_ik x_k) / L_ii
    T x[16];
    if (INV_FROM_LDS) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);      // the tile (L) written above has landed in LDS
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            T acc = (li == i) ? (T)1 : (T)0;
#pragma unroll
            for (int k = 0; k < i; ++k) acc = fma(-A[(size_t)i * ld + k], x[k], acc);   // wave-uniform address: broadcast
            x[i] = acc * rinv[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            T acc = (li == i) ? (T)1 : (T)0;
#pragma unroll
            for (int k = 0; k < i; ++k) acc -= lane_bcast(a[k], i) * x[k];
            x[i] = acc * rinv[i];
        }
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
            diag_tile<T, true, false>(A + (size_t)(16 * k) * ld + 16 * k, ld, dinv, dinv_glob ? dinv_glob + k * 256 : nullptr,
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
// LDS-resident blocked Cholesky: lower-triangle tiles (I,J), J <= I < nbf, followed by (nbr - nbf) border VECTORS: one row of
// nbf * LDT elements each (border vector b, tile column J at b[J * LDT .. J * LDT + 15]).  A border vector rides through the
// factorisation as row 0 of an otherwise zero tile row (on exit it holds L^-1 v), but only that row is stored: with a full
// tile row the 128 x 128 fp64 case needs 90 KB of LDS and only one workgroup fits a compute unit; this way it is 80 KB.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_tile_index(int I, int J, int nbf) { return I * (I + 1) / 2 + J; }     // J <= I < nbf
static inline size_t lds_chol_elems(int nbf, int nborder) {      // dinv tile + lower tiles + border vectors
    return (size_t)TSZ * (size_t)(1 + nbf * (nbf + 1) / 2) + (size_t)nborder * nbf * LDT;
}

// tiles: LDS array of TSZ-element tiles followed by the border vectors; dinv: one more LDS tile.  On exit the tiles hold L
// and the border vectors L^-1 v.  dinv_glob (optional): [nbf][256] global array that receives the inverted diagonal tiles.
// OCC only separates instantiations: a kernel bounded to 2 workgroups per CU (256 VGPRs) must not share this function's
// register allocation with an unbounded one (the fp64 diagonal tile wants ~330 registers and spills when held to 256).
template <typename T, int OCC = 1>
__device__ void potrf_lds(T *tiles, T *dinv, int nbf, int nbr, int *fail, T *dinv_glob = nullptr) {
    typedef typename Mfma<T>::acc_t acc_t;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, kk = lane >> 4;
    T *border = tiles + (size_t)(nbf * (nbf + 1) / 2) * TSZ;
    if (wv == 0) diag_tile<T, true, true>(tiles, LDT, dinv, dinv_glob, fail, 0);
    __syncthreads();
    for (int k = 0; k < nbf; ++k) {
        // panel: P_I = A_Ik * Linv_kk^T, in place
        { ACC_BEGIN();
        for (int I = k + 1 + wv; I < nbr; I += 4) {
            acc_t c = {0, 0, 0, 0};
            if (I < nbf) {
                T *tile = tiles + lds_tile_index(I, k, nbf) * TSZ;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) c = Mfma<T>::mma(tile[li * LDT + 4 * ks + kk], dinv[li * LDT + 4 * ks + kk], c);
#pragma unroll
                for (int v = 0; v < 4; ++v) tile[Mfma<T>::row(lane, v) * LDT + li] = c[v];
            } else {
                T *bk = border + ((I - nbf) * nbf + k) * LDT;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    c = Mfma<T>::mma(li == 0 ? bk[4 * ks + kk] : (T)0, dinv[li * LDT + 4 * ks + kk], c);
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (Mfma<T>::row(lane, v) == 0) bk[li] = c[v];
            }
        }
        ACC_END(5); }
        __syncthreads();
        // trailing update A_IJ -= P_I P_J^T (k < J < nbf, J <= I < nbr).  Wave 0 takes the next diagonal tile first and
        // factors it right away; waves 1-3 share the rest of the update.
        if (wv == 0) {
            if (k + 1 < nbf) {
                T *tile = tiles + lds_tile_index(k + 1, k + 1, nbf) * TSZ;
                const T *pI = tiles + lds_tile_index(k + 1, k, nbf) * TSZ;
                acc_t c;
#pragma unroll
                for (int v = 0; v < 4; ++v) c[v] = tile[Mfma<T>::row(lane, v) * LDT + li];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) c = Mfma<T>::mma(-pI[li * LDT + 4 * ks + kk], pI[li * LDT + 4 * ks + kk], c);
#pragma unroll
                for (int v = 0; v < 4; ++v) tile[Mfma<T>::row(lane, v) * LDT + li] = c[v];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                ACC_BEGIN();
                diag_tile<T, true, true>(tile, LDT, dinv, dinv_glob ? dinv_glob + (k + 1) * 256 : (T *)nullptr, fail,
                                         16 * (k + 1));
                ACC_END(4);
            }
        } else {
            ACC_BEGIN();
            int cnt = 0;
            for (int I = k + 1; I < nbr; ++I) {
                const int jmax = min(I, nbf - 1);
                for (int J = k + 1; J <= jmax; ++J) {
                    if (I == k + 1 && J == k + 1) continue;
                    if ((cnt++ % 3) != wv - 1) continue;
                    const T *pJ = tiles + lds_tile_index(J, k, nbf) * TSZ;
                    acc_t c;
                    if (I < nbf) {
                        T *tile = tiles + lds_tile_index(I, J, nbf) * TSZ;
                        const T *pI = tiles + lds_tile_index(I, k, nbf) * TSZ;
#pragma unroll
                        for (int v = 0; v < 4; ++v) c[v] = tile[Mfma<T>::row(lane, v) * LDT + li];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks)
                            c = Mfma<T>::mma(-pI[li * LDT + 4 * ks + kk], pJ[li * LDT + 4 * ks + kk], c);
#pragma unroll
                        for (int v = 0; v < 4; ++v) tile[Mfma<T>::row(lane, v) * LDT + li] = c[v];
                    } else {
                        T *bJ = border + ((I - nbf) * nbf + J) * LDT;
                        const T *bk = border + ((I - nbf) * nbf + k) * LDT;
#pragma unroll
                        for (int v = 0; v < 4; ++v) c[v] = (Mfma<T>::row(lane, v) == 0) ? bJ[li] : (T)0;
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks)
                            c = Mfma<T>::mma(li == 0 ? -bk[4 * ks + kk] : (T)0, pJ[li * LDT + 4 * ks + kk], c);
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            if (Mfma<T>::row(lane, v) == 0) bJ[li] = c[v];
                    }
                }
            }
            ACC_END(6);
        }
        __syncthreads();
    }
}

// In-place inverse of the LDS-resident Cholesky factor: tiles (lower, nb x nb) hold L on entry and the lower triangle of
// (L L^T)^-1 = L^-T L^-1 on exit.  dinv_glob: the inverted diagonal tiles saved by potrf_lds; dinv: one LDS tile.
//   1. W = L^-1 row by row (rows above already hold W):  L'_ik = Linv_ii L_ik (k < i);  W_ij = delta_ij Linv_ii -
//      sum_{k=j}^{i-1} L'_ik W_kj, collected in registers and written once the whole row has been read;
//   2. out_ij = sum_{k >= i} W_ki^T W_kj row by row (row i of W is not read again by later rows).
// Every product is the tile primitive C += X Y^T of Mfma<T>::mma on 16x16 LDS tiles (Y read transposed where needed).
template <typename T>
__device__ void trtri_lds(T *tiles, T *dinv, const T *dinv_glob, int nb) {       // step 1 alone: tiles <- L^-1
    typedef typename Mfma<T>::acc_t acc_t;
    constexpr int MAXT = 3;                                   // tiles of one row per wave: nb <= 12
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kk = lane >> 4;
    for (int i = 0; i < nb; ++i) {
        dinv[(t >> 4) * LDT + (t & 15)] = dinv_glob[i * 256 + t];
        __syncthreads();
        for (int k = wv; k < i; k += 4) {                     // L'_ik = Linv_ii L_ik, in place
            T *tile = tiles + lds_tile_index(i, k, nb) * TSZ;
            acc_t c = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) c = Mfma<T>::mma(dinv[li * LDT + 4 * ks + kk], tile[(4 * ks + kk) * LDT + li], c);
#pragma unroll
            for (int v = 0; v < 4; ++v) tile[Mfma<T>::row(lane, v) * LDT + li] = c[v];
        }
        __syncthreads();
        acc_t cw[MAXT];
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
            const int j = wv + 4 * u;
            if (j > i) continue;
            acc_t c = {0, 0, 0, 0};
            if (j == i) {
#pragma unroll
                for (int v = 0; v < 4; ++v) c[v] = dinv[Mfma<T>::row(lane, v) * LDT + li];
            } else {
                for (int k = j; k < i; ++k) {
                    const T *lt = tiles + lds_tile_index(i, k, nb) * TSZ, *wt = tiles + lds_tile_index(k, j, nb) * TSZ;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        c = Mfma<T>::mma(-lt[li * LDT + 4 * ks + kk], wt[(4 * ks + kk) * LDT + li], c);
                }
            }
            cw[u] = c;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
            const int j = wv + 4 * u;
            if (j > i) continue;
            T *tile = tiles + lds_tile_index(i, j, nb) * TSZ;
#pragma unroll
            for (int v = 0; v < 4; ++v) tile[Mfma<T>::row(lane, v) * LDT + li] = cw[u][v];
        }
        __syncthreads();
    }
}
template <typename T>
__device__ void potri_lds(T *tiles, T *dinv, const T *dinv_glob, int nb) {
    typedef typename Mfma<T>::acc_t acc_t;
    constexpr int MAXT = 3;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kk = lane >> 4;
    trtri_lds<T>(tiles, dinv, dinv_glob, nb);
    // (the diagonal tiles of W are lower triangular with explicit zeros above the diagonal: diag_tile)
    for (int i = 0; i < nb; ++i) {
        acc_t cw[MAXT];
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
            const int j = wv + 4 * u;
            if (j > i) continue;
            acc_t c = {0, 0, 0, 0};
            for (int k = i; k < nb; ++k) {
                const T *wi = tiles + lds_tile_index(k, i, nb) * TSZ, *wj = tiles + lds_tile_index(k, j, nb) * TSZ;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    c = Mfma<T>::mma(wi[(4 * ks + kk) * LDT + li], wj[(4 * ks + kk) * LDT + li], c);
            }
            cw[u] = c;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < MAXT; ++u) {
            const int j = wv + 4 * u;
            if (j > i) continue;
            T *tile = tiles + lds_tile_index(i, j, nb) * TSZ;
#pragma unroll
            for (int v = 0; v < 4; ++v) tile[Mfma<T>::row(lane, v) * LDT + li] = cw[u][v];
        }
        __syncthreads();
    }
}

// lower tiles of Wm^T Wm for a lower-triangular Wm (nb x nb tiles, global): out_IJ = sum_{k >= I} W_kI^T W_kJ
template <typename T> __device__ void wtw_lower_blocked(const T *Wm, int ldw, T *out, int ldo, int nb) {
    typedef typename Mfma<T>::acc_t acc_t;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, kk = lane >> 4;
    int cnt = 0;
    for (int I = 0; I < nb; ++I)
        for (int J = 0; J <= I; ++J, ++cnt) {
            if ((cnt & 3) != wv) continue;
            acc_t c = {0, 0, 0, 0};
            for (int k = I; k < nb; ++k) {
                const T *wr = Wm + (size_t)(16 * k) * ldw;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    c = Mfma<T>::mma(wr[(size_t)(4 * ks + kk) * ldw + 16 * I + li], wr[(size_t)(4 * ks + kk) * ldw + 16 * J + li], c);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) out[(size_t)(16 * I + Mfma<T>::row(lane, v)) * ldo + 16 * J + li] = c[v];
        }
    __syncthreads();
}


// per-output workspace of the Cholesky chain in elements (layout: linalg.hip)
static inline size_t la_chain_ws_elems_inline(int M) {
    const int Mp = dpgp_round_up(M, 16);
    return (size_t)3 * Mp * Mp + (size_t)(Mp + 16) * Mp + (size_t)(Mp / 16) * 256;
}
static inline size_t la_lds_bytes(int Mp, size_t elem) {     // global-memory blocked routines: dinv + one panel of nb+1 tiles
    return LA_LDS_HDR + elem * (size_t)TSZ * (size_t)(Mp / 16 + 2);
}
// chain_k keeps K_uu in LDS (dinv + the lower-triangle tiles) when that leaves room for two workgroups per CU next to
// the psi2 workgroups of the same dispatch (80 KB each) and a row of tiles fits the per-wave register arrays of potri_lds
#define LA_CHAIN_K_LDS_MAX (80 * 1024)
__host__ __device__ inline size_t chain_k_resident_bytes(int Mp, size_t elem) {
    const int nb = Mp / 16;
    return LA_LDS_HDR + elem * (size_t)TSZ * (size_t)(1 + nb * (nb + 1) / 2);
}
__host__ __device__ inline bool chain_k_resident(int Mp, size_t elem) {
    return Mp / 16 <= 12 && chain_k_resident_bytes(Mp, elem) <= LA_CHAIN_K_LDS_MAX;
}
static inline size_t chain_k_lds_bytes(int Mp, size_t elem) {
    return chain_k_resident(Mp, elem) ? chain_k_resident_bytes(Mp, elem) : la_lds_bytes(Mp, elem);
}

// ---- chain_k: everything that depends on K_uu only (dp_gp_lvm.py:115-116) for output dim d; one 256-thread workgroup.
// smem_raw: >= la_lds_bytes(Mp) bytes of (dynamic) LDS.  Called from chain_k_kernel and, as an extra task slice, from the
// psi2 kernels (so that it is dispatched together with — in front of — the psi2 workgroups and overlaps them).
template <typename TL, int OCC>      // OCC: see potrf_lds
__device__ void chain_k_body(int d, int M, int Mp, TL *__restrict__ ws, size_t ws_stride, double *__restrict__ logdet_k,
                             int *__restrict__ info_k, int plain, unsigned char *smem_raw) {
    double *scratch = reinterpret_cast<double *>(smem_raw);
    int &fail = *reinterpret_cast<int *>(smem_raw + 64);
    TL *lds = reinterpret_cast<TL *>(smem_raw + LA_LDS_HDR);
    const int t = threadIdx.x, nb = Mp / 16;
    TL *K0 = ws + (size_t)d * ws_stride, *Kb = K0 + (size_t)Mp * Mp, *Wb = Kb + (size_t)Mp * Mp,
       *KI = Wb + (size_t)(Mp + 16) * Mp, *dinv = KI + (size_t)Mp * Mp;
    if (t == 0) fail = 0;
    if (!plain && chain_k_resident(Mp, sizeof(TL))) {
        // ---- LDS-resident: K_uu -> LDS, Cholesky, log-det, in-place inverse, K_uu^-1 -> KI; no other global traffic ----
        typedef TL tl4 __attribute__((ext_vector_type(4)));
        TL *dl = lds, *tiles = lds + TSZ;
        const int nlow = nb * (nb + 1) / 2;
        const int u = t >> 6, r = (t & 63) >> 2, c4 = (t & 3) * 4;
        for (int t0 = 0; t0 < nlow; t0 += 4) {                // 4 tiles per pass: thread = (tile, row, 4 columns)
            const int tt = t0 + u;
            if (tt >= nlow) continue;
            int I = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
            while ((I + 1) * (I + 2) / 2 <= tt) ++I;
            while (I * (I + 1) / 2 > tt) --I;
            const int J = tt - I * (I + 1) / 2, i = 16 * I + r, j = 16 * J + c4;
            const tl4 k0 = *reinterpret_cast<const tl4 *>(K0 + (size_t)i * Mp + j);
            TL *dst = tiles + lds_tile_index(I, J, nb) * TSZ + r * LDT + c4;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] = (i < M && j + e < M) ? k0[e] : ((i == j + e) ? (TL)1 : (TL)0);
        }
        __syncthreads();
        potrf_lds<TL, OCC>(tiles, dl, nb, nb, &fail, dinv);
        __syncthreads();
        double ld = 0.0;
        for (int i = t; i < M; i += 256) ld += log((double)tiles[lds_tile_index(i >> 4, i >> 4, nb) * TSZ + (i & 15) * LDT + (i & 15)]);
        ld = block_sum(ld, scratch);
        if (t == 0) {
            logdet_k[d] = ld;
            info_k[d] = fail;
        }
        __threadfence_block();
        __syncthreads();                                       // dinv (global) written by wave 0 is read by all waves below
        potri_lds<TL>(tiles, dl, dinv, nb);
        for (int t0 = 0; t0 < nlow; t0 += 4) {
            const int tt = t0 + u;
            if (tt >= nlow) continue;
            int I = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
            while ((I + 1) * (I + 2) / 2 <= tt) ++I;
            while (I * (I + 1) / 2 > tt) --I;
            const int J = tt - I * (I + 1) / 2, i = 16 * I + r, j = 16 * J + c4;
            const TL *src = tiles + lds_tile_index(I, J, nb) * TSZ + r * LDT + c4;
            tl4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = src[e];
            *reinterpret_cast<tl4 *>(KI + (size_t)i * Mp + j) = v;
        }
        return;
    }
    // the copy that rides in the 2-per-CU psi2 kernel is only ever launched for the LDS-resident case (the host checks
    // chain_k_resident): keeping the global-memory routines out of it keeps their registers out of that kernel's budget
    if constexpr (OCC == 2) return;
    for (int e = t; e < Mp * Mp; e += 256) {
        const int i = e / Mp, j = e - i * Mp;
        Kb[e] = (i < M && j < M) ? K0[e] : ((i == j) ? (TL)1 : (TL)0);
        Wb[e] = (i == j) ? (TL)1 : (TL)0;
    }
    __syncthreads();
    if (plain) potrf_plain<TL>(Kb, Mp, Mp, Mp, &fail, 0);
    else potrf_blocked<TL>(Kb, Mp, nb, nb, lds, dinv, &fail, 0);
    __syncthreads();
    double ld = 0.0;
    for (int i = t; i < M; i += 256) ld += log((double)Kb[(size_t)i * Mp + i]);
    ld = block_sum(ld, scratch);
    if (t == 0) {
        logdet_k[d] = ld;
        info_k[d] = fail;
    }
    // W = L^-1, K^-1 = W^T W (lower)
    if (plain) {
        trsm_left_plain<TL>(Kb, Mp, Wb, Mp, Mp, Mp);
        for (int e = t; e < Mp * Mp; e += 256) {
            const int i = e / Mp, j = e - i * Mp;
            if (j > i) continue;
            TL a = 0;
            for (int k = i; k < Mp; ++k) a += Wb[(size_t)k * Mp + i] * Wb[(size_t)k * Mp + j];
            KI[e] = a;
        }
    } else {
        trsm_left_blocked<TL>(Kb, Mp, dinv, Wb, Mp, nb, nb, lds);
        wtw_lower_blocked<TL>(Wb, Mp, KI, Mp, nb);
    }
}

