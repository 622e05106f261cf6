// K4/K5/K6  batched blocked Cholesky, triangular solves and the per-output ELBO reduction on the matrix cores.
// Reference ops replaced: tf.cholesky / tf.matrix_triangular_solve / trace / log-det at
// /root/reference/src/models/dp_gp_lvm.py:115-145.
//
// One 256-thread workgroup (4 waves) owns one matrix.  Everything is tiled in 16x16 blocks (v_mfma_*_16x16x4):
//   potrf  : right-looking.  Step k: wave 0 factors the diagonal tile in registers (one row per lane, columns via
//            v_readlane broadcasts) and also inverts it; the panel below becomes a GEMM with that inverse
//            (P_I = A_Ik Linv_kk^T, MFMA); the trailing lower triangle gets the SYRK/GEMM update A_IJ -= P_I P_J^T (MFMA).
//            "Border" tile-rows below the SPD part are carried along (panel + trailing steps only): a border row
//            holding v^T comes out as (L^-1 v)^T — the M-vector solve of the data-fit term costs nothing extra.
//            Two variants: matrix in global memory (any M), and LDS-resident lower triangle (M <= ~160 in fp64), where
//            wave 0 factors diagonal tile k+1 while waves 1-3 finish the trailing update of step k.
//   trsm   : with the inverted diagonal tiles a triangular solve is a pure MFMA GEMM sweep.
//
// The per-output chain of dp_gp_lvm.py:115-145 is evaluated in the algebraically identical form
//      B = K_uu + beta Psi2 = L (beta L^-1 Psi2 L^-T + I) L^T = L A L^T
//      sum log diag L_A = sum log diag L_B - sum log diag L          tr(L^-1 Psi2 L^-T) = <K_uu^-1, Psi2>_F
//      |L_A^-1 L^-1 v|^2 = v^T B^-1 v = |L_B^-1 v|^2
//  so that everything that depends only on K_uu (its Cholesky, log-det and inverse: chain_k_kernel) can run on a second
//  stream WHILE the psi2 kernel runs, and the part after Psi2 (chain_b_kernel) is one fused assemble + one bordered
//  Cholesky.  Matrices are padded to a multiple of 16 with an identity block so no tile needs bounds checks.
#include "internal.h"

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

#ifdef DPGP_PROFILE_CHAIN
// diagnostic build only (scratch/): per-phase clock stamps of workgroup 0, read back with dpgp_debug_stamps()
__device__ long long g_chain_stamps[16];
#define STAMP(i) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) g_chain_stamps[i] = wall_clock64(); } while (0)
extern "C" void dpgp_debug_stamps(long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_stamps), sizeof(long long) * 16); }
#define ACC_BEGIN() long long t__ = wall_clock64()
#define ACC_END(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == ((i) == 4 ? 0 : 1)) g_chain_stamps[i] += wall_clock64() - t__; } while (0)
#else
#define STAMP(i)
#define ACC_BEGIN()
#define ACC_END(i)
#endif

// ---------------------------------------------------------------------------------------------------------------
// LDS-resident blocked Cholesky: lower-triangle tiles (I,J), J <= I < nbf, then (nbr - nbf) border tile-rows of nbf tiles.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_tile_index(int I, int J, int nbf) {
    return I < nbf ? I * (I + 1) / 2 + J : nbf * (nbf + 1) / 2 + (I - nbf) * nbf + J;
}
static inline int lds_tile_count(int nbf, int nbr) { return nbf * (nbf + 1) / 2 + (nbr - nbf) * nbf; }

// tiles: LDS array of TSZ-element tiles; dinv: one more LDS tile.  On exit the tiles hold L (and the solved border rows).
template <typename T>
__device__ void potrf_lds(T *tiles, T *dinv, int nbf, int nbr, int *fail) {
    typedef typename Mfma<T>::acc_t acc_t;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, kk = lane >> 4;
    if (wv == 0) diag_tile<T, true, true>(tiles, LDT, dinv, (T *)nullptr, fail, 0);
    __syncthreads();
    for (int k = 0; k < nbf; ++k) {
        // panel: P_I = A_Ik * Linv_kk^T, in place
        { ACC_BEGIN();
        for (int I = k + 1 + wv; I < nbr; I += 4) {
            T *tile = tiles + lds_tile_index(I, k, nbf) * TSZ;
            acc_t c = {0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) c = Mfma<T>::mma(tile[li * LDT + 4 * ks + kk], dinv[li * LDT + 4 * ks + kk], c);
#pragma unroll
            for (int v = 0; v < 4; ++v) tile[Mfma<T>::row(lane, v) * LDT + li] = c[v];
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
                diag_tile<T, true, true>(tile, LDT, dinv, (T *)nullptr, fail, 16 * (k + 1));
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
                    T *tile = tiles + lds_tile_index(I, J, nbf) * TSZ;
                    const T *pI = tiles + lds_tile_index(I, k, nbf) * TSZ, *pJ = tiles + lds_tile_index(J, k, nbf) * TSZ;
                    acc_t c;
#pragma unroll
                    for (int v = 0; v < 4; ++v) c[v] = tile[Mfma<T>::row(lane, v) * LDT + li];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        c = Mfma<T>::mma(-pI[li * LDT + 4 * ks + kk], pJ[li * LDT + 4 * ks + kk], c);
#pragma unroll
                    for (int v = 0; v < 4; ++v) tile[Mfma<T>::row(lane, v) * LDT + li] = c[v];
                }
            }
            ACC_END(6);
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

// ---------------------------------------------------------------------------------------------------------------
// Per-output workspace (elements of TL), see la_chain_ws_elems:
//   K0 [Mp x Mp]      : K_uu + jitter I as written by the gram kernel into [0,M)x[0,M)   (read by both chain kernels)
//   Kb [Mp x Mp]      : identity-padded copy -> L_uu                                       (chain_k)
//   Wb [(Mp+16) x Mp] : L_uu^-1 (chain_k); reused as B = K + beta Psi2 (+ border row) when B does not fit in LDS (chain_b)
//   KI [Mp x Mp]      : K_uu^-1, lower triangle                                            (chain_k -> chain_b)
//   dinv [Mp/16][256] : inverted diagonal tiles of L_uu
// ---------------------------------------------------------------------------------------------------------------
size_t la_chain_ws_elems(int M) {
    const int Mp = dpgp_round_up(M, 16);
    return (size_t)3 * Mp * Mp + (size_t)(Mp + 16) * Mp + (size_t)(Mp / 16) * 256;
}

// ---- chain_k: everything that depends on K_uu only (dp_gp_lvm.py:115-116) -----------------------------------------
template <typename TL>
__global__ __launch_bounds__(256) void chain_k_kernel(int M, int Mp, TL *__restrict__ ws, size_t ws_stride,
                                                      double *__restrict__ logdet_k, int *__restrict__ info_k,
                                                      int plain) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double *scratch = reinterpret_cast<double *>(smem_raw);
    int &fail = *reinterpret_cast<int *>(smem_raw + 64);
    TL *lds = reinterpret_cast<TL *>(smem_raw + LA_LDS_HDR);
    const int d = blockIdx.x, t = threadIdx.x, nb = Mp / 16;
    TL *K0 = ws + (size_t)d * ws_stride, *Kb = K0 + (size_t)Mp * Mp, *Wb = Kb + (size_t)Mp * Mp,
       *KI = Wb + (size_t)(Mp + 16) * Mp, *dinv = KI + (size_t)Mp * Mp;
    if (t == 0) fail = 0;
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

// ---- chain_b: everything after Psi2 (dp_gp_lvm.py:118-145 in the B = K + beta Psi2 form) -----------------------------
// mode 0: B lives in LDS (potrf_lds); mode 1: B in global memory (Wb), blocked MFMA; mode 2: plain VALU cross-check.
template <typename TP, typename TL>
__global__ __launch_bounds__(256) void chain_b_kernel(int D, int N, int M, int Mp, const TP *__restrict__ psi2_part,
                                                      int ns2, const double *__restrict__ v_part, int ns1,
                                                      const double *__restrict__ alpha,
                                                      const double *__restrict__ beta,
                                                      const double *__restrict__ yy_part,
                                                      const double *__restrict__ logdet_k,
                                                      const int *__restrict__ info_k, double *__restrict__ terms,
                                                      int *__restrict__ info, TL *__restrict__ ws, size_t ws_stride,
                                                      int mode) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double *scratch = reinterpret_cast<double *>(smem_raw);
    int &fail = *reinterpret_cast<int *>(smem_raw + 64);
    TL *dinv = reinterpret_cast<TL *>(smem_raw + LA_LDS_HDR);
    TL *tiles = dinv + TSZ;
    const int d = blockIdx.x, t = threadIdx.x, nb = Mp / 16;
    const TL *K0 = ws + (size_t)d * ws_stride;
    TL *Wb = ws + (size_t)d * ws_stride + (size_t)2 * Mp * Mp;
    const TL *KI = Wb + (size_t)(Mp + 16) * Mp;
    if (t == 0) fail = 0;
    STAMP(0);
    // ---- assemble B = K + beta Psi2 (lower) with the border row v^T; <K^-1, Psi2>_F on the fly ----
    const TL be = (TL)beta[d];
    double ip = 0.0;
    const int ii = t >> 4, jj = t & 15;
    const int nlow = nb * (nb + 1) / 2;
    {
        // 4 tiles per pass: thread = (tile u, row r, 4 consecutive columns) -> 16/32-byte loads, all issued before use
        typedef TP tp4 __attribute__((ext_vector_type(4)));
        typedef TL tl4 __attribute__((ext_vector_type(4)));
        const int u = t >> 6, r = (t & 63) >> 2, c4 = (t & 3) * 4;
#pragma unroll 2
        for (int t0 = 0; t0 < nlow; t0 += 4) {
            const int tt = min(t0 + u, nlow - 1);
            int I = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
            while ((I + 1) * (I + 2) / 2 <= tt) ++I;
            while (I * (I + 1) / 2 > tt) --I;
            const int J = tt - I * (I + 1) / 2;
            const int i = 16 * I + r, j = 16 * J + c4;
            const size_t off = (size_t)i * Mp + j;
            tl4 k0 = *reinterpret_cast<const tl4 *>(K0 + off);
            tl4 ki = *reinterpret_cast<const tl4 *>(KI + off);
            double p2[4] = {0.0, 0.0, 0.0, 0.0};
            for (int kb = 0; kb < ns2; kb += 8) {      // up to 8 slab loads in flight (psi2_nsplit() never exceeds 8)
                tp4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (kb + k < ns2)
                        v[k] = *reinterpret_cast<const tp4 *>(psi2_part + ((size_t)(kb + k) * D + d) * (size_t)Mp * Mp + off);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (kb + k < ns2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) p2[e] += (double)v[k][e];
                    }
            }
            if (t0 + u < nlow) {
                TL *dst = (mode == 0) ? tiles + lds_tile_index(I, J, nb) * TSZ + r * LDT + c4 : Wb + off;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int je = j + e;
                    TL bv;
                    if (je <= i && i < M) {            // inside the real lower triangle
                        bv = k0[e] + be * (TL)p2[e];
                        ip += (double)ki[e] * p2[e] * (i == je ? 1.0 : 2.0);
                    } else {
                        bv = (i == je) ? (TL)1 : (TL)0;   // identity padding (and don't-care zeros above the diagonal)
                    }
                    dst[e] = bv;
                }
            }
        }
    }
    for (int J = 0; J < nb; ++J) {                          // border tile-row: row 0 holds v^T
        const int j = 16 * J + jj;
        TL bv = 0;
        if (ii == 0 && j < M) {
            double a = 0.0;
            for (int k = 0; k < ns1; ++k) a += v_part[((size_t)k * D + d) * M + j];
            bv = (TL)a;
        }
        if (mode == 0) tiles[lds_tile_index(nb, J, nb) * TSZ + ii * LDT + jj] = bv;
        else Wb[(size_t)(Mp + ii) * Mp + j] = bv;
    }
    ip = block_sum(ip, scratch);
    __syncthreads();
    STAMP(1);
    // ---- L_B = chol(B), border -> L_B^-1 v ----
    if (mode == 0) potrf_lds<TL>(tiles, dinv, nb, nb + 1, &fail);
    else if (mode == 1) potrf_blocked<TL>(Wb, Mp, nb, nb + 1, dinv, (TL *)nullptr, &fail, 0);
    else potrf_plain<TL>(Wb, Mp, Mp, Mp + 1, &fail, 0);
    __syncthreads();
    STAMP(2);
    double ld = 0.0, cc = 0.0;
    for (int i = t; i < M; i += 256) {
        const int I = i >> 4, r = i & 15;
        const double lii = mode == 0 ? (double)tiles[lds_tile_index(I, I, nb) * TSZ + r * LDT + r]
                                     : (double)Wb[(size_t)i * Mp + i];
        const double c = mode == 0 ? (double)tiles[lds_tile_index(nb, I, nb) * TSZ + r] : (double)Wb[(size_t)Mp * Mp + i];
        ld += log(lii);
        cc += c * c;
    }
    ld = block_sum(ld, scratch);
    cc = block_sum(cc, scratch);
    double yy = 0.0;
    for (int k = t; k < DPGP_YY_NCH; k += 256) yy += yy_part[(size_t)k * D + d];
    yy = block_sum(yy, scratch);
    STAMP(3);
    if (t == 0) {
        const double b_ = beta[d], a_ = alpha[d];
        double *o = terms + (size_t)d * 5;
        const int fk = info_k[d];
        const int f = fk ? fk : (fail ? M + fail : 0);
        info[d] = f;
        const double nan_ = __longlong_as_double(0x7ff8000000000000LL);
        o[0] = 0.5 * N * (log(b_) - DPGP_LOG_2PI);
        o[1] = f ? nan_ : -(ld - logdet_k[d]);                 // -sum log diag L_A
        o[2] = f ? nan_ : 0.5 * b_ * (ip - a_ * N);            // tr(L^-1 Psi2 L^-T) = <K^-1, Psi2>
        o[3] = -0.5 * b_ * yy;
        o[4] = f ? nan_ : 0.5 * b_ * b_ * cc;
    }
}

static size_t la_lds_bytes(int Mp, size_t elem) {     // global-memory blocked routines: dinv + one panel of nb+1 tiles
    return LA_LDS_HDR + elem * (size_t)TSZ * (size_t)(Mp / 16 + 2);
}
static size_t chain_b_lds_bytes(int Mp, size_t elem) {   // LDS-resident B: dinv + lower triangle + border row
    const int nb = Mp / 16;
    return LA_LDS_HDR + elem * (size_t)TSZ * (size_t)(1 + lds_tile_count(nb, nb + 1));
}

template <typename TL>
int launch_chain_k(int D, int M, TL *ws, double *logdet_k, int *info_k, int algo, hipStream_t st) {
    const int Mp = dpgp_round_up(M, 16);
    size_t lds = la_lds_bytes(Mp, sizeof(TL));
    auto kern = chain_k_kernel<TL>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3(D), dim3(256), lds, st, M, Mp, ws, la_chain_ws_elems(M), logdet_k, info_k,
                       algo == DPGP_ALGO_PLAIN ? 1 : 0);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
template int launch_chain_k<float>(int, int, float *, double *, int *, int, hipStream_t);
template int launch_chain_k<double>(int, int, double *, double *, int *, int, hipStream_t);

template <typename TP, typename TL>
int launch_chain_b(int D, int N, int M, const TP *psi2_part, int ns2, const double *v_part, int ns1,
                   const double *alpha, const double *beta, const double *yy_part, const double *logdet_k,
                   const int *info_k, double *terms, int *info, TL *ws, int algo, hipStream_t st) {
    const int Mp = dpgp_round_up(M, 16);
    int mode = 2;
    size_t lds = la_lds_bytes(Mp, sizeof(TL));
    if (algo != DPGP_ALGO_PLAIN) {
        const size_t need = chain_b_lds_bytes(Mp, sizeof(TL));
        mode = need <= LA_LDS_LIMIT ? 0 : 1;
        if (mode == 0) lds = need;
    }
    auto kern = chain_b_kernel<TP, TL>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3(D), dim3(256), lds, st, D, N, M, Mp, psi2_part, ns2, v_part, ns1, alpha, beta, yy_part,
                       logdet_k, info_k, terms, info, ws, la_chain_ws_elems(M), mode);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
#define INST_CHAIN_B(TP, TL)                                                                                        \
    template int launch_chain_b<TP, TL>(int, int, int, const TP *, int, const double *, int, const double *,      \
                                        const double *, const double *, const double *, const int *, double *, int *, \
                                        TL *, int, hipStream_t);
INST_CHAIN_B(float, float)
INST_CHAIN_B(float, double)
INST_CHAIN_B(double, double)

__global__ __launch_bounds__(256) void sum_terms_kernel(int D, const double *__restrict__ terms,
                                                        const double *__restrict__ kl_part,
                                                        double *__restrict__ sums,
                                                        const double *__restrict__ model_scal,
                                                        double *__restrict__ model_pack,
                                                        double *__restrict__ model_out) {
    __shared__ double scratch[8];
    double a = 0.0;
    for (int i = threadIdx.x; i < D * 5; i += 256) a += terms[i];
    a = block_sum(a, scratch);
    if (threadIdx.x == 0) {
        sums[0] = a;
        double kl = sums[1];
        if (kl_part) {
            kl = 0.0;
            for (int i = 0; i < DPGP_KL_NBLK; ++i) kl += kl_part[i];
            sums[1] = kl;
        }
        if (model_scal) {    // dp_gp_lvm.py:151-154 (see dpgp_model_pack / dpgp_model_finalize)
            double dp = model_scal[0];
            const int nrb = (D + DPGP_PREP_ROWS - 1) / DPGP_PREP_ROWS;
            for (int i = 0; i < nrb; ++i) dp += model_scal[2 + i];
            dp = -dp;
            if (model_pack) {
                model_pack[0] = a;
                model_pack[1] = dp;
            }
            if (model_out) {
                const double hyper = model_scal[1];
                model_out[0] = dp - (a - kl) - hyper;
                model_out[1] = a;
                model_out[2] = kl;
                model_out[3] = dp;
                model_out[4] = hyper;
            }
        }
    }
}
int launch_sum_terms(int D, const double *terms, const double *kl_part, double *sums, const double *model_scal,
                     double *model_pack, double *model_out, hipStream_t st) {
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(sum_terms_kernel, dim3(1), dim3(256), 0, st, D, terms, kl_part, sums, model_scal, model_pack,
                       model_out);
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
            diag_tile<T, false, false>(Lw + (size_t)(16 * k) * Mp + 16 * k, Mp, lds + (size_t)wv * 16 * LDT, dinv + k * 256,
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
    if (algo < 0 || algo > DPGP_ALGO_MFMA_F32) return -7;
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
    if (algo < 0 || algo > DPGP_ALGO_MFMA_F32) return -8;
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
