// Device-side linear algebra shared by linalg.hip and psi2.hip (16x16-tiled MFMA Cholesky / triangular solves).
#pragma once
#include "internal.h"

#ifndef STAMP          // diagnostic stamps are only compiled into linalg.hip of the scratch build (-DDPGP_PROFILE_CHAIN)
#define STAMP(i)
#define ACC_BEGIN()
#define ACC_END(i)
#endif
#ifndef SUB_BEGIN
#define SUB_BEGIN()
#define SUB_END(i)
#endif

#define LDT 17          // LDS tile row stride (16 + 1 pad)
#define TSZ (16 * LDT)  // elements per LDS tile
#define LA_LDS_HDR 128  // bytes at the start of the dynamic LDS region: 8 doubles of reduction scratch + fail flag
#define LA_LDS_LIMIT (150 * 1024)

// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic, not for its global loads and
// stores (__syncthreads() drains both; potrf_persist.hip lets 393 KB of zero stores drain behind the LDS-resident
// factorisation, which has eight of these per block column).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
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
            if (dinv_lds) dinv_lds[i * LDT + li] = x[i];
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

// ---- pivot reciprocal root to working precision, for the pivot chain: native seed + Newton steps written out so that the
// dependent chain is known (v_rsq_f64 is good to 5.2e-8 relative on gfx950, scratch/ubench/lat64.hip: two steps; v_rsq_f32 to
// one ulp: one step) ----
__device__ __forceinline__ double pivot_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    double e = __builtin_fma(-(h * y), y, 0.5);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-(h * y), y, 0.5);
    return __builtin_fma(y, e, y);
}
__device__ __forceinline__ float pivot_rsqrt(float d) {
    float y = __builtin_amdgcn_rsqf(d);
    const float e = __builtin_fmaf(-(0.5f * d * y), y, 0.5f);
    return __builtin_fmaf(y, e, y);
}

// ---- the register panel of potrf_lds: DPP row broadcasts (row = 16 lanes) ------------------------------------------------
// On gfx950 v_fmac_f64 with a row_newbcast source costs what a plain v_fma_f64 does (4.9 cycles of issue,
// scratch/ubench/lat64b.hip) against 17 + 5 for a v_readlane pair into SGPRs + fma: the broadcast is free.  Rules this code
// keeps: (1) a VALU write needs two wait states before a DPP read of the same register and the compiler's hazard recognizer
// does not see into asm statements — the one place where a producer stands directly in front of its DPP consumer carries its
// own s_nop; (2) the statements are volatile, so they stay in source order; (3) everything that belongs between two column
// updates sits INSIDE one asm statement: around every separate asm statement the compiler puts a conservative s_nop
// (125 per block column in the first version).
// One column c > J of pivot J for this lane's two rows (a: its row below the tile, g: its copy of row lane & 15 of the
// tile), preceded by operation N of the reciprocal chain of the NEXT pivot (PivotChain; N >= NOPS: none).
#define DPGP_FMAC2_F64                                                                                          \
    "v_fmac_f64_dpp %[a], -%[g], %[ta] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"                        \
    "v_fmac_f64_dpp %[g], -%[g], %[tg] row_newbcast:%[j] row_mask:0xf bank_mask:0xf"
#define DPGP_FMAC2_F32                                                                                          \
    "v_fmac_f32_dpp %[a], -%[g], %[ta] row_newbcast:%[j] row_mask:0xf bank_mask:0xf\n\t"                        \
    "v_fmac_f32_dpp %[g], -%[g], %[tg] row_newbcast:%[j] row_mask:0xf bank_mask:0xf"
// The reciprocal of the next pivot, two Newton steps from v_rcp_f64 (good to 4.6e-8 relative, scratch/ubench/lat64.hip; one
// step from v_rcp_f32), as single operations that ride in front of the column updates: the compiler does not move this
// dependent chain in between them by itself (16 x ~30 cycles of exposed latency per block column).
template <typename T> struct PivotChain {
    T d, y, e, rcp;
    static constexpr int NOPS = sizeof(T) == 8 ? 5 : 3;
    __device__ __forceinline__ void finish_from(int n0) {     // operations n0 .. NOPS - 1 on their own (short columns)
        if constexpr (sizeof(T) == 8) {
            if (n0 <= 0) y = __builtin_amdgcn_rcp((double)d);
            if (n0 <= 1) e = fma(-d, y, (T)1);
            if (n0 <= 2) y = fma(y, e, y);
            if (n0 <= 3) e = fma(-d, y, (T)1);
            if (n0 <= 4) rcp = fma(y, e, y);
        } else {
            if (n0 <= 0) y = __builtin_amdgcn_rcpf((float)d);
            if (n0 <= 1) e = fma(-d, y, (T)1);
            if (n0 <= 2) rcp = fma(y, e, y);
        }
    }
};
template <typename T, int J, int N> struct PanelColumn {
    static __device__ __forceinline__ void run(T &a, T &g, T ta, T tg, PivotChain<T> &nx) {
        if constexpr (sizeof(T) == 8) {
            if constexpr (N == 0)
                asm volatile("v_rcp_f64 %[y], %[d]\n\t" DPGP_FMAC2_F64 : [a] "+v"(a), [g] "+v"(g), [y] "=&v"(nx.y) : [ta] "v"(ta), [tg] "v"(tg), [d] "v"(nx.d), [j] "n"(J));
            else if constexpr (N == 1 || N == 3)
                asm volatile("v_fma_f64 %[e], -%[d], %[y], 1.0\n\t" DPGP_FMAC2_F64 : [a] "+v"(a), [g] "+v"(g), [e] "=&v"(nx.e) : [ta] "v"(ta), [tg] "v"(tg), [d] "v"(nx.d), [y] "v"(nx.y), [j] "n"(J));
            else if constexpr (N == 2)
                asm volatile("v_fma_f64 %[y], %[y], %[e], %[y]\n\t" DPGP_FMAC2_F64 : [a] "+v"(a), [g] "+v"(g), [y] "+v"(nx.y) : [ta] "v"(ta), [tg] "v"(tg), [e] "v"(nx.e), [j] "n"(J));
            else if constexpr (N == 4)
                asm volatile("v_fma_f64 %[r], %[y], %[e], %[y]\n\t" DPGP_FMAC2_F64 : [a] "+v"(a), [g] "+v"(g), [r] "=&v"(nx.rcp) : [ta] "v"(ta), [tg] "v"(tg), [y] "v"(nx.y), [e] "v"(nx.e), [j] "n"(J));
            else
                asm volatile(DPGP_FMAC2_F64 : [a] "+v"(a), [g] "+v"(g) : [ta] "v"(ta), [tg] "v"(tg), [j] "n"(J));
        } else {
            if constexpr (N == 0)
                asm volatile("v_rcp_f32 %[y], %[d]\n\t" DPGP_FMAC2_F32 : [a] "+v"(a), [g] "+v"(g), [y] "=&v"(nx.y) : [ta] "v"(ta), [tg] "v"(tg), [d] "v"(nx.d), [j] "n"(J));
            else if constexpr (N == 1)
                asm volatile("v_fma_f32 %[e], -%[d], %[y], 1.0\n\t" DPGP_FMAC2_F32 : [a] "+v"(a), [g] "+v"(g), [e] "=&v"(nx.e) : [ta] "v"(ta), [tg] "v"(tg), [d] "v"(nx.d), [y] "v"(nx.y), [j] "n"(J));
            else if constexpr (N == 2)
                asm volatile("v_fma_f32 %[r], %[y], %[e], %[y]\n\t" DPGP_FMAC2_F32 : [a] "+v"(a), [g] "+v"(g), [r] "=&v"(nx.rcp) : [ta] "v"(ta), [tg] "v"(tg), [y] "v"(nx.y), [e] "v"(nx.e), [j] "n"(J));
            else
                asm volatile(DPGP_FMAC2_F32 : [a] "+v"(a), [g] "+v"(g) : [ta] "v"(ta), [tg] "v"(tg), [j] "n"(J));
        }
    }
};
// column J + 1 of pivot J, then the next pivot d = (updated) g[J + 1] of lane J + 1, broadcast to the row of lanes
template <typename T, int J> struct PanelNextPivot {
    static __device__ __forceinline__ T run(T &a, T &g, T ta, T tg) {
        T d;
        if constexpr (sizeof(T) == 8)
            asm volatile(DPGP_FMAC2_F64 "\n\ts_nop 1\n\tv_mov_b64_dpp %[d], %[g] row_newbcast:%[j1] row_mask:0xf bank_mask:0xf"
                         : [a] "+v"(a), [g] "+v"(g), [d] "=&v"(d) : [ta] "v"(ta), [tg] "v"(tg), [j] "n"(J), [j1] "n"(J + 1));
        else
            asm volatile(DPGP_FMAC2_F32 "\n\ts_nop 1\n\tv_mov_b32_dpp %[d], %[g] row_newbcast:%[j1] row_mask:0xf bank_mask:0xf"
                         : [a] "+v"(a), [g] "+v"(g), [d] "=&v"(d) : [ta] "v"(ta), [tg] "v"(tg), [j] "n"(J), [j1] "n"(J + 1));
        return d;
    }
};
// lane J of the row of lanes -> all its lanes (NOP: wait states in front, for a source a VALU instruction has just written)
template <typename T, int J, int NOP> __device__ __forceinline__ T row_bcast(T v) {
    T r;
    if constexpr (sizeof(T) == 8) {
        if constexpr (NOP) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
        else asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
    } else {
        if constexpr (NOP) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
        else asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
    }
    return r;
}

// Pivots J .. 15 of one block column held one matrix row per lane, in the square-root-free (L D L^T) form: g = this lane's
// copy of row (lane & 15) of the diagonal tile (full symmetric row; every row of 16 lanes holds the whole tile), a = the
// lane's own row below the tile.  pc: the finished chain of pivot J.  dsel collects pivot (lane & 15) on each lane.  On exit
// g[j] / a[j] hold column j of the Schur complement at pivot j: the entries of L are those times rsqrt(d_j) (the caller's).
// Which chain operation rides in front of the q-th of the 14 - J columns behind the next pivot's: every other column while
// there is room (an asm statement that reads what the statement directly before it wrote gets a conservative s_nop from the
// compiler), every column otherwise; 99 = none.
template <typename T> constexpr int panel_chain_op(int J, int q) {
    const int ncols = 14 - J, nops = PivotChain<T>::NOPS;
    if (ncols >= 2 * nops - 1) return (q % 2 == 0 && q / 2 < nops) ? q / 2 : 99;
    return q < nops ? q : 99;
}
template <typename T> constexpr int panel_chain_placed(int J) {
    const int ncols = 14 - J, nops = PivotChain<T>::NOPS;
    return ncols >= 2 * nops - 1 ? nops : (ncols < nops ? (ncols < 0 ? 0 : ncols) : nops);
}
template <typename T, int J, int C> struct PanelColumns {       // columns C .. 15 of pivot J
    static __device__ __forceinline__ void run(T (&g)[16], T (&a)[16], T ta, T tg, PivotChain<T> &nx) {
        if constexpr (C < 16) {
            PanelColumn<T, J, panel_chain_op<T>(J, C - J - 2)>::run(a[C], g[C], ta, tg, nx);   // (reads row J of the tile before changing its own)
            PanelColumns<T, J, C + 1>::run(g, a, ta, tg, nx);
        }
    }
};
template <typename T, int J> struct PanelStep {
    static __device__ __forceinline__ void run(T (&g)[16], T (&a)[16], const PivotChain<T> &pc, T &dsel, int li) {
        if constexpr (J < 15) {
            const T tg = g[J] * pc.rcp, ta = a[J] * pc.rcp;   // multipliers a_iJ / d_J of the two rows
            // the next pivot's column first, so that its reciprocal is under way while the other columns are updated
            PivotChain<T> nx;
            nx.d = PanelNextPivot<T, J>::run(a[J + 1], g[J + 1], ta, tg);
            dsel = (li == J + 1) ? nx.d : dsel;
            PanelColumns<T, J, J + 2>::run(g, a, ta, tg, nx);
            nx.finish_from(panel_chain_placed<T>(J));         // what the remaining columns had no room for
            PanelStep<T, J + 1>::run(g, a, nx, dsel, li);
        }
    }
};
// g[j], a[j] *= r[lane j of the row]   (r = rsqrt of the pivots, one per lane)
template <typename T, int J> struct PanelScale {
    static __device__ __forceinline__ void run(T (&g)[16], T (&a)[16], T r) {
        const T rj = row_bcast<T, J, J == 0>(r);
        g[J] *= rj;
        a[J] *= rj;
        if constexpr (J < 15) PanelScale<T, J + 1>::run(g, a, r);
    }
};

// ---- inverse of a lower-triangular 16 x 16 LDS tile with DPP row broadcasts: one tile per ROW of 16 lanes (four tiles per
// wave at a time), lane = row.  X = L^-1 by forward substitution on rows: x_i = (e_i - sum_{j<i} L_ij x_j) / L_ii, kept
// unscaled (x'_i = x_i L_ii) until the end so that a finished row needs no write of its own before it is broadcast.
// x -= x[lane J of the row] * t.  The s_nop rides INSIDE the statement: x may have been written by a compiler-scheduled vector
// instruction (its initialisation lands wherever the scheduler likes, e.g. directly in front of this statement), and a DPP
// read needs two wait states behind such a write — found on the GPU as a first column of the inverse scaled by 1 / L_00.
template <typename T, int J> __device__ __forceinline__ void dpp_fnma_self(T &x, T t) {
    if constexpr (sizeof(T) == 8) asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(t), "n"(J));
    else asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, -%0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(t), "n"(J));
}
template <typename T, int J, int C> struct TriInvCols {         // columns C .. J of step J
    static __device__ __forceinline__ void run(T (&x)[16], T t) {
        if constexpr (C <= J) {
            dpp_fnma_self<T, J>(x[C], t);
            TriInvCols<T, J, C + 1>::run(x, t);
        }
    }
};
template <typename T, int J> struct TriInvStep {
    static __device__ __forceinline__ void run(T (&x)[16], const T (&gz)[16], T rown) {
        if constexpr (J < 15) {                               // (row 15 is nobody's source)
            const T t = gz[J] * row_bcast<T, J, 1>(rown);         // L_iJ / L_JJ for rows i > J, 0 for the others
            TriInvCols<T, J, 0>::run(x, t);
            TriInvStep<T, J + 1>::run(x, gz, rown);
        }
    }
};
// tile: this lane row's lower-triangular tile (row stride LDT, zeros above the diagonal); out: row stride ldo
template <typename T> __device__ __forceinline__ void tri_inverse_dpp(const T *tile, T *out, int ldo, int lane) {
    const int li = lane & 15;
    T gz[16], x[16], d = (T)1;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const T v = tile[li * LDT + c];
        d = (c == li) ? v : d;
        gz[c] = (c == li) ? (T)0 : v;                         // the row's own multiplier must be 0
        x[c] = (c == li) ? (T)1 : (T)0;
    }
    const T rown = (T)1 / d;
    asm volatile("s_nop 4");                                  // (EXEC / VALU writes settled in front of the first DPP read)
    TriInvStep<T, 0>::run(x, gz, rown);
#pragma unroll
    for (int c = 0; c < 16; ++c) out[li * ldo + c] = x[c] * rown;
}

// ---- trailing update of potrf_lds -------------------------------------------------------------------------------------
// Step k updates the tiles behind block column k in two phases (see potrf_lds): COLUMN = block column k + 1 alone (what the
// next panel needs), REST = the lower triangle from tile column k + 2 on.  Work items of a phase: its tiles (I, J) in
// row-major order, then the border vectors' tile columns; a wave is handed a contiguous range of that list.
// UpdPlan: the tile offsets of a wave's items, decoded ONCE per phase with one item per lane (vector integer work) and handed
// to the item loop by v_readlane — decoding per item on the scalar unit (triangular index -> row / column, three tile
// offsets) cost ~120 scalar instructions with a dozen branches, 700-900 cycles per item against 260 for the item's four
// MFMAs (stamps of the first version).
template <typename T> struct UpdPlan {
    int oa, ob, oc;          // per lane: element offsets from `tiles` of the A / B operands' and the C tile (or border entries)
    // rest != 0: item t of the lower triangle from tile index k + 2 on; rest == 0: item t of block column k + 1
    __device__ __forceinline__ void regular(int rest, int t, int nbf, int k) {
        const int tt = max(t, 0);
        int r = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
        r += ((r + 1) * (r + 2) / 2 <= tt) ? 1 : 0;
        r -= (r * (r + 1) / 2 > tt) ? 1 : 0;
        const int I = rest ? k + 2 + r : k + 1 + tt, J = rest ? k + 2 + (tt - r * (r + 1) / 2) : k + 1;
        oa = lds_tile_index(I, k, nbf) * TSZ;
        ob = lds_tile_index(J, k, nbf) * TSZ;
        oc = lds_tile_index(I, J, nbf) * TSZ;
    }
    // border item t: vector t / ncol, tile column (rest ? k + 2 : k + 1) + t % ncol   (ncol = 1 for the column phase)
    __device__ __forceinline__ void border(int rest, int t, int nbf, int k, int ncol, int border_ofs) {
        const int tt = max(t, 0), nc = max(ncol, 1);
        const int bi = (int)(((float)tt + 0.5f) * (1.0f / (float)nc)), J = (rest ? k + 2 : k + 1) + (tt - bi * nc);
        oa = border_ofs + (bi * nbf + k) * LDT;
        ob = lds_tile_index(J, k, nbf) * TSZ;
        oc = border_ofs + (bi * nbf + J) * LDT;
    }
};
#ifdef UPD_DIAG_NO_MFMA       // scratch/ubench/chol_phases.hip only: the item loop without its matrix instructions
#define UPD_MMA(x, y, z) (z)
#else
#define UPD_MMA(x, y, z) Mfma<T>::mma(x, y, z)
#endif
// One item: C -= A B^T on 16 x 16 LDS tiles (BORDER = false), or the one row c -= a B^T as row 0 of such a product (BORDER:
// row i of an MFMA result depends on row i of A only, so every lane row simply reads the border vector; the lanes that own no
// result element aim their one store at the dummy tile).  No branch anywhere, constant strides (ds_read2 / ds_write2 with
// immediate offsets).  step() runs an item while fetching the wave's next one: the compiler clusters the four dependent
// MFMAs (64 cycles each on gfx950) and the wave sits out each in front of the next; scheduling barriers keep the next item's
// address arithmetic and LDS reads in those gaps, in source order.  (fp64 MFMA and vector instructions of the same wave do
// not overlap — what the gaps hide is LDS latency, not issue.)
template <typename T, bool BORDER> struct UpdItem {
    typedef typename Mfma<T>::acc_t acc_t;
    T a[4], b[4];
    acc_t c;
    const T *pa, *pb;
    T *pc;                   // this lane's first C element; its four are pc[v * cst]   (BORDER: its one)
    static constexpr int cst = (sizeof(T) == 8 ? 4 : 1) * LDT;     // rows of a lane's C elements: Mfma<T>::row(lane, v)
    __device__ __forceinline__ void locate(T *tiles, T *dummy, const UpdPlan<T> &pl, int i, int lane) {
        const int li = lane & 15, kk = lane >> 4;
        const int oa = __builtin_amdgcn_readlane(pl.oa, i), ob = __builtin_amdgcn_readlane(pl.ob, i),
                  oc = __builtin_amdgcn_readlane(pl.oc, i);
        pb = tiles + ob + li * LDT + kk;
        if constexpr (BORDER) {
            pa = tiles + oa + kk;
            pc = Mfma<T>::row(lane, 0) == 0 ? tiles + oc + li : dummy + lane;
        } else {
            pa = tiles + oa + li * LDT + kk;
            pc = tiles + oc + Mfma<T>::row(lane, 0) * LDT + li;
        }
    }
    __device__ __forceinline__ void load_ab() {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            a[ks] = pa[4 * ks];
            b[ks] = pb[4 * ks];
        }
    }
    __device__ __forceinline__ void load_c() {
        c[0] = pc[0];
#pragma unroll
        for (int v = 1; v < 4; ++v) c[v] = BORDER ? (T)0 : pc[v * cst];
    }
    __device__ __forceinline__ void store_c() {
        pc[0] = c[0];
        if constexpr (!BORDER) {
#pragma unroll
            for (int v = 1; v < 4; ++v) pc[v * cst] = c[v];
        }
    }
    __device__ __forceinline__ void fetch(T *tiles, T *dummy, const UpdPlan<T> &pl, int i, int lane) {
        locate(tiles, dummy, pl, i, lane);
        load_ab();
        load_c();
    }
    __device__ __forceinline__ void step(UpdItem &nx, T *tiles, T *dummy, const UpdPlan<T> &pl, int inext, int lane) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) a[ks] = -a[ks];
        __builtin_amdgcn_sched_barrier(0);
        c = UPD_MMA(a[0], b[0], c);
        __builtin_amdgcn_sched_barrier(0);
        nx.locate(tiles, dummy, pl, inext, lane);
        __builtin_amdgcn_sched_barrier(0);
        c = UPD_MMA(a[1], b[1], c);
        __builtin_amdgcn_sched_barrier(0);
        nx.load_ab();
        __builtin_amdgcn_sched_barrier(0);
        c = UPD_MMA(a[2], b[2], c);
        __builtin_amdgcn_sched_barrier(0);
        nx.load_c();
        __builtin_amdgcn_sched_barrier(0);
        c = UPD_MMA(a[3], b[3], c);
        __builtin_amdgcn_sched_barrier(0);
        store_c();
    }
};
// items 0 .. n - 1 of a plan (n <= 64), two register sets
template <typename T, bool BORDER>
__device__ __forceinline__ void upd_run_items(T *tiles, T *dummy, const UpdPlan<T> &plan, int n, int lane) {
    UpdItem<T, BORDER> X, Y;
    if (n > 0) X.fetch(tiles, dummy, plan, 0, lane);
    for (int i = 0; i < n; i += 2) {
        X.step(Y, tiles, dummy, plan, min(i + 1, n - 1), lane);        // (past the end: the last item once more, never run)
        if (i + 1 >= n) break;
        Y.step(X, tiles, dummy, plan, min(i + 2, n - 1), lane);
    }
}
// number of items of a phase of step k (rest: see UpdPlan)
static __device__ __forceinline__ int potrf_lds_items(int nbf, int nborder, int k, int rest) {
    const int m = nbf - 1 - k;                                          // tiles below the diagonal one
    if (m <= 0) return 0;
    return rest ? (m - 1) * m / 2 + nborder * (m - 1) : m + nborder;
}
// This wave's share [start, start + cnt) of the item list of a phase of step k
template <typename T>
__device__ __forceinline__ void potrf_lds_update(T *tiles, T *dummy, int nbf, int nborder, int k, int rest, int start, int cnt,
                                                 int lane) {
    const int m = nbf - 1 - k;
    const int ncol = rest ? m - 1 : 1, nreg = rest ? (m - 1) * m / 2 : m;
    const int r0 = min(start, nreg), r1 = min(start + cnt, nreg);
    const int b0 = max(start, nreg) - nreg, b1 = max(start + cnt, nreg) - nreg;
    if (r1 > r0) {
        UpdPlan<T> plan;
        plan.regular(rest, r0 + lane, nbf, k);
        upd_run_items<T, false>(tiles, dummy, plan, r1 - r0, lane);
    }
    if (b1 > b0) {
        UpdPlan<T> plan;
        plan.border(rest, b0 + lane, nbf, k, ncol, (nbf * (nbf + 1) / 2) * TSZ);
        upd_run_items<T, true>(tiles, dummy, plan, b1 - b0, lane);
    }
}

// tiles: LDS array of TSZ-element tiles followed by the border vectors.  On exit the tiles hold L and the border vectors
// L^-1 v.  dinv_glob (optional): [nbf][256] global array that receives the inverted diagonal tiles (computed after the
// factorisation, off its critical path, one tile per wave at a time).  `dinv`: one LDS tile of scratch (UpdItem's dummy).
// OCC only separates instantiations: a kernel bounded to 2 workgroups per CU (256 VGPRs) must not share this function's
// register allocation with an unbounded one.
//
// Right-looking over block columns of 16, with a look-ahead of one block column:
//   PANEL k (registers, one matrix row per lane): every row of 16 lanes holds the 16 rows of the diagonal tile (full symmetric
//     rows, redundantly) and every lane one row below it (64 per wave; the border vectors are rows like any other).  Pivot j:
//     each lane scales its own entries of column j by 1/d and updates its two rows with one v_fmac_f64 per column whose
//     second factor — the entry of row j of the tile — comes from lane j of its own row of lanes by DPP.  The rows come out
//     as L_kk and A_Ik L_kk^-T: no inverse of the diagonal tile, no panel product, no cross-wave traffic.
//     (Round 2 factored + inverted the diagonal tile on one wave with v_readlane broadcasts — 3.7 us of the 5-6 us per
//     step — and formed the panel with that inverse on the matrix pipe.)
//   UPDATE k (matrix pipe): A_IJ -= P_I P_J^T for k < J <= I, in two parts: block column k + 1 by all four waves, then — at
//     the same time — panel k + 1 on the wave(s) holding its rows and the rest (J >= k + 2) on the others.  The panel is a
//     chain of 16 dependent pivots (~3100 cycles) that nothing shortens; the look-ahead takes the bulk of the matrix-pipe
//     work out from between two panels.
// PANEL phase of step k (see potrf_lds): on exit the rows below the diagonal tile are stored; g holds row (lane & 15) of L_kk,
// which the caller stores after its barrier.  Returns the first non-positive pivot of the tile (1-based) or 0.
template <typename T>
__device__ __forceinline__ int potrf_lds_panel(T *tiles, T *border, int nbf, int nborder, int k, int wv, int lane, T (&g)[16]) {
    const int li = lane & 15;
    const T *tkk = tiles + lds_tile_index(k, k, nbf) * TSZ;
    const int ntile_below = nbf - 1 - k, nrows_below = 16 * ntile_below + nborder;
    int bad = 0;
    if (64 * wv < nrows_below || wv == 0) {                   // (nrows_below <= 256: nbf <= 16)
        const int p = 64 * wv + lane;                         // this lane's row below the diagonal tile
        T *rowp = nullptr;
        if (p < 16 * ntile_below) rowp = tiles + lds_tile_index(k + 1 + (p >> 4), k, nbf) * TSZ + li * LDT;
        else if (p < nrows_below) rowp = border + ((p - 16 * ntile_below) * nbf + k) * LDT;
        T a[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            g[c] = tkk[li * LDT + c];                         // full symmetric row (potrf_lds mirrors the diagonal tiles)
            a[c] = rowp ? rowp[c] : (T)0;
        }
        asm volatile("s_nop 4");                              // (EXEC restored by the predicated loads -> DPP)
        PivotChain<T> pc;
        pc.d = row_bcast<T, 0, 0>(g[0]);
        T dsel = pc.d;                                        // lane li ends up with pivot li
        pc.finish_from(0);
        PanelStep<T, 0>::run(g, a, pc, dsel, li);
        // non-positive (or NaN) pivots: found once per tile from the 16 collected pivots; the factorisation runs on (its
        // numbers mean nothing from there on; the callers report info and poison the results)
        const unsigned long long nonpos = __builtin_amdgcn_ballot_w64(!(dsel > (T)0)) & 0xffffull;
        bad = nonpos ? __builtin_ctzll(nonpos) + 1 : 0;
        PanelScale<T, 0>::run(g, a, pivot_rsqrt(dsel));
        if (rowp) {
#pragma unroll
            for (int c = 0; c < 16; ++c) rowp[c] = a[c];
        }
    }
    return bad;
}
#define LA_PANEL_CREDIT 5     // one block column's panel phase ~ this many update items of wave time (3100 vs ~450-700 cycles)
template <typename T, int OCC = 1>
__device__ __forceinline__ void potrf_lds(T *tiles, T *dinv, int nbf_, int nbr_, int *fail, T *dinv_glob = nullptr) {
    const int lane = threadIdx.x & 63, li = lane & 15, kk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Inlined into its kernels on purpose: as a called function it received its arguments in vector registers (every loop bound
    // and tile offset treated as divergent, control flow on the vector unit under exec masks) and, the LDS pointers being
    // propagated into it as constants, re-read the dynamic-LDS base from the compiler's per-kernel offset table in memory.
    const int nbf = __builtin_amdgcn_readfirstlane(nbf_), nbr = __builtin_amdgcn_readfirstlane(nbr_);
    T *border = tiles + (size_t)(nbf * (nbf + 1) / 2) * TSZ;
    const int nborder = nbr - nbf;
    int failed = 0;                                           // first non-positive pivot, 1-based
    // the panel phase reads full symmetric rows of the diagonal tiles: mirror their lower triangles once (the updates
    // A_II -= P_I P_I^T keep them symmetric bit for bit: the same products in the same order on both sides)
    for (int e = threadIdx.x; e < nbf * 256; e += 256) {
        const int I = e >> 8, r = (e >> 4) & 15, c = e & 15;
        T *t = tiles + lds_tile_index(I, I, nbf) * TSZ;
        if (c > r) t[r * LDT + c] = t[c * LDT + r];
    }
    lds_barrier();
    T g[16];                                                  // row li of the diagonal tile -> of L_kk (panel waves)
    {
        const int k = 0;
        (void)k;
        ACC_BEGIN();
        failed = potrf_lds_panel<T>(tiles, border, nbf, nborder, 0, wv, lane, g);
        ACC_END(4);
    }
    lds_barrier();
    for (int k = 0; k < nbf; ++k) {
        // L_kk goes back only now: every panel wave read the unfactored diagonal tile during the panel phase
        if (wv == 0 && kk == 0) {
            T *tkk = tiles + lds_tile_index(k, k, nbf) * TSZ;
#pragma unroll
            for (int c = 0; c < 16; ++c) tkk[li * LDT + c] = (c <= li) ? g[c] : (T)0;
        }
        if (k + 1 == nbf) break;
        // COLUMN phase: block column k + 1, dealt evenly to the four waves
        {
            ACC_BEGIN();
            const int nc = potrf_lds_items(nbf, nborder, k, 0), per = (nc + 3) >> 2;
            potrf_lds_update<T>(tiles, dinv, nbf, nborder, k, 0, min(wv * per, nc), min(per, max(nc - wv * per, 0)), lane);
            ACC_END(5);
        }
        lds_barrier();
        // REST phase: the waves that hold rows of block column k + 1 factor it first (panel k + 1) and take LA_PANEL_CREDIT
        // items fewer of the rest of update k, which touches neither that block column nor anything the panel reads
        {
            ACC_BEGIN();
            const int nrows_next = 16 * (nbf - 2 - k) + nborder;
            const int np = min(max((nrows_next + 63) >> 6, 1), 4);         // panel waves: 0 .. np - 1
            const int nr = potrf_lds_items(nbf, nborder, k, 1);
            const int share = (nr + np * LA_PANEL_CREDIT + 3) >> 2, pshare = max(share - LA_PANEL_CREDIT, 0);
            const int mine = wv < np ? pshare : share;
            const int before = wv < np ? wv * pshare : np * pshare + (wv - np) * share;
            if (wv < np) {
                const int bad = potrf_lds_panel<T>(tiles, border, nbf, nborder, k + 1, wv, lane, g);
                failed = (bad && failed == 0) ? 16 * (k + 1) + bad : failed;
            }
            potrf_lds_update<T>(tiles, dinv, nbf, nborder, k, 1, min(before, nr), min(mine, max(nr - before, 0)), lane);
            ACC_END(6);
        }
        lds_barrier();
    }
    if (threadIdx.x == 0 && failed && *fail == 0) *fail = failed;
    if (dinv_glob) {                       // inverted diagonal tiles for the callers that go on to L^-1 (trtri_lds / potri_lds)
        lds_barrier();
        for (int k = wv; k < nbf; k += 4)
            diag_tile<T, false, true>(tiles + lds_tile_index(k, k, nbf) * TSZ, LDT, (T *)nullptr, dinv_glob + k * 256, (int *)nullptr, 0);
    }
}

// In-place inverse of the LDS-resident Cholesky factor: tiles (lower, nb x nb) hold L on entry and the lower triangle of
// (L L^T)^-1 = L^-T L^-1 on exit.  dinv_glob: the inverted diagonal tiles saved by potrf_lds; dinv: one LDS tile.
//   1. W = L^-1 row by row (rows above already hold W):  L'_ik = Linv_ii L_ik (k < i);  W_ij = delta_ij Linv_ii -
//      sum_{k=j}^{i-1} L'_ik W_kj, collected in registers and written once the whole row has been read;
//   2. out_ij = sum_{k >= i} W_ki^T W_kj row by row (row i of W is not read again by later rows).
// Every product is the tile primitive C += X Y^T of Mfma<T>::mma on 16x16 LDS tiles (Y read transposed where needed).
template <typename T>
__device__ __forceinline__ void trtri_lds(T *tiles, T *dinv, const T *dinv_glob, int nb) {       // step 1 alone: tiles <- L^-1
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
__device__ __forceinline__ void potri_lds(T *tiles, T *dinv, const T *dinv_glob, int nb) {
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
// (M > 128: at least the layout of the persistent-workgroup chain, chain_big.hip: K0 | three Mw x Mw matrices | 128 scalars)
static inline size_t chain_big_elems_inline(int M) {
    const size_t Mp = (size_t)dpgp_round_up(M, 16), Mw = (size_t)dpgp_round_up(M, 128);
    return Mp * Mp + 3 * Mw * Mw + 128;
}
static inline size_t la_chain_ws_elems_inline(int M) {
    const int Mp = dpgp_round_up(M, 16);
    const size_t plain = (size_t)3 * Mp * Mp + (size_t)(Mp + 16) * Mp + (size_t)(Mp / 16) * 256;
    const size_t big = M > 128 ? chain_big_elems_inline(M) : 0;
    return plain > big ? plain : big;
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

