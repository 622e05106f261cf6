// Batched fp64 Cholesky for M > 128 when there are enough matrices to give every compute unit one of its own (config 4:
// B = D = 256 matrices of 512 x 512 on 256 CUs; reference: tf.cholesky on [D, M, M], dp_gp_lvm.py:116,127): ONE persistent
// workgroup (4 waves, the CU's whole LDS and register file) factors a matrix from start to end — no kernel boundaries, no
// panel buffer, no explicit inverse of a diagonal block.  Right-looking over 128-wide block columns; per block column k:
//   (a) DIAGONAL BLOCK  L_kk = chol(A_kk), LDS-resident (potrf_lds: register panels with DPP row broadcasts + MFMA updates);
//       the zeros of block row k to the right of it are stored first and drain while the block is factored.
//   (b) its eight 16 x 16 diagonal tiles inverted, one tile per row of 16 lanes (tri_inverse_dpp).
//   (c) PANEL  P_I = A_Ik L_kk^-T for the 16-row tiles I below, as a chain of 16 x 16 MFMA products that never leaves the
//       registers: with X = P_I^T held in the accumulator layout, register v of a result tile IS the B operand of K-step v of
//       the next product (fp64 16x16x4: C/D row = (lane >> 4) + 4 v), so forward substitution over the eight tile columns
//       X_c = Linv_cc (A_Ic^T - sum_{c' < c} L_cc' X_c') needs only the L tiles (A operands, from LDS).  144 MFMAs per row
//       tile instead of the 256 of a product with an explicit 128 x 128 inverse, and no trtri.  Rows travel between
//       global memory and the transposed register layout through a per-wave LDS staging tile (coalesced 1 KB rows).
//   (d) UPDATE  A_IJ -= P_I P_J^T on 128 x 128 blocks, K = 128 in four chunks of 32 staged in LDS with two buffers: the next
//       chunk's global loads are in flight during the current chunk's 72-128 MFMAs per wave.  A wave owns row tiles w and
//       7 - w of a block, so that a diagonal block (lower tiles only) costs every wave the same 9 of 16 tiles.  The block's
//       old values are fetched during its last chunk and added at the end (the accumulation starts from zero).
// The multi-workgroup routine (potrf_big.hip) remains for few matrices (B < 128), fp32 and as cross-check.
#include <cstdlib>
#include "internal.h"
#include "linalg_dev.h"

#define PP_PW 128                 // block column width
#define PP_NT (PP_PW / 16)        // tiles per block edge
#define PP_SLD (PP_PW + 2)        // row stride of a wave's staging tile [16][PP_SLD] (16-byte aligned rows)
#define PP_KQ 32                  // K staged per chunk of the update
#define PP_KLD (PP_KQ + 2)        // row stride of a staged chunk [128][PP_KLD]

typedef double pp_f2 __attribute__((ext_vector_type(2)));
#ifdef PP_STAMPS              // diagnostic build (scratch/build_variant.sh stamps2 potrf_persist.hip -DPP_STAMPS): phase clocks of workgroup 0
__device__ long long g_pp_stamps[64];
extern "C" void dpgp_debug_persist_stamps(long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_stamps), sizeof(long long) * 64); }
#define PP_STAMP(i) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) g_pp_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define PP_T0() const long long pt0__ = __builtin_amdgcn_s_memtime()
#define PP_T1(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_pp_stamps[i] += __builtin_amdgcn_s_memtime() - pt0__; } while (0)
#else
#define PP_STAMP(i)
#define PP_T0()
#define PP_T1(i)
#endif

static size_t pp_lds_bytes() {
    const size_t fact = LA_LDS_HDR + sizeof(double) * ((size_t)TSZ * (PP_NT * (PP_NT + 1) / 2 + PP_NT) + (size_t)4 * 16 * PP_SLD);
    const size_t upd = LA_LDS_HDR + sizeof(double) * (size_t)2 * 2 * PP_PW * PP_KLD;
    return fact > upd ? fact : upd;
}

// ---- (c) one wave, one 16-row tile I below the diagonal block: S (staging, [16][PP_SLD]) holds A_Ik on entry and P_I on exit
// (INPLACE: the inverted diagonal tiles sit in the diagonal slots of `tiles` themselves, linv unused)
template <bool INPLACE = false>
__device__ __forceinline__ void pp_trsm_tile(double *S, const double *tiles, const double *linv, int lane) {
    typedef f64x4 acc_t;
    const int li = lane & 15, kk = lane >> 4;
    acc_t X[PP_NT];
#pragma unroll
    for (int c = 0; c < PP_NT; ++c) {
        acc_t a;
#pragma unroll
        for (int v = 0; v < 4; ++v) a[v] = S[li * PP_SLD + 16 * c + 4 * v + kk];          // (A_Ic)^T in the accumulator layout
#pragma unroll
        for (int cp = 0; cp < c; ++cp) {
            const double *lt = tiles + lds_tile_index(c, cp, PP_NT) * TSZ + li * LDT + kk;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a = __builtin_amdgcn_mfma_f64_16x16x4f64(-lt[4 * ks], X[cp][ks], a, 0, 0, 0);
        }
        const double *iv = (INPLACE ? tiles + lds_tile_index(c, c, PP_NT) * TSZ : linv + c * TSZ) + li * LDT + kk;
#ifdef PP_DEBUG_NOPS
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
        acc_t x = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) x = __builtin_amdgcn_mfma_f64_16x16x4f64(iv[4 * ks], a[ks], x, 0, 0, 0);
        X[c] = x;
#ifdef PP_DEBUG_NOPS
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
    }
#pragma unroll
    for (int c = 0; c < PP_NT; ++c)
#pragma unroll
        for (int v = 0; v < 4; ++v) S[li * PP_SLD + 16 * c + 4 * v + kk] = X[c][v];
}

__global__ __launch_bounds__(256) void pbig_persistent_kernel(int Mw, double *__restrict__ w, size_t wstride,
                                                              int *__restrict__ info) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    int &fail = *reinterpret_cast<int *>(smem_raw + 64);
    double *base = reinterpret_cast<double *>(smem_raw + LA_LDS_HDR);
    double *tiles = base;                                              // factorisation phases: 36 lower tiles of the diagonal block,
    double *linv = tiles + (size_t)TSZ * (PP_NT * (PP_NT + 1) / 2);   //   its 8 inverted diagonal tiles,
    double *stage = linv + (size_t)TSZ * PP_NT;                        //   4 staging tiles [16][PP_SLD]
    const int t = threadIdx.x, lane = t & 63, li = lane & 15, kk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = blockIdx.x, nblk = Mw / PP_PW;
    double *A = w + (size_t)b * wstride;
    constexpr int nlow = PP_NT * (PP_NT + 1) / 2;
    int first_fail = 0;
    for (int k = 0; k < nblk; ++k) {
        double *Akk = A + (size_t)(PP_PW * k) * Mw + PP_PW * k;
        const int nbelow = Mw - PP_PW * (k + 1);                       // rows below / columns to the right of the block
        PP_STAMP(8 * k + 0);
        // ---- (a) block -> LDS tiles; then the zeros to the right of the diagonal block: fire-and-forget stores that drain
        //      during the factorisation (issued in front of the loads they made the loads queue behind 393 KB of stores) ----
        if (t == 0) fail = 0;
        for (int e = t; e < nlow * 128; e += 256) {                    // (tile, row, column pair)
            const int tt = e >> 7, r = (e >> 3) & 15, c2 = (e & 7) * 2;
            int I = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
            while ((I + 1) * (I + 2) / 2 <= tt) ++I;
            while (I * (I + 1) / 2 > tt) --I;
            const int J = tt - I * (I + 1) / 2;
            const pp_f2 v = *reinterpret_cast<const pp_f2 *>(Akk + (size_t)(16 * I + r) * Mw + 16 * J + c2);
            tiles[tt * TSZ + r * LDT + c2] = v[0];
            tiles[tt * TSZ + r * LDT + c2 + 1] = v[1];
        }
        for (int e = t; e < PP_PW * (nbelow / 2); e += 256) {
            const int r = e / (nbelow / 2), c2 = (e - r * (nbelow / 2)) * 2;
            *reinterpret_cast<pp_f2 *>(Akk + (size_t)r * Mw + PP_PW + c2) = (pp_f2){0.0, 0.0};
        }
        lds_barrier();                                                 // (not __syncthreads(): the zero stores keep draining)
        PP_STAMP(8 * k + 1);
        potrf_lds<double, 1>(tiles, linv, PP_NT, PP_NT, &fail);        // (no border: `linv` is only nominally its scratch tile)
        lds_barrier();
        PP_STAMP(8 * k + 2);
        if (fail && first_fail == 0) first_fail = PP_PW * k + fail;
        for (int e = t; e < PP_PW * PP_PW / 2; e += 256) {             // L_kk back, zeros above its diagonal
            const int i = e >> 6, j = (e & 63) * 2;
            pp_f2 v;
#pragma unroll
            for (int u = 0; u < 2; ++u)
                v[u] = (j + u <= i) ? tiles[lds_tile_index(i >> 4, (j + u) >> 4, PP_NT) * TSZ + (i & 15) * LDT + ((j + u) & 15)] : 0.0;
            *reinterpret_cast<pp_f2 *>(Akk + (size_t)i * Mw + j) = v;
        }
        if (k + 1 == nblk) break;
        PP_STAMP(8 * k + 3);
        // ---- (b) inverted diagonal tiles: waves 0 and 1, one tile per row of 16 lanes ----
        if (wv < 2) {
            const int c = 4 * wv + kk;
            tri_inverse_dpp<double>(tiles + lds_tile_index(c, c, PP_NT) * TSZ, linv + c * TSZ, LDT, lane);
        }
        __syncthreads();
        PP_STAMP(8 * k + 4);
        // ---- (c) panel: row tiles I = wv, wv + 4, ... below the block; the next tile's rows are fetched while this one runs ----
        {
            const int nrt = nbelow / 16;
            double *S = stage + (size_t)wv * 16 * PP_SLD;
            double *Pk = A + (size_t)(PP_PW * (k + 1)) * Mw + PP_PW * k;          // first row below the block, block column k
            pp_f2 pre[16];
            if (wv < nrt) {
#pragma unroll
                for (int q = 0; q < 16; ++q) pre[q] = *reinterpret_cast<const pp_f2 *>(Pk + (size_t)(16 * wv + q) * Mw + 2 * lane);
            }
            for (int I = wv; I < nrt; I += 4) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {                       // (LDS is only ever accessed as double: a vector-typed store
                    S[q * PP_SLD + 2 * lane] = pre[q][0];             //  next to scalar loads of the same bytes invites
                    S[q * PP_SLD + 2 * lane + 1] = pre[q][1];         //  type-based reordering)
                }
                if (I + 4 < nrt) {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        pre[q] = *reinterpret_cast<const pp_f2 *>(Pk + (size_t)(16 * (I + 4) + q) * Mw + 2 * lane);
                }
                pp_trsm_tile(S, tiles, linv, lane);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const pp_f2 v = {S[q * PP_SLD + 2 * lane], S[q * PP_SLD + 2 * lane + 1]};
                    *reinterpret_cast<pp_f2 *>(Pk + (size_t)(16 * I + q) * Mw + 2 * lane) = v;
                }
            }
        }
        __threadfence_block();
        __syncthreads();                                               // P complete in memory; the LDS tiles are free
        PP_STAMP(8 * k + 5);
        // ---- (d) update: 128 x 128 blocks (I, J), J <= I, of the trailing matrix; stream of (block, chunk) steps ----
#ifndef PP_DEBUG_NO_UPDATE
        {
            typedef f64x4 acc_t;
            const int nub = nbelow / PP_PW, nblocks = nub * (nub + 1) / 2, nsteps = nblocks * (PP_PW / PP_KQ);
            double *T = A + (size_t)(PP_PW * (k + 1)) * Mw + PP_PW * (k + 1);     // trailing matrix
            const double *P = A + (size_t)(PP_PW * (k + 1)) * Mw + PP_PW * k;     // panel: row r at P + r * Mw, 128 columns
            const int rt0 = wv, rt1 = PP_NT - 1 - wv;                             // this wave's two row tiles of a block
            acc_t acc[2][PP_NT];
            pp_f2 px[8], py[8];                                                    // this thread's share of a staged chunk pair
            // thread -> (row, column pair) of a [128][32] chunk: 16 threads per row
            const int srow = t >> 4, sc2 = (t & 15) * 2;
            int bI = 0, bJ = 0;                                                    // block of the step being FETCHED
            auto fetch = [&](int step) {
                const int h = step & 3;
                const double *pi = P + (size_t)(PP_PW * bI) * Mw + PP_KQ * h, *pj = P + (size_t)(PP_PW * bJ) * Mw + PP_KQ * h;
#pragma unroll
                for (int q = 0; q < 8; ++q) px[q] = *reinterpret_cast<const pp_f2 *>(pi + (size_t)(srow + 16 * q) * Mw + sc2);
                if (bI != bJ) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) py[q] = *reinterpret_cast<const pp_f2 *>(pj + (size_t)(srow + 16 * q) * Mw + sc2);
                }
                if (h == 3) {                                                      // advance to the next block (row-major lower)
                    if (bJ < bI) ++bJ;
                    else { ++bI; bJ = 0; }
                }
            };
            auto stash = [&](int step, bool diag) {                                // registers -> LDS buffer (step & 1): -P_I, P_J
                double *xs = base + (size_t)(step & 1) * 2 * PP_PW * PP_KLD, *ys = xs + (size_t)PP_PW * PP_KLD;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const pp_f2 vx = px[q], vy = diag ? px[q] : py[q];
                    double *dx = xs + (srow + 16 * q) * PP_KLD + sc2, *dy = ys + (srow + 16 * q) * PP_KLD + sc2;
                    dx[0] = -vx[0]; dx[1] = -vx[1];
                    dy[0] = vy[0]; dy[1] = vy[1];
                }
            };
            int cI = 0, cJ = 0;                                                    // block of the step being COMPUTED
            int fI = 0, fJ = 0;                                                    // block whose chunk sits in px / py
            if (nsteps > 0) {
                fI = bI; fJ = bJ;
                fetch(0);
                stash(0, fI == fJ);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int Jt = 0; Jt < PP_NT; ++Jt) acc[i][Jt] = (acc_t){0, 0, 0, 0};
            for (int step = 0; step < nsteps; ++step) {
                const int h = step & 3;
                const bool diag = (cI == cJ);
                if (step + 1 < nsteps) {
                    fI = bI; fJ = bJ;
                    fetch(step + 1);
                }
                double *Cb = T + (size_t)(PP_PW * cI) * Mw + PP_PW * cJ;
                acc_t cold[2][PP_NT];
                if (h == 3) {                                                      // the block's old values, needed after this chunk
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int rt = i ? rt1 : rt0;
#pragma unroll
                        for (int Jt = 0; Jt < PP_NT; ++Jt)
                            if (!diag || Jt <= rt) {
#pragma unroll
                                for (int v = 0; v < 4; ++v) cold[i][Jt][v] = Cb[(size_t)(16 * rt + kk + 4 * v) * Mw + 16 * Jt + li];
                            }
                    }
                }
                const double *xs = base + (size_t)(step & 1) * 2 * PP_PW * PP_KLD, *ys = xs + (size_t)PP_PW * PP_KLD;
#pragma unroll 2
                for (int ks = 0; ks < PP_KQ / 4; ++ks) {
                    const double x0 = xs[(16 * rt0 + li) * PP_KLD + 4 * ks + kk], x1 = xs[(16 * rt1 + li) * PP_KLD + 4 * ks + kk];
#pragma unroll
                    for (int Jt = 0; Jt < PP_NT; ++Jt) {
                        if (diag && Jt > rt1) continue;                            // (rt0 <= rt1: nothing of this column for the wave)
                        const double yv = ys[(16 * Jt + li) * PP_KLD + 4 * ks + kk];
                        if (!diag || Jt <= rt0) acc[0][Jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, yv, acc[0][Jt], 0, 0, 0);
                        acc[1][Jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, yv, acc[1][Jt], 0, 0, 0);
                    }
                }
                if (h == 3) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int rt = i ? rt1 : rt0;
#pragma unroll
                        for (int Jt = 0; Jt < PP_NT; ++Jt) {
                            if (!diag || Jt <= rt) {
#pragma unroll
                                for (int v = 0; v < 4; ++v)
                                    Cb[(size_t)(16 * rt + kk + 4 * v) * Mw + 16 * Jt + li] = cold[i][Jt][v] + acc[i][Jt][v];
                            }
                            acc[i][Jt] = (acc_t){0, 0, 0, 0};
                        }
                    }
                    if (cJ < cI) ++cJ;
                    else { ++cI; cJ = 0; }
                }
                if (step + 1 < nsteps) stash(step + 1, fI == fJ);
                __syncthreads();
            }
        }
#endif
        __threadfence_block();
        __syncthreads();
        PP_STAMP(8 * k + 6);
    }
    PP_STAMP(63);
    if (t == 0) info[b] = first_fail;
}

// ---- LEFT-LOOKING form of the same factorisation (round 3, second half): block column k is first brought up to date with
// all block columns to its left — the accumulators of a 128 x 128 block run over K = 128 k without touching memory — and is
// then finished where it sits: the diagonal block goes from the accumulators straight into the LDS tiles of potrf_lds, a
// block below it from the accumulators through the wave's staging tile into the register-chained solve of pp_trsm_tile and
// only then to memory.  Every block of the matrix is read once (its old values) and written once (its part of L); the
// right-looking kernel above reads and writes a trailing block once per block column to its left and a panel block twice more
// (update out, panel in / out): 4.9 against 6.8 MB per 512 x 512 matrix, and HBM is what binds this kernel.
// LDS: the 36 tiles of the diagonal block (their diagonal tiles replaced by their inverses once L_kk is stored) + ONE staged
// chunk pair [128][32] x 2 (the next chunk waits in registers), which the four staging tiles of the solve alias.
#define PL_KQ 32
#define PL_KLD (PL_KQ + 2)
static size_t pl_lds_bytes() {
    return LA_LDS_HDR + sizeof(double) * ((size_t)TSZ * (PP_NT * (PP_NT + 1) / 2) + (size_t)2 * PP_PW * PL_KLD);
}
static_assert(2 * PP_PW * PL_KLD >= 4 * 16 * PP_SLD, "the staging tiles of the solve alias the chunk pair");

__global__ __launch_bounds__(256) void pleft_persistent_kernel(int Mw, double *__restrict__ w, size_t wstride,
                                                               int *__restrict__ info) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    int &fail = *reinterpret_cast<int *>(smem_raw + 64);
    double *tiles = reinterpret_cast<double *>(smem_raw + LA_LDS_HDR);
    double *chunk = tiles + (size_t)TSZ * (PP_NT * (PP_NT + 1) / 2);   // xs [128][PL_KLD] | ys [128][PL_KLD]
    double *stage = chunk;                                             // (aliases: 4 x [16][PP_SLD], used between update loops)
    typedef f64x4 acc_t;
    const int t = threadIdx.x, lane = t & 63, li = lane & 15, kk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = blockIdx.x, nblk = Mw / PP_PW;
    double *A = w + (size_t)b * wstride;
    constexpr int nlow = PP_NT * (PP_NT + 1) / 2;
    const int rt0 = wv, rt1 = PP_NT - 1 - wv;                          // this wave's two row tiles of a block
    const int srow = t >> 4, sc2 = (t & 15) * 2;                       // thread -> (row, column pair) of a [128][32] chunk
    int first_fail = 0;
#ifdef PL_DEPHASE_SLEEPS           // (experiment, scratch/dephase.sh: scratch/build_variant.sh dp<n> potrf_persist.hip -DPL_DEPHASE_SLEEPS=<n>)
    if (b & 1)                     // odd workgroups start late: their HBM bursts fall between the even ones' — measured: no gain
        for (int i = 0; i < PL_DEPHASE_SLEEPS; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    acc_t acc[2][PP_NT];
    // acc <- old values of block (Iblk, k) - sum_{j < k} P_Iblk,j P_k,j^T   (k >= 1; diag: Iblk == k, lower tiles only).  The
    // accumulators START from the old values (their loads are issued with the first chunk's and land beneath its staging): a
    // second register set for them at the end of the loop, as in the right-looking kernel, spilled 91 registers here.
    auto update_block = [&](int Iblk, int k, bool diag) __attribute__((always_inline)) {
        const int nch = (PP_PW / PL_KQ) * k;
        const double *Pi = A + (size_t)(PP_PW * Iblk) * Mw, *Pk = A + (size_t)(PP_PW * k) * Mw;
        const double *Cb = Pi + PP_PW * k;
        pp_f2 px[8], py[8];
        auto fetch = [&](int m) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 8; ++q) px[q] = *reinterpret_cast<const pp_f2 *>(Pi + (size_t)(srow + 16 * q) * Mw + PL_KQ * m + sc2);
            if (!diag) {
#pragma unroll
                for (int q = 0; q < 8; ++q) py[q] = *reinterpret_cast<const pp_f2 *>(Pk + (size_t)(srow + 16 * q) * Mw + PL_KQ * m + sc2);
            }
        };
        double *xs = chunk, *ys = chunk + (size_t)PP_PW * PL_KLD;
        fetch(0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rt = i ? rt1 : rt0;
#pragma unroll
            for (int Jt = 0; Jt < PP_NT; ++Jt) {
                if (!diag || Jt <= rt) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[i][Jt][v] = Cb[(size_t)(16 * rt + kk + 4 * v) * Mw + 16 * Jt + li];
                } else {
                    acc[i][Jt] = (acc_t){0, 0, 0, 0};
                }
            }
        }
        for (int m = 0; m < nch; ++m) {
            __syncthreads();                                           // the previous chunk (or the staging tiles) is no longer read
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const pp_f2 vx = px[q], vy = diag ? px[q] : py[q];
                double *dx = xs + (srow + 16 * q) * PL_KLD + sc2, *dy = ys + (srow + 16 * q) * PL_KLD + sc2;
                dx[0] = -vx[0]; dx[1] = -vx[1];
                dy[0] = vy[0]; dy[1] = vy[1];
            }
            __syncthreads();
            if (m + 1 < nch) fetch(m + 1);
#pragma unroll 2
            for (int ks = 0; ks < PL_KQ / 4; ++ks) {
                const double x0 = xs[(16 * rt0 + li) * PL_KLD + 4 * ks + kk], x1 = xs[(16 * rt1 + li) * PL_KLD + 4 * ks + kk];
#pragma unroll
                for (int Jt = 0; Jt < PP_NT; ++Jt) {
                    if (diag && Jt > rt1) continue;                    // (rt0 <= rt1: nothing of this column for the wave)
                    const double yv = ys[(16 * Jt + li) * PL_KLD + 4 * ks + kk];
                    if (!diag || Jt <= rt0) acc[0][Jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, yv, acc[0][Jt], 0, 0, 0);
                    acc[1][Jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, yv, acc[1][Jt], 0, 0, 0);
                }
            }
        }
    };
    for (int k = 0; k < nblk; ++k) {
        double *Akk = A + (size_t)(PP_PW * k) * Mw + PP_PW * k;
        const int nbelow = Mw - PP_PW * (k + 1);
        if (t == 0) fail = 0;
        PP_STAMP(8 * k + 0);
        // ---- the diagonal block, up to date, into the LDS tiles ----
        if (k == 0) {
            // (tile, row, column pair); six independent loads in flight per thread: one at a time this loop took 47k cycles
            for (int e0 = t; e0 < nlow * 128; e0 += 256 * 6) {
                pp_f2 v[6];
                int dst[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    const int e = e0 + 256 * u;
                    const int tt = min(e >> 7, nlow - 1), r = (e >> 3) & 15, c2 = (e & 7) * 2;
                    int I = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
                    while ((I + 1) * (I + 2) / 2 <= tt) ++I;
                    while (I * (I + 1) / 2 > tt) --I;
                    const int J = tt - I * (I + 1) / 2;
                    v[u] = *reinterpret_cast<const pp_f2 *>(Akk + (size_t)(16 * I + r) * Mw + 16 * J + c2);
                    dst[u] = (e < nlow * 128) ? tt * TSZ + r * LDT + c2 : -1;
                }
#pragma unroll
                for (int u = 0; u < 6; ++u)
                    if (dst[u] >= 0) {
                        tiles[dst[u]] = v[u][0];
                        tiles[dst[u] + 1] = v[u][1];
                    }
            }
        } else {
            update_block(k, k, true);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rt = i ? rt1 : rt0;
#pragma unroll
                for (int Jt = 0; Jt < PP_NT; ++Jt)
                    if (Jt <= rt) {
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            tiles[lds_tile_index(rt, Jt, PP_NT) * TSZ + (kk + 4 * v) * LDT + li] = acc[i][Jt][v];
                    }
            }
        }
        for (int e = t; e < PP_PW * (nbelow / 2); e += 256) {          // zeros to the right of the block: drain behind the factorisation
            const int r = e / (nbelow / 2), c2 = (e - r * (nbelow / 2)) * 2;
            *reinterpret_cast<pp_f2 *>(Akk + (size_t)r * Mw + PP_PW + c2) = (pp_f2){0.0, 0.0};
        }
        __syncthreads();                                               // (also: nobody reads the chunk pair any more)
        PP_STAMP(8 * k + 1);
        potrf_lds<double, 1>(tiles, chunk, PP_NT, PP_NT, &fail);       // (`chunk`: its scratch tile)
        lds_barrier();
        PP_STAMP(8 * k + 2);
        if (fail && first_fail == 0) first_fail = PP_PW * k + fail;
        for (int e = t; e < PP_PW * PP_PW / 2; e += 256) {             // L_kk out, zeros above its diagonal
            const int i = e >> 6, j = (e & 63) * 2;
            pp_f2 v;
#pragma unroll
            for (int u = 0; u < 2; ++u)
                v[u] = (j + u <= i) ? tiles[lds_tile_index(i >> 4, (j + u) >> 4, PP_NT) * TSZ + (i & 15) * LDT + ((j + u) & 15)] : 0.0;
            *reinterpret_cast<pp_f2 *>(Akk + (size_t)i * Mw + j) = v;
        }
        if (k + 1 == nblk) break;
        lds_barrier();                                                 // the diagonal tiles have been read out
        if (wv < 2) {                                                  // their inverses, in place
            const int c = 4 * wv + kk;
            double *tc = tiles + lds_tile_index(c, c, PP_NT) * TSZ;
            tri_inverse_dpp<double>(tc, tc, LDT, lane);
        }
        __syncthreads();
        PP_STAMP(8 * k + 3);
        // ---- the blocks below: up to date in the accumulators, solved through the staging tile, stored once ----
        double *S = stage + (size_t)wv * 16 * PP_SLD;
        pp_f2 pre[16];                                                 // k == 0: the next row tile's rows, fetched beneath the current solve
        if (k == 0) {
            const double *P0 = A + (size_t)(PP_PW * 1) * Mw;
#pragma unroll
            for (int q = 0; q < 16; ++q) pre[q] = *reinterpret_cast<const pp_f2 *>(P0 + (size_t)(16 * rt0 + q) * Mw + 2 * lane);
        }
        for (int I = k + 1; I < nblk; ++I) {
            double *Pik = A + (size_t)(PP_PW * I) * Mw + PP_PW * k;
            PP_T0();
            if (k > 0) {
                update_block(I, k, false);
                __syncthreads();                                       // every wave is done with the chunk pair: staging may begin
            }
            PP_T1(40 + k);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rt = i ? rt1 : rt0;
                if (k == 0) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        S[q * PP_SLD + 2 * lane] = pre[q][0];
                        S[q * PP_SLD + 2 * lane + 1] = pre[q][1];
                    }
                    const int In = i ? I + 1 : I, rn = i ? rt0 : rt1;  // the tile after this one
                    if (In < nblk) {
                        const double *Pn = A + (size_t)(PP_PW * In) * Mw;
#pragma unroll
                        for (int q = 0; q < 16; ++q) pre[q] = *reinterpret_cast<const pp_f2 *>(Pn + (size_t)(16 * rn + q) * Mw + 2 * lane);
                    }
                } else {
#pragma unroll
                    for (int Jt = 0; Jt < PP_NT; ++Jt)
#pragma unroll
                        for (int v = 0; v < 4; ++v) S[(kk + 4 * v) * PP_SLD + 16 * Jt + li] = acc[i][Jt][v];
                }
                pp_trsm_tile<true>(S, tiles, nullptr, lane);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const pp_f2 v = {S[q * PP_SLD + 2 * lane], S[q * PP_SLD + 2 * lane + 1]};
                    *reinterpret_cast<pp_f2 *>(Pik + (size_t)(16 * rt + q) * Mw + 2 * lane) = v;
                }
            }
        }
        __threadfence_block();
        __syncthreads();                                               // block column k complete in memory
        PP_STAMP(8 * k + 4);
    }
    PP_STAMP(63);
    if (t == 0) info[b] = first_fail;
}

// ---- X = L^-1 R in place, R lower triangular (zeros above the diagonal), same persistent scheme as the factorisation above:
// right-looking over 128-row block rows; per block row k
//   (a), (b) L_kk -> LDS tiles, its diagonal tiles inverted;
//   (c) X_k. = L_kk^-1 G_k. for the 8 (k + 1) tiles of 16 columns of the block row: pp_trsm_tile solves L_kk X^T... exactly this
//       left solve on a [16 columns][128 rows] staging tile (there it is the transposed panel tile);
//   (d) X_IJ -= L_Ik X_kJ for the blocks I > k, J <= k: the update loop of the factorisation with the A operand chunk from L
//       (negated) and the B operand chunk [32][128] from block row k of X (row-major in k: no transposition).
// Used by the fused ELBO for M > 128 (chain_big.hip): only |X|_F^2 leaves the kernel (nrm2[b]); X itself is scratch (the
// last block row is not even stored).  L[B][Mw][Mw] read-only (lower + zeros), X[B][Mw][Mw] in/out.
#define PT_YLD 144                 // row stride of a staged [32][128] chunk of X: lanes (kk, li) of an operand read hit 64 banks
static size_t pt_lds_bytes() {
    const size_t fact = LA_LDS_HDR + sizeof(double) * ((size_t)TSZ * (PP_NT * (PP_NT + 1) / 2 + PP_NT) + (size_t)4 * 16 * PP_SLD);
    const size_t upd = LA_LDS_HDR + sizeof(double) * (size_t)2 * (PP_PW * PP_KLD + PP_KQ * PT_YLD);
    return fact > upd ? fact : upd;
}

__global__ __launch_bounds__(256) void ptrsm_persistent_kernel(int Mw, const double *__restrict__ Lall, size_t lstride,
                                                               double *__restrict__ Xall, size_t xstride,
                                                               double *__restrict__ nrm2, size_t nstride, int identity) {
    // identity bit 1 (dpgp_trtri_lower_batched_f64): X itself is the result — the last block row is stored as well
    const bool keep_all = (identity & 2) != 0;
    identity &= 1;
    // identity != 0: R = I — X need not be initialised (its diagonal blocks are taken as I, the blocks below as 0 where they
    // are first touched): the caller saves writing Mw^2 / 2 doubles per matrix and this kernel reading them
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double *scratch = reinterpret_cast<double *>(smem_raw);
    double *base = reinterpret_cast<double *>(smem_raw + LA_LDS_HDR);
    double *tiles = base;
    double *linv = tiles + (size_t)TSZ * (PP_NT * (PP_NT + 1) / 2);
    double *stage = linv + (size_t)TSZ * PP_NT;
    const int t = threadIdx.x, lane = t & 63, li = lane & 15, kk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = blockIdx.x, nblk = Mw / PP_PW;
    const double *L = Lall + (size_t)b * lstride;
    double *X = Xall + (size_t)b * xstride;
    constexpr int nlow = PP_NT * (PP_NT + 1) / 2;
    double fro = 0.0;
    for (int k = 0; k < nblk; ++k) {
        const double *Lkk = L + (size_t)(PP_PW * k) * Mw + PP_PW * k;
        // ---- (a) L_kk -> LDS tiles ----
        for (int e = t; e < nlow * 128; e += 256) {
            const int tt = e >> 7, r = (e >> 3) & 15, c2 = (e & 7) * 2;
            int I = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
            while ((I + 1) * (I + 2) / 2 <= tt) ++I;
            while (I * (I + 1) / 2 > tt) --I;
            const int J = tt - I * (I + 1) / 2;
            const pp_f2 v = *reinterpret_cast<const pp_f2 *>(Lkk + (size_t)(16 * I + r) * Mw + 16 * J + c2);
            tiles[tt * TSZ + r * LDT + c2] = v[0];
            tiles[tt * TSZ + r * LDT + c2 + 1] = v[1];
        }
        __syncthreads();
        // ---- (b) inverted diagonal tiles ----
        if (wv < 2) {
            const int c = 4 * wv + kk;
            tri_inverse_dpp<double>(tiles + lds_tile_index(c, c, PP_NT) * TSZ, linv + c * TSZ, LDT, lane);
        }
        __syncthreads();
        // ---- (c) block row k: column tiles ct = wv, wv + 4, ...; lane -> (row r0 + 8 q of the block, column pair c2 of the tile) ----
        {
            const int nct = PP_NT * (k + 1), r0 = lane >> 3, c2 = (lane & 7) * 2;
            double *S = stage + (size_t)wv * 16 * PP_SLD;
            double *Xk = X + (size_t)(PP_PW * k) * Mw;
            const bool keep = (k + 1 < nblk) || keep_all;      // the last block row is only summed (fused ELBO)
            pp_f2 pre[16];
            if (wv < nct) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int col = 16 * wv + c2 - PP_PW * k, r = r0 + 8 * q;             // (column inside the diagonal block if >= 0)
                    if (identity && col >= 0) pre[q] = (pp_f2){r == col ? 1.0 : 0.0, r == col + 1 ? 1.0 : 0.0};
                    else pre[q] = *reinterpret_cast<const pp_f2 *>(Xk + (size_t)r * Mw + 16 * wv + c2);
                }
            }
            for (int ct = wv; ct < nct; ct += 4) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    S[c2 * PP_SLD + r0 + 8 * q] = pre[q][0];
                    S[(c2 + 1) * PP_SLD + r0 + 8 * q] = pre[q][1];
                }
                if (ct + 4 < nct) {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                    {
                        const int col = 16 * (ct + 4) + c2 - PP_PW * k, r = r0 + 8 * q;
                        if (identity && col >= 0) pre[q] = (pp_f2){r == col ? 1.0 : 0.0, r == col + 1 ? 1.0 : 0.0};
                        else pre[q] = *reinterpret_cast<const pp_f2 *>(Xk + (size_t)r * Mw + 16 * (ct + 4) + c2);
                    }
                }
                pp_trsm_tile(S, tiles, linv, lane);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const pp_f2 v = {S[c2 * PP_SLD + r0 + 8 * q], S[(c2 + 1) * PP_SLD + r0 + 8 * q]};
                    fro += v[0] * v[0] + v[1] * v[1];
                    if (keep) *reinterpret_cast<pp_f2 *>(Xk + (size_t)(r0 + 8 * q) * Mw + 16 * ct + c2) = v;
                }
            }
        }
        if (k + 1 == nblk) break;
        __threadfence_block();
        __syncthreads();                                               // block row k complete in memory; the LDS tiles are free
        // ---- (d) update of the blocks (I, J), I > k, J <= k ----
        {
            typedef f64x4 acc_t;
            const int nI = nblk - 1 - k, nJ = k + 1, nsteps = nI * nJ * (PP_PW / PP_KQ);
            const double *Lk = L + (size_t)(PP_PW * (k + 1)) * Mw + PP_PW * k;    // L_Ik: row r of block I at Lk + (128 I + r) Mw
            const double *Xk = X + (size_t)(PP_PW * k) * Mw;                      // block row k of X
            double *T = X + (size_t)(PP_PW * (k + 1)) * Mw;                       // the block rows below
            const int rt0 = wv, rt1 = PP_NT - 1 - wv;
            acc_t acc[2][PP_NT];
            pp_f2 px[8], py[8];
            const int srow = t >> 4, sc2 = (t & 15) * 2;                           // [128][32] chunk of L: 16 threads per row
            const int yrow = t >> 6, yc2 = (t & 63) * 2;                           // [32][128] chunk of X: 64 threads per row
            int bI = 0, bJ = 0;
            auto fetch = [&](int step) {
                const int h = step & 3;
                const double *pi = Lk + (size_t)(PP_PW * bI) * Mw + PP_KQ * h;
                const double *pj = Xk + (size_t)(PP_KQ * h) * Mw + PP_PW * bJ;
#pragma unroll
                for (int q = 0; q < 8; ++q) px[q] = *reinterpret_cast<const pp_f2 *>(pi + (size_t)(srow + 16 * q) * Mw + sc2);
#pragma unroll
                for (int q = 0; q < 8; ++q) py[q] = *reinterpret_cast<const pp_f2 *>(pj + (size_t)(yrow + 4 * q) * Mw + yc2);
                if (h == 3) {
                    if (bJ + 1 < nJ) ++bJ;
                    else { ++bI; bJ = 0; }
                }
            };
            auto stash = [&](int step) {
                double *xs = base + (size_t)(step & 1) * (PP_PW * PP_KLD + PP_KQ * PT_YLD), *ys = xs + (size_t)PP_PW * PP_KLD;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    double *dx = xs + (srow + 16 * q) * PP_KLD + sc2, *dy = ys + (yrow + 4 * q) * PT_YLD + yc2;
                    dx[0] = -px[q][0]; dx[1] = -px[q][1];
                    dy[0] = py[q][0]; dy[1] = py[q][1];
                }
            };
            int cI = 0, cJ = 0;
            if (nsteps > 0) {
                fetch(0);
                stash(0);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int Jt = 0; Jt < PP_NT; ++Jt) acc[i][Jt] = (acc_t){0, 0, 0, 0};
            for (int step = 0; step < nsteps; ++step) {
                const int h = step & 3;
                if (step + 1 < nsteps) fetch(step + 1);
                double *Cb = T + (size_t)(PP_PW * cI) * Mw + PP_PW * cJ;
                acc_t cold[2][PP_NT];
                if (h == 3) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int rt = i ? rt1 : rt0;
#pragma unroll
                        for (int Jt = 0; Jt < PP_NT; ++Jt)
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                cold[i][Jt][v] = (identity && cJ == k) ? 0.0 : Cb[(size_t)(16 * rt + kk + 4 * v) * Mw + 16 * Jt + li];
                    }
                }
                const double *xs = base + (size_t)(step & 1) * (PP_PW * PP_KLD + PP_KQ * PT_YLD), *ys = xs + (size_t)PP_PW * PP_KLD;
#pragma unroll 2
                for (int ks = 0; ks < PP_KQ / 4; ++ks) {
                    const double x0 = xs[(16 * rt0 + li) * PP_KLD + 4 * ks + kk], x1 = xs[(16 * rt1 + li) * PP_KLD + 4 * ks + kk];
#pragma unroll
                    for (int Jt = 0; Jt < PP_NT; ++Jt) {
                        const double yv = ys[(4 * ks + kk) * PT_YLD + 16 * Jt + li];
                        acc[0][Jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, yv, acc[0][Jt], 0, 0, 0);
                        acc[1][Jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, yv, acc[1][Jt], 0, 0, 0);
                    }
                }
                if (h == 3) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int rt = i ? rt1 : rt0;
#pragma unroll
                        for (int Jt = 0; Jt < PP_NT; ++Jt) {
#pragma unroll
                            for (int v = 0; v < 4; ++v)
                                Cb[(size_t)(16 * rt + kk + 4 * v) * Mw + 16 * Jt + li] = cold[i][Jt][v] + acc[i][Jt][v];
                            acc[i][Jt] = (acc_t){0, 0, 0, 0};
                        }
                    }
                    if (cJ + 1 < nJ) ++cJ;
                    else { ++cI; cJ = 0; }
                }
                if (step + 1 < nsteps) stash(step + 1);
                __syncthreads();
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    __syncthreads();
    fro = block_sum(fro, scratch);
    if (t == 0) nrm2[(size_t)b * nstride] = fro;
}

// L^-1 of B lower-triangular factors [M][M], M a multiple of 128 (the block width of the persistent kernels): out = L^-1, lower, zeros
// above the diagonal.  The solve X = L^-1 I of the persistent kernel above with every block row stored; the blocks above the diagonal
// are zeroed here (the kernel never touches them).  ws: B doubles (the squared norms the kernel also forms).   (stage A of the backward
// pass for M > 128, ops._elbo_grad_chain_large: K^-1 = W^T W, B^-1 likewise; tf.matrix_triangular_solve on the identity)
extern "C" int dpgp_trtri_lower_batched_f64(int B, int M, const double *l, double *out, void *ws, size_t ws_bytes, void *stream) {
    if (B <= 0) return -1;
    if (M <= 0 || M % PP_PW != 0) return -2;
    if (!l) return -3;
    if (!out) return -4;
    if (!ws) return -5;
    if (ws_bytes < sizeof(double) * (size_t)B) return -6;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(double) * (size_t)B * M * M, st) != hipSuccess) return DPGP_ERR_LAUNCH;
    return launch_ptrsm_persist(B, M, l, (size_t)M * M, out, (size_t)M * M, (double *)ws, 1, st, 3);
}

int launch_ptrsm_persist(int B, int Mw, const double *l, size_t lstride, double *x, size_t xstride, double *nrm2,
                         size_t nstride, hipStream_t st, int identity) {
    const size_t lds = pt_lds_bytes();
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(ptrsm_persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(ptrsm_persistent_kernel, dim3(B), dim3(256), lds, st, Mw, l, lstride, x, xstride, nrm2, nstride, identity);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

// ---- c = L_t^-1 V_t for all D columns of V_t (the over-T model, dp_gp_lvm.py:657-667: every atom t is solved against all
// columns of Psi1_t^T Y) and quad[t][d] = scale_t^2 |c_td|^2.  M <= 128; L_t arrives as the LDS image of its lower tiles
// (chain_b_kernel, lb_out), V_t as nsl slabs of a split-k product.  Workgroup = (64 columns, t), wave = 16 columns: the
// register-chained left solve of pp_trsm_tile on a [16 columns][128 rows] staging tile, as in ptrsm_persistent_kernel (c).
__global__ __launch_bounds__(256) void tcols_quad_kernel(int M, int Mp, int D, const double *__restrict__ lb,
                                                         const double *__restrict__ vp, int nsl, long long v_ss,
                                                         const double *__restrict__ scale, double *__restrict__ quad) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double *tiles = reinterpret_cast<double *>(smem_raw + LA_LDS_HDR);
    double *linv = tiles + (size_t)TSZ * (PP_NT * (PP_NT + 1) / 2);
    double *stage = linv + (size_t)TSZ * PP_NT;
    const int t = threadIdx.x, lane = t & 63, kk = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int at = blockIdx.y, nb = Mp / 16, nlow = nb * (nb + 1) / 2;
    const double *lt = lb + (size_t)at * nlow * TSZ;
    {   // the image verbatim, 8 independent 16-byte loads in flight per thread (nlow TSZ is even, the image 16-byte aligned)
        const pp_f2 *src = reinterpret_cast<const pp_f2 *>(lt);
        const int npair = nlow * TSZ / 2;
        for (int e0 = t; e0 < npair; e0 += 256 * 8) {
            pp_f2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + 256 * u < npair) v[u] = src[e0 + 256 * u];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + 256 * u < npair) {
                    tiles[2 * (e0 + 256 * u)] = v[u][0];
                    tiles[2 * (e0 + 256 * u) + 1] = v[u][1];
                }
        }
    }
    for (int e = nlow * TSZ + t; e < PP_NT * (PP_NT + 1) / 2 * TSZ; e += 256) {      // identity padding up to 128 rows
        const int tt = e / TSZ, r = (e - tt * TSZ) / LDT, c = e - tt * TSZ - r * LDT;
        int I = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
        while ((I + 1) * (I + 2) / 2 <= tt) ++I;
        while (I * (I + 1) / 2 > tt) --I;
        tiles[e] = (tt == I * (I + 1) / 2 + I && r == c) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (wv < 2) {
        const int c = 4 * wv + kk;
        tri_inverse_dpp<double>(tiles + lds_tile_index(c, c, PP_NT) * TSZ, linv + c * TSZ, LDT, lane);
    }
    __syncthreads();
    const int col0 = 64 * blockIdx.x + 16 * wv;
    if (col0 >= D) return;
    const int r0 = lane >> 3, c2 = (lane & 7) * 2;
    double *S = stage + (size_t)wv * 16 * PP_SLD;
    const double *vt = vp + (size_t)at * M * D;
    // the slabs of the split-k product are added on the way in; all loads of a slab are issued before its sums
    {
        double a0[16], a1[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a0[q] = a1[q] = 0.0;
        const bool in0 = col0 + c2 < D, in1 = col0 + c2 + 1 < D;
        for (int ks = 0; ks < nsl; ++ks) {
            const double *sl = vt + (long long)ks * v_ss + col0 + c2;
            double x0[16], x1[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = r0 + 8 * q;
                x0[q] = (r < M && in0) ? sl[(size_t)r * D] : 0.0;
                x1[q] = (r < M && in1) ? sl[(size_t)r * D + 1] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                a0[q] += x0[q];
                a1[q] += x1[q];
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            S[c2 * PP_SLD + r0 + 8 * q] = a0[q];
            S[(c2 + 1) * PP_SLD + r0 + 8 * q] = a1[q];
        }
    }
    pp_trsm_tile(S, tiles, linv, lane);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const double x0 = S[c2 * PP_SLD + r0 + 8 * q], x1 = S[(c2 + 1) * PP_SLD + r0 + 8 * q];
        s0 += x0 * x0;
        s1 += x1 * x1;
    }
#pragma unroll
    for (int o = 8; o <= 32; o <<= 1) {
        s0 += __shfl_xor(s0, o, 64);
        s1 += __shfl_xor(s1, o, 64);
    }
    if (r0 == 0) {
        const double sc = scale[at] * scale[at];
        if (col0 + c2 < D) quad[(size_t)at * D + col0 + c2] = sc * s0;
        if (col0 + c2 + 1 < D) quad[(size_t)at * D + col0 + c2 + 1] = sc * s1;
    }
}

int launch_tcols_quad(int T, int M, int D, const double *lb, const double *vp, int nsl, long long v_ss, const double *scale,
                      double *quad, hipStream_t st) {
    const int Mp = dpgp_round_up(M, 16);
    if (Mp > PP_PW) return -30;
    const size_t lds = pp_lds_bytes();
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(tcols_quad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(tcols_quad_kernel, dim3(dpgp_ceil_div(D, 64), T), dim3(256), lds, st, M, Mp, D, lb, vp, nsl, v_ss, scale, quad);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

bool potrf_persist_applicable(int B, int M, int elem_size) { return elem_size == 8 && M > 128 && B >= 128; }

int launch_potrf_persist(int B, int Mw, double *w, int *info, hipStream_t st, size_t wstride) {
    if (!wstride) wstride = (size_t)Mw * Mw;
    const char *le_ = getenv("DPGP_POTRF_LEFT");                 // (0: the right-looking kernel, cross-checks / profiling)
    if (!le_ || le_[0] != '0') {
        const size_t lds = pl_lds_bytes();
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(pleft_persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return DPGP_ERR_LAUNCH;
        DPGP_PRELAUNCH();
        hipLaunchKernelGGL(pleft_persistent_kernel, dim3(B), dim3(256), lds, st, Mw, w, wstride, info);
        DPGP_LAUNCH_CHECK();
        return DPGP_OK;
    }
    const size_t lds = pp_lds_bytes();
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(pbig_persistent_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(pbig_persistent_kernel, dim3(B), dim3(256), lds, st, Mw, w, wstride, info);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
