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
#ifdef DPGP_PROFILE_CHAIN
// diagnostic build only (scratch/): per-phase clock stamps of workgroup 0, read back with dpgp_debug_stamps()
__device__ long long g_chain_stamps[128];
#define STAMP(i) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) { g_chain_stamps[i] = wall_clock64(); g_chain_stamps[8 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
extern "C" void dpgp_debug_stamps(long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_stamps), sizeof(long long) * 128); }
#define ACC_BEGIN() long long t__ = __builtin_amdgcn_s_memtime()
// slots 16 + k / 32 + k: shader cycles of panel / update phase k of the last evaluation (wave 0 / wave 1 of workgroup 0)
#define ACC_END(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == ((i) == 4 ? 0 : 1)) { const long long dt__ = __builtin_amdgcn_s_memtime() - t__; g_chain_stamps[i] += dt__; g_chain_stamps[((i) == 4 ? 16 : 32) + k] = dt__; } } while (0)
// slot 48 + k: cycles inside UpdItem::run of update phase k (wave 1), summed over its items
#define SUB_BEGIN() const long long ts__ = __builtin_amdgcn_s_memtime()
#define SUB_END(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == 1) g_chain_stamps[(i) + k] += __builtin_amdgcn_s_memtime() - ts__; } while (0)
#endif
#include "linalg_dev.h"
static_assert(TSZ == DPGP_LB_TILE_ELEMS, "lb_out image: tile size");

// ---------------------------------------------------------------------------------------------------------------
// Per-output workspace (elements of TL), see la_chain_ws_elems:
//   K0 [Mp x Mp]      : K_uu + jitter I as written by the gram kernel into [0,M)x[0,M)   (read by both chain kernels)
//   Kb [Mp x Mp]      : identity-padded copy -> L_uu                                       (chain_k)
//   Wb [(Mp+16) x Mp] : L_uu^-1 (chain_k); reused as B = K + beta Psi2 (+ border row) when B does not fit in LDS (chain_b)
//   KI [Mp x Mp]      : K_uu^-1, lower triangle                                            (chain_k -> chain_b)
//   dinv [Mp/16][256] : inverted diagonal tiles of L_uu
// ---------------------------------------------------------------------------------------------------------------
size_t la_chain_ws_elems(int M) { return la_chain_ws_elems_inline(M); }
bool la_chain_k_resident(int M, int elem) { return chain_k_resident(dpgp_round_up(M, 16), (size_t)elem); }

// ---- chain_k as a kernel of its own (plain cross-check path and callers without a psi2 launch) ----------------------
template <typename TL>
__global__ __launch_bounds__(256) void chain_k_kernel(int M, int Mp, TL *__restrict__ ws, size_t ws_stride,
                                                      double *__restrict__ logdet_k, int *__restrict__ info_k,
                                                      int plain) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    chain_k_body<TL, 1>(blockIdx.x, M, Mp, ws, ws_stride, logdet_k, info_k, plain, smem_raw);
}

// f_hat = sum of the per-output terms, KL from its partials, the model-level tail: one 256-thread workgroup, fixed order.
template <bool COHERENT = false>    // COHERENT: the terms were written by other workgroups of the SAME launch (agent-scope loads)
__device__ __forceinline__ void sum_terms_body(int D, const double *terms, const double *kl_part, double *sums,
                                               const double *model_scal, double *model_pack, double *model_out, double *scratch) {
    double a = 0.0;
    for (int i = threadIdx.x; i < D * 5; i += 256)
        a += COHERENT ? __hip_atomic_load(terms + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : terms[i];
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
// ---- chain_b: everything after Psi2 (dp_gp_lvm.py:118-145 in the B = K + beta Psi2 form) -----------------------------
// mode 0: B lives in LDS (potrf_lds); mode 1: B in global memory (Wb), blocked MFMA; mode 2: plain VALU cross-check.
// OCC = 2: bounded to 256 VGPRs so that two workgroups share a compute unit (D = 512: one round instead of two, 135 -> 71 us);
// launched whenever B is LDS-resident within 80 KB.  It is also the better compile for a single round (D = 64: 69 -> 62 us;
// the unbounded instantiation of the fp64 diagonal tile takes 328 registers AND spills more).  OCC = 1 carries the
// global-memory and plain fallbacks.
template <typename TP, typename TL, int OCC>
__global__ __launch_bounds__(256, OCC) void chain_b_kernel(int D, int N, int M, int Mp, const TP *__restrict__ psi2_part,
                                                      int ns2, const double *__restrict__ v_part, int ns1,
                                                      const double *__restrict__ alpha,
                                                      const double *__restrict__ beta,
                                                      const double *__restrict__ yy_part,
                                                      const double *__restrict__ logdet_k,
                                                      const int *__restrict__ info_k, double *__restrict__ terms,
                                                      int *__restrict__ info, double *__restrict__ guard,
                                                      TL *__restrict__ ws, size_t ws_stride, int mode,
                                                      const double *kl_part, double *sums, const double *model_scal,
                                                      double *model_pack, double *model_out, TL *lb_out) {
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
    double ip = 0.0, kin2 = 0.0, p2n2 = 0.0;     // <K^-1, Psi2>, |K^-1|_F^2, |Psi2|_F^2 (the last two: conditioning guard)
    const int ii = t >> 4, jj = t & 15;
    const int nlow = nb * (nb + 1) / 2;
    // the border vector v^T = sum of the ns1 partial Psi1^T y slabs: its loads are issued first and drain beneath the assembly
    // below (behind it they were ~2 us of exposed latency per evaluation); thread j < Mp holds entry j
    double vborder = 0.0;
    if (mode == 0 && t < M) {                              // (LDS-resident sizes: Mp <= 256, one entry per thread)
        for (int k0 = 0; k0 < ns1; k0 += 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (k0 + k < ns1) ? v_part[((size_t)(k0 + k) * D + d) * M + t] : 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) vborder += v[k];
        }
    }
    {
        // 4 tiles per pass: thread = (tile u, row r, 4 consecutive columns) -> 16/32-byte loads, all issued before use
        typedef TP tp4 __attribute__((ext_vector_type(4)));
        typedef TL tl4 __attribute__((ext_vector_type(4)));
        const int u = t >> 6, r = (t & 63) >> 2, c4 = (t & 3) * 4;
        // tile of pass t0 for this thread's quarter u, and what is done with its loaded values
        auto locate = [&](int t0, int &I, int &J) __attribute__((always_inline)) {
            const int tt = min(t0 + u, nlow - 1);
            I = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
            while ((I + 1) * (I + 2) / 2 <= tt) ++I;
            while (I * (I + 1) / 2 > tt) --I;
            J = tt - I * (I + 1) / 2;
        };
        auto finish = [&](int t0, int I, int J, const tl4 &k0, const tl4 &ki, const double (&p2)[4]) __attribute__((always_inline)) {
            if (t0 + u >= nlow) return;
            const int i = 16 * I + r, j = 16 * J + c4;
            // (the 2-per-CU instantiation is LDS-resident by construction: a destination that is provably not global
            //  memory lets the loads of the next passes be issued ahead of these stores)
            TL *dst = (OCC == 2 || mode == 0) ? tiles + lds_tile_index(I, J, nb) * TSZ + r * LDT + c4 : Wb + (size_t)i * Mp + j;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int je = j + e;
                TL bv;
                if (je <= i && i < M) {            // inside the real lower triangle
                    bv = k0[e] + be * (TL)p2[e];
                    const double wt = (i == je ? 1.0 : 2.0);
                    ip += (double)ki[e] * p2[e] * wt;
                    kin2 += (double)ki[e] * (double)ki[e] * wt;
                    p2n2 += p2[e] * p2[e] * wt;
                } else {
                    bv = (i == je) ? (TL)1 : (TL)0;   // identity padding (and don't-care zeros above the diagonal)
                }
                dst[e] = bv;
            }
        };
        constexpr int NB = sizeof(TP) == 4 ? 2 : 1;        // passes whose loads are in flight together (fp32 slabs: 48 registers each;
                                                           // three: 185 registers = two waves per SIMD, config 5's 560 workgroups then need two rounds)
        if (NB > 1 && ns2 <= 8) {
            // (the loop below left to the compiler waits for every pass's loads in turn — its tile search and the slab loop keep
            //  it from hoisting them: 9 passes x ~1.2 us at M = 128 on the critical path of every evaluation)
            for (int t0 = 0; t0 < nlow; t0 += 4 * NB) {
                int I[NB], J[NB];
                tl4 k0[NB], ki[NB];
                tp4 v[NB][8];
#pragma unroll
                for (int b_ = 0; b_ < NB; ++b_) {
                    locate(t0 + 4 * b_, I[b_], J[b_]);
                    const size_t off = (size_t)(16 * I[b_] + r) * Mp + 16 * J[b_] + c4;
                    k0[b_] = *reinterpret_cast<const tl4 *>(K0 + off);
                    ki[b_] = *reinterpret_cast<const tl4 *>(KI + off);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k < ns2) v[b_][k] = *reinterpret_cast<const tp4 *>(psi2_part + ((size_t)k * D + d) * (size_t)Mp * Mp + off);
                }
#pragma unroll
                for (int b_ = 0; b_ < NB; ++b_) {
                    double p2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (k < ns2) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) p2[e] += (double)v[b_][k][e];
                        }
                    finish(t0 + 4 * b_, I[b_], J[b_], k0[b_], ki[b_], p2);
                }
            }
        } else {
            for (int t0 = 0; t0 < nlow; t0 += 4) {
                int I, J;
                locate(t0, I, J);
                const size_t off = (size_t)(16 * I + r) * Mp + 16 * J + c4;
                tl4 k0 = *reinterpret_cast<const tl4 *>(K0 + off);
                tl4 ki = *reinterpret_cast<const tl4 *>(KI + off);
                double p2[4] = {0.0, 0.0, 0.0, 0.0};
                for (int kb = 0; kb < ns2; kb += 8) {      // up to 8 slab loads in flight
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
                finish(t0, I, J, k0, ki, p2);
            }
        }
    }
    // border tile-row: row 0 holds v^T, rows 1..15 are zero
    if (mode == 0) {
        if (t < Mp) tiles[(size_t)nlow * TSZ + (t >> 4) * LDT + (t & 15)] = (TL)vborder;     // border vector (linalg_dev.h)
    } else {
        for (int e = t; e < 16 * Mp; e += 256) {
            const int row = e / Mp, j = e - row * Mp;
            TL bv = 0;
            if (row == 0 && j < M) {
                double a = 0.0;
                for (int k0 = 0; k0 < ns1; k0 += 8) {
                    double v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (k0 + k < ns1) ? v_part[((size_t)(k0 + k) * D + d) * M + j] : 0.0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) a += v[k];
                }
                bv = (TL)a;
            }
            Wb[(size_t)(Mp + row) * Mp + j] = bv;
        }
    }
    ip = block_sum(ip, scratch);
    kin2 = block_sum(kin2, scratch);
    p2n2 = block_sum(p2n2, scratch);
    __syncthreads();
    STAMP(1);
    // ---- L_B = chol(B), border -> L_B^-1 v ----
    if (mode == 0) {
        potrf_lds<TL, OCC>(tiles, dinv, nb, nb + 1, &fail);
    } else if constexpr (OCC == 1) {        // (the 2-per-CU instantiation is only launched LDS-resident: keep the global-memory
        if (mode == 1) potrf_blocked<TL>(Wb, Mp, nb, nb + 1, dinv, (TL *)nullptr, &fail, 0);   // routines' registers out of it)
        else potrf_plain<TL>(Wb, Mp, Mp, Mp + 1, &fail, 0);
    }
    __syncthreads();
    STAMP(2);
    // the over-T model solves L_B against all D columns of Psi1^T Y afterwards (elbo_t.hip): it gets the factor as the LDS image
    // of its lower tiles [nb (nb + 1) / 2][16][LDT], verbatim
    if (lb_out && mode == 0)
        for (int e = t; e < nlow * TSZ; e += 256) lb_out[(size_t)d * nlow * TSZ + e] = tiles[e];
    double ld = 0.0, cc = 0.0;
    for (int i = t; i < M; i += 256) {
        const int I = i >> 4, r = i & 15;
        const double lii = mode == 0 ? (double)tiles[lds_tile_index(I, I, nb) * TSZ + r * LDT + r]
                                     : (double)Wb[(size_t)i * Mp + i];
        const double c = mode == 0 ? (double)tiles[(size_t)nlow * TSZ + I * LDT + r] : (double)Wb[(size_t)Mp * Mp + i];
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
        int f = fk ? fk : (fail ? M + fail : 0);
        // Conditioning guard.  Psi2 computed in fp32 carries a relative rounding error eps = 2^-23 (Frobenius norm); through
        // B = K + beta Psi2 it moves  -log det L_A + beta/2 <K^-1, Psi2>  by at most  beta |K^-1|_F eps |Psi2|_F  and
        // beta^2/2 v^T B^-1 v  by at most  beta^3/2 |K^-1| eps |Psi2| v^T B^-1 v   (B >= K, so |B^-1| <= |K^-1|).  The bound is
        // written to guard[d] in every precision mode; with an fp32 Psi2 (TP = float) an output dim whose bound exceeds
        // DPGP_GUARD_REL * N is reported as info = DPGP_INFO_ILL_CONDITIONED: its terms are still written, but they can no
        // longer be trusted to the mixed-precision tolerance — evaluate with DPGP_PREC_F64.  (Entries are uncorrelated
        // rounding errors, so the typical deviation is ~1/M of the bound.)
        const double errb = 1.1920928955078125e-07 * b_ * sqrt(kin2 * p2n2) * (1.0 + 0.5 * b_ * b_ * cc);
        if (guard) guard[d] = errb;
        if (sizeof(TP) == 4 && !f && !(errb <= DPGP_GUARD_REL * (double)N)) f = DPGP_INFO_ILL_CONDITIONED;
        const bool flagged_only = (f == DPGP_INFO_ILL_CONDITIONED);
        info[d] = f;
        if (flagged_only) f = 0;                               // the terms below stay numbers
        const double nan_ = __longlong_as_double(0x7ff8000000000000LL);
        const double o_[5] = {0.5 * N * (log(b_) - DPGP_LOG_2PI),
                              f ? nan_ : -(ld - logdet_k[d]),                 // -sum log diag L_A
                              f ? nan_ : 0.5 * b_ * (ip - a_ * N),            // tr(L^-1 Psi2 L^-T) = <K^-1, Psi2>
                              -0.5 * b_ * yy,
                              f ? nan_ : 0.5 * b_ * b_ * cc};
        // agent-scope (sc1) stores: the workgroup that finishes last — possibly on another XCD, whose L2 is not coherent with
        // this one's — reads them with agent-scope loads; a full release fence (L2 write-back) per workgroup cost 4-11 us
#pragma unroll
        for (int i = 0; i < 5; ++i) __hip_atomic_store(o + i, o_[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the final reduction, by the workgroup that finishes last (sums != nullptr: the fused ELBO; round 2 launched
    // sum_terms_kernel for it: 5-7 us of launch + latency on the critical path of every evaluation).  The counter sits behind
    // the KL partials and is zeroed by the front launch; the sum runs in the fixed order of sum_terms_body either way, so the
    // result does not depend on which workgroup is last. ----
    if (sums) {
        int *counter = reinterpret_cast<int *>(const_cast<double *>(kl_part) + DPGP_KL_NBLK);
        int &last = *reinterpret_cast<int *>(smem_raw + 68);
        if (t == 0) {
            // Order: terms (relaxed agent-scope stores, written through to memory: sc1) -> all of them acknowledged (s_waitcnt) ->
            // arrival counter.  The language's memory model has no edge between relaxed operations on different addresses, and the
            // waitcnt builtin is not a compiler barrier: the two signal fences pin the order for the COMPILER (no instruction), the
            // waitcnt for the hardware.  (A release fetch_add would do both but writes back the whole L2: 4-11 us per workgroup.)  The
            // last workgroup reads the terms with agent-scope loads (sum_terms_body<true>): they do not hit in its XCD's L2.
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            __builtin_amdgcn_s_waitcnt(0);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
            last = (__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == D - 1) ? 1 : 0;
            __atomic_signal_fence(__ATOMIC_SEQ_CST);
        }
        __syncthreads();
        if (last) sum_terms_body<true>(D, terms, kl_part, sums, model_scal, model_pack, model_out, scratch);
    }
}

static size_t chain_b_lds_bytes(int Mp, size_t elem) {   // LDS-resident B: dinv + lower triangle + border row
    const int nb = Mp / 16;
    return LA_LDS_HDR + elem * lds_chol_elems(nb, 1);
}

template <typename TL>
int launch_chain_k(int D, int M, TL *ws, double *logdet_k, int *info_k, int algo, hipStream_t st) {
    const int Mp = dpgp_round_up(M, 16);
    size_t lds = chain_k_lds_bytes(Mp, sizeof(TL));
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
                   const int *info_k, double *terms, int *info, double *guard, TL *ws, int algo, hipStream_t st,
                   const double *kl_part, double *sums, const double *model_scal, double *model_pack, double *model_out,
                   TL *lb_out) {
    const int Mp = dpgp_round_up(M, 16);
    int mode = 2;
    size_t lds = la_lds_bytes(Mp, sizeof(TL));
    if (algo != DPGP_ALGO_PLAIN) {
        const size_t need = chain_b_lds_bytes(Mp, sizeof(TL));
        mode = need <= LA_LDS_LIMIT ? 0 : 1;
        if (mode == 0) lds = need;
    }
    if (lb_out && mode != 0) return -30;                       // (the factor is only exported from the LDS-resident form)
    const bool two = (mode == 0) && lds <= 80 * 1024;
    auto kern = two ? chain_b_kernel<TP, TL, 2> : chain_b_kernel<TP, TL, 1>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3(D), dim3(256), lds, st, D, N, M, Mp, psi2_part, ns2, v_part, ns1, alpha, beta, yy_part,
                       logdet_k, info_k, terms, info, guard, ws, la_chain_ws_elems(M), mode, kl_part, sums, model_scal, model_pack,
                       model_out, lb_out);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
#ifdef DPGP_PROFILE_CHAIN
extern "C" int dpgp_debug_chain_b_occupancy(int Mp, int extra) {
    int nb = 0;
    const size_t lds = chain_b_lds_bytes(Mp, 8) + extra;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(chain_b_kernel<float, double, 2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, chain_b_kernel<float, double, 2>, 256, lds);
    return nb * 1000000 + (int)lds;
}
#endif
#define INST_CHAIN_B(TP, TL)                                                                                        \
    template int launch_chain_b<TP, TL>(int, int, int, const TP *, int, const double *, int, const double *,      \
                                        const double *, const double *, const double *, const int *, double *, int *, \
                                        double *, TL *, int, hipStream_t, const double *, double *, const double *, double *, \
                                        double *, TL *);
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
    sum_terms_body(D, terms, kl_part, sums, model_scal, model_pack, model_out, scratch);
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

// LDS-resident form of the stand-alone batched potrf (matrices whose lower triangle fits 80 KB: M <= 128 in fp64): tiles
// in, potrf_lds, tiles out; no workspace traffic, two workgroups per compute unit.
template <typename T>
__global__ __launch_bounds__(256, 2) void potrf_batched_lds_kernel(int M, int Mp, T *__restrict__ a, int *__restrict__ info) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    int &fail = *reinterpret_cast<int *>(smem_raw + 64);
    T *dinv = reinterpret_cast<T *>(smem_raw + LA_LDS_HDR), *tiles = dinv + TSZ;
    const int b = blockIdx.x, t = threadIdx.x, nb = Mp / 16, nlow = nb * (nb + 1) / 2;
    T *A = a + (size_t)b * M * M;
    if (t == 0) fail = 0;
    STAMP(60);
    if ((M & 1) == 0 && sizeof(T) == 8) {
        // even M: 16-byte loads along the rows, four in flight per thread; pairs in tiles above the diagonal are skipped
        typedef T t2 __attribute__((ext_vector_type(2)));
        const int hp = Mp / 2, total = Mp * hp;
        for (int e0 = t; e0 < total; e0 += 256 * 4) {
            t2 v[4];
            int dst[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + 256 * u, i = e / hp, j = 2 * (e - i * hp);
                const bool want = e < total && (j >> 4) <= (i >> 4);
                dst[u] = want ? lds_tile_index(i >> 4, j >> 4, nb) * TSZ + (i & 15) * LDT + (j & 15) : -1;
                v[u] = (t2){(i == j) ? (T)1 : (T)0, (i == j + 1) ? (T)1 : (T)0};      // identity padding
                if (want && i < M && j < M) v[u] = *reinterpret_cast<const t2 *>(A + (size_t)i * M + j);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u] >= 0) {
                    tiles[dst[u]] = v[u][0];
                    tiles[dst[u] + 1] = v[u][1];
                }
        }
    } else {
        for (int e = t; e < nlow * 256; e += 256) {           // (tile, row, column): consecutive threads along a row
            const int tt = e >> 8, r = (e >> 4) & 15, c = e & 15;
            int I = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
            while ((I + 1) * (I + 2) / 2 <= tt) ++I;
            while (I * (I + 1) / 2 > tt) --I;
            const int J = tt - I * (I + 1) / 2, i = 16 * I + r, j = 16 * J + c;
            tiles[tt * TSZ + r * LDT + c] = (i < M && j < M) ? A[(size_t)i * M + j] : ((i == j) ? (T)1 : (T)0);
        }
    }
    __syncthreads();
    STAMP(61);
    potrf_lds<T, 2>(tiles, dinv, nb, nb, &fail);
    __syncthreads();
    STAMP(62);
    if ((M & 1) == 0 && sizeof(T) == 8) {
        typedef T t2 __attribute__((ext_vector_type(2)));
        const int hm = M / 2;
        for (int e = t; e < M * hm; e += 256) {               // 16-byte stores
            const int i = e / hm, j = 2 * (e - i * hm);
            const T *src = tiles + lds_tile_index(i >> 4, j >> 4, nb) * TSZ + (i & 15) * LDT + (j & 15);
            const t2 v = {(j <= i) ? src[0] : (T)0, (j + 1 <= i) ? src[1] : (T)0};
            *reinterpret_cast<t2 *>(A + (size_t)i * M + j) = v;
        }
    } else {
        for (int e = t; e < M * M; e += 256) {
            const int i = e / M, j = e - i * M;
            A[e] = (j <= i) ? tiles[lds_tile_index(i >> 4, j >> 4, nb) * TSZ + (i & 15) * LDT + (j & 15)] : (T)0;
        }
    }
    STAMP(63);
    if (t == 0) info[b] = fail;
}

// The right-hand sides are independent column by column: workgroup = (batch element, chunk of TRSM_KC columns), so that a
// few matrices with many right-hand sides (the over-T model: T = 8 matrices, D = 512 columns; stage A of the backward pass
// for M > 128: the identity) still fill the GPU.  K = leading dimension of rhs, kc0 .. kc0 + kw = this chunk.
#define TRSM_KC 64
template <typename T>
__global__ __launch_bounds__(256) void trsm_batched_kernel(int M, int K, int Mp, int Kcp, int nchunk, const T *__restrict__ l,
                                                           T *__restrict__ rhs, T *__restrict__ ws, int plain) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *lds = reinterpret_cast<T *>(smem_raw + LA_LDS_HDR);
    const int b = blockIdx.x / nchunk, ch = blockIdx.x - b * nchunk, t = threadIdx.x, nb = Mp / 16;
    const int kc0 = ch * TRSM_KC, kw = min(nchunk == 1 ? K : TRSM_KC, K - kc0), Kp = 16 * ((kw + 15) / 16);
    const T *Lg = l + (size_t)b * M * M;
    T *R = rhs + (size_t)b * M * K + kc0;
    T *Lw = ws + (size_t)blockIdx.x * ((size_t)Mp * Mp + (size_t)Mp * Kcp + (size_t)nb * 256), *Rw = Lw + (size_t)Mp * Mp,
      *dinv = Rw + (size_t)Mp * Kcp;
    for (int e = t; e < Mp * Mp; e += 256) {
        const int i = e / Mp, j = e - i * Mp;
        Lw[e] = (i < M && j < M) ? ((j <= i) ? Lg[(size_t)i * M + j] : (T)0) : ((i == j) ? (T)1 : (T)0);
    }
    for (int e = t; e < Mp * Kp; e += 256) {
        const int i = e / Kp, j = e - i * Kp;
        Rw[e] = (i < M && j < kw) ? R[(size_t)i * K + j] : (T)0;
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
    for (int e = t; e < M * kw; e += 256) {
        const int i = e / kw, j = e - i * kw;
        R[(size_t)i * K + j] = Rw[(size_t)i * Kp + j];
    }
}

extern "C" size_t dpgp_potrf_workspace_bytes(int B, int M, int elem_size) {
    if (B <= 0 || M <= 0) return 0;
    const int Mp = dpgp_round_up(M, 16);
    size_t elems = (size_t)B * Mp * Mp;                                         // single-workgroup routine
    if (potrf_big_ws_elems(B, M) > elems) elems = potrf_big_ws_elems(B, M);     // multi-workgroup routine (large M)
    return dpgp_align256((size_t)elem_size * elems);
}
extern "C" size_t dpgp_trsm_workspace_bytes(int B, int M, int K, int elem_size) {
    if (B <= 0 || M <= 0 || K <= 0) return 0;
    const int Mp = dpgp_round_up(M, 16);
    const int nchunk = K > TRSM_KC ? dpgp_ceil_div(K, TRSM_KC) : 1, Kcp = nchunk == 1 ? dpgp_round_up(K, 16) : TRSM_KC;
    return dpgp_align256((size_t)elem_size * B * nchunk * ((size_t)Mp * Mp + (size_t)Mp * Kcp + (size_t)(Mp / 16) * 256));
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
    if (algo != DPGP_ALGO_PLAIN && chain_k_resident(Mp, sizeof(T))) {
        const size_t lds = chain_k_resident_bytes(Mp, sizeof(T));
        auto kern = potrf_batched_lds_kernel<T>;
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
                hipSuccess)
            return DPGP_ERR_LAUNCH;
        DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, (hipStream_t)stream, M, Mp, a, info);
        DPGP_LAUNCH_CHECK();
        return DPGP_OK;
    }
    if (algo != DPGP_ALGO_PLAIN) return launch_potrf_big<T>(B, M, a, info, (T *)ws, (hipStream_t)stream);
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
    const int Mp = dpgp_round_up(M, 16);
    const int nchunk = K > TRSM_KC ? dpgp_ceil_div(K, TRSM_KC) : 1, Kcp = nchunk == 1 ? dpgp_round_up(K, 16) : TRSM_KC;
    size_t tiles = (size_t)(Kcp / 16 + 1);
    if (tiles < 4) tiles = 4;   // the diagonal-tile inversion uses one dinv slot per wave
    size_t lds = LA_LDS_HDR + sizeof(T) * (size_t)(16 * LDT) * tiles;
    auto kern = trsm_batched_kernel<T>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3((unsigned)((size_t)B * nchunk)), dim3(256), lds, (hipStream_t)stream, M, K, Mp, Kcp,
                       nchunk, l, rhs, (T *)ws, algo == DPGP_ALGO_PLAIN ? 1 : 0);
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
