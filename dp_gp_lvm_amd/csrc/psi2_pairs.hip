// K3, pair-tile form: the Psi2 statistic (reference: /root/reference/src/kernels/rbf_kernel.py:164-199) as ONE plain GEMM
// per output dim between observations and PAIRS of inducing points, followed by exp2 and a column sum.
//
// With den = 2 g s_n + 1, w = g / den, z and mu centred by the column mean c of z (mu' = mu - c, the statistic is translation
// invariant) and s_p = z_m + z_m' for the pair p = (m, m'), m' <= m, SURVEY.md Appendix A reads in log2 units
//      log2 psi2[n, p] = 2 log2 alpha + beta_p + E[n, p],
//      E[n, p] = c''_n + sum_q ( a_nq s_pq^2 + b_nq s_pq ),   a = -1/4 w log2e,  b = w mu' log2e,
//      c''_n   = - sum_q ( w mu'^2 log2e + 1/2 log2 den ),     beta_p = -1/4 log2e sum_q g_q (z_mq - z_m'q)^2
// so E = A B^T with A[n, :] = (a_n., b_n., c''_n) (depends on the output dim and the observation) and B[p, :] = (s_p.^2, s_p.,
// 1) (depends on the inducing inputs only: built once per evaluation for ALL output dims, psi2_consts.h), K = 2Q + 1.  Both
// operands are split into f16 (hi, lo) pairs once, where they are built — a_h f_h + a_h f_l + a_l f_h, three K slots per
// product, fp32 accumulation in v_mfma_f32_32x32x16_f16 — so the hot loop holds NO operand arithmetic at all:
//      per 32 observations x 32 pairs:  KS MFMAs,  16 v_exp_f32 and 16 adds per lane.
// The earlier patch kernel (psi2.hip, psi2_patch_f16p: A_n[m, :] B_n[:, m'] per observation, K = Q + 2) had to form and
// split the products X[n,q] z[m,q] for every observation and patch row (30 of ~300 instructions per 1024 exponentials) and
// the P[n, m] rows (two more phases), and computed full 64 x 64 patches on the diagonal; here the pairs are exactly the
// M (M + 1) / 2 lower-triangle entries.
//
// Decomposition: workgroup = (output dim b, n-split sp, tile range rg), 4 autonomous waves.  Per chunk of <= R observations
// the workgroup builds the A image in LDS (thread = observation); each wave then owns every 4th pair tile of the range, keeps
// the B operands of G tiles in registers, and streams the chunk's 32-row tiles through them: MFMA chain of the next tile
// issued before the exponentials of the current one (2 waves per SIMD do not hide the matrix-pipe latency by switching).
// Per (tile, chunk) the two lane halves are added and alpha^2 exp2(beta_p) * sum is stored (first chunk) or added (later
// chunks: the owner is the only writer) at [m][m'] of the partial slab part[sp][b][Mp][Mp] — the layout the consumers
// (chain_b_kernel, chain_grad_kernel, psi2_finish_kernel) already read; only entries m' <= m < M are written and read.
// The K_uu branch of the ELBO rides in the same dispatch (psi2_chain_task.h).
#include <type_traits>
#include "internal.h"
#include "psi2_consts.h"
#include "psi2_chain_task.h"

typedef _Float16 pp_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 pp_h2 __attribute__((ext_vector_type(2)));
typedef float pp_f16v __attribute__((ext_vector_type(16)));

#define PP_APAD 8                  // f16 padding of an A-image row in LDS: row stride 16 KS + 8 halves, conflict-free b128 reads
#define PP_LDS_HDR 512             // gamma_b, column means, flags

// resident pair tiles per wave by K-steps (register budget: 4 KS (G + 2) operand registers + 32 result + 4 G accumulators
// within 256 without spills), even
#ifndef PP_G
#define PP_G(KS) ((KS) <= 4 ? 6 : ((KS) <= 6 ? 4 : 2))
#endif
#ifndef PP_WAVES
#define PP_WAVES 2                 // waves per SIMD the register allocation must allow
#endif
template <int KS> struct PairsCfg { static constexpr int G = PP_G(KS); };

#ifndef PP_NW8_MIN_TILES
#define PP_NW8_MIN_TILES 512       // pair tiles from which the eight-wave workgroup is used (see psi2_pairs_kernel, pairs_geom)
#endif
#ifndef PP_DEEP_PREFETCH
#define PP_DEEP_PREFETCH 1         // groups of two resident tiles fetch their row operands two row tiles ahead
#endif
#ifndef PP_SCALAR_ADD
#define PP_SCALAR_ADD 1            // 1: the 16 accumulations per tile as v_add_f32 (inline asm keeps the compiler from packing
#endif                             //    them into v_pk_add_f32, which costs more issue time beside MFMAs: MI355X guide)

// the column operands of GG pair tiles tb, tb + 4, ...: operand order (psi2_consts.h), 64 lanes x 16 bytes contiguous per
// (tile, K-step)
template <int KS, int GG, int NW>
__device__ __forceinline__ void pairs_load(pp_h8 (&bop)[GG][KS], int tb, const _Float16 *__restrict__ img, int l5, int half) {
#pragma unroll
    for (int g = 0; g < GG; ++g) {
        const pp_h8 *row = reinterpret_cast<const pp_h8 *>(img) + (size_t)(tb + NW * g) * KS * 64 + 32 * half + l5;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bop[g][ks] = row[ks * 64];
    }
}

// GG pair tiles tb, tb + 4, ... of one wave against the ntile row tiles of the A image in LDS.  The GG column operands stay in
// registers; the exponent tiles alternate between two result registers sets (c0, c1): the MFMA chain of the next tile is
// issued before the exponentials of the current one are evaluated.  GG is even or 1.
// bop: the operands of this group (pairs_load); tb_next >= 0: the operands of the group at tb_next are fetched into bop as soon
// as the last MFMA of this group has been issued, i.e. beneath this group's epilogue.
template <int KS, int GG, int NW>
__device__ __forceinline__ void pairs_group(int tb, pp_h8 (&bop)[GG][KS], int tb_next, const _Float16 *__restrict__ aimg,
                                            const _Float16 *__restrict__ img,
                                            const unsigned *__restrict__ pmap, const float *__restrict__ scale, int ntile,
                                            int l5, int half, float poison, float *__restrict__ out, int Mp, int chunk) {
    constexpr int SLP = 16 * KS, LDA = SLP + PP_APAD;
    static_assert(GG == 1 || (GG & 1) == 0, "ping-pong needs an even number of resident tiles");
    // the epilogue's per-pair data (index map, alpha^2 exp2(beta_p) of this output dim, the sum of the earlier chunks): fetched
    // now, used after the row loop
    unsigned pm[GG];
    float sc[GG], old[GG];
#pragma unroll
    for (int g = 0; g < GG; ++g) {
        pm[g] = pmap[32 * (tb + NW * g) + l5];
        sc[g] = scale[32 * (tb + NW * g) + l5];
    }
#pragma unroll
    for (int g = 0; g < GG; ++g) {
        old[g] = 0.0f;
        if (chunk && pm[g] != 0xffffffffu && half == 0) old[g] = out[(size_t)(pm[g] >> 16) * Mp + (pm[g] & 0xffffu)];
    }
    float acc[GG][4];
#pragma unroll
    for (int g = 0; g < GG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[g][i] = 0.0f;
    const _Float16 *arow = aimg + (size_t)l5 * LDA + 8 * half;
    auto load_a = [&](pp_h8 (&a)[KS], int nt) __attribute__((always_inline)) {
#ifdef PP_DIAG_NO_ALOAD                // (timing experiments only: wrong results)
        if (nt > 1) return;
#endif
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = *reinterpret_cast<const pp_h8 *>(arow + (size_t)nt * 32 * LDA + 16 * ks);
    };
    auto mma = [&](pp_f16v &c, const pp_h8 (&a)[KS], const pp_h8 (&bq)[KS]) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < 16; ++v) c[v] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], bq[ks], c, 0, 0, 0);
    };
    // One pipeline stage: the KS dependent MFMAs of the NEXT tile's exponent chain, each followed by its share of the 16
    // exponentials + accumulations of the CURRENT tile: a wave never waits for its own chain (the next MFMA's accumulator is
    // ready by the time its share of VALU work has been issued) and the matrix pipe runs under the VALU work.
    auto stage = [&](pp_f16v &cn, const pp_h8 (&a)[KS], const pp_h8 (&bq)[KS], float (&ac)[4], const pp_f16v &cu)
                     __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < 16; ++v) cn[v] = 0.0f;
#ifndef PP_MFMA_PER_STEP
#define PP_MFMA_PER_STEP 1         // MFMAs issued back to back between two shares of the exponentials (experiments)
#endif
#pragma unroll
        for (int ks = 0; ks < KS; ks += PP_MFMA_PER_STEP) {
#ifndef PP_DIAG_NO_MFMA            // (timing experiments only: wrong results)
#pragma unroll
            for (int k2 = ks; k2 < ks + PP_MFMA_PER_STEP && k2 < KS; ++k2)
                cn = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[k2], bq[k2], cn, 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            const int kend = (ks + PP_MFMA_PER_STEP < KS) ? ks + PP_MFMA_PER_STEP : KS;
            const int v0 = (16 * ks) / KS, v1 = (16 * kend) / KS;
#pragma unroll
            for (int v = v0; v < v1; ++v) {
#ifdef PP_DIAG_NO_EXP              // (timing experiments only: wrong results)
                const float e = cu[v];
#else
                const float e = __builtin_amdgcn_exp2f(cu[v]);
#endif
#if PP_SCALAR_ADD
                asm("v_add_f32 %0, %0, %1" : "+v"(ac[v & 3]) : "v"(e));
#else
                ac[v & 3] += e;
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    pp_f16v c0, c1;
    if constexpr (GG == 2 && PP_DEEP_PREFETCH) {
        // Two resident tiles (Q > 16): the row tile fetched at the top of a step is needed by the step's SECOND stage already,
        // ~250 cycles later, less than the latency of 8 ds_read_b128 on a busy LDS.  Three operand buffers, fetched two row
        // tiles ahead, rotated by unrolling the loop three times (no register moves).
        pp_h8 a0[KS], a1[KS], a2[KS];
        load_a(a0, 0);
        load_a(a1, min(1, ntile - 1));
        mma(c0, a0, bop[0]);
#pragma unroll 1
        for (int nt = 0; nt < ntile; nt += 3) {
            load_a(a2, min(nt + 2, ntile - 1));
            stage(c1, a0, bop[1], acc[0], c0);
            stage(c0, a1, bop[0], acc[1], c1);                     // (behind the last row tile: one surplus MFMA chain)
            if (nt + 1 >= ntile) break;
            load_a(a0, min(nt + 3, ntile - 1));
            stage(c1, a1, bop[1], acc[0], c0);
            stage(c0, a2, bop[0], acc[1], c1);
            if (nt + 2 >= ntile) break;
            load_a(a1, min(nt + 4, ntile - 1));
            stage(c1, a2, bop[1], acc[0], c0);
            stage(c0, a0, bop[0], acc[1], c1);
        }
    } else {
        // (the row loop is unrolled twice so that the two operand buffers swap roles without register moves: the 16 v_mov of
        //  a_cur = a_nxt were 7 % of the vector issue time of a row tile at G = 6)
        pp_h8 a_0[KS], a_1[KS];
        load_a(a_0, 0);
        mma(c0, a_0, bop[0]);
        auto row = [&](const pp_h8 (&a_cur)[KS], const pp_h8 (&a_nxt)[KS]) __attribute__((always_inline)) {
            if constexpr (GG == 1) {
                stage(c1, a_nxt, bop[0], acc[0], c0);              // (behind the last row tile: one surplus MFMA chain)
                c0 = c1;
            } else {
#pragma unroll
                for (int g = 0; g < GG; g += 2) {
                    stage(c1, a_cur, bop[g + 1], acc[g], c0);
                    if (g + 2 < GG) stage(c0, a_cur, bop[g + 2], acc[g + 1], c1);
                    else stage(c0, a_nxt, bop[0], acc[g + 1], c1);
                }
            }
        };
#pragma unroll 1
        for (int nt = 0; nt < ntile; nt += 2) {
            load_a(a_1, min(nt + 1, ntile - 1));
            row(a_0, a_1);
            if (nt + 1 >= ntile) break;
            load_a(a_0, min(nt + 2, ntile - 1));
            row(a_1, a_0);
        }
    }
    if (tb_next >= 0) pairs_load<KS, GG, NW>(bop, tb_next, img, l5, half);
    // ---- column sums: add the lane halves, scale by alpha^2 exp2(beta_p), store (first chunk) or accumulate ----
#pragma unroll
    for (int g = 0; g < GG; ++g) {
        float tot = (acc[g][0] + acc[g][1]) + (acc[g][2] + acc[g][3]);
        tot += __shfl_xor(tot, 32, 64);
        if (pm[g] != 0xffffffffu && half == 0)                      // (poison: 0, or NaN after a range-guard hit)
            out[(size_t)(pm[g] >> 16) * Mp + (pm[g] & 0xffffu)] = old[g] + (sc[g] * tot + poison);
    }
}

template <typename TIN>
__global__ __launch_bounds__(256) void psi2_pair_scale_kernel(int M, int Q, const TIN *__restrict__ z,
                                                              const TIN *__restrict__ gamma, const TIN *__restrict__ alpha,
                                                              float *__restrict__ scale) {
    psi2_pair_scale_block<TIN, TIN>((int)blockIdx.y, (int)blockIdx.x, M, Q, z, gamma, alpha, scale);
}

// NW waves per workgroup: 4 (two workgroups per compute unit, 80 KB of LDS each) or 8 (one workgroup with the whole LDS: twice
// the observations per chunk behind the same two waves per SIMD — the column operands of a group of G pair tiles are fetched once
// per chunk, so at KS = 8 (288 rows per 80 KB, G = 2) their traffic and the group prologue/epilogue set the pace: measured at
// config 4, 24.2 / 19.9 ms for 144 / 288 rows per chunk)
template <typename TIN, int KS, int NW>
__global__ __launch_bounds__(64 * NW, PP_WAVES) void psi2_pairs_kernel(int N, int M, int Q, int B, const unsigned char *__restrict__ consts,
                                                            const TIN *__restrict__ mu, const TIN *__restrict__ s,
                                                            const TIN *__restrict__ gamma, const float *__restrict__ scale,
                                                            float *__restrict__ part, int Mp, int n_per_split, int n_splits,
                                                            int n_ranges, int tiles_per_range, int R, ChainKTask task) {
    constexpr int G = PairsCfg<KS>::G, SLP = 16 * KS, LDA = SLP + PP_APAD;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    int item;
    if (psi2_task_1d(blockIdx.x, task.ws ? B : 0, task.last != 0, item)) {
#ifndef PP_NO_CHAIN
        if constexpr (std::is_same<TIN, double>::value && NW == 4) chain_k_task<2>(task, item, smem_raw);
#endif
        return;
    }
    const int b = item % B, sp = (item / B) % n_splits, rg = item / (B * n_splits);
    float *gq = reinterpret_cast<float *>(smem_raw);            // [32] gamma_b
    float *zc = gq + 32;                                        // [32] column means of z
    _Float16 *aimg = reinterpret_cast<_Float16 *>(smem_raw + PP_LDS_HDR);    // [R][LDA]
    const Psi2Consts C = psi2_consts_layout(M, Q);
    const _Float16 *img = reinterpret_cast<const _Float16 *>(consts + C.off_pairs);
    const unsigned *pmap = reinterpret_cast<const unsigned *>(consts + C.off_pmap);
    const float *sc_b = scale + (size_t)b * C.Ppad;
    const int t = threadIdx.x, lane = t & 63, l5 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);      // (scalar: the tile loops are wave-uniform)
    if (t < 32) {
        gq[t] = (t < Q) ? (float)gamma[(size_t)b * Q + t] : 0.0f;
        zc[t] = reinterpret_cast<const float *>(consts)[t];
    }
    const int tiles_total = C.Ppad / 32;
    const int t0 = rg * tiles_per_range, t1 = min(tiles_total, t0 + tiles_per_range);
    const int nbeg = sp * n_per_split, nend = min(N, nbeg + n_per_split);
    float poison = 0.0f;
    float *out = part + ((size_t)sp * B + b) * (size_t)Mp * Mp;
    bool oor = false;
    __syncthreads();

    for (int n0 = nbeg, chunk = 0; n0 < nend; n0 += R, ++chunk) {
        const int rows = min(R, nend - n0), ntile = (rows + 31) >> 5;
        if (chunk) __syncthreads();                              // the previous chunk's image is no longer read
        // ---- phase A: the A image of this chunk, thread = observation ----
        for (int r = t; r < 32 * ntile; r += 64 * NW) {
            const int n = n0 + r;
            unsigned *dst = reinterpret_cast<unsigned *>(aimg + (size_t)r * LDA);
            float cc = -60000.0f;                                // rows past the end: exp2(-60000) = 0
            if (n < nend) {
                cc = 0.0f;
                for (int q = 0; q < Q; ++q) {
                    const float g = gq[q], sv = (float)s[(size_t)n * Q + q], mc = (float)mu[(size_t)n * Q + q] - zc[q];
                    const float den = 2.0f * g * sv + 1.0f, w = g / den;
                    // (a carries the 64 that the s^2 features were divided by: psi2_consts.h, PSI2_PAIR_S2_SCALE)
                    const float a = dpgp_pin((float)(-0.25 * DPGP_LOG2E / PSI2_PAIR_S2_SCALE) * w);
                    const float bb = dpgp_pin((float)DPGP_LOG2E * w * mc);              // (pinned: see dpgp_pin)
                    cc -= bb * mc + 0.5f * __builtin_amdgcn_logf(den);          // (v_log_f32 = log2, den >= 1)
                    const _Float16 ah = (_Float16)a, alo = (_Float16)(a - (float)ah);
                    const _Float16 bh = (_Float16)bb, blo = (_Float16)(bb - (float)bh);
                    const pp_h2 w0 = {ah, ah}, w1 = {alo, bh}, w2 = {bh, blo};  // slots {ah, ah, al | bh, bh, bl}
                    dst[3 * q] = __builtin_bit_cast(unsigned, w0);
                    dst[3 * q + 1] = __builtin_bit_cast(unsigned, w1);
                    dst[3 * q + 2] = __builtin_bit_cast(unsigned, w2);
                }
                // Range guard.  The terms of the exponent cancel (E = c'' + sum a s^2 + b s is a sum of squares in disguise) and
                // are as large as |c''| ~ (|mu - c| / length scale)^2: with 22-bit operands and fp32 accumulation the exponent
                // of a row is good to ~|c''| 2^-21, i.e. the row's terms to 0.4 % at |c''| = 8192 (|mu - c| ~ 50 length scales
                // in every latent dim).  Beyond that the row is DETECTED and this workgroup's results become NaN — never a
                // silently wrong Psi2 (DPGP_ALGO_MFMA_F32 and fp64 have no such limit).
                oor |= !(cc >= -8192.0f);
                cc = fmaxf(cc, -60000.0f);
            } else {
                for (int q = 0; q < 3 * Q; ++q) dst[q] = 0u;
            }
            // c'' is the largest number in the exponent and common to a whole row: three f16 pieces (33 bits) where a K slot
            // is free (always, except Q = 5), so that its rounding does not show in every term of the observation
            cc = dpgp_pin(cc);
            const _Float16 ch = (_Float16)cc;
            const float r1 = dpgp_pin(cc - (float)ch);
            const _Float16 cm = (_Float16)r1;
            const pp_h2 cw = {ch, cm}, cw2 = {(_Float16)(r1 - (float)cm), (_Float16)0.0f};
            dst[3 * Q] = __builtin_bit_cast(unsigned, cw);
            for (int k = 3 * Q + 1; k < SLP / 2; ++k) dst[k] = 0u;
            if (6 * Q + 2 < SLP) dst[3 * Q + 1] = __builtin_bit_cast(unsigned, cw2);
        }
        if (__syncthreads_or(oor ? 1 : 0)) poison = __builtin_nanf("");

        // ---- the wave's pair tiles: groups of G, the remainder in groups of 2 and 1 ----
        int tb = t0 + wv;
        if (tb + NW * (G - 1) < t1) {
            pp_h8 bop[G][KS];
            pairs_load<KS, G, NW>(bop, tb, img, l5, half);
            for (; tb + NW * (G - 1) < t1; tb += NW * G) {
                const int nx = tb + NW * G;
                pairs_group<KS, G, NW>(tb, bop, nx + NW * (G - 1) < t1 ? nx : -1, aimg, img, pmap, sc_b, ntile, l5, half, poison, out,
                                   Mp, chunk);
            }
        }
        if constexpr (G > 2)
            for (; tb + NW < t1; tb += 2 * NW) {
                pp_h8 bop[2][KS];
                pairs_load<KS, 2, NW>(bop, tb, img, l5, half);
                pairs_group<KS, 2, NW>(tb, bop, -1, aimg, img, pmap, sc_b, ntile, l5, half, poison, out, Mp, chunk);
            }
        for (; tb < t1; tb += NW) {
            pp_h8 bop[1][KS];
            pairs_load<KS, 1, NW>(bop, tb, img, l5, half);
            pairs_group<KS, 1, NW>(tb, bop, -1, aimg, img, pmap, sc_b, ntile, l5, half, poison, out, Mp, chunk);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct PairsGeom {
    int KS, R, n_ranges, tiles_per_range, NW;
    size_t lds;
};
static PairsGeom pairs_geom(int B, int N, int M, int Q, int ns, size_t chain_lds) {
    const Psi2Consts C = psi2_consts_layout(M, Q);
    PairsGeom g;
    g.KS = C.KS;
    const size_t row = sizeof(_Float16) * (size_t)(16 * C.KS + PP_APAD);
    // two workgroups per compute unit: 80 KB each (the K_uu task slice of M = 128 in fp64 needs 79 KB of it anyway)
    size_t budget = 80 * 1024;
    if (const char *e = getenv("DPGP_PP_LDS_KB")) budget = (size_t)atoi(e) * 1024;   // (experiments only)
    if (chain_lds > budget) budget = chain_lds;
    int rmax = (int)((budget - PP_LDS_HDR) / row) & ~31;
    const int nper = dpgp_ceil_div(N, ns);
    // eight waves on the whole LDS (psi2_pairs_kernel) when the observations of a workgroup do not fit one 80 KB chunk, no
    // K_uu task rides in the dispatch (those are 256-thread workgroups) and every wave has enough pair tiles to pay for the
    // image build of the larger chunk (operator alone, 4 against 8 waves: config 4 (4104 tiles, KS = 8) 19.96 -> 18.24 ms;
    // config 3 (258 tiles, KS = 4) 1.467 -> 1.443 ms; config 5 (65 tiles, KS = 8) 0.79 -> 0.89 ms)
    g.NW = (!chain_lds && nper > rmax && C.Ppad / 32 >= PP_NW8_MIN_TILES) ? 8 : 4;
    if (const char *e = getenv("DPGP_PP_NW")) {                 // (experiments only)
        const int v = atoi(e);
        if ((v == 4 || v == 8) && !(v == 8 && chain_lds)) g.NW = v;
    }
    if (g.NW == 8) {
        budget = 160 * 1024;
        rmax = (int)((budget - PP_LDS_HDR) / row) & ~31;
    }
    int r = dpgp_round_up(nper, 32);
    if (r > rmax) {                                             // several chunks per workgroup: equal ones
        const int chunks = dpgp_ceil_div(nper, rmax);
        r = dpgp_round_up(dpgp_ceil_div(nper, chunks), 32);
    }
    g.R = r;
    g.lds = PP_LDS_HDR + row * (size_t)r;
    if (chain_lds > g.lds) g.lds = chain_lds;
    // tile ranges: only when the (output dim, n-split) workgroups alone cannot fill the GPU (small batches)
    const int tiles = C.Ppad / 32;
    // (with the K_uu tasks in the dispatch: one round of <= 512 workgroups, less the B slots the tasks hold from the start
    //  when there are few output dims — the rule psi2_nsplit chose ns by)
    int nr = chain_lds ? (B < 128 ? 512 - B : 512) / (B * ns) : dpgp_ceil_div(g.NW == 8 ? 256 : 512, B * ns);
    if (nr > dpgp_ceil_div(tiles, 32)) nr = dpgp_ceil_div(tiles, 32);   // >= 32 tiles (8 per wave) per range
    if (nr < 1) nr = 1;
    if (const char *e = getenv("DPGP_PP_RANGES")) {             // (experiments only)
        const int v = atoi(e);
        if (v >= 1 && v <= tiles) nr = v;
    }
    g.tiles_per_range = dpgp_ceil_div(tiles, nr);
    g.n_ranges = dpgp_ceil_div(tiles, g.tiles_per_range);
    return g;
}

template <typename TIN, int KS, int NW>
static int launch_pairs_ks(int B, int N, int M, int Q, const TIN *mu, const TIN *s, const TIN *gamma, const float *scale,
                           float *part, int ns, const ChainKTask &task, const unsigned char *consts, const PairsGeom &g,
                           hipStream_t st) {
    const int Mp = dpgp_round_up(M, 16);
    const int nper = dpgp_ceil_div(N, ns);
    const long long nwg = (long long)B * ns * g.n_ranges + (task.ws ? B : 0);
    if (nwg > 0x7fffffffLL) return -1;
    auto kern = psi2_pairs_kernel<TIN, KS, NW>;
    if (NW == 8 && task.ws) return -16;
    if (g.lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds) !=
            hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * NW), g.lds, st, N, M, Q, B, consts, mu, s, gamma, scale, part,
                       Mp, nper, ns, g.n_ranges, g.tiles_per_range, g.R, task);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

// part[ns][B][Mp][Mp] (fp32; entries m' <= m < M written), consts: psi2_consts_bytes(M, Q) bytes, built by launch_psi2_consts
// or by the KL / y^T y launch of the fused ELBO before this call; scale: psi2_pairs_scale_bytes(B, M) bytes of scratch.  task.ws != nullptr: the K_uu branch rides in the dispatch
// (LDS-resident sizes only).
size_t psi2_pairs_scale_bytes(int B, int M) {
    const size_t ppad = ((size_t)M * (M + 1) / 2 + 31) & ~(size_t)31;
    return dpgp_align256(sizeof(float) * (size_t)B * ppad);
}
template <typename TIN>
int launch_psi2_pairs(int B, int N, int M, int Q, const TIN *z, const TIN *mu, const TIN *s, const TIN *gamma,
                      const TIN *alpha, float *part, int ns, const ChainKTask &task, const unsigned char *consts, float *scale,
                      int scale_ready, hipStream_t st) {
    if (!consts || !scale) return -18;
    if (!scale_ready) {        // (the fused ELBO builds the table in its front launch)
        const Psi2Consts C = psi2_consts_layout(M, Q);
        DPGP_PRELAUNCH(); hipLaunchKernelGGL((psi2_pair_scale_kernel<TIN>), dim3(dpgp_ceil_div(C.Ppad, 256), B), dim3(256), 0, st, M, Q, z,
                           gamma, alpha, scale);
        DPGP_LAUNCH_CHECK();
    }
    size_t chain_lds = 0;
    if (task.ws) {
        if (!chain_k_resident(task.Mp, task.elem)) return -16;
        chain_lds = chain_k_lds_bytes(task.Mp, task.elem);
    }
    const PairsGeom g = pairs_geom(B, N, M, Q, ns, chain_lds);
    switch (g.KS) {
#define CASE(k)                                                                                                       \
    case k:                                                                                                           \
        return g.NW == 8 ? launch_pairs_ks<TIN, k, 8>(B, N, M, Q, mu, s, gamma, scale, part, ns, task, consts, g, st) \
                         : launch_pairs_ks<TIN, k, 4>(B, N, M, Q, mu, s, gamma, scale, part, ns, task, consts, g, st);
        CASE(2) CASE(4) CASE(6) CASE(8)
#undef CASE
    }
    return -4;
}
template int launch_psi2_pairs<float>(int, int, int, int, const float *, const float *, const float *, const float *,
                                      const float *, float *, int, const ChainKTask &, const unsigned char *, float *, int,
                                      hipStream_t);
template int launch_psi2_pairs<double>(int, int, int, int, const double *, const double *, const double *, const double *,
                                       const double *, float *, int, const ChainKTask &, const unsigned char *, float *, int,
                                       hipStream_t);
