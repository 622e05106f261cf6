// Per-evaluation constants of the f16 psi2 kernel that depend on the inducing inputs z only (not on the output dim):
//   zc   [32]            column means of z (fp32; the exponent GEMM works on centred z)
//   zs   [Mp64][ZLD]     centred z rows, zero padded (rows >= M and columns >= Q are zero)
//   bimg [Mp64][SL] f16  column-side operand image of the P-row GEMM: per column m and term tt = 2 q + {0: z^2, 1: z} the
//                        f16 slots {hi, lo, hi}; slots 6Q, 6Q+1 = 1; rest 0   (psi2.hip, psi2_patch_f16p)
// Every psi2 workgroup used to rebuild its slice of these in its prologue (~4 us of ~65 us at D = 64); now one small
// block per 64 rows builds them once per evaluation (as an extra role of the KL / y^T y launch that precedes psi2 in the
// fused ELBO, or a launch of its own for the stand-alone operator) and the workgroups copy / read them.
#pragma once
#include "common.h"

struct Psi2Consts {
    int ZLD, SL, Mp64;
    size_t off_zs, off_bimg, bytes;
    // pair image of the pair-tile psi2 kernel (psi2_pairs.hip): SLP f16 slots per pair p = m (m + 1) / 2 + m' (m' <= m), sums
    // s = (z_m - c) + (z_m' - c): per latent dim q the slots {hi, lo, hi} of s_q^2 (term 2q) and of s_q (term 2q + 1), slots
    // 6Q, 6Q + 1 = 1, rest 0; pairs [P, Ppad) are zero.  Independent of the output dim.  Stored in MFMA OPERAND ORDER: for the
    // tile of 32 pairs T = p / 32 and the K-step ks = slot / 16 the 64 lanes' 16-byte operand words are contiguous (lane =
    // 32 (slot % 16 / 8) + p % 32 holds slots 16 ks + 8 (lane / 32) .. + 7 of pair 32 T + lane % 32): one wave-wide load is one
    // coalesced KB (with one row of slots per pair the 64 lanes of a load touched 32 different cache lines, and the address
    // path of the texture unit, not the arithmetic, set the kernel's pace).
    int KS, SLP, P, Ppad;
    size_t off_pairs;
    // per pair, for the kernel's epilogue: pmap[p] = m << 16 | m' (0xffffffff for p >= P)
    size_t off_pmap;
};
// K-steps of 16 slots that hold the 6Q + 2 slots of a row; only these instantiations of the kernel exist
__host__ __device__ inline int psi2_pairs_ksteps(int Q) {
    const int ks = (6 * Q + 2 + 15) / 16;
    return ks <= 2 ? 2 : (ks <= 4 ? 4 : (ks <= 6 ? 6 : (ks <= 8 ? 8 : 12)));
}
__host__ __device__ inline Psi2Consts psi2_consts_layout(int M, int Q) {
    Psi2Consts c;
    const int KQ = 4 * ((Q + 3) / 4);
    c.ZLD = ((KQ / 4) & 1) ? KQ : KQ + 4;                      // = Psi2F16Lds<KB>::ZLD
    c.SL = 32 * ((6 * Q + 2 + 31) / 32);                       // = psi2p_layout<KB>(Q).SL
    c.Mp64 = (M + 63) & ~63;
    c.off_zs = 128;
    c.off_bimg = c.off_zs + sizeof(float) * (size_t)c.Mp64 * c.ZLD;
    c.off_pairs = (c.off_bimg + sizeof(_Float16) * (size_t)c.Mp64 * c.SL + 255) & ~(size_t)255;
    c.KS = psi2_pairs_ksteps(Q);
    c.SLP = 16 * c.KS;
    c.P = (int)((long long)M * (M + 1) / 2);
    c.Ppad = (c.P + 31) & ~31;
    c.off_pmap = (c.off_pairs + sizeof(_Float16) * (size_t)c.Ppad * c.SLP + 255) & ~(size_t)255;
    c.bytes = (c.off_pmap + sizeof(unsigned) * (size_t)c.Ppad + 255) & ~(size_t)255;
    return c;
}

// rows [64 blk, 64 blk + 64) by one 256-thread workgroup; scratch: 8 * 32 doubles + 32 floats of LDS
template <typename TIN>
__device__ __forceinline__ void psi2_consts_rows(const TIN *__restrict__ z, int M, int Q, unsigned char *__restrict__ dst,
                                                 int blk, double *scratch) {
    const Psi2Consts c = psi2_consts_layout(M, Q);
    float *zc = reinterpret_cast<float *>(scratch + 8 * 32);
    block_column_means(z, M, Q, zc, scratch);
    const int t = threadIdx.x;
    if (blk == 0 && t < 32) reinterpret_cast<float *>(dst)[t] = (t < Q) ? zc[t] : 0.0f;
    float *zs = reinterpret_cast<float *>(dst + c.off_zs);
    _Float16 *bimg = reinterpret_cast<_Float16 *>(dst + c.off_bimg);
    const int m0 = 64 * blk;
    for (int e = t; e < 64 * c.ZLD; e += 256) {
        const int r = e / c.ZLD, k = e - r * c.ZLD, m = m0 + r;
        zs[(size_t)m * c.ZLD + k] = (k < Q && m < M) ? (float)z[(size_t)m * Q + k] - zc[k] : 0.0f;
    }
    for (int e = t; e < 64 * c.SL / 2; e += 256) {              // one 32-bit word = two f16 slots
        const int r = e / (c.SL / 2), w = e - r * (c.SL / 2), m = m0 + r;
        _Float16 h2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int slot = 2 * w + i;
            float val = 0.0f;
            if (slot < 6 * Q) {
                const int tt = slot / 3, part = slot - 3 * tt, q = tt >> 1;
                const float zz = (m < M) ? (float)z[(size_t)m * Q + q] - zc[q] : 0.0f;
                const float v = dpgp_pin((tt & 1) ? zz : zz * zz);     // (pinned before the (hi, lo) split: see dpgp_pin)
                const _Float16 h = (_Float16)v;
                val = (part == 1) ? (float)(_Float16)(v - (float)h) : (float)h;
            } else if (slot < 6 * Q + 2) {
                val = 1.0f;
            }
            h2[i] = (_Float16)val;
        }
        typedef _Float16 h2v __attribute__((ext_vector_type(2)));
        const h2v hv = {h2[0], h2[1]};
        reinterpret_cast<unsigned *>(bimg)[(size_t)m * (c.SL / 2) + w] = __builtin_bit_cast(unsigned, hv);
    }
}

// pair (m, m'), m' <= m, of the row-major lower-triangle index p
__host__ __device__ inline void psi2_pair_of(int p, int &m, int &mp) {
    int i = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
    while ((long long)(i + 1) * (i + 2) / 2 <= p) ++i;
    while ((long long)i * (i + 1) / 2 > p) --i;
    m = i;
    mp = p - (int)((long long)i * (i + 1) / 2);
}
#define PSI2_PAIR_ROWS_PER_BLOCK 256
#define PSI2_PAIR_S2_SCALE 0.015625f     // 1 / 64 (see psi2_pair_rows); the kernel's a-coefficients carry the 64
// pair-image rows [256 blk, 256 blk + 256) by one 256-thread workgroup; scratch as psi2_consts_rows
template <typename TIN>
__device__ __forceinline__ void psi2_pair_rows(const TIN *__restrict__ z, int M, int Q, unsigned char *__restrict__ dst,
                                               int blk, double *scratch) {
    const Psi2Consts c = psi2_consts_layout(M, Q);
    float *zc = reinterpret_cast<float *>(scratch + 8 * 32);
    block_column_means(z, M, Q, zc, scratch);
    const int t = threadIdx.x, wpr = c.SLP / 2;                  // 32-bit words per row
    unsigned *img = reinterpret_cast<unsigned *>(dst + c.off_pairs);
    const int p0 = PSI2_PAIR_ROWS_PER_BLOCK * blk;
    if (p0 + t < c.Ppad) {                                       // thread = pair: index map
        const int p = p0 + t;
        int m = 0, mp = 0;
        if (p < c.P) psi2_pair_of(p, m, mp);
        reinterpret_cast<unsigned *>(dst + c.off_pmap)[p] = (p < c.P) ? ((unsigned)m << 16 | (unsigned)mp) : 0xffffffffu;
    }
    // item = (pair r of the block, latent dim q; q == Q: the tail words): the six slots {h, l, h | h, l, h} of (s_q^2 / 64, s_q)
    // are the three words 3q .. 3q + 2 of the pair, written in operand order (layout: struct Psi2Consts)
    auto put = [&](int p, int w, unsigned word) {
        const int ks = w >> 3, hf = (w >> 2) & 1, wi = w & 3, lane = 32 * hf + (p & 31);
        img[(((size_t)(p >> 5) * c.KS + ks) * 64 + lane) * 4 + wi] = word;
    };
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    for (int e = t; e < PSI2_PAIR_ROWS_PER_BLOCK * (Q + 1); e += 256) {
        const int r = e % PSI2_PAIR_ROWS_PER_BLOCK, q = e / PSI2_PAIR_ROWS_PER_BLOCK, p = p0 + r;
        if (p >= c.Ppad) continue;
        const bool real = p < c.P;
        if (q == Q) {                                           // slots 6Q, 6Q + 1 (and 6Q + 2 where it exists) = 1, rest 0
            const h2v one2 = {(_Float16)1.0f, (_Float16)1.0f}, one1 = {(_Float16)1.0f, (_Float16)0.0f};
            for (int w = 3 * Q; w < wpr; ++w) {
                unsigned word = 0u;
                if (real && w == 3 * Q) word = __builtin_bit_cast(unsigned, one2);
                if (real && w == 3 * Q + 1 && 6 * Q + 2 < c.SLP) word = __builtin_bit_cast(unsigned, one1);
                put(p, w, word);
            }
            continue;
        }
        unsigned w0 = 0u, w1 = 0u, w2 = 0u;
        if (real) {
            int m, mp;
            psi2_pair_of(p, m, mp);
            const float sq = ((float)z[(size_t)m * Q + q] - zc[q]) + ((float)z[(size_t)mp * Q + q] - zc[q]);
            // s^2 / 64 against 64 a on the other side keeps both f16 lo pieces well inside the normal range
            const float f1 = dpgp_pin(sq * sq * PSI2_PAIR_S2_SCALE), f2 = dpgp_pin(sq);     // (pinned: see dpgp_pin)
            const _Float16 f1h = (_Float16)f1, f1l = (_Float16)(f1 - (float)f1h);
            const _Float16 f2h = (_Float16)f2, f2l = (_Float16)(f2 - (float)f2h);
            const h2v a = {f1h, f1l}, b = {f1h, f2h}, d = {f2l, f2h};
            w0 = __builtin_bit_cast(unsigned, a); w1 = __builtin_bit_cast(unsigned, b); w2 = __builtin_bit_cast(unsigned, d);
        }
        put(p, 3 * q, w0); put(p, 3 * q + 1, w1); put(p, 3 * q + 2, w2);
    }
}

// scale[b][p] = alpha_b^2 exp2(beta_bp),  beta_bp = -1/4 log2e sum_q gamma_bq (z_mq - z_m'q)^2: the per-output factor of every
// pair of the pair-tile psi2 kernel (applied once per column sum instead of once per exponent); 256 pairs per block, thread =
// pair, straight from z (no dependency on the other constants).  Runs as extra blocks of the front launch of the fused ELBO,
// or as a launch of its own (psi2_pairs.hip).
// output dims per workgroup of the scale role: 16 at D = 512 (the gather amortised), fewer for few output dims (D = 64: 2 —
// with 16 the role is 132 long workgroups and the front launch of the per-GPU share takes 16.0 instead of 14.6 us)
__host__ __device__ inline int psi2_scale_dchunk(int D) {
    const int c = D / 32;
    return c < 1 ? 1 : (c > 16 ? 16 : c);
}
// output dims [b0, b0 + nb) of pair block pblk: the pair's squared differences are formed ONCE (the gather of two z rows and
// the index arithmetic of psi2_pair_of were the cost of the per-(output dim, pair) form: 43.6 us at config 3 for 17 MB of
// output) and every output dim adds Q FMAs, one v_exp_f32 and one coalesced store
// lds != nullptr (>= nb (Q + 1) floats of LDS, all 256 threads of the workgroup call): the chunk's gamma rows and alpha^2 are
// staged there first — read from global memory inside the loop every output dim waited for its own Q + 1 scalar loads
template <typename TIN, typename TG>
__device__ __forceinline__ void psi2_pair_scale_chunk(int b0, int nb, int pblk, int M, int Q, const TIN *__restrict__ z,
                                                      const TG *__restrict__ gamma, const TG *__restrict__ alpha,
                                                      float *__restrict__ scale, float *__restrict__ lds = nullptr) {
    const int P = (int)((long long)M * (M + 1) / 2), Ppad = (P + 31) & ~31;
    const int p = pblk * 256 + (int)threadIdx.x;
    if (lds) {
        for (int e = (int)threadIdx.x; e < nb * (Q + 1); e += 256) {
            const int bb = e / (Q + 1), q = e - bb * (Q + 1);
            const float al = (float)alpha[b0 + bb];
            lds[e] = q < Q ? (float)gamma[(size_t)(b0 + bb) * Q + q] : al * al;
        }
        __syncthreads();
    }
    if (p >= Ppad) return;
    float d2[DPGP_MAX_Q];
#pragma unroll
    for (int q = 0; q < DPGP_MAX_Q; ++q) d2[q] = 0.0f;
    if (p < P) {
        int m, mp;
        psi2_pair_of(p, m, mp);
#pragma unroll
        for (int q = 0; q < DPGP_MAX_Q; ++q)
            if (q < Q) {
                const float d = (float)((double)z[(size_t)m * Q + q] - (double)z[(size_t)mp * Q + q]);
                d2[q] = d * d;
            }
    }
    for (int b = b0; b < b0 + nb; ++b) {
        float bsum = 0.0f, al2;
        if (lds) {
            const float *gl = lds + (b - b0) * (Q + 1);
#pragma unroll
            for (int q = 0; q < DPGP_MAX_Q; ++q)
                if (q < Q) bsum += gl[q] * d2[q];
            al2 = gl[Q];
        } else {
#pragma unroll
            for (int q = 0; q < DPGP_MAX_Q; ++q)
                if (q < Q) bsum += (float)gamma[(size_t)b * Q + q] * d2[q];
            const float al = (float)alpha[b];
            al2 = al * al;
        }
        scale[(size_t)b * Ppad + p] = al2 * __builtin_amdgcn_exp2f((float)(-0.25 * DPGP_LOG2E) * bsum);
    }
}
template <typename TIN, typename TG>
__device__ __forceinline__ void psi2_pair_scale_block(int b, int pblk, int M, int Q, const TIN *__restrict__ z,
                                                      const TG *__restrict__ gamma, const TG *__restrict__ alpha,
                                                      float *__restrict__ scale) {
    psi2_pair_scale_chunk<TIN, TG>(b, 1, pblk, M, Q, z, gamma, alpha, scale);
}
