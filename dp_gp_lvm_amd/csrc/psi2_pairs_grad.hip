// Stage B of the backward pass, Psi2 term, in the PAIR-TILE form of psi2_pairs.hip (reference: tf.gradients through
// /root/reference/src/kernels/rbf_kernel.py:164-199).  With the forward's notation (psi2_pairs.hip)
//      psi2[d, p] = alpha_d^2 exp2(beta_dp) sum_n exp2(E_dnp),     E = A_dn . B_p  (a' s2' + b s + c'' 1 over the K slots),
// and L = sum_d sum_p g_dp psi2[d, p]  (g = the adjoint of Psi2 folded onto the pairs: 2 G[m, m'] off the diagonal), every
// derivative is a contraction of the SAME matrix W_dnp = exp2(E_dnp) with the operands of the exponent GEMM:
//      pair side   R2[d, p, :] = sum_n W_dnp (a'_dn., b_dn., 1)            -> d/dz (through s and beta), d/dgamma (through beta)
//      obs. side   R1[d, n, :] = sum_p W_dnp u_dp (s2'_p., s_p., 1)        -> d/dmu, d/dS, d/dgamma (through a', b, c'')
// with u_dp = g_dp alpha_d^2 exp2(beta_dp).  Both are second GEMMs on the matrix pipe whose one operand is the exponential
// tile just computed: the 16 result registers of a lane ARE the B operand (8 k-slots per K-step) of
// v_mfma_f32_32x32x16_f16 when the contraction runs over the tile's ROWS — so the pass whose rows are the observations gives
// R2 and a second pass with the roles swapped (rows = pairs, columns = observations) gives R1; a transposition inside the
// wave would cost as much as recomputing the exponentials.  The per-element work is exp2 + the (hi, lo) f16 split of W; the
// factor u_dp never touches an element (it multiplies a column of R2 afterwards, and is folded into the features of pass 2).
// The round-1 kernel (psi2_grad_kernel, psi2.hip) walks the FULL M x M square per observation patch by patch (twice the
// exponentials) and re-forms operands per observation; it stays for Q > 10.
//
// Workgroup = (output dim, 16 column tiles: 4 waves x 4 resident tiles); it loops over ALL row chunks with the second-product
// accumulators in registers and writes them once — no partial slabs, no atomics (bit-reproducible).
#include <type_traits>
#include <utility>
#include "internal.h"
#include "psi2_consts.h"

typedef _Float16 pg_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 pg_h2 __attribute__((ext_vector_type(2)));
typedef float pg_f16v __attribute__((ext_vector_type(16)));
typedef float pg_f4 __attribute__((ext_vector_type(4)));
typedef unsigned pg_u4 __attribute__((ext_vector_type(4)));

#define PG_HDR 512
#define PG_WSHIFT 12.0f              // the exponent tiles carry + PG_WSHIFT: W = 2^12 exp2(E) <= 4096 uses the f16 range downwards
                                     // (f16 pairs resolve W / max W down to ~2^-36 instead of 2^-24); undone in the finishing kernels
#define PG_FB 32                     // feature rows of one block of the second product (one 32-row matrix instruction)

// Kernel configuration by the K-steps of the exponent product (KS = psi2_pairs_ksteps(Q)):
//   NFB  feature blocks of 32 rows of the second product (2Q + 1 features: Q <= 15 one block, Q <= 20 two)
//   G    resident column tiles per wave;  NW waves per workgroup, ONE workgroup per compute unit (two waves per SIMD, 256
//        registers per lane, the whole LDS): with G = 4 and four waves (round 3) the new loop spilled its column operands
//   WPS  waves per SIMD the register allocation allows (workgroups per compute unit = 4 WPS / NW, each with its share of the LDS)
#ifndef PG_G4
#define PG_G4 2
#endif
#ifndef PG_NW4
#define PG_NW4 8
#endif
#ifndef PG_WPS4
#define PG_WPS4 2
#endif
template <int KS> struct PgCfg { static constexpr int NFB = 1, G = PG_G4, NW = PG_NW4, WPS = PG_WPS4; };
template <> struct PgCfg<6> { static constexpr int NFB = 1, G = 2, NW = 8, WPS = 2; };
#ifndef PG_G8
#define PG_G8 1
#define PG_NW8 8
#define PG_WPS8 2
#endif
template <> struct PgCfg<8> { static constexpr int NFB = 2, G = PG_G8, NW = PG_NW8, WPS = PG_WPS8; };   // (G = 2 at two waves per SIMD spills 130 registers)
__host__ __device__ inline int pg_nfb(int KS) { return KS >= 8 ? 2 : 1; }
// 1 KB pieces of the feature image of one row tile.  One block: [hi | lo] x [K-step 0 | 1].  Two blocks (33 .. 41 features, Q = 16 .. 20):
// the second block holds at most 9 features — its hi words sit in rows 0-15 and its lo words in rows 16-31 of ONE operand, so that
// (X_hi, W_hi) and (X_lo, W_hi) are one matrix instruction and the block needs no lo plane: 6 pieces, 5 products per K-step half instead
// of 8 and 6 (the rows r and r + 16 of its result are added in the pass kernel's epilogue)
__host__ __device__ constexpr int pg_xp(int NFB) { return NFB == 2 ? 6 : 4; }
// a pass result: [column tile][feature f < NF][32 columns]; col = (set * padded columns) + column, padded to tiles of 32
template <int NF> __device__ __forceinline__ size_t pg_oix(size_t col, int f) { return ((col >> 5) * NF + f) * 32 + (col & 31); }

// position of (feature f < 32 of a block, row rr of a 32-row tile) in the transposed feature image of one (row tile, block,
// kind): the k-slot order of the second product's operands = the register order the exponent tile arrives in
__device__ __forceinline__ int pg_xt_index(int f, int rr) {
    const int c = rr >> 3, h = (rr >> 2) & 1, j = rr & 3, s_ = c >> 1, t_ = 4 * (c & 1) + j;
    return (((s_ * 2 + h) * 32) + f) * 8 + t_;
}
// feature f of row rr as an f16 (hi, lo) pair into the transposed images of this row tile: xt = [hi | lo][1024] of block 0, then the one plane of block 1 (pg_xp)
__device__ __forceinline__ void pg_put(_Float16 *xt, int f, int rr, float v) {
    v = dpgp_pin(v);
    const _Float16 vh = (_Float16)v, vl = (_Float16)(v - (float)vh);
    if (f < 32) {
        const int ix = pg_xt_index(f, rr);
        xt[ix] = vh;
        xt[1024 + ix] = vl;
    } else {                                                  // second block (f < 48): hi in row f - 32, lo in row f - 16 of its one plane
        xt[2048 + pg_xt_index(f - 32, rr)] = vh;
        xt[2048 + pg_xt_index(f - 16, rr)] = vl;
    }
}

// ---- the row of the exponent GEMM's A operand for observation n of output dim d (as phase A of psi2_pairs_kernel) --------
// dst: 8 KS words (16 KS f16 slots) in REGISTERS (the loop over the latent dims is unrolled to the most a K-step count holds, so every
// index is a constant); the features a'_q, b_q (the values the slots were split from) of this row (rr within its tile) go into the
// transposed images of its row tile.  mu and s of the row are fetched up front — inside the loop every latent dim waited for its own
// two loads (config 5: 309 of the kernel's microseconds).
template <int KS>
__device__ __forceinline__ bool pg_obs_row(bool valid, int n, int Q, const double *__restrict__ mu, const double *__restrict__ s,
                                           const float *gq, const float *zc, unsigned (&dst)[8 * KS], _Float16 *xt, int rr,
                                           float denfac = 2.0f, float half = 1.0f, float xw = 1.0f) {
    constexpr int SLP = 16 * KS, QM = (SLP - 2) / 6;              // 6 Q + 2 slots <= 16 KS (the third part of the row constant only where a slot is left)
    bool oor = false;
    float cc = -60000.0f;
    float svv[QM], mcv[QM];
#pragma unroll
    for (int q = 0; q < QM; ++q) {
        const bool lq = valid && q < Q;
        svv[q] = lq ? (float)s[(size_t)n * Q + q] : 0.0f;
        mcv[q] = lq ? (float)mu[(size_t)n * Q + q] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < SLP / 2; ++k) dst[k] = 0u;
    if (valid) {
        cc = 0.0f;
#pragma unroll
        for (int q = 0; q < QM; ++q) {
            if (q < Q) {
                const float g = gq[q], sv = svv[q], mc = mcv[q] - zc[q];
                // Psi2: denfac = 2, half = 1.  Psi1 on the DIAGONAL pairs (s = 2 z', see launch_psi1_pgrad): den = g s + 1 and half the
                // coefficients — log2 psi1 / alpha = c + sum_q (-8 log2e w1) (s^2 / 64) + (1/2 log2e w1 mu') s
                const float den = denfac * g * sv + 1.0f, w = g / den;
                const float a = dpgp_pin(half * (float)(-0.25 * DPGP_LOG2E / PSI2_PAIR_S2_SCALE) * w);
                const float bb = dpgp_pin(half * (float)DPGP_LOG2E * w * mc);
                cc -= bb * mc + 0.5f * __builtin_amdgcn_logf(den);
                const _Float16 ah = (_Float16)a, alo = (_Float16)(a - (float)ah);
                const _Float16 bh = (_Float16)bb, blo = (_Float16)(bb - (float)bh);
                const pg_h2 w0 = {ah, ah}, w1 = {alo, bh}, w2 = {bh, blo};
                dst[3 * q] = __builtin_bit_cast(unsigned, w0);
                dst[3 * q + 1] = __builtin_bit_cast(unsigned, w1);
                dst[3 * q + 2] = __builtin_bit_cast(unsigned, w2);
                // (xw: a per-row weight of the features — y_nd for the Psi1 term)
                pg_put(xt, 2 * q, rr, xw * ((float)ah + (float)alo));
                pg_put(xt, 2 * q + 1, rr, xw * ((float)bh + (float)blo));
            }
        }
        oor = !(cc >= -8192.0f);                                  // range guard of the f16-split exponent (psi2_pairs.hip)
        cc = fmaxf(cc, -60000.0f) + PG_WSHIFT;
    } else {
#pragma unroll
        for (int q = 0; q < QM; ++q) {
            if (q < Q) {
                pg_put(xt, 2 * q, rr, 0.0f);
                pg_put(xt, 2 * q + 1, rr, 0.0f);
            }
        }
    }
    cc = dpgp_pin(cc);
    const _Float16 ch = (_Float16)cc;
    const float r1 = dpgp_pin(cc - (float)ch);
    const _Float16 cm = (_Float16)r1;
    const pg_h2 cw = {ch, cm}, cw2 = {(_Float16)(r1 - (float)cm), (_Float16)0.0f};
    // (the row constant sits behind the last latent dim: a runtime position in the register array — selected, not indexed)
#pragma unroll
    for (int q = 0; q <= QM; ++q)
        if (q == Q) {
            dst[3 * q] = __builtin_bit_cast(unsigned, cw);
            if (3 * q + 1 < SLP / 2 && 6 * q + 2 < SLP) dst[3 * q + 1] = __builtin_bit_cast(unsigned, cw2);
        }
    return oor;
}

// ---- u[d][p] = g_dp alpha_d^2 exp2(beta_dp), and kap[d] = the power of two that brings max_p |u_dp| into [2^7, 2^8) (times |s| <= ~100: inside the f16 range) ----
__global__ __launch_bounds__(256) void pg_u_kernel(int M, int Q, int Mp, const double *__restrict__ z,
                                                   const double *__restrict__ gamma, const double *__restrict__ alpha,
                                                   const double *__restrict__ GP, float *__restrict__ u, float *__restrict__ kap) {
    __shared__ float red[256];
    const int d = blockIdx.x, t = threadIdx.x;
    const int P = (int)((long long)M * (M + 1) / 2), Ppad = (P + 31) & ~31;
    const float al = (float)alpha[d], al2 = al * al;
    const double *Gd = GP + (size_t)d * Mp * Mp;
    float mx = 0.0f;
    for (int p = t; p < Ppad; p += 256) {
        float val = 0.0f;
        if (p < P) {
            int m, mp;
            psi2_pair_of(p, m, mp);
            float bsum = 0.0f;
            for (int q = 0; q < Q; ++q) {
                const float dd = (float)(z[(size_t)m * Q + q] - z[(size_t)mp * Q + q]);
                bsum += (float)gamma[(size_t)d * Q + q] * dd * dd;
            }
            const float g = (float)Gd[(size_t)m * Mp + mp] * (m == mp ? 1.0f : 2.0f);
            val = g * al2 * __builtin_amdgcn_exp2f((float)(-0.25 * DPGP_LOG2E) * bsum);
        }
        u[(size_t)d * Ppad + p] = val;
        mx = fmaxf(mx, fabsf(val));
    }
    red[t] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) red[t] = fmaxf(red[t], red[t + o]);
        __syncthreads();
    }
    if (t == 0) {
        float k = 1.0f;
        const float m0 = red[0];
        if (m0 > 0.0f && m0 < 3.0e38f) {
            int ex;
            (void)frexpf(m0, &ex);
            ex = max(-100, min(100, ex));
            k = ldexpf(1.0f, 8 - ex);
        }
        kap[d] = k;
    }
}

// the same from the forward's per-pair factors scale[d][p] = alpha_d^2 exp2(beta_dp) (training step: the table exists)
__global__ __launch_bounds__(1024) void pg_u_scale_kernel(int Ppad, int Mp, int Q, const unsigned *__restrict__ pmap,
                                                          const float *__restrict__ scale, const double *__restrict__ GP,
                                                          float *__restrict__ u, float *__restrict__ kap,
                                                          const float *__restrict__ psi2, const double *z,
                                                          double *__restrict__ dgamma, int zlds) {
    // psi2 != nullptr ([D][Mp][Mp], slab 0 of the forward's partial slabs = the column sums of pass 1 with their per-pair factors):
    // the derivative through beta_dp = -1/4 log2e sum_q gamma_dq delta_pq^2 needs only g_dp psi2_dp per pair,
    //     dgamma[d][q] += sum_p -1/4 delta_pq^2 g_dp psi2_dp        (what pg_dgamma_pairs_kernel forms from u_dp and R2's constant feature)
    // block = output dim, 1024 threads (a 256-thread block walked 33 dependent trips at M = 128), wave sums, one LDS hand-over
    __shared__ float red[16];
    __shared__ double redd[16][DPGP_MAX_Q];
    extern __shared__ __align__(16) unsigned char u_smem[];    // z [M][Q] where it fits (zlds): the 2 Q gathered reads per pair come from
    double *zs = reinterpret_cast<double *>(u_smem);           // LDS instead of 8-byte gathers through the texture path (39 us of them)
    const int d = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const double *Gd = GP + (size_t)d * Mp * Mp;
    const float *Pd = psi2 ? psi2 + (size_t)d * Mp * Mp : nullptr;
    if (zlds && Pd) {
        for (int i = t; i < zlds; i += 1024) zs[i] = z[i];
        __syncthreads();
        z = zs;
    }
    float mx = 0.0f;
    double a[DPGP_MAX_Q];
#pragma unroll
    for (int q = 0; q < DPGP_MAX_Q; ++q) a[q] = 0.0;
    for (int p = t; p < Ppad; p += 1024) {
        const unsigned pm = pmap[p];
        float val = 0.0f;
        if (pm != 0xffffffffu) {
            const int m = pm >> 16, mp = pm & 0xffffu;
            const float g = (float)Gd[(size_t)m * Mp + mp] * (m == mp ? 1.0f : 2.0f);
            val = g * scale[(size_t)d * Ppad + p];
            if (Pd && m != mp) {
                const double gp2 = -0.25 * (double)g * (double)Pd[(size_t)m * Mp + mp];
#pragma unroll
                for (int q = 0; q < DPGP_MAX_Q; ++q)
                    if (q < Q) {
                        const double dd = z[(size_t)m * Q + q] - z[(size_t)mp * Q + q];
                        a[q] += dd * dd * gp2;
                    }
            }
        }
        u[(size_t)d * Ppad + p] = val;
        mx = fmaxf(mx, fabsf(val));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wv] = mx;
    if (Pd) {
#pragma unroll
        for (int q = 0; q < DPGP_MAX_Q; ++q)
            if (q < Q) {
                const double v = wave_sum(a[q]);
                if (lane == 0) redd[wv][q] = v;
            }
    }
    __syncthreads();
    if (t == 0) {
        float k = 1.0f, m0 = 0.0f;
        for (int w = 0; w < 16; ++w) m0 = fmaxf(m0, red[w]);
        if (m0 > 0.0f && m0 < 3.0e38f) {
            int ex;
            (void)frexpf(m0, &ex);
            ex = max(-100, min(100, ex));
            k = ldexpf(1.0f, 8 - ex);
        }
        kap[d] = k;
    }
    if (Pd && t >= 64 && t - 64 < Q) {
        double v = 0.0;
        for (int w = 0; w < 16; ++w) v += redd[w][t - 64];
        dgamma[(size_t)d * Q + t - 64] += v;
    }
}

// ---- precomputed images -----------------------------------------------------------------------------------------------
// Every workgroup of a pass re-reads the row images of its output dim chunk by chunk; they are built ONCE per evaluation
// (observation side: per output dim; pair side: the exponent rows are shared by all output dims — the forward's pair image in
// psi2_consts — and the features carry u_dp):
//   operand order   cimg[set][tile][ks][lane 0..63][8 halves]   lane = 32 half + row % 32 holds slots 16 ks + 8 half .. + 7:
//                   the SAME image serves as the column operand of one pass (registers) and as the row operand of the other
//                   (LDS: a wave-wide read of one K-step is 1 KB contiguous, no padding, and the fill is a linear copy)
//   features        ximg[set][tile][block][kind hi / lo][K-step 0 / 1][lane][8 halves]   (pg_xt_index: the second product's A operand)
template <int KS>
__global__ __launch_bounds__(256) void pg_obs_images_kernel(int N, int Q, const unsigned char *__restrict__ consts,
                                                            const double *__restrict__ mu, const double *__restrict__ s,
                                                            const double *__restrict__ gamma, _Float16 *__restrict__ cimg,
                                                            _Float16 *__restrict__ ximg, int NT, int *__restrict__ flag,
                                                            const double *__restrict__ y, int ldy) {
    // y != nullptr: the images of the Psi1 term (den = g s + 1, half coefficients, features weighted by y_nd)
    constexpr int NFB = PgCfg<KS>::NFB, XP = pg_xp(NFB), NFZ = NFB == 2 ? 48 : 32;   // (NFZ: feature slots that exist in the image)
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float *gq = reinterpret_cast<float *>(smem_raw), *zc = gq + 32;
    _Float16 *xt = reinterpret_cast<_Float16 *>(smem_raw + 256);                   // [8 tiles][XP pieces][64][8]
    const int d = blockIdx.y, t = threadIdx.x, n0 = 256 * blockIdx.x;
    if (t < 32) {
        gq[t] = (t < Q) ? (float)gamma[(size_t)d * Q + t] : 0.0f;
        zc[t] = reinterpret_cast<const float *>(consts)[t];
    }
    __syncthreads();
    const int tile0 = n0 / 32, ntl = min(8, NT - tile0);
    {
        _Float16 *xr = xt + (size_t)(t >> 5) * 512 * XP;
        const bool valid = n0 + t < N;
        const float xw = (y && valid) ? (float)y[(size_t)(n0 + t) * ldy + d] : 1.0f;
        unsigned row[8 * KS];
        const bool oor = pg_obs_row<KS>(valid, n0 + t, Q, mu, s, gq, zc, row, xr, t & 31, y ? 1.0f : 2.0f, y ? 0.5f : 1.0f, xw);
        pg_put(xr, 2 * Q, t & 31, valid ? xw : 0.0f);
        for (int f = 2 * Q + 1; f < NFZ; ++f) pg_put(xr, f, t & 31, 0.0f);
        if (oor) atomicOr(flag, 1);
        // the row's 2 KS sixteen-byte pieces straight into the operand-order image: piece (ks, half) of the 32 rows of a tile is a
        // 512-byte run
        if ((t >> 5) < ntl) {
            pg_u4 *cd = reinterpret_cast<pg_u4 *>(cimg) + ((size_t)d * NT + tile0 + (t >> 5)) * KS * 64 + (t & 31);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    cd[ks * 64 + 32 * hf] = (pg_u4){row[8 * ks + 4 * hf], row[8 * ks + 4 * hf + 1], row[8 * ks + 4 * hf + 2], row[8 * ks + 4 * hf + 3]};
        }
    }
    __syncthreads();
    pg_u4 *xd = reinterpret_cast<pg_u4 *>(ximg) + ((size_t)d * NT + tile0) * 64 * XP;
    for (int e = t; e < ntl * 64 * XP; e += 256) xd[e] = reinterpret_cast<const pg_u4 *>(xt)[e];
}

// pair side: thread = pair; ximg per output dim (features x kap_d u_dp).  pimg: an operand-order image of Ppad rows (the pair image
// of psi2_consts, or its diagonal pairs for the Psi1 term: pg_diag_image_kernel), u: [D][Ppad]
template <int KS>
__global__ __launch_bounds__(256) void pg_pair_images_kernel(int Ppad, int Q, const _Float16 *__restrict__ pimg,
                                                             const float *__restrict__ u, const float *__restrict__ kap,
                                                             _Float16 *__restrict__ ximg) {
    constexpr int SLP = 16 * KS, NFB = PgCfg<KS>::NFB, XP = pg_xp(NFB), NFZ = NFB == 2 ? 48 : 32;
    __shared__ __align__(16) _Float16 xt[8 * 512 * XP];
    struct { int Ppad; } C = {Ppad};
    const int d = blockIdx.y, t = threadIdx.x, p0 = 256 * blockIdx.x, p = p0 + t, PT = C.Ppad / 32;
    const int tile0 = p0 / 32, ntl = min(8, PT - tile0);
    if (p < C.Ppad) {
        unsigned row[SLP / 2];
        const pg_u4 *src = reinterpret_cast<const pg_u4 *>(pimg) + ((size_t)(p >> 5) * KS) * 64 + (p & 31);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const pg_u4 w = src[ks * 64 + 32 * hf];
                row[8 * ks + 4 * hf] = w[0]; row[8 * ks + 4 * hf + 1] = w[1];
                row[8 * ks + 4 * hf + 2] = w[2]; row[8 * ks + 4 * hf + 3] = w[3];
            }
        const float up = kap[d] * u[(size_t)d * C.Ppad + p];
        _Float16 *xr = xt + (size_t)(t >> 5) * 512 * XP;
        const int rr = t & 31;
#pragma unroll
        for (int q = 0; q < DPGP_MAX_Q; ++q) {                    // slots {h, l, h | h, l, h} of (s^2 / 64, s): words 3q .. 3q + 2
            if (q < Q && 3 * q + 2 < SLP / 2) {
                const pg_h2 w0 = __builtin_bit_cast(pg_h2, row[3 * q]), w1 = __builtin_bit_cast(pg_h2, row[3 * q + 1]),
                            w2 = __builtin_bit_cast(pg_h2, row[3 * q + 2]);
                pg_put(xr, 2 * q, rr, up * ((float)w0[0] + (float)w0[1]));
                pg_put(xr, 2 * q + 1, rr, up * ((float)w1[1] + (float)w2[0]));
            }
        }
        pg_put(xr, 2 * Q, rr, up);
        for (int f = 2 * Q + 1; f < NFZ; ++f) pg_put(xr, f, rr, 0.0f);
    }
    __syncthreads();
    pg_u4 *xd = reinterpret_cast<pg_u4 *>(ximg) + ((size_t)d * PT + tile0) * 64 * XP;
    for (int e = t; e < ntl * 64 * XP; e += 256) xd[e] = reinterpret_cast<const pg_u4 *>(xt)[e];
}

// ---- Psi1 term through the same passes --------------------------------------------------------------------------------------
// psi1[n, m] = alpha exp2(c_n + sum_q a_nq z'_mq^2 + b_nq z'_mq) is the exponent product of the pair form on the DIAGONAL pairs
// p = (m, m): s = 2 z'_m, s^2 / 64 = z'^2 / 16, with half the observation-side coefficients and den = g s + 1 (pg_obs_row).
// dimg: the diagonal pairs' rows of the pair image, regathered in operand order (Mpad = 32 ceil(M / 32) rows, rows >= M zero).
template <int KS>
__global__ __launch_bounds__(256) void pg_diag_image_kernel(int M, int Mpad, const _Float16 *__restrict__ pimg, _Float16 *__restrict__ dimg) {
    const int e = blockIdx.x * 256 + threadIdx.x;               // (point m, K-step ks, lane half hf)
    if (e >= Mpad * KS * 2) return;
    const int hf = e & 1, ks = (e >> 1) % KS, m = e / (2 * KS);
    pg_u4 w = {0u, 0u, 0u, 0u};
    if (m < M) {
        const long long p = (long long)m * (m + 1) / 2 + m;
        w = reinterpret_cast<const pg_u4 *>(pimg)[((size_t)(p >> 5) * KS + ks) * 64 + 32 * hf + (int)(p & 31)];
    }
    reinterpret_cast<pg_u4 *>(dimg)[((size_t)(m >> 5) * KS + ks) * 64 + 32 * hf + (m & 31)] = w;
}
// u1[d][m] = alpha_d g_v[d][m] (the adjoint of (Psi1^T y)_dm with psi1's factor alpha), kap1[d]: the power of two that brings
// max_m |u1| into [2^7, 2^8)
__global__ __launch_bounds__(64) void pg_u1_kernel(int M, int Mp, int Mpad, const double *__restrict__ alpha, const double *__restrict__ Gv,
                                                   float *__restrict__ u1, float *__restrict__ kap1) {
    const int d = blockIdx.x, t = threadIdx.x;
    const float al = (float)alpha[d];
    float mx = 0.0f;
    for (int m = t; m < Mpad; m += 64) {
        const float v = m < M ? al * (float)Gv[(size_t)d * Mp + m] : 0.0f;
        u1[(size_t)d * Mpad + m] = v;
        mx = fmaxf(mx, fabsf(v));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (t == 0) {
        float k = 1.0f;
        if (mx > 0.0f && mx < 3.0e38f) {
            int ex;
            (void)frexpf(mx, &ex);
            ex = max(-100, min(100, ex));
            k = ldexpf(1.0f, 8 - ex);
        }
        kap1[d] = k;
    }
}
// d/dz of the Psi1 term: block = inducing point m, threads over the output dims:
//   dz[m][q] += 2 ln2 2^-12 sum_d u1_dm (2 S2 s_mq R[d][m][2q] + R[d][m][2q + 1]),  s = 2 z'
// (R: pass "rows = observations, columns = inducing points" with the y-weighted features; the 2: d s / d z')
template <int NF, int QP>
__global__ __launch_bounds__(256) void pg_finish_m1_kernel(int M, int Mpad, int Q, int D, const double *__restrict__ z,
                                                           const unsigned char *__restrict__ consts, const float *__restrict__ u1,
                                                           const float *__restrict__ r, double *__restrict__ dz) {
    __shared__ double red[4][2 * QP];
    const int m = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    double ta[QP], tb[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) { ta[q] = 0.0; tb[q] = 0.0; }
    for (int d = t; d < D; d += 256) {
        const double ud = (double)u1[(size_t)d * Mpad + m];
        const pg_f4 *row = reinterpret_cast<const pg_f4 *>(r + ((size_t)d * Mpad + m) * NF);   // (this pass writes [column][feature])
        float rv[NF];
#pragma unroll
        for (int k = 0; k < NF / 4; ++k) {
            const pg_f4 v = row[k];
            rv[4 * k] = v[0]; rv[4 * k + 1] = v[1]; rv[4 * k + 2] = v[2]; rv[4 * k + 3] = v[3];
        }
#pragma unroll
        for (int q = 0; q < QP; ++q)
            if (q < Q && 2 * q + 1 < NF) { ta[q] += ud * (double)rv[2 * q]; tb[q] += ud * (double)rv[2 * q + 1]; }
    }
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        if (q < Q) {
            const double va = wave_sum(ta[q]), vb = wave_sum(tb[q]);
            if (lane == 0) { red[wv][2 * q] = va; red[wv][2 * q + 1] = vb; }
        }
    }
    __syncthreads();
    if (t < Q) {
        const double sq = 2.0 * (z[(size_t)m * Q + t] - (double)reinterpret_cast<const float *>(consts)[t]);
        const double sa = red[0][2 * t] + red[1][2 * t] + red[2][2 * t] + red[3][2 * t];
        const double sb = red[0][2 * t + 1] + red[1][2 * t + 1] + red[2][2 * t + 1] + red[3][2 * t + 1];
        dz[(size_t)m * Q + t] += 2.0 * 0.6931471805599453 * (1.0 / 4096.0) * (2.0 * (double)PSI2_PAIR_S2_SCALE * sq * sa + sb);
    }
}

// ---- one pass: rows (LDS ring, LDS-DMA copies of an operand-order image and of ximg) x resident column tiles (cimg) --------
// out[set][column tile][32 NFB features][32 columns] (pg_oix) = sum_rows exp2(E[row, column]) X[row, :].  Pass 1: rows = observations of output dim d, columns =
// pairs; pass 2: rows = pairs, columns = observations of output dim d.  row_per_d / col_per_d: 1 = that operand image is per
// output dim, 0 = shared (the pair image).
//
// Per tile step (32 rows x 32 columns) the matrix pipe has NM = KS + 6 NFB instructions (KS of the NEXT step's exponent chain,
// 3 NFB second-product instructions on the K-step-1 half of the PREVIOUS step's exponentials and 3 NFB on the K-step-0 half of
// this step's) and the vector unit NV = 40: per two exponents  v_exp_f32 x 2, v_cvt_pk_f16_f32 (the hi words),
// v_fma_mixlo_f16 + v_fma_mixhi_f16 (lo = f16(e - hi), straight into the register of e: no second conversion).  Round 3's
// loop issued them as 7 matrix + 4 x 9 vector instructions in the first half of a step and 3 + 4 x 9 in the second (s_nops
// included): both pipes were busy ~340 of 488 cycles per step and overlapped for 199 (rocprofv3: SQ_VALU_MFMA_BUSY_CYCLES,
// SQ_ACTIVE_INST_VALU, SQ_VALU_MFMA_COEXEC_CYCLES).  Here ONE matrix instruction is followed by NV / NM vector instructions
// throughout (the order is generated at compile time: pg_slot), the exponentials of element pair i + 1 sit between those of pair
// i and their conversion (no wait state behind a transcendental), and the pipeline runs through the chunk boundaries.
struct PgPsi2Out {                    // Psi2 as a by-product of pass 1 (see the kernel's epilogue)
    float *part;                      // [D][Mp][Mp] (fp32; entries m' <= m written), or nullptr
    const float *scale;               // [D][Ppad] alpha_d^2 exp2(beta_dp) (psi2_pairs.hip)
    const unsigned *pmap;             // [Ppad] m << 16 | m'
    const int *flag;                  // range-guard flag of the image build
    int Mp, Ppad, fsel;
    int col_major;                    // != 0: out[set][column][32 NFB features] (the reader is a thread per (output dim, column): pg_finish_m1_kernel)
    double *vout;                     // [D][vM] or nullptr: alpha_d 2^-12 x the column sums of feature fsel (the Psi1 pass with rows =
    const double *valpha;             //   observations, y-weighted features: Psi1^T y of the forward pass, elbo_run's v)
    int vM;
};
struct PgSlot { int kind, i; };      // kind 0: exponent chain K-step i; 1: product i of the previous step's K-step-1 half; 2: of this step's K-step-0 half
struct PgUnit { int kind, i; };      // kind 0: v_exp_f32 of element i; 1: v_cvt_pk_f16_f32 (hi) of element pair i; 2: the two v_fma_mix_f32 (e - hi) of pair i; 3: v_cvt_pk_f16_f32 (lo)
// WLO = false (DPGP_PREC_MIXED_FAST): the exponentials enter the second product as their f16 roundings alone — no residuals, no lo
// words, no (X_hi, W_lo) products: 24 vector units and KS + 4 NFB matrix instructions per step.
template <int KS, int NFB, bool WLO> struct PgSched;
template <int KS, int NFB> struct PgSchedUnits1 {
    //   E0 E1 | (E_{2i+2} E_{2i+3} C_i) for i = 0 .. 6 | C7
    static constexpr int NU = 24, CYC = 160, U_D3 = 13, NPR = 2;
    static constexpr PgUnit unit(int u) {
        if (u < 2) return PgUnit{0, u};
        if (u == 23) return PgUnit{1, 7};
        const int i = (u - 2) / 3, r = (u - 2) % 3;
        return r < 2 ? PgUnit{0, 2 * (i + 1) + r} : PgUnit{1, i};
    }
    static constexpr int done_kind = 1;
};
template <int KS, int NFB> struct PgSchedUnits2 {
    // vector units of a step, in order (E: exponential, C: hi words, L: the two residuals, D: lo words)
    //   E0 E1 | E2 E3 C0 L0 | (E_{2i+2} E_{2i+3} C_i L_i D_{i-1}) for i = 1 .. 6 | C7 L7 D6 D7
    // the exponentials of pair i + 1 sit between those of pair i and their conversion (no wait state behind a transcendental), and
    // the lo conversion of a pair follows the NEXT pair's residuals (the compiler pads an asm statement whose result is read by
    // the very next instruction with an s_nop)
    static constexpr int NU = 40, CYC = 256, U_D3 = 25, NPR = 3;  // (U_D3: unit D_3: behind it the words 0-3 are complete)
    static constexpr PgUnit unit(int u) {
        if (u < 2) return PgUnit{0, u};
        if (u < 6) return u < 4 ? PgUnit{0, u} : PgUnit{u - 3, 0};
        if (u >= 36) return u == 36 ? PgUnit{1, 7} : (u == 37 ? PgUnit{2, 7} : PgUnit{3, u - 32});
        const int i = (u - 6) / 5 + 1, r = (u - 6) % 5;
        return r < 2 ? PgUnit{0, 2 * (i + 1) + r} : (r == 4 ? PgUnit{3, i - 1} : PgUnit{r - 1, i});
    }
    static constexpr int done_kind = 3;
};
template <int KS, int NFB, bool WLO> struct PgSched : std::conditional<WLO, PgSchedUnits2<KS, NFB>, PgSchedUnits1<KS, NFB>>::type {
    typedef typename std::conditional<WLO, PgSchedUnits2<KS, NFB>, PgSchedUnits1<KS, NFB>>::type U;
    using U::NU; using U::CYC; using U::U_D3; using U::NPR; using U::unit;
    // products per K-step half: (X_hi, W_hi), (X_hi, W_lo), (X_lo, W_hi) [WLO] or (X_hi, W_hi), (X_lo, W_hi) of block 0; of the stacked second
    // block (pg_xp) its one operand against W_hi [and W_lo].  Blocks interleaved.
    static constexpr int NK = NPR * NFB - (NFB == 2 ? 1 : 0), NM = KS + 2 * NK;
    static constexpr int prod_fb(int i) { return NFB == 1 ? 0 : (WLO ? (i < 4 ? (i & 1) : 0) : (i == 1 ? 1 : 0)); }
    static constexpr int prod_pr(int i) { return NFB == 1 ? (WLO ? i : 2 * i) : (WLO ? (i < 4 ? (i >> 1) : 2) : (i == 2 ? 2 : 0)); }
    // issue cycles (MI355X guide).  (v_fma_mixlo_f16 + v_fma_mixhi_f16 into one register would save the second conversion: measured
    // slower, adjacent (the second waits for the first) as well as spaced apart (the compiler pads each asm statement))
    static constexpr int cost(int u) { return (unit(u).kind & 1) ? 4 : 8; }
    static constexpr int cum(int u) { int c = 0; for (int k = 0; k < u; ++k) c += cost(k); return c; }
    static_assert(unit(U_D3).kind == U::done_kind && unit(U_D3).i == 3, "unit table");
    // the first J0 slots hold the KS-step exponent chain of the next step (NE1 of its instructions) and the NK products of the
    // previous step's K-step-1 half; the others the products of this step's K-step-0 half, which need the words 0-3
    static constexpr int J0raw = (cum(U_D3 + 1) * NM + CYC - 1) / CYC;
    static constexpr int J0 = J0raw < KS + NK ? J0raw : KS + NK;
    static constexpr int NE1 = J0 - NK;
    static_assert(NE1 >= 0 && NE1 <= KS && NM - J0 >= NK, "slot table");
    static constexpr PgSlot slot(int j) {
        int e = 0, k = 0;
        if (j < J0) {                                             // chain first on a tie
            for (int jj = 0;; ++jj) {
                const bool pe = (e < NE1) && (k >= NK || e * NK <= k * NE1);
                if (jj == j) return pe ? PgSlot{0, e} : PgSlot{1, k};
                if (pe) ++e; else ++k;
            }
        }
        const int ne2 = KS - NE1;
        for (int jj = J0;; ++jj) {                                // products first on a tie
            const bool pk = (k < NK) && (e >= ne2 || k * ne2 <= e * NK);
            if (jj == j) return pk ? PgSlot{2, k} : PgSlot{0, NE1 + e};
            if (pk) ++k; else ++e;
        }
    }
    static constexpr int first_of(int kind) { for (int j = 0; j < NM; ++j) if (slot(j).kind == kind) return j; return -1; }
    // matrix slot j sits in front of the first unit that starts at or behind cycle CYC j / NM of the vector stream (slots >= J0:
    // not in front of unit U_D3 + 1)
    static constexpr int unit_of_slot(int j) {
        const int want = (CYC * j + NM - 1) / NM;
        int u = 0;
        while (u < NU - 1 && cum(u) < want) ++u;
        return (j >= J0 && u <= U_D3) ? U_D3 + 1 : u;
    }
};
template <int... I, typename F> __device__ __forceinline__ void pg_unroll(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
__device__ __forceinline__ pg_h8 pg_quad(unsigned a, unsigned b, unsigned c, unsigned d) {
    return __builtin_bit_cast(pg_h8, (pg_u4){a, b, c, d});
}

// One tile step.  c_cur: this step's exponent tile (16 values per lane); c_nxt: the next step's (chain a_n x b_n, issued here);
// acc_p / x1h, x1l / (wh1, wl1): accumulators, K-step-1 feature operands and K-step-1 words of the PREVIOUS step (wh1, wl1 are
// replaced by this step's on return); acc_c / x0h, x0l: this step's accumulators and K-step-0 feature operands.
// hook(j): called behind matrix slot j (the caller's LDS reads ride there).
template <int KS, int NFB, bool WLO, typename HOOK>
__device__ __forceinline__ void pg_step(pg_f16v &c_nxt, const pg_f16v &c_cur, const pg_h8 (&a_n)[KS], const pg_h8 (&b_n)[KS],
                                        pg_f16v (&acc_p)[NFB], pg_f16v (&acc_c)[NFB], const pg_h8 (&x0h)[NFB],
                                        const pg_h8 (&x0l)[NFB], const pg_h8 (&x1h)[NFB], const pg_h8 (&x1l)[NFB], pg_h8 &wh1,
                                        pg_h8 &wl1, HOOK &&hook) {
    typedef PgSched<KS, NFB, WLO> S;
    float ex[16];
    unsigned hw[8], lw[8];
    const pg_h8 wh1p = wh1, wl1p = wl1;
    pg_h8 wh0, wl0;
    auto matrix = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr PgSlot sl = S::slot(j);
        if constexpr (sl.kind == 0) {
            if constexpr (sl.i == 0) {
                const pg_f16v zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                c_nxt = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_n[0], b_n[0], zero, 0, 0, 0);
            } else {
                c_nxt = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_n[sl.i], b_n[sl.i], c_nxt, 0, 0, 0);
            }
        } else {
            constexpr int fb = S::prod_fb(sl.i), pr = S::prod_pr(sl.i);
            if constexpr (sl.kind == 1)
                acc_p[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 2 ? x1l[fb] : x1h[fb], pr == 1 ? wl1p : wh1p, acc_p[fb], 0, 0, 0);
            else
                acc_c[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 2 ? x0l[fb] : x0h[fb], pr == 1 ? wl0 : wh0, acc_c[fb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        hook(jc);
    };
    auto vector = [&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        pg_unroll(std::make_integer_sequence<int, S::NM>{}, [&](auto jc) __attribute__((always_inline)) {
            if constexpr (S::unit_of_slot(decltype(jc)::value) == u) matrix(jc);
        });
        constexpr PgUnit un = S::unit(u);
        constexpr int i = un.i;
        if constexpr (un.kind == 0) {
            ex[i] = __builtin_amdgcn_exp2f(c_cur[i]);
        } else if constexpr (un.kind == 1) {
            const pg_h2 hi = {(_Float16)ex[2 * i], (_Float16)ex[2 * i + 1]};
            hw[i] = __builtin_bit_cast(unsigned, hi);
            if constexpr (i == 3) wh0 = pg_quad(hw[0], hw[1], hw[2], hw[3]);
        } else if constexpr (un.kind == 2) {
            // e - hi in place, from the f16 halves (one instruction each instead of v_cvt_f32_f16 + v_sub_f32).  The registers were
            // written by the compiler-visible v_exp_f32 just before, so the hazard recognizer (which does not see inside asm) has
            // already cleared them against matrix instructions in flight; the operand words themselves are written by compiler-
            // visible conversions.  ONE statement: the compiler pads every separate asm statement with an s_nop.
            asm("v_fma_mix_f32 %0, %2, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %1, %2, -1.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                : "+v"(ex[2 * i]), "+v"(ex[2 * i + 1]) : "v"(hw[i]));
        } else {
            const pg_h2 lo = {(_Float16)ex[2 * i], (_Float16)ex[2 * i + 1]};
            lw[i] = __builtin_bit_cast(unsigned, lo);
            if constexpr (i == 3) wl0 = pg_quad(lw[0], lw[1], lw[2], lw[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    pg_unroll(std::make_integer_sequence<int, S::NU>{}, vector);
    wh1 = pg_quad(hw[4], hw[5], hw[6], hw[7]);
    wl1 = pg_quad(lw[4], lw[5], lw[6], lw[7]);
}

template <int KS, bool WLO, int G>
__global__ __launch_bounds__(64 * PgCfg<KS>::NW, PgCfg<KS>::WPS) void pg_pass_kernel(const _Float16 *__restrict__ rimg, int row_per_d,
                                                                   const _Float16 *__restrict__ ximg,
                                                                   const _Float16 *__restrict__ cimg, int col_per_d,
                                                                   float *__restrict__ out, int n_row_tiles, int n_col_tiles,
                                                                   int groups_per_d, int NTb, PgPsi2Out po) {
    constexpr int NFB = PgCfg<KS>::NFB, NW = PgCfg<KS>::NW, NF = PG_FB * NFB;
    constexpr int XP = pg_xp(NFB), PIECES = KS + XP, TILE_BYTES = 1024 * PIECES;   // LDS bytes of one row tile: K-steps of the exponent operand, then the features
    typedef PgSched<KS, NFB, WLO> S;
    constexpr int LA = G == 1 ? 2 : 1;                         // row tiles between an LDS read of the exponent operand and its row tile
    extern __shared__ __align__(16) unsigned char smem_raw[];
    typedef __attribute__((address_space(3))) void lds_void;
    // same-d workgroups on one XCD (blocks are dealt round-robin over the 8 XCDs: speed only): they re-read the same row images
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int d = bid / groups_per_d, cg = bid - d * groups_per_d;
    const int t = threadIdx.x, lane = t & 63, l5 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int RING = 3 * NTb;                                    // row-tile slots of the LDS ring: three chunks of NTb
    const unsigned char *rsrc = reinterpret_cast<const unsigned char *>(rimg) + (row_per_d ? (size_t)d * n_row_tiles * KS * 1024 : 0);
    const unsigned char *xsrc = reinterpret_cast<const unsigned char *>(ximg) + (size_t)d * n_row_tiles * XP * 1024;
    const _Float16 *csrc = cimg + (col_per_d ? (size_t)d * n_col_tiles * KS * 64 * 8 : 0);
    // chunk c = row tiles [c NTb, (c + 1) NTb) -> ring slots (c % 3) NTb ...: 1 KB pieces by LDS-DMA (global_load_lds_dwordx4: lane-linear
    // destination, no staging registers), piece i of the chunk by wave i % NW
    auto fill = [&](int c) __attribute__((always_inline)) {
        const int rt0 = c * NTb, ntile = min(NTb, n_row_tiles - rt0);
        unsigned char *base = smem_raw + (size_t)(c % 3) * NTb * TILE_BYTES;
        for (int i = wv; i < ntile * PIECES; i += NW) {           // (wave-uniform)
            const int tl = i / PIECES, pc = i - tl * PIECES;
            const unsigned char *src = pc < KS ? rsrc + ((size_t)(rt0 + tl) * KS + pc) * 1024
                                               : xsrc + ((size_t)(rt0 + tl) * XP + (pc - KS)) * 1024;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const pg_u4 *>(src) + lane, (lds_void *)(base + (size_t)1024 * i), 16, 0, 0);
        }
    };
    // KS = 8, chunks >= 2 (inside the loop): the same pieces through registers (global_load_dwordx4 at the chunk barrier, ds_write_b128
    // one row tile later) — an LDS-DMA piece among the loop's matrix instructions costs the issuing wave ~100 cycles, and with G = 1 there
    // are 16 pieces per row tile and ONE step per row tile to carry them: config 5 4.74 -> 4.58 ms per training step.  KS = 4 (8 pieces
    // per 2-3 steps): 1.3 % SLOWER through registers (24 more of them), so the DMA stays there.
    constexpr bool STAGED = KS >= 8 && G == 1;
    // KS = 8 with G >= 2: ONE register set for the exponent operand — K-step ks of the next row tile is read into the register of K-step
    // ks right behind the last matrix instruction that reads it (the chain of step G - 2), a whole step ahead of its first use
    constexpr bool SINGLE_A = (KS >= 8 && G >= 2) || G >= 4;
    constexpr int NTB_MAX = (160 * 1024 * NW / (4 * PgCfg<KS>::WPS)) / (3 * 1024 * PIECES), MAXP = STAGED ? (NTB_MAX * PIECES + NW - 1) / NW : 1;
    pg_u4 stage[MAXP];
    int st_chunk = -1;
    auto fill_load = [&](int c) __attribute__((always_inline)) {
        const int rt0 = c * NTb, npc = min(NTb, n_row_tiles - rt0) * PIECES;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const int i = wv + k * NW;
            if (i < npc) {
                const int tl = i / PIECES, pc = i - tl * PIECES;
                const unsigned char *src = pc < KS ? rsrc + ((size_t)(rt0 + tl) * KS + pc) * 1024
                                                   : xsrc + ((size_t)(rt0 + tl) * XP + (pc - KS)) * 1024;
                stage[k] = reinterpret_cast<const pg_u4 *>(src)[lane];
            }
        }
        st_chunk = c;
    };
    auto fill_store = [&]() __attribute__((always_inline)) {
        const int c = st_chunk, rt0 = c * NTb, npc = min(NTb, n_row_tiles - rt0) * PIECES;
        unsigned char *base = smem_raw + (size_t)(c % 3) * NTb * TILE_BYTES;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const int i = wv + k * NW;
            if (i < npc) reinterpret_cast<pg_u4 *>(base + (size_t)1024 * i)[lane] = stage[k];
        }
        st_chunk = -1;
    };
    const int n_chunks = (n_row_tiles + NTb - 1) / NTb;
    fill(0);
    if (n_chunks > 1) fill(1);
    pg_h8 bop[G][KS];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int ct = min(cg * NW * G + wv + NW * g, n_col_tiles - 1);   // (a surplus tile of the last group repeats the last one; not stored)
        const pg_h8 *row = reinterpret_cast<const pg_h8 *>(csrc) + (size_t)ct * KS * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bop[g][ks] = row[ks * 64];
    }
    pg_f16v acc[G][NFB];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[g][fb][v] = 0.0f;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();

    // LDS reads of ring slot pos: the exponent operand (KS x 16 bytes per lane) and the feature operands of one K-step half
    auto load_a = [&](pg_h8 (&a)[KS], int pos) __attribute__((always_inline)) {
        const pg_h8 *p = reinterpret_cast<const pg_h8 *>(smem_raw + (size_t)pos * TILE_BYTES) + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[ks] = p[ks * 64];
    };
    auto load_x = [&](pg_h8 (&xh)[NFB], pg_h8 (&xl)[NFB], int pos, int kstep) __attribute__((always_inline)) {
        const pg_h8 *p = reinterpret_cast<const pg_h8 *>(smem_raw + (size_t)pos * TILE_BYTES + 1024 * KS) + lane;
        xh[0] = p[kstep * 64];                                    // block 0: [hi | lo][K-step][64 lanes]
        xl[0] = p[(2 + kstep) * 64];
        if constexpr (NFB == 2) xh[1] = p[(4 + kstep) * 64];      // block 1: its one plane (hi rows 0-15, lo rows 16-31)
    };
    pg_h8 a_0[KS], a_1[KS], x0h[NFB], x0l[NFB], x1h[NFB], x1l[NFB];
    pg_h8 wh1 = pg_quad(0u, 0u, 0u, 0u), wl1 = wh1;             // (the first step's "previous" products add zero)
    pg_f16v c[2];
    load_a(a_0, 0);
    if constexpr (G == 1) load_a(a_1, n_row_tiles > 1 ? 1 : 0);
    load_x(x1h, x1l, 0, 1);
    {
        const pg_f16v zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        c[0] = zero;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_0[ks], bop[0][ks], c[0], 0, 0, 0);
    }
    int pos = 0, in_chunk = 0, chunk = 0;                         // ring slot of row tile rt, its index in its chunk, its chunk
    // one row tile: G steps against the resident column tiles; a_cur / a_nxt swap roles from one row tile to the next
    // G >= 2: a_cur holds this row tile's exponent operand, a_nxt receives the next one's (read one step ahead of its first use);
    // G == 1: a_nxt holds the NEXT row tile's (the chain issued in this step), a_cur receives the one after it.  PAR: parity of the
    // steps in front of this row tile (the two exponent tiles alternate from step to step).
    auto row_tile = [&](auto par, pg_h8 (&a_cur)[KS], pg_h8 (&a_nxt)[KS], int rt) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par)::value;
        if constexpr (STAGED)
            if (st_chunk >= 0) fill_store();                      // (the row tile behind the one that issued the loads)
        if (in_chunk == NTb - LA && chunk + 1 < n_chunks) {
            // LA row tiles in front of the chunk's end: the next chunk is complete in LDS (own pieces: vmcnt, everybody's: barrier),
            // and everybody has left the previous chunk, whose ring slots take the chunk after the next
#ifndef PG_DIAG_NOBAR                                              // (timing experiments only: wrong results)
            __builtin_amdgcn_s_waitcnt(0x0f70);                   // vmcnt(0)
            __syncthreads();
#endif
#ifndef PG_DIAG_NOFILL                                             // (timing experiments only: wrong results)
            if (chunk + 2 < n_chunks) {
                if constexpr (STAGED) fill_load(chunk + 2);
                else fill(chunk + 2);
            }
#endif
        }
        // ring slot of the row tile LA ahead (behind the last row tile: surplus chains on the last one)
        int pos_n = pos;
        if (rt + LA < n_row_tiles) { pos_n = pos + LA; if (pos_n >= RING) pos_n -= RING; }
        else if (LA == 2 && rt + 1 < n_row_tiles) { pos_n = pos + 1 == RING ? 0 : pos + 1; }
        pg_unroll(std::make_integer_sequence<int, G>{}, [&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value, st = PAR * G + g;
            auto hook = [&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                // LDS reads of the row tile, each behind the last matrix instruction that reads the registers it replaces and a
                // few slots ahead of its first use (a read issued just in front of a wait for an OLDER one is waited for as well:
                // lgkmcnt counts in order).  Step 0: this row tile's K-step-0 features behind the first product of the previous
                // step (the previous step's K-step-0 products are all issued), its K-step-1 features behind the first K-step-0
                // product (the previous row tile's K-step-1 products are all issued); the exponent operand one step ahead of
                // its first use.
#ifdef PG_DIAG_NOLDS                                               // (timing experiments only: wrong results)
                if (rt > 1) return;
#endif
                if constexpr (g == 0 && j == S::first_of(1)) load_x(x0h, x0l, pos, 0);
                if constexpr (G == 1) {
                    if constexpr (j == S::first_of(1)) load_a(a_cur, pos_n);
                } else if constexpr (SINGLE_A) {
                    if constexpr (g == G - 2 && S::slot(j).kind == 0)
                        a_cur[S::slot(j).i] = (reinterpret_cast<const pg_h8 *>(smem_raw + (size_t)pos_n * TILE_BYTES) + lane)[S::slot(j).i * 64];
                } else {
                    if constexpr (g == G - 2 && j == S::first_of(1)) load_a(a_nxt, pos_n);
                }
                if constexpr (g == 0 && j == S::J0) load_x(x1h, x1l, pos, 1);
            };
            if constexpr (g + 1 < G)
                pg_step<KS, NFB, WLO>(c[(st + 1) & 1], c[st & 1], a_cur, bop[(g + 1) % G], acc[(g + G - 1) % G], acc[g], x0h, x0l, x1h, x1l, wh1, wl1, hook);
            else
                pg_step<KS, NFB, WLO>(c[(st + 1) & 1], c[st & 1], a_nxt, bop[(g + 1) % G], acc[(g + G - 1) % G], acc[g], x0h, x0l, x1h, x1l, wh1, wl1, hook);
        });
        pos = pos + 1 == RING ? 0 : pos + 1;
        if (++in_chunk == NTb) { in_chunk = 0; ++chunk; }
    };
    typedef std::integral_constant<int, 0> P0;
    typedef std::integral_constant<int, G & 1> P1;
#pragma unroll 1
    for (int rt = 0; rt + 1 < n_row_tiles; rt += 2) {          // (pairs: the two operand buffers swap roles without register moves;
        if constexpr (SINGLE_A) {                                 //  a break between the two made the compiler copy the accumulators)
            row_tile(P0{}, a_0, a_0, rt);
            row_tile(P1{}, a_0, a_0, rt + 1);
        } else {
            row_tile(P0{}, a_0, a_1, rt);
            row_tile(P1{}, a_1, a_0, rt + 1);
        }
    }
    if (n_row_tiles & 1) {
        if constexpr (SINGLE_A) row_tile(P0{}, a_0, a_0, n_row_tiles - 1);
        else row_tile(P0{}, a_0, a_1, n_row_tiles - 1);
    }
    // the last step's K-step-1 half
#pragma unroll
    for (int i = 0; i < S::NK; ++i) {
        const int fb = S::prod_fb(i), pr = S::prod_pr(i);
        acc[G - 1][fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 2 ? x1l[fb] : x1h[fb], pr == 1 ? wl1 : wh1, acc[G - 1][fb], 0, 0, 0);
    }
    // the stacked second block: feature 32 + r = rows r and r + 16 of its result (registers v and v + 8)
    if constexpr (NFB == 2) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int v = 0; v < 8; ++v) acc[g][1][v] += acc[g][1][v + 8];
    }
    // ---- out[d][column][f]: register v of lane (column l5, half) of block fb is feature 32 fb + 8 (v / 4) + 4 half + v % 4 ----
    // po.part != nullptr (pass 1 of a training step): feature po.fsel (the constant 1) is the column sum of the exponentials, i.e.
    // Psi2 of output dim d up to its per-pair factor — written where the forward's psi2 kernel writes it (slab 0 of the partial
    // slabs, entries m' <= m), so the forward needs no psi2 dispatch of its own
    float poison = 0.0f;
    if ((po.part || po.vout) && *po.flag) poison = __builtin_nanf("");
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int tile = cg * NW * G + wv + NW * g;
        if (tile >= n_col_tiles) continue;
        if (po.part || po.vout) {
            const int fi = po.fsel & 31, vs = (fi >> 3) * 4 + (fi & 3), hs = (fi >> 2) & 1;
            float cs = 0.0f;
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                for (int v = 0; v < 16; ++v) cs = (fb == (po.fsel >> 5) && v == vs) ? acc[g][fb][v] : cs;
            const int p = 32 * tile + l5;
            if (po.part) {
                const unsigned pm = po.pmap[p];
                if (half == hs && pm != 0xffffffffu)
                    po.part[(size_t)d * po.Mp * po.Mp + (size_t)(pm >> 16) * po.Mp + (pm & 0xffffu)] =
                        po.scale[(size_t)d * po.Ppad + p] * (cs * (1.0f / 4096.0f)) + poison;      // (x 2^-PG_WSHIFT)
            } else if (half == hs && p < po.vM) {
                po.vout[(size_t)d * po.vM + p] = po.valpha[d] * (double)(cs * (1.0f / 4096.0f) + poison);
            }
        }
        // feature-major inside a column tile (pg_oix): one store instruction = two 128-byte runs, and the finishing kernels (thread =
        // column) read every feature as a coalesced run
        if (po.col_major) {
            float *oc = out + (((size_t)d * n_col_tiles + tile) * 32 + l5) * NF + 4 * half;
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                for (int vq = 0; vq < (fb ? 2 : 4); ++vq)
                    *reinterpret_cast<pg_f4 *>(oc + 32 * fb + 8 * vq) =
                        (pg_f4){acc[g][fb][4 * vq], acc[g][fb][4 * vq + 1], acc[g][fb][4 * vq + 2], acc[g][fb][4 * vq + 3]};
            continue;
        }
        float *o = out + (((size_t)d * n_col_tiles + tile) * NF + 4 * half) * 32 + l5;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
            for (int v = 0; v < (fb ? 8 : 16); ++v) o[(32 * fb + 8 * (v >> 2) + (v & 3)) * 32] = acc[g][fb][v];   // (block 1: features 32 .. 47)
    }
}

// ---- finishing, pair side -----------------------------------------------------------------------------------------------
// thread = pair p, blockIdx.y = chunk of output dims, blockIdx.z = block of four latent dims: partial sums over the chunk's d of
//   tp[c][0][q][p] = sum_d u_dp R2[2q],  tp[c][1][q][p] = sum_d u_dp R2[2q+1],  tp[c][2][q][p] = sum_d u_dp C_dp gamma_dq  (C = R2[2Q])
// (tp: [chunk][3][Q][P]).  Every load is a coalesced run (pg_oix); two output dims per trip so that their loads are in flight
// together; four latent dims per thread keep it at a few dozen registers (a thread with all Q of them needed 200).
template <int NF>
__global__ __launch_bounds__(256) void pg_finish_pairs_kernel(int M, int Q, int D, int Ppad, int dchunk,
                                                              const double *__restrict__ gamma, const float *__restrict__ u,
                                                              const float *__restrict__ r2, double *__restrict__ tp) {
    constexpr int QB = 4, KD = 2;                             // (KD output dims per trip: their 10 KD loads in flight together; 4: slower, 122 registers)
    __shared__ double gsh[32][QB];
    const int t = threadIdx.x, p = blockIdx.x * 256 + t, c = blockIdx.y, q0 = QB * blockIdx.z;
    const int P = (int)((long long)M * (M + 1) / 2);
    const bool ok = p < P;
    const int pp = ok ? p : P - 1;
    double a1[QB], a2[QB], a3[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) { a1[q] = 0.0; a2[q] = 0.0; a3[q] = 0.0; }
    const int d1 = min(D, (c + 1) * dchunk);
    for (int d0 = c * dchunk; d0 < d1; d0 += 32) {
        const int nd = min(32, d1 - d0);
        __syncthreads();
        if (t < nd * QB) {
            const int dd = t / QB, q = t - dd * QB;
            gsh[dd][q] = q0 + q < Q ? gamma[(size_t)(d0 + dd) * Q + q0 + q] : 0.0;
        }
        __syncthreads();
        for (int dd = 0; dd < nd; dd += KD) {
            float ud[KD], cc[KD], ra[KD][QB], rb[KD][QB];
#pragma unroll
            for (int k = 0; k < KD; ++k) {
                const bool on = dd + k < nd;
                const int d = d0 + (on ? dd + k : dd);
                ud[k] = on ? u[(size_t)d * Ppad + pp] * (1.0f / 4096.0f) : 0.0f;          // (x 2^-PG_WSHIFT)
                const float *row = r2 + pg_oix<NF>((size_t)d * Ppad + pp, 0);
                cc[k] = row[64 * Q];
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    const bool lq = q0 + q < Q;
                    ra[k][q] = lq ? row[64 * (q0 + q)] : 0.0f;
                    rb[k][q] = lq ? row[64 * (q0 + q) + 32] : 0.0f;
                }
            }
#pragma unroll
            for (int k = 0; k < KD; ++k) {
                const double udd = (double)ud[k], uc = udd * (double)cc[k];
                const int dl = dd + k < nd ? dd + k : dd;
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    a1[q] += udd * (double)ra[k][q];
                    a2[q] += udd * (double)rb[k][q];
                    a3[q] += uc * gsh[dl][q];
                }
            }
        }
    }
    if (!ok) return;
    double *o = tp + (size_t)c * 3 * P * Q;
#pragma unroll
    for (int q = 0; q < QB; ++q)
        if (q0 + q < Q) {
            o[(size_t)(q0 + q) * P + p] = a1[q];
            o[((size_t)Q + q0 + q) * P + p] = a2[q];
            o[((size_t)2 * Q + q0 + q) * P + p] = a3[q];
        }
}
// block = inducing point m, thread = (latent dim q = t % 32, group of the other ends o = t / 32 + 8 i):
//   dz[m][q] += sum over the pairs that hold m of  ln2 (2 S2 s_pq t[0] + t[1]) -+ 1/2 delta_pq t[2]          (tt: [3][Q][P])
__global__ __launch_bounds__(256) void pg_gather_dz_kernel(int M, int Q, const double *__restrict__ z,
                                                           const unsigned char *__restrict__ consts,
                                                           const double *__restrict__ tt, double *__restrict__ dz) {
    __shared__ double red[8][32];
    const int m = blockIdx.x, t = threadIdx.x, q = t & 31, og = t >> 5;
    const size_t P = (size_t)M * (M + 1) / 2;
    double acc = 0.0;
    if (q < Q) {
        const double c = (double)reinterpret_cast<const float *>(consts)[q];     // (the centring constant the images were built with)
        const double zm = z[(size_t)m * Q + q];
        for (int o = og; o < M; o += 8) {
            const int hi = o > m ? o : m, lo = o > m ? m : o;
            const size_t p = (size_t)hi * (hi + 1) / 2 + lo;
            const double zo = z[(size_t)o * Q + q];
            const double sp = (zm - c) + (zo - c);
            const double d1 = 0.6931471805599453 * (2.0 * (double)PSI2_PAIR_S2_SCALE * sp * tt[(size_t)q * P + p] + tt[((size_t)Q + q) * P + p]);
            const double d2 = -0.5 * (zm - zo) * tt[((size_t)2 * Q + q) * P + p];   // (delta is antisymmetric in (m, o): one formula for both ends)
            acc += d1 + d2;
            if (o == m) acc += d1;
        }
    }
    red[og][q] = acc;
    __syncthreads();
    if (t < Q) {
        double a = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) a += red[g][t];
        dz[(size_t)m * Q + t] += a;
    }
}
// block = output dim d: dgamma[d][q] += sum_p -1/4 delta_pq^2 u_dp C_dp
template <int NF>
__global__ __launch_bounds__(256) void pg_dgamma_pairs_kernel(int M, int Q, int Ppad, const double *__restrict__ z,
                                                              const float *__restrict__ u, const float *__restrict__ r2,
                                                              double *__restrict__ dgamma) {
    __shared__ double red[256];
    const int d = blockIdx.x, t = threadIdx.x;
    const int P = (int)((long long)M * (M + 1) / 2);
    double a[DPGP_MAX_Q];
#pragma unroll
    for (int q = 0; q < DPGP_MAX_Q; ++q) a[q] = 0.0;
    for (int p = t; p < P; p += 256) {
        int m, mp;
        psi2_pair_of(p, m, mp);
        const double uc = (double)u[(size_t)d * Ppad + p] * (double)r2[pg_oix<NF>((size_t)d * Ppad + p, 2 * Q)] * (1.0 / 4096.0);
#pragma unroll
        for (int q = 0; q < DPGP_MAX_Q; ++q)
            if (q < Q) {
                const double dd = z[(size_t)m * Q + q] - z[(size_t)mp * Q + q];
                a[q] += -0.25 * dd * dd * uc;
            }
    }
#pragma unroll
    for (int q = 0; q < DPGP_MAX_Q; ++q) {
        if (q < Q) {
            red[t] = a[q];
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) {
                if (t < o) red[t] += red[t + o];
                __syncthreads();
        }
        if (t == 0) dgamma[(size_t)d * Q + q] += red[0];
        __syncthreads();
        }
    }
}

// ---- finishing, observation side ------------------------------------------------------------------------------------------
// The chain rule of one (output dim d, observation n) from its row of a pass with rows = pairs / inducing points and columns =
// observations: (ra, rb, rc) = the a-, b- and constant-feature sums, scaled.  src 0: the Psi2 term (den = 2 g s + 1); src 1: the Psi1
// term on the diagonal pairs (den = g s + 1, half the coefficients — carried by `ik` —, the row weighted by y_nd).
// one latent dim q of that chain rule: adds to d/dmu', d/dS and d/dgamma_q
__device__ __forceinline__ void pg_obs_chain_q(double ra, double rb, double rc, double g, double mc, double sv, int src, double &am,
                                               double &as_, double &dg) {
    const double df = src ? 1.0 : 2.0;
    // (one fp32 reciprocal: three fp64 divisions per (d, n, q) were most of this kernel's time — ~40 instructions each)
    const double id = (double)(1.0f / (float)(df * g * sv + 1.0)), w = g * id;
    const double dw = (-0.25 / (double)PSI2_PAIR_S2_SCALE) * ra + mc * rb - mc * mc * rc;
    const double dmc = w * (rb - 2.0 * mc * rc);
    // (the 1/2 log2 den of the row constant carries no "half": undo it for the Psi1 rows, whose scale holds it)
    const double dden = -0.5 * rc * (src ? 2.0 : 1.0) * id - w * id * dw;
    am += dmc;
    as_ += df * g * dden;
    dg += dw * id + df * sv * dden;
}
// four values per lane summed over the 64 lanes of a wave, the value set halved in the first two butterfly steps (7 shuffles instead
// of 24): lane L returns the total of value L / 16 (the 16 lanes of one value hold copies)
__device__ __forceinline__ double pg_wave_reduce4(const double (&v)[4], int lane) {
    const bool u5 = (lane & 32) != 0, u4 = (lane & 16) != 0;
    const double w0 = (u5 ? v[2] : v[0]) + __shfl_xor(u5 ? v[0] : v[2], 32, 64);
    const double w1 = (u5 ? v[3] : v[1]) + __shfl_xor(u5 ? v[1] : v[3], 32, 64);
    double w = (u4 ? w1 : w0) + __shfl_xor(u4 ? w0 : w1, 16, 64);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) w += __shfl_xor(w, off, 64);
    return w;
}
// thread = observation n, blockIdx.y = chunk of output dims, blockIdx.z = block of four latent dims.  dmu_part / ds_part
// [chunk][N][Q] = partial sums over the chunk's d (registers, no cross-thread reduction); dg_part [4 blockIdx.x + wave][D][Q] = the
// wave's sum over its 64 observations of the d/dgamma_dq terms.  r1p / kap1 / y (optional): the Psi1 term's rows in the same sweep.
// Every feature of a row is a coalesced run (pg_oix); gamma and the scales of eight output dims at a time through LDS; four latent
// dims per thread: ~60 registers and five times the workgroups of a thread that carries all Q (246 registers, one wave per SIMD).
template <int NF>
__global__ __launch_bounds__(256) void pg_finish_obs_kernel(int N, int Q, int D, int NT, int dchunk,
                                                            const unsigned char *__restrict__ consts,
                                                            const double *__restrict__ mu, const double *__restrict__ s,
                                                            const double *__restrict__ gamma, const float *__restrict__ kap,
                                                            const float *__restrict__ r1, const float *__restrict__ kap1,
                                                            const float *__restrict__ r1p, const double *__restrict__ y, int ldy,
                                                            double *__restrict__ dmu_part, double *__restrict__ ds_part,
                                                            double *__restrict__ dg_part) {
    constexpr int QB = 4;
    __shared__ double gsh[8][QB];
    __shared__ double iksh[8][2];
    const int t = threadIdx.x, n = blockIdx.x * 256 + t, c = blockIdx.y, q0 = QB * blockIdx.z, lane = t & 63, wv = t >> 6;
    const bool ok = n < N;
    const int nn = ok ? n : N - 1;
    double mc[QB], sv[QB], am[QB], as_[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const bool lq = q0 + q < Q;
        mc[q] = lq ? mu[(size_t)nn * Q + q0 + q] - (double)reinterpret_cast<const float *>(consts)[q0 + q] : 0.0;
        sv[q] = lq ? s[(size_t)nn * Q + q0 + q] : 1.0;
        am[q] = 0.0; as_[q] = 0.0;
    }
    double *dgw = dg_part + ((size_t)(4 * blockIdx.x + wv) * D) * Q;
    const int d1 = min(D, (c + 1) * dchunk);
    for (int d0 = c * dchunk; d0 < d1; d0 += 8) {
        const int nd = min(8, d1 - d0);
        __syncthreads();
        if (t < nd * QB) {
            const int dd = t / QB, q = t - dd * QB;
            gsh[dd][q] = q0 + q < Q ? gamma[(size_t)(d0 + dd) * Q + q0 + q] : 0.0;
        }
        if (t >= 64 && t - 64 < nd) {
            iksh[t - 64][0] = 1.0 / ((double)kap[d0 + t - 64] * 4096.0);
            iksh[t - 64][1] = r1p ? 0.5 / ((double)kap1[d0 + t - 64] * 4096.0) : 0.0;
        }
        __syncthreads();
        // rows of output dim dd + 1 are fetched while those of dd are worked on (two register sets, the loop unrolled by two)
        struct Rows { float ra[2][QB], rb[2][QB], rcf[2]; double yv; };
        auto fetch = [&](Rows &R, int dd) __attribute__((always_inline)) {
            const int d = d0 + (dd < nd ? dd : nd - 1);
            R.yv = 0.0; R.rcf[0] = 0.0f; R.rcf[1] = 0.0f;
#pragma unroll
            for (int src = 0; src < 2; ++src) {
                if (src && !r1p) break;
                const float *row = (src ? r1p : r1) + pg_oix<NF>((size_t)d * NT * 32 + nn, 0);
                R.rcf[src] = row[64 * Q];
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    const bool lq = q0 + q < Q;
                    R.ra[src][q] = lq ? row[64 * (q0 + q)] : 0.0f;
                    R.rb[src][q] = lq ? row[64 * (q0 + q) + 32] : 0.0f;
                }
                if (src) R.yv = y[(size_t)nn * ldy + d];
            }
        };
        auto work = [&](const Rows &R, int dd) __attribute__((always_inline)) {
            const int d = d0 + dd;
            double dg[QB];
#pragma unroll
            for (int q = 0; q < QB; ++q) dg[q] = 0.0;
#pragma unroll
            for (int src = 0; src < 2; ++src) {
                if (src && !r1p) break;
                const double ik = src ? iksh[dd][1] * R.yv : iksh[dd][0];
                const double rc = (double)R.rcf[src] * ik;
#pragma unroll
                for (int q = 0; q < QB; ++q)
                    if (q0 + q < Q)
                        pg_obs_chain_q((double)R.ra[src][q] * ik, (double)R.rb[src][q] * ik, rc, gsh[dd][q], mc[q], sv[q], src, am[q], as_[q], dg[q]);
            }
            if (!ok) {
#pragma unroll
                for (int q = 0; q < QB; ++q) dg[q] = 0.0;
            }
            const double tot = pg_wave_reduce4(dg, lane);
            const int qi = q0 + (lane >> 4);
            if ((lane & 15) == 0 && qi < Q) dgw[(size_t)d * Q + qi] = tot;
        };
        Rows Ra, Rb;
        fetch(Ra, 0);
        for (int dd = 0; dd < nd; dd += 2) {
            if (dd + 1 < nd) fetch(Rb, dd + 1);
            work(Ra, dd);
            if (dd + 1 < nd) {
                if (dd + 2 < nd) fetch(Ra, dd + 2);
                work(Rb, dd + 1);
            }
        }
    }
    if (ok) {
#pragma unroll
        for (int q = 0; q < QB; ++q)
            if (q0 + q < Q) {
                dmu_part[((size_t)c * N + n) * Q + q0 + q] = am[q];
                ds_part[((size_t)c * N + n) * Q + q0 + q] = as_[q];
            }
    }
}

// a range-guard hit anywhere: the outputs become NaN (never a silently wrong gradient)
__global__ void pg_poison_kernel(const int *__restrict__ flag, double *dmu, double *ds, double *dz, double *dgamma) {
    if (*flag) {
        const double nan = (double)__builtin_nanf("");
        dmu[0] = nan; ds[0] = nan; dz[0] = nan; dgamma[0] = nan;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
bool psi2_pgrad_supported(int M, int Q) {
    const int ks = psi2_pairs_ksteps(Q);
    return ks <= 8 && 2 * Q + 1 <= (pg_nfb(ks) == 2 ? 48 : PG_FB) && M >= 1 && M <= 4096;   // (a second block holds 16 features: pg_xp)
}

template <int KS> static int pg_ring_tiles() {                   // row tiles per chunk of the pass kernel's LDS ring
    // three chunks of NTb row tiles (KS + pg_xp KB each) in the workgroup's share of the 160 KB
    int NTb = (int)((size_t)(160 * 1024 * PgCfg<KS>::NW / (4 * PgCfg<KS>::WPS)) / ((size_t)3 * 1024 * (KS + pg_xp(PgCfg<KS>::NFB))));
    if (const char *e = getenv("DPGP_PG_NTB")) {                 // (experiments only)
        const int v = atoi(e);
        if (v >= (PgCfg<KS>::G == 1 ? 2 : 1) && v <= NTb) NTb = v;
    }
    return NTb;
}
// finishing kernels: per-thread arrays of QP = 4 ceil(Q / 4) entries (Q <= 20 here; arrays of DPGP_MAX_Q doubles spill)
#define PG_QP_SWITCH(Q, CALL)                                   \
    switch (((Q) + 3) / 4) {                                    \
        case 1: { constexpr int QP_ = 4; CALL; } break;         \
        case 2: { constexpr int QP_ = 8; CALL; } break;         \
        case 3: { constexpr int QP_ = 12; CALL; } break;        \
        case 4: { constexpr int QP_ = 16; CALL; } break;        \
        default: { constexpr int QP_ = 20; CALL; } break;       \
    }
#define PG_DC_PAIRS 16               // chunks of output dims of the finishing kernels
#define PG_DC_OBS 64
struct PgLayout {
    int KS, NF, P, Ppad, NT, PT, nblk_obs, MT;
    size_t off_u, off_kap, off_flag, off_cobs, off_xobs, off_xpair, off_r2, off_r1, off_tp, off_tt, off_dmup, off_dsp, off_dgp,
        off_cobs1, off_xobs1, off_dimg, off_xm1, off_r1p, off_r2p, off_u1, off_kap1, total;   // (..1 / ..p: the Psi1 term)
};
static PgLayout pg_layout(int D, int N, int M, int Q) {
    PgLayout L;
    const Psi2Consts C = psi2_consts_layout(M, Q);
    L.KS = C.KS; L.P = C.P; L.Ppad = C.Ppad; L.NT = dpgp_ceil_div(N, 32); L.PT = C.Ppad / 32;
    L.NF = PG_FB * pg_nfb(C.KS);
    L.nblk_obs = dpgp_ceil_div(N, 256);
    const size_t h = sizeof(_Float16), xp = (size_t)pg_xp(pg_nfb(C.KS));
    size_t o = 0;
    L.off_u = o;     o += dpgp_align256(sizeof(float) * (size_t)D * L.Ppad);
    L.off_kap = o;   o += dpgp_align256(sizeof(float) * (size_t)D);
    L.off_flag = o;  o += 256;
    L.off_cobs = o;  o += dpgp_align256(h * (size_t)D * L.NT * L.KS * 64 * 8);
    L.off_xobs = o;  o += dpgp_align256(h * (size_t)D * L.NT * 512 * xp);
    L.off_xpair = o; o += dpgp_align256(h * (size_t)D * L.PT * 512 * xp);
    L.off_r2 = o;    o += dpgp_align256(sizeof(float) * (size_t)D * L.Ppad * L.NF);
    L.off_r1 = o;    o += dpgp_align256(sizeof(float) * (size_t)D * L.NT * 32 * L.NF);
    L.off_tp = o;    o += dpgp_align256(sizeof(double) * (size_t)PG_DC_PAIRS * 3 * L.P * Q);
    L.off_tt = o;    o += dpgp_align256(sizeof(double) * (size_t)3 * L.P * Q);
    L.off_dmup = o;  o += dpgp_align256(sizeof(double) * (size_t)PG_DC_OBS * N * Q);
    L.off_dsp = o;   o += dpgp_align256(sizeof(double) * (size_t)PG_DC_OBS * N * Q);
    L.off_dgp = o;   o += dpgp_align256(sizeof(double) * (size_t)4 * L.nblk_obs * D * Q);
    L.MT = dpgp_ceil_div(M, 32);
    L.off_cobs1 = o; o += dpgp_align256(h * (size_t)D * L.NT * L.KS * 64 * 8);
    L.off_xobs1 = o; o += dpgp_align256(h * (size_t)D * L.NT * 512 * xp);
    L.off_dimg = o;  o += dpgp_align256(h * (size_t)L.MT * L.KS * 64 * 8);
    L.off_xm1 = o;   o += dpgp_align256(h * (size_t)D * L.MT * 512 * xp);
    L.off_r1p = o;   o += dpgp_align256(sizeof(float) * (size_t)D * L.NT * 32 * L.NF);
    L.off_r2p = o;   o += dpgp_align256(sizeof(float) * (size_t)D * L.MT * 32 * L.NF);
    L.off_u1 = o;    o += dpgp_align256(sizeof(float) * (size_t)D * L.MT * 32);
    L.off_kap1 = o;  o += dpgp_align256(sizeof(float) * (size_t)D);
    L.total = o;
    return L;
}
size_t psi2_pgrad_ws_bytes(int D, int N, int M, int Q) { return psi2_pgrad_supported(M, Q) ? pg_layout(D, N, M, Q).total : 0; }

// One pass launch.  Resident column tiles per wave: PgCfg<KS>::G, or (KS <= 4, where the registers allow up to four) the one of
// {2, 3, 4} with the least (tile slots x cost per step) — 258 pair tiles: 11 groups of 24 instead of 17 of 16; 63 observation tiles: 2
// groups of 32 rather than 4 of 16 (half the LDS-DMA pieces per step and half the re-reads of the pair image).
template <int KS, bool WLO, int G>
static int pg_launch_pass_g(int D, const _Float16 *rimg, int row_per_d, const _Float16 *ximg, const _Float16 *cimg, int col_per_d,
                            float *out, int n_row_tiles, int n_col_tiles, const PgPsi2Out &po, hipStream_t st) {
    constexpr int NFB = PgCfg<KS>::NFB, NW = PgCfg<KS>::NW;
    const int NTb = pg_ring_tiles<KS>();
    const size_t lds = (size_t)3 * NTb * 1024 * (KS + pg_xp(NFB));
    auto kern = pg_pass_kernel<KS, WLO, G>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DPGP_ERR_LAUNCH;
    const int groups = dpgp_ceil_div(n_col_tiles, NW * G);
    const long long nwg = (long long)D * groups;
    if (nwg > 0x7fffffffLL) return -1;
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(64 * NW), lds, st, rimg, row_per_d, ximg, cimg, col_per_d, out, n_row_tiles, n_col_tiles,
                       groups, NTb, po);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
static int pg_cu_count() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ncu = n;
        else ncu = 256;
    }
    return ncu;
}
template <int KS, bool WLO>
static int pg_launch_pass(int D, const _Float16 *rimg, int row_per_d, const _Float16 *ximg, const _Float16 *cimg, int col_per_d, float *out,
                          int n_row_tiles, int n_col_tiles, const PgPsi2Out &po, hipStream_t st) {
    constexpr int G0 = PgCfg<KS>::G, NW = PgCfg<KS>::NW;
    // cost of a pass with G resident column tiles per wave: rounds of workgroups (one per compute unit) x G tile steps per row tile x the
    // measured cost of a step (KS <= 4, config 3: G = 3 and 4 — the latter with one register set for the exponent operand — take 0.945 /
    // 0.94 of a G = 2 step: a third / half the LDS-DMA pieces per step; KS = 8: G = 2 takes 0.92 of a G = 1 step).  The rounds count the
    // surplus tile slots of the last group of a row AND an under-filled chip: at D = 64 the 63 observation tiles in two groups of 32 are
    // 128 workgroups on 256 compute units (config 2: 380 us where four groups of 16 take 270)
    const int ncu = pg_cu_count();
    auto cost = [&](int G, int permille) {
        const long long nwg = (long long)D * dpgp_ceil_div(n_col_tiles, NW * G);
        return ((nwg + ncu - 1) / ncu) * G * permille;
    };
    if constexpr (KS <= 4 && G0 == 2) {
        const long long c2 = cost(2, 1000), c3 = cost(3, 945), c4 = cost(4, 940);
        const char *e = getenv("DPGP_PG_G");                       // (experiments: 2, 3 or 4)
        const int pick = e ? atoi(e) : (c4 < c3 && c4 < c2 ? 4 : (c3 < c2 ? 3 : 2));
        if (pick == 4) return pg_launch_pass_g<KS, WLO, 4>(D, rimg, row_per_d, ximg, cimg, col_per_d, out, n_row_tiles, n_col_tiles, po, st);
        if (pick == 3) return pg_launch_pass_g<KS, WLO, 3>(D, rimg, row_per_d, ximg, cimg, col_per_d, out, n_row_tiles, n_col_tiles, po, st);
    }
    if constexpr (KS >= 8 && G0 == 1) {
        // two resident column tiles per wave (one register set for the exponent operand: 248 registers) halve the ring fills and the LDS
        // reads per step
        if (cost(2, 920) <= cost(1, 1000) && !getenv("DPGP_PG_G1"))   // (DPGP_PG_G1: experiments)
            return pg_launch_pass_g<KS, WLO, 2>(D, rimg, row_per_d, ximg, cimg, col_per_d, out, n_row_tiles, n_col_tiles, po, st);
    }
    return pg_launch_pass_g<KS, WLO, G0>(D, rimg, row_per_d, ximg, cimg, col_per_d, out, n_row_tiles, n_col_tiles, po, st);
}

// The half of the Psi1 term that does not depend on the adjoints: its observation images (den = g s + 1, half coefficients, features
// weighted by y_nd), the diagonal rows of the pair image, and the pass with rows = observations, columns = inducing points -> R2'.
// psi1v != nullptr (training step): the same pass's y-weighted constant feature gives Psi1^T y of the forward pass, [D][M] — the
// forward's Psi1^T y launch is not needed (as Psi2 out of pass 1).
template <int KS>
static int launch_psi1_front(int D, int N, int M, int Q, const unsigned char *consts, const double *mu, const double *s,
                             const double *gamma, const double *alpha, unsigned char *ws, const double *y, int ldy, double *psi1v,
                             hipStream_t st, bool wlo) {
    constexpr int NFB = PgCfg<KS>::NFB;
    const PgLayout L = pg_layout(D, N, M, Q);
    const Psi2Consts C = psi2_consts_layout(M, Q);
    int *flag = reinterpret_cast<int *>(ws + L.off_flag);
    const _Float16 *pimg = reinterpret_cast<const _Float16 *>(consts + C.off_pairs);
    _Float16 *cobs1 = reinterpret_cast<_Float16 *>(ws + L.off_cobs1), *xobs1 = reinterpret_cast<_Float16 *>(ws + L.off_xobs1),
             *dimg = reinterpret_cast<_Float16 *>(ws + L.off_dimg);
    float *r2p = reinterpret_cast<float *>(ws + L.off_r2p);
    const int Mpad = 32 * L.MT;
    {
        const size_t lds = 256 + sizeof(_Float16) * 8 * 512 * pg_xp(NFB);
        auto kern = pg_obs_images_kernel<KS>;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return DPGP_ERR_LAUNCH;
        DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3(dpgp_ceil_div(N, 256), D), dim3(256), lds, st, N, Q, consts, mu, s, gamma, cobs1, xobs1,
                           L.NT, flag, y, ldy);
        DPGP_LAUNCH_CHECK();
    }
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((pg_diag_image_kernel<KS>), dim3(dpgp_ceil_div(Mpad * KS * 2, 256)), dim3(256), 0, st, M, Mpad, pimg, dimg);
    DPGP_LAUNCH_CHECK();
    // rows: the observations of output dim d (cobs1, y-weighted features xobs1); columns: the inducing points -> R2' [d][m][.]
    // (the pass that also yields Psi1^T y keeps the lo half of the exponentials whatever the mode, as the one that yields Psi2)
    PgPsi2Out po = {nullptr, nullptr, nullptr, flag, 0, 0, 2 * Q, 1, psi1v, alpha, M};
    if (wlo || psi1v) return pg_launch_pass<KS, true>(D, cobs1, 1, xobs1, dimg, 0, r2p, L.NT, L.MT, po, st);
    return pg_launch_pass<KS, false>(D, cobs1, 1, xobs1, dimg, 0, r2p, L.NT, L.MT, po, st);
}

// part 1 (does not depend on the adjoints): observation images, pass 1 -> R2 [and Psi2: psi2_part != nullptr]
template <int KS>
static int launch_pgrad_part1(int D, int N, int M, int Q, const unsigned char *consts, const double *mu, const double *s,
                              const double *gamma, unsigned char *ws, float *psi2_part, const float *scale, hipStream_t st, bool wlo) {
    constexpr int NFB = PgCfg<KS>::NFB;
    const PgLayout L = pg_layout(D, N, M, Q);
    const Psi2Consts C = psi2_consts_layout(M, Q);
    int *flag = reinterpret_cast<int *>(ws + L.off_flag);
    _Float16 *cobs = reinterpret_cast<_Float16 *>(ws + L.off_cobs), *xobs = reinterpret_cast<_Float16 *>(ws + L.off_xobs);
    float *r2 = reinterpret_cast<float *>(ws + L.off_r2);
    const _Float16 *pimg = reinterpret_cast<const _Float16 *>(consts + C.off_pairs);
    if (hipMemsetAsync(flag, 0, sizeof(int), st) != hipSuccess) return DPGP_ERR_LAUNCH;
    {
        const size_t lds = 256 + sizeof(_Float16) * 8 * 512 * pg_xp(NFB);
        auto kern = pg_obs_images_kernel<KS>;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return DPGP_ERR_LAUNCH;
        DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3(dpgp_ceil_div(N, 256), D), dim3(256), lds, st, N, Q, consts, mu, s, gamma, cobs, xobs, L.NT,
                           flag, (const double *)nullptr, 0);
        DPGP_LAUNCH_CHECK();
    }
    PgPsi2Out po = {psi2_part, scale, reinterpret_cast<const unsigned *>(consts + C.off_pmap), flag, dpgp_round_up(M, 16), L.Ppad, 2 * Q, 0, nullptr, nullptr, 0};
    // rows: the observations of output dim d (cobs, xobs); columns: the pairs (the forward's pair image)
    // (the pass that also yields Psi2 keeps the lo half of the exponentials whatever the mode: the objective is not a "fast" quantity)
    if (wlo || psi2_part) return pg_launch_pass<KS, true>(D, cobs, 1, xobs, pimg, 0, r2, L.NT, L.PT, po, st);
    return pg_launch_pass<KS, false>(D, cobs, 1, xobs, pimg, 0, r2, L.NT, L.PT, po, st);
}

// part 2: u_dp from the adjoint of Psi2, pair features, pass 2 -> R1, finishing kernels.  scale != nullptr: the forward's
// per-pair factors (then u is one multiplication per pair; otherwise it is formed from z, gamma, alpha)
template <int KS>
static int launch_pgrad_part2(int D, int N, int M, int Q, const unsigned char *consts, const double *z, const double *mu,
                              const double *s, const double *gamma, const double *alpha, const double *GP, const float *scale,
                              const float *psi2, unsigned char *ws, double *dmu, double *ds, double *dz, double *dgamma,
                              hipStream_t st, const double *y, int ldy, const double *Gv, bool wlo, bool psi1_front_done) {
    constexpr int NFB = PgCfg<KS>::NFB, NF = PG_FB * NFB;
    const PgLayout L = pg_layout(D, N, M, Q);
    const Psi2Consts C = psi2_consts_layout(M, Q);
    const int Mp = dpgp_round_up(M, 16);
    float *u = reinterpret_cast<float *>(ws + L.off_u), *kap = reinterpret_cast<float *>(ws + L.off_kap);
    int *flag = reinterpret_cast<int *>(ws + L.off_flag);
    _Float16 *cobs = reinterpret_cast<_Float16 *>(ws + L.off_cobs), *xpair = reinterpret_cast<_Float16 *>(ws + L.off_xpair);
    float *r2 = reinterpret_cast<float *>(ws + L.off_r2), *r1 = reinterpret_cast<float *>(ws + L.off_r1);
    double *tp = reinterpret_cast<double *>(ws + L.off_tp), *tt = reinterpret_cast<double *>(ws + L.off_tt);
    double *dmup = reinterpret_cast<double *>(ws + L.off_dmup),
           *dsp = reinterpret_cast<double *>(ws + L.off_dsp);
    const _Float16 *pimg = reinterpret_cast<const _Float16 *>(consts + C.off_pairs);
    DPGP_PRELAUNCH();
    if (scale)
    {
        const int zl = (size_t)M * Q * sizeof(double) <= 32 * 1024 ? M * Q : 0;
        hipLaunchKernelGGL(pg_u_scale_kernel, dim3(D), dim3(1024), sizeof(double) * (size_t)zl, st, L.Ppad, Mp, Q,
                           reinterpret_cast<const unsigned *>(consts + C.off_pmap), scale, GP, u, kap, psi2, z, dgamma, zl);
    }
    else
        hipLaunchKernelGGL(pg_u_kernel, dim3(D), dim3(256), 0, st, M, Q, Mp, z, gamma, alpha, GP, u, kap);
    DPGP_LAUNCH_CHECK();
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((pg_pair_images_kernel<KS>), dim3(dpgp_ceil_div(L.Ppad, 256), D), dim3(256), 0, st, L.Ppad, Q, pimg,
                       (const float *)u, (const float *)kap, xpair);
    DPGP_LAUNCH_CHECK();
    int rc;
    {
        PgPsi2Out po = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr, 0};
        // rows: the pairs (pair image, xpair of output dim d); columns: the observations of output dim d
        rc = wlo ? pg_launch_pass<KS, true>(D, pimg, 0, xpair, cobs, 1, r1, L.PT, L.NT, po, st)
                 : pg_launch_pass<KS, false>(D, pimg, 0, xpair, cobs, 1, r1, L.PT, L.NT, po, st);
        if (rc != DPGP_OK) return rc;
    }
    const int dcp = dpgp_ceil_div(D, PG_DC_PAIRS), ncp = dpgp_ceil_div(D, dcp);
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL((pg_finish_pairs_kernel<NF>), dim3(dpgp_ceil_div(L.P, 256), ncp, dpgp_ceil_div(Q, 4)), dim3(256), 0, st, M, Q, D, L.Ppad, dcp,
                       gamma, (const float *)u, (const float *)r2, tp);
    DPGP_LAUNCH_CHECK();
    const size_t n3 = (size_t)3 * L.P * Q;
    rc = launch_reduce_rows<double>(n3, n3, ncp, tp, tt, 0, nullptr, st);
    if (rc != DPGP_OK) return rc;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(pg_gather_dz_kernel, dim3(M), dim3(256), 0, st, M, Q, z, consts, (const double *)tt, dz);
    DPGP_LAUNCH_CHECK();
    if (!(scale && psi2)) {          // (training step: formed by pg_u_scale_kernel from Psi2 itself)
        DPGP_PRELAUNCH(); hipLaunchKernelGGL((pg_dgamma_pairs_kernel<NF>), dim3(D), dim3(256), 0, st, M, Q, L.Ppad, z, (const float *)u,
                           (const float *)r2, dgamma);
        DPGP_LAUNCH_CHECK();
    }
    // ---- the Psi1 term (y != nullptr) through the same pass kernel, on the diagonal pairs: its own observation images (den = g s + 1,
    // half coefficients, features weighted by y_nd), the pair image's diagonal rows, features alpha_d g_v[d][m] (s^2 / 64, s, 1)
    float *r1p = reinterpret_cast<float *>(ws + L.off_r1p), *kap1 = reinterpret_cast<float *>(ws + L.off_kap1);
    if (y) {
        _Float16 *cobs1 = reinterpret_cast<_Float16 *>(ws + L.off_cobs1), *dimg = reinterpret_cast<_Float16 *>(ws + L.off_dimg),
                 *xm1 = reinterpret_cast<_Float16 *>(ws + L.off_xm1);
        float *r2p = reinterpret_cast<float *>(ws + L.off_r2p), *u1 = reinterpret_cast<float *>(ws + L.off_u1);
        const int Mpad = 32 * L.MT;
        // (training step: images, diagonal rows and the pass over the observations ran in part 1 and gave the forward its Psi1^T y)
        if (!psi1_front_done) {
            rc = launch_psi1_front<KS>(D, N, M, Q, consts, mu, s, gamma, alpha, ws, y, ldy, nullptr, st, wlo);
            if (rc != DPGP_OK) return rc;
        }
        DPGP_PRELAUNCH(); hipLaunchKernelGGL(pg_u1_kernel, dim3(D), dim3(64), 0, st, M, Mp, Mpad, alpha, Gv, u1, kap1);
        DPGP_LAUNCH_CHECK();
        DPGP_PRELAUNCH(); hipLaunchKernelGGL((pg_pair_images_kernel<KS>), dim3(dpgp_ceil_div(Mpad, 256), D), dim3(256), 0, st, Mpad, Q,
                           (const _Float16 *)dimg, (const float *)u1, (const float *)kap1, xm1);
        DPGP_LAUNCH_CHECK();
        PgPsi2Out po = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr, 0};
        // rows: the inducing points (dimg, features xm1 of output dim d); columns: the observations of output dim d -> R1' [d][n][.]
        rc = wlo ? pg_launch_pass<KS, true>(D, dimg, 0, xm1, cobs1, 1, r1p, L.MT, L.NT, po, st)
                 : pg_launch_pass<KS, false>(D, dimg, 0, xm1, cobs1, 1, r1p, L.MT, L.NT, po, st);
        if (rc != DPGP_OK) return rc;
        DPGP_PRELAUNCH();
        PG_QP_SWITCH(Q, hipLaunchKernelGGL((pg_finish_m1_kernel<NF, QP_>), dim3(M), dim3(256), 0, st, M, Mpad, Q, D, z, consts, (const float *)u1,
                                           (const float *)r2p, dz));
        DPGP_LAUNCH_CHECK();
    }
    const int dco = dpgp_ceil_div(D, PG_DC_OBS), nco = dpgp_ceil_div(D, dco);
    const float *r1p_ = y ? (const float *)r1p : (const float *)nullptr;
    double *dgp = reinterpret_cast<double *>(ws + L.off_dgp);
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL((pg_finish_obs_kernel<NF>), dim3(L.nblk_obs, nco, dpgp_ceil_div(Q, 4)), dim3(256), 0, st, N, Q, D, L.NT, dco, consts, mu, s,
                       gamma, (const float *)kap, (const float *)r1, (const float *)kap1, r1p_, y, ldy, dmup, dsp, dgp);
    DPGP_LAUNCH_CHECK();
    const size_t dq = (size_t)D * Q;
    rc = launch_reduce_rows<double>(dq, dq, 4 * L.nblk_obs, dgp, dgamma, 1, nullptr, st);
    const size_t nq = (size_t)N * Q;
    // (with the Psi1 term in the sweep nothing has written dmu, ds before: overwrite; otherwise launch_psi1_grad has: add)
    if (rc == DPGP_OK) rc = launch_reduce_rows<double>(nq, nq, nco, dmup, dmu, y ? 0 : 1, nullptr, st);
    if (rc == DPGP_OK) rc = launch_reduce_rows<double>(nq, nq, nco, dsp, ds, y ? 0 : 1, nullptr, st);
    if (rc != DPGP_OK) return rc;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(pg_poison_kernel, dim3(1), dim3(1), 0, st, (const int *)flag, dmu, ds, dz, dgamma);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

// Psi2 part of stage B, ADDED to dmu [N,Q], ds [N,Q], dz [M,Q], dgamma [D,Q]; GP [D][Mp][Mp]: the adjoint of Psi2 (lower
// triangle read); consts: psi2_consts (launch_psi2_consts); ws: psi2_pgrad_ws_bytes.
// A range-guard hit (see psi2_pairs.hip) poisons the outputs with NaN (the caller's trouble flag sees it).
// which: 1 = part 1 only (images of the observations, pass 1; with psi2_part / scale also Psi2 into slab 0 of the forward's
// partial slabs), 2 = part 2 only (after part 1 on the same ws), 3 = both.
// y != nullptr (with Gv [D][Mp] = d f_hat / d (Psi1^T y)): the Psi1 term as well, through the same pass kernel; dmu and ds are then
// OVERWRITTEN (nothing else has written them), dz and dgamma added to.
// fast != 0 (DPGP_PREC_MIXED_FAST): the second products take the exponentials as their f16 roundings alone (11 bits; the pass that
// also yields Psi2 keeps both halves).
int launch_psi2_pgrad(int D, int N, int M, int Q, const unsigned char *consts, const double *z, const double *mu,
                      const double *s, const double *gamma, const double *alpha, const double *GP, unsigned char *ws,
                      double *stage, double *dmu, double *ds, double *dz, double *dgamma, hipStream_t st, int which,
                      float *psi2_part, const float *scale, const double *y, int ldy, const double *Gv, int fast, double *psi1v) {
    // psi1v != nullptr: with which = 1 (and y) part 1 also runs the adjoint-free half of the Psi1 term and writes Psi1^T y there; with
    // which = 2 it says that part 1 has done so (the pointer itself is not used)
    if (!psi2_pgrad_supported(M, Q)) return -4;
    (void)stage;
    int rc = DPGP_OK;
    switch (psi2_pairs_ksteps(Q)) {
#define CASE(k)                                                                                                                  \
    case k:                                                                                                                      \
        if (which & 1) rc = launch_pgrad_part1<k>(D, N, M, Q, consts, mu, s, gamma, ws, psi2_part, scale, st, !fast);           \
        if (rc == DPGP_OK && which == 1 && y && psi1v)                                                                           \
            rc = launch_psi1_front<k>(D, N, M, Q, consts, mu, s, gamma, alpha, ws, y, ldy, psi1v, st, !fast);                    \
        if (rc == DPGP_OK && (which & 2))                                                                                        \
            rc = launch_pgrad_part2<k>(D, N, M, Q, consts, z, mu, s, gamma, alpha, GP, scale, (which & 1) ? nullptr : psi2_part, ws, dmu, ds, \
                                       dz, dgamma, st, y, ldy, Gv, !fast, which == 2 && psi1v != nullptr);                       \
        return rc;
        CASE(2) CASE(4) CASE(6) CASE(8)
#undef CASE
    }
    return -4;
}
