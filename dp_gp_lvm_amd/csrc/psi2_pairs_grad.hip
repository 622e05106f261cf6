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
#include "internal.h"
#include "psi2_consts.h"

typedef _Float16 pg_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 pg_h2 __attribute__((ext_vector_type(2)));
typedef float pg_f16v __attribute__((ext_vector_type(16)));
typedef float pg_f4 __attribute__((ext_vector_type(4)));
typedef unsigned pg_u4 __attribute__((ext_vector_type(4)));

#define PG_APAD 8
#define PG_HDR 512
#ifndef PG_G
#define PG_G 4                       // resident column tiles per wave (2: the row images are re-read twice as often, 7.3 vs 6.6 ms at config 3)
#endif
#define PG_WSHIFT 12.0f              // the exponent tiles carry + PG_WSHIFT: W = 2^12 exp2(E) <= 4096 uses the f16 range downwards
                                     // (f16 pairs resolve W / max W down to ~2^-36 instead of 2^-24); undone in the finishing kernels
#ifndef PG_OCC
#define PG_OCC 2                     // workgroups per compute unit the pass kernel is built for (registers, LDS buffers)
#endif
#define PG_NF 32                     // feature rows of the second product (2Q + 1 used)

// position of (feature f, row rr of a 32-row tile) in the transposed feature image of one (kind, row tile): the k-slot order
// of the second product's operands = the register order the exponent tile arrives in
__device__ __forceinline__ int pg_xt_index(int f, int rr) {
    const int c = rr >> 3, h = (rr >> 2) & 1, j = rr & 3, s_ = c >> 1, t_ = 4 * (c & 1) + j;
    return (((s_ * 2 + h) * 32) + f) * 8 + t_;
}
// feature f of row rr as an f16 (hi, lo) pair into the two transposed images (xh, xl: this row tile's)
__device__ __forceinline__ void pg_put(_Float16 *xh, _Float16 *xl, int f, int rr, float v) {
    v = dpgp_pin(v);
    const _Float16 vh = (_Float16)v;
    const int ix = pg_xt_index(f, rr);
    xh[ix] = vh;
    xl[ix] = (_Float16)(v - (float)vh);
}

// ---- the row of the exponent GEMM's A operand for observation n of output dim d (as phase A of psi2_pairs_kernel) --------
// dst: 8 KS words (16 KS f16 slots); xh != nullptr: also the features a'_q, b_q (the values the slots were split from) of
// this row (rr within its tile) into the transposed images
template <int KS>
__device__ __forceinline__ bool pg_obs_row(bool valid, int n, int Q, const double *__restrict__ mu, const double *__restrict__ s,
                                           const float *gq, const float *zc, unsigned *dst, _Float16 *xh, _Float16 *xl, int rr) {
    constexpr int SLP = 16 * KS;
    bool oor = false;
    float cc = -60000.0f;
    if (valid) {
        cc = 0.0f;
        for (int q = 0; q < Q; ++q) {
            const float g = gq[q], sv = (float)s[(size_t)n * Q + q], mc = (float)mu[(size_t)n * Q + q] - zc[q];
            const float den = 2.0f * g * sv + 1.0f, w = g / den;
            const float a = dpgp_pin((float)(-0.25 * DPGP_LOG2E / PSI2_PAIR_S2_SCALE) * w);
            const float bb = dpgp_pin((float)DPGP_LOG2E * w * mc);
            cc -= bb * mc + 0.5f * __builtin_amdgcn_logf(den);
            const _Float16 ah = (_Float16)a, alo = (_Float16)(a - (float)ah);
            const _Float16 bh = (_Float16)bb, blo = (_Float16)(bb - (float)bh);
            const pg_h2 w0 = {ah, ah}, w1 = {alo, bh}, w2 = {bh, blo};
            dst[3 * q] = __builtin_bit_cast(unsigned, w0);
            dst[3 * q + 1] = __builtin_bit_cast(unsigned, w1);
            dst[3 * q + 2] = __builtin_bit_cast(unsigned, w2);
            if (xh) {
                pg_put(xh, xl, 2 * q, rr, (float)ah + (float)alo);
                pg_put(xh, xl, 2 * q + 1, rr, (float)bh + (float)blo);
            }
        }
        oor = !(cc >= -8192.0f);                                  // range guard of the f16-split exponent (psi2_pairs.hip)
        cc = fmaxf(cc, -60000.0f) + PG_WSHIFT;
    } else {
        for (int q = 0; q < 3 * Q; ++q) dst[q] = 0u;
        if (xh)
            for (int f = 0; f < 2 * Q; ++f) pg_put(xh, xl, f, rr, 0.0f);
    }
    cc = dpgp_pin(cc);
    const _Float16 ch = (_Float16)cc;
    const float r1 = dpgp_pin(cc - (float)ch);
    const _Float16 cm = (_Float16)r1;
    const pg_h2 cw = {ch, cm}, cw2 = {(_Float16)(r1 - (float)cm), (_Float16)0.0f};
    dst[3 * Q] = __builtin_bit_cast(unsigned, cw);
    for (int k = 3 * Q + 1; k < SLP / 2; ++k) dst[k] = 0u;
    if (6 * Q + 2 < SLP) dst[3 * Q + 1] = __builtin_bit_cast(unsigned, cw2);
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

// ---- precomputed images -----------------------------------------------------------------------------------------------
// Every workgroup of a pass re-reads the row images of its output dim chunk by chunk; they are built ONCE per evaluation
// (observation side: per output dim; pair side: the exponent rows are shared by all output dims, the features carry u_dp):
//   operand order   cimg[set][tile][ks][lane 0..63][8 halves]   lane = 32 half + row % 32 holds slots 16 ks + 8 half .. + 7
//   row major       rimg[set][row][16 KS halves]                (the LDS copy adds the bank padding)
//   features        ximg[set][tile][kind hi / lo][K-step 0 / 1][lane][8 halves]   (pg_xt_index: the second product's A operand)
template <int KS>
__global__ __launch_bounds__(256) void pg_obs_images_kernel(int N, int Q, const unsigned char *__restrict__ consts,
                                                            const double *__restrict__ mu, const double *__restrict__ s,
                                                            const double *__restrict__ gamma, _Float16 *__restrict__ cimg,
                                                            _Float16 *__restrict__ rimg, _Float16 *__restrict__ ximg, int NT,
                                                            int *__restrict__ flag) {
    constexpr int SLP = 16 * KS, RW = SLP / 2 + 4;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float *gq = reinterpret_cast<float *>(smem_raw), *zc = gq + 32;
    unsigned *rows = reinterpret_cast<unsigned *>(smem_raw + 256);                 // [256][RW]
    _Float16 *xt = reinterpret_cast<_Float16 *>(rows + 256 * RW);                  // [8 tiles][2][2][64][8]
    const int d = blockIdx.y, t = threadIdx.x, n0 = 256 * blockIdx.x;
    if (t < 32) {
        gq[t] = (t < Q) ? (float)gamma[(size_t)d * Q + t] : 0.0f;
        zc[t] = reinterpret_cast<const float *>(consts)[t];
    }
    __syncthreads();
    {
        _Float16 *xh = xt + (size_t)(t >> 5) * 2048, *xl = xh + 1024;
        const bool valid = n0 + t < N;
        const bool oor = pg_obs_row<KS>(valid, n0 + t, Q, mu, s, gq, zc, rows + t * RW, xh, xl, t & 31);
        pg_put(xh, xl, 2 * Q, t & 31, valid ? 1.0f : 0.0f);
        for (int f = 2 * Q + 1; f < PG_NF; ++f) pg_put(xh, xl, f, t & 31, 0.0f);
        if (oor) atomicOr(flag, 1);
    }
    __syncthreads();
    const int tile0 = n0 / 32, ntl = min(8, NT - tile0);
    pg_u4 *cd = reinterpret_cast<pg_u4 *>(cimg) + ((size_t)d * NT + tile0) * KS * 64;
    for (int e = t; e < ntl * KS * 64; e += 256) {
        const int lane = e & 63, ks = (e >> 6) % KS, tl = e / (64 * KS);
        const unsigned *r = rows + (32 * tl + (lane & 31)) * RW + 8 * ks + 4 * (lane >> 5);
        cd[e] = (pg_u4){r[0], r[1], r[2], r[3]};
    }
    pg_u4 *rd = reinterpret_cast<pg_u4 *>(rimg) + ((size_t)d * NT + tile0) * 32 * (SLP / 8);
    for (int e = t; e < ntl * 32 * (SLP / 8); e += 256) {
        const int row = e / (SLP / 8), w = e - row * (SLP / 8);
        const unsigned *r = rows + row * RW + 4 * w;
        rd[e] = (pg_u4){r[0], r[1], r[2], r[3]};
    }
    pg_u4 *xd = reinterpret_cast<pg_u4 *>(ximg) + ((size_t)d * NT + tile0) * 256;
    for (int e = t; e < ntl * 256; e += 256) xd[e] = reinterpret_cast<const pg_u4 *>(xt)[e];
}

// pair side: thread = pair; ximg per output dim (features x kap_d u_dp), rimg once (blockIdx.y == 0)
template <int KS>
__global__ __launch_bounds__(256) void pg_pair_images_kernel(int M, int Q, const unsigned char *__restrict__ consts,
                                                             const float *__restrict__ u, const float *__restrict__ kap,
                                                             _Float16 *__restrict__ rimg, _Float16 *__restrict__ ximg) {
    constexpr int SLP = 16 * KS;
    __shared__ __align__(16) _Float16 xt[8 * 2048];
    const Psi2Consts C = psi2_consts_layout(M, Q);
    const _Float16 *pimg = reinterpret_cast<const _Float16 *>(consts + C.off_pairs);
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
        if (d == 0) {
            pg_u4 *rd = reinterpret_cast<pg_u4 *>(rimg) + (size_t)p * (SLP / 8);
#pragma unroll
            for (int w = 0; w < SLP / 8; ++w) rd[w] = (pg_u4){row[4 * w], row[4 * w + 1], row[4 * w + 2], row[4 * w + 3]};
        }
        const float up = kap[d] * u[(size_t)d * C.Ppad + p];
        _Float16 *xh = xt + (size_t)(t >> 5) * 2048, *xl = xh + 1024;
        const int rr = t & 31;
#pragma unroll
        for (int q = 0; q < DPGP_MAX_Q; ++q) {                    // slots {h, l, h | h, l, h} of (s^2 / 64, s): words 3q .. 3q + 2
            if (q < Q && 3 * q + 2 < SLP / 2) {
                const pg_h2 w0 = __builtin_bit_cast(pg_h2, row[3 * q]), w1 = __builtin_bit_cast(pg_h2, row[3 * q + 1]),
                            w2 = __builtin_bit_cast(pg_h2, row[3 * q + 2]);
                pg_put(xh, xl, 2 * q, rr, up * ((float)w0[0] + (float)w0[1]));
                pg_put(xh, xl, 2 * q + 1, rr, up * ((float)w1[1] + (float)w2[0]));
            }
        }
        pg_put(xh, xl, 2 * Q, rr, up);
        for (int f = 2 * Q + 1; f < PG_NF; ++f) pg_put(xh, xl, f, rr, 0.0f);
    }
    __syncthreads();
    pg_u4 *xd = reinterpret_cast<pg_u4 *>(ximg) + ((size_t)d * PT + tile0) * 256;
    for (int e = t; e < ntl * 256; e += 256) xd[e] = reinterpret_cast<const pg_u4 *>(xt)[e];
}

// ---- one pass: rows (LDS, chunked copies of rimg / ximg) x resident column tiles (cimg) ------------------------------------
// out[set][column][PG_NF] = sum_rows exp2(E[row, column]) X[row, :].  Pass 1: rows = observations of output dim d, columns =
// pairs; pass 2: rows = pairs, columns = observations of output dim d.  row_set / x_set / col_set: 1 = the images are per
// output dim, 0 = shared.
template <int KS>
__global__ __launch_bounds__(256, PG_OCC) void pg_pass_kernel(const _Float16 *__restrict__ rimg, int row_per_d,
                                                         const _Float16 *__restrict__ ximg, const _Float16 *__restrict__ cimg,
                                                         int col_per_d, float *__restrict__ out, int n_row_tiles, int n_col_tiles,
                                                         int groups_per_d, int R) {
    constexpr int SLP = 16 * KS, LDA = SLP + PG_APAD, G = PG_G;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // Two LDS buffers of NTb = R / 32 row tiles each, filled by LDS-DMA (global_load_lds_dwordx4: no staging registers, the
    // data lands while the waves compute on the other buffer).  The destination of one wave-instruction is lane-linear (base +
    // 16 lane), so a buffer is one array of 16-byte slots — [row][LDA halves] rows, then the feature image [kind][tile][128
    // slots] — and every lane computes the SOURCE address of its slot (the padding slots of a row fetch any valid word).
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int SPR = LDA / 8, DPR = SLP / 8;                 // 16-byte slots per LDS row / per image row
    const int NTb = R / 32, ri_slots = ((NTb * 32 * SPR + 63) / 64) * 64;
    const size_t buf_bytes = (size_t)16 * (ri_slots + NTb * 256);
    const int d = blockIdx.x / groups_per_d, cg = blockIdx.x - d * groups_per_d;
    const int t = threadIdx.x, lane = t & 63, l5 = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int NTc = NTb;
    const pg_u4 *rsrc = reinterpret_cast<const pg_u4 *>(rimg) + (row_per_d ? (size_t)d * n_row_tiles * 32 * DPR : 0);
    const pg_u4 *xsrc = reinterpret_cast<const pg_u4 *>(ximg) + (size_t)d * n_row_tiles * 256;
    const _Float16 *csrc = cimg + (col_per_d ? (size_t)d * n_col_tiles * KS * 64 * 8 : 0);
    auto fill = [&](int buf, int rt0, int ntile) __attribute__((always_inline)) {
        unsigned char *base = smem_raw + (size_t)buf * buf_bytes;
        const int n_ri = ntile * 32 * SPR, n_ri_instr = (n_ri + 63) >> 6;
        for (int i = wv; i < n_ri_instr; i += 4) {               // (wave-uniform trip count)
            const int j = 64 * i + lane, row = j / SPR, col = j - row * SPR;
            const pg_u4 *src = rsrc;
            if (j < n_ri && col < DPR) src = rsrc + ((size_t)rt0 * 32 + row) * DPR + col;
            __builtin_amdgcn_global_load_lds(src, (lds_void *)(base + (size_t)1024 * i), 16, 0, 0);
        }
        for (int i = wv; i < ntile * 4; i += 4) {                 // feature image: (tile, kind, half of 128 slots) per instruction
            const int rt = i >> 2, kind = (i >> 1) & 1, hf = i & 1;
            const pg_u4 *src = xsrc + ((size_t)(rt0 + rt) * 256 + kind * 128 + hf * 64 + lane);
            __builtin_amdgcn_global_load_lds(src, (lds_void *)(base + (size_t)16 * (ri_slots + (kind * NTb + rt) * 128 + hf * 64)), 16, 0, 0);
        }
    };
    pg_h8 bop[G][KS];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int ct = min(cg * 4 * G + wv + 4 * g, n_col_tiles - 1);   // (a surplus tile of the last group repeats the last one; not stored)
        const pg_h8 *row = reinterpret_cast<const pg_h8 *>(csrc) + (size_t)ct * KS * 64 + 32 * half + l5;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bop[g][ks] = row[ks * 64];
    }
    pg_f16v acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[g][v] = 0.0f;

    fill(0, 0, min(NTb, n_row_tiles));
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int rt0 = 0, cb = 0; rt0 < n_row_tiles; rt0 += NTb, cb ^= 1) {
        const int ntile = min(NTb, n_row_tiles - rt0);
        if (rt0 + NTb < n_row_tiles) fill(cb ^ 1, rt0 + NTb, min(NTb, n_row_tiles - rt0 - NTb));   // lands during the work below
        const _Float16 *ri = reinterpret_cast<const _Float16 *>(smem_raw + (size_t)cb * buf_bytes);
        const _Float16 *xt = ri + (size_t)8 * ri_slots;
        // ---- software pipeline over the tile steps (row tile nt, resident tile g) --------------------------------------
        // Per step the matrix pipe has KS + 6 MFMAs and the vector unit 16 exp + the (hi, lo) split of W; one wave's
        // instruction stream alternates them.  Step k runs
        //   first half  (values 0-7 of its exponent tile): the exponent chain of step k + 1  and  the K-step-1 products of step k - 1
        //   second half (values 8-15):                      the K-step-0 products of step k (their operands are complete by then)
        // two independent accumulation chains alternate on the pipe.  The last step of a row tile finishes its own products
        // (the feature operands change with the row tile).
        auto load_a = [&](pg_h8 (&a)[KS], int nt) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                a[ks] = *reinterpret_cast<const pg_h8 *>(ri + (size_t)(32 * nt + l5) * LDA + 16 * ks + 8 * half);
        };
        // values 2i, 2i + 1 of the exponent tile -> word i of the (hi, lo) operands: v_exp_f32 x 2, v_cvt_pk_f16_f32 (hi pair),
        // v_fma_mix_f32 x 2 (e - hi from the f16 halves: one instruction instead of v_cvt_f32_f16 + v_sub_f32), v_cvt_pk_f16_f32
        // (lo pair).  Only the fma_mix is inline asm, and it works IN PLACE on the register of e: inline-asm VALU writes are
        // invisible to the compiler's hazard recognizer — with the conversions that write the MFMA OPERANDS in asm, operand
        // registers of an MFMA still in flight were overwritten (measured: non-deterministic garbage as soon as registers were
        // reused across tiles).  e's register was just written by v_exp_f32 (checked by the compiler, never an MFMA operand),
        // and the operand words are written by compiler-visible conversions.
        typedef _Float16 pg_h2v __attribute__((ext_vector_type(2)));
        auto pair = [&](const pg_f16v &c, int i, pg_h8 (&wh)[2], pg_h8 (&wl)[2]) __attribute__((always_inline)) {
            float e0 = __builtin_amdgcn_exp2f(c[2 * i]), e1 = __builtin_amdgcn_exp2f(c[2 * i + 1]);
            const pg_h2v hi = {(_Float16)e0, (_Float16)e1};
            const unsigned ph = __builtin_bit_cast(unsigned, hi);
            asm("s_nop 0\n\tv_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(e0) : "v"(ph));
            asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(e1) : "v"(ph));
            const pg_h2v lo = {(_Float16)e0, (_Float16)e1};
            const int v = 2 * i;
            wh[v >> 3][v & 7] = hi[0]; wh[v >> 3][(v & 7) + 1] = hi[1];
            wl[v >> 3][v & 7] = lo[0]; wl[v >> 3][(v & 7) + 1] = lo[1];
        };
#ifdef PG_DIAG_NO_MMA               // (timing experiments only: wrong results)
#define PG_MMA(A, B, C) __builtin_amdgcn_sched_barrier(0)
#else
#define PG_MMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0); __builtin_amdgcn_sched_barrier(0)
#endif
#ifdef PG_DIAG_NO_VALU              // (timing experiments only: wrong results)
#define PG_PAIR(C, I) __builtin_amdgcn_sched_barrier(0)
#else
#define PG_PAIR(C, I) pair(C, I, wh, wl); __builtin_amdgcn_sched_barrier(0)
#endif
        pg_h8 a_cur[KS], a_nxt[KS], xh[2], xl[2], wh[2], wl[2];
        pg_f16v c_cur, c_nxt;
        load_a(a_cur, 0);
#pragma unroll
        for (int v = 0; v < 16; ++v) c_cur[v] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) c_cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_cur[ks], bop[0][ks], c_cur, 0, 0, 0);
#pragma unroll 1
        for (int nt = 0; nt < ntile; ++nt) {
            load_a(a_nxt, min(nt + 1, ntile - 1));
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                xh[s_] = *reinterpret_cast<const pg_h8 *>(xt + (((size_t)(0 * NTc + nt) * 2 + s_) * 64 + lane) * 8);
                xl[s_] = *reinterpret_cast<const pg_h8 *>(xt + (((size_t)(1 * NTc + nt) * 2 + s_) * 64 + lane) * 8);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                // the exponent chain of the next step: tile g + 1 of this row tile, or tile 0 of the next one (behind the last
                // row tile of the chunk: one surplus chain)
                const pg_h8 (&an)[KS] = (g + 1 < G) ? a_cur : a_nxt;
                const pg_h8 (&bn)[KS] = bop[(g + 1) % G];
#pragma unroll
                for (int v = 0; v < 16; ++v) c_nxt[v] = 0.0f;
                const pg_h8 wh1 = wh[1], wl1 = wl[1];             // (of step k - 1)
                // ---- first half ----
                PG_MMA(an[0], bn[0], c_nxt);
                PG_PAIR(c_cur, 0);
                if (g > 0) { PG_MMA(xh[1], wh1, acc[g > 0 ? g - 1 : 0]); }
                if (KS > 1) { PG_MMA(an[1 % KS], bn[1 % KS], c_nxt); }
                PG_PAIR(c_cur, 1);
                if (g > 0) { PG_MMA(xh[1], wl1, acc[g > 0 ? g - 1 : 0]); }
                if (KS > 2) { PG_MMA(an[2 % KS], bn[2 % KS], c_nxt); }
                PG_PAIR(c_cur, 2);
                if (g > 0) { PG_MMA(xl[1], wh1, acc[g > 0 ? g - 1 : 0]); }
                if (KS > 3) { PG_MMA(an[3 % KS], bn[3 % KS], c_nxt); }
                PG_PAIR(c_cur, 3);
                // ---- second half ----
                const pg_h8 wh0 = wh[0], wl0 = wl[0];
                PG_MMA(xh[0], wh0, acc[g]);
                PG_PAIR(c_cur, 4);
                PG_MMA(xh[0], wl0, acc[g]);
                PG_PAIR(c_cur, 5);
                PG_MMA(xl[0], wh0, acc[g]);
                PG_PAIR(c_cur, 6);
                PG_PAIR(c_cur, 7);
                if (g == G - 1) {                                  // the row tile's last step finishes its own K-step-1 products
                    const pg_h8 wh1e = wh[1], wl1e = wl[1];
                    PG_MMA(xh[1], wh1e, acc[g]);
                    PG_MMA(xh[1], wl1e, acc[g]);
                    PG_MMA(xl[1], wh1e, acc[g]);
                }
                c_cur = c_nxt;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a_cur[ks] = a_nxt[ks];
        }
#undef PG_MMA
#undef PG_PAIR
        __builtin_amdgcn_s_waitcnt(0);                            // this wave's LDS-DMAs of the next chunk have landed ...
        __syncthreads();                                           // ... everybody's have, and everybody is done with this buffer
    }
    // ---- out[d][column][f]: register v of lane (column l5, half) is feature 8 (v / 4) + 4 half + v % 4 ----
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int tile = cg * 4 * G + wv + 4 * g;
        if (tile >= n_col_tiles) continue;
        float *o = out + (((size_t)d * n_col_tiles + tile) * 32 + l5) * PG_NF + 4 * half;
#pragma unroll
        for (int vq = 0; vq < 4; ++vq)
            *reinterpret_cast<pg_f4 *>(o + 8 * vq) = (pg_f4){acc[g][4 * vq], acc[g][4 * vq + 1], acc[g][4 * vq + 2], acc[g][4 * vq + 3]};
    }
}

// ---- finishing, pair side -----------------------------------------------------------------------------------------------
// thread = pair p, block row = chunk of output dims: partial sums over the chunk's d of
//   tp[c][0][p][q] = sum_d u_dp R2[2q],  tp[c][1][p][q] = sum_d u_dp R2[2q+1],  tp[c][2][p][q] = sum_d u_dp C_dp gamma_dq  (C = R2[2Q])
__global__ __launch_bounds__(256) void pg_finish_pairs_kernel(int M, int Q, int D, int Ppad, int dchunk,
                                                              const double *__restrict__ gamma, const float *__restrict__ u,
                                                              const float *__restrict__ r2, double *__restrict__ tp) {
    const int p = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
    const int P = (int)((long long)M * (M + 1) / 2);
    if (p >= P) return;
    double a1[DPGP_MAX_Q], a2[DPGP_MAX_Q], a3[DPGP_MAX_Q];
#pragma unroll
    for (int q = 0; q < DPGP_MAX_Q; ++q) { a1[q] = 0.0; a2[q] = 0.0; a3[q] = 0.0; }
    const int d1 = min(D, (c + 1) * dchunk);
    for (int d = c * dchunk; d < d1; ++d) {
        const float ud = u[(size_t)d * Ppad + p] * (1.0f / 4096.0f);          // (x 2^-PG_WSHIFT)
        const pg_f4 *row = reinterpret_cast<const pg_f4 *>(r2 + ((size_t)d * Ppad + p) * PG_NF);
        float rv[PG_NF];
#pragma unroll
        for (int k = 0; k < PG_NF / 4; ++k) {
            const pg_f4 v = row[k];
            rv[4 * k] = v[0]; rv[4 * k + 1] = v[1]; rv[4 * k + 2] = v[2]; rv[4 * k + 3] = v[3];
        }
        float cc = 0.0f;
#pragma unroll
        for (int k = 0; k < PG_NF; ++k) cc = (k == 2 * Q) ? rv[k] : cc;
        const double uc = (double)ud * (double)cc;
#pragma unroll
        for (int q = 0; q < DPGP_MAX_Q; ++q)
            if (q < Q && 2 * q + 1 < PG_NF) {
                a1[q] += (double)ud * (double)rv[2 * q];
                a2[q] += (double)ud * (double)rv[2 * q + 1];
                a3[q] += uc * gamma[(size_t)d * Q + q];
            }
    }
    double *o = tp + (size_t)c * 3 * P * Q;
#pragma unroll
    for (int q = 0; q < DPGP_MAX_Q; ++q)
        if (q < Q) {
            o[(size_t)p * Q + q] = a1[q];
            o[((size_t)P + p) * Q + q] = a2[q];
            o[((size_t)2 * P + p) * Q + q] = a3[q];
        }
}
// thread = (m, q): dz[m][q] += sum over the pairs that hold m of  ln2 (2 S2 s_pq t[0] + t[1]) -+ 1/2 delta_pq t[2]
__global__ __launch_bounds__(256) void pg_gather_dz_kernel(int M, int Q, const double *__restrict__ z,
                                                           const unsigned char *__restrict__ consts,
                                                           const double *__restrict__ tt, double *__restrict__ dz) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= M * Q) return;
    const int m = e / Q, q = e - m * Q;
    const size_t P = (size_t)M * (M + 1) / 2;
    const double c = (double)reinterpret_cast<const float *>(consts)[q];     // (the centring constant the images were built with)
    const double zm = z[(size_t)m * Q + q];
    double acc = 0.0;
    for (int o = 0; o < M; ++o) {
        const int hi = o > m ? o : m, lo = o > m ? m : o;
        const size_t p = (size_t)hi * (hi + 1) / 2 + lo;
        const double zo = z[(size_t)o * Q + q];
        const double sp = (zm - c) + (zo - c);
        const double d1 = 0.6931471805599453 * (2.0 * (double)PSI2_PAIR_S2_SCALE * sp * tt[p * Q + q] + tt[(P + p) * Q + q]);
        const double d2 = -0.5 * (zm - zo) * tt[(2 * P + p) * Q + q];      // (delta is antisymmetric in (m, o): one formula for both ends)
        acc += d1 + d2;
        if (o == m) acc += d1;
    }
    dz[e] += acc;
}
// block = output dim d: dgamma[d][q] += sum_p -1/4 delta_pq^2 u_dp C_dp
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
        const double uc = (double)u[(size_t)d * Ppad + p] * (double)r2[((size_t)d * Ppad + p) * PG_NF + 2 * Q] * (1.0 / 4096.0);
#pragma unroll
        for (int q = 0; q < DPGP_MAX_Q; ++q)
            if (q < Q) {
                const double dd = z[(size_t)m * Q + q] - z[(size_t)mp * Q + q];
                a[q] += -0.25 * dd * dd * uc;
            }
    }
#pragma unroll
    for (int q = 0; q < DPGP_MAX_Q; ++q) {
        if (q >= Q) break;
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

// ---- finishing, observation side: thread = observation n, block row = chunk of output dims ---------------------------------
// dmu_part / ds_part [chunk][N][Q]: partial sums over the chunk's d;  dg_part[n-block][d][q]: this block's share of dgamma
__global__ __launch_bounds__(256) void pg_finish_obs_kernel(int N, int Q, int D, int NT, int dchunk,
                                                            const unsigned char *__restrict__ consts,
                                                            const double *__restrict__ mu, const double *__restrict__ s,
                                                            const double *__restrict__ gamma, const float *__restrict__ kap,
                                                            const float *__restrict__ r1, double *__restrict__ dmu_part,
                                                            double *__restrict__ ds_part, double *__restrict__ dg_part) {
    __shared__ double red[4][DPGP_MAX_Q + 2];
    const int t = threadIdx.x, n = blockIdx.x * 256 + t, lane = t & 63, wv = t >> 6, c = blockIdx.y;
    const bool ok = n < N;
    double mc[DPGP_MAX_Q], sv[DPGP_MAX_Q], am[DPGP_MAX_Q], as_[DPGP_MAX_Q];
#pragma unroll
    for (int q = 0; q < DPGP_MAX_Q; ++q) {
        const bool on = ok && q < Q;
        mc[q] = on ? mu[(size_t)n * Q + q] - (double)reinterpret_cast<const float *>(consts)[q] : 0.0;
        sv[q] = on ? s[(size_t)n * Q + q] : 1.0;
        am[q] = 0.0;
        as_[q] = 0.0;
    }
    const int d1 = min(D, (c + 1) * dchunk);
    for (int d = c * dchunk; d < d1; ++d) {
        const double ik = 1.0 / ((double)kap[d] * 4096.0);            // (kap_d and 2^PG_WSHIFT)
        const pg_f4 *row = reinterpret_cast<const pg_f4 *>(r1 + ((size_t)d * NT * 32 + (ok ? n : 0)) * PG_NF);
        float rv[PG_NF];
#pragma unroll
        for (int k = 0; k < PG_NF / 4; ++k) {
            const pg_f4 v = row[k];
            rv[4 * k] = v[0]; rv[4 * k + 1] = v[1]; rv[4 * k + 2] = v[2]; rv[4 * k + 3] = v[3];
        }
        float cc = 0.0f;
#pragma unroll
        for (int k = 0; k < PG_NF; ++k) cc = (k == 2 * Q) ? rv[k] : cc;
        const double rc = ok ? (double)cc * ik : 0.0;
#pragma unroll
        for (int q = 0; q < DPGP_MAX_Q; ++q) {
            if (q >= Q || 2 * q + 1 >= PG_NF) break;
            const double g = gamma[(size_t)d * Q + q];
            const double ra = ok ? (double)rv[2 * q] * ik : 0.0, rb = ok ? (double)rv[2 * q + 1] * ik : 0.0;
            const double den = 2.0 * g * sv[q] + 1.0, w = g / den;
            const double dw = (-0.25 / (double)PSI2_PAIR_S2_SCALE) * ra + mc[q] * rb - mc[q] * mc[q] * rc;
            const double dmc = w * (rb - 2.0 * mc[q] * rc);
            const double dden = -0.5 * rc / den - (g / (den * den)) * dw;
            am[q] += dmc;
            as_[q] += 2.0 * g * dden;
            double dgq = dw / den + 2.0 * sv[q] * dden;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dgq += __shfl_xor(dgq, o, 64);
            if (lane == 0) red[wv][q] = dgq;
        }
        __syncthreads();
        if (t < Q) dg_part[((size_t)blockIdx.x * D + d) * Q + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
        __syncthreads();
    }
    if (ok) {
#pragma unroll
        for (int q = 0; q < DPGP_MAX_Q; ++q)
            if (q < Q) {
                dmu_part[((size_t)c * N + n) * Q + q] = am[q];
                ds_part[((size_t)c * N + n) * Q + q] = as_[q];
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
bool psi2_pgrad_supported(int M, int Q) { return psi2_pairs_ksteps(Q) <= 4 && 2 * Q + 1 <= PG_NF && M >= 1 && M <= 4096; }

#define PG_DC_PAIRS 16               // chunks of output dims of the finishing kernels
#define PG_DC_OBS 64
struct PgLayout {
    int KS, P, Ppad, NT, PT, nblk_obs;
    size_t off_u, off_kap, off_flag, off_cobs, off_robs, off_xobs, off_rpair, off_xpair, off_r2, off_r1, off_tp, off_tt, off_dgp,
        off_dmup, off_dsp, total;
};
static PgLayout pg_layout(int D, int N, int M, int Q) {
    PgLayout L;
    const Psi2Consts C = psi2_consts_layout(M, Q);
    L.KS = C.KS; L.P = C.P; L.Ppad = C.Ppad; L.NT = dpgp_ceil_div(N, 32); L.PT = C.Ppad / 32;
    L.nblk_obs = dpgp_ceil_div(N, 256);
    const size_t h = sizeof(_Float16);
    size_t o = 0;
    L.off_u = o;     o += dpgp_align256(sizeof(float) * (size_t)D * L.Ppad);
    L.off_kap = o;   o += dpgp_align256(sizeof(float) * (size_t)D);
    L.off_flag = o;  o += 256;
    L.off_cobs = o;  o += dpgp_align256(h * (size_t)D * L.NT * L.KS * 64 * 8);
    L.off_robs = o;  o += dpgp_align256(h * (size_t)D * L.NT * 32 * 16 * L.KS);
    L.off_xobs = o;  o += dpgp_align256(h * (size_t)D * L.NT * 2048);
    L.off_rpair = o; o += dpgp_align256(h * (size_t)L.Ppad * 16 * L.KS);
    L.off_xpair = o; o += dpgp_align256(h * (size_t)D * L.PT * 2048);
    L.off_r2 = o;    o += dpgp_align256(sizeof(float) * (size_t)D * L.Ppad * PG_NF);
    L.off_r1 = o;    o += dpgp_align256(sizeof(float) * (size_t)D * L.NT * 32 * PG_NF);
    L.off_tp = o;    o += dpgp_align256(sizeof(double) * (size_t)PG_DC_PAIRS * 3 * L.P * Q);
    L.off_tt = o;    o += dpgp_align256(sizeof(double) * (size_t)3 * L.P * Q);
    L.off_dgp = o;   o += dpgp_align256(sizeof(double) * (size_t)L.nblk_obs * D * Q);
    L.off_dmup = o;  o += dpgp_align256(sizeof(double) * (size_t)PG_DC_OBS * N * Q);
    L.off_dsp = o;   o += dpgp_align256(sizeof(double) * (size_t)PG_DC_OBS * N * Q);
    L.total = o;
    return L;
}
size_t psi2_pgrad_ws_bytes(int D, int N, int M, int Q) { return psi2_pgrad_supported(M, Q) ? pg_layout(D, N, M, Q).total : 0; }

template <int KS>
static int launch_pgrad_ks(int D, int N, int M, int Q, const unsigned char *consts, const double *z, const double *mu,
                           const double *s, const double *gamma, const double *alpha, const double *GP, unsigned char *ws,
                           double *stage, double *dmu, double *ds, double *dz, double *dgamma, hipStream_t st) {
    const PgLayout L = pg_layout(D, N, M, Q);
    const int Mp = dpgp_round_up(M, 16);
    float *u = reinterpret_cast<float *>(ws + L.off_u), *kap = reinterpret_cast<float *>(ws + L.off_kap);
    int *flag = reinterpret_cast<int *>(ws + L.off_flag);
    _Float16 *cobs = reinterpret_cast<_Float16 *>(ws + L.off_cobs), *robs = reinterpret_cast<_Float16 *>(ws + L.off_robs),
             *xobs = reinterpret_cast<_Float16 *>(ws + L.off_xobs), *rpair = reinterpret_cast<_Float16 *>(ws + L.off_rpair),
             *xpair = reinterpret_cast<_Float16 *>(ws + L.off_xpair);
    float *r2 = reinterpret_cast<float *>(ws + L.off_r2), *r1 = reinterpret_cast<float *>(ws + L.off_r1);
    double *tp = reinterpret_cast<double *>(ws + L.off_tp), *tt = reinterpret_cast<double *>(ws + L.off_tt);
    double *dgp = reinterpret_cast<double *>(ws + L.off_dgp), *dmup = reinterpret_cast<double *>(ws + L.off_dmup),
           *dsp = reinterpret_cast<double *>(ws + L.off_dsp);
    const _Float16 *pimg = reinterpret_cast<const _Float16 *>(consts + psi2_consts_layout(M, Q).off_pairs);
    if (hipMemsetAsync(flag, 0, sizeof(int), st) != hipSuccess) return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(pg_u_kernel, dim3(D), dim3(256), 0, st, M, Q, Mp, z, gamma, alpha, GP, u, kap);
    DPGP_LAUNCH_CHECK();
    {
        const size_t lds = 256 + sizeof(unsigned) * 256 * (8 * KS + 4) + sizeof(_Float16) * 8 * 2048;
        auto kern = pg_obs_images_kernel<KS>;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return DPGP_ERR_LAUNCH;
        DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3(dpgp_ceil_div(N, 256), D), dim3(256), lds, st, N, Q, consts, mu, s, gamma, cobs, robs, xobs,
                           L.NT, flag);
        DPGP_LAUNCH_CHECK();
    }
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((pg_pair_images_kernel<KS>), dim3(dpgp_ceil_div(L.Ppad, 256), D), dim3(256), 0, st, M, Q, consts,
                       (const float *)u, (const float *)kap, rpair, xpair);
    DPGP_LAUNCH_CHECK();
    // two LDS buffers per workgroup within 80 KB (2 workgroups per CU): row tiles per buffer from the bytes of a row (row image
    // LDA halves + feature image 64 halves); the row-image region is rounded up to whole wave-instructions of the LDS-DMA fill
    const size_t row = sizeof(_Float16) * (size_t)(16 * KS + PG_APAD + 64);
    const size_t buf_budget = (size_t)(160 * 1024 / PG_OCC) / 2;
    int NTb = (int)(buf_budget / (32 * row));
    while (NTb > 1 && (size_t)16 * (((NTb * 32 * (16 * KS + PG_APAD) / 8 + 63) / 64) * 64 + NTb * 256) > buf_budget) --NTb;
    const int R = 32 * NTb;
    const size_t lds = (size_t)2 * 16 * (((NTb * 32 * (16 * KS + PG_APAD) / 8 + 63) / 64) * 64 + NTb * 256);
    auto kern = pg_pass_kernel<KS>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return DPGP_ERR_LAUNCH;
    for (int pass = 1; pass <= 2; ++pass) {
        const int n_row_tiles = pass == 1 ? L.NT : L.PT, n_col_tiles = pass == 1 ? L.PT : L.NT;
        const int groups = dpgp_ceil_div(n_col_tiles, 4 * PG_G);
        const long long nwg = (long long)D * groups;
        if (nwg > 0x7fffffffLL) return -1;
        DPGP_PRELAUNCH();
        if (pass == 1)
            hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, st, (const _Float16 *)robs, 1, (const _Float16 *)xobs, pimg, 0, r2,
                               n_row_tiles, n_col_tiles, groups, R);
        else
            hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), lds, st, (const _Float16 *)rpair, 0, (const _Float16 *)xpair,
                               (const _Float16 *)cobs, 1, r1, n_row_tiles, n_col_tiles, groups, R);
        DPGP_LAUNCH_CHECK();
    }
    const int dcp = dpgp_ceil_div(D, PG_DC_PAIRS), ncp = dpgp_ceil_div(D, dcp);
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(pg_finish_pairs_kernel, dim3(dpgp_ceil_div(L.P, 256), ncp), dim3(256), 0, st, M, Q, D, L.Ppad, dcp, gamma,
                       (const float *)u, (const float *)r2, tp);
    DPGP_LAUNCH_CHECK();
    const size_t n3 = (size_t)3 * L.P * Q;
    int rc = launch_reduce_rows<double>(n3, n3, ncp, tp, tt, 0, nullptr, st);
    if (rc != DPGP_OK) return rc;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(pg_gather_dz_kernel, dim3(dpgp_ceil_div(M * Q, 256)), dim3(256), 0, st, M, Q, z, consts, (const double *)tt, dz);
    DPGP_LAUNCH_CHECK();
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(pg_dgamma_pairs_kernel, dim3(D), dim3(256), 0, st, M, Q, L.Ppad, z, (const float *)u, (const float *)r2,
                       dgamma);
    DPGP_LAUNCH_CHECK();
    const int dco = dpgp_ceil_div(D, PG_DC_OBS), nco = dpgp_ceil_div(D, dco);
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(pg_finish_obs_kernel, dim3(L.nblk_obs, nco), dim3(256), 0, st, N, Q, D, L.NT, dco, consts, mu, s, gamma,
                       (const float *)kap, (const float *)r1, dmup, dsp, dgp);
    DPGP_LAUNCH_CHECK();
    const size_t dq = (size_t)D * Q, nq = (size_t)N * Q;
    rc = launch_reduce_rows<double>(dq, dq, L.nblk_obs, dgp, dgamma, 1, nullptr, st);
    if (rc == DPGP_OK) rc = launch_reduce_rows<double>(nq, nq, nco, dmup, dmu, 1, nullptr, st);
    if (rc == DPGP_OK) rc = launch_reduce_rows<double>(nq, nq, nco, dsp, ds, 1, nullptr, st);
    if (rc != DPGP_OK) return rc;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(pg_poison_kernel, dim3(1), dim3(1), 0, st, (const int *)flag, dmu, ds, dz, dgamma);
    DPGP_LAUNCH_CHECK();
    (void)stage;
    return DPGP_OK;
}

// Psi2 part of stage B, ADDED to dmu [N,Q], ds [N,Q], dz [M,Q], dgamma [D,Q]; GP [D][Mp][Mp]: the adjoint of Psi2 (lower
// triangle read); consts: psi2_consts (launch_psi2_consts); ws: psi2_pgrad_ws_bytes.
// A range-guard hit (see psi2_pairs.hip) poisons the outputs with NaN (the caller's trouble flag sees it).
int launch_psi2_pgrad(int D, int N, int M, int Q, const unsigned char *consts, const double *z, const double *mu,
                      const double *s, const double *gamma, const double *alpha, const double *GP, unsigned char *ws,
                      double *stage, double *dmu, double *ds, double *dz, double *dgamma, hipStream_t st) {
    if (!psi2_pgrad_supported(M, Q)) return -4;
    switch (psi2_pairs_ksteps(Q)) {
        case 2: return launch_pgrad_ks<2>(D, N, M, Q, consts, z, mu, s, gamma, alpha, GP, ws, stage, dmu, ds, dz, dgamma, st);
        case 4: return launch_pgrad_ks<4>(D, N, M, Q, consts, z, mu, s, gamma, alpha, GP, ws, stage, dmu, ds, dz, dgamma, st);
    }
    return -4;
}
