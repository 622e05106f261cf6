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
};
__host__ __device__ inline Psi2Consts psi2_consts_layout(int M, int Q) {
    Psi2Consts c;
    const int KQ = 4 * ((Q + 3) / 4);
    c.ZLD = ((KQ / 4) & 1) ? KQ : KQ + 4;                      // = Psi2F16Lds<KB>::ZLD
    c.SL = 32 * ((6 * Q + 2 + 31) / 32);                       // = psi2p_layout<KB>(Q).SL
    c.Mp64 = (M + 63) & ~63;
    c.off_zs = 128;
    c.off_bimg = c.off_zs + sizeof(float) * (size_t)c.Mp64 * c.ZLD;
    c.bytes = (c.off_bimg + sizeof(_Float16) * (size_t)c.Mp64 * c.SL + 255) & ~(size_t)255;
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
                const float v = (tt & 1) ? zz : zz * zz;
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
