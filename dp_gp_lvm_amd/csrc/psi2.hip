// K3  psi2 statistic, streamed over n on the matrix cores.
// Reference: /root/reference/src/kernels/rbf_kernel.py:164-199 (materialises [B,N,M,M,Q]; here nothing larger than the
// [B,M,M] result ever exists).
//
// Algebra (per batch entry b; den = 2 g s_n + 1, w = g/den, everything in log2 units, z centred by its column mean c):
//   log2 psi2[n,m,m'] = 2 log2 alpha + beta_mm' + P[n,m] + P[n,m'] + sum_q X[n,q] z_mq z_m'q
//     X[n,q]   = -1/2 w_nq log2e
//     P[n,m]   = log2e * sum_q ( 1/2 w_nq (mu_nq-c_q)^2 - 1/4 log den_nq - 1/4 w_nq (z_mq - 2 (mu_nq-c_q))^2 )
//     beta_mm' = -1/4 log2e sum_q g_q (z_mq - z_m'q)^2                 (n-independent: applied once, at the end)
//   so for every n the [m,m'] block of exponents is ONE small GEMM  E_n = A_n B_n  with K = Q+2:
//     A_n[m, 0:Q] = X[n,:] * z[m,:],  A_n[m,Q] = P[n,m],  A_n[m,Q+1] = 1
//     B_n[0:Q,m'] = z[m',:]^T,        B_n[Q,m'] = 1,      B_n[Q+1,m'] = P[n,m']
//   evaluated with v_mfma_{f32,f64}_16x16x4; then psi2[m,m'] += exp2(E_n[m,m']) element-wise in registers.
//   (identical to the reference formula: tests/test_psi2.py checks against the literal oracle)
//
// Decomposition: workgroup = (patch of the lower triangle, b, n-split); patch = PT x PT tiles of 16x16 (64x64 fp32,
// 32x32 fp64); the 4 waves of a workgroup work on the SAME patch and split the n of every 32-row tile 4 ways
// (identical work per wave, z operands shared), their accumulators are summed through LDS at the end.
// Output: partial slabs part[split][b][Mp][Mp], lower block-triangle of 16x16 tiles only.
#include <type_traits>
#include "internal.h"
#include "linalg_dev.h"
#include "psi2_consts.h"
#include "psi2_chain_task.h"

#define PSI2_NT 32   // n per LDS tile
#ifndef PSI2_PF_NARROW_KB
#define PSI2_PF_NARROW_KB 5   // from this many groups of 4 latent dims on, the prefetched q(X) rows are held as fp32
#endif

// Workgroup coordinates.  grid = (B, n-splits, patches); the patch index is the SLOWEST dimension and enumerates the
// off-diagonal patches (16 tiles of work) before the diagonal ones (10 tiles), so the long workgroups are dispatched
// first and the short ones fill the tail.
__device__ __forceinline__ void psi2_patch_coords(int nps, int p, int &pi, int &pj) {
    const int noff = nps * (nps - 1) / 2;
    if (p < noff) {                    // strict lower triangle: p = pi (pi - 1) / 2 + pj, pj < pi
        int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
        while (i * (i - 1) / 2 > p) --i;
        while ((i + 1) * i / 2 <= p) ++i;
        pi = i;
        pj = p - i * (i - 1) / 2;
    } else {
        pi = pj = p - noff;
    }
}
__device__ __forceinline__ void psi2_block_coords(int nps, int zoff, int &b, int &sp, int &pi, int &pj) {
    b = blockIdx.x;
    sp = blockIdx.y;
    psi2_patch_coords(nps, blockIdx.z - zoff, pi, pj);
}
// LDS geometry shared by the kernel and the host-side size computation
template <typename T, int KS, int PT> struct Psi2Lds {
    static constexpr int PS = 16 * PT;                       // patch edge
    static constexpr int KP = 4 * KS;                        // K padded to the MFMA step
    static constexpr int ZLD = ((KP / 4) & 1) ? KP : KP + 4; // z row stride: multiple of 4, odd number of 16-B slots
    static constexpr int PLD = 2 * PS + 4;                   // P row stride
    static constexpr int NR = 8;                             // rows (n) per wave-private chunk
    static constexpr int WSZ = 3 * NR * KP + 2 * NR * KP / 2 + NR * PLD;   // xa, w4, tm, (cn + cr), pm  per wave
    static constexpr int OFF_W = 2 * PS * ZLD + 2 * (DPGP_MAX_Q + 2);      // start of the per-wave regions
    static constexpr int FILL = OFF_W + 4 * WSZ;
    static constexpr int RED = 2 * PS * ZLD + 2 * (DPGP_MAX_Q + 2) + 4 * PT * 4 * 64;
    static constexpr int ELEMS0 = FILL > RED ? FILL : RED;
    static constexpr int OFF_TAB = (ELEMS0 + 3) & ~3;        // exp2 table of the fp64 kernel (dpgp_exp2_tab), 16-B aligned
    static constexpr int ELEMS = OFF_TAB + (sizeof(T) == 8 ? DPGP_EXP2_TAB_ELEMS : 0);
};

// DIAG is a compile-time property of the patch (pi == pj: only tiles J <= I are computed) so that the MFMA / exp
// sequences are straight-line code.  (With a run-time `diag` predicate around each MFMA, hipcc 7.2 shuffled the
// accumulators through AGPRs and overwrote a SrcC register of an in-flight v_mfma_f32_16x16x4_f32 with
// v_accvgpr_write_b32 with no wait states in between: one register of one tile came out wrong on the GPU.)
//
// The four waves of the workgroup are AUTONOMOUS in the main loop: wave w owns the observations n = nbeg + w + 4 j and
// a private LDS region; for every chunk of NR of its rows it (A) computes the per-(n,q) factors, (B) the P[n, .] rows
// of both column blocks, (C) runs the MFMA + exp2 accumulation.  No workgroup barrier separates the phases, so the
// waves drift apart and one wave's VALU phases overlap another wave's matrix-core phase on the same SIMD.
template <typename TIN, typename T, int KS, int PT, bool DIAG>
__device__ __forceinline__ void psi2_patch(int N, int M, int Q, int B, const TIN *__restrict__ z,
                                           const TIN *__restrict__ mu, const TIN *__restrict__ s,
                                           const TIN *__restrict__ gamma, const TIN *__restrict__ alpha,
                                           T *__restrict__ part, int Mp, int n_per_split, int b, int sp, int pi, int pj,
                                           unsigned char *smem_raw) {
    typedef Psi2Lds<T, KS, PT> G;
    constexpr int PS = G::PS, KP = G::KP, ZLD = G::ZLD, PLD = G::PLD, NR = G::NR;
    typedef typename Mfma<T>::acc_t acc_t;
    T *zs = reinterpret_cast<T *>(smem_raw);        // [2*PS][ZLD] centred z rows: m-block then m'-block (0 beyond Q / M)
    T *zc = zs + 2 * PS * ZLD;                       // [Q]  column means of z
    T *gq = zc + DPGP_MAX_Q + 2;                     // [Q]  gamma_b
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kk = lane >> 4;
    const int m_base = pi * PS, mp_base = pj * PS;
    T *wp = zs + G::OFF_W + wv * G::WSZ;             // this wave's private region
    T *xa = wp;                                      // [NR][KP]  X[n,k]  (0 for k >= Q)
    T *w4 = xa + NR * KP;                            // [NR][KP]  1/4 w log2e      (0 for k >= Q)
    T *tm = w4 + NR * KP;                            // [NR][KP]  2 (mu - c)
    T *cn = tm + NR * KP;                            // [NR][KP]  (1/2 w (mu-c)^2 - 1/4 log den) log2e
    T *pm = cn + NR * KP;                            // [NR][PLD] P[n, m-block | m'-block]

    // ---- prologue (workgroup-wide): gamma_b, z column means, centred z rows of both blocks -----------------------
    if (t < Q) gq[t] = (T)gamma[(size_t)b * Q + t];
    const double *etab = reinterpret_cast<const double *>(zs + G::OFF_TAB);       // fp64 only (the fp32 exp2 is v_exp_f32)
    if (sizeof(T) == 8) dpgp_exp2_tab_init(reinterpret_cast<double *>(zs + G::OFF_TAB));
    block_column_means(z, M, Q, zc, reinterpret_cast<double *>(zs + G::OFF_W));   // scratch: the (not yet used) wave regions
    for (int e = t; e < 2 * PS * ZLD; e += 256) {
        int r = e / ZLD, k = e - r * ZLD;
        int m = (r < PS) ? (m_base + r) : (mp_base + r - PS);
        zs[e] = (k < Q && m < M) ? (T)z[(size_t)m * Q + k] - zc[k] : (T)0;
    }
    __syncthreads();

    // ---- per-lane constant MFMA operands ---------------------------------------------------------------------
    // A side (rows m of tile I): z value for k < Q, else 0;  B side (cols m' of tile J): z value for k < Q, 1 for k == Q.
    T zA[PT][KS], zB[PT][KS];
    T cPa[KS], c1a[KS], cPb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + kk;
        cPa[ks] = (k == Q) ? (T)1 : (T)0;
        c1a[ks] = (k == Q + 1) ? (T)1 : (T)0;
        cPb[ks] = (k == Q + 1) ? (T)1 : (T)0;
#pragma unroll
        for (int I = 0; I < PT; ++I) {
            zA[I][ks] = zs[(16 * I + li) * ZLD + k];                         // already 0 for k >= Q
            zB[I][ks] = (k == Q) ? (T)1 : zs[(PS + 16 * I + li) * ZLD + k];
        }
    }

    acc_t acc[PT][PT];
#pragma unroll
    for (int I = 0; I < PT; ++I)
#pragma unroll
        for (int J = 0; J < PT; ++J) acc[I][J] = (acc_t){0, 0, 0, 0};

    const int nbeg = sp * n_per_split, nend = min(N, nbeg + n_per_split);
    constexpr int pb_off = DIAG ? 0 : PS;
    constexpr int NCOL = DIAG ? PS : 2 * PS;         // columns whose P rows are needed (m-block, then m'-block)
    constexpr int CPL = (NCOL + 63) / 64;            // ... per lane

    for (int nc = nbeg + wv; nc < nend; nc += 4 * NR) {     // this wave's chunk: rows nc, nc+4, ..., nc+4(NR-1)
        // ---- phase A: per-(row,k) factors ----
#pragma unroll
        for (int e0 = 0; e0 < NR * KP; e0 += 64) {
            const int e = e0 + lane;
            if (e < NR * KP) {
                const int r = e / KP, k = e - r * KP, n = nc + 4 * r;
                T vx = 0, vw = 0, vt = 0, vc = 0;
                if (k < Q) {
                    if (n < nend) {
                        const T g = gq[k];
                        const T sv = (T)s[(size_t)n * Q + k];
                        const T mc = (T)mu[(size_t)n * Q + k] - zc[k];
                        const T den = (T)2 * g * sv + (T)1;
                        const T w = g / den;
                        vx = (T)(-0.5 * DPGP_LOG2E) * w;
                        vw = (T)(0.25 * DPGP_LOG2E) * w;
                        vt = (T)2 * mc;
                        vc = (T)DPGP_LOG2E * ((T)0.5 * w * mc * mc - (T)0.25 * dpgp_log(den));
                    } else {
                        vc = (T)-1.0e30;             // rows past the end of this split contribute exp2(-huge) = 0
                    }
                }
                xa[e] = vx; w4[e] = vw; tm[e] = vt; cn[e] = vc;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // row sums C_r of cn (lanes 0..NR-1), kept in registers across the overwrite
        T crv = 0;
        if (lane < NR) {
#pragma unroll
            for (int k = 0; k < KP; ++k) crv += cn[lane * KP + k];
        }
        // ---- phase B: P[r, c] = C_r - sum_k w4_rk (z_ck - tm_rk)^2 for this lane's column of each block ----
        T zr[CPL][KP];
#pragma unroll
        for (int h = 0; h < CPL; ++h)
#pragma unroll
            for (int k = 0; k < KP; ++k) zr[h][k] = (h * 64 + lane < NCOL) ? zs[(h * 64 + lane) * ZLD + k] : (T)0;
#pragma unroll 2
        for (int r = 0; r < NR; ++r) {
            const T c0 = __shfl(crv, r, 64);
            T p[CPL];
#pragma unroll
            for (int h = 0; h < CPL; ++h) p[h] = c0;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const T wk = w4[r * KP + k], tk = tm[r * KP + k];
#pragma unroll
                for (int h = 0; h < CPL; ++h) {
                    const T d = zr[h][k] - tk;
                    p[h] = fma(-wk * d, d, p[h]);
                }
            }
#pragma unroll
            for (int h = 0; h < CPL; ++h)
                if (h * 64 + lane < NCOL) pm[r * PLD + h * 64 + lane] = p[h];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- phase C: per row, E = A_n B_n on the matrix cores, psi2 += exp2(E) ----
#pragma unroll 1
        for (int r = 0; r < NR; ++r) {
            T a[PT][KS], bq[PT][KS];
            T xk[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xk[ks] = xa[r * KP + 4 * ks + kk];
            T pa[PT], pb[PT];
#pragma unroll
            for (int I = 0; I < PT; ++I) {
                pa[I] = pm[r * PLD + 16 * I + li];
                pb[I] = pm[r * PLD + pb_off + 16 * I + li];
            }
#pragma unroll
            for (int I = 0; I < PT; ++I)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    if (ks < KS - 2) {          // k < Q for certain: pure z steps
                        a[I][ks] = zA[I][ks] * xk[ks];
                        bq[I][ks] = zB[I][ks];
                    } else {
                        a[I][ks] = fma(zA[I][ks], xk[ks], fma(cPa[ks], pa[I], c1a[ks]));
                        bq[I][ks] = fma(cPb[ks], pb[I], zB[I][ks]);
                    }
                }
#pragma unroll
            for (int I = 0; I < PT; ++I) {
                acc_t c[PT];
#pragma unroll
                for (int J = 0; J < PT; ++J) c[J] = (acc_t){0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int J = 0; J < PT; ++J)
                        if (!(DIAG && J > I)) c[J] = Mfma<T>::mma(a[I][ks], bq[J][ks], c[J]);
#pragma unroll
                for (int J = 0; J < PT; ++J)
                    if (!(DIAG && J > I)) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[I][J][v] += dpgp_exp2_hot(c[J][v], etab);
                    }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // next chunk's phase A overwrites what phase C read
    }

    // ---- epilogue: sum the 4 waves' accumulators through LDS, apply alpha^2 exp2(beta_mm'), store the lower tiles ----
    T *red = zs + G::OFF_W;   // [4 waves][PT][4][64] (reuses the per-wave regions)
    const T al = (T)alpha[b];
    const T al2 = al * al;
    T *out = part + ((size_t)sp * B + b) * (size_t)Mp * Mp;
#pragma unroll
    for (int I = 0; I < PT; ++I) {
        __syncthreads();
#pragma unroll
        for (int J = 0; J < PT; ++J)
#pragma unroll
            for (int v = 0; v < 4; ++v) red[((wv * PT + J) * 4 + v) * 64 + lane] = acc[I][J][v];
        __syncthreads();
        for (int e = t; e < PT * 256; e += 256) {
            const int J = e >> 8, v = (e >> 6) & 3, l = e & 63;
            if (DIAG && J > I) continue;
            T sum = 0;
#pragma unroll
            for (int w_ = 0; w_ < 4; ++w_) sum += red[((w_ * PT + J) * 4 + v) * 64 + l];
            const int row = 16 * I + Mfma<T>::row(l, v), col = 16 * J + (l & 15);
            const int m = m_base + row, mp = mp_base + col;
            if (m < Mp && mp < Mp) {
                T val = 0;
                if (m < M && mp < M) {
                    const T *z1 = zs + row * ZLD, *z2 = zs + (PS + col) * ZLD;
                    T bsum = 0;
                    for (int q = 0; q < Q; ++q) {
                        const T d = z1[q] - z2[q];
                        bsum += gq[q] * d * d;
                    }
                    val = al2 * sum * dpgp_exp2((T)(-0.25 * DPGP_LOG2E) * bsum);
                }
                out[(size_t)m * Mp + mp] = val;
            }
        }
    }
}

// ===============================================================================================================
// fp32 psi2 on the REAL matrix pipe: f16 hi/lo-split operands, v_mfma_f32_16x16x32_f16, fp32 accumulate.
//
// Why: on gfx950 the fp32-input MFMA runs at the vector rate and does NOT co-execute with VALU work (measured:
// SQ_VALU_MFMA_COEXEC_CYCLES = 0, micro-benchmark scratch/ubench/coexec.hip: mfma 26.7 ms + valu 15.0 ms -> 44.9 ms
// together), so the fp32-MFMA kernel above pays (matrix + exp) serially.  The f16 MFMA is a separate pipe (14.9 + 14.9
// -> 23.3 ms, the remainder is its 8 issue cycles per instruction).  Each fp32 operand x is split as x = hi + lo with
// hi = f16(x), lo = f16(x - hi) (22 significant bits; f16 subnormals are honoured by the MFMA, verified on the GPU) and
//      a*b ~= ah*bh + ah*bl + al*bh          (the dropped al*bl is 2^-22 relative)
// so one fp32 product becomes three f16 K-columns.  Exponent error vs fp64 is the same ~1e-6 (log2 units) as the fp32
// kernel's (scratch/f16split_sim.py).  |P| must stay below the f16 range: it is clamped to +-30000 (2^-30000 = 0).
//
// K layout: 8-slot groups, group g = kk + 4 s (kk = lane >> 4, s = K-step of 32), two latent dims q0 = 2g, q1 = 2g+1:
//      A slots { ah0, ah1, ah0, ah1, al0, al1, spA0, spA1 }      A_q = X[n,q] * z[m,q]
//      B slots { bh0, bh1, bl0, bl1, bh0, bh1, spB0, spB1 }      B_q = z[m',q]
//   spare slots: group 0: A = (Ph, Pl) of P[n,m], B = (1, 1);  group 1: A = (1, 1), B = (Ph, Pl) of P[n,m'];  else 0.
// ===============================================================================================================
typedef _Float16 dpgp_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 dpgp_h8 __attribute__((ext_vector_type(8)));
typedef unsigned dpgp_u4 __attribute__((ext_vector_type(4)));
#define DPGP_H2_ONES 0x3C003C00u
#ifndef PSI2_F16_WAVES
#define PSI2_F16_WAVES 2   // waves per SIMD the register allocator must allow (3 = 168 VGPRs spills in the hot loop: slower)
#endif

__device__ __forceinline__ unsigned pack_h2(float a, float b) {
    dpgp_h2 h = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, h);
}
// hi/lo split of two fp32 products x0*z0, x1*z1 into packed f16 pairs:
//   p = x * z (packed fp32 multiply);  hi = (f16(p0), f16(p1))  (v_cvt_pk_f16_f32, round to nearest)
//   lo = (f16(x0*z0 - hi0), f16(x1*z1 - hi1))                   (exact fp32 FMA reading the f16 addend through op_sel)
// Five instructions of the cheap VOP3 class (measured on gfx950, scratch/ubench/issue2.hip: ~4.5 cycles each with two
// waves per SIMD); the four-instruction v_fma_mixlo/mixhi_f16 form runs at the transcendental rate (8.5 cycles each).
// Plain VALU RAW dependencies only: the hardware interlocks them (no wait states).
typedef float dpgp_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_products(dpgp_f2 x, dpgp_f2 z, unsigned &hi, unsigned &lo) {
    dpgp_f2 p;
    unsigned h, l;
    float l0, l1;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(z));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(p[0]), "v"(p[1]));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(x[0]), "v"(z[0]), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(x[1]), "v"(z[1]), "v"(h));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(l0), "v"(l1));
    hi = h;
    lo = l;
}
__device__ __forceinline__ void split_products(float x0, float z0, float x1, float z1, unsigned &hi, unsigned &lo) {
    split_products((dpgp_f2){x0, x1}, (dpgp_f2){z0, z1}, hi, lo);
}

template <int KB> struct Psi2F16Lds {
    static constexpr int PT = 4, PS = 64;
    static constexpr int KF = (KB + 1) / 2;                  // K-steps of 32 (4 groups of 2 latent dims each)
    static constexpr int KQ = 4 * KB;                        // latent dims padded to a multiple of 4 (phase B unroll)
    static constexpr int XLD = 8 * KF;                       // row stride of the per-(n,q) arrays (>= KQ)
    static constexpr int ZLD = ((KQ / 4) & 1) ? KQ : KQ + 4;
    static constexpr int PLD = 2 * PS + 4;
    static constexpr int NR = 8;
    static constexpr int WSZ = 4 * NR * XLD + NR * PLD + 4;  // xa, w4, tm, cn, packed P rows, 2 constant words (+pad)
    static constexpr int OFF_W = 2 * PS * ZLD + 2 * (DPGP_MAX_Q + 2);
    static constexpr int FILL = OFF_W + 4 * WSZ;
    static constexpr int RED = OFF_W + 4 * PT * 4 * 64;
    static constexpr int ELEMS = FILL > RED ? FILL : RED;
};

template <typename TIN, int KB, bool DIAG>
__device__ __forceinline__ void psi2_patch_f16(int N, int M, int Q, int B, const TIN *__restrict__ z,
                                               const TIN *__restrict__ mu, const TIN *__restrict__ s,
                                               const TIN *__restrict__ gamma, const TIN *__restrict__ alpha,
                                               float *__restrict__ part, int Mp, int n_per_split, int b, int sp, int pi,
                                               int pj, unsigned char *smem_raw) {
    typedef Psi2F16Lds<KB> G;
    constexpr int PT = G::PT, PS = G::PS, KF = G::KF, KQ = G::KQ, XLD = G::XLD, ZLD = G::ZLD, PLD = G::PLD, NR = G::NR;
    float *zs = reinterpret_cast<float *>(smem_raw);     // [2*PS][ZLD] centred z rows: m-block then m'-block
    float *zc = zs + 2 * PS * ZLD;                        // [Q] column means of z
    float *gq = zc + DPGP_MAX_Q + 2;                      // [Q] gamma_b
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kk = lane >> 4;
    const int m_base = pi * PS, mp_base = pj * PS;
    float *wp = zs + G::OFF_W + wv * G::WSZ;
    float *xa = wp;                                       // [NR][XLD]  X[n,q] = -1/2 w log2e  (0 for q >= Q)
    float *w4 = xa + NR * XLD;                            // [NR][XLD]  1/4 w log2e
    float *tm = w4 + NR * XLD;                            // [NR][XLD]  2 (mu - c)
    float *cn = tm + NR * XLD;                            // [NR][XLD]  (1/2 w (mu-c)^2 - 1/4 log den) log2e
    unsigned *pw = reinterpret_cast<unsigned *>(cn + NR * XLD);   // [NR][PLD] packed (Ph, Pl) + [2] constants
    constexpr int CONST_ONE = NR * PLD, CONST_ZERO = NR * PLD + 1;

    if (t < Q) gq[t] = (float)gamma[(size_t)b * Q + t];
    block_column_means(z, M, Q, zc, reinterpret_cast<double *>(zs + G::OFF_W));   // scratch: the (not yet used) wave regions
    for (int e = t; e < 2 * PS * ZLD; e += 256) {
        int r = e / ZLD, k = e - r * ZLD;
        int m = (r < PS) ? (m_base + r) : (mp_base + r - PS);
        zs[e] = (k < Q && m < M) ? (float)z[(size_t)m * Q + k] - zc[k] : 0.0f;
    }
    if (lane == 0) { pw[CONST_ONE] = DPGP_H2_ONES; pw[CONST_ZERO] = 0u; }
    __syncthreads();

    // ---- per-lane constants: A-side z values (fp32) and B-side split z (packed f16) of this lane's groups ----
    float zA[PT][KF][2];
    unsigned bh[PT][KF], bl[PT][KF];
#pragma unroll
    for (int ks = 0; ks < KF; ++ks) {
        const int q0 = 2 * (kk + 4 * ks);                 // < XLD <= ZLD? zs rows hold ZLD >= KQ entries; guard by Q
#pragma unroll
        for (int I = 0; I < PT; ++I) {
            const float a0 = (q0 < Q) ? zs[(16 * I + li) * ZLD + q0] : 0.0f;
            const float a1 = (q0 + 1 < Q) ? zs[(16 * I + li) * ZLD + q0 + 1] : 0.0f;
            zA[I][ks][0] = a0;
            zA[I][ks][1] = a1;
            const float b0 = (q0 < Q) ? zs[(PS + 16 * I + li) * ZLD + q0] : 0.0f;
            const float b1 = (q0 + 1 < Q) ? zs[(PS + 16 * I + li) * ZLD + q0 + 1] : 0.0f;
            const _Float16 h0 = (_Float16)b0, h1 = (_Float16)b1;
            dpgp_h2 hv = {h0, h1};
            bh[I][ks] = __builtin_bit_cast(unsigned, hv);
            bl[I][ks] = pack_h2(b0 - (float)h0, b1 - (float)h1);
        }
    }
    // spare-slot sources (K-step 0 only): lane group 0 carries P[n,m] on the A side, group 1 carries P[n,m'] on the B side
    constexpr int pb_off = DIAG ? 0 : PS;
    const int rmulA = (kk == 0) ? PLD : 0, rmulB = (kk == 1) ? PLD : 0;
    int offA[PT], offB[PT];
#pragma unroll
    for (int I = 0; I < PT; ++I) {
        offA[I] = (kk == 0) ? 16 * I + li : (kk == 1 ? CONST_ONE : CONST_ZERO);
        offB[I] = (kk == 1) ? pb_off + 16 * I + li : (kk == 0 ? CONST_ONE : CONST_ZERO);
    }

    f32x4 acc[PT][PT];
#pragma unroll
    for (int I = 0; I < PT; ++I)
#pragma unroll
        for (int J = 0; J < PT; ++J) acc[I][J] = (f32x4){0, 0, 0, 0};

    const int nbeg = sp * n_per_split, nend = min(N, nbeg + n_per_split);
    constexpr int NCOL = DIAG ? PS : 2 * PS;
    constexpr int CPL = (NCOL + 63) / 64;

    // q(X) rows of the first chunk; every later chunk is fetched one chunk ahead (global-load latency off the critical path)
    constexpr int NPA = (NR * XLD + 63) / 64;
    TIN pf_s[NPA], pf_m[NPA];
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
        const int e = 64 * u + lane, r = e / XLD, k = e - r * XLD, n = nbeg + wv + 4 * r;
        const bool ok = (e < NR * XLD) && (k < Q) && (n < nend);
        pf_s[u] = ok ? s[(size_t)n * Q + k] : (TIN)1;
        pf_m[u] = ok ? mu[(size_t)n * Q + k] : (TIN)0;
    }
    for (int nc = nbeg + wv; nc < nend; nc += 4 * NR) {
        // ---- phase A: per-(row,q) factors ----
#pragma unroll
        for (int u = 0; u < NPA; ++u) {
            const int e = 64 * u + lane;
            if (e < NR * XLD) {
                const int r = e / XLD, k = e - r * XLD, n = nc + 4 * r;
                float vx = 0, vw = 0, vt = 0, vc = 0;
                if (k < Q) {
                    if (n < nend) {
                        const float g = gq[k];
                        const float sv = (float)pf_s[u];
                        const float mc = (float)pf_m[u] - zc[k];
                        const float den = 2.0f * g * sv + 1.0f;
                        const float w = g / den;
                        vx = (float)(-0.5 * DPGP_LOG2E) * w;
                        vw = (float)(0.25 * DPGP_LOG2E) * w;
                        vt = 2.0f * mc;
                        vc = (float)DPGP_LOG2E * (0.5f * w * mc * mc - 0.25f * dpgp_log(den));
                    } else {
                        vc = -1.0e30f;
                    }
                }
                xa[e] = vx; w4[e] = vw; tm[e] = vt; cn[e] = vc;
            }
        }
#pragma unroll
        for (int u = 0; u < NPA; ++u) {      // issue the next chunk's loads now; they land during phases B and C
            const int e = 64 * u + lane, r = e / XLD, k = e - r * XLD, n = nc + 4 * NR + 4 * r;
            const bool ok = (e < NR * XLD) && (k < Q) && (n < nend);
            pf_s[u] = ok ? s[(size_t)n * Q + k] : (TIN)1;
            pf_m[u] = ok ? mu[(size_t)n * Q + k] : (TIN)0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float crv = 0;
        if (lane < NR) {
#pragma unroll
            for (int k = 0; k < KQ; ++k) crv += cn[lane * XLD + k];
        }
        // ---- phase B: P[r, c], clamped to the f16 range and stored as packed (hi, lo) ----
        float zr[CPL][KQ];
#pragma unroll
        for (int h = 0; h < CPL; ++h)
#pragma unroll
            for (int k = 0; k < KQ; ++k) zr[h][k] = (h * 64 + lane < NCOL) ? zs[(h * 64 + lane) * ZLD + k] : 0.0f;
#pragma unroll 2
        for (int r = 0; r < NR; ++r) {
            const float c0 = __shfl(crv, r, 64);
            float p[CPL];
#pragma unroll
            for (int h = 0; h < CPL; ++h) p[h] = c0;
#pragma unroll
            for (int k = 0; k < KQ; ++k) {
                const float wk = w4[r * XLD + k], tk = tm[r * XLD + k];
#pragma unroll
                for (int h = 0; h < CPL; ++h) {
                    const float d = zr[h][k] - tk;
                    p[h] = fmaf(-wk * d, d, p[h]);
                }
            }
#pragma unroll
            for (int h = 0; h < CPL; ++h) {
                const float pc = fminf(fmaxf(p[h], -30000.0f), 30000.0f);
                const _Float16 ph = (_Float16)pc;
                dpgp_h2 hv = {ph, (_Float16)(pc - (float)ph)};
                if (h * 64 + lane < NCOL) pw[r * PLD + h * 64 + lane] = __builtin_bit_cast(unsigned, hv);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- phase C: per row, exponent tile on the f16 matrix pipe, psi2 += exp2(E) ----
#pragma unroll 1
        for (int r = 0; r < NR; ++r) {
            float xk[KF][2];
#pragma unroll
            for (int ks = 0; ks < KF; ++ks) {
                xk[ks][0] = xa[r * XLD + 2 * (kk + 4 * ks)];
                xk[ks][1] = xa[r * XLD + 2 * (kk + 4 * ks) + 1];
            }
            unsigned spA[PT], spB[PT];
#pragma unroll
            for (int I = 0; I < PT; ++I) {
                spA[I] = pw[r * rmulA + offA[I]];
                spB[I] = pw[r * rmulB + offB[I]];
            }
#pragma unroll
            for (int I = 0; I < PT; ++I) {
                dpgp_u4 aop[KF];
#pragma unroll
                for (int ks = 0; ks < KF; ++ks) {
                    unsigned hi, lo;
                    split_products(xk[ks][0], zA[I][ks][0], xk[ks][1], zA[I][ks][1], hi, lo);
                    aop[ks] = (dpgp_u4){hi, hi, lo, ks == 0 ? spA[I] : 0u};
                }
                f32x4 c[PT];
#pragma unroll
                for (int J = 0; J < PT; ++J) c[J] = (f32x4){0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KF; ++ks)
#pragma unroll
                    for (int J = 0; J < PT; ++J)
                        if (!(DIAG && J > I)) {
                            const dpgp_u4 bop = {bh[J][ks], bl[J][ks], bh[J][ks], ks == 0 ? spB[J] : 0u};
                            c[J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dpgp_h8, aop[ks]),
                                                                          __builtin_bit_cast(dpgp_h8, bop), c[J], 0, 0, 0);
                        }
#pragma unroll
                for (int J = 0; J < PT; ++J)
                    if (!(DIAG && J > I)) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[I][J][v] += dpgp_exp2(c[J][v]);
                    }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }

    // ---- epilogue (as in the fp32-MFMA kernel) ----
    float *red = zs + G::OFF_W;
    const float al = (float)alpha[b];
    const float al2 = al * al;
    float *out = part + ((size_t)sp * B + b) * (size_t)Mp * Mp;
#pragma unroll
    for (int I = 0; I < PT; ++I) {
        __syncthreads();
#pragma unroll
        for (int J = 0; J < PT; ++J)
#pragma unroll
            for (int v = 0; v < 4; ++v) red[((wv * PT + J) * 4 + v) * 64 + lane] = acc[I][J][v];
        __syncthreads();
        for (int e = t; e < PT * 256; e += 256) {
            const int J = e >> 8, v = (e >> 6) & 3, l = e & 63;
            if (DIAG && J > I) continue;
            float sum = 0;
#pragma unroll
            for (int w_ = 0; w_ < 4; ++w_) sum += red[((w_ * PT + J) * 4 + v) * 64 + l];
            const int row = 16 * I + Mfma<float>::row(l, v), col = 16 * J + (l & 15);
            const int m = m_base + row, mp = mp_base + col;
            if (m < Mp && mp < Mp) {
                float val = 0;
                if (m < M && mp < M) {
                    const float *z1 = zs + row * ZLD, *z2 = zs + (PS + col) * ZLD;
                    float bsum = 0;
                    for (int q = 0; q < Q; ++q) {
                        const float d = z1[q] - z2[q];
                        bsum += gq[q] * d * d;
                    }
                    val = al2 * sum * dpgp_exp2((float)(-0.25 * DPGP_LOG2E) * bsum);
                }
                out[(size_t)m * Mp + mp] = val;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Variant with the P rows ALSO on the matrix pipe (the default).  Phase B above costs ~78 VALU instructions per row;
// here P[n,m] = c'_n + sum_q ( a_nq z_mq^2 + b_nq z_mq ),  a = -1/4 w log2e,  b = w (mu-c) log2e,
// c'_n = -log2e sum_q ( 1/2 w (mu-c)^2 + 1/4 log den ), is one [16 n] x [K] x [K x 16 m] product per column tile with
// f16 hi/lo-split operands (K = 6Q + 2 slots, the slot layout of psi1T_y_f16_kernel), i.e. ~20 VALU instructions per
// row for clamping / splitting / storing the result.  Chunks are 16 rows (one MFMA tile).
// LDS (4-byte units): zs | zc | gq | bimg (128 columns x SL f16) | 4 x wave { xa[16][XLD] | aimg[16][SL] f16 |
// cq[16][QS] | pw[16][PLD] + 2 constants }.
// ---------------------------------------------------------------------------------------------------------------
#ifdef PSI2_PROFILE            // diagnostic build only (scratch/): phase clocks (10 ns units) of wave 0 of one workgroup
__device__ long long g_psi2_stamps[16];
__device__ long long g_psi2_wg[3 * 8192];     // per workgroup: start, end (10 ns units), XCC/SE/CU id
extern "C" void dpgp_debug_psi2_stamps(long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_psi2_stamps), sizeof(long long) * 16); }
extern "C" void dpgp_debug_psi2_wg(long long *out, int n) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_psi2_wg), sizeof(long long) * 3 * n); }
#define P2_WG(i) do { if (threadIdx.x == 0) { const int id__ = blockIdx.x; \
        if (id__ < 8192) { g_psi2_wg[3 * id__ + (i)] = wall_clock64(); if ((i) == 0) { unsigned hw__, xc__; \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw__)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xc__)); \
        g_psi2_wg[3 * id__ + 2] = ((long long)xc__ << 32) | hw__; } } } } while (0)
#define P2_ON (blockIdx.x == (PSI2_PROFILE) && threadIdx.x == 0)
#define P2_MARK(i) do { if (P2_ON) g_psi2_stamps[i] = wall_clock64(); } while (0)
#define P2_BEGIN() long long t2__ = wall_clock64()
#define P2_RESTART() t2__ = wall_clock64()
#define P2_END(i) do { if (P2_ON) g_psi2_stamps[i] += wall_clock64() - t2__; } while (0)
#else
#define P2_MARK(i)
#define P2_WG(i)
#define P2_BEGIN()
#define P2_RESTART()
#define P2_END(i)
#endif
// (hi, lo) f16 words of two fp32 values (phase B of the P-on-MFMA kernel): w = hi | lo << 16 with hi = f16(p),
// lo = f16(p - hi); six instructions per pair.
__device__ __forceinline__ void split_pair_words(float p0, float p1, unsigned &w0, unsigned &w1) {
    unsigned h, l;
    float l0, l1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(p0), "v"(p1));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(p0), "v"(h));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(p1), "v"(h));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(l0), "v"(l1));
    w0 = __builtin_amdgcn_perm(l, h, 0x05040100u);
    w1 = __builtin_amdgcn_perm(l, h, 0x07060302u);
}

struct Psi2PLayout {
    int SL, QS, off_bimg, off_wave, wsz, o_aimg, o_cq, o_pw, elems;
};
template <int KB> __host__ __device__ inline Psi2PLayout psi2p_layout(int Q) {
    typedef Psi2F16Lds<KB> G;
    Psi2PLayout L;
    L.SL = 32 * ((6 * Q + 2 + 31) / 32);                       // f16 slots per image row
    L.QS = G::KQ;                                              // row stride of the c' pieces (zero padded)
    L.off_bimg = 2 * G::PS * G::ZLD + 2 * (DPGP_MAX_Q + 2);
    L.off_wave = L.off_bimg + (KB >= 5 ? 0 : 128 * L.SL / 2);  // many latent dims: the column image stays in global memory
                                                               // (L2), with it in LDS only one workgroup fits a CU
    L.o_aimg = 16 * G::XLD;
    L.o_cq = L.o_aimg + 16 * L.SL / 2;
    L.o_pw = L.o_cq + 16 * L.QS;
    L.wsz = L.o_pw + 16 * G::PLD + 36;                         // + the constant word (1,1) at [0] and [32]
    const int fill = L.off_wave + 4 * L.wsz, red = L.off_wave + 4 * 2 * 16 * 64;
    L.elems = fill > red ? fill : red;
    return L;
}

template <typename TIN, int KB, bool DIAG>
__device__ __forceinline__ void psi2_patch_f16p(int N, int M, int Q, int B, const unsigned char *__restrict__ consts,
                                                const TIN *__restrict__ mu, const TIN *__restrict__ s,
                                                const TIN *__restrict__ gamma, const TIN *__restrict__ alpha,
                                                float *__restrict__ part, int Mp, int n_per_split, int b, int sp,
                                                int pi, int pj, unsigned char *smem_raw) {
    typedef Psi2F16Lds<KB> G;
    constexpr int PT = G::PT, PS = G::PS, KF = G::KF, XLD = G::XLD, ZLD = G::ZLD, PLD = G::PLD, NR = 16;
    const Psi2PLayout L = psi2p_layout<KB>(Q);
    const int SL = L.SL, QS = L.QS, kf1 = SL / 32;
    float *zs = reinterpret_cast<float *>(smem_raw);     // [2*PS][ZLD] centred z rows: m-block then m'-block
    float *zc = zs + 2 * PS * ZLD;
    float *gq = zc + DPGP_MAX_Q + 2;
    _Float16 *bimg = reinterpret_cast<_Float16 *>(zs + L.off_bimg);          // [128][SL] column-side image of P
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kk = lane >> 4;
    const int m_base = pi * PS, mp_base = pj * PS;
    float *wp = zs + L.off_wave + wv * L.wsz;
    float *xa = wp;                                                           // [16][XLD] X[n,q]
    _Float16 *aimg = reinterpret_cast<_Float16 *>(wp + L.o_aimg);             // [16][SL]  row-side image of P
    float *cq = wp + L.o_cq;                                                  // [16][QS]  per-(row,q) pieces of c'_n
    unsigned *pw = reinterpret_cast<unsigned *>(wp + L.o_pw);                 // [16][PLD] packed (Ph, Pl) + 2 constants
    constexpr int CONST_ONE = NR * PLD;
    constexpr int NCOL = DIAG ? PS : 2 * PS;

    P2_MARK(0);
    P2_WG(0);
#ifdef PSI2_PROFILE
    if (P2_ON) { g_psi2_stamps[2] = 0; g_psi2_stamps[3] = 0; g_psi2_stamps[4] = 0; }
#endif
    // ---- prologue: gamma of this output dim; the z-only constants (psi2_consts.h) are copied, not rebuilt ----
    const Psi2Consts C = psi2_consts_layout(M, Q);           // C.ZLD == ZLD, C.SL == SL
    const float *zs_g = reinterpret_cast<const float *>(consts + C.off_zs);
    const _Float16 *bimg_g = reinterpret_cast<const _Float16 *>(consts + C.off_bimg);
    if (t < G::KQ) gq[t] = (t < Q) ? (float)gamma[(size_t)b * Q + t] : 0.0f;
    if (t < 32) zc[t] = reinterpret_cast<const float *>(consts)[t];
    for (int e = t; e < 2 * PS * (ZLD / 4); e += 256) {
        const int r = e / (ZLD / 4), k4 = e - r * (ZLD / 4);
        const int m = (r < PS) ? (m_base + r) : (mp_base + r - PS);
        reinterpret_cast<f32x4 *>(zs)[e] = reinterpret_cast<const f32x4 *>(zs_g + (size_t)m * ZLD)[k4];
    }
    if (KB < 5) {
        const int v16 = SL / 8;                              // 16-byte vectors per image row
        for (int e = t; e < NCOL * v16; e += 256) {
            const int c = e / v16, k = e - c * v16;
            const int m = (c < PS) ? (m_base + c) : (mp_base + c - PS);
            reinterpret_cast<dpgp_u4 *>(bimg)[e] = reinterpret_cast<const dpgp_u4 *>(bimg_g + (size_t)m * SL)[k];
        }
    }
    for (int e = lane; e < 16 * SL / 2; e += 64) reinterpret_cast<unsigned *>(aimg)[e] = 0u;
    if (lane < 2) pw[CONST_ONE + 32 * lane] = DPGP_H2_ONES;
    __syncthreads();

    // ---- per-lane constants of the exponent GEMM: 32x32x16 tiles, lane = (row li5, K-half k2), K-step ks holds the two
    //      latent dims q0 = 2 (k2 + 2 ks), q0 + 1 in the slot layout of the header comment ----
    const int li5 = lane & 31, k2 = lane >> 5;
    dpgp_f2 zA[2][KB];
    unsigned bh[2][KB], bl[2][KB];
#pragma unroll
    for (int ks = 0; ks < KB; ++ks) {
        const int q0 = 2 * (k2 + 2 * ks);
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            zA[I][ks][0] = (q0 < Q) ? zs[(32 * I + li5) * ZLD + q0] : 0.0f;
            zA[I][ks][1] = (q0 + 1 < Q) ? zs[(32 * I + li5) * ZLD + q0 + 1] : 0.0f;
            const float b0 = (q0 < Q) ? zs[(PS + 32 * I + li5) * ZLD + q0] : 0.0f;
            const float b1 = (q0 + 1 < Q) ? zs[(PS + 32 * I + li5) * ZLD + q0 + 1] : 0.0f;
            const _Float16 h0 = (_Float16)b0, h1 = (_Float16)b1;
            dpgp_h2 hv = {h0, h1};
            bh[I][ks] = __builtin_bit_cast(unsigned, hv);
            bl[I][ks] = pack_h2(b0 - (float)h0, b1 - (float)h1);
        }
    }
    // spare slots of K-step 0: lanes k2 = 0 carry A = (Ph, Pl) of P[n,m], B = (1,1); lanes k2 = 1 carry A = (1,1),
    // B = (Ph, Pl) of P[n,m'].  The constant word sits at pw[CONST_ONE] and pw[CONST_ONE + 32] so that both row tiles
    // are one two-word read at a fixed distance.
    constexpr int pb_off = DIAG ? 0 : PS;
    const unsigned *pwA = pw + ((k2 == 0) ? li5 : CONST_ONE);
    const unsigned *pwB = pw + ((k2 == 1) ? pb_off + li5 : CONST_ONE);
    const int stepA = (k2 == 0) ? PLD : 0, stepB = (k2 == 1) ? PLD : 0;
    const float *xaq = xa + 2 * k2;

    f32x16 acc[2][2];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int J = 0; J < 2; ++J)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[I][J][v] = 0.0f;

    // f16 range guard: P[n,m] (and the row constant c'_n) are clamped to +-30000 so that their (hi, lo) f16 words exist;
    // a clamped value of a real row makes the exponent tiles wrong (the large terms of P[n,m] + P[n,m'] + cross term cancel).
    // The clamp is DETECTED and the whole patch of this workgroup comes out as NaN: never a silently wrong Psi2.  Reached
    // when |z - c| or |mu - c| is ~90 length scales or more; DPGP_ALGO_MFMA_F32 / fp64 have no such limit.
    bool oor = false;
    const int nbeg = sp * n_per_split, nend = min(N, nbeg + n_per_split);
    constexpr int NPA = (NR * XLD + 63) / 64;
    // q(X) rows of the next chunk stay in flight in registers during this chunk.  Raw input type: a conversion here would
    // wait for the loads on the spot (measured +70 us at config 3); only with many latent dims, where 256 VGPRs are short,
    // they are narrowed to fp32 at once.
    typedef typename std::conditional<(KB >= PSI2_PF_NARROW_KB), float, TIN>::type PF;
    PF pf_s[NPA], pf_m[NPA];
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
        const int e = 64 * u + lane, r = e / XLD, k = e - r * XLD, n = nbeg + wv + 4 * r;
        const bool ok = (e < NR * XLD) && (k < Q) && (n < nend);
        pf_s[u] = ok ? (PF)s[(size_t)n * Q + k] : (PF)1;
        pf_m[u] = ok ? (PF)mu[(size_t)n * Q + k] : (PF)0;
    }
    P2_MARK(1);
    for (int nc = nbeg + wv; nc < nend; nc += 4 * NR) {
        const int nr = min(NR, (nend - nc + 3) >> 2);               // rows of this wave in the batch that exist (>= 1)
        P2_BEGIN();
#ifdef PSI2_DIAG_SKIP_AB      // timing experiment only (wrong results): phases A and B for the first chunk only
        if (nc != nbeg + wv) goto phase_c;
#endif
        // ---- phase A: per-(row,q) factors: X for the exponent GEMM, the split (a, b) image and the c' pieces for P ----
#pragma unroll
        for (int u = 0; u < NPA; ++u) {
            const int e = 64 * u + lane;
            if (e < NR * XLD) {
                const int r = e / XLD, k = e - r * XLD, n = nc + 4 * r;
                float vx = 0.0f;
                if (k < Q) {
                    float a = 0.0f, bb = 0.0f, cc = (k == 0) ? -30000.0f : 0.0f;    // rows past the end: P = -30000
                    if (n < nend) {
                        const float g = gq[k];
                        const float mc = (float)pf_m[u] - zc[k];
                        const float den = 2.0f * g * (float)pf_s[u] + 1.0f;
                        const float w = g / den;
                        vx = (float)(-0.5 * DPGP_LOG2E) * w;
                        a = (float)(-0.25 * DPGP_LOG2E) * w;
                        bb = (float)DPGP_LOG2E * w * mc;
                        cc = (float)(-0.5 * DPGP_LOG2E) * w * mc * mc - 0.25f * __builtin_amdgcn_logf(den);   // (v_log_f32 = log2, den >= 1)
                    }
                    a = dpgp_pin(a);                          // (pinned before the (hi, lo) split: see dpgp_pin)
                    bb = dpgp_pin(bb);
                    const _Float16 ah = (_Float16)a, al = (_Float16)(a - (float)ah);
                    const _Float16 bhh = (_Float16)bb, bll = (_Float16)(bb - (float)bhh);
                    unsigned *dst = reinterpret_cast<unsigned *>(aimg + r * SL + 6 * k);     // slots {ah, ah, al, bh, bh, bl}
                    const dpgp_h2 w0 = {ah, ah}, w1 = {al, bhh}, w2 = {bhh, bll};
                    dst[0] = __builtin_bit_cast(unsigned, w0);
                    dst[1] = __builtin_bit_cast(unsigned, w1);
                    dst[2] = __builtin_bit_cast(unsigned, w2);
                    cq[r * QS + k] = cc;
                } else if (k < QS) {
                    cq[r * QS + k] = 0.0f;
                }
                xa[e] = vx;
            }
        }
#pragma unroll
        for (int u = 0; u < NPA; ++u) {      // next chunk's q(X) rows: in flight during the rest of this chunk
            const int e = 64 * u + lane, r = e / XLD, k = e - r * XLD, n = nc + 4 * NR + 4 * r;
            const bool ok = (e < NR * XLD) && (k < Q) && (n < nend);
            pf_s[u] = ok ? (PF)s[(size_t)n * Q + k] : (PF)1;
            pf_m[u] = ok ? (PF)mu[(size_t)n * Q + k] : (PF)0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 16) {
            float c = 0.0f;
#pragma unroll
            for (int q4 = 0; q4 < G::KQ / 4; ++q4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(cq + lane * G::KQ + 4 * q4);
                c += (v[0] + v[1]) + (v[2] + v[3]);
            }
            oor |= !(c >= -30000.0f);              // f16 range guard: see the epilogue
            c = fmaxf(c, -30000.0f);
            c = dpgp_pin(c);
            const _Float16 ch = (_Float16)c;
            const dpgp_h2 cw = {ch, (_Float16)(c - (float)ch)};
            *reinterpret_cast<unsigned *>(aimg + lane * SL + 6 * Q) = __builtin_bit_cast(unsigned, cw);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        P2_END(2);
        P2_RESTART();
        // ---- phase B: P rows of this chunk on the matrix pipe, stored as packed (hi, lo) f16 ----
        {
            constexpr int NJ = NCOL / 16;
            f32x4 pc[NJ];
#pragma unroll
            for (int J = 0; J < NJ; ++J) pc[J] = (f32x4){0, 0, 0, 0};
            if constexpr (KB >= 5) {
                // column image from global memory (L2 resident, shared by all workgroups), K-steps double buffered
                auto bload = [&](dpgp_h8 (&bq)[NJ], int ks) __attribute__((always_inline)) {
#pragma unroll
                    for (int J = 0; J < NJ; ++J) {
                        const int m = (J < PS / 16) ? (m_base + 16 * J + li) : (mp_base + 16 * (J - PS / 16) + li);
                        bq[J] = *reinterpret_cast<const dpgp_h8 *>(bimg_g + (size_t)m * SL + 32 * ks + 8 * kk);
                    }
                };
                auto bmma = [&](const dpgp_h8 (&bq)[NJ], int ks) __attribute__((always_inline)) {
                    const dpgp_h8 av = *reinterpret_cast<const dpgp_h8 *>(aimg + li * SL + 32 * ks + 8 * kk);
#pragma unroll
                    for (int J = 0; J < NJ; ++J) pc[J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bq[J], pc[J], 0, 0, 0);
                };
                dpgp_h8 b0[NJ], b1[NJ];
                bload(b0, 0);
                for (int ks = 0; ks < kf1; ks += 2) {
                    if (ks + 1 < kf1) bload(b1, ks + 1);
                    bmma(b0, ks);
                    if (ks + 2 < kf1) bload(b0, ks + 2);
                    if (ks + 1 < kf1) bmma(b1, ks + 1);
                }
            } else
            for (int ks = 0; ks < kf1; ++ks) {       // K-step outermost: NJ independent accumulation chains in flight
                const dpgp_h8 av = *reinterpret_cast<const dpgp_h8 *>(aimg + li * SL + 32 * ks + 8 * kk);
#pragma unroll
                for (int J = 0; J < NJ; ++J) {
                    const dpgp_h8 bv = *reinterpret_cast<const dpgp_h8 *>(bimg + (16 * J + li) * SL + 32 * ks + 8 * kk);
                    pc[J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, pc[J], 0, 0, 0);
                }
            }
#pragma unroll
            for (int J = 0; J < NJ; ++J)
#pragma unroll
                for (int v = 0; v < 4; v += 2) {
                    unsigned w0, w1;
                    oor |= !(fabsf(pc[J][v]) <= 30000.0f) | !(fabsf(pc[J][v + 1]) <= 30000.0f);
                    split_pair_words(fminf(fmaxf(pc[J][v], -30000.0f), 30000.0f),
                                     fminf(fmaxf(pc[J][v + 1], -30000.0f), 30000.0f), w0, w1);
                    pw[(4 * kk + v) * PLD + 16 * J + li] = w0;
                    pw[(4 * kk + v + 1) * PLD + 16 * J + li] = w1;
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        P2_END(3);
        P2_RESTART();
#ifdef PSI2_DIAG_SKIP_AB
    phase_c:
#endif
#ifndef PSI2_DIAG_SKIP_C      // timing experiment only (wrong results)
        // ---- phase C: per row, the 64 x 64 exponent patch as 32x32x16 f16 MFMA tiles, psi2 += exp2(E) ----
        // Software pipeline over the tiles T0 = (0,0), T1 = (0,1), T2 = (1,0), T3 = (1,1) of consecutive rows: the MFMA
        // chain of a tile is issued two exp stages before its results are read, so a wave never waits on the matrix pipe
        // (2 waves per SIMD are too few to hide that latency by switching).  Diagonal patches skip T1.
        {
            dpgp_f2 xk[KB];
            unsigned spA[2], spB[2], spBn[2];
            dpgp_u4 a0[KB], a1[KB];
            f32x16 c0, c1, c2, c3;
            auto load_row = [&](int r, unsigned (&sb)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int ks = 0; ks < KB; ++ks) xk[ks] = *reinterpret_cast<const dpgp_f2 *>(xaq + r * XLD + 4 * ks);
#pragma unroll
                for (int I = 0; I < 2; ++I) {
                    spA[I] = pwA[r * stepA + 32 * I];
                    sb[I] = pwB[r * stepB + 32 * I];
                }
            };
            auto split = [&](int I, dpgp_u4 (&aop)[KB]) __attribute__((always_inline)) {
#pragma unroll
                for (int ks = 0; ks < KB; ++ks) {
                    unsigned hi, lo;
                    split_products(xk[ks], zA[I][ks], hi, lo);
                    aop[ks] = (dpgp_u4){hi, hi, lo, ks == 0 ? spA[I] : 0u};
                }
            };
            auto issue = [&](const dpgp_u4 (&aop)[KB], int J, const unsigned (&sb)[2]) __attribute__((always_inline)) {
                f32x16 c;
#pragma unroll
                for (int v = 0; v < 16; ++v) c[v] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < KB; ++ks) {
                    const dpgp_u4 bop = {bh[J][ks], bl[J][ks], bh[J][ks], ks == 0 ? sb[J] : 0u};
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dpgp_h8, aop[ks]),
                                                               __builtin_bit_cast(dpgp_h8, bop), c, 0, 0, 0);
                }
                return c;
            };
            auto expacc = [&](f32x16 &a, const f32x16 &c) __attribute__((always_inline)) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < 16; ++v) a[v] += dpgp_exp2(c[v]);
                __builtin_amdgcn_sched_barrier(0);
            };
            if constexpr (KB >= 5) {
                // many latent dims: the operand registers of two row tiles plus four result tiles no longer fit 256 VGPRs
                // (the pipelined form spills ~200 dwords per chunk); one row tile at a time, results read right away
#pragma unroll 1
                for (int r = 0; r < nr; ++r) {
                    load_row(r, spB);
#pragma unroll
                    for (int I = 0; I < 2; ++I) {
                        split(I, a0);
                        c0 = issue(a0, 0, spB);
                        if (!(DIAG && I == 0)) c1 = issue(a0, 1, spB);
                        expacc(acc[I][0], c0);
                        if (!(DIAG && I == 0)) expacc(acc[I][1], c1);
                    }
                }
            } else {
            load_row(0, spB);
            split(0, a0);
            split(1, a1);
            c0 = issue(a0, 0, spB);
            if (!DIAG) c1 = issue(a0, 1, spB);
#pragma unroll 1
            for (int r = 0; r < nr - 1; ++r) {            // (rows past the end of the split are skipped: they would add exp2(-30000) = 0)
                c2 = issue(a1, 0, spB);
                load_row(r + 1, spBn);
                expacc(acc[0][0], c0);
                c3 = issue(a1, 1, spB);
                if (!DIAG) expacc(acc[0][1], c1);
                split(0, a0);
                c0 = issue(a0, 0, spBn);
                expacc(acc[1][0], c2);
                if (!DIAG) c1 = issue(a0, 1, spBn);
                split(1, a1);
                expacc(acc[1][1], c3);
                spB[0] = spBn[0];
                spB[1] = spBn[1];
            }
            c2 = issue(a1, 0, spB);
            expacc(acc[0][0], c0);
            c3 = issue(a1, 1, spB);
            if (!DIAG) expacc(acc[0][1], c1);
            expacc(acc[1][0], c2);
            expacc(acc[1][1], c3);
            }
        }
#endif
        P2_END(4);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    P2_MARK(5);

    // ---- epilogue ----
    float *red = zs + L.off_wave;
    const float al = __syncthreads_or(oor ? 1 : 0) ? __builtin_nanf("") : (float)alpha[b];     // range guard: NaN patch
    const float al2 = al * al;
    float *out = part + ((size_t)sp * B + b) * (size_t)Mp * Mp;
    // 32x32 result tile: register v of lane l holds (row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32)
#pragma unroll
    for (int I = 0; I < 2; ++I) {
        __syncthreads();
#pragma unroll
        for (int J = 0; J < 2; ++J)
            if (!(DIAG && J > I)) {
#pragma unroll
                for (int v = 0; v < 16; ++v) red[((wv * 2 + J) * 16 + v) * 64 + lane] = acc[I][J][v];
            }
        __syncthreads();
        // thread t sums registers v = wv + 4 i (i = 0..3) of lane l = t & 63 of both column tiles: rows
        // 32 I + 8 i + 4 (l / 32) + wv, columns 32 J + l % 32; the z rows are zero padded up to KQ
        const int l = t & 63;
#pragma unroll
        for (int J = 0; J < 2; ++J) {
            if (DIAG && J > I) continue;
            const int col = 32 * J + (l & 31), mp = mp_base + col;
            float z2[G::KQ], gg[G::KQ];
#pragma unroll
            for (int k = 0; k < G::KQ; ++k) {
                z2[k] = zs[(PS + col) * ZLD + k];
                gg[k] = gq[k];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = wv + 4 * i, row = 32 * I + 8 * i + 4 * (l >> 5) + wv, m = m_base + row;
                float sum = 0;
#pragma unroll
                for (int w_ = 0; w_ < 4; ++w_) sum += red[((w_ * 2 + J) * 16 + v) * 64 + l];
                if (m < Mp && mp < Mp) {
                    float val = 0;
                    if (m < M && mp < M) {
                        float bsum = 0;
#pragma unroll
                        for (int k = 0; k < G::KQ; ++k) {
                            const float d = zs[row * ZLD + k] - z2[k];
                            bsum += gg[k] * d * d;
                        }
                        val = al2 * sum * dpgp_exp2((float)(-0.25 * DPGP_LOG2E) * bsum);
                    }
                    out[(size_t)m * Mp + mp] = val;
                }
            }
        }
    }
    P2_MARK(6);
    P2_WG(1);
}

template <typename TIN, int KB>
__global__ __launch_bounds__(256, PSI2_F16_WAVES) void psi2_f16_kernel(int N, int M, int Q, int B, const TIN *__restrict__ z,
                                                       const unsigned char *__restrict__ consts,
                                                       const TIN *__restrict__ mu, const TIN *__restrict__ s,
                                                       const TIN *__restrict__ gamma, const TIN *__restrict__ alpha,
                                                       float *__restrict__ part, int Mp, int n_per_split,
                                                       int n_splits, ChainKTask task) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int nps = (Mp + 63) / 64;
    int item;
    if (psi2_task_1d(blockIdx.x, task.ws ? B : 0, task.last != 0, item)) {
        chain_k_task<2>(task, item, smem_raw);
        return;
    }
    int b, sp, pi, pj;
    b = item % B;
    sp = (item / B) % n_splits;
    psi2_patch_coords(nps, item / (B * n_splits), pi, pj);
#ifdef PSI2_P_VALU       // diagnostic build: P rows by the direct squared-distance form on the VALU
    if (pi == pj)
        psi2_patch_f16<TIN, KB, true>(N, M, Q, B, z, mu, s, gamma, alpha, part, Mp, n_per_split, b, sp, pi, pj, smem_raw);
    else
        psi2_patch_f16<TIN, KB, false>(N, M, Q, B, z, mu, s, gamma, alpha, part, Mp, n_per_split, b, sp, pi, pj, smem_raw);
#else
    if (pi == pj)
        psi2_patch_f16p<TIN, KB, true>(N, M, Q, B, consts, mu, s, gamma, alpha, part, Mp, n_per_split, b, sp, pi, pj,
                                        smem_raw);
    else
        psi2_patch_f16p<TIN, KB, false>(N, M, Q, B, consts, mu, s, gamma, alpha, part, Mp, n_per_split, b, sp, pi, pj,
                                        smem_raw);
#endif
}

template <typename TIN, typename T, int KS, int PT>
__global__ __launch_bounds__(256) void psi2_mfma_kernel(int N, int M, int Q, int B, const TIN *__restrict__ z,
                                                        const TIN *__restrict__ mu, const TIN *__restrict__ s,
                                                        const TIN *__restrict__ gamma, const TIN *__restrict__ alpha,
                                                        T *__restrict__ part, int Mp, int n_per_split,
                                                        ChainKTask task) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int zoff = task.ws ? 1 : 0;
    if (zoff && blockIdx.z == 0) {
        if (blockIdx.y == 0) chain_k_task<1>(task, blockIdx.x, smem_raw);
        return;
    }
    int b, sp, pi, pj;
    psi2_block_coords((Mp + 16 * PT - 1) / (16 * PT), zoff, b, sp, pi, pj);
    if (pi == pj)
        psi2_patch<TIN, T, KS, PT, true>(N, M, Q, B, z, mu, s, gamma, alpha, part, Mp, n_per_split, b, sp, pi, pj, smem_raw);
    else
        psi2_patch<TIN, T, KS, PT, false>(N, M, Q, B, z, mu, s, gamma, alpha, part, Mp, n_per_split, b, sp, pi, pj, smem_raw);
}

// Plain-VALU variant (cross-check of the matrix-core kernel): thread per (m, m') of the lower block-triangle, literal
// reference formula rbf_kernel.py:189-199 with the per-(n,q) factors staged through LDS.  Writes slab 0 only.
template <typename TIN, typename T>
__global__ __launch_bounds__(256) void psi2_plain_kernel(int N, int M, int Q, const TIN *__restrict__ z,
                                                         const TIN *__restrict__ mu, const TIN *__restrict__ s,
                                                         const TIN *__restrict__ gamma, const TIN *__restrict__ alpha,
                                                         T *__restrict__ part, int Mp) {
    __shared__ T w[PSI2_NT][DPGP_MAX_Q], mm[PSI2_NT][DPGP_MAX_Q], hl[PSI2_NT];
    const int b = blockIdx.y, t = threadIdx.x;
    const int e = blockIdx.x * 256 + t;
    const int m = e / Mp, mp = e - m * Mp;
    const bool active = (m < Mp) && ((mp >> 4) <= (m >> 4));
    const bool real = active && m < M && mp < M;
    T zb[DPGP_MAX_Q], t1 = 0;
    const TIN *g = gamma + (size_t)b * Q;
    for (int q = 0; q < Q; ++q) {
        T z1 = real ? (T)z[(size_t)m * Q + q] : (T)0, z2 = real ? (T)z[(size_t)mp * Q + q] : (T)0;
        zb[q] = (T)0.5 * (z1 + z2);
        t1 += (T)0.25 * (T)g[q] * (z1 - z2) * (z1 - z2);
    }
    T acc = 0;
    for (int n0 = 0; n0 < N; n0 += PSI2_NT) {
        __syncthreads();
        for (int i = t; i < PSI2_NT * Q; i += 256) {
            int r = i / Q, q = i - r * Q, n = n0 + r;
            T sv = n < N ? (T)s[(size_t)n * Q + q] : (T)1;
            w[r][q] = (T)g[q] / ((T)2 * (T)g[q] * sv + (T)1);
            mm[r][q] = n < N ? (T)mu[(size_t)n * Q + q] : (T)0;
        }
        if (t < PSI2_NT) {
            int n = n0 + t;
            T a = 0;
            for (int q = 0; q < Q; ++q) a += dpgp_log((T)2 * (T)g[q] * (n < N ? (T)s[(size_t)n * Q + q] : (T)1) + (T)1);
            hl[t] = (T)0.5 * a;
        }
        __syncthreads();
        const int nn = min(PSI2_NT, N - n0);
        for (int r = 0; r < nn; ++r) {
            T ex = hl[r] + t1;
            for (int q = 0; q < Q; ++q) {
                T d = mm[r][q] - zb[q];
                ex += w[r][q] * d * d;
            }
            acc += dpgp_exp2((T)(-DPGP_LOG2E) * ex);
        }
    }
    if (active) {
        T al = (T)alpha[b];
        part[(size_t)b * Mp * Mp + (size_t)m * Mp + mp] = real ? al * al * acc : (T)0;
    }
}

// ===============================================================================================================
// Backward pass, matrix-core version of the Psi2 term of stage B (grad.hip has the plain first version and the Psi1 / K_uu
// terms).  For every (output dim d, observation n) it needs, with W[a,m'] = G_d[a,m'] psi2(n,a,m') (G = d f_hat / d Psi2,
// symmetric), the column sums C[m'] = sum_a W and T'[m',q] = sum_a W z_aq; everything else follows per (n, m', q):
//     d/dz[m',q] += (gamma_q - a2_nq) T' + (2 a2_nq mu_nq - (gamma_q + a2_nq) z_m'q) C            a2 = gamma / (2 gamma S + 1)
//     S0 = sum C,  S1_q = sum C z,  S2_q = sum C z^2,  S3_q = sum z T'   (sums over m')  ->  d/dmu_n, d/dS_n, d/dgamma_d
// (the row-sum halves of the symmetric expressions are the column sums of the transposed patch, so every patch of the FULL
// M x M square is processed and contributes its column side only).  Workgroup = (64 x 64 patch, output dim, n-split) as in
// the forward kernel, same phases A (per-(n,q) factors) and B (P rows on the matrix pipe), same exponent tiles (32x32x16 f16
// MFMA, hi/lo-split operands); then, instead of accumulating exp2(E):  w = G' .* exp2(E)  (G' = G alpha^2 exp2(beta_mm') held
// in the registers the forward uses for its accumulators),  C and T' by per-lane FMAs over the 16 rows a lane holds of each
// tile (z rows broadcast from LDS), the two lane halves combined, and per observation a DPP reduction over the columns.
// Outputs are partial sums per workgroup, added up in fixed order by psi2_grad_reduce_* (deterministic).
// ===============================================================================================================
// W_LO (template parameter of the kernel): 1: the products w go to the second MFMA product as f16 hi + lo pairs (what the
// launcher uses); 0: rounded to f16, 0.8 ms faster at config 3 and 4e-6 of the largest gradient entry off there, but the sums
// over (a, m') cancel heavily when K_uu is ill-conditioned (G = -B^-1/2 - ...): 5e-3 on a random M = 70 problem -- not used
template <int KB> __host__ __device__ inline Psi2PLayout psi2g_layout(int Q) {
    typedef Psi2F16Lds<KB> G;
    Psi2PLayout L;
    L.SL = 32 * ((6 * Q + 2 + 31) / 32);
    L.QS = G::KQ;
    // [G' tiles 64 x 64 floats; the row block of z lives in its first PS * ZLD words during the prologue][column block of z]
    L.off_bimg = 64 * 64 + G::PS * G::ZLD + 2 * (DPGP_MAX_Q + 2);
    L.off_wave = L.off_bimg;                                   // (column image read from global memory)
    L.o_aimg = 16 * G::XLD;
    L.o_cq = L.o_aimg + 16 * L.SL / 2;
    L.o_pw = L.o_cq + 16 * L.QS;
    L.wsz = L.o_pw + 16 * G::PLD + 36 + 2 * 16 * G::XLD;       // + mu'[16][XLD], S[16][XLD]
    const int fill = L.off_wave + 4 * L.wsz, red = L.off_wave + 4 * 64 * (G::KQ + 1);
    L.elems = fill > red ? fill : red;
    return L;
}

// sum over the 32 lanes of each wave half, result in lanes 31 and 63 (DPP: row_shr 1, 2, 4, 8; row_bcast:15 into rows 1, 3)
__device__ __forceinline__ float half_sum_dpp(float x) {
    int v = __builtin_bit_cast(int, x);
#define DPP_ADD(ctrl, rmask)                                                                                          \
    v = __builtin_bit_cast(int, __builtin_bit_cast(float, v) +                                                        \
                                    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, v, ctrl, rmask, 0xf, false)))
    DPP_ADD(0x111, 0xf);
    DPP_ADD(0x112, 0xf);
    DPP_ADD(0x114, 0xf);
    DPP_ADD(0x118, 0xf);
    DPP_ADD(0x142, 0xa);
#undef DPP_ADD
    return __builtin_bit_cast(float, v);
}

// the same for four values at once as fused v_add_f32_dpp (the builtin form costs a v_mov of the `old` value, the DPP move
// and the add per step); the s_nop covers the 2 wait states a DPP read needs after a VALU write of its source
__device__ __forceinline__ void half_sum_dpp4(float &a, float &b, float &c, float &d) {
#define DPP_STEP4(ctrl)                                                                                                \
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " ctrl "\n\tv_add_f32_dpp %1, %1, %1 " ctrl "\n\tv_add_f32_dpp %2, %2, %2 " ctrl \
        "\n\tv_add_f32_dpp %3, %3, %3 " ctrl                                                                           \
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
    DPP_STEP4("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    DPP_STEP4("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    DPP_STEP4("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    DPP_STEP4("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1");
    DPP_STEP4("row_bcast:15 row_mask:0xa bank_mask:0xf");
#undef DPP_STEP4
}

template <int KB, int PSI2G_W_LO>
__global__ __launch_bounds__(256, (KB <= 3 ? 2 : 1)) void psi2_grad_kernel(int N, int M, int Q, int B, const unsigned char *__restrict__ consts,
                                                           const double *__restrict__ mu, const double *__restrict__ s,
                                                           const double *__restrict__ gamma, const double *__restrict__ alpha,
                                                           const double *__restrict__ GP, int Mp, int n_per_split,
                                                           int n_splits, float *__restrict__ dmu_part,
                                                           float *__restrict__ ds_part, double *__restrict__ dz_part,
                                                           double *__restrict__ dg_part) {
    typedef Psi2F16Lds<KB> G;
    constexpr int PS = G::PS, XLD = G::XLD, ZLD = G::ZLD, PLD = G::PLD, NR = 16, KQ = G::KQ;
    // latent dims per lane half, and the form of the second product: up to 12 latent dims z_hi, z_lo and the ones are 2 QH + 1
    // <= 13 of the 16 result rows a lane half owns (one A operand); beyond that z_hi and z_lo are two A operands (QH + 1 rows,
    // one more MFMA per step) and the kernel runs one workgroup per CU (register budget)
    constexpr bool SPLITZ = KB > 3;
    constexpr int QH = KQ / 2 > 15 ? 15 : KQ / 2, QHE = QH + (QH & 1), ONES = SPLITZ ? QH : 2 * QH;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const Psi2PLayout L = psi2g_layout<KB>(Q);
    const int SL = L.SL, QS = L.QS, kf1 = SL / 32;
    const int nps = (Mp + 63) / 64, npatch = nps * nps;
    const int item = blockIdx.x, b = item % B, sp = (item / B) % n_splits, patch = item / (B * n_splits);
    const int pi = patch / nps, pj = patch - pi * nps;
    float *zs = reinterpret_cast<float *>(smem_raw);          // [PS][ZLD] centred z rows of the row block pi (prologue only), then
    float *gl = zs;                                           // G' tiles [tile 2 I + J][v / 4][lane][v % 4] in the same place
    float *zcb = zs + 64 * 64;                                // [PS][ZLD] centred z rows of the column block pj
    float *zc = zcb + PS * ZLD;
    float *gq = zc + DPGP_MAX_Q + 2;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kk = lane >> 4;
    const int m_base = pi * PS, mp_base = pj * PS;
    float *wp = zs + L.off_wave + wv * L.wsz;
    float *xa = wp;
    _Float16 *aimg = reinterpret_cast<_Float16 *>(wp + L.o_aimg);
    float *cq = wp + L.o_cq;
    unsigned *pw = reinterpret_cast<unsigned *>(wp + L.o_pw);
    float *mus = wp + L.o_pw + 16 * PLD + 36, *sss = mus + 16 * XLD;          // mu - zc and S of this wave's 16 rows
    constexpr int CONST_ONE = NR * PLD;
    constexpr int NCOL = 2 * PS;
    const Psi2Consts C = psi2_consts_layout(M, Q);
    const float *zs_g = reinterpret_cast<const float *>(consts + C.off_zs);
    const _Float16 *bimg_g = reinterpret_cast<const _Float16 *>(consts + C.off_bimg);
    if (t < KQ) gq[t] = (t < Q) ? (float)gamma[(size_t)b * Q + t] : 0.0f;
    if (t < 32) zc[t] = reinterpret_cast<const float *>(consts)[t];
    for (int e = t; e < 2 * PS * (ZLD / 4); e += 256) {
        const int r = e / (ZLD / 4), k4 = e - r * (ZLD / 4);
        const int m = (r < PS) ? (m_base + r) : (mp_base + r - PS);
        reinterpret_cast<f32x4 *>(r < PS ? zs : zcb - PS * ZLD)[e] = reinterpret_cast<const f32x4 *>(zs_g + (size_t)m * ZLD)[k4];
    }
    for (int e = lane; e < 16 * SL / 2; e += 64) reinterpret_cast<unsigned *>(aimg)[e] = 0u;
    if (lane < 2) pw[CONST_ONE + 32 * lane] = DPGP_H2_ONES;
    __syncthreads();

    const int li5 = lane & 31, k2 = lane >> 5;
    dpgp_f2 zA[2][KB];
    unsigned bh[2][KB], bl[2][KB];
#pragma unroll
    for (int ks = 0; ks < KB; ++ks) {
        const int q0 = 2 * (k2 + 2 * ks);
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            zA[I][ks][0] = (q0 < Q) ? zs[(32 * I + li5) * ZLD + q0] : 0.0f;
            zA[I][ks][1] = (q0 + 1 < Q) ? zs[(32 * I + li5) * ZLD + q0 + 1] : 0.0f;
            const float b0 = (q0 < Q) ? zcb[(32 * I + li5) * ZLD + q0] : 0.0f;
            const float b1 = (q0 + 1 < Q) ? zcb[(32 * I + li5) * ZLD + q0 + 1] : 0.0f;
            const _Float16 h0 = (_Float16)b0, h1 = (_Float16)b1;
            dpgp_h2 hv = {h0, h1};
            bh[I][ks] = __builtin_bit_cast(unsigned, hv);
            bl[I][ks] = pack_h2(b0 - (float)h0, b1 - (float)h1);
        }
    }
    const unsigned *pwA = pw + ((k2 == 0) ? li5 : CONST_ONE);
    const unsigned *pwB = pw + ((k2 == 1) ? PS + li5 : CONST_ONE);
    const int stepA = (k2 == 0) ? PLD : 0, stepB = (k2 == 1) ? PLD : 0;
    const float *xaq = xa + 2 * k2;

    // G' tiles in the 32x32 MFMA result layout: register v of lane l = (row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32)
    // (wave w computes tile (w / 2, w % 2); they go to LDS, scaled by a common power of two so that the products
    //  w = G' psi2 fit f16 hi/lo pairs: max |G'| -> [2^13, 2^14))
    float gt[16];
    float unscale = 1.0f, gscale = 1.0f;
    {
        const int I = wv >> 1, J = wv & 1;
        const float al = (float)alpha[b], al2 = al * al;
        const double *Gd = GP + (size_t)b * Mp * Mp;
        float mx = 0.0f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = 32 * I + 8 * (v >> 2) + 4 * k2 + (v & 3), col = 32 * J + li5;
            const int m = m_base + row, mp = mp_base + col;
            float val = 0.0f;
            if (m < M && mp < M) {
                const float *z1 = zs + row * ZLD, *z2 = zcb + col * ZLD;
                float bsum = 0;
                for (int q = 0; q < Q; ++q) {
                    const float dd = z1[q] - z2[q];
                    bsum += gq[q] * dd * dd;
                }
                const double gv = Gd[(size_t)(m >= mp ? m : mp) * Mp + (m >= mp ? mp : m)];
                val = (float)gv * al2 * dpgp_exp2((float)(-0.25 * DPGP_LOG2E) * bsum);
            }
            gt[v] = val;
            mx = fmaxf(mx, fabsf(val));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) xa[0] = mx;                             // (this wave's own area, not in use before phase A)
    }
    // Second product on the matrix pipe: per observation  T'^T[slot, m'] = sum_a Zt[slot, a] w[a, m'],  the w tile taken as
    // the B operand straight from the registers the exponent tile arrived in (k-slot 8 (l / 32) + j of step s <-> register
    // 8 s + j <-> patch row 8 (2 s + j / 4) + 4 (l / 32) + j % 4, the A operand is permuted to match).  Output rows (slots,
    // 16 per lane half h): u < QH: z_hi[., QH h + u];  QH <= u < 2 QH: z_lo[., QH h + u - QH];  u = 2 QH: ones (column sums);
    // split form (more than 12 latent dims): u < QH: z_hi resp. z_lo in two operands, u = QH: ones.
    dpgp_h8 za[2][2], zal[SPLITZ ? 2 : 1][2];                  // (zal: the z_lo operand of the split form)
    {
        const int rho = li5, u = 4 * (rho >> 3) + (rho & 3), hh = (rho >> 2) & 1;
        const int qq = QH * hh + (SPLITZ ? u : (u < QH ? u : u - QH));
#pragma unroll
        for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int a = 32 * I + 8 * (2 * s_ + (j >> 2)) + 4 * k2 + (j & 3);
                    const float zv = (qq < Q && u < (SPLITZ ? QH : 2 * QH)) ? zs[a * ZLD + qq] : 0.0f;
                    const _Float16 zh = (_Float16)zv, zl = (_Float16)(zv - (float)zh);
                    _Float16 val = (_Float16)0.0f;
                    if (u < QH) val = zh;
                    else if (!SPLITZ && u < 2 * QH) val = zl;
                    else if (u == ONES) val = (_Float16)1.0f;
                    za[I][s_][j] = val;
                    if (SPLITZ) zal[I][s_][j] = (u < QH) ? zl : (_Float16)0.0f;
                }
    }
    __syncthreads();                                            // every wave is done with the row block of z: tiles may overwrite it
    {
        const float *mxs = zs + L.off_wave;
        const float mx = fmaxf(fmaxf(mxs[0], mxs[L.wsz]), fmaxf(mxs[2 * L.wsz], mxs[3 * L.wsz]));
        if (mx > 0.0f && mx < 3.0e38f) {
            int ex;
            (void)frexpf(mx, &ex);
            ex = max(-100, min(100, ex));
            gscale = ldexpf(1.0f, 14 - ex);
            unscale = ldexpf(1.0f, ex - 14);
        }
#pragma unroll
        for (int vq = 0; vq < 4; ++vq)
            *reinterpret_cast<f32x4 *>(gl + ((wv * 4 + vq) * 64 + lane) * 4) =
                (f32x4){gt[4 * vq] * gscale, gt[4 * vq + 1] * gscale, gt[4 * vq + 2] * gscale, gt[4 * vq + 3] * gscale};
    }
    __syncthreads();
    float dz[2][QH], dgam[QH];                                  // lane = column (l % 32), half l / 32 owns q = QH (l / 32) + i
#pragma unroll
    for (int i = 0; i < QH; ++i) { dz[0][i] = 0.0f; dz[1][i] = 0.0f; dgam[i] = 0.0f; }
    const int qb = QH * k2;
    float zcr[2][QH], gqr[QH], igq[QH];                         // this lane's columns' z', gamma and 1 / gamma of its latent dims
#pragma unroll
    for (int i = 0; i < QH; ++i) {
        zcr[0][i] = zcb[li5 * ZLD + qb + i];
        zcr[1][i] = zcb[(32 + li5) * ZLD + qb + i];
        gqr[i] = gq[qb + i];
        igq[i] = gqr[i] > 0.0f ? 1.0f / gqr[i] : 0.0f;
    }
    const size_t slot = (size_t)b * npatch + patch;
    bool oor = false;                                           // f16 range guard (see psi2_patch_f16p): poisons d/dgamma

    const int nbeg = sp * n_per_split, nend = min(N, nbeg + n_per_split);
    constexpr int NPA = (NR * XLD + 63) / 64;
    float pf_s[NPA], pf_m[NPA];
#pragma unroll
    for (int u = 0; u < NPA; ++u) {
        const int e = 64 * u + lane, r = e / XLD, k = e - r * XLD, n = nbeg + wv + 4 * r;
        const bool ok = (e < NR * XLD) && (k < Q) && (n < nend);
        pf_s[u] = ok ? (float)s[(size_t)n * Q + k] : 1.0f;
        pf_m[u] = ok ? (float)mu[(size_t)n * Q + k] : 0.0f;
    }
    for (int nc = nbeg + wv; nc < nend; nc += 4 * NR) {
        const int nr = min(NR, (nend - nc + 3) >> 2);               // rows of this wave in the batch that exist (>= 1)
        // ---- phase A (as the forward kernel) + mu', S of the rows for the finishing step ----
#pragma unroll
        for (int u = 0; u < NPA; ++u) {
            const int e = 64 * u + lane;
            if (e < NR * XLD) {
                const int r = e / XLD, k = e - r * XLD, n = nc + 4 * r;
                float vx = 0.0f, mc = 0.0f, sv = 1.0f;
                if (k < Q) {
                    float a = 0.0f, bb = 0.0f, cc = (k == 0) ? -30000.0f : 0.0f;
                    if (n < nend) {
                        const float gg = gq[k];
                        mc = pf_m[u] - zc[k];
                        sv = pf_s[u];
                        const float den = 2.0f * gg * sv + 1.0f;
                        const float w = gg / den;
                        vx = (float)(-0.5 * DPGP_LOG2E) * w;
                        a = (float)(-0.25 * DPGP_LOG2E) * w;
                        bb = (float)DPGP_LOG2E * w * mc;
                        cc = (float)(-0.5 * DPGP_LOG2E) * w * mc * mc - 0.25f * __builtin_amdgcn_logf(den);   // (v_log_f32 = log2, den >= 1)
                    }
                    a = dpgp_pin(a);                          // (pinned before the (hi, lo) split: see dpgp_pin)
                    bb = dpgp_pin(bb);
                    const _Float16 ah = (_Float16)a, al_ = (_Float16)(a - (float)ah);
                    const _Float16 bhh = (_Float16)bb, bll = (_Float16)(bb - (float)bhh);
                    unsigned *dst = reinterpret_cast<unsigned *>(aimg + r * SL + 6 * k);
                    const dpgp_h2 w0 = {ah, ah}, w1 = {al_, bhh}, w2 = {bhh, bll};
                    dst[0] = __builtin_bit_cast(unsigned, w0);
                    dst[1] = __builtin_bit_cast(unsigned, w1);
                    dst[2] = __builtin_bit_cast(unsigned, w2);
                    cq[r * QS + k] = cc;
                } else if (k < QS) {
                    cq[r * QS + k] = 0.0f;
                }
                xa[e] = vx;
                mus[e] = mc;
                sss[e] = sv;
            }
        }
#pragma unroll
        for (int u = 0; u < NPA; ++u) {
            const int e = 64 * u + lane, r = e / XLD, k = e - r * XLD, n = nc + 4 * NR + 4 * r;
            const bool ok = (e < NR * XLD) && (k < Q) && (n < nend);
            pf_s[u] = ok ? (float)s[(size_t)n * Q + k] : 1.0f;
            pf_m[u] = ok ? (float)mu[(size_t)n * Q + k] : 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 16) {
            float c = 0.0f;
#pragma unroll
            for (int q4 = 0; q4 < KQ / 4; ++q4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(cq + lane * KQ + 4 * q4);
                c += (v[0] + v[1]) + (v[2] + v[3]);
            }
            oor |= !(c >= -30000.0f);              // f16 range guard: see the epilogue
            c = fmaxf(c, -30000.0f);
            c = dpgp_pin(c);
            const _Float16 ch = (_Float16)c;
            const dpgp_h2 cw = {ch, (_Float16)(c - (float)ch)};
            *reinterpret_cast<unsigned *>(aimg + lane * SL + 6 * Q) = __builtin_bit_cast(unsigned, cw);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- phase B (as the forward kernel, column image from global memory) ----
        {
            constexpr int NJ = NCOL / 16;
            f32x4 pc[NJ];
#pragma unroll
            for (int J = 0; J < NJ; ++J) pc[J] = (f32x4){0, 0, 0, 0};
            auto bload = [&](dpgp_h8 (&bq)[NJ], int ks) __attribute__((always_inline)) {
#pragma unroll
                for (int J = 0; J < NJ; ++J) {
                    const int m = (J < PS / 16) ? (m_base + 16 * J + li) : (mp_base + 16 * (J - PS / 16) + li);
                    bq[J] = *reinterpret_cast<const dpgp_h8 *>(bimg_g + (size_t)m * SL + 32 * ks + 8 * kk);
                }
            };
            auto bmma = [&](const dpgp_h8 (&bq)[NJ], int ks) __attribute__((always_inline)) {
                const dpgp_h8 av = *reinterpret_cast<const dpgp_h8 *>(aimg + li * SL + 32 * ks + 8 * kk);
#pragma unroll
                for (int J = 0; J < NJ; ++J) pc[J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bq[J], pc[J], 0, 0, 0);
            };
            dpgp_h8 b0[NJ], b1[NJ];
            bload(b0, 0);
            for (int ks = 0; ks < kf1; ks += 2) {
                if (ks + 1 < kf1) bload(b1, ks + 1);
                bmma(b0, ks);
                if (ks + 2 < kf1) bload(b0, ks + 2);
                if (ks + 1 < kf1) bmma(b1, ks + 1);
            }
#pragma unroll
            for (int J = 0; J < NJ; ++J)
#pragma unroll
                for (int v = 0; v < 4; v += 2) {
                    unsigned w0, w1;
                    oor |= !(fabsf(pc[J][v]) <= 30000.0f) | !(fabsf(pc[J][v + 1]) <= 30000.0f);
                    split_pair_words(fminf(fmaxf(pc[J][v], -30000.0f), 30000.0f),
                                     fminf(fmaxf(pc[J][v + 1], -30000.0f), 30000.0f), w0, w1);
                    pw[(4 * kk + v) * PLD + 16 * J + li] = w0;
                    pw[(4 * kk + v + 1) * PLD + 16 * J + li] = w1;
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- phase C': per row, w = G' exp2(E) tile by tile, T'^T += Zt w on the matrix pipe; then the finishing step ----
        // (a hand-made software pipeline over the tiles as in the forward kernel was slower here: 8.0 vs 7.1 ms at config 3)
#pragma unroll 1
        for (int r = 0; r < nr; ++r) {
            dpgp_f2 xk[KB];
            unsigned spA[2], spB[2];
#pragma unroll
            for (int ks = 0; ks < KB; ++ks) xk[ks] = *reinterpret_cast<const dpgp_f2 *>(xaq + r * XLD + 4 * ks);
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                spA[I] = pwA[r * stepA + 32 * I];
                spB[I] = pwB[r * stepB + 32 * I];
            }
            f32x16 acc[2];
#pragma unroll
            for (int v = 0; v < 16; ++v) { acc[0][v] = 0.0f; acc[1][v] = 0.0f; }
            int goff = lane * 4;                                 // (opaque per row: keeps the 64 tile values out of registers)
            asm volatile("" : "+v"(goff));
            const float *glane = gl + goff;
#pragma unroll
            for (int I = 0; I < 2; ++I) {
                dpgp_u4 aop[KB];
#pragma unroll
                for (int ks = 0; ks < KB; ++ks) {
                    unsigned hi, lo;
                    split_products(xk[ks], zA[I][ks], hi, lo);
                    aop[ks] = (dpgp_u4){hi, hi, lo, ks == 0 ? spA[I] : 0u};
                }
#pragma unroll
                for (int J = 0; J < 2; ++J) {
                    f32x4 gv[4];
#pragma unroll
                    for (int vq = 0; vq < 4; ++vq) gv[vq] = *reinterpret_cast<const f32x4 *>(glane + ((2 * I + J) * 4 + vq) * 256);
                    f32x16 c;
#pragma unroll
                    for (int v = 0; v < 16; ++v) c[v] = 0.0f;
#pragma unroll
                    for (int ks = 0; ks < KB; ++ks) {
                        const dpgp_u4 bop = {bh[J][ks], bl[J][ks], bh[J][ks], ks == 0 ? spB[J] : 0u};
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(dpgp_h8, aop[ks]),
                                                                   __builtin_bit_cast(dpgp_h8, bop), c, 0, 0, 0);
                    }
#pragma unroll
                    for (int s_ = 0; s_ < 2; ++s_) {
                        dpgp_u4 whi, wlo;
#pragma unroll
                        for (int j = 0; j < 8; j += 2) {
                            const float w0 = gv[2 * s_ + (j >> 2)][j & 3] * dpgp_exp2(c[8 * s_ + j]);
                            const float w1 = gv[2 * s_ + (j >> 2)][(j & 3) + 1] * dpgp_exp2(c[8 * s_ + j + 1]);
                            unsigned h;
                            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(w0), "v"(w1));
                            whi[j >> 1] = h;
                            if (PSI2G_W_LO) {
                                unsigned l;
                                float l0, l1;
                                asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(w0), "v"(h));
                                asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(w1), "v"(h));
                                asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(l0), "v"(l1));
                                wlo[j >> 1] = l;
                            }
                        }
#ifdef PSI2G_DIAG_SKIP_MFMA2          // (timing experiment only: wrong results)
                        acc[J][s_] += __builtin_bit_cast(float, whi[0] ^ whi[1] ^ whi[2] ^ whi[3] ^ wlo[0] ^ wlo[1] ^ wlo[2] ^ wlo[3]);
#else
                        acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(za[I][s_], __builtin_bit_cast(dpgp_h8, whi), acc[J], 0, 0, 0);
                        if (SPLITZ)
                            acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zal[SPLITZ ? I : 0][s_], __builtin_bit_cast(dpgp_h8, whi), acc[J], 0, 0, 0);
                        if (PSI2G_W_LO)
                            acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(za[I][s_], __builtin_bit_cast(dpgp_h8, wlo), acc[J], 0, 0, 0);
#endif
                    }
                }
            }
            // finishing: this lane = column l % 32 of both column tiles, latent dims q = qb + i
#ifndef PSI2G_DIAG_SKIP_FINISH      // (timing experiments only: wrong results)
            // Per (n, q) everything is linear in the column sums, with coefficients that depend on (n, q) only: each lane forms
            // its columns' share of d/dmu_nq, d/dS_nq (summed over the 32 lanes of the half: 2 QH DPP chains per row, results
            // staged in the consumed P row of the observation and written out after the 16 rows) and of d/dgamma_q (accumulated
            // per lane, summed over the lanes once at the very end).   a2 = gamma / den2, id2 = 1 / den2:
            //   d/dmu = -2 a2 (mu' S0 - S1),  d/dS = -a2 S0 + 2 a2^2 q2,  q2 = mu'^2 S0 - 2 mu' S1 + (S2 + S3) / 2,
            //   d/dgamma = -S id2 S0 - id2^2 q2 - (S2 - S3) / 2     (S0 = sum C, S1 = sum C z', S2 = sum C z'^2, S3 = sum z' T')
            const float cs[2] = {acc[0][ONES], acc[1][ONES]}, c01 = cs[0] + cs[1];
            float xs[QHE], ys[QHE];
            xs[QHE - 1] = 0.0f;
            ys[QHE - 1] = 0.0f;
#pragma unroll
            for (int i = 0; i < QH; ++i) {
                const float gg = gqr[i], a2 = xa[r * XLD + qb + i] * (float)(-2.0 / DPGP_LOG2E), mq = mus[r * XLD + qb + i];
                const float sv = sss[r * XLD + qb + i];
                const float cT = gg - a2, c2n = gg + a2, tt = 2.0f * a2 * mq;
                float a1 = 0.0f, a2s = 0.0f, a3 = 0.0f;
#pragma unroll
                for (int J = 0; J < 2; ++J) {
                    const float tq = SPLITZ ? acc[J][i] : acc[J][i] + acc[J][QH + i], zq = zcr[J][i];
                    dz[J][i] += cT * tq + (tt - c2n * zq) * cs[J];
                    const float cz = cs[J] * zq;
                    a1 += cz;
                    a2s += cz * zq;
                    a3 += zq * tq;
                }
                const float A2 = 0.5f * (a2s + a3), Dh = 0.5f * (a2s - a3);
                const float q2 = mq * (mq * c01 - 2.0f * a1) + A2;
                xs[i] = 2.0f * a2 * (a1 - mq * c01);
                ys[i] = a2 * (2.0f * a2 * q2 - c01);
                const float id2 = a2 * igq[i];
                dgam[i] -= sv * id2 * c01 + id2 * id2 * q2 + Dh;
            }
#pragma unroll
            for (int i = 0; i < QHE; i += 2) half_sum_dpp4(xs[i], ys[i], xs[i + 1], ys[i + 1]);
            if (li5 == 31) {
#pragma unroll
                for (int i = 0; i < QH; ++i) {
                    pw[r * PLD + qb + i] = __builtin_bit_cast(unsigned, xs[i]);
                    pw[r * PLD + KQ + qb + i] = __builtin_bit_cast(unsigned, ys[i]);
                }
            }
#else
            dz[0][0] += acc[0][0] + acc[1][1];
#endif
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int e = lane; e < NR * 2 * KQ; e += 64) {
            const int r = e / (2 * KQ), c = e - r * (2 * KQ), q = c < KQ ? c : c - KQ, n = nc + 4 * r;
            if (n < nend && q < Q)
                (c < KQ ? dmu_part : ds_part)[(slot * N + n) * Q + q] = unscale * __builtin_bit_cast(float, pw[r * PLD + c]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // ---- this workgroup's partial d/dz (64 columns x Q) and d/dgamma (Q): 4 waves through LDS ----
    float *red = zs + L.off_wave;                               // [4][64][KQ + 1]
    __syncthreads();
#pragma unroll
    for (int J = 0; J < 2; ++J)
#pragma unroll
        for (int i = 0; i < QH; ++i) red[(wv * 64 + 32 * J + li5) * (KQ + 1) + qb + i] = dz[J][i];
    __syncthreads();
    for (int e = t; e < 64 * Q; e += 256) {
        const int col = e / Q, q = e - col * Q;
        float v = 0.0f;
        for (int w_ = 0; w_ < 4; ++w_) v += red[(w_ * 64 + col) * (KQ + 1) + q];
        dz_part[(((size_t)(pi * n_splits + sp) * B + b) * (nps * 64) + 64 * pj + col) * Q + q] = (double)(unscale * v);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < QH; ++i) dgam[i] = half_sum_dpp(dgam[i]);
    if (li5 == 31)
#pragma unroll
        for (int i = 0; i < QH; ++i) red[wv * KQ + qb + i] = dgam[i];
    __syncthreads();
    const bool any_oor = __syncthreads_or(oor ? 1 : 0) != 0;
    if (t < Q) dg_part[((size_t)(patch * n_splits + sp) * B + b) * Q + t] =
        any_oor ? (double)__builtin_nanf("") : (double)(unscale * (red[t] + red[KQ + t] + red[2 * KQ + t] + red[3 * KQ + t]));
}

bool psi2_grad_supported(int M, int Q) { return Q <= DPGP_MAX_Q && M >= 1; }
int psi2_grad_nsplit(int B, int N, int M) {
    const int nps = dpgp_ceil_div(dpgp_round_up(M, 16), 64);
    int ns = dpgp_ceil_div(1024, B * nps * nps);
    if (ns > dpgp_ceil_div(N, 256)) ns = dpgp_ceil_div(N, 256);
    return ns < 1 ? 1 : ns;
}
// in doubles: d/dmu, d/dS partials [B npatch][N][Q] (float each); d/dz [nps ns B][64 nps][Q]; d/dgamma [npatch ns][B][Q]
size_t psi2_grad_part_elems(int B, int N, int M, int Q) {
    const int nps = dpgp_ceil_div(dpgp_round_up(M, 16), 64), np = nps * nps, ns = psi2_grad_nsplit(B, N, M);
    return (size_t)B * np * N * Q + 2 + (size_t)B * np * ns * 64 * Q + (size_t)np * ns * B * Q;
}

template <int KB, int W_LO>
static int launch_psi2_grad_kb(int B, int N, int M, int Q, const unsigned char *consts, const double *mu, const double *s,
                               const double *gamma, const double *alpha, const double *GP, double *part, double *stage,
                               double *dmu, double *ds, double *dz, double *dgamma, hipStream_t st) {
    const int Mp = dpgp_round_up(M, 16), nps = dpgp_ceil_div(Mp, 64), np = nps * nps, ns = psi2_grad_nsplit(B, N, M);
    const int nper = dpgp_round_up(dpgp_ceil_div(N, ns), 4);
    const size_t slab = (size_t)B * np * N * Q;
    float *dmu_part = reinterpret_cast<float *>(part), *ds_part = dmu_part + slab;
    double *dz_part = part + slab + 2;                          // (2 float slabs = `slab` doubles, + alignment slack)
    double *dg_part = dz_part + (size_t)B * np * ns * 64 * Q;
    const size_t lds = sizeof(float) * (size_t)psi2g_layout<KB>(Q).elems;
    auto kern = psi2_grad_kernel<KB, W_LO>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
        return DPGP_ERR_LAUNCH;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, dim3((unsigned)((size_t)B * ns * np)), dim3(256), lds, st, N, M, Q, B, consts, mu, s, gamma, alpha, GP, Mp,
                       nper, ns, dmu_part, ds_part, dz_part, dg_part);
    DPGP_LAUNCH_CHECK();
    // the Psi1 / K_uu parts are already in the outputs: add the Psi2 part on top
    const size_t nq = (size_t)N * Q, dq = (size_t)B * Q;
    int rc = launch_reduce_rows<float>(nq, nq, B * np, dmu_part, dmu, 1, stage, st);
    if (rc == DPGP_OK) rc = launch_reduce_rows<float>(nq, nq, B * np, ds_part, ds, 1, stage, st);
    if (rc == DPGP_OK) rc = launch_reduce_rows<double>(dq, dq, np * ns, dg_part, dgamma, 1, stage, st);
    if (rc == DPGP_OK) rc = launch_reduce_rows<double>((size_t)M * Q, (size_t)nps * 64 * Q, nps * ns * B, dz_part, dz, 1, stage, st);
    return rc;
}
// Psi2 part of stage B on the matrix pipe, ADDED to dmu [N,Q], ds [N,Q], dz [M,Q], dgamma [B,Q] (which hold the other parts)
int launch_psi2_grad(int B, int N, int M, int Q, const unsigned char *consts, const double *mu, const double *s,
                     const double *gamma, const double *alpha, const double *GP, double *part, double *stage, double *dmu,
                     double *ds, double *dz, double *dgamma, hipStream_t st) {
    if (!psi2_grad_supported(M, Q)) return -4;
    switch (dpgp_ceil_div(Q, 4)) {
#define CASE(k) case k: return launch_psi2_grad_kb<k, 1>(B, N, M, Q, consts, mu, s, gamma, alpha, GP, part, stage, dmu, ds, dz, dgamma, st);
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
    return -4;
}

// slabs -> dense symmetric [B,M,M]
template <typename T>
__global__ void psi2_finish_kernel(int B, int M, int Mp, int ns, const T *__restrict__ part, T *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * M * M) return;
    int b = (int)(i / ((size_t)M * M));
    int r = (int)(i - (size_t)b * M * M);
    int m = r / M, mp = r - m * M;
    const int a = m > mp ? m : mp, c = m > mp ? mp : m;   // element-wise lower triangle: exactly symmetric output
    double acc = 0.0;
    for (int k = 0; k < ns; ++k) acc += (double)part[((size_t)k * B + b) * (size_t)Mp * Mp + (size_t)a * Mp + c];
    out[i] = (T)acc;
}

template <typename T> struct Psi2Cfg;
template <> struct Psi2Cfg<float> { static constexpr int PT = 4; };
template <> struct Psi2Cfg<double> { static constexpr int PT = 2; };

template <typename T, int KS> static size_t psi2_lds_bytes() {
    return sizeof(T) * (size_t)Psi2Lds<T, KS, Psi2Cfg<T>::PT>::ELEMS;
}

int psi2_nsplit(int B, int N, int M) {
    // Split the observations so that the workgroups fill whole "rounds" of the GPU: R = 256 CUs x 2 resident workgroups.
    // Jobs: W = B * patches * ns patch workgroups of L = ceil(N / ns) rows (rounded up to the 64 rows the 4 waves take per
    // step) and, when the K_uu branch rides in the same dispatch (LDS-resident sizes, M <= 128), B chain tasks of ~60 us =
    // CK row-equivalents each -- before the patches for B < 256, after them otherwise (psi2_task_1d).  The makespan of that
    // list schedule, in rows, is minimised over ns <= 8 (>= 128 rows per split).  Config 2 (B = 64): 8 splits are 1536 + 64
    // workgroups = 3 rounds and a 4th one for the chain tasks (4 x 256 rows); 5 splits are 1024 = 2 rounds of 448 rows.
    if (const char *e = getenv("DPGP_PSI2_NS")) {                // (experiments only)
        const int v = atoi(e);
        if (v >= 1 && v <= 8 && N / v >= 1) return v;
    }
    const int np64 = dpgp_ceil_div(M, 64), patches = np64 * (np64 + 1) / 2, R = 512, CK = 250;
    const long long nchain = (M <= 128) ? B : 0;
    int max_ns = N / 128;
    if (max_ns > 8) max_ns = 8;
    if (max_ns < 1) max_ns = 1;
    if (nchain > 0) {
        // The pair-tile kernel — the default for these sizes — runs B ns nr workgroups (nr tile ranges, pairs_geom).  Measured
        // on config 3 sliced to B output dims (scratch/time_by_d.py, ms per evaluation, (ns, nr)): the best split has ONE round
        // of workgroups, 512 of them — B = 512: (1,1) 1.376 against (2,1) 1.400; B = 256: (2,1) 0.729, (1,2) 0.731 against
        // (4,1) 0.755 and 0.766 for the list-schedule model below (which describes the patch kernels); B = 128: (4,1) 0.432
        // against (3,1) 0.476 — except that few output dims leave the B slots their K_uu tasks hold from the start free:
        // B = 64: (7,1) 0.182 against (8,1) 0.202.  Hence: as many workgroups as fit the target, among equals the most n-splits.
        const long long target = B < 128 ? R - nchain : R;
        if (B > target) {
            // more output dims than slots: several rounds of workgroups; n-splits shorten the partly filled last round
            // (config 5, B = 560: 1 split 0.797 ms (1.09 rounds), 4 splits 0.660, 8 splits 0.665)
            int best_ns = 1;
            double best_score = -1.0;
            for (int ns = 1; ns <= max_ns; ++ns) {
                const long long wg = (long long)B * ns, rounds = (wg + R - 1) / R;
                const double score = (double)wg / (double)(rounds * R) - 0.01 * ns;
                if (score > best_score) { best_score = score; best_ns = ns; }
            }
            return best_ns;
        }
        int best_ns = 1;
        long long best_wg = 0;
        for (int ns = 1; ns <= max_ns; ++ns) {
            const long long per = (long long)B * ns;
            long long nr = target / per;
            if (nr < 1) nr = 1;
            const long long wg = per * nr;
            if (wg > target && ns > 1) break;
            if (wg >= best_wg) { best_wg = wg; best_ns = ns; }
        }
        return best_ns;
    }
    int best = 1;
    double best_t = 1e300;
    for (int ns = 1; ns <= max_ns; ++ns) {
        const long long W = (long long)B * patches * ns;
        long long L = dpgp_round_up(dpgp_ceil_div(N, ns), 4) + 8;   // (+ the per-chunk phases A and B, ~8 row-equivalents per split)
        if (L > N + 8) L = N + 8;
        double t;
        if (nchain == 0) {
            t = (double)((W + R - 1) / R) * (double)L;
        } else if (B >= 256) {                                   // patches first, chain tasks fill up behind them
            const long long f = W / R, rem = W % R, free_slots = R - rem;
            const double tail = (double)f * L + (double)((nchain + free_slots - 1) / free_slots) * CK;
            const double body = (double)(rem ? f + 1 : f) * L;
            t = tail > body ? tail : body;
        } else {                                                 // chain tasks first on B slots, patches on the earliest free slot
            const long long busy = nchain < R ? nchain : R;
            t = 1e300;
            for (long long k = 1; k <= W; ++k) {                 // candidate makespans k L and CK + k L
                const double t1 = (double)k * L;
                const long long k_busy1 = t1 >= CK ? (long long)((t1 - CK) / L) : 0;
                if ((R - busy) * k + busy * k_busy1 >= W && t1 < t) t = t1;
                const double t2 = (double)CK + (double)k * L;
                if ((R - busy) * (long long)(t2 / L) + busy * k >= W && t2 < t) t = t2;
                if (t < 1e299 && t1 > t) break;
            }
        }
        if (t < best_t * 0.999) { best_t = t; best = ns; }
    }
    return best;
}

template <typename TIN, typename T, int KS>
static int launch_psi2_ks(int B, int N, int M, int Q, const TIN *z, const TIN *mu, const TIN *s, const TIN *gamma,
                          const TIN *alpha, T *part, int ns, const ChainKTask &task, hipStream_t st) {
    constexpr int PT = Psi2Cfg<T>::PT, PS = 16 * PT;
    const int Mp = dpgp_round_up(M, 16);
    const int nps = dpgp_ceil_div(Mp, PS);
    const int nper = dpgp_round_up(dpgp_ceil_div(N, ns), PSI2_NT);
    dim3 grid(B, ns, nps * (nps + 1) / 2 + (task.ws ? 1 : 0));
    size_t lds = psi2_lds_bytes<T, KS>();
    if (task.ws && chain_k_lds_bytes(task.Mp, task.elem) > lds) lds = chain_k_lds_bytes(task.Mp, task.elem);
    auto kern = psi2_mfma_kernel<TIN, T, KS, PT>;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return DPGP_ERR_LAUNCH;
    }
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, N, M, Q, B, z, mu, s, gamma, alpha, part, Mp, nper, task);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

template <typename TIN>
__global__ __launch_bounds__(256) void psi2_consts_kernel(const TIN *__restrict__ z, int M, int Q, unsigned char *__restrict__ dst) {
    __shared__ double scratch[5 * 64];
    const int nrow = (M + 63) / 64;
    if ((int)blockIdx.x < nrow) psi2_consts_rows(z, M, Q, dst, (int)blockIdx.x, scratch);
    else psi2_pair_rows(z, M, Q, dst, (int)blockIdx.x - nrow, scratch);
}
size_t psi2_consts_bytes(int M, int Q) { return psi2_consts_layout(M, Q).bytes; }
template <typename TIN> int launch_psi2_consts(const TIN *z, int M, int Q, unsigned char *consts, hipStream_t st) {
    const int blocks = dpgp_ceil_div(M, 64) + dpgp_ceil_div(psi2_consts_layout(M, Q).Ppad, PSI2_PAIR_ROWS_PER_BLOCK);
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((psi2_consts_kernel<TIN>), dim3(blocks), dim3(256), 0, st, z, M, Q, consts);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
template int launch_psi2_consts<float>(const float *, int, int, unsigned char *, hipStream_t);
template int launch_psi2_consts<double>(const double *, int, int, unsigned char *, hipStream_t);

template <typename TIN, int KB>
static int launch_psi2_f16_kb(int B, int N, int M, int Q, const TIN *z, const TIN *mu, const TIN *s, const TIN *gamma,
                              const TIN *alpha, float *part, int ns, const ChainKTask &task, unsigned char *consts,
                              int consts_ready, hipStream_t st) {
    const int Mp = dpgp_round_up(M, 16);
    const int nps = dpgp_ceil_div(Mp, 64);
    if (!consts) return -18;
    if (task.ws && !chain_k_resident(task.Mp, task.elem)) return -16;    // (callers fuse only the LDS-resident K_uu branch)
    if (!consts_ready) {
        const int rc = launch_psi2_consts<TIN>(z, M, Q, consts, st);
        if (rc) return rc;
    }
    const int nper = dpgp_round_up(dpgp_ceil_div(N, ns), 4);       // even splits; a wave skips the rows of its last 16-row chunk that do not exist
    const long long nwg = (long long)B * ns * (nps * (nps + 1) / 2) + (task.ws ? B : 0);   // see psi2_task_1d
    if (nwg > 0x7fffffffLL) return -1;
    dim3 grid((unsigned)nwg);
#ifdef PSI2_P_VALU
    size_t lds = sizeof(float) * (size_t)Psi2F16Lds<KB>::ELEMS;
#else
    size_t lds = sizeof(float) * (size_t)psi2p_layout<KB>(Q).elems;
#endif
    if (task.ws && chain_k_lds_bytes(task.Mp, task.elem) > lds) lds = chain_k_lds_bytes(task.Mp, task.elem);
    auto kern = psi2_f16_kernel<TIN, KB>;
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return DPGP_ERR_LAUNCH;
    }
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, N, M, Q, B, z, (const unsigned char *)consts, mu, s, gamma, alpha,
                       part, Mp, nper, ns, task);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

// psi2_pairs.hip: the pair-tile kernel (fp32 results; the default)
template <typename TIN>
int launch_psi2_pairs(int B, int N, int M, int Q, const TIN *z, const TIN *mu, const TIN *s, const TIN *gamma,
                      const TIN *alpha, float *part, int ns, const ChainKTask &task, const unsigned char *consts, float *scale,
                      int scale_ready, hipStream_t st);
template <typename TIN, typename T> struct Psi2PairsDispatch {
    static int run(int, int, int, int, const TIN *, const TIN *, const TIN *, const TIN *, const TIN *, T *, int,
                   const ChainKTask &, unsigned char *, int, float *, hipStream_t) {
        return -13;
    }
};
template <typename TIN> struct Psi2PairsDispatch<TIN, float> {
    static int run(int B, int N, int M, int Q, const TIN *z, const TIN *mu, const TIN *s, const TIN *gamma,
                   const TIN *alpha, float *part, int ns, const ChainKTask &task, unsigned char *consts, int consts_ready,
                   float *pair_scale, hipStream_t st) {
        if (!consts || !pair_scale) return -18;
        if (!consts_ready) {
            const int rc = launch_psi2_consts<TIN>(z, M, Q, consts, st);
            if (rc) return rc;
        }
        // consts_ready > 1: the caller's front launch has also filled the pair-scale table
        return launch_psi2_pairs<TIN>(B, N, M, Q, z, mu, s, gamma, alpha, part, ns, task, consts, pair_scale, consts_ready > 1, st);
    }
};

// the f16-split kernel exists for fp32 results only
template <typename TIN, typename T> struct Psi2F16Dispatch {
    static int run(int, int, int, int, const TIN *, const TIN *, const TIN *, const TIN *, const TIN *, T *, int,
                   const ChainKTask &, unsigned char *, int, hipStream_t) {
        return -13;
    }
};
template <typename TIN> struct Psi2F16Dispatch<TIN, float> {
    static int run(int B, int N, int M, int Q, const TIN *z, const TIN *mu, const TIN *s, const TIN *gamma,
                   const TIN *alpha, float *part, int ns, const ChainKTask &task, unsigned char *consts, int consts_ready,
                   hipStream_t st) {
        switch (dpgp_ceil_div(Q, 4)) {
#define CASE(k) \
    case k: return launch_psi2_f16_kb<TIN, k>(B, N, M, Q, z, mu, s, gamma, alpha, part, ns, task, consts, consts_ready, st);
            CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
        }
        return -4;
    }
};

template <typename TIN, typename T>
int launch_psi2_partial(int B, int N, int M, int Q, const TIN *z, const TIN *mu, const TIN *s, const TIN *gamma,
                        const TIN *alpha, T *part, int ns, int algo, hipStream_t st, void *chain_ws, int chain_elem,
                        double *logdet_k, int *info_k, unsigned char *consts, int consts_ready, float *pair_scale) {
    const int Mp = dpgp_round_up(M, 16);
    int chain_last = B >= 256 ? 1 : 0;
    if (const char *e = getenv("DPGP_CHAIN_LAST")) chain_last = atoi(e) ? 1 : 0;      // (experiments only)
    ChainKTask task = {chain_ws, la_chain_ws_elems_inline(M), logdet_k, info_k, M, Mp, chain_elem, chain_last};
    if (algo == DPGP_ALGO_PLAIN && chain_ws) return -16;     // the plain path launches chain_k on its own
    if (algo == DPGP_ALGO_PLAIN) {
        // slabs 1.. are expected to exist by the consumer: zero them, slab 0 carries the result
        if (ns > 1 && hipMemsetAsync(part + (size_t)B * Mp * Mp, 0, sizeof(T) * (size_t)(ns - 1) * B * Mp * Mp, st) !=
                          hipSuccess)
            return DPGP_ERR_LAUNCH;
        dim3 grid(dpgp_ceil_div(Mp * Mp, 256), B);
        DPGP_PRELAUNCH(); hipLaunchKernelGGL((psi2_plain_kernel<TIN, T>), grid, dim3(256), 0, st, N, M, Q, z, mu, s, gamma, alpha, part,
                           Mp);
        DPGP_LAUNCH_CHECK();
        return DPGP_OK;
    }
    // fp32 results: f16-split operands on the matrix pipe unless the exact-fp32 MFMA kernel is asked for; the pair-tile
    // kernel (psi2_pairs.hip) by default, the per-observation patch kernel on request
    // (more than 20 latent dims: 12 K-steps per pair tile leave no room for resident column operands; patch kernel)
    if (sizeof(T) == 4 && (algo == DPGP_ALGO_PATCH_F16 || (algo != DPGP_ALGO_MFMA_F32 && psi2_pairs_ksteps(Q) > 8)))
        return Psi2F16Dispatch<TIN, T>::run(B, N, M, Q, z, mu, s, gamma, alpha, part, ns, task, consts, consts_ready, st);
    if (sizeof(T) == 4 && algo != DPGP_ALGO_MFMA_F32)
        return Psi2PairsDispatch<TIN, T>::run(B, N, M, Q, z, mu, s, gamma, alpha, part, ns, task, consts, consts_ready, pair_scale,
                                              st);
    const int KS = dpgp_ceil_div(Q + 2, 4);
    switch (KS) {
#define CASE(k) \
    case k: return launch_psi2_ks<TIN, T, k>(B, N, M, Q, z, mu, s, gamma, alpha, part, ns, task, st);
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)
#undef CASE
    }
    return -4;
}
template int launch_psi2_partial<float, float>(int, int, int, int, const float *, const float *, const float *,
                                               const float *, const float *, float *, int, int, hipStream_t, void *, int,
                                               double *, int *, unsigned char *, int, float *);
template int launch_psi2_partial<double, double>(int, int, int, int, const double *, const double *, const double *,
                                                 const double *, const double *, double *, int, int, hipStream_t, void *,
                                                 int, double *, int *, unsigned char *, int, float *);
template int launch_psi2_partial<double, float>(int, int, int, int, const double *, const double *, const double *,
                                                const double *, const double *, float *, int, int, hipStream_t, void *,
                                                int, double *, int *, unsigned char *, int, float *);

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
extern "C" size_t dpgp_psi2_workspace_bytes(int B, int N, int M, int Q, int elem_size) {
    if (B <= 0 || N <= 0 || M <= 0 || Q <= 0) return 0;
    int Mp = dpgp_round_up(M, 16);
    return dpgp_align256((size_t)elem_size * psi2_nsplit(B, N, M) * B * Mp * Mp) + psi2_consts_bytes(M, Q) +
           psi2_pairs_scale_bytes(B, M);
}

template <typename T>
static int psi2_api(int B, int N, int M, int Q, const T *z, const T *mu, const T *s, const T *gamma, const T *alpha,
                    T *out, void *ws, size_t ws_bytes, int algo, void *stream) {
    if (B <= 0) return -1;
    if (N <= 0) return -2;
    if (M <= 0) return -3;
    if (Q <= 0 || Q > DPGP_MAX_Q) return -4;
    if (!z) return -5;
    if (!mu) return -6;
    if (!s) return -7;
    if (!gamma) return -8;
    if (!alpha) return -9;
    if (!out) return -10;
    if (!ws) return -11;
    if (ws_bytes < dpgp_psi2_workspace_bytes(B, N, M, Q, sizeof(T))) return -12;
    if (algo < 0 || algo > DPGP_ALGO_PATCH_F16) return -13;
    const int ns = psi2_nsplit(B, N, M), Mp = dpgp_round_up(M, 16);
    unsigned char *consts = (unsigned char *)ws + dpgp_align256(sizeof(T) * (size_t)ns * B * Mp * Mp);
    int rc = launch_psi2_partial<T, T>(B, N, M, Q, z, mu, s, gamma, alpha, (T *)ws, ns, algo, (hipStream_t)stream,
                                       nullptr, 0, nullptr, nullptr, consts, 0,
                                       reinterpret_cast<float *>(consts + psi2_consts_bytes(M, Q)));
    if (rc) return rc;
    size_t tot = (size_t)B * M * M;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((psi2_finish_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       B, M, Mp, ns, (const T *)ws, out);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
extern "C" int dpgp_psi2_f32(int B, int N, int M, int Q, const float *z, const float *mu, const float *s,
                             const float *gamma, const float *alpha, float *out, void *ws, size_t ws_bytes, int algo,
                             void *stream) {
    return psi2_api<float>(B, N, M, Q, z, mu, s, gamma, alpha, out, ws, ws_bytes, algo, stream);
}
extern "C" int dpgp_psi2_f64(int B, int N, int M, int Q, const double *z, const double *mu, const double *s,
                             const double *gamma, const double *alpha, double *out, void *ws, size_t ws_bytes, int algo,
                             void *stream) {
    return psi2_api<double>(B, N, M, Q, z, mu, s, gamma, alpha, out, ws, ws_bytes, algo, stream);
}
