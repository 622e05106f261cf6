// ARD-RBF gram / diag / psi0 / psi1 / Psi1^T y / KL kernels (HBM- or exp-bound, no matrix cores needed).
// Reference semantics: /root/reference/src/kernels/rbf_kernel.py:58-161, src/models/expressions/gp_expressions.py:10-24.
#include <cstdlib>
#include "internal.h"
#include "psi2_consts.h"

// ---------------------------------------------------------------------------------------------------------------
// K1  gram: out[b, i, j] = alpha_b exp(-1/2 sum_q gamma_bq (x0_iq - x1_jq)^2)  (+ noise/jitter on the diagonal)
//   one 256-thread workgroup per 64x64 output tile; both input tiles are staged through LDS pre-scaled by
//   sqrt(gamma_b) (so the inner loop is sub+fma per q); each thread owns a 4x4 patch whose rows are written as one
//   16-byte (fp32) / 32-byte (fp64) store -> 256 B contiguous per 16 lanes.
// ---------------------------------------------------------------------------------------------------------------
#define GRAM_T 64
template <typename TIN, typename T>
__device__ __forceinline__ void gram_tile(int b, int i0, int j0, int N0, int N1, int Q, const TIN *__restrict__ x0,
                                          const TIN *__restrict__ x1, const TIN *__restrict__ gamma,
                                          const TIN *__restrict__ alpha, const TIN *__restrict__ beta, int flags, T jitter,
                                          T *__restrict__ out, int ld_out, size_t batch_stride, int symmetric,
                                          unsigned char *smem_raw) {
    // both input tiles in LDS, pre-scaled by sqrt(gamma_b), TRANSPOSED: [Q][64], so that a thread's four rows / columns of
    // one latent dim are ONE 16-byte (fp32) read instead of four 4-byte ones (round 2: [64][Q + 1], eight ds_read_b32 per 32
    // vector instructions — the LDS pipe, not the vector unit or HBM, held the 1 GB gram at 4.1 TB/s)
    typedef T vec4 __attribute__((ext_vector_type(4)));
    T *xs = reinterpret_cast<T *>(smem_raw);          // [Q][64]
    T *zs = xs + GRAM_T * Q;                           // [Q][64]
    // fp64: the table of dpgp_exp2_tab behind the tiles (64 doubles; callers size the LDS for it) — 12 instead of ~24 fp64
    // operations per entry next to the 2 Q of the squared distance
    const double *etab = nullptr;
    if constexpr (sizeof(T) == 8) {
        double *et = reinterpret_cast<double *>(zs + GRAM_T * Q);
        dpgp_exp2_tab_init(et);
        etab = et;
    }
    const int t = threadIdx.x;
    const TIN *g = gamma + (size_t)b * Q;
    {   // fill: thread = (point r = t & 63, latent dims q = t >> 6, + 4, ...): no division, one square root per (thread, q)
        const int r = t & 63;
        const bool in0 = i0 + r < N0, in1 = j0 + r < N1;
        for (int q = t >> 6; q < Q; q += 4) {
            const T sg = sqrt((T)g[q]);
            xs[q * GRAM_T + r] = in0 ? sg * (T)x0[(size_t)(i0 + r) * Q + q] : (T)0;
            zs[q * GRAM_T + r] = in1 ? sg * (T)x1[(size_t)(j0 + r) * Q + q] : (T)0;
        }
    }
    __syncthreads();
    const int ty = t >> 4, tx = t & 15;
    T acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0;
    for (int q = 0; q < Q; ++q) {
        const vec4 a = *reinterpret_cast<const vec4 *>(xs + q * GRAM_T + ty * 4);
        const vec4 c_ = *reinterpret_cast<const vec4 *>(zs + q * GRAM_T + tx * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                T d = a[r] - c_[c];
                acc[r][c] = fma(d, d, acc[r][c]);
            }
    }
    const T al = (T)alpha[b];
    T diag_add = 0;
    if (symmetric & 1) {
        if (flags & DPGP_FLAG_NOISE) diag_add += (T)1 / (T)beta[b];
        if (flags & DPGP_FLAG_JITTER) diag_add += jitter;
    }
    const T scale = (T)(-0.5 * DPGP_LOG2E);
    T *ob = out + (size_t)b * batch_stride;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int i = i0 + ty * 4 + r;
        if (i >= N0) continue;
        T v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int j = j0 + tx * 4 + c;
            v[c] = al * dpgp_exp2_hot(scale * acc[r][c], etab);
            if ((symmetric & 1) && i == j) v[c] += diag_add;
        }
        int j = j0 + tx * 4;
        T *p = ob + (size_t)i * ld_out + j;
        if (j + 3 < N1 && ((ld_out & 3) == 0)) {
            vec4 vv = {v[0], v[1], v[2], v[3]};
            if (symmetric & 2) __builtin_nontemporal_store(vv, reinterpret_cast<vec4 *>(p));   // (bit 1: the output is not re-read soon)
            else *reinterpret_cast<vec4 *>(p) = vv;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (j + c < N1) p[c] = v[c];
        }
    }
}
template <typename TIN, typename T>
__global__ __launch_bounds__(256) void gram_kernel(int N0, int N1, int Q, const TIN *__restrict__ x0,
                                                   const TIN *__restrict__ x1, const TIN *__restrict__ gamma,
                                                   const TIN *__restrict__ alpha, const TIN *__restrict__ beta,
                                                   int flags, T jitter, T *__restrict__ out, int ld_out,
                                                   size_t batch_stride, int symmetric) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    gram_tile<TIN, T>(blockIdx.z, blockIdx.y * GRAM_T, blockIdx.x * GRAM_T, N0, N1, Q, x0, x1, gamma, alpha, beta, flags, jitter,
                      out, ld_out, batch_stride, symmetric, smem_raw);
}

template <typename TIN, typename T>
int launch_gram(int B, int N0, int N1, int Q, const TIN *x0, const TIN *x1, const TIN *gamma, const TIN *alpha,
                const TIN *beta, int flags, double jitter, T *out, int ld_out, size_t batch_stride, hipStream_t st) {
    int sym = (x1 == nullptr) ? 1 : 0;
    if (sym) { x1 = x0; N1 = N0; }
    // outputs beyond the L2 + MALL working set are streamed with non-temporal stores (DPGP_GRAM_NT=0 / 1 overrides: profiling)
    const char *nt_ = getenv("DPGP_GRAM_NT");
    if (nt_ ? nt_[0] == '1' : (size_t)B * N0 * N1 * sizeof(T) > ((size_t)256 << 20)) sym |= 2;
    dim3 grid(dpgp_ceil_div(N1, GRAM_T), dpgp_ceil_div(N0, GRAM_T), B);
    size_t lds = sizeof(T) * 2 * GRAM_T * Q + (sizeof(T) == 8 ? sizeof(double) * DPGP_EXP2_TAB_ELEMS : 0);
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((gram_kernel<TIN, T>), grid, dim3(256), lds, st, N0, N1, Q, x0, x1, gamma, alpha, beta, flags,
                       (T)jitter, out, ld_out, batch_stride, sym);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
template int launch_gram<float, float>(int, int, int, int, const float *, const float *, const float *, const float *,
                                       const float *, int, double, float *, int, size_t, hipStream_t);
template int launch_gram<double, double>(int, int, int, int, const double *, const double *, const double *,
                                         const double *, const double *, int, double, double *, int, size_t,
                                         hipStream_t);
template int launch_gram<double, float>(int, int, int, int, const double *, const double *, const double *,
                                        const double *, const double *, int, double, float *, int, size_t,
                                        hipStream_t);

// ---------------------------------------------------------------------------------------------------------------
// diag / psi0: trivial broadcasts (rbf_kernel.py:96-132)
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void diag_kernel(int B, int N, const T *alpha, const T *beta, int flags, T jitter, T *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * N) return;
    int b = (int)(i / N);
    T v = alpha[b];
    if (flags & DPGP_FLAG_NOISE) v += (T)1 / beta[b];
    if (flags & DPGP_FLAG_JITTER) v += jitter;
    out[i] = v;
}
template <typename T> __global__ void psi0_kernel(int B, int N, const T *alpha, T *out) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out[b] = alpha[b] * (T)N;
}

// ---------------------------------------------------------------------------------------------------------------
// K2  psi1 materialised: out[b,n,m] = alpha_b exp(-1/2 sum_q( g (mu_nq - z_mq)^2 / (g s_nq + 1) + log(g s_nq + 1) ))
//   workgroup = (b, 32 rows n); per-row factors w = g/(g s+1) and c = -1/2 sum log(g s+1) are computed once into LDS;
//   z is staged through LDS in chunks of 64 rows; thread = (row, 4 consecutive m) -> 16-byte stores.
// ---------------------------------------------------------------------------------------------------------------
#define PSI1_NT 32
template <typename T>
__global__ __launch_bounds__(256) void psi1_kernel(int N, int M, int Q, const T *__restrict__ z,
                                                   const T *__restrict__ mu, const T *__restrict__ s,
                                                   const T *__restrict__ gamma, const T *__restrict__ alpha,
                                                   T *__restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *w = reinterpret_cast<T *>(smem_raw);   // [32][Q]
    T *mus = w + PSI1_NT * Q;                  // [32][Q]
    T *cn = mus + PSI1_NT * Q;                 // [32]
    T *zs = cn + PSI1_NT;                      // [64][Q+1]
    const int b = blockIdx.y, n0 = blockIdx.x * PSI1_NT, t = threadIdx.x;
    const T *g = gamma + (size_t)b * Q;
    for (int e = t; e < PSI1_NT * Q; e += 256) {
        int r = e / Q, q = e - r * Q, n = n0 + r;
        T sv = n < N ? s[(size_t)n * Q + q] : (T)1, mv = n < N ? mu[(size_t)n * Q + q] : (T)0;
        T den = g[q] * sv + (T)1;
        w[e] = g[q] / den;
        mus[e] = mv;
    }
    if (t < PSI1_NT) {
        int n = n0 + t;
        T a = 0;
        for (int q = 0; q < Q; ++q) a += dpgp_log(g[q] * (n < N ? s[(size_t)n * Q + q] : (T)1) + (T)1);
        cn[t] = (T)(-0.5 * DPGP_LOG2E) * a;
    }
    const T al = alpha[b];
    const int r = t >> 3, c8 = t & 7;   // 32 rows x 8 threads; each thread 2 groups of 4 consecutive m per 64-chunk
    const int n = n0 + r;
    for (int m0 = 0; m0 < M; m0 += 64) {
        __syncthreads();
        for (int e = t; e < 64 * Q; e += 256) {
            int rr = e / Q, q = e - rr * Q;
            zs[rr * (Q + 1) + q] = (m0 + rr < M) ? z[(size_t)(m0 + rr) * Q + q] : (T)0;
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int mb = (h * 8 + c8) * 4;
            T acc[4] = {0, 0, 0, 0};
            for (int q = 0; q < Q; ++q) {
                T wq = w[r * Q + q], mq = mus[r * Q + q];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    T d = mq - zs[(mb + c) * (Q + 1) + q];
                    acc[c] = fma(wq * d, d, acc[c]);
                }
            }
            if (n < N) {
                T *p = out + ((size_t)b * N + n) * M + m0 + mb;
                T v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = al * dpgp_exp2((T)(-0.5 * DPGP_LOG2E) * acc[c] + cn[r]);
                if (m0 + mb + 3 < M && ((M & 3) == 0)) {
                    typedef T vec4 __attribute__((ext_vector_type(4)));
                    vec4 vv = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<vec4 *>(p) = vv;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (m0 + mb + c < M) p[c] = v[c];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K2f  Psi1^T y (fused reduction over n, Psi1 never written):  part[ns][b][m] = alpha_b sum_{n in chunk} y_nb psi1[b,n,m]
//   grid (ns, ceil(M/128), B), 256 threads = 4 waves; a wave takes every 4th n of the tile, a lane owns m = lane and
//   m = lane + 64 of the 128-chunk with their z rows in registers; per-n factors go through LDS.
// ---------------------------------------------------------------------------------------------------------------
#define P1Y_NT 32
template <typename TIN, typename T>
__global__ __launch_bounds__(256) void psi1T_y_kernel(int N, int M, int Q, int B, const TIN *__restrict__ z,
                                                      const TIN *__restrict__ mu, const TIN *__restrict__ s,
                                                      const TIN *__restrict__ gamma, const TIN *__restrict__ alpha,
                                                      const TIN *__restrict__ y, int ldy, double *__restrict__ part,
                                                      int n_per_split) {
    __shared__ T w[P1Y_NT][DPGP_MAX_Q + 2];
    __shared__ T mus[P1Y_NT][DPGP_MAX_Q + 2];
    __shared__ T lg[P1Y_NT][DPGP_MAX_Q + 2];
    __shared__ T cn[P1Y_NT], yn[P1Y_NT];
    __shared__ double red[4][128];
    __shared__ double etab[DPGP_EXP2_TAB_ELEMS];             // fp64: table-based exp2 (dpgp_exp2_tab, common.h); unused in fp32
    if (sizeof(T) == 8) dpgp_exp2_tab_init(etab);
    const int b = blockIdx.z, mc = blockIdx.y * 128, sp = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const TIN *g = gamma + (size_t)b * Q;
    T zr[2][DPGP_MAX_Q];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int m = mc + lane + 64 * h;
#pragma unroll
        for (int q = 0; q < DPGP_MAX_Q; ++q) zr[h][q] = (q < Q && m < M) ? (T)z[(size_t)m * Q + q] : (T)0;
    }
    double acc[2] = {0.0, 0.0};
    const int nbeg = sp * n_per_split, nend = min(N, nbeg + n_per_split);
    for (int n0 = nbeg; n0 < nend; n0 += P1Y_NT) {
        __syncthreads();
        for (int e = t; e < P1Y_NT * Q; e += 256) {
            int r = e / Q, q = e - r * Q, n = n0 + r;
            bool ok = n < nend;
            T sv = ok ? (T)s[(size_t)n * Q + q] : (T)1;
            T den = (T)g[q] * sv + (T)1;
            w[r][q] = (T)(-0.5 * DPGP_LOG2E) * (T)g[q] / den;
            mus[r][q] = ok ? (T)mu[(size_t)n * Q + q] : (T)0;
            lg[r][q] = dpgp_log(den);                         // (one logarithm per thread: P1Y_NT threads taking Q each in a row
        }                                                     //  kept the other 200 waiting at the barrier — fp64: 0.72 ms of 10.4)
        __syncthreads();
        if (t < P1Y_NT) {
            int n = n0 + t;
            bool ok = n < nend;
            T a = 0;
            for (int q = 0; q < Q; ++q) a += lg[t][q];
            cn[t] = (T)(-0.5 * DPGP_LOG2E) * a;
            yn[t] = ok ? (T)y[(size_t)n * ldy + b] : (T)0;
        }
        __syncthreads();
        T tacc[2] = {0, 0};
        for (int r = wv; r < P1Y_NT; r += 4) {
            T e0 = cn[r], e1 = cn[r];
#pragma unroll
            for (int q = 0; q < DPGP_MAX_Q; ++q) {
                if (q < Q) {
                    T wq = w[r][q], mq = mus[r][q];
                    T d0 = mq - zr[0][q], d1 = mq - zr[1][q];
                    e0 = fma(wq * d0, d0, e0);
                    e1 = fma(wq * d1, d1, e1);
                }
            }
            tacc[0] = fma(yn[r], dpgp_exp2_hot(e0, etab), tacc[0]);
            tacc[1] = fma(yn[r], dpgp_exp2_hot(e1, etab), tacc[1]);
        }
        acc[0] += (double)tacc[0];
        acc[1] += (double)tacc[1];
    }
    __syncthreads();
    red[wv][lane] = acc[0];
    red[wv][lane + 64] = acc[1];
    __syncthreads();
    if (t < 128 && mc + t < M) {
        double v = red[0][t] + red[1][t] + red[2][t] + red[3][t];
        part[((size_t)sp * B + b) * M + mc + t] = (double)alpha[b] * v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K2f on the matrix pipe (fp32 results): the exponent of psi1 is a rank-(2Q+1) form in (n, m),
//      log2 psi1[n,m] / alpha = c_n + sum_q a_nq z_mq^2 + sum_q b_nq z_mq,     a = -1/2 w1 log2e,  b = w1 (mu-c) log2e,
//      c_n = -1/2 log2e sum_q ( w1 (mu-c)^2 + log(g s + 1) ),   w1 = g / (g s + 1),   z centred by its column mean c,
// evaluated tile by tile as [16 n x K] x [K x 16 m] with v_mfma_f32_16x16x32_f16 on f16 hi/lo-split operands (three
// products per term, see psi2.hip), K = 6Q + 2 slots; then v[m] += y_n exp2(E[n,m]).  Per 256 (n,m) pairs: 2-6 MFMAs,
// 4 v_exp_f32 and 4 FMAs per lane instead of ~40 VALU ops per pair.
//   slot 3t+{0,1,2} of term t (t = 2q: (a, z^2), t = 2q+1: (b, z)):  A = {h, h, l},  B = {h, l, h};   slots 6Q, 6Q+1: (c_h, c_l) x (1, 1)
// grid (ns, ceil(M/128), B); each wave takes every 4th 16-row tile of the split and owns a 2 KB LDS operand image.
// ---------------------------------------------------------------------------------------------------------------
typedef _Float16 p1_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 p1_h2 __attribute__((ext_vector_type(2)));
typedef float p1_f4 __attribute__((ext_vector_type(4)));

// NJ: 16-column tiles a workgroup covers (8 = 128 columns; 4 when M <= 64: half the operand registers, MFMAs and exponentials
// — config 5 (M = 64, Q = 20) spent them on zero padding: 183 us for 7e7 exponentials)
template <typename TIN, int KF1, int NJ>
__global__ __launch_bounds__(256) void psi1T_y_f16_kernel(int N, int M, int Q, int B, const TIN *__restrict__ z,
                                                          const unsigned char *__restrict__ consts,
                                                          const TIN *__restrict__ mu, const TIN *__restrict__ s,
                                                          const TIN *__restrict__ gamma,
                                                          const TIN *__restrict__ alpha, const TIN *__restrict__ y,
                                                          int ldy, double *__restrict__ part, int n_per_split) {
    constexpr int SL = 32 * KF1;                          // K slots (f16) per row of an operand image
    constexpr int QP = 4 * ((32 * KF1 - 2) / 6 / 4 + 1);  // row stride of the per-(row,q) arrays: >= Q, multiple of 4
    constexpr int QPC = QP < 32 ? QP : 32;                // (Q <= DPGP_MAX_Q = 30)
    __shared__ __align__(16) _Float16 bimg[16 * NJ * SL]; // m-side image of this chunk of 16 NJ columns
    __shared__ __align__(16) _Float16 aimg[4][16 * SL];   // n-side image, one per wave
    __shared__ float yv[4][16];
    __shared__ float zc[32], gq[32];
    __shared__ __align__(16) float red[16][128];
    const int b = blockIdx.z, mc = blockIdx.y * (16 * NJ), sp = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, kk = lane >> 4;
    // the z-only constants (column means, m-side image: the same image as the psi2 kernel's, psi2_consts.h) are copied
    const Psi2Consts C = psi2_consts_layout(M, Q);        // C.SL == SL
    const _Float16 *bimg_g = reinterpret_cast<const _Float16 *>(consts + C.off_bimg);
    if (t < 32) {
        gq[t] = (t < Q) ? (float)gamma[(size_t)b * Q + t] : 0.0f;
        zc[t] = reinterpret_cast<const float *>(consts)[t];
    }
    typedef unsigned p1_u4 __attribute__((ext_vector_type(4)));
    for (int e = t; e < 16 * NJ * (SL / 8); e += 256) {       // 16-byte vectors; rows up to round_up(M, 64) exist
        const int m = e / (SL / 8), k = e - m * (SL / 8);
        p1_u4 v = {0u, 0u, 0u, 0u};
        if (mc + m < C.Mp64) v = reinterpret_cast<const p1_u4 *>(bimg_g + (size_t)(mc + m) * SL)[k];
        reinterpret_cast<p1_u4 *>(bimg)[e] = v;
    }
    for (int e = t; e < 4 * 16 * SL / 2; e += 256) reinterpret_cast<unsigned *>(&aimg[0][0])[e] = 0u;
    __syncthreads();
    // m-side operands of this lane: 8 column tiles x KF1 K-steps, resident in registers
    p1_h8 bop[NJ][KF1];
#pragma unroll
    for (int J = 0; J < NJ; ++J)
#pragma unroll
        for (int ks = 0; ks < KF1; ++ks)
            bop[J][ks] = *reinterpret_cast<const p1_h8 *>(bimg + (16 * J + li) * SL + 32 * ks + 8 * kk);
    float acc[NJ];
#pragma unroll
    for (int J = 0; J < NJ; ++J) acc[J] = 0.0f;
    _Float16 *am = &aimg[wv][0];
    const int nbeg = sp * n_per_split, nend = min(N, nbeg + n_per_split);
    // q(X) rows and y of a 16-row tile are fetched one tile ahead (registers), so that their global-load latency overlaps
    // the previous tile's MFMA / exp work; the per-row constant c_n is reduced from per-(row,q) pieces through LDS
    // (element e of a tile = (row e / QPC, latent dim e % QPC), zero padded: vector reads, no division by Q).
    constexpr int NPF = (16 * QPC + 63) / 64;
    float *cq = &red[wv * 4][0];                          // [16][QPC] scratch of this wave (red is only used at the very end)
    TIN pf_s[NPF], pf_m[NPF], pf_y = (TIN)0;
    auto prefetch = [&](int n0) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = 64 * u + lane, r = e / QPC, q = e - r * QPC, n = n0 + r;
            const bool ok = (e < 16 * QPC) && (q < Q) && (n < nend);
            pf_s[u] = ok ? s[(size_t)n * Q + q] : (TIN)1;
            pf_m[u] = ok ? mu[(size_t)n * Q + q] : (TIN)0;
        }
        pf_y = (lane < 16 && n0 + lane < nend) ? y[(size_t)(n0 + lane) * ldy + b] : (TIN)0;
    };
    prefetch(nbeg + 16 * wv);
    bool oor = false;
    for (int n0 = nbeg + 16 * wv; n0 < nend; n0 += 64) {
        // ---- n-side image of rows n0 .. n0+15 ----
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int e = 64 * u + lane;
            if (e < 16 * QPC) {
                const int r = e / QPC, q = e - r * QPC, n = n0 + r;
                float a = 0.0f, bb = 0.0f, cc = 0.0f;
                if (q < Q) {
                    if (n < nend) {
                        // (v_rcp_f32 / v_log_f32, 1 ulp each, instead of the IEEE division and the library logf: ~35 of the ~60
                        //  instructions of an element — this image build, not the exponentials, sets the kernel's pace: per
                        //  observation Q elements against M exponentials)
                        const float g = gq[q], den = g * (float)pf_s[u] + 1.0f;
                        const float w1 = g * __builtin_amdgcn_rcpf(den), mcq = (float)pf_m[u] - zc[q];
                        a = (float)(-0.5 * DPGP_LOG2E) * w1;
                        bb = (float)DPGP_LOG2E * w1 * mcq;
                        cc = w1 * mcq * mcq + 0.6931471805599453f * __builtin_amdgcn_logf(den);
                    }
                    a = dpgp_pin(a);                          // (pinned before the (hi, lo) split: see dpgp_pin)
                    bb = dpgp_pin(bb);
                    const _Float16 ah = (_Float16)a, al = (_Float16)(a - (float)ah);
                    const _Float16 bh = (_Float16)bb, bl = (_Float16)(bb - (float)bh);
                    unsigned *dst = reinterpret_cast<unsigned *>(am + r * SL + 6 * q);      // slots {ah, ah, al, bh, bh, bl}
                    const p1_h2 w0 = {ah, ah}, w1_ = {al, bh}, w2 = {bh, bl};
                    dst[0] = __builtin_bit_cast(unsigned, w0);
                    dst[1] = __builtin_bit_cast(unsigned, w1_);
                    dst[2] = __builtin_bit_cast(unsigned, w2);
                }
                cq[e] = cc;
            }
        }
        const float yn = (float)pf_y;
        prefetch(n0 + 64);                                  // next tile of this wave
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 16) {
            float c = 0.0f;
#pragma unroll
            for (int q4 = 0; q4 < QPC / 4; ++q4) {
                const p1_f4 v = *reinterpret_cast<const p1_f4 *>(cq + lane * QPC + 4 * q4);
                c += (v[0] + v[1]) + (v[2] + v[3]);
            }
            c = (float)(-0.5 * DPGP_LOG2E) * c;
            oor |= !(c >= -60000.0f);                       // f16 range guard (as in psi2_patch_f16p): the result becomes NaN
            c = fmaxf(c, -60000.0f);
            c = dpgp_pin(c);
            const _Float16 ch = (_Float16)c;
            const p1_h2 cw = {ch, (_Float16)(c - (float)ch)};
            *reinterpret_cast<unsigned *>(am + lane * SL + 6 * Q) = __builtin_bit_cast(unsigned, cw);
            yv[wv][lane] = yn;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        p1_h8 aop[KF1];
#pragma unroll
        for (int ks = 0; ks < KF1; ++ks) aop[ks] = *reinterpret_cast<const p1_h8 *>(am + li * SL + 32 * ks + 8 * kk);
        float y4[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) y4[v] = yv[wv][4 * kk + v];
        p1_f4 c[NJ];
#pragma unroll
        for (int J = 0; J < NJ; ++J) c[J] = (p1_f4){0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KF1; ++ks)                    // K-step outermost: 8 independent accumulation chains in flight
#pragma unroll
            for (int J = 0; J < NJ; ++J) c[J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(aop[ks], bop[J][ks], c[J], 0, 0, 0);
#pragma unroll
        for (int J = 0; J < NJ; ++J)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[J] = fmaf(y4[v], dpgp_exp2(c[J][v]), acc[J]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the next tile overwrites the image just read
    }
    // ---- sum the 4 row groups of each wave and the 4 waves ----
    const bool any_oor = __syncthreads_or(oor ? 1 : 0) != 0;   // (also the barrier cq -> red needs: cq aliases red)
#pragma unroll
    for (int J = 0; J < NJ; ++J) red[wv * 4 + kk][16 * J + li] = acc[J];
    __syncthreads();
    if (t < 16 * NJ && mc + t < M) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += (double)red[k][t];
        part[((size_t)sp * B + b) * M + mc + t] = any_oor ? (double)__builtin_nanf("") : (double)alpha[b] * v;
    }
}

template <typename TIN, int KF1>
static int launch_psi1T_y_f16_kf(int B, int N, int M, int Q, const TIN *z, const TIN *mu, const TIN *s,
                                 const TIN *gamma, const TIN *alpha, const TIN *y, int ldy, double *part, int ns,
                                 const unsigned char *consts, hipStream_t st) {
    int nper = dpgp_round_up(dpgp_ceil_div(N, ns), 64);
    if (M <= 64) {
        dim3 grid(ns, 1, B);
        DPGP_PRELAUNCH(); hipLaunchKernelGGL((psi1T_y_f16_kernel<TIN, KF1, 4>), grid, dim3(256), 0, st, N, M, Q, B, z, consts, mu, s, gamma,
                           alpha, y, ldy, part, nper);
        DPGP_LAUNCH_CHECK();
        return DPGP_OK;
    }
    dim3 grid(ns, dpgp_ceil_div(M, 128), B);
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((psi1T_y_f16_kernel<TIN, KF1, 8>), grid, dim3(256), 0, st, N, M, Q, B, z, consts, mu, s, gamma, alpha, y,
                       ldy, part, nper);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

int psi1T_y_nsplit(int B, int N, int M) {
    if (const char *e = getenv("DPGP_PSI1_NS")) {                // (experiments only)
        const int v = atoi(e);
        if (v >= 1 && v <= 64 && v <= dpgp_ceil_div(N, 4 * P1Y_NT)) return v;
    }
    int wgs = B * dpgp_ceil_div(M, 128);
    int ns = dpgp_ceil_div(512, wgs);           // aim for ~2 workgroups per CU (measured: config 3 best with 1 split, config 2 with 8)
    int max_ns = dpgp_ceil_div(N, 4 * P1Y_NT);
    if (ns > max_ns) ns = max_ns;
    return ns < 1 ? 1 : ns;
}

template <typename TIN, typename T>
int launch_psi1T_y_partial(int B, int N, int M, int Q, const TIN *z, const TIN *mu, const TIN *s, const TIN *gamma,
                           const TIN *alpha, const TIN *y, int ldy, double *part, int ns, unsigned char *consts,
                           int consts_ready, hipStream_t st) {
    if (sizeof(T) == 4) {        // fp32 results: f16-split operands on the matrix pipe
        if (!consts) return -18;
        if (!consts_ready) {
            int rc = launch_psi2_consts<TIN>(z, M, Q, consts, st);
            if (rc) return rc;
        }
        switch (dpgp_ceil_div(6 * Q + 2, 32)) {
#define CASE(k) \
    case k: return launch_psi1T_y_f16_kf<TIN, k>(B, N, M, Q, z, mu, s, gamma, alpha, y, ldy, part, ns, consts, st);
            CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6)
#undef CASE
        }
        return -4;
    }
    int nper = dpgp_round_up(dpgp_ceil_div(N, ns), P1Y_NT);
    dim3 grid(ns, dpgp_ceil_div(M, 128), B);
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((psi1T_y_kernel<TIN, T>), grid, dim3(256), 0, st, N, M, Q, B, z, mu, s, gamma, alpha, y, ldy,
                       part, nper);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
template int launch_psi1T_y_partial<float, float>(int, int, int, int, const float *, const float *, const float *,
                                                  const float *, const float *, const float *, int, double *, int,
                                                  unsigned char *, int, hipStream_t);
template int launch_psi1T_y_partial<double, double>(int, int, int, int, const double *, const double *, const double *,
                                                    const double *, const double *, const double *, int, double *, int,
                                                    unsigned char *, int, hipStream_t);
template int launch_psi1T_y_partial<double, float>(int, int, int, int, const double *, const double *, const double *,
                                                   const double *, const double *, const double *, int, double *, int,
                                                   unsigned char *, int, hipStream_t);

template <typename T>
__global__ void sum_slabs_kernel(size_t n, int ns, const double *__restrict__ part, T *__restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = 0;
    for (int k = 0; k < ns; ++k) a += part[(size_t)k * n + i];
    out[i] = (T)a;
}

// ---------------------------------------------------------------------------------------------------------------
// KL(q(X) || N(0,I)) (gp_expressions.py:18-23) and, in the same launch, y_d^T y_d for the data-fit term
// (dp_gp_lvm.py:143-144; the reference forms the whole [D,D] product and takes its diagonal).
//   blocks 0..DPGP_KL_NBLK-1 : KL partial sums -> kl_out[DPGP_KL_NBLK] (summed in fixed order by the consumer)
//   other blocks : (64 consecutive d) x (one of YY_NCH chunks of n); coalesced along d, 4 waves stride over n, partial
//                  sums written to yy_out[chunk][d] (DPGP_YY_NCH slabs, summed in fixed order by the consumer: deterministic)
//   blocks >= first_consts_block (optional): 64 rows each of the psi2 kernel's z-only constants (psi2_consts.h)
// ---------------------------------------------------------------------------------------------------------------
#define YY_NCH DPGP_YY_NCH
template <typename TIN>
__device__ __forceinline__ void kl_yy_block(int blk, int N, int Q, const TIN *__restrict__ mu, const TIN *__restrict__ s,
                                            double *__restrict__ kl_out, int D, const TIN *__restrict__ y, int ldy,
                                            double *__restrict__ yy_out, const TIN *__restrict__ z, int M,
                                            unsigned char *__restrict__ consts, int first_consts_block,
                                            double (*scratch)[64]) {
    const int t = threadIdx.x;
    if (blk >= first_consts_block) {      // third role: z-only constants of the psi kernels (psi2_consts.h)
        const int cb = blk - first_consts_block, nrow = (M + 63) / 64;
        if (cb < nrow) psi2_consts_rows(z, M, Q, consts, cb, &scratch[0][0]);
        else psi2_pair_rows(z, M, Q, consts, cb - nrow, &scratch[0][0]);        // pair image of psi2_pairs.hip
        return;
    }
    if (blk < DPGP_KL_NBLK) {      // KL partials: block k handles every DPGP_KL_NBLK-th run of 256 elements
        if (kl_out == nullptr) return;
        double a0 = 0.0;
        const size_t tot = (size_t)N * Q;
        for (size_t i = (size_t)blk * 256 + t; i < tot; i += (size_t)DPGP_KL_NBLK * 256) {
            const double m0 = (double)mu[i], v0 = (double)s[i];
            a0 += m0 * m0 + v0 - log(v0) - 1.0;
        }
        const double a = block_sum(a0, &scratch[0][0]);
        if (t == 0) kl_out[blk] = 0.5 * a;     // sum over blocks = 1/2 (sum mu^2 + sum (s - log s) - N Q)
        // the arrival counter of chain_b_kernel's final reduction (linalg.hip) sits behind the partials: reset per evaluation
        if (t == 0 && blk == 0) *reinterpret_cast<int *>(kl_out + DPGP_KL_NBLK) = 0;
        return;
    }
    const int bid = blk - DPGP_KL_NBLK, dblk = bid / YY_NCH, ch = bid - dblk * YY_NCH;
    const int d = dblk * 64 + (t & 63), wv = t >> 6;
    const int nper = (N + YY_NCH - 1) / YY_NCH, n0 = ch * nper, n1 = min(N, n0 + nper);
    double a0 = 0.0, a1 = 0.0;
    if (d < D) {
        int n = n0 + wv;
        for (; n + 4 < n1; n += 8) {
            const double v0 = (double)y[(size_t)n * ldy + d], v1 = (double)y[(size_t)(n + 4) * ldy + d];
            a0 += v0 * v0;
            a1 += v1 * v1;
        }
        for (; n < n1; n += 4) {
            const double v0 = (double)y[(size_t)n * ldy + d];
            a0 += v0 * v0;
        }
    }
    scratch[wv][t & 63] = a0 + a1;
    __syncthreads();
    if (t < 64 && d < D) yy_out[(size_t)ch * D + d] = scratch[0][t] + scratch[1][t] + scratch[2][t] + scratch[3][t];
}
template <typename TIN>
__global__ __launch_bounds__(256) void kl_yy_kernel(int N, int Q, const TIN *__restrict__ mu,
                                                    const TIN *__restrict__ s, double *__restrict__ kl_out, int D,
                                                    const TIN *__restrict__ y, int ldy, double *__restrict__ yy_out,
                                                    const TIN *__restrict__ z, int M, unsigned char *__restrict__ consts,
                                                    int first_consts_block) {
    __shared__ double scratch[5][64];
    kl_yy_block<TIN>((int)blockIdx.x, N, Q, mu, s, kl_out, D, y, ldy, yy_out, z, M, consts, first_consts_block, scratch);
}
// The front launch of the fused ELBO: the roles of kl_yy_kernel and, behind them, the 64 x 64 tiles of K_uu + jitter I of all
// D output dims (gram_tile) -- none of them depends on another, so one launch replaces two (the evaluation at D / 8 output
// dims per GPU is a chain of latency-bound launches).
template <typename TL>
__global__ __launch_bounds__(256) void elbo_front_kernel(int N, int Q, const double *__restrict__ mu,
                                                         const double *__restrict__ s, double *__restrict__ kl_out, int D,
                                                         const double *__restrict__ y, int ldy, double *__restrict__ yy_out,
                                                         const double *__restrict__ z, int M, unsigned char *__restrict__ consts,
                                                         int first_consts_block, int first_gram_block,
                                                         const double *__restrict__ gamma, const double *__restrict__ alpha,
                                                         const double *__restrict__ beta, TL jitter, TL *__restrict__ kuu,
                                                         int ld_kuu, size_t kuu_stride, int first_scale_block,
                                                         float *__restrict__ pair_scale) {
    __shared__ double scratch[5][64];
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int blk = (int)blockIdx.x;
#ifdef FRONT_DIAG_SKIP             // (timing experiments only: wrong results) bit 0: KL / y'y / constants, 1: K_uu tiles, 2: scale table
    if ((FRONT_DIAG_SKIP & 1) && blk < first_gram_block) return;
    // (bits 1 / 2 are tested where the interleaved blocks are told apart)
#endif
    if (blk < first_gram_block) {
        kl_yy_block<double>(blk, N, Q, mu, s, kl_out, D, y, ldy, yy_out, z, M, consts, first_consts_block, scratch);
        return;
    }
    // K_uu tiles (fp64 vector work) and scale-table blocks (gather, exp, 17 MB of stores at config 3) are INTERLEAVED in dispatch
    // order: one after the other they took 22.9 + 20.9 us of the launch's 44 (scratch/front_roles.sh), neither filling the GPU's
    // other half.  Position i of the ng + ns blocks is a scale block iff floor((i + 1) ns / (ng + ns)) > floor(i ns / (ng + ns)).
    const int ng = first_scale_block - first_gram_block, nsb = (int)gridDim.x - first_scale_block;
    int g = blk - first_gram_block;
    if (nsb > 0) {
        const long long T = (long long)ng + nsb;
        const int c0 = (int)((long long)g * nsb / T), c1 = (int)((long long)(g + 1) * nsb / T);
        if (c1 > c0) {                                    // the pair-scale table of the pair-tile psi2 kernel (psi2_consts.h)
#ifdef FRONT_DIAG_SKIP
            if (FRONT_DIAG_SKIP & 4) return;
#endif
            const int nb = (psi2_consts_layout(M, Q).Ppad + 255) / 256, sb = c0;
            const int dch = psi2_scale_dchunk(D), b0 = (sb / nb) * dch;
            psi2_pair_scale_chunk<double, double>(b0, min(dch, D - b0), sb % nb, M, Q, z, gamma, alpha, pair_scale,
                                                  reinterpret_cast<float *>(smem_raw));   // (16 (Q + 1) floats of the launch's LDS)
            return;
        }
        g -= c0;
    }
#ifdef FRONT_DIAG_SKIP
    if (FRONT_DIAG_SKIP & 2) return;
#endif
    const int tm = (M + GRAM_T - 1) / GRAM_T, b = g / (tm * tm), r = g - b * tm * tm;
    gram_tile<double, TL>(b, (r / tm) * GRAM_T, (r % tm) * GRAM_T, M, M, Q, z, z, gamma, alpha, beta, DPGP_FLAG_JITTER, jitter,
                          kuu, ld_kuu, kuu_stride, 1, smem_raw);
}
template <typename TL>
int launch_elbo_front(int N, int Q, const double *mu, const double *s, double *kl_out, int D, const double *y, int ldy,
                      double *yy_out, const double *z, int M, unsigned char *psi2_consts, const double *gamma,
                      const double *alpha, const double *beta, double jitter, TL *kuu, int ld_kuu, size_t kuu_stride,
                      float *pair_scale, hipStream_t st) {
    int blocks = DPGP_KL_NBLK + dpgp_ceil_div(D, 64) * YY_NCH;
    const int first_consts = blocks;
    blocks += dpgp_ceil_div(M, 64) + dpgp_ceil_div(psi2_consts_layout(M, Q).Ppad, PSI2_PAIR_ROWS_PER_BLOCK);
    const int first_gram = blocks, tm = dpgp_ceil_div(M, GRAM_T);
    blocks += D * tm * tm;
    const int first_scale = blocks;                     // pair_scale != nullptr: ceil(D / chunk) x ceil(Ppad / 256) more blocks
    if (pair_scale) blocks += dpgp_ceil_div(D, psi2_scale_dchunk(D)) * dpgp_ceil_div(psi2_consts_layout(M, Q).Ppad, 256);
    const size_t lds = sizeof(TL) * 2 * GRAM_T * (Q + 1);
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((elbo_front_kernel<TL>), dim3(blocks), dim3(256), lds, st, N, Q, mu, s, kl_out, D, y, ldy, yy_out,
                       z, M, psi2_consts, first_consts, first_gram, gamma, alpha, beta, (TL)jitter, kuu, ld_kuu, kuu_stride,
                       first_scale, pair_scale);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
template int launch_elbo_front<float>(int, int, const double *, const double *, double *, int, const double *, int, double *,
                                      const double *, int, unsigned char *, const double *, const double *, const double *,
                                      double, float *, int, size_t, float *, hipStream_t);
template int launch_elbo_front<double>(int, int, const double *, const double *, double *, int, const double *, int, double *,
                                       const double *, int, unsigned char *, const double *, const double *, const double *,
                                       double, double *, int, size_t, float *, hipStream_t);

template <typename TIN>
int launch_kl_yy(int N, int Q, const TIN *mu, const TIN *s, double *kl_out, int D, const TIN *y, int ldy,
                 double *yy_out, const TIN *z, int M, unsigned char *psi2_consts, hipStream_t st) {
    int blocks = DPGP_KL_NBLK;
    if (yy_out) {
        blocks += dpgp_ceil_div(D, 64) * YY_NCH;
    }
    const int first_consts = blocks;
    if (psi2_consts) blocks += dpgp_ceil_div(M, 64) + dpgp_ceil_div(psi2_consts_layout(M, Q).Ppad, PSI2_PAIR_ROWS_PER_BLOCK);
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((kl_yy_kernel<TIN>), dim3(blocks), dim3(256), 0, st, N, Q, mu, s, kl_out, D, y, ldy, yy_out, z, M,
                       psi2_consts, first_consts);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
template int launch_kl_yy<float>(int, int, const float *, const float *, double *, int, const float *, int, double *,
                                 const float *, int, unsigned char *, hipStream_t);
template int launch_kl_yy<double>(int, int, const double *, const double *, double *, int, const double *, int,
                                  double *, const double *, int, unsigned char *, hipStream_t);

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
#define CHECK_ARG(cond, idx) \
    do {                     \
        if (!(cond)) return -(idx); \
    } while (0)

template <typename T>
static int gram_api(int B, int N0, int N1, int Q, const T *x0, const T *x1, const T *gamma, const T *alpha,
                    const T *beta, int flags, double jitter, T *out, void *stream) {
    CHECK_ARG(B > 0, 1); CHECK_ARG(N0 > 0, 2); CHECK_ARG(x1 == nullptr || N1 > 0, 3);
    CHECK_ARG(Q > 0 && Q <= DPGP_MAX_Q, 4); CHECK_ARG(x0, 5); CHECK_ARG(gamma, 7); CHECK_ARG(alpha, 8);
    CHECK_ARG(beta || !(flags & DPGP_FLAG_NOISE), 9); CHECK_ARG(out, 12);
    int n1 = x1 ? N1 : N0;
    return launch_gram<T, T>(B, N0, n1, Q, x0, x1, gamma, alpha, beta, flags, jitter, out, n1, (size_t)N0 * n1,
                             (hipStream_t)stream);
}
extern "C" int dpgp_ard_rbf_gram_f32(int B, int N0, int N1, int Q, const float *x0, const float *x1, const float *gamma,
                                     const float *alpha, const float *beta, int flags, double jitter, float *out,
                                     void *stream) {
    return gram_api<float>(B, N0, N1, Q, x0, x1, gamma, alpha, beta, flags, jitter, out, stream);
}
extern "C" int dpgp_ard_rbf_gram_f64(int B, int N0, int N1, int Q, const double *x0, const double *x1,
                                     const double *gamma, const double *alpha, const double *beta, int flags,
                                     double jitter, double *out, void *stream) {
    return gram_api<double>(B, N0, N1, Q, x0, x1, gamma, alpha, beta, flags, jitter, out, stream);
}

template <typename T>
static int diag_api(int B, int N, const T *alpha, const T *beta, int flags, double jitter, T *out, void *stream) {
    CHECK_ARG(B > 0, 1); CHECK_ARG(N > 0, 2); CHECK_ARG(alpha, 3); CHECK_ARG(beta || !(flags & DPGP_FLAG_NOISE), 4);
    CHECK_ARG(out, 7);
    size_t tot = (size_t)B * N;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((diag_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, N,
                       alpha, beta, flags, (T)jitter, out);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
extern "C" int dpgp_ard_rbf_diag_f32(int B, int N, const float *alpha, const float *beta, int flags, double jitter,
                                     float *out, void *stream) {
    return diag_api<float>(B, N, alpha, beta, flags, jitter, out, stream);
}
extern "C" int dpgp_ard_rbf_diag_f64(int B, int N, const double *alpha, const double *beta, int flags, double jitter,
                                     double *out, void *stream) {
    return diag_api<double>(B, N, alpha, beta, flags, jitter, out, stream);
}

template <typename T> static int psi0_api(int B, int N, const T *alpha, T *out, void *stream) {
    CHECK_ARG(B > 0, 1); CHECK_ARG(N > 0, 2); CHECK_ARG(alpha, 3); CHECK_ARG(out, 4);
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((psi0_kernel<T>), dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, N, alpha, out);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
extern "C" int dpgp_psi0_f32(int B, int N, const float *alpha, float *out, void *stream) {
    return psi0_api<float>(B, N, alpha, out, stream);
}
extern "C" int dpgp_psi0_f64(int B, int N, const double *alpha, double *out, void *stream) {
    return psi0_api<double>(B, N, alpha, out, stream);
}

template <typename T>
static int psi1_api(int B, int N, int M, int Q, const T *z, const T *mu, const T *s, const T *gamma, const T *alpha,
                    T *out, void *stream) {
    CHECK_ARG(B > 0, 1); CHECK_ARG(N > 0, 2); CHECK_ARG(M > 0, 3); CHECK_ARG(Q > 0 && Q <= DPGP_MAX_Q, 4);
    CHECK_ARG(z, 5); CHECK_ARG(mu, 6); CHECK_ARG(s, 7); CHECK_ARG(gamma, 8); CHECK_ARG(alpha, 9); CHECK_ARG(out, 10);
    size_t lds = sizeof(T) * (2 * PSI1_NT * Q + PSI1_NT + 64 * (Q + 1));
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((psi1_kernel<T>), dim3(dpgp_ceil_div(N, PSI1_NT), B), dim3(256), lds, (hipStream_t)stream, N, M,
                       Q, z, mu, s, gamma, alpha, out);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
extern "C" int dpgp_psi1_f32(int B, int N, int M, int Q, const float *z, const float *mu, const float *s,
                             const float *gamma, const float *alpha, float *out, void *stream) {
    return psi1_api<float>(B, N, M, Q, z, mu, s, gamma, alpha, out, stream);
}
extern "C" int dpgp_psi1_f64(int B, int N, int M, int Q, const double *z, const double *mu, const double *s,
                             const double *gamma, const double *alpha, double *out, void *stream) {
    return psi1_api<double>(B, N, M, Q, z, mu, s, gamma, alpha, out, stream);
}

extern "C" size_t dpgp_psi1T_y_workspace_bytes(int B, int N, int M) {
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    // partial slabs + the z-only operand constants of the f16 kernel (sized for the largest Q: no Q in this signature)
    return dpgp_align256(sizeof(double) * (size_t)psi1T_y_nsplit(B, N, M) * B * M) + psi2_consts_bytes(M, DPGP_MAX_Q);
}
template <typename T>
static int psi1T_y_api(int B, int N, int M, int Q, const T *z, const T *mu, const T *s, const T *gamma, const T *alpha,
                       const T *y, int ldy, T *out, void *ws, size_t ws_bytes, void *stream) {
    CHECK_ARG(B > 0, 1); CHECK_ARG(N > 0, 2); CHECK_ARG(M > 0, 3); CHECK_ARG(Q > 0 && Q <= DPGP_MAX_Q, 4);
    CHECK_ARG(z, 5); CHECK_ARG(mu, 6); CHECK_ARG(s, 7); CHECK_ARG(gamma, 8); CHECK_ARG(alpha, 9); CHECK_ARG(y, 10);
    CHECK_ARG(ldy >= B, 11); CHECK_ARG(out, 12); CHECK_ARG(ws, 13);
    CHECK_ARG(ws_bytes >= dpgp_psi1T_y_workspace_bytes(B, N, M), 14);
    int ns = psi1T_y_nsplit(B, N, M);
    unsigned char *consts = (unsigned char *)ws + dpgp_align256(sizeof(double) * (size_t)ns * B * M);
    int rc = launch_psi1T_y_partial<T, T>(B, N, M, Q, z, mu, s, gamma, alpha, y, ldy, (double *)ws, ns, consts, 0,
                                          (hipStream_t)stream);
    if (rc) return rc;
    size_t tot = (size_t)B * M;
    DPGP_PRELAUNCH(); hipLaunchKernelGGL((sum_slabs_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       tot, ns, (const double *)ws, out);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
extern "C" int dpgp_psi1T_y_f32(int B, int N, int M, int Q, const float *z, const float *mu, const float *s,
                                const float *gamma, const float *alpha, const float *y, int ldy, float *out, void *ws,
                                size_t ws_bytes, void *stream) {
    return psi1T_y_api<float>(B, N, M, Q, z, mu, s, gamma, alpha, y, ldy, out, ws, ws_bytes, stream);
}
extern "C" int dpgp_psi1T_y_f64(int B, int N, int M, int Q, const double *z, const double *mu, const double *s,
                                const double *gamma, const double *alpha, const double *y, int ldy, double *out,
                                void *ws, size_t ws_bytes, void *stream) {
    return psi1T_y_api<double>(B, N, M, Q, z, mu, s, gamma, alpha, y, ldy, out, ws, ws_bytes, stream);
}

__global__ void kl_single_block_kernel(int N, int Q, const float *muf, const float *sf, const double *mud,
                                       const double *sd, double *out) {
    __shared__ double scratch[8];
    double a = 0.0;
    const size_t tot = (size_t)N * Q;
    for (size_t i = threadIdx.x; i < tot; i += 256) {
        const double m0 = muf ? (double)muf[i] : mud[i], v0 = sf ? (double)sf[i] : sd[i];
        a += m0 * m0 + v0 - log(v0) - 1.0;
    }
    a = block_sum(a, scratch);
    if (threadIdx.x == 0) out[0] = 0.5 * a;
}
static int kl_api_launch(int N, int Q, const float *muf, const float *sf, const double *mud, const double *sd,
                         double *out, void *stream) {
    CHECK_ARG(N > 0, 1); CHECK_ARG(Q > 0, 2); CHECK_ARG(muf || mud, 3); CHECK_ARG(sf || sd, 4); CHECK_ARG(out, 5);
    DPGP_PRELAUNCH(); hipLaunchKernelGGL(kl_single_block_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, N, Q, muf, sf, mud, sd, out);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
extern "C" int dpgp_kl_qx_f32(int N, int Q, const float *mu, const float *s, double *out, void *stream) {
    return kl_api_launch(N, Q, mu, s, nullptr, nullptr, out, stream);
}
extern "C" int dpgp_kl_qx_f64(int N, int Q, const double *mu, const double *s, double *out, void *stream) {
    return kl_api_launch(N, Q, nullptr, nullptr, mu, s, out, stream);
}
extern "C" int dpgp_version(void) { return 200; }
