// Strided batched fp64 matrix product on the matrix cores: C[b] = alpha A[b] B[b] + beta C[b] with element strides for every
// index, so that a transposed, sliced or broadcast operand is a choice of strides and no copy.  It carries the plain products
// of the composed models — the places where the reference calls tf.matmul on [D,M,M] / [T,M,N] x [N,D] operands
// (/root/reference/src/models/dp_gp_lvm.py:657-658 Psi1^T Y of the over-T model; :640-676 its chain; the prediction bound
// gaussian_process.py / dp_gp_lvm.py:300-420) and the adjoint algebra of the backward pass for M > 128 (B^-1 = L^-T L^-1,
// K^-1 Psi2 K^-1): sizes of a few hundred, batches of 8 .. 512.
//
// Workgroup = one 64 x 64 tile of C for one batch element, 4 waves x (16 rows x 64 columns = 4 v_mfma_f64_16x16x4 tiles);
// K is staged 32 at a time through LDS as As[i][k], Bs[j][k] (k contiguous, row stride 36 doubles: conflict-free operand
// reads); the staging loop runs along whichever index is contiguous in memory.  Bounded by the fp64 matrix pipe only for
// large K; at the sizes above a launch is a few microseconds of latency.
#include "internal.h"
#include "linalg_dev.h"

#define CHECK_ARG(cond, idx) \
    do {                     \
        if (!(cond)) return -(idx); \
    } while (0)

#define GM_T 64
#define GM_KC 32
#define GM_LD (GM_KC + 4)

// one operand chunk [GM_T][GM_KC] in two halves: global -> registers (8 per thread, issued before the products of the
// previous chunk so that their latency hides beneath the MFMAs), registers -> LDS after the barrier
struct GmChunk {
    double v[GM_T * GM_KC / 256];
    __device__ __forceinline__ static void index(int e, bool kfast, int &i, int &k) {
        if (kfast) { i = e / GM_KC; k = e % GM_KC; } else { k = e / GM_T; i = e % GM_T; }
    }
    __device__ __forceinline__ void load(const double *__restrict__ g, long long si, long long sk, int i0, int imax, int k0,
                                         int kmax, bool kfast) {
#pragma unroll
        for (int u = 0; u < GM_T * GM_KC / 256; ++u) {
            int i, k;
            index((int)threadIdx.x + 256 * u, kfast, i, k);
            v[u] = (i0 + i < imax && k0 + k < kmax) ? g[(long long)(i0 + i) * si + (long long)(k0 + k) * sk] : 0.0;
        }
    }
    __device__ __forceinline__ void store(double *__restrict__ s, bool kfast) const {
#pragma unroll
        for (int u = 0; u < GM_T * GM_KC / 256; ++u) {
            int i, k;
            index((int)threadIdx.x + 256 * u, kfast, i, k);
            s[i * GM_LD + k] = v[u];
        }
    }
};

__global__ __launch_bounds__(256, 2) void gemm_strided_f64_kernel(int batch, int m, int n, int k, double alpha,
                                                                  const double *__restrict__ a, long long a_sb, long long a_si,
                                                                  long long a_sk, const double *__restrict__ b, long long b_sb,
                                                                  long long b_sk, long long b_sj, double beta,
                                                                  double *__restrict__ c, long long c_sb, long long c_si,
                                                                  long long c_sj, int ksplit, long long c_ss) {
    __shared__ __align__(16) double As[GM_T * GM_LD], Bs[GM_T * GM_LD];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, li = lane & 15, kk = lane >> 4;
    const int i0 = GM_T * blockIdx.y, j0 = GM_T * blockIdx.x;
    const bool a_kfast = (a_sk == 1), b_kfast = (b_sk == 1);
    // ksplit > 1 (internal callers only): the k range in ksplit pieces, piece ks of batch element bi into c + ks c_ss (beta = 0)
    const int kchunk = ksplit > 1 ? GM_KC * ((k + ksplit * GM_KC - 1) / (ksplit * GM_KC)) : k;
    for (int bz = blockIdx.z; bz < batch * ksplit; bz += gridDim.z) {
        const int bi = bz / ksplit, ks_ = bz - bi * ksplit;
        const int kbeg = ks_ * kchunk, kend = min(k, kbeg + kchunk);
        const double *ab = a + (long long)bi * a_sb, *bb = b + (long long)bi * b_sb;
        double *cb = c + (long long)bi * c_sb + (long long)ks_ * c_ss;
        f64x4 acc[4];
#pragma unroll
        for (int J = 0; J < 4; ++J) acc[J] = (f64x4){0.0, 0.0, 0.0, 0.0};
        GmChunk ca, cb_;
        if (kbeg < kend) {
            ca.load(ab, a_si, a_sk, i0, m, kbeg, kend, a_kfast);
            cb_.load(bb, b_sj, b_sk, j0, n, kbeg, kend, b_kfast);
        }
        for (int k0 = kbeg; k0 < kend; k0 += GM_KC) {
            __syncthreads();
            ca.store(As, a_kfast);
            cb_.store(Bs, b_kfast);
            __syncthreads();
            if (k0 + GM_KC < kend) {
                ca.load(ab, a_si, a_sk, i0, m, k0 + GM_KC, kend, a_kfast);
                cb_.load(bb, b_sj, b_sk, j0, n, k0 + GM_KC, kend, b_kfast);
            }
#pragma unroll
            for (int ks = 0; ks < GM_KC / 4; ++ks) {
                const double av = As[(16 * wv + li) * GM_LD + 4 * ks + kk];
#pragma unroll
                for (int J = 0; J < 4; ++J) acc[J] = Mfma<double>::mma(av, Bs[(16 * J + li) * GM_LD + 4 * ks + kk], acc[J]);
            }
        }
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = i0 + 16 * wv + Mfma<double>::row(lane, v), col = j0 + 16 * J + li;
                if (r < m && col < n) {
                    double *o = cb + (long long)r * c_si + (long long)col * c_sj;
                    *o = (beta != 0.0) ? alpha * acc[J][v] + beta * *o : alpha * acc[J][v];
                }
            }
    }
}

extern "C" int dpgp_gemm_strided_f64(int batch, int m, int n, int k, double alpha, const double *a, long long a_sb,
                                     long long a_si, long long a_sk, const double *b, long long b_sb, long long b_sk,
                                     long long b_sj, double beta, double *c, long long c_sb, long long c_si, long long c_sj,
                                     void *stream) {
    CHECK_ARG(batch >= 1, 1);
    CHECK_ARG(m >= 1, 2);
    CHECK_ARG(n >= 1, 3);
    CHECK_ARG(k >= 0, 4);
    CHECK_ARG(a != nullptr || k == 0, 6);
    CHECK_ARG(b != nullptr || k == 0, 10);
    CHECK_ARG(c != nullptr, 15);
    const dim3 grid(dpgp_ceil_div(n, GM_T), dpgp_ceil_div(m, GM_T), batch < 65535 ? batch : 65535);
    CHECK_ARG(grid.y <= 65535u, 2);
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(gemm_strided_f64_kernel, grid, dim3(256), 0, (hipStream_t)stream, batch, m, n, k, alpha, a, a_sb, a_si,
                       a_sk, b, b_sb, b_sk, b_sj, beta, c, c_sb, c_si, c_sj, 1, 0LL);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}

// internal: the k range in ksplit pieces, each into its own slab c + ks c_ss (beta = 0; the consumer adds the slabs) — a product
// with few output tiles and a long k (Psi1^T Y of the over-T model: 8 x 128 x 512 outputs, k = N = 2000) fills the GPU this way
int launch_gemm_splitk_f64(int batch, int m, int n, int k, const double *a, long long a_sb, long long a_si, long long a_sk,
                           const double *b, long long b_sb, long long b_sk, long long b_sj, double *c, long long c_sb,
                           long long c_si, long long c_sj, int ksplit, long long c_ss, hipStream_t st) {
    if (batch < 1 || m < 1 || n < 1 || k < 1 || ksplit < 1 || (long long)batch * ksplit > 65535) return -1;
    const dim3 grid(dpgp_ceil_div(n, GM_T), dpgp_ceil_div(m, GM_T), batch * ksplit);
    DPGP_PRELAUNCH();
    hipLaunchKernelGGL(gemm_strided_f64_kernel, grid, dim3(256), 0, st, batch, m, n, k, 1.0, a, a_sb, a_si, a_sk, b, b_sb, b_sk,
                       b_sj, 0.0, c, c_sb, c_si, c_sj, ksplit, c_ss);
    DPGP_LAUNCH_CHECK();
    return DPGP_OK;
}
