// Shared device helpers for libdpgp_hip (gfx950 / CDNA4 only: 64-wide wavefronts, MFMA 16x16x4 f32/f64).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "../../include/dpgp.h"

#define DPGP_WAVE 64
#define DPGP_TB 16            // MFMA tile edge (v_mfma_*_16x16x4_*)
#define DPGP_LOG2E 1.4426950408889634074
#define DPGP_LOG_2PI 1.8378770664093454835606594728112

static inline int dpgp_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int dpgp_round_up(int a, int b) { return dpgp_ceil_div(a, b) * b; }
static inline size_t dpgp_align256(size_t x) { return (x + 255) & ~(size_t)255; }

// hipGetLastError() is sticky across unrelated runtime calls of the process (torch's own included): clear it first.
#define DPGP_PRELAUNCH() (void)hipGetLastError()
// The last HIP error seen by a launch check of this thread (diagnostics only; see dpgp_last_hip_error()).
inline hipError_t &dpgp_last_error_slot() {
    static thread_local hipError_t e = hipSuccess;
    return e;
}
#define DPGP_LAUNCH_CHECK()                              \
    do {                                                 \
        hipError_t e_ = hipGetLastError();               \
        if (e_ != hipSuccess) {                          \
            dpgp_last_error_slot() = e_;                 \
            return DPGP_ERR_LAUNCH;                      \
        }                                                \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// One 16x16x4 matrix-core step D = A(16x4) * B(4x16) + C for one wavefront.
//   A operand: lane l holds A[i = l & 15][k = l >> 4];  B operand: lane l holds B[k = l >> 4][j = l & 15].
//   C/D: 4 values per lane, column j = l & 15, row given by Mfma<T>::row(l, r)   (f32 and f64 differ!).
template <typename T> struct Mfma;
template <> struct Mfma<float> {
    typedef f32x4 acc_t;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return ((lane >> 4) << 2) + r; }
};
template <> struct Mfma<double> {
    typedef f64x4 acc_t;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + (r << 2); }
};

// base-2 exponential: fp32 -> the bare v_exp_f32 (results below 2^-126 flush to 0, which is what a sum of such terms
// next to O(1) terms needs); fp64 -> 2^x = 2^k 2^f, k = rint(x), |f| <= 1/2, 2^f by the degree-12 Taylor polynomial of
// exp(f ln 2) (truncation (ln2 / 2)^13 / 13! = 1.7e-16 relative), 2^k by an exponent-field add; arguments below -1020 give
// 0, above 1020 inf, NaN stays NaN (the kernels only exponentiate log-densities <= ~0).  About half the
// instructions of the device library's exp2, which handles the full range and denormal results.
__device__ __forceinline__ float dpgp_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ double dpgp_exp2(double x) {
    const double xc = x < -1020.0 ? -1020.0 : x;
    const double k = __builtin_rint(xc), f = xc - k;
    const double c[13] = {
        1.00000000000000000e+00,
        6.93147180559945286e-01,
        2.40226506959100694e-01,
        5.55041086648215762e-02,
        9.61812910762847688e-03,
        1.33335581464284411e-03,
        1.54035303933816061e-04,
        1.52527338040598377e-05,
        1.32154867901443053e-06,
        1.01780860092396960e-07,
        7.05491162080112088e-09,
        4.44553827187081007e-10,
        2.56784359934881958e-11};
    double p = c[12];
#pragma unroll
    for (int i = 11; i >= 0; --i) p = __builtin_fma(p, f, c[i]);
    long long bits = __builtin_bit_cast(long long, p) + ((long long)(int)k << 52);
    if (!(x <= 1020.0)) return (x != x) ? x : __builtin_inf();     // NaN stays NaN, overflow -> inf
    return x < -1020.0 ? 0.0 : __builtin_bit_cast(double, bits);
}
__device__ __forceinline__ float dpgp_log(float x) { return logf(x); }
__device__ __forceinline__ double dpgp_log(double x) { return log(x); }

// Wavefront (64 lanes) sum.
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Workgroup sum through LDS (scratch: >= blockDim.x/64 elements). Result valid in every thread.
template <typename T> __device__ __forceinline__ T block_sum(T v, T *scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    T t = 0;
    for (int i = 0; i < nw; ++i) t += scratch[i];
    return t;
}

// Column means of z[M,Q] (Q <= 30) by the whole 256-thread workgroup: thread = (column t & 31, row group t >> 5), partial
// sums in fp64, 8-way reduction through LDS (scratch: 8 * 32 doubles).  Result in zc[0..Q) (type T), valid after the
// trailing barrier.  Replaces a per-column serial loop over M dependent loads in every workgroup's prologue.
template <typename TIN, typename T>
__device__ __forceinline__ void block_column_means(const TIN *__restrict__ z, int M, int Q, T *zc, double *scratch) {
    const int t = threadIdx.x, q = t & 31, rg = t >> 5;
    double a = 0.0;
    if (q < Q)
        for (int m = rg; m < M; m += 8) a += (double)z[(size_t)m * Q + q];
    scratch[rg * 32 + q] = a;
    __syncthreads();
    if (t < Q) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += scratch[k * 32 + t];
        zc[t] = (T)(v / (double)M);
    }
    __syncthreads();
}
