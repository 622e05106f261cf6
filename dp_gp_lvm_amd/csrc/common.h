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
// fp64 base-2 exponential for the streaming hot loops (one exp2 per (n, m, m')): 2^x = 2^k T[j] 2^r with x = k + j/64 + r,
// |r| <= 1/128.  k and j come out of ONE addition of 1.5 * 2^46 (the sum's ulp is 1/64, so its low dword is round(64 x) in
// two's complement), T[j] = 2^(j/64) from a 64-entry table that the workgroup keeps in LDS (512 B; lanes reading the same
// entry broadcast), 2^r by the degree-5 Taylor polynomial of exp(r ln 2) (truncation (ln2/128)^6/720 = 3.5e-17 relative),
// 2^k by v_ldexp_f64, which also gives 0 / inf for arguments far outside the exponent range; NaN in gives NaN out.
// 12 fp64 + 3 integer instructions + one ds_read_b64 per value against ~24 fp64 instructions of dpgp_exp2(double)
// (measured on gfx950, scratch/ubench/f64exp.hip: 70 vs 170 issue cycles per wave-instruction-equivalent; worst relative
// error 2.5e-16 on [-1100, 50]).  Valid for x < 2^25 (larger arguments mean an infinite result anyway).
static __device__ const double dpgp_exp2_table[64] = {
    1.00000000000000000e+00, 1.01088928605170048e+00, 1.02189714865411663e+00, 1.03302487902122841e+00,
    1.04427378242741375e+00, 1.05564517836055716e+00, 1.06714040067682370e+00, 1.07876079775711986e+00,
    1.09050773266525769e+00, 1.10238258330784089e+00, 1.11438674259589243e+00, 1.12652161860824185e+00,
    1.13878863475669156e+00, 1.15118922995298267e+00, 1.16372485877757748e+00, 1.17639699165028122e+00,
    1.18920711500272103e+00, 1.20215673145270308e+00, 1.21524735998046896e+00, 1.22848053610687002e+00,
    1.24185781207348400e+00, 1.25538075702469110e+00, 1.26905095719173322e+00, 1.28287001607877826e+00,
    1.29683955465100964e+00, 1.31096121152476441e+00, 1.32523664315974132e+00, 1.33966752405330292e+00,
    1.35425554693689265e+00, 1.36900242297459052e+00, 1.38390988196383202e+00, 1.39897967253831124e+00,
    1.41421356237309515e+00, 1.42961333839197002e+00, 1.44518080697704665e+00, 1.46091779418064704e+00,
    1.47682614593949935e+00, 1.49290772829126484e+00, 1.50916442759342284e+00, 1.52559815074453842e+00,
    1.54221082540794074e+00, 1.55900440023783693e+00, 1.57598084510788650e+00, 1.59314215134226700e+00,
    1.61049033194925428e+00, 1.62802742185734783e+00, 1.64575547815396495e+00, 1.66367658032673638e+00,
    1.68179283050742900e+00, 1.70010635371852348e+00, 1.71861929812247793e+00, 1.73733383527370622e+00,
    1.75625216037329945e+00, 1.77537649252652119e+00, 1.79470907500310717e+00, 1.81425217550039886e+00,
    1.83400808640934243e+00, 1.85397912508338547e+00, 1.87416763411029996e+00, 1.89457598158696561e+00,
    1.91520656139714740e+00, 1.93606179349229435e+00, 1.95714412417540018e+00, 1.97845602638795093e+00};
#define DPGP_EXP2_TAB_ELEMS 64
// fills tab[0..64) (LDS) by the calling workgroup's first 64 threads; the caller's next barrier publishes it
__device__ __forceinline__ void dpgp_exp2_tab_init(double *tab) {
    if (threadIdx.x < DPGP_EXP2_TAB_ELEMS) tab[threadIdx.x] = dpgp_exp2_table[threadIdx.x];
}
__device__ __forceinline__ double dpgp_exp2_tab(double x, const double *tab) {
    const double MAGIC = 105553116266496.0;                       // 1.5 * 2^46
    const double xc = x < -1100.0 ? -1100.0 : x;                  // 2^-1100 = 0; keeps round(64 x) inside 32 bits.  A select,
                                                                  // not fmax: a NaN argument stays NaN through every step below
                                                                  // (fmax would turn it into 2^-1100 = 0)
    const double t = xc + MAGIC;
    const int lo = (int)__builtin_bit_cast(long long, t);         // round(64 x)
    const double r = xc - (t - MAGIC);
    double p = 1.33335581464284433e-03;
    p = __builtin_fma(p, r, 9.61812910762847688e-03);
    p = __builtin_fma(p, r, 5.55041086648215831e-02);
    p = __builtin_fma(p, r, 2.40226506959100722e-01);
    p = __builtin_fma(p, r, 6.93147180559945286e-01);
    const double tj = tab[lo & 63];
    return __builtin_ldexp(__builtin_fma(tj * r, p, tj), lo >> 6);
}
// the streaming kernels' one call site for both types: fp32 ignores the table
__device__ __forceinline__ float dpgp_exp2_hot(float x, const double *) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ double dpgp_exp2_hot(double x, const double *tab) { return dpgp_exp2_tab(x, tab); }
__device__ __forceinline__ float dpgp_log(float x) { return logf(x); }
__device__ __forceinline__ double dpgp_log(double x) { return log(x); }

// Fixes a value in its fp32 register before it is split into an f16 (hi, lo) pair.  Without it the compiler (with
// -ffp-contract=fast) forms  hi = f16(fl32(x y))  with v_cvt but the residual  x y - hi'  with hi' = v_fma_mixlo_f16(x, y, 0),
// the f16 rounding of the EXACT product: near a rounding tie hi' != hi and the pair is off by one f16 ulp of hi
// (found on the GPU: 2e-3 relative on single terms of Psi2; ISA: v_cvt_pk_f16_f32 next to v_fma_mixlo_f16 of the same product).
__device__ __forceinline__ float dpgp_pin(float x) {
    asm volatile("" : "+v"(x));
    return x;
}
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
