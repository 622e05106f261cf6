// Micro-benchmark (scratch, not part of the product): fp64 exponent tiles.
//  1. does v_mfma_f64_16x16x4_f64 co-execute with fp64 VALU work of other waves / of the same wave?
//  2. cost and accuracy of candidate fp64 exp2 forms: degree-12 polynomial (common.h) vs 32-entry table + degree 5.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double exp2_poly12(double x) {
    const double xc = x < -1020.0 ? -1020.0 : x;
    const double k = __builtin_rint(xc), f = xc - k;
    const double c[13] = {1.00000000000000000e+00, 6.93147180559945286e-01, 2.40226506959100694e-01, 5.55041086648215762e-02,
                          9.61812910762847688e-03, 1.33335581464284411e-03, 1.54035303933816061e-04, 1.52527338040598377e-05,
                          1.32154867901443053e-06, 1.01780860092396960e-07, 7.05491162080112088e-09, 4.44553827187081007e-10,
                          2.56784359934881958e-11};
    double p = c[12];
#pragma unroll
    for (int i = 11; i >= 0; --i) p = __builtin_fma(p, f, c[i]);
    long long bits = __builtin_bit_cast(long long, p) + ((long long)(int)k << 52);
    if (!(x <= 1020.0)) return (x != x) ? x : __builtin_inf();
    return x < -1020.0 ? 0.0 : __builtin_bit_cast(double, bits);
}

// 2^x = 2^k * T[j] * 2^r,  x = k + j/32 + r, |r| <= 1/64: magic-number rounding, table in LDS, degree-5 polynomial
// of 2^r - 1 (Taylor in r ln2: truncation (ln2/64)^6/720 = 2.2e-15 relative), v_ldexp_f64 for the scaling.
template <int DEG>
__device__ __forceinline__ double exp2_tab(double x, const double *tab) {
    const double MAGIC = 211106232532992.0;   // 1.5 * 2^47: ulp = 2^-5
    const double t = x + MAGIC;
    const int lo = (int)__builtin_bit_cast(long long, t);        // low dword: round(32 x) (two's complement)
    const double r = x - (t - MAGIC);
    const double L = 0.693147180559945309417232;
    const double c1 = L, c2 = L * L / 2, c3 = L * L * L / 6, c4 = L * L * L * L / 24, c5 = L * L * L * L * L / 120,
                 c6 = L * L * L * L * L * L / 720;
    double p = DEG >= 6 ? c6 : c5;
    if (DEG >= 6) p = __builtin_fma(p, r, c5);
    p = __builtin_fma(p, r, c4);
    p = __builtin_fma(p, r, c3);
    p = __builtin_fma(p, r, c2);
    p = __builtin_fma(p, r, c1);
    const double tj = tab[lo & 31];
    const double v = __builtin_fma(tj * r, p, tj);
    return __builtin_ldexp(v, lo >> 5);
}

template <int MODE, int EXPK>   // MODE bit0: mfma f64, bit1: valu exp2; EXPK 0 poly12, 1 tab5, 2 tab6
__global__ __launch_bounds__(256, 2) void k(double *out, int iters) {
    __shared__ double tab[32];
    if (threadIdx.x < 32) tab[threadIdx.x] = exp2((double)threadIdx.x / 32.0);
    __syncthreads();
    const int lane = threadIdx.x;
    f64x4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = (f64x4){0, 0, 0, 0};
    double a = lane * 1e-3, b = 1.0 - lane * 1e-3;
    double e[16], s = 0.0;
    for (int i = 0; i < 16; ++i) e[i] = -1.0 - 0.37 * i - lane * 1e-2;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s += EXPK == 0 ? exp2_poly12(e[i]) : (EXPK == 1 ? exp2_tab<5>(e[i], tab) : exp2_tab<6>(e[i], tab));
                e[i] += 1e-6;
            }
        }
    }
    double r = s;
    for (int i = 0; i < 4; ++i) r += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int EXPK> __global__ void acc_kernel(const double *x, double *y, int n) {
    __shared__ double tab[32];
    if (threadIdx.x < 32) tab[threadIdx.x] = exp2((double)threadIdx.x / 32.0);
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = EXPK == 0 ? exp2_poly12(x[i]) : (EXPK == 1 ? exp2_tab<5>(x[i], tab) : exp2_tab<6>(x[i], tab));
}

template <int MODE, int EXPK> float run(double *d, int iters, int wgs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, EXPK><<<wgs, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, EXPK><<<wgs, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int EXPK> void accuracy(const char *name) {
    const int n = 1 << 20;
    std::vector<double> hx(n), hy(n);
    for (int i = 0; i < n; ++i) hx[i] = -1100.0 + 1150.0 * ((double)rand() / RAND_MAX) * ((double)rand() / RAND_MAX);
    hx[0] = 0.0; hx[1] = -0.5; hx[2] = 0.5; hx[3] = -1074.0; hx[4] = -5000.0; hx[5] = 1023.5; hx[6] = 5000.0; hx[7] = NAN;
    hx[8] = 1.0 / 64; hx[9] = -1.0 / 64; hx[10] = 3.0 / 64; hx[11] = 1e-300;
    double *dx, *dy; hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8);
    hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
    acc_kernel<EXPK><<<n / 256, 256>>>(dx, dy, n);
    hipMemcpy(hy.data(), dy, n * 8, hipMemcpyDeviceToHost);
    double worst = 0; int wi = 0;
    for (int i = 12; i < n; ++i) {
        double ref = exp2(hx[i]);
        if (ref < 1e-290) continue;
        double e = fabs(hy[i] - ref) / ref;
        if (e > worst) { worst = e; wi = i; }
    }
    printf("%s: worst rel err %.3e at x=%.6f; specials:", name, worst, hx[wi]);
    for (int i = 0; i < 12; ++i) printf(" [%g -> %g]", hx[i], hy[i]);
    printf("\n");
    hipFree(dx); hipFree(dy);
}
int main() {
    double *d; hipMalloc(&d, 256 * 8 * 256 * 8);
    int it = 4000;
    for (int wgs : {256 * 8, 256 * 2, 256}) {
        printf("wgs %d (%d per CU)\n", wgs, wgs / 256);
        printf(" poly12: mfma %.3f ms, valu %.3f ms, both %.3f ms\n", run<1, 0>(d, it, wgs), run<2, 0>(d, it, wgs), run<3, 0>(d, it, wgs));
        printf(" tab5  : mfma %.3f ms, valu %.3f ms, both %.3f ms\n", run<1, 1>(d, it, wgs), run<2, 1>(d, it, wgs), run<3, 1>(d, it, wgs));
        printf(" tab6  : mfma %.3f ms, valu %.3f ms, both %.3f ms\n", run<1, 2>(d, it, wgs), run<2, 2>(d, it, wgs), run<3, 2>(d, it, wgs));
    }
    accuracy<0>("poly12"); accuracy<1>("tab5"); accuracy<2>("tab6");
    return 0;
}
