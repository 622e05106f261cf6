// Micro-benchmarks (scratch, not part of the product):
//  1. do v_mfma_f32_16x16x4_f32 / v_mfma_f32_16x16x32_f16 co-execute with VALU (v_exp_f32 + v_pk_add)?
//  2. does the f16 MFMA flush subnormal inputs?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE, bool F16>   // MODE bit0: mfma, bit1: valu
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    const int lane = threadIdx.x;
    f32x4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = (f32x4){0, 0, 0, 0};
    float a = lane * 1e-3f, b = 1.0f - lane * 1e-3f;
    f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(a + i); bh[i] = (_Float16)(b - i); }
    float e[16], s = 0.f;
    for (int i = 0; i < 16; ++i) e[i] = -1.0f - 0.01f * i - lane * 1e-4f;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (F16) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c[i], 0, 0, 0);
                    else c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
                }
        }
        if (MODE & 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { s += __builtin_amdgcn_exp2f(e[i]); e[i] += 1e-6f; }
        }
    }
    float r = s;
    for (int i = 0; i < 4; ++i) r += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

__global__ void denorm(float *out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0; b[i] = (_Float16)0; }
    // lane's k-slot 0: subnormal f16 (2^-20) times 2^10 -> expect 2^-10 if subnormals are honoured, 0 if flushed
    a[0] = (_Float16)9.5367431640625e-07f;   // 2^-20 (subnormal in f16: min normal 2^-14)
    b[0] = (_Float16)1024.0f;
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
    // packed conversion check: round-to-nearest?
    float x = 1.0f + 1.5f * 0.0009765625f;   // 1 + 1.5 ulp(f16)
    _Float16 h = (_Float16)x;
    if (threadIdx.x == 0) { out[64] = (float)h; out[65] = x; }
}

template <int MODE, bool F16> float run(float *d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, F16><<<256 * 8, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, F16><<<256 * 8, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    int it = 20000;
    printf("f32 mfma 16x16x4 : mfma %.3f ms, valu %.3f ms, both %.3f ms\n", run<1, false>(d, it), run<2, false>(d, it), run<3, false>(d, it));
    printf("f16 mfma 16x16x32: mfma %.3f ms, valu %.3f ms, both %.3f ms\n", run<1, true>(d, it), run<2, true>(d, it), run<3, true>(d, it));
    denorm<<<1, 64>>>(d); hipDeviceSynchronize();
    float h[66]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("f16 MFMA subnormal input 2^-20 * 2^10: lane0 got %g (expect %g if honoured, 0 if flushed)\n", h[0], 0.0009765625);
    printf("cvt f32->f16 of %.10f gives %.10f\n", h[65], h[64]);
    return 0;
}
