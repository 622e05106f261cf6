// scratch: fp32 FMA throughput of the whole chip, packed (v_pk_fma_f32) vs plain (v_fma_f32), at full occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK> __global__ __launch_bounds__(256) void k(float *sink, int iters) {
    f2 a[8], y = {1.0001f, 0.9999f}, z = {1e-7f, 2e-7f};
    for (int i = 0; i < 8; ++i) a[i] = (f2){threadIdx.x * 1e-3f + i, i * 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(y), "v"(z));
                else {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i][0]) : "v"(y[0]), "v"(z[0]));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i][1]) : "v"(y[1]), "v"(z[1]));
                }
            }
    }
    float t = 0; for (int i = 0; i < 8; ++i) t += a[i][0] + a[i][1];
    sink[blockIdx.x * 256 + threadIdx.x] = t;
}
int main() {
    float *sink; (void)hipMalloc(&sink, 4096 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000, blocks = 4096;
    for (int pk = 0; pk < 2; ++pk) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (pk) k<1><<<blocks, 256>>>(sink, iters); else k<0><<<blocks, 256>>>(sink, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        const double fl = 2.0 * 128 * iters * 256.0 * blocks;
        printf("%s: %.3f ms -> %.1f TFLOP/s fp32\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms, fl / ms / 1e9);
    }
    return 0;
}
