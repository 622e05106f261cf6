// Micro-benchmark (scratch): the pipeline stage of psi2_pairs.hip — KS dependent v_mfma_f32_32x32x16_f16 into one VGPR
// accumulator tile, interleaved with 16 v_exp_f32 + 16 v_add_f32 on another tile — cycles per stage by variant.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int MODE, int WAVES, int KS>   // bit0: mfma chain, bit1: exps, bit2: adds, bit3: two independent half-chains (c = cA + cB not added)
__global__ __launch_bounds__(256, WAVES) void k(float *out, int iters, const h8 *src) {
    h8 a[KS], b[KS];
    for (int i = 0; i < KS; ++i) { a[i] = src[threadIdx.x + 64 * (i & 3)]; b[i] = src[threadIdx.x + 64 * ((i & 3) + 4)]; }
    f16v c0, c1, c2;
    for (int v = 0; v < 16; ++v) { c0[v] = -1.0f - 0.01f * v; c1[v] = -2.0f; c2[v] = 0.f; }
    float acc[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f16v &cn = half ? c0 : c1;
            f16v &cu = half ? c1 : c0;
            if (MODE & 1) {
#pragma unroll
                for (int v = 0; v < 16; ++v) cn[v] = 0.0f;
                if (MODE & 8) for (int v = 0; v < 16; ++v) c2[v] = 0.0f;
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (MODE & 1) {
                    if ((MODE & 8) && (ks & 1)) c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b[ks], c2, 0, 0, 0);
                    else cn = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b[ks], cn, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 16 * ks / KS; v < 16 * (ks + 1) / KS; ++v) {
                    float e = cu[v];
                    if (MODE & 2) e = __builtin_amdgcn_exp2f(e);
                    if (MODE & 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[v & 3]) : "v"(e));
                    else asm volatile("" :: "v"(e));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!(MODE & 1)) for (int v = 0; v < 16; ++v) cn[v] = cu[v] * 0.999f - 0.5f;
        }
    }
    float r = acc[0] + acc[1] + acc[2] + acc[3];
    for (int v = 0; v < 16; ++v) r += c0[v] + c1[v] + c2[v];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE, int WAVES, int KS> float run(float *d, const h8 *src, int iters, int wgs) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, WAVES, KS><<<wgs, 256>>>(d, 10, src);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, WAVES, KS><<<wgs, 256>>>(d, iters, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int KS> void suite(float *d, const h8 *src) {
    const int it = 20000;
    for (int wpc : {2}) {
        int wgs = 256 * wpc;
        auto rep = [&](const char *name, float ms) {
            double stages = (double)wpc * it * 2;
            printf("  KS=%d %-34s %8.3f ms  -> %6.1f ns per stage per SIMD\n", KS, name, ms, ms * 1e6 / stages);
        };
        rep("mfma chain only", run<1, 2, KS>(d, src, it, wgs));
        rep("exp + add only", run<6, 2, KS>(d, src, it, wgs));
        rep("mfma chain + exp + add", run<7, 2, KS>(d, src, it, wgs));
        rep("two half-chains + exp + add", run<15, 2, KS>(d, src, it, wgs));
    }
}
int main() {
    float *d; hipMalloc(&d, 4096 * 256 * 4);
    h8 *src; hipMalloc(&src, 64 * 8 * 16 * 4); hipMemset(src, 0x3c, 64 * 8 * 16 * 4);
    suite<4>(d, src); suite<6>(d, src); suite<8>(d, src);
    return 0;
}
