// scratch: issue-bound floor of the psi2 phase-C instruction mix (per row: 64 v_exp_f32, 32 v_pk_add_f32, 12 mfma 32x32x16 f16,
// 12 v_fma_mix_f32, 12 v_cvt_pk_f16_f32, 6 v_pk_mul_f32, 15 v_mov_b32) with no data dependencies between the groups
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE>   // bit0: exps, bit1: adds, bit2: mfma, bit3: split, bit4: movs
__global__ __launch_bounds__(256, 2) void k(float *out, int iters) {
    const int lane = threadIdx.x;
    f32x16 c[4], acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) { c[i][j] = -1.0f - 0.01f * j - lane * 1e-4f; acc[i][j] = 0; }
    f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(lane * 1e-3f + i); bh[i] = (_Float16)(1.0f - i); }
    f32x2 x = {1.0f + lane * 1e-3f, 0.5f}, z = {0.25f, 1.5f - lane * 1e-3f}, p = {0, 0};
    unsigned h = 0, l = 0; float l0 = 0, l1 = 0, mv = 0;
    f32x16 e[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) e[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE & 4) {
#pragma unroll
                for (int ks = 0; ks < 3; ++ks) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[t]) : "v"(ah), "v"(bh));
            }
            if (MODE & 8) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (t * 2 + u < 6) {
                        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(z));
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x[0]), "v"(x[1]));
                        asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l0) : "v"(x[0]), "v"(z[0]), "v"(l));
                        asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(l1) : "v"(x[1]), "v"(z[1]), "v"(l));
                        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(z[0]), "v"(z[1]));
                    }
                }
            }
            if (MODE & 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) asm volatile("v_mov_b32 %0, %1" : "=v"(mv) : "v"(x[0]));
            }
            const int tp = (t + 2) & 3;       // exps read a tile issued two stages earlier
            if (MODE & 1) {
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_exp_f32 %0, %1" : "=v"(e[tp][j]) : "v"(c[tp][j]));
            }
            if (MODE & 2) {
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    f32x2 a = {acc[tp][j], acc[tp][j + 1]}, b = {e[tp][j], e[tp][j + 1]};
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
                    acc[tp][j] = a[0]; acc[tp][j + 1] = a[1];
                }
            }
        }
    }
    float r = p[0] + p[1] + l0 + l1 + mv + (float)h + (float)l;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j] + c[i][j] + e[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(float *d, const char *name) {
    const int it = 4000;
    for (int w = 1; w <= 2; ++w) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        k<MODE><<<256 * w, 256>>>(d, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k<MODE><<<256 * w, 256>>>(d, it);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s waves/SIMD %d: %.3f ms -> %.0f cycles@2.4GHz per row per wave, %.0f per row per SIMD\n", name, w, ms,
               ms * 1e-3 / it * 2.4e9, ms * 1e-3 / it * 2.4e9 / w);
    }
}
int main() {
    float *d;
    (void)hipMalloc(&d, 256 * 2 * 256 * 4);
    run<1>(d, "64 exp");
    run<3>(d, "64 exp + 32 pk_add");
    run<7>(d, "  + 12 mfma32");
    run<15>(d, "  + split (30)");
    run<31>(d, "  + 16 mov (full row mix)");
    run<4>(d, "12 mfma32 only");
    run<8>(d, "split only");
    return 0;
}
