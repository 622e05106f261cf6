// Micro-benchmark (scratch, not part of the product): issue cost of the instructions in the psi2 hot loop on gfx950.
// Each kernel runs ITER iterations of a block of independent instructions; time is reported as ns per wave-instruction
// per SIMD (x 2.4 = cycles at 2.4 GHz).  WAVES = waves per SIMD (1 or 2).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { EXP, ADD, PKADD, MIX, MFMA16, MFMA32, MFMA16_EXP2, MFMA16_ADD4, MFMA32_EXP8, EXP_ADD, MFMA16_EXP2_ADD2, MFMA16_C0, NCASE };
static const char *names[] = {"v_exp_f32", "v_add_f32", "v_pk_add_f32", "v_fma_mixlo_f16", "mfma16x16x32f16", "mfma32x32x16f16",
                              "mfma16 + 2 exp", "mfma16 + 4 add", "mfma32 + 8 exp", "exp + add", "mfma16 + 2exp + 2add", "mfma16 srcC=0"};
static const int ninstr[] = {16, 16, 8, 16, 8, 4, 8 * 3, 8 * 5, 4 * 9, 32, 8 * 5, 8};

template <int CASE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    const int lane = threadIdx.x;
    float e[16];
    for (int i = 0; i < 16; ++i) e[i] = -1.0f - 0.01f * i - lane * 1e-4f;
    float s[16];
    for (int i = 0; i < 16; ++i) s[i] = 0.0f;
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = (f32x4){0, 0, 0, 0};
    f32x16 cc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) cc[i][j] = 0;
    f16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(lane * 1e-3f + i); bh[i] = (_Float16)(1.0f - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep) {
        if (CASE == EXP) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %1" : "=v"(s[i]) : "v"(e[i]));
        } else if (CASE == ADD) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(s[i]) : "v"(e[i]));
        } else if (CASE == PKADD) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 a = {s[i], s[i + 1]}, b = {e[i], e[i + 1]};
                asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a) : "v"(b));
                s[i] = a[0]; s[i + 1] = a[1];
            }
        } else if (CASE == MIX) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(s[i]) : "v"(e[i]), "v"(e[(i + 1) & 15]));
        } else if (CASE == MFMA16) {
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c[i], 0, 0, 0);
        } else if (CASE == MFMA16_C0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(c[i]) : "v"(ah), "v"(bh));
        } else if (CASE == MFMA32) {
#pragma unroll
            for (int i = 0; i < 4; ++i) cc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, cc[i], 0, 0, 0);
        } else if (CASE == MFMA16_EXP2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(ah), "v"(bh));
                asm volatile("v_exp_f32 %0, %1" : "=v"(s[2 * i]) : "v"(e[2 * i]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(s[2 * i + 1]) : "v"(e[2 * i + 1]));
            }
        } else if (CASE == MFMA16_EXP2_ADD2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(ah), "v"(bh));
                asm volatile("v_exp_f32 %0, %1" : "=v"(s[2 * i]) : "v"(e[2 * i]));
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(e[2 * i]) : "v"(e[2 * i + 1]));
                asm volatile("v_exp_f32 %0, %1" : "=v"(s[2 * i + 1]) : "v"(e[2 * i + 1]));
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(e[2 * i + 1]) : "v"(e[2 * i]));
            }
        } else if (CASE == MFMA16_ADD4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[i]) : "v"(ah), "v"(bh));
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(s[2 * i]) : "v"(e[2 * i]));
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(s[2 * i + 1]) : "v"(e[2 * i + 1]));
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(s[(2 * i + 2) & 15]) : "v"(e[2 * i]));
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(s[(2 * i + 3) & 15]) : "v"(e[2 * i + 1]));
            }
        } else if (CASE == MFMA32_EXP8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(cc[i]) : "v"(ah), "v"(bh));
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_exp_f32 %0, %1" : "=v"(s[4 * i + j]) : "v"(e[4 * i + j]));
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("v_exp_f32 %0, %1" : "=v"(s[4 * i + j]) : "v"(e[(4 * i + j + 1) & 15]));
            }
        } else if (CASE == EXP_ADD) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_exp_f32 %0, %1" : "=v"(s[i]) : "v"(e[i]));
                asm volatile("v_add_f32 %0, %1, %0" : "+v"(e[i]) : "v"(e[(i + 1) & 15]));
            }
        }
      }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += s[i] + e[i];
    for (int i = 0; i < 8; ++i) r += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) r += cc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int CASE> float run(float *d, int waves, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves;
    k<CASE><<<blocks, 256>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CASE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int CASE> void report(float *d) {
    const int it = 2500;
    for (int w = 1; w <= 2; ++w) {
        const float ms = run<CASE>(d, w, it);
        // per SIMD: w waves x it x ninstr wave-instructions in ms
        const double ns_per = ms * 1e6 / ((double)w * it * 8 * ninstr[CASE]);
        printf("%-22s waves/SIMD %d: %8.3f ms  %6.2f ns per wave-instr = %5.1f cycles @2.4GHz (block of %d: %.0f cycles)\n",
               names[CASE], w, ms, ns_per, ns_per * 2.4, ninstr[CASE], ns_per * 2.4 * ninstr[CASE]);
    }
}

int main() {
    float *d;
    hipMalloc(&d, 256 * 2 * 256 * 4);
    report<EXP>(d); report<ADD>(d); report<PKADD>(d); report<MIX>(d); report<MFMA16>(d); report<MFMA16_C0>(d); report<MFMA32>(d);
    report<MFMA16_EXP2>(d); report<MFMA16_ADD4>(d); report<MFMA32_EXP8>(d); report<EXP_ADD>(d); report<MFMA16_EXP2_ADD2>(d);
    return 0;
}
