// scratch: issue cost (2 waves per SIMD) of candidate instructions for the hi/lo operand split
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define KERNEL(NAME, ASM, ...)                                                                  \
    __global__ __launch_bounds__(256) void NAME(float *out, int iters) {                        \
        float e[16], s[16];                                                                     \
        for (int i = 0; i < 16; ++i) { e[i] = 1.0f + 0.01f * i + threadIdx.x * 1e-4f; s[i] = 0.5f * i; } \
        for (int it = 0; it < iters; ++it) {                                                    \
            _Pragma("unroll") for (int rep = 0; rep < 16; ++rep) _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(s[i]) : "v"(e[i]), "v"(e[(i + 5) & 15])); \
        }                                                                                       \
        float r = 0;                                                                            \
        for (int i = 0; i < 16; ++i) r += s[i];                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                \
    }
KERNEL(k_add, "v_add_f32 %0, %1, %2")
KERNEL(k_mul, "v_mul_f32 %0, %1, %2")
KERNEL(k_fma, "v_fma_f32 %0, %1, %2, %0")
KERNEL(k_cvtpk, "v_cvt_pk_f16_f32 %0, %1, %2")
KERNEL(k_cvtpkrtz, "v_cvt_pkrtz_f16_f32 %0, %1, %2")
KERNEL(k_cvt16, "v_cvt_f16_f32 %0, %1")
KERNEL(k_cvt32, "v_cvt_f32_f16 %0, %1")
KERNEL(k_cvt32s, "v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
KERNEL(k_mixlo, "v_fma_mixlo_f16 %0, %1, %2, 0")
KERNEL(k_mix, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,0,0]")
KERNEL(k_mix16, "v_fma_mix_f32 %0, %1, %2, -%0 op_sel_hi:[0,0,1]")
KERNEL(k_max, "v_max_f32 %0, %1, %2")
KERNEL(k_perm, "v_perm_b32 %0, %1, %2, %0")
KERNEL(k_pkaddf16, "v_pk_add_f16 %0, %1, %2")
KERNEL(k_pkfmaf16, "v_pk_fma_f16 %0, %1, %2, %0")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %1, %2")
KERNEL(k_log, "v_log_f32 %0, %1")
KERNEL(k_rcp, "v_rcp_f32 %0, %1")
KERNEL(k_mov, "v_mov_b32 %0, %1")
KERNEL(k_addu, "v_add_u32 %0, %1, %2")
KERNEL(k_and, "v_and_b32 %0, %1, %2")
KERNEL(k_mad24, "v_mad_u32_u24 %0, %1, %2, %0")
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %1, 2, %2")

__global__ __launch_bounds__(256) void k_pkmul(float *out, int iters) {
    f32x2 e[8], s[8];
    for (int i = 0; i < 8; ++i) { e[i] = (f32x2){1.0f + 0.01f * i, 1.0f + threadIdx.x * 1e-4f}; s[i] = (f32x2){0.5f * i, 1.0f}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(s[i]) : "v"(e[i]), "v"(e[(i + 3) & 7]));
#pragma unroll
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(s[i]) : "v"(e[i]), "v"(e[(i + 5) & 7]));
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += s[i][0] + s[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_pkfma(float *out, int iters) {
    f32x2 e[8], s[8];
    for (int i = 0; i < 8; ++i) { e[i] = (f32x2){1.0f + 0.01f * i, 1.0f + threadIdx.x * 1e-4f}; s[i] = (f32x2){0.5f * i, 1.0f}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(e[i]), "v"(e[(i + 3) & 7]));
#pragma unroll
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(e[i]), "v"(e[(i + 5) & 7]));
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += s[i][0] + s[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k_lds(float *out, int iters) {
    __shared__ float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = i;
    __syncthreads();
    float s[16];
    for (int i = 0; i < 16; ++i) s[i] = 0;
    int a = (threadIdx.x * 4) & 4095;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(s[i]) : "v"(a), "n"(i * 256));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

typedef void (*kern_t)(float *, int);
static void report(const char *name, kern_t f, float *d) {
    const int it = 2000;
    for (int w = 1; w <= 2; ++w) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        f<<<256 * w, 256>>>(d, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        f<<<256 * w, 256>>>(d, it);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double ns = ms * 1e6 / ((double)w * it * 256);
        printf("%-28s waves/SIMD %d: %7.3f ms %6.2f ns/instr = %5.1f cycles @2.4GHz\n", name, w, ms, ns, ns * 2.4);
    }
}
int main() {
    float *d;
    (void)hipMalloc(&d, 256 * 2 * 256 * 4);
#define R(k) report(#k, k, d)
    R(k_add); R(k_mul); R(k_fma); R(k_cvtpk); R(k_cvtpkrtz); R(k_cvt16); R(k_cvt32); R(k_cvt32s); R(k_mixlo); R(k_mix); R(k_mix16);
    R(k_max); R(k_perm); R(k_pkaddf16); R(k_pkfmaf16); R(k_ldexp); R(k_log); R(k_rcp); R(k_mov); R(k_addu); R(k_and); R(k_mad24);
    R(k_lshladd); R(k_pkmul); R(k_pkfma); R(k_lds);
    return 0;
}
