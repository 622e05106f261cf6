// scratch: issue cost of packed fp32 vector instructions vs scalar-per-lane ones on gfx950 (one wave; long loops, s_memtime)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 2000
#define BEGIN() long long m0 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); for (int it = 0; it < ITERS; ++it) {
#define END(slot, n) } long long m1 = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (threadIdx.x == 0) { out[2 * (slot)] = m1 - m0; out[2 * (slot) + 1] = (long long)ITERS * (n); }
__global__ __launch_bounds__(64) void k(long long *out, float *sink) {
    f2 a[16], y = {1.0001f, 0.9999f}, z = {1e-7f, 2e-7f};
    float s[16], ys = 1.0001f, zs = 1e-7f;
    for (int i = 0; i < 16; ++i) { a[i] = (f2){threadIdx.x * 1e-3f + i, i * 0.5f}; s[i] = threadIdx.x * 1e-3f + i; }
    { BEGIN()
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(y), "v"(z));
      END(0, 64) }
    { BEGIN()
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(z));
      END(1, 64) }
    { BEGIN()
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
      END(2, 64) }
    { BEGIN()
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(ys), "v"(zs));
      END(3, 64) }
    { BEGIN()
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s[i]) : "v"(zs));
      END(4, 64) }
    { BEGIN()
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(s[i]));
      END(5, 64) }
    float t = 0; for (int i = 0; i < 16; ++i) t += a[i][0] + a[i][1] + s[i];
    sink[threadIdx.x] = t;
}
int main() {
    long long *out; float *sink; (void)hipMalloc(&out, 256); (void)hipMalloc(&sink, 256);
    for (int w = 0; w < 2; ++w) k<<<1, 64>>>(out, sink);
    (void)hipDeviceSynchronize();
    long long h[16]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[] = {"v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_fma_f32", "v_sub_f32", "v_exp_f32"};
    for (int i = 0; i < 6; ++i) printf("%-14s %.2f cycles per instruction (independent, one wave)\n", nm[i], (double)h[2 * i] / h[2 * i + 1]);
    return 0;
}
