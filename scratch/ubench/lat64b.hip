// Micro-benchmark (scratch): steady-state cost per instruction of the fp64 / cross-lane primitives of the Cholesky pivot
// chain, from long loops (2000 x 64 instructions) timed with s_memtime (shader cycles) and s_memrealtime (100 MHz): one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define ITERS 2000
#define BEGIN() long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); for (int it = 0; it < ITERS; ++it) {
#define END(slot, n) } long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (threadIdx.x == 0) { out[3 * (slot)] = m1 - m0; out[3 * (slot) + 1] = r1 - r0; out[3 * (slot) + 2] = (long long)ITERS * (n); }

__global__ __launch_bounds__(64) void k(long long *out, double *sink, double seed) {
    __shared__ double lds[1024];
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-3, y = 1.0 + lane * 1e-7, z = 1e-9;
    for (int i = lane; i < 1024; i += 64) lds[i] = i;
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = x + i;
    { BEGIN()                                                       // 0: dependent v_fma_f64
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
      END(0, 64) }
    { BEGIN()                                                       // 1: 16 independent v_fma_f64 chains
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(y), "v"(z));
      END(1, 64) }
    { BEGIN()                                                       // 2: dependent v_fma_f32
      float f = (float)x, g = (float)y, h = (float)z;
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(g), "v"(h));
      x += f;
      END(2, 64) }
    { float f[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) f[i] = (float)a[i];
      float g = (float)y, h = (float)z;
      BEGIN()                                                       // 3: 16 independent v_fma_f32
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(g), "v"(h));
      END(3, 64)
#pragma unroll
      for (int i = 0; i < 16; ++i) x += f[i]; }
    { BEGIN()                                                       // 4: 16 independent readlane pairs (32 v_readlane) + nothing else, x2
      int acc = 0;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
          int lo, hi;
          asm volatile("v_readlane_b32 %0, %2, 5\n\tv_readlane_b32 %1, %3, 5" : "=s"(lo), "=s"(hi) : "v"((int)__builtin_bit_cast(long long, a[i])), "v"((int)(__builtin_bit_cast(long long, a[i]) >> 32)));
          acc ^= lo ^ hi;
      }
      if (acc == 0x12345) x += 1;
      END(4, 64) }
    { BEGIN()                                                       // 5: 16 x (readlane pair + fma with the SGPR pair) independent
#pragma unroll
      for (int i = 0; i < 16; ++i) {
          const unsigned long long u = __builtin_bit_cast(unsigned long long, a[(i + 1) & 15]);
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 5);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 5);
          const double s = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
          asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(z), "s"(s));
      }
      END(5, 48) }
    { BEGIN()                                                       // 6: dependent v_rsq_f64 (+ mul to keep range)
#pragma unroll
      for (int i = 0; i < 32; ++i) { asm volatile("v_rsq_f64 %0, %0" : "+v"(x)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(y)); }
      END(6, 32) }
    { f64x4 c0 = {x, x, x, x}, c1 = {y, x, x, x}, c2 = {x, y, x, x}, c3 = {x, x, y, x};
      BEGIN()                                                       // 7: mfma f64 16x16x4, 4 independent accumulators
#pragma unroll
      for (int i = 0; i < 4; ++i) {
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(z, y, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(z, z, c3, 0, 0, 0);
      }
      END(7, 16)
      x += c0[0] + c1[1] + c2[2] + c3[3]; }
    { f64x4 c0 = {x, x, x, x};
      BEGIN()                                                       // 8: mfma f64 16x16x4 one accumulator chain
#pragma unroll
      for (int i = 0; i < 16; ++i) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, c0, 0, 0, 0);
      END(8, 16)
      x += c0[0] + c0[1]; }
    { f64x4 c0 = {x, x, x, x}, c1 = {y, x, x, x};
      BEGIN()                                                       // 9: 2 mfma f64 + 16 independent v_fma_f64 interleaved (co-issue?)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, c0, 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 8; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(y), "v"(z));
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(z, y, c1, 0, 0, 0);
#pragma unroll
          for (int j = 8; j < 16; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(y), "v"(z));
      }
      END(9, 8)
      x += c0[0] + c1[1]; }
    { double t[8];
      BEGIN()                                                       // 10: 8 ds_read_b64 (strided rows) + wait
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = lds[(lane * 17 + i + it) & 1023];
#pragma unroll
      for (int i = 0; i < 8; ++i) x += t[i];
      END(10, 8) }
    { BEGIN()                                                       // 11: dependent v_mul_f64
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(y));
      END(11, 64) }
    { BEGIN()                                                       // 12: v_mov_b64 dpp row_newbcast, 16 independent
      double b[16];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(b[i]) : "v"(a[i]));
#pragma unroll
      for (int i = 0; i < 16; ++i) x += b[i];
      END(12, 64) }
    { BEGIN()                                                       // 13: s_barrier alone (one wave)
#pragma unroll
      for (int i = 0; i < 8; ++i) __syncthreads();
      END(13, 8) }

    { BEGIN()                                                       // 14: v_fmac_f64_dpp row_newbcast, 16 independent destinations
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(y), "v"(z));
      END(14, 64) }
    { BEGIN()                                                       // 15: v_fmac_f64_dpp in place (src0 = dst), 16 independent
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f64_dpp %0, -%0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(z));
      END(15, 64) }
    { BEGIN()                                                       // 16: v_fmac_f64_e32 (VOP2), 16 independent
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "v"(y), "v"(z));
      END(16, 64) }
    { BEGIN()                                                       // 17: v_fmac_f64_dpp dependent chain
#pragma unroll
      for (int i = 0; i < 64; ++i) asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y), "v"(z));
      END(17, 64) }
    { BEGIN()                                                       // 18: 2 waves worth? (placeholder) v_rsq_f64 independent x16
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rsq_f64 %0, %1" : "=v"(a[i]) : "v"(y));
      END(18, 32) }
#pragma unroll
    for (int i = 0; i < 16; ++i) x += a[i];
    sink[lane] = x;
}
int main() {
    long long *out; double *sink;
    (void)hipMalloc(&out, 128 * 8); (void)hipMalloc(&sink, 64 * 8);
    for (int w = 0; w < 2; ++w) k<<<1, 64>>>(out, sink, 1.0);
    (void)hipDeviceSynchronize();
    long long h[128]; (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const char *nm[] = {"v_fma_f64 dependent", "v_fma_f64 16 independent", "v_fma_f32 dependent", "v_fma_f32 16 independent",
                        "v_readlane_b32 pairs (per pair)", "readlane pair + fma_f64(sgpr) (per instr of 3)", "v_rsq_f64 + fma dependent (per pair)",
                        "mfma_f64_16x16x4 4 indep acc", "mfma_f64_16x16x4 1 acc chain", "mfma_f64 + 8 fma_f64 interleaved (per mfma+8fma)",
                        "8 ds_read_b64 + wait (per read)", "v_mul_f64 dependent", "v_mov_b64_dpp newbcast indep", "s_barrier (1 wave)", "v_fmac_f64_dpp indep", "v_fmac_f64_dpp in-place indep", "v_fmac_f64_e32 indep", "v_fmac_f64_dpp dependent", "v_rsq_f64 indep"};
    for (int i = 0; i < 19; ++i)
        printf("%-52s %7.2f cyc each   (clock %.0f MHz)\n", nm[i], (double)h[3 * i] / h[3 * i + 2], 100.0 * h[3 * i] / h[3 * i + 1]);
    return 0;
}
