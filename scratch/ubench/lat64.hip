// Micro-benchmark (scratch, not part of the product): latencies / issue costs of the fp64 primitives on the critical path of
// a Cholesky pivot chain on gfx950: v_fma_f64 (dependent / independent), v_rsq_f64, v_rcp_f64 (+ their accuracy), the f32-seeded
// reciprocal root, v_readlane -> SGPR operand of v_fma_f64, v_mov_b64 DPP row_newbcast, v_mfma_f64_16x16x4 / 4x4x4 dependent
// and back-to-back, and an LDS write -> uniform-address read round trip.  One wave, s_memtime around N repetitions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef double f64x4 __attribute__((ext_vector_type(4)));

#define NREP 64
#define T0() __builtin_amdgcn_sched_barrier(0); long long t0_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)"); __builtin_amdgcn_sched_barrier(0)
#define T1(slot) __builtin_amdgcn_sched_barrier(0); long long t1_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)"); __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) cyc[slot] = t1_ - t0_

__global__ __launch_bounds__(64) void lat_kernel(long long *cyc, double *sink, double seed) {
    __shared__ double lds[64];
    const int lane = threadIdx.x;
    double x = seed + lane * 1e-3, y = 1.0 + lane * 1e-4, z = 0.5;
    // 0: empty
    { T0(); T1(0); }
    // 1: dependent v_fma_f64
    { T0();
#pragma unroll
      for (int i = 0; i < NREP; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z));
      T1(1); }
    // 2: 8 independent v_fma_f64 chains (issue cost)
    { double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
      T0();
#pragma unroll
      for (int i = 0; i < NREP / 8; ++i) {
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(y), "v"(z));
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a1) : "v"(y), "v"(z));
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a2) : "v"(y), "v"(z));
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a3) : "v"(y), "v"(z));
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a4) : "v"(y), "v"(z));
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a5) : "v"(y), "v"(z));
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a6) : "v"(y), "v"(z));
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a7) : "v"(y), "v"(z));
      }
      T1(2);
      x = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; }
    // 3: dependent v_rsq_f64
    { double r = 1.0 + x * 1e-9;
      T0();
#pragma unroll
      for (int i = 0; i < NREP; ++i) asm volatile("v_rsq_f64 %0, %0" : "+v"(r));
      T1(3); x += r; }
    // 4: dependent v_rcp_f64
    { double r = 1.0 + x * 1e-9;
      T0();
#pragma unroll
      for (int i = 0; i < NREP; ++i) asm volatile("v_rcp_f64 %0, %0" : "+v"(r));
      T1(4); x += r; }
    // 5: f32-seeded chain: cvt_f32_f64, rsq_f32, cvt_f64_f32 (dependent)
    { double r = 1.0 + x * 1e-9; float f;
      T0();
#pragma unroll
      for (int i = 0; i < NREP; ++i) {
          asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(r));
          asm volatile("v_rsq_f32 %0, %0" : "+v"(f));
          asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r) : "v"(f));
      }
      T1(5); x += r; }
    // 6: readlane(2) -> fma with SGPR operand, dependent through the vector register
    { double a = 1.0 + x * 1e-9;
      T0();
#pragma unroll
      for (int i = 0; i < NREP; ++i) {
          const unsigned long long u = __builtin_bit_cast(unsigned long long, a);
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 5);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 5);
          const double s = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
          asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(y), "s"(s));
      }
      T1(6); x += a; }
    // 7: v_mov_b64 dpp row_newbcast:5 -> fma, dependent
    { double a = 1.0 + x * 1e-9, b;
      T0();
#pragma unroll
      for (int i = 0; i < NREP; ++i) {
          asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(b) : "v"(a));
          asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a) : "v"(y), "v"(b));
      }
      T1(7); x += a; }
    // 8: dependent mfma_f64_16x16x4 (accumulator chain)
    { f64x4 c = {x, x, x, x};
      T0();
#pragma unroll
      for (int i = 0; i < NREP; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, c, 0, 0, 0);
      T1(8); x += c[0] + c[1] + c[2] + c[3]; }
    // 9: 4 independent mfma_f64_16x16x4 accumulators back to back
    { f64x4 c0 = {x, x, x, x}, c1 = {y, x, x, x}, c2 = {x, y, x, x}, c3 = {x, x, y, x};
      T0();
#pragma unroll
      for (int i = 0; i < NREP / 4; ++i) {
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, z, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(z, y, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(z, z, c3, 0, 0, 0);
      }
      T1(9); x += c0[0] + c1[1] + c2[2] + c3[3]; }
    // 10: dependent mfma_f64_4x4x4 (4 blocks)
    { double c = x;
      T0();
#pragma unroll
      for (int i = 0; i < NREP; ++i) c = __builtin_amdgcn_mfma_f64_4x4x4f64(y, z, c, 0, 0, 0);
      T1(10); x += c; }
    // 11: mfma 16x16x4 result -> used as B operand of the next (operand chain)
    { f64x4 c = {x, x, x, x};
      T0();
#pragma unroll
      for (int i = 0; i < NREP; ++i) { f64x4 zc = {0, 0, 0, 0}; c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, c[0], zc, 0, 0, 0); }
      T1(11); x += c[0] + c[1] + c[2] + c[3]; }
    // 12: LDS write -> wait -> uniform-address read -> wait, dependent
    { double a = 1.0 + x * 1e-9;
      T0();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
          lds[lane] = a;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_s_waitcnt(0xc07f);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          a = lds[5] + 1.0;
      }
      T1(12); x += a; }
    // 13: v_mul_f64 dependent
    { double a = 1.0 + x * 1e-12;
      T0();
#pragma unroll
      for (int i = 0; i < NREP; ++i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(y));
      T1(13); x += a; }
    // 14: readlane x2 only, dependent via v_mov from sgpr
    { double a = 1.0 + x * 1e-9;
      T0();
#pragma unroll
      for (int i = 0; i < NREP; ++i) {
          const unsigned long long u = __builtin_bit_cast(unsigned long long, a);
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 5);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 5);
          a = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
          asm volatile("" : "+v"(a));
      }
      T1(14); x += a; }
    // 15: 16 independent readlane pairs + fma into 16 registers (issue cost of a broadcast row update)
    { double r[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) r[i] = x + i;
      double src[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) src[i] = y + i;
      T0();
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
          const unsigned long long u = __builtin_bit_cast(unsigned long long, src[i]);
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, 5);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 5);
          const double s = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
          asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(r[i]) : "v"(z), "s"(s));
      }
      T1(15);
#pragma unroll
      for (int i = 0; i < 16; ++i) x += r[i]; }
    sink[lane] = x;
}


__global__ __launch_bounds__(64) void calib_kernel(long long *out, float *sink, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 1e-7f;
    long long m0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
    }
    long long m1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (threadIdx.x == 0) { out[0] = m1 - m0; out[1] = r1 - r0; }
    sink[threadIdx.x] = a;
}

__global__ void acc_kernel(const double *in, double *rsq, double *rcp, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a, b;
    asm volatile("v_rsq_f64 %0, %1" : "=v"(a) : "v"(in[i]));
    asm volatile("v_rcp_f64 %0, %1" : "=v"(b) : "v"(in[i]));
    rsq[i] = a;
    rcp[i] = b;
}

int main() {
    long long *cyc; double *sink;
    hipMalloc(&cyc, 32 * sizeof(long long));
    hipMalloc(&sink, 64 * sizeof(double));
    hipMemset(cyc, 0, 32 * sizeof(long long));
    for (int it = 0; it < 3; ++it) lat_kernel<<<1, 64>>>(cyc, sink, 1.0 + it);
    hipDeviceSynchronize();
    long long h[32];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char *name[16] = {"empty", "fma_f64 dep", "fma_f64 indep x8", "rsq_f64 dep", "rcp_f64 dep", "cvt+rsq_f32+cvt dep",
                            "readlane x2 + fma(sgpr) dep", "mov_b64 dpp newbcast + fma dep", "mfma16x16x4 f64 dep acc",
                            "mfma16x16x4 f64 4 indep", "mfma4x4x4 f64 dep", "mfma16 result->B operand dep", "lds wr->rd uniform (x16)",
                            "mul_f64 dep", "readlane x2 dep", "16x(readlane x2 + fma) indep (x64)"};
    const int reps[16] = {1, NREP, NREP, NREP, NREP, NREP, NREP, NREP, NREP, NREP, NREP, NREP, 16, NREP, NREP, 64};
    for (int i = 0; i < 16; ++i)
        printf("%-40s total %6lld  per rep %.1f cyc\n", name[i], h[i], (double)(h[i] - h[0]) / reps[i]);

    {
        long long *co; float *fs; hipMalloc(&co, 16); hipMalloc(&fs, 256);
        calib_kernel<<<1, 64>>>(co, fs, 20000);
        hipDeviceSynchronize();
        long long hc[2]; hipMemcpy(hc, co, 16, hipMemcpyDeviceToHost);
        printf("calibration: %d dependent v_fma_f32: memtime ticks %lld (%.2f per fma), memrealtime ticks %lld -> memtime = %.1f MHz if realtime is 100 MHz\n",
               20000 * 64, hc[0], (double)hc[0] / (20000.0 * 64), hc[1], 100.0 * hc[0] / hc[1]);
    }
    // accuracy of the native seeds
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-20.0, 20.0);
    for (int i = 0; i < n; ++i) x[i] = std::exp2(u(g)) * (1.0 + (g() % 1000003) * 1e-6);
    double *dx, *da, *db;
    hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    acc_kernel<<<n / 256, 256>>>(dx, da, db, n);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        e1 = std::fmax(e1, std::fabs(a[i] * std::sqrt(x[i]) - 1.0));
        e2 = std::fmax(e2, std::fabs(b[i] * x[i] - 1.0));
    }
    printf("max rel err  v_rsq_f64 %.3e   v_rcp_f64 %.3e\n", e1, e2);
    return 0;
}
