// Micro-benchmark (scratch): the two phases of the LDS-resident Cholesky (linalg_dev.h) in isolation, one workgroup of four
// waves on tiles of random data (the arithmetic is meaningless; the instruction streams are the product's): shader cycles per
// phase per wave for step k = 0 .. nb - 1 of a 128 x 128 matrix with one border vector.  VARIANT switches for experiments.
#include "../../dp_gp_lvm_amd/csrc/linalg_dev.h"
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void phases(long long *out, double *sink, int nb, int reps) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double *dinv = reinterpret_cast<double *>(smem_raw + LA_LDS_HDR), *tiles = dinv + TSZ;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nlow = nb * (nb + 1) / 2;
    for (int e = threadIdx.x; e < (nlow + 1) * TSZ; e += 256) tiles[e] = 1.0 + 1e-3 * (e % 977);
    double *border = tiles + (size_t)nlow * TSZ;
    __syncthreads();
    for (int k = 0; k < nb; ++k) {
        long long tp = 0, tu = 0;
        for (int r = 0; r < reps; ++r) {
            double g[16];
            __syncthreads();
            long long t0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            int bad = potrf_lds_panel<double>(tiles, border, nb, 1, k, wv, lane, g);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            long long t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tp += t1 - t0;
            if (bad == 12345) sink[threadIdx.x] = g[3];
            __syncthreads();
            t0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            potrf_lds_update<double>(tiles, dinv, nb, 1, k, wv, lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            t1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tu += t1 - t0;
        }
        if (lane == 0) {
            out[(k * 4 + wv) * 2] = tp / reps;
            out[(k * 4 + wv) * 2 + 1] = tu / reps;
        }
    }
    sink[threadIdx.x] += tiles[threadIdx.x];
}
int main() {
    const int nb = 8, reps = 20;
    long long *out; double *sink;
    (void)hipMalloc(&out, 8 * 64 * 8); (void)hipMalloc(&sink, 256 * 8);
    const size_t lds = LA_LDS_HDR + 8 * lds_chol_elems(nb, 1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(phases), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int w = 0; w < 2; ++w) phases<<<1, 256, lds>>>(out, sink, nb, reps);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(nb * 8);
    (void)hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    printf("cycles per phase (incl. ~2 x 200 of stamp overhead), waves 0-3\n");
    for (int k = 0; k < nb; ++k) {
        const int m = nb - 1 - k, ntot = m * (m + 1) / 2 + m;
        printf("k %d: panel %6lld %6lld %6lld %6lld   update %6lld %6lld %6lld %6lld  (%d items)\n", k, h[(k * 4) * 2], h[(k * 4 + 1) * 2],
               h[(k * 4 + 2) * 2], h[(k * 4 + 3) * 2], h[(k * 4) * 2 + 1], h[(k * 4 + 1) * 2 + 1], h[(k * 4 + 2) * 2 + 1], h[(k * 4 + 3) * 2 + 1], ntot);
    }
    printf("hip error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
