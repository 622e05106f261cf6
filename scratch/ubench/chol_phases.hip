// Micro-benchmark (scratch): the two phases of the LDS-resident Cholesky (linalg_dev.h) in isolation, one workgroup of four
// waves on tiles of random data (the arithmetic is meaningless; the instruction streams are the product's): shader cycles per
// phase per wave for step k = 0 .. nb - 1 of a 128 x 128 matrix with one border vector.  VARIANT switches for experiments.
#include <hip/hip_runtime.h>
__device__ long long g_ph[8][16][4];      // [slot][k][wave] cycles, summed over reps
#define STAMP(i)
#define ACC_BEGIN() const long long t__ = __builtin_amdgcn_s_memtime()
#define ACC_END(i) do { if ((threadIdx.x & 63) == 0) g_ph[i][k][threadIdx.x >> 6] += __builtin_amdgcn_s_memtime() - t__; } while (0)
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
    int *fail = reinterpret_cast<int *>(smem_raw + 64);
    long long tot = 0;
    for (int r = 0; r < reps; ++r) {
        for (int e = threadIdx.x; e < (nlow + 1) * TSZ; e += 256) tiles[e] = ((e % TSZ) % 18 == 0 ? 40.0 : 0.0) + 1e-3 * (e % 977);
        __syncthreads();
        long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        potrf_lds<double, 2>(tiles, dinv, nb, nb + 1, fail);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tot += t1 - t0;
        __syncthreads();
    }
    if (lane == 0) out[wv] = tot / reps;
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
    printf("cycles of potrf_lds (nb = %d, one border vector), waves 0-3: %lld %lld %lld %lld\n", nb, h[0], h[1], h[2], h[3]);
    {
        long long ph[8][16][4];
        (void)hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_ph), sizeof(ph));
        const int runs = 2 * reps;
        for (int k = 0; k < nb; ++k)
            printf("k %d: panel %5lld %5lld %5lld %5lld | column %5lld %5lld %5lld %5lld | rest (incl. next panel) %5lld %5lld %5lld %5lld\n", k,
                   ph[4][k][0] / runs, ph[4][k][1] / runs, ph[4][k][2] / runs, ph[4][k][3] / runs, ph[5][k][0] / runs, ph[5][k][1] / runs,
                   ph[5][k][2] / runs, ph[5][k][3] / runs, ph[6][k][0] / runs, ph[6][k][1] / runs, ph[6][k][2] / runs, ph[6][k][3] / runs);
    }
    printf("hip error: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
