// scratch: verify the operand / result lane layout assumed for v_mfma_f32_32x32x16_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const float *A, const float *B, float *D) {   // A[32][16], B[16][32] row-major, D[32][32]
    const int l = threadIdx.x, i = l & 31, g = l >> 5;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[i * 16 + 8 * g + j]; b[j] = (_Float16)B[(8 * g + j) * 32 + i]; }
    f32x16 c;
    for (int v = 0; v < 16; ++v) c[v] = 0;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int v = 0; v < 16; ++v) D[(8 * (v / 4) + 4 * g + (v % 4)) * 32 + i] = c[v];
}
int main() {
    float hA[512], hB[512], hD[1024], *dA, *dB, *dD;
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((rand() % 17) - 8) / 8.0f; hB[i] = (float)((rand() % 13) - 6) / 4.0f; }
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    double err = 0;
    for (int i = 0; i < 32; ++i)
        for (int n = 0; n < 32; ++n) {
            double s = 0;
            for (int kk = 0; kk < 16; ++kk) s += (double)hA[i * 16 + kk] * hB[kk * 32 + n];
            err = fmax(err, fabs(s - hD[i * 32 + n]));
        }
    printf("32x32x16 f16 layout check: max abs err %g (0 expected)\n", err);
    return 0;
}
