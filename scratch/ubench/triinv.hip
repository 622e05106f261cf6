// scratch: tri_inverse_dpp (linalg_dev.h) against a host inverse, 8 tiles on waves 0 and 1 as potrf_persist.hip calls it
#include "../../dp_gp_lvm_amd/csrc/linalg_dev.h"
#include <cstdio>
#include <vector>
#include <cmath>
__global__ __launch_bounds__(256) void k(const double *Lg, double *out) {
    __shared__ double tiles[8 * TSZ], linv[8 * TSZ];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, kk = lane >> 4;
    for (int e = t; e < 8 * 256; e += 256) tiles[(e >> 8) * TSZ + ((e >> 4) & 15) * LDT + (e & 15)] = Lg[e];
    __syncthreads();
    if (wv < 2) {
        const int c = 4 * wv + kk;
        tri_inverse_dpp<double>(tiles + c * TSZ, linv + c * TSZ, LDT, lane);
    }
    __syncthreads();
    for (int e = t; e < 8 * 256; e += 256) out[e] = linv[(e >> 8) * TSZ + ((e >> 4) & 15) * LDT + (e & 15)];
}
int main() {
    std::vector<double> L(8 * 256, 0.0), X(8 * 256);
    for (int tI = 0; tI < 8; ++tI)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j <= i; ++j) L[tI * 256 + i * 16 + j] = (i == j) ? 2.0 + 0.1 * i + tI : 0.3 * std::sin(1.0 + i * 7 + j * 3 + tI);
    double *dL, *dX;
    (void)hipMalloc(&dL, L.size() * 8); (void)hipMalloc(&dX, X.size() * 8);
    (void)hipMemcpy(dL, L.data(), L.size() * 8, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        k<<<1, 256>>>(dL, dX);
        (void)hipMemcpy(X.data(), dX, X.size() * 8, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int tI = 0; tI < 8; ++tI) {
            double e = 0;
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double s = 0;
                    for (int q = 0; q < 16; ++q) s += L[tI * 256 + i * 16 + q] * X[tI * 256 + q * 16 + j];
                    e = std::fmax(e, std::fabs(s - (i == j)));
                }
            printf("tile %d: |L X - I| = %.2e  ", tI, e);
            worst = std::fmax(worst, e);
        }
        printf("\n");
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
