// scratch: are f16 subnormals produced by float -> f16 conversions and honoured by v_mfma_f32_32x32x16_f16 inputs?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float x) {
    _Float16 h = (_Float16)x;                      // compiler conversion
    unsigned pk;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(pk) : "v"(x));
    out[0] = (float)h;
    out[1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(pk & 0xffff));
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0; b[i] = (_Float16)0; }
    a[0] = __builtin_bit_cast(_Float16, (unsigned short)0x0200);   // 2^-15 (subnormal), exact bit pattern
    b[0] = (_Float16)1024.0f;
    f16v c;
    for (int v = 0; v < 16; ++v) c[v] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    out[2] = c[0];                                  // expect 2^-5 = 0.03125 if honoured
    a[0] = (_Float16)1024.0f;
    b[0] = __builtin_bit_cast(_Float16, (unsigned short)0x0200);
    for (int v = 0; v < 16; ++v) c[v] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    out[3] = c[0];
}
int main() {
    float *d; hipMalloc(&d, 64);
    k<<<1, 64>>>(d, 3.0e-5f);
    float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("cast(3e-5f) -> %g ; v_cvt_pk_f16_f32 -> %g (subnormal kept if ~3e-5)\n", h[0], h[1]);
    printf("mfma 32x32x16: subnormal A * 1024 = %g, 1024 * subnormal B = %g (expect 0.03125 each if honoured)\n", h[2], h[3]);
    return 0;
}
