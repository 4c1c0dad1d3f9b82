// Probe: does v_mfma_f32_32x32x16_f16 honour f16 subnormal INPUTS on gfx950?
// (the f16 hi+lo split of the fp32-class conv mode stores lo = f16(x - f16(x)), which is subnormal for |x| < 0.125)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__global__ void probe(const uint16_t *a_bits, const uint16_t *b_bits, float *out) {
    f16x8_t a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = __builtin_bit_cast(_Float16, a_bits[0]);
        b[i] = __builtin_bit_cast(_Float16, b_bits[0]);
    }
    f32x16_t c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
    // plain conversion path for comparison
    if (threadIdx.x == 0) out[1] = (float)__builtin_bit_cast(_Float16, a_bits[0]) * (float)__builtin_bit_cast(_Float16, b_bits[0]) * 16.f;
    if (threadIdx.x == 0) { _Float16 h = (_Float16)(3.1e-6f); out[2] = (float)h; }
}
int main() {
    uint16_t ha = 0x0001 /* 2^-24, smallest subnormal */, hb = 0x4400 /* 4.0 */;
    uint16_t *da, *db; float *dout; float hout[3];
    hipMalloc(&da, 2); hipMalloc(&db, 2); hipMalloc(&dout, 12);
    for (uint16_t bits : {(uint16_t)0x0001, (uint16_t)0x0155, (uint16_t)0x03ff, (uint16_t)0x0400}) {
        ha = bits;
        hipMemcpy(da, &ha, 2, hipMemcpyHostToDevice); hipMemcpy(db, &hb, 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout);
        hipMemcpy(hout, dout, 12, hipMemcpyDeviceToHost);
        printf("a_bits=0x%04x  mfma=%.9g  expected=%.9g  cvt(3.1e-6)=%.9g\n", bits, hout[0], hout[1], hout[2]);
    }
    return 0;
}
