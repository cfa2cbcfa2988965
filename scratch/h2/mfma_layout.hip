// checks the operand layout assumed for v_mfma_f32_32x32x16_f16 on gfx950:
//   A: lane l holds row l%32, k = 8*(l/32) + j (j = 0..7);  B: lane l holds column l%32, k = 8*(l/32) + j
//   D: lane l, reg i -> row (i&3) + 8*(i>>2) + 4*(l/32), column l%32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float *A, const float *B, float *D)   // A[32][16], B[16][32] row-major
{
    int l = threadIdx.x, r = l & 31, g = l >> 5;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[r * 16 + 8 * g + j]; b[j] = (_Float16)B[(8 * g + j) * 32 + r]; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * g) * 32 + r] = c[i];
}
int main()
{
    float hA[512], hB[512], hD[1024], ref[1024];
    for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6) / 4.f; hB[i] = (float)((i * 5) % 11 - 5) / 8.f; }
    // subnormal probe: one operand 2^-20 (fp16 subnormal), partner 2^10
    hA[0] = 9.5367431640625e-07f; hB[0] = 1024.f;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { double s = 0; for (int kk = 0; kk < 16; ++kk) s += (double)hA[m * 16 + kk] * hB[kk * 32 + n]; ref[m * 32 + n] = (float)s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    double worst = 0; for (int i = 0; i < 1024; ++i) worst = fmax(worst, fabs(hD[i] - ref[i]));
    printf("max |D - ref| = %g  (D[0][0] = %.10g, ref %.10g: subnormal operand %s)\n", worst, hD[0], ref[0],
           fabs(hD[0] - ref[0]) < 1e-6 ? "kept" : "FLUSHED");
    return worst < 2e-3 ? 0 : 1;
}
