// micro-benchmark: v_mfma_f32_32x32x16_f16 issue rate, operands in registers vs four ds_read_b128 fragments
// per three MFMAs (the split-f16 kernel's pattern at 64 columns)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, const uint4 *in)
{
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i];
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    union U { uint4 u; f16x8 v; };
    U x, y; x.u = in[threadIdx.x]; y.u = in[threadIdx.x + 256];
    const uint4 *p = lds + (threadIdx.x & 63);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            U ah = x, al = y, bh0 = y, bl0 = x, bh1 = x, bl1 = y;
            if (MODE == 1) { ah.u = p[t * 64]; al.u = p[1024 + t * 64]; bh0.u = p[2048 + t * 128]; bl0.u = p[2048 + t * 128 + 64]; bh1.u = p[t * 128 + 32]; bl1.u = p[3000 + t * 64]; }
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bh0.v, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bl0.v, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bh1.v, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.v, bl1.v, a3, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, bh0.v, a1, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.v, bh1.v, a3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *out; uint4 *in;
    hipMalloc(&out, 8192 * 256 * 4); hipMalloc(&in, 4096 * 16); hipMemset(in, 0, 4096 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 1; w <= 4; ++w)
        for (int mode = 0; mode < 2; ++mode) {
            int grid = 256 * w, iters = 300; float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<grid, 256, 65536>>>(out, iters, in); else k<1><<<grid, 256, 65536>>>(out, iters, in);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            double fl = (double)grid * 4 * iters * 14 * 6 * 32768.0;
            printf("%d WG/CU mode %d (%s): %.3f ms  %.0f TFLOP/s (f16 MFMA)\n", w, mode, mode ? "LDS fragments" : "registers", ms, fl / (ms * 1e-3) / 1e12);
        }
    return 0;
}
