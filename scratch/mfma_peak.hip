// micro-benchmark: fp32 MFMA 32x32x2 issue rate with/without LDS operand reads (not shipped)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, const float *in)
{
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = in[i];
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0};
    float x = in[threadIdx.x], y = in[threadIdx.x + 256];
    const float *ap = lds + (threadIdx.x & 63) * 6, *bp = lds + 4096 + (threadIdx.x & 31);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 1) {
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                float2 av = *(const float2 *)(ap + t * 12);
                float b0 = bp[t * 64], b1 = bp[t * 64 + 32], b2 = bp[t * 64 + 2048], b3 = bp[t * 64 + 2080];
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b2, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b3, a1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a1, 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    float *out, *in;
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&in, 8192 * 4);
    std::vector<float> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = (float)((i * 2654435761u) >> 8) / 16777216.f - 0.5f;
    hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; ++wgs_per_cu)
        for (int mode = 0; mode < 2; ++mode) {
            int grid = 256 * wgs_per_cu, iters = 400;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<grid, 256, 40000>>>(out, iters, in);
                else k<1><<<grid, 256, 40000>>>(out, iters, in);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)grid * 4 * iters * 108 * 4096.0;
            printf("waves/SIMD %d mode %s: %.3f ms  %.1f TF/s\n", wgs_per_cu, mode ? "lds-operands" : "reg-operands", ms, flop / ms / 1e9);
        }
    return 0;
}
