// Does a SIMD overlap one wave's MFMAs with another wave's VALU work?  512-thread work-groups (2 waves per SIMD), one per CU.
// mode 0: waves 0-3 MFMA, 4-7 idle; 1: waves 0-3 idle, 4-7 VALU; 2: waves 0-3 MFMA, 4-7 VALU; 3: all MFMA; 4: all VALU
// 5: every wave alternates 84 MFMAs and 312 VALU ops (phases), the two waves of a SIMD in opposite phase
// 6: waves 0-3: 14 x (6 ds_read_b128 + 6 MFMA that consume them), 4-7 idle; 7: all eight waves do that; 8: as 7 plus a barrier and
//    22 ds_write_b128 per iteration (the conv_h2 stage without its VALU work and global loads)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma_block(f32x16 &a0, f32x16 &a1, f32x16 &a2, f32x16 &a3, f16x8 x, f16x8 y)
{
#pragma unroll
    for (int i = 0; i < 21; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, a3, 0, 0, 0);
    }
}
__device__ __forceinline__ void valu_block(float &v0, float &v1, float &v2, float &v3, float c)
{
#pragma unroll
    for (int i = 0; i < 78; ++i) {
        v0 = __builtin_fmaf(v0, c, 1.0f); v1 = __builtin_fmaf(v1, c, 2.0f); v2 = __builtin_fmaf(v2, c, 3.0f); v3 = __builtin_fmaf(v3, c, 4.0f);
    }
}
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_mfma_block(f32x16 &a0, f32x16 &a1, f32x16 &a2, f32x16 &a3, const u32x4v *lds, int lane)
{
    union FR { u32x4v q; f16x8 v; };
#pragma unroll
    for (int p = 0; p < 14; ++p) {
        FR f[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) f[r].q = lds[((p * 6 + r) % 42) * 64 + lane];
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0].v, f[2].v, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0].v, f[3].v, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1].v, f[2].v, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0].v, f[4].v, a3, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[0].v, f[5].v, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[1].v, f[4].v, a1, 0, 0, 0);
    }
}
__global__ __launch_bounds__(512) void k(int mode, int iters, float *out)
{
    __shared__ u32x4v lds[14 * 6 * 64 / 2 + 64];             // 43 KB: two work-groups fit a CU
    for (int i = threadIdx.x; i < 14 * 6 * 64 / 2; i += blockDim.x) lds[i] = u32x4v{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, (uint32_t)i};
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    f16x8 x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(threadIdx.x * 0.001f + i); y[i] = (_Float16)(1.0f + i * 0.01f); }
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    const float c = 0.999f + out[0] * 1e-9f;
    const bool do_m = mode == 3 || mode == 5 || ((mode == 0 || mode == 2) && wave < 4);
    const bool do_v = mode == 4 || mode == 5 || ((mode == 1 || mode == 2) && wave >= 4);
    if (mode >= 6) {
        const int lane = threadIdx.x & 63;
        if (mode == 6 && wave >= 4) { /* idle */ }
        else for (int it = 0; it < iters; ++it) {
            lds_mfma_block(a0, a1, a2, a3, lds, lane);
            if (mode == 8) {
                __syncthreads();
#pragma unroll
                for (int w = 0; w < 22; ++w) lds[((it + w) % 42) * 64 + lane] = u32x4v{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, (uint32_t)w};
                __syncthreads();
            }
        }
    } else if (mode == 5) {
        const bool odd = wave >= 4;
        if (odd) valu_block(v0, v1, v2, v3, c);                  // offset by one phase
        for (int it = 0; it < iters; ++it) { mfma_block(a0, a1, a2, a3, x, y); valu_block(v0, v1, v2, v3, c); }
    } else {
        for (int it = 0; it < iters; ++it) {
            if (do_m) mfma_block(a0, a1, a2, a3, x, y);
            if (do_v) valu_block(v0, v1, v2, v3, c);
        }
    }
    float s = v0 + v1 + v2 + v3;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    if (s == 12345.678f) out[1] = s;
}
int main()
{
    float *d; hipMalloc(&d, 1024); hipMemset(d, 0, 1024);
    const int iters = 2000;
    for (int mode = 0; mode <= 8; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<256, 512>>>(mode, 10, d);
        hipEventRecord(e0);
        k<<<256, 512>>>(mode, iters, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per iteration per wave: 84 MFMAs = 2688 pipe cycles; 312 VALU = 1248 issue cycles
        printf("mode %d: %.3f ms  -> %.0f cycles per iteration at 2.4 GHz\n", mode, ms, ms * 1e-3 * 2.4e9 / iters);
    }
    // the conv_h2 arrangement: two 256-thread work-groups per CU (60 KB of dynamic LDS each on top of the 86 KB static would not
    // fit, so 40 KB: 2 x (86 + ... ) -- the static array is 86 KB; one work-group per CU fits twice only with less: use 512 groups
    // of 256 threads and let the hardware place them; occupancy is reported by the time ratio against modes 7 / 8)
    for (int mode = 7; mode <= 8; ++mode) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<512, 256>>>(mode, 10, d);
        hipEventRecord(e0);
        k<<<512, 256>>>(mode, iters, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d as 512 work-groups of 256 threads: %.3f ms  -> %.0f cycles per iteration at 2.4 GHz\n", mode, ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
