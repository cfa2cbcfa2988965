// LDS read bandwidth per CU for b128 / b64 / b32 lane-contiguous reads (4 waves per CU, then 8)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W>
__global__ __launch_bounds__(512) void k(float *out, int iters)
{
    extern __shared__ uint4 lds[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned acc = 0;
    unsigned base = (wave * 64 + lane) * (W * 4);            // bytes, lane-contiguous
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            unsigned addr = (base + t * 4096 + (i & 1) * 65536) & 131071;
            if (W == 4) { uint4 v; asm volatile("ds_read_b128 %0, %1\n" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); acc += v.x; }
            if (W == 2) { uint2 v; asm volatile("ds_read_b64 %0, %1\n" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); acc += v.x; }
            if (W == 1) { unsigned v; asm volatile("ds_read_b32 %0, %1\n" : "=v"(v) : "v"(addr)); asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); acc += v; }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}
int main()
{
    float *out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads = 256; threads <= 512; threads += 256)
        for (int w = 1; w <= 4; w *= 2) {
            float ms = 0; int iters = 2000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (w == 4) k<4><<<256, threads, 131072>>>(out, iters); else if (w == 2) k<2><<<256, threads, 131072>>>(out, iters); else k<1><<<256, threads, 131072>>>(out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            double bytes = (double)threads * iters * 16 * w * 4;          // per CU
            printf("%d waves/CU, %3d-bit reads: %.3f ms  %.1f B/clk/CU (at 2.4 GHz)\n", threads / 64, w * 32, ms, bytes / (ms * 1e-3) / 2.4e9);
        }
    return 0;
}
