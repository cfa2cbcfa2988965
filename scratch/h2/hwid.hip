// prints the HW_ID fields of the first work-groups of a launch that fits two 256-thread work-groups per CU (70 KB LDS each)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(uint32_t *out, unsigned long long *t)
{
    extern __shared__ uint32_t sm[];
    const uint32_t id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));      // XCC_ID
    if (threadIdx.x % 64 == 0) { out[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = id; out[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = xcc; }
    if (threadIdx.x == 0) t[blockIdx.x] = __builtin_amdgcn_s_memtime();
    sm[threadIdx.x] = id;
    for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(100);
    if (sm[(threadIdx.x + 1) % 256] == 12345) out[0] = 0;
}
int main()
{
    const int nb = 2048;
    uint32_t *d; unsigned long long *t;
    hipMalloc(&d, nb * 8 * 4); hipMalloc(&t, nb * 8);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024);
    k<<<nb, 256, 70 * 1024>>>(d, t);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(nb * 8); std::vector<unsigned long long> ht(nb);
    hipMemcpy(h.data(), d, nb * 8 * 4, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), t, nb * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < 1100; b += (b < 40 ? 1 : 37)) {
        printf("block %4d t %llu:", b, ht[b] - ht[0]);
        for (int w = 0; w < 4; ++w) {
            uint32_t id = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1];
            printf("  [wave %u simd %u cu %u sh %u se %u tg %u xcc %u]", id & 15, (id >> 4) & 3, (id >> 8) & 15, (id >> 12) & 1, (id >> 13) & 7, (id >> 16) & 15, x & 15);
        }
        printf("\n");
    }
    return 0;
}
