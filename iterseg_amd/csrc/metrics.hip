// metrics.hip -- contingency table of two label volumes on the GPU (the part of the
// reference's segmentation assessment that can be pinned here).
//
// Replaces the counting inside skimage.metrics.variation_of_information(gt, mr) as the reference calls it
// (src/iterseg/metrics.py:105-108; skimage builds a sparse contingency matrix from all voxel pairs on one core) and
// the np.unique object counts (metrics.py:111-119): voxel pairs -> 64-bit keys -> device radix sort -> run heads ->
// unique (gt, mr) pairs with their voxel counts.  The entropies are then a few thousand float64 terms on the host
// (iterseg_amd/metrics.py).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "isx_common.h"
#include "prims.h"

namespace isx {

__global__ void pair_key_kernel(const int32_t *__restrict__ a, const int32_t *__restrict__ b, int64_t n,
                                unsigned long long *__restrict__ keys)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) keys[i] = ((unsigned long long)(uint32_t)a[i] << 32) | (unsigned long long)(uint32_t)b[i];
}

__global__ void run_head_kernel(const unsigned long long *__restrict__ keys, int64_t n, uint32_t *__restrict__ head)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ void run_scatter_kernel(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ head,
                                   const uint32_t *__restrict__ pos, int64_t n, int64_t cap,
                                   unsigned long long *__restrict__ pairs, long long *__restrict__ starts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && head[i] && pos[i] < cap) {
        pairs[pos[i]] = keys[i];
        starts[pos[i]] = i;
    }
}

__global__ void run_count_kernel(const long long *__restrict__ starts, int64_t m, int64_t n, long long *__restrict__ counts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) counts[i] = (i + 1 < m ? starts[i + 1] : n) - starts[i];
}

}  // namespace isx

using namespace isx;

extern "C" int isx_label_contingency(isx_ctx *ctx, const int32_t *gt, const int32_t *mr, int64_t n, uint64_t *pairs,
                                     int64_t *counts, int64_t cap, int64_t *n_pairs)
{
    ISX_REQUIRE(ctx && gt && mr && pairs && counts && n_pairs && n > 0 && cap > 0, "null argument");
    ISX_REQUIRE(n < (int64_t)0xFFFFFFF0ll, "volume too large for 32-bit run positions");
    ISX_HIP(hipSetDevice(ctx->device));
    Scratch<unsigned long long> keys(ctx, (size_t)n), sorted(ctx, (size_t)n);
    Scratch<uint32_t> head(ctx, (size_t)n), pos(ctx, (size_t)n);
    Scratch<long long> starts(ctx, (size_t)cap);
    if (!keys.ok() || !sorted.ok() || !head.ok() || !pos.ok() || !starts.ok()) return ISX_ERR_NOMEM;
    const unsigned grid = cdiv(n, 256);
    pair_key_kernel<<<grid, 256, 0, ctx->stream>>>(gt, mr, n, keys.p);
    ISX_CHECK_LAUNCH();
    ISX_TRY(sort_u64(ctx, (const uint64_t *)keys.p, (uint64_t *)sorted.p, (size_t)n));
    run_head_kernel<<<grid, 256, 0, ctx->stream>>>(sorted.p, n, head.p);
    ISX_CHECK_LAUNCH();
    ISX_TRY(exclusive_scan_u32(ctx, head.p, pos.p, (size_t)n));
    uint32_t last_pos = 0, last_head = 0;
    ISX_HIP(hipMemcpyAsync(&last_pos, pos.p + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipMemcpyAsync(&last_head, head.p + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t m = (int64_t)last_pos + last_head;
    *n_pairs = m;
    if (m > cap) { set_error("%lld distinct label pairs, report buffer holds %lld", (long long)m, (long long)cap); return ISX_ERR_ARG; }
    run_scatter_kernel<<<grid, 256, 0, ctx->stream>>>(sorted.p, head.p, pos.p, n, cap, (unsigned long long *)pairs, starts.p);
    ISX_CHECK_LAUNCH();
    run_count_kernel<<<cdiv(m, 256), 256, 0, ctx->stream>>>(starts.p, m, n, (long long *)counts);
    ISX_CHECK_LAUNCH();
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    return ISX_OK;
}
