// labels.hip -- training-target generation on the GPU (the producer side of train_unet).
//
// Replaces, for the channel kinds the default training uses, the reference's get_training_labels
// (src/iterseg/labels.py:23-68): nth_affinity (:87-108), get_centreness (:142-205, log / power off) and
// get_semantic_labels (:283-285).  The reference walks regionprops objects in Python and computes one
// np.linalg.norm per voxel; here: one pass of integer atomics for the per-object coordinate sums, one pass for the
// per-object maximum distance, one pass writing 1 - d / d_max.  Distances are float64 like upstream.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "isx_common.h"

#pragma clang fp contract(off)

namespace isx {

static constexpr int LTPB = 256;

// out[c][v] = (labels[v] != labels[v shifted by -n along axis, reflected at the start]) ? 1 : 0
__global__ void aff_target_kernel(const int32_t *__restrict__ lab, int64_t Z, int64_t Y, int64_t X, int axis, int n,
                                  float *__restrict__ out)
{
    const int64_t N = Z * Y * X;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int64_t x = i % X, y = (i / X) % Y, z = i / (X * Y);
    int64_t p[3] = {z, y, x};
    int64_t q = p[axis] - n;
    if (q < 0) q = -q;                              // np.pad(mode='reflect')
    p[axis] = q;
    const int32_t o = lab[(p[0] * Y + p[1]) * X + p[2]];
    out[i] = (lab[i] != o) ? 1.0f : 0.0f;
}

__global__ void mask_target_kernel(const int32_t *__restrict__ lab, int64_t N, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[i] = lab[i] > 1 ? 1.0f : 0.0f;   // "> 1" as upstream (labels.py:284)
}

__global__ void label_max_kernel(const int32_t *__restrict__ lab, int64_t N, int32_t *mx, int32_t *mn)
{
    int32_t hi = 0, lo = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t v = lab[i];
        hi = v > hi ? v : hi;
        lo = v < lo ? v : lo;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const int32_t h2 = __shfl_down(hi, o), l2 = __shfl_down(lo, o);
        hi = h2 > hi ? h2 : hi;
        lo = l2 < lo ? l2 : lo;
    }
    if ((threadIdx.x & 63) == 0) { atomicMax(mx, hi); atomicMin(mn, lo); }
}

// per object: sums of z, y, x and the voxel count.  Lanes of a wave that hold the same label are combined first
// (objects are spatially compact: one atomic per run instead of one per voxel).
__global__ void centroid_sums_kernel(const int32_t *__restrict__ lab, int64_t Z, int64_t Y, int64_t X,
                                     unsigned long long *__restrict__ sums /* [L+1][4] */)
{
    const int64_t N = Z * Y * X;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t l = i < N ? lab[i] : 0;
    const int lane = threadIdx.x & 63;
    bool todo = l > 0;
    unsigned long long z = 0, y = 0, x = 0;
    if (todo) { x = (unsigned long long)(i % X); y = (unsigned long long)((i / X) % Y); z = (unsigned long long)(i / (X * Y)); }
    while (__any(todo)) {
        const int leader = __ffsll((long long)__ballot(todo)) - 1;
        const int32_t ll = __shfl(l, leader);
        const bool same = todo && l == ll;
        unsigned long long sz = same ? z : 0ull, sy = same ? y : 0ull, sx = same ? x : 0ull, sc = same ? 1ull : 0ull;
        for (int o = 32; o > 0; o >>= 1) {
            sz += __shfl_xor(sz, o); sy += __shfl_xor(sy, o); sx += __shfl_xor(sx, o); sc += __shfl_xor(sc, o);
        }
        if (lane == leader) {
            unsigned long long *s = sums + (int64_t)ll * 4;
            atomicAdd(s + 0, sz); atomicAdd(s + 1, sy); atomicAdd(s + 2, sx); atomicAdd(s + 3, sc);
        }
        if (same) todo = false;
    }
}

__device__ __forceinline__ double centre_dist(const unsigned long long *s, int64_t z, int64_t y, int64_t x, double s0,
                                              double s1, double s2)
{
    const double n = (double)s[3];
    const double dz = ((double)s[0] / n - (double)z) * s0, dy = ((double)s[1] / n - (double)y) * s1,
                 dx = ((double)s[2] / n - (double)x) * s2;
    return sqrt(dz * dz + dy * dy + dx * dx);
}

__global__ void centre_dmax_kernel(const int32_t *__restrict__ lab, int64_t Z, int64_t Y, int64_t X,
                                   const unsigned long long *__restrict__ sums, double s0, double s1, double s2,
                                   unsigned long long *__restrict__ dmax /* [L+1] bit pattern of a non-negative double */)
{
    const int64_t N = Z * Y * X;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t l = lab[i];
    if (l <= 0) return;
    const double d = centre_dist(sums + (int64_t)l * 4, i / (X * Y), (i / X) % Y, i % X, s0, s1, s2);
    atomicMax(dmax + l, (unsigned long long)__double_as_longlong(d));      // non-negative doubles order like their bits
}

__global__ void centre_write_kernel(const int32_t *__restrict__ lab, int64_t Z, int64_t Y, int64_t X,
                                    const unsigned long long *__restrict__ sums, double s0, double s1, double s2,
                                    const unsigned long long *__restrict__ dmax, float *__restrict__ out)
{
    const int64_t N = Z * Y * X;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t l = lab[i];
    float v = 0.0f;
    if (l > 0) {
        const double d = centre_dist(sums + (int64_t)l * 4, i / (X * Y), (i / X) % Y, i % X, s0, s1, s2);
        const double m = __longlong_as_double((long long)dmax[l]);
        const float f = (float)(1.0 - d / m);
        v = (f != f) ? 0.0f : f;                      // 0/0 for single-voxel objects -> nan -> np.nan_to_num -> 0
    }
    out[i] = v;
}

}  // namespace isx

using namespace isx;

extern "C" int isx_training_labels(isx_ctx *ctx, const int32_t *labels, const int64_t shape[3], int n_channels,
                                   const int32_t *kind, const int32_t *axis, const int32_t *n, const double scale[3],
                                   float *out)
{
    ISX_REQUIRE(ctx && labels && shape && kind && axis && n && scale && out && n_channels >= 1, "null argument");
    ISX_HIP(hipSetDevice(ctx->device));
    const int64_t Z = shape[0], Y = shape[1], X = shape[2], N = Z * Y * X;
    ISX_REQUIRE(Z > 0 && Y > 0 && X > 0, "empty volume");
    const unsigned grid = cdiv(N, LTPB);
    bool centre_done = false;
    int centre_channel = -1;
    for (int c = 0; c < n_channels; ++c) {
        float *dst = out + (int64_t)c * N;
        if (kind[c] == ISX_TARGET_AFFINITY) {
            ISX_REQUIRE(axis[c] >= 0 && axis[c] < 3 && n[c] >= 1 && n[c] < shape[axis[c]], "channel %d: affinity axis %d, n %d", c,
                        axis[c], n[c]);
            aff_target_kernel<<<grid, LTPB, 0, ctx->stream>>>(labels, Z, Y, X, axis[c], n[c], dst);
            ISX_CHECK_LAUNCH();
        } else if (kind[c] == ISX_TARGET_MASK) {
            mask_target_kernel<<<grid, LTPB, 0, ctx->stream>>>(labels, N, dst);
            ISX_CHECK_LAUNCH();
        } else if (kind[c] == ISX_TARGET_CENTRENESS) {
            if (centre_done) {           // a second centreness channel is a copy of the first
                ISX_HIP(hipMemcpyAsync(dst, out + (int64_t)centre_channel * N, N * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
                continue;
            }
            Scratch<int32_t> mm(ctx, 2);
            if (!mm.ok()) return ISX_ERR_NOMEM;
            ISX_HIP(hipMemsetAsync(mm.p, 0, 8, ctx->stream));
            label_max_kernel<<<1024, LTPB, 0, ctx->stream>>>(labels, N, mm.p, mm.p + 1);
            ISX_CHECK_LAUNCH();
            int32_t h[2];
            ISX_HIP(hipMemcpyAsync(h, mm.p, 8, hipMemcpyDeviceToHost, ctx->stream));
            ISX_HIP(hipStreamSynchronize(ctx->stream));
            ISX_REQUIRE(h[1] >= 0, "negative labels are not supported (regionprops ignores them upstream)");
            const int64_t L = h[0];
            Scratch<unsigned long long> sums(ctx, (size_t)(L + 1) * 4), dmax(ctx, (size_t)(L + 1));
            if (!sums.ok() || !dmax.ok()) return ISX_ERR_NOMEM;
            ISX_HIP(hipMemsetAsync(sums.p, 0, (size_t)(L + 1) * 32, ctx->stream));
            ISX_HIP(hipMemsetAsync(dmax.p, 0, (size_t)(L + 1) * 8, ctx->stream));
            centroid_sums_kernel<<<grid, LTPB, 0, ctx->stream>>>(labels, Z, Y, X, sums.p);
            ISX_CHECK_LAUNCH();
            centre_dmax_kernel<<<grid, LTPB, 0, ctx->stream>>>(labels, Z, Y, X, sums.p, scale[0], scale[1], scale[2], dmax.p);
            ISX_CHECK_LAUNCH();
            centre_write_kernel<<<grid, LTPB, 0, ctx->stream>>>(labels, Z, Y, X, sums.p, scale[0], scale[1], scale[2], dmax.p, dst);
            ISX_CHECK_LAUNCH();
            ISX_HIP(hipStreamSynchronize(ctx->stream));          // scratch goes back to the cache after this
            centre_done = true;
            centre_channel = c;
        } else {
            set_error("channel %d: unknown target kind %d", c, kind[c]);
            return ISX_ERR_ARG;
        }
    }
    return ISX_OK;
}
