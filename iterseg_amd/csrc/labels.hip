// labels.hip -- training-target generation on the GPU (the producer side of train_unet).
//
// Replaces the reference's get_training_labels (src/iterseg/labels.py:23-68) for: nth_affinity (:87-108),
// get_centreness (:142-205, plain and log), get_centre_offsets (:212-276) and get_semantic_labels (:283-285).  The reference walks regionprops objects in Python and computes one
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

// ---- centreness-log (labels.py:178-205, log=True): t = ln d for d > 0 (0 stays 0), per object t += |min t|, then
// 1 - t / max t.  Doubles of either sign are ordered through a monotone map to uint64.
__device__ __forceinline__ unsigned long long ord_key(double v)
{
    const long long b = __double_as_longlong(v);
    return b < 0 ? ~(unsigned long long)b : (unsigned long long)b | 0x8000000000000000ull;
}
__device__ __forceinline__ double ord_val(unsigned long long k)
{
    return __longlong_as_double((long long)((k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
}
__device__ __forceinline__ double log_dist(double d) { return fabs(d) > 0.0 ? log(d) : d; }

__global__ void centre_log_range_kernel(const int32_t *__restrict__ lab, int64_t Z, int64_t Y, int64_t X,
                                        const unsigned long long *__restrict__ sums, double s0, double s1, double s2,
                                        unsigned long long *__restrict__ tmin, unsigned long long *__restrict__ tmax)
{
    const int64_t N = Z * Y * X;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t l = lab[i];
    if (l <= 0) return;
    const unsigned long long k = ord_key(log_dist(centre_dist(sums + (int64_t)l * 4, i / (X * Y), (i / X) % Y, i % X, s0, s1, s2)));
    atomicMin(tmin + l, k);
    atomicMax(tmax + l, k);
}

__global__ void centre_log_write_kernel(const int32_t *__restrict__ lab, int64_t Z, int64_t Y, int64_t X,
                                        const unsigned long long *__restrict__ sums, double s0, double s1, double s2,
                                        const unsigned long long *__restrict__ tmin, const unsigned long long *__restrict__ tmax,
                                        float *__restrict__ out)
{
    const int64_t N = Z * Y * X;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t l = lab[i];
    float v = 0.0f;
    if (l > 0) {
        const double t = log_dist(centre_dist(sums + (int64_t)l * 4, i / (X * Y), (i / X) % Y, i % X, s0, s1, s2));
        const double shift = fabs(ord_val(tmin[l]));
        const double m = ord_val(tmax[l]) + shift;
        const float f = (float)(1.0 - (t + shift) / m);
        v = (f != f) ? 0.0f : f;                      // nan -> np.nan_to_num -> 0 (single-voxel objects)
    }
    out[i] = v;
}

// ---- centre offsets (labels.py:212-276): the per-object extremes of (centroid - voxel) * scale along an axis are
// attained at the ends of the object's bounding box, so one pass of integer min / max atomics is enough.
__global__ void bbox_kernel(const int32_t *__restrict__ lab, int64_t Z, int64_t Y, int64_t X, int32_t *__restrict__ box /* [L+1][6] */)
{
    const int64_t N = Z * Y * X;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t l = lab[i];
    if (l <= 0) return;
    const int32_t p[3] = {(int32_t)(i / (X * Y)), (int32_t)((i / X) % Y), (int32_t)(i % X)};
    int32_t *b = box + (int64_t)l * 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) { atomicMin(b + a, p[a]); atomicMax(b + 3 + a, p[a]); }
}

__global__ void bbox_init_kernel(int32_t *box, int64_t L1)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < L1 * 6) box[i] = (i % 6) < 3 ? 0x7FFFFFFF : -1;
}

__global__ void offset_write_kernel(const int32_t *__restrict__ lab, int64_t Z, int64_t Y, int64_t X,
                                    const unsigned long long *__restrict__ sums, const int32_t *__restrict__ box, int axis,
                                    double sc, float *__restrict__ out)
{
    const int64_t N = Z * Y * X;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int32_t l = lab[i];
    float v = 0.5f;                                   // background (labels.py:228)
    if (l > 0) {
        const int64_t p[3] = {i / (X * Y), (i / X) % Y, i % X};
        const unsigned long long *s = sums + (int64_t)l * 4;
        const double c = (double)s[axis] / (double)s[3];
        const double d = (c - (double)p[axis]) * sc;
        const double e0 = (c - (double)box[(int64_t)l * 6 + axis]) * sc, e1 = (c - (double)box[(int64_t)l * 6 + 3 + axis]) * sc;
        const double mx = e0 > e1 ? e0 : e1, mn = e0 < e1 ? e0 : e1;
        double r = 0.0;
        if (d > 0.0) r = d / mx;
        else if (d < 0.0) r = -(d / mn);
        r = (r - (-1.0)) / 2.0;
        v = (float)r;
    }
    out[i] = v;
}

// ---- centroid-gauss (labels.py:293-309): impulses at the rounded centroids (np.round: half to even)
__global__ void centroid_scatter_kernel(const unsigned long long *__restrict__ sums, int64_t L1, int64_t Y, int64_t X,
                                        float *__restrict__ out)
{
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l < 1 || l >= L1) return;
    const unsigned long long *s = sums + l * 4;
    if (s[3] == 0ull) return;                         // label value not present
    const double n = (double)s[3];
    const int64_t z = (int64_t)rint((double)s[0] / n), y = (int64_t)rint((double)s[1] / n), x = (int64_t)rint((double)s[2] / n);
    out[(z * Y + y) * X + x] = 1.0f;
}

__global__ void fminmax_kernel(const float *__restrict__ a, int64_t N, unsigned long long *mn, unsigned long long *mx)
{
    unsigned long long lo = ~0ull, hi = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const unsigned long long k = ord_key((double)a[i]);
        lo = k < lo ? k : lo;
        hi = k > hi ? k : hi;
    }
    atomicMin(mn, lo);
    atomicMax(mx, hi);
}

__global__ void shift_scale_kernel(float *__restrict__ a, int64_t N, const unsigned long long *mn, const unsigned long long *mx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double lo = ord_val(*mn), hi = ord_val(*mx);
    a[i] = (float)(((double)a[i] - lo) / (hi - lo));
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
    // per-object statistics shared by the centreness and offset channels (computed once, on first use)
    int64_t L = -1;
    Scratch<unsigned long long> sums(ctx, 0), dmax(ctx, 0), tmin(ctx, 0), tmax(ctx, 0);
    Scratch<int32_t> box(ctx, 0);
    bool have_dmax = false, have_log = false, have_box = false;
    auto need_sums = [&]() -> int {
        if (L >= 0) return ISX_OK;
        Scratch<int32_t> mm(ctx, 2);
        if (!mm.ok()) return ISX_ERR_NOMEM;
        ISX_HIP(hipMemsetAsync(mm.p, 0, 8, ctx->stream));
        label_max_kernel<<<1024, LTPB, 0, ctx->stream>>>(labels, N, mm.p, mm.p + 1);
        ISX_CHECK_LAUNCH();
        int32_t h[2];
        ISX_HIP(hipMemcpyAsync(h, mm.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        ISX_HIP(hipStreamSynchronize(ctx->stream));
        ISX_REQUIRE(h[1] >= 0, "negative labels are not supported (regionprops ignores them upstream)");
        L = h[0];
        sums.resize((size_t)(L + 1) * 4);
        if (!sums.ok()) return ISX_ERR_NOMEM;
        ISX_HIP(hipMemsetAsync(sums.p, 0, (size_t)(L + 1) * 32, ctx->stream));
        centroid_sums_kernel<<<grid, LTPB, 0, ctx->stream>>>(labels, Z, Y, X, sums.p);
        ISX_CHECK_LAUNCH();
        return ISX_OK;
    };
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
            ISX_TRY(need_sums());
            if (!have_dmax) {
                dmax.resize((size_t)(L + 1));
                if (!dmax.ok()) return ISX_ERR_NOMEM;
                ISX_HIP(hipMemsetAsync(dmax.p, 0, (size_t)(L + 1) * 8, ctx->stream));
                centre_dmax_kernel<<<grid, LTPB, 0, ctx->stream>>>(labels, Z, Y, X, sums.p, scale[0], scale[1], scale[2], dmax.p);
                ISX_CHECK_LAUNCH();
                have_dmax = true;
            }
            centre_write_kernel<<<grid, LTPB, 0, ctx->stream>>>(labels, Z, Y, X, sums.p, scale[0], scale[1], scale[2], dmax.p, dst);
            ISX_CHECK_LAUNCH();
        } else if (kind[c] == ISX_TARGET_CENTRENESS_LOG) {
            ISX_TRY(need_sums());
            if (!have_log) {
                tmin.resize((size_t)(L + 1));
                tmax.resize((size_t)(L + 1));
                if (!tmin.ok() || !tmax.ok()) return ISX_ERR_NOMEM;
                ISX_HIP(hipMemsetAsync(tmin.p, 0xFF, (size_t)(L + 1) * 8, ctx->stream));
                ISX_HIP(hipMemsetAsync(tmax.p, 0, (size_t)(L + 1) * 8, ctx->stream));
                centre_log_range_kernel<<<grid, LTPB, 0, ctx->stream>>>(labels, Z, Y, X, sums.p, scale[0], scale[1], scale[2], tmin.p, tmax.p);
                ISX_CHECK_LAUNCH();
                have_log = true;
            }
            centre_log_write_kernel<<<grid, LTPB, 0, ctx->stream>>>(labels, Z, Y, X, sums.p, scale[0], scale[1], scale[2], tmin.p, tmax.p, dst);
            ISX_CHECK_LAUNCH();
        } else if (kind[c] == ISX_TARGET_OFFSET) {
            ISX_REQUIRE(axis[c] >= 0 && axis[c] < 3, "channel %d: offset axis %d", c, axis[c]);
            ISX_TRY(need_sums());
            if (!have_box) {
                box.resize((size_t)(L + 1) * 6);
                if (!box.ok()) return ISX_ERR_NOMEM;
                bbox_init_kernel<<<cdiv((L + 1) * 6, LTPB), LTPB, 0, ctx->stream>>>(box.p, L + 1);
                ISX_CHECK_LAUNCH();
                bbox_kernel<<<grid, LTPB, 0, ctx->stream>>>(labels, Z, Y, X, box.p);
                ISX_CHECK_LAUNCH();
                have_box = true;
            }
            offset_write_kernel<<<grid, LTPB, 0, ctx->stream>>>(labels, Z, Y, X, sums.p, box.p, axis[c], scale[axis[c]], dst);
            ISX_CHECK_LAUNCH();
        } else if (kind[c] == ISX_TARGET_CENTROID_GAUSS) {
            ISX_TRY(need_sums());
            ISX_HIP(hipMemsetAsync(dst, 0, (size_t)N * sizeof(float), ctx->stream));
            centroid_scatter_kernel<<<cdiv(L + 1, LTPB), LTPB, 0, ctx->stream>>>(sums.p, L + 1, Y, X, dst);
            ISX_CHECK_LAUNCH();
            Scratch<float> g(ctx, (size_t)N);
            Scratch<unsigned long long> mm(ctx, 2);
            if (!g.ok() || !mm.ok()) return ISX_ERR_NOMEM;
            const int32_t sig[3] = {0, 1, 1};
            ISX_TRY(isx_ws_gaussian(ctx, dst, g.p, shape, sig));
            ISX_HIP(hipMemsetAsync(mm.p, 0xFF, 8, ctx->stream));
            ISX_HIP(hipMemsetAsync(mm.p + 1, 0, 8, ctx->stream));
            fminmax_kernel<<<1024, LTPB, 0, ctx->stream>>>(g.p, N, mm.p, mm.p + 1);
            ISX_CHECK_LAUNCH();
            ISX_HIP(hipMemcpyAsync(dst, g.p, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
            shift_scale_kernel<<<grid, LTPB, 0, ctx->stream>>>(dst, N, mm.p, mm.p + 1);
            ISX_CHECK_LAUNCH();
            ISX_HIP(hipStreamSynchronize(ctx->stream));
        } else {
            set_error("channel %d: unknown target kind %d", c, kind[c]);
            return ISX_ERR_ARG;
        }
    }
    ISX_HIP(hipStreamSynchronize(ctx->stream));          // the scratch buffers go back to the cache after this
    return ISX_OK;
}
