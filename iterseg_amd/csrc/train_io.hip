// train_io.hip -- the patch sampler that feeds train_unet (the producer side).
//
// Replaces the array work of the reference's get_image_chunks (src/iterseg/train_io.py:224-275): for a candidate
// offset the brightness test  x.mean() / x.max() > min_brightness_prop  over a (z, y, x) crop of the image, and the crop
// itself.  The random draws stay on the host (the caller's RNG, in the reference's order: one randint per axis and
// candidate), so a run with the same draws picks the same crops.  The image stays resident in HBM across candidates;
// a candidate costs one reduction over the crop (sum in float64, maximum) and an accepted one a strided gather.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "isx_common.h"

namespace isx {

// one work-group per candidate: sum (float64) and maximum of vol[z0:z0+cz, y0:y0+cy, x0:x0+cx]
__global__ __launch_bounds__(256) void patch_stats_kernel(const float *__restrict__ vol, int64_t Y, int64_t X,
                                                          const int32_t *__restrict__ starts, int cz, int cy, int cx,
                                                          double *__restrict__ sum_out, float *__restrict__ max_out,
                                                          float *__restrict__ min_out)
{
    __shared__ double ssum[256];
    __shared__ float smax[256], smin[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int64_t z0 = starts[3 * c], y0 = starts[3 * c + 1], x0 = starts[3 * c + 2];
    const int64_t n = (int64_t)cz * cy * cx;
    double s = 0.0;
    float m = -INFINITY, lo = INFINITY;
    for (int64_t i = tid; i < n; i += 256) {
        const int64_t x = i % cx, y = (i / cx) % cy, z = i / ((int64_t)cx * cy);
        const float v = vol[((z0 + z) * Y + (y0 + y)) * X + (x0 + x)];
        s += (double)v;
        m = v > m ? v : m;
        lo = v < lo ? v : lo;
    }
    ssum[tid] = s;
    smax[tid] = m;
    smin[tid] = lo;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            ssum[tid] += ssum[tid + o];
            smax[tid] = smax[tid + o] > smax[tid] ? smax[tid + o] : smax[tid];
            smin[tid] = smin[tid + o] < smin[tid] ? smin[tid + o] : smin[tid];
        }
        __syncthreads();
    }
    if (tid == 0) { sum_out[c] = ssum[0]; max_out[c] = smax[0]; min_out[c] = smin[0]; }
}

// out[c][z][y][x] = vol[z0 + z][y0 + y][x0 + x] for every crop c of the list; ELEM = bytes per element (4: image, labels)
__global__ void patch_gather_kernel(const uint32_t *__restrict__ vol, int64_t Y, int64_t X, const int32_t *__restrict__ starts,
                                    int cz, int cy, int cx, int64_t total, uint32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t n = (int64_t)cz * cy * cx;
    const int64_t c = i / n, r = i % n;
    const int64_t x = r % cx, y = (r / cx) % cy, z = r / ((int64_t)cx * cy);
    const int64_t z0 = starts[3 * c], y0 = starts[3 * c + 1], x0 = starts[3 * c + 2];
    out[i] = vol[((z0 + z) * Y + (y0 + y)) * X + (x0 + x)];
}

static int check_patches(const int64_t dims[3], const int32_t *starts, int n, const int64_t shape[3])
{
    ISX_REQUIRE(dims && starts && shape && n >= 1, "null argument or empty candidate list");
    for (int a = 0; a < 3; ++a) ISX_REQUIRE(shape[a] >= 1 && shape[a] <= dims[a], "crop axis %d: %ld of %ld", a, (long)shape[a], (long)dims[a]);
    for (int c = 0; c < n; ++c)
        for (int a = 0; a < 3; ++a)
            ISX_REQUIRE(starts[3 * c + a] >= 0 && starts[3 * c + a] + shape[a] <= dims[a], "crop %d starts outside the volume on axis %d", c, a);
    return ISX_OK;
}

}  // namespace isx

using namespace isx;

extern "C" int isx_patch_stats(isx_ctx *ctx, const float *vol, const int64_t dims[3], const int32_t *starts, int n,
                               const int64_t shape[3], double *sum_out, float *max_out, float *min_out)
{
    ISX_REQUIRE(ctx && vol && sum_out && max_out, "null argument");
    ISX_TRY(check_patches(dims, starts, n, shape));
    ISX_HIP(hipSetDevice(ctx->device));
    Scratch<int32_t> ds(ctx, (size_t)3 * n);
    Scratch<double> dsum(ctx, (size_t)n);
    Scratch<float> dmax(ctx, (size_t)2 * n);               // maxima, then minima
    if (!ds.ok() || !dsum.ok() || !dmax.ok()) return ISX_ERR_NOMEM;
    ISX_HIP(hipMemcpyAsync(ds.p, starts, (size_t)3 * n * 4, hipMemcpyHostToDevice, ctx->stream));
    patch_stats_kernel<<<(unsigned)n, 256, 0, ctx->stream>>>(vol, dims[1], dims[2], ds.p, (int)shape[0], (int)shape[1], (int)shape[2], dsum.p, dmax.p,
                                                             dmax.p + n);
    ISX_CHECK_LAUNCH();
    ISX_HIP(hipMemcpyAsync(sum_out, dsum.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipMemcpyAsync(max_out, dmax.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (min_out) ISX_HIP(hipMemcpyAsync(min_out, dmax.p + n, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    return ISX_OK;
}

extern "C" int isx_patch_gather(isx_ctx *ctx, const void *vol, const int64_t dims[3], const int32_t *starts, int n,
                                const int64_t shape[3], void *out)
{
    ISX_REQUIRE(ctx && vol && out, "null argument");
    ISX_TRY(check_patches(dims, starts, n, shape));
    ISX_HIP(hipSetDevice(ctx->device));
    Scratch<int32_t> ds(ctx, (size_t)3 * n);
    if (!ds.ok()) return ISX_ERR_NOMEM;
    ISX_HIP(hipMemcpyAsync(ds.p, starts, (size_t)3 * n * 4, hipMemcpyHostToDevice, ctx->stream));
    const int64_t total = (int64_t)n * shape[0] * shape[1] * shape[2];
    patch_gather_kernel<<<(unsigned)((total + 255) / 256), 256, 0, ctx->stream>>>((const uint32_t *)vol, dims[1], dims[2], ds.p, (int)shape[0],
                                                                                  (int)shape[1], (int)shape[2], total, (uint32_t *)out);
    ISX_CHECK_LAUNCH();
    ISX_HIP(hipStreamSynchronize(ctx->stream));           // `starts` was read from the caller's host buffer
    return ISX_OK;
}

namespace isx {

struct AugGeom {
    int64_t in_shape[4], out_shape[4];
    int perm[4];           // out axis k runs over in axis perm[k] (np.transpose)
    int mirror_axis;       // in axis mirrored before the transposition, -1 none
};

// One pass of the reference's augment_images over one 4-D array (src/iterseg/augment.py:8-118), 4-byte elements:
//   intensity (image only, augment_intensity :36-73): v = v * gain + shift  (gain = scale / image.max()), then
//     noise 1: v + g, 2 "speckle": v + v * g, 0 none (g = the caller's field, indexed like the INPUT), clipped to
//     [clip_low, 1] -- clip_low = 0, or -1 where skimage's random_noise saw a negative value (signed range);
//   order (every array, augment_order :76-118): mirror along one axis, then np.transpose(axes).  Arrays without the
//     intensity part are moved as raw 32-bit words (labels and ground truth keep every bit).
__global__ void augment_kernel(const uint32_t *__restrict__ in, AugGeom g, int intensity, float gain, float shift, int noise_mode,
                               const float *__restrict__ noise, float clip_low, uint32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = g.out_shape[0] * g.out_shape[1] * g.out_shape[2] * g.out_shape[3];
    if (i >= n) return;
    int64_t o[4], idx[4], r = i;
#pragma unroll
    for (int k = 3; k >= 0; --k) { o[k] = r % g.out_shape[k]; r /= g.out_shape[k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (g.perm[k] == a) idx[a] = o[k];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
        if (g.mirror_axis == a) idx[a] = g.in_shape[a] - 1 - idx[a];
    const int64_t src = ((idx[0] * g.in_shape[1] + idx[1]) * g.in_shape[2] + idx[2]) * g.in_shape[3] + idx[3];
    uint32_t w = in[src];
    if (intensity) {
        float v = __uint_as_float(w) * gain + shift;
        if (noise_mode == 1) v = v + noise[src];
        else if (noise_mode == 2) v = v + v * noise[src];
        v = v < clip_low ? clip_low : (v > 1.f ? 1.f : v);
        w = __float_as_uint(v);
    }
    out[i] = w;
}

}  // namespace isx

extern "C" int isx_augment_array(isx_ctx *ctx, const void *in, const int64_t shape[4], int intensity, float gain, float shift,
                                 int noise_mode, const float *noise, float clip_low, int mirror_axis, const int32_t *perm, void *out)
{
    ISX_REQUIRE(ctx && in && out && shape, "null argument");
    ISX_REQUIRE(in != out, "augmentation is not in place");
    ISX_REQUIRE(mirror_axis >= -1 && mirror_axis <= 3, "mirror axis: -1 none, 0..3 an axis of the 4-D input");
    ISX_REQUIRE(noise_mode >= 0 && noise_mode <= 2 && (noise_mode == 0 || noise), "noise mode %d needs its draws", noise_mode);
    ISX_REQUIRE(clip_low == 0.f || clip_low == -1.f, "clip_low is 0 or -1 (skimage's unsigned / signed range)");
    AugGeom g;
    unsigned used = 0;
    for (int k = 0; k < 4; ++k) {
        g.perm[k] = perm ? perm[k] : k;
        ISX_REQUIRE(g.perm[k] >= 0 && g.perm[k] <= 3, "perm[%d] = %d", k, g.perm[k]);
        used |= 1u << g.perm[k];
        g.in_shape[k] = shape[k];
    }
    ISX_REQUIRE(used == 15u, "perm is not a permutation of (0, 1, 2, 3)");
    for (int k = 0; k < 4; ++k) g.out_shape[k] = shape[g.perm[k]];
    g.mirror_axis = mirror_axis;
    ISX_HIP(hipSetDevice(ctx->device));
    const int64_t n = shape[0] * shape[1] * shape[2] * shape[3];
    ISX_REQUIRE(n >= 1, "empty array");
    augment_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>((const uint32_t *)in, g, intensity, gain, shift, noise_mode, noise,
                                                                         clip_low, (uint32_t *)out);
    ISX_CHECK_LAUNCH();
    return ISX_OK;
}
