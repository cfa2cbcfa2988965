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
                                                          double *__restrict__ sum_out, float *__restrict__ max_out)
{
    __shared__ double ssum[256];
    __shared__ float smax[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int64_t z0 = starts[3 * c], y0 = starts[3 * c + 1], x0 = starts[3 * c + 2];
    const int64_t n = (int64_t)cz * cy * cx;
    double s = 0.0;
    float m = -INFINITY;
    for (int64_t i = tid; i < n; i += 256) {
        const int64_t x = i % cx, y = (i / cx) % cy, z = i / ((int64_t)cx * cy);
        const float v = vol[((z0 + z) * Y + (y0 + y)) * X + (x0 + x)];
        s += (double)v;
        m = v > m ? v : m;
    }
    ssum[tid] = s;
    smax[tid] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { ssum[tid] += ssum[tid + o]; smax[tid] = smax[tid + o] > smax[tid] ? smax[tid + o] : smax[tid]; }
        __syncthreads();
    }
    if (tid == 0) { sum_out[c] = ssum[0]; max_out[c] = smax[0]; }
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
                               const int64_t shape[3], double *sum_out, float *max_out)
{
    ISX_REQUIRE(ctx && vol && sum_out && max_out, "null argument");
    ISX_TRY(check_patches(dims, starts, n, shape));
    ISX_HIP(hipSetDevice(ctx->device));
    Scratch<int32_t> ds(ctx, (size_t)3 * n);
    Scratch<double> dsum(ctx, (size_t)n);
    Scratch<float> dmax(ctx, (size_t)n);
    if (!ds.ok() || !dsum.ok() || !dmax.ok()) return ISX_ERR_NOMEM;
    ISX_HIP(hipMemcpyAsync(ds.p, starts, (size_t)3 * n * 4, hipMemcpyHostToDevice, ctx->stream));
    patch_stats_kernel<<<(unsigned)n, 256, 0, ctx->stream>>>(vol, dims[1], dims[2], ds.p, (int)shape[0], (int)shape[1], (int)shape[2], dsum.p, dmax.p);
    ISX_CHECK_LAUNCH();
    ISX_HIP(hipMemcpyAsync(sum_out, dsum.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipMemcpyAsync(max_out, dmax.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
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

// One pass of the reference's augment_images over one (C, Z, Y, X) array (src/iterseg/augment.py:8-118):
//   intensity (image only, augment_intensity :36-73): v = v * gain + shift  (gain = scale / image.max()), then
//     noise 1 "gaussian": v + g, 2 "speckle": v + v * g, 0 none (g = the caller's N(0, var) draws), clipped to [0, 1];
//   order (every array, augment_order :76-118): mirror along y or x, then the (y, x) transposition.
// out[c][z][yo][xo] with (yo, xo) running over the OUTPUT shape (transposed: (X, Y)).
__global__ void augment_kernel(const float *__restrict__ in, int64_t C, int64_t Z, int64_t Y, int64_t X, int intensity, float gain,
                               float shift, int noise_mode, const float *__restrict__ noise, int mirror_axis, int transpose,
                               float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = C * Z * Y * X;
    if (i >= n) return;
    const int64_t Yo = transpose ? X : Y, Xo = transpose ? Y : X;
    const int64_t xo = i % Xo, yo = (i / Xo) % Yo, z = (i / (Xo * Yo)) % Z, c = i / (Xo * Yo * Z);
    // out = transpose(mirror(in)): undo the transposition, then the mirror
    int64_t y = transpose ? xo : yo, x = transpose ? yo : xo;
    if (mirror_axis == 1) y = Y - 1 - y;
    else if (mirror_axis == 2) x = X - 1 - x;
    const int64_t src = ((c * Z + z) * Y + y) * X + x;
    float v = in[src];
    if (intensity) {
        v = v * gain + shift;
        if (noise_mode == 1) v = v + noise[src];
        else if (noise_mode == 2) v = v + v * noise[src];
        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    }
    out[i] = v;
}

}  // namespace isx

extern "C" int isx_augment_array(isx_ctx *ctx, const float *in, const int64_t shape[4], int intensity, float gain, float shift,
                                 int noise_mode, const float *noise, int mirror_axis, int transpose, float *out)
{
    ISX_REQUIRE(ctx && in && out && shape, "null argument");
    ISX_REQUIRE(in != out, "augmentation is not in place");
    ISX_REQUIRE(mirror_axis == 0 || mirror_axis == 1 || mirror_axis == 2, "mirror axis: 0 none, 1 y, 2 x");
    ISX_REQUIRE(noise_mode >= 0 && noise_mode <= 2 && (noise_mode == 0 || noise), "noise mode %d needs its draws", noise_mode);
    ISX_HIP(hipSetDevice(ctx->device));
    const int64_t n = shape[0] * shape[1] * shape[2] * shape[3];
    ISX_REQUIRE(n >= 1, "empty array");
    augment_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(in, shape[0], shape[1], shape[2], shape[3], intensity, gain, shift,
                                                                         noise_mode, noise, mirror_axis, transpose, out);
    ISX_CHECK_LAUNCH();
    return ISX_OK;
}
