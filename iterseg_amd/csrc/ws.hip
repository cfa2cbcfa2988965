// ws.hip -- watershed post-processor of libitersegx.so (gfx950).
//
// Replaces src/iterseg/watershed.py of the reference: segment_output_image (:165-223),
// _get_mask (:226-229), _get_centroids (:232-236), _remove_unwanted_objects (:239-251),
// _prep_data (:38-63) and raveled_affinity_watershed (:95-159).  Results are bit-exact
// with the reference for the integer/boolean outputs and for the float32 intermediates,
// so every floating-point expression below follows the rounding sequence of the code it
// replaces (double accumulation per separable pass, float32 histogram index, double
// Otsu statistics).  FP contraction is therefore disabled for this translation unit.
//
// All of it is HBM-bound byte/integer work: kernels are one thread per voxel with
// x-contiguous (coalesced) accesses; reductions use wave64 shuffles + one atomic per wave.
#pragma clang fp contract(off)

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "isx_common.h"
#include "prims.h"

namespace isx {

static constexpr int TPB = 256;
static constexpr uint32_t NO_PARENT = 0xFFFFFFFFu;

// Gaussian taps: scipy.ndimage._gaussian_kernel1d(sigma, 0, int(4*sigma+0.5)) as produced
// by the numpy 1.26.4 the golden vectors were captured with (np.exp is not bit-stable
// across numpy releases, so these are data).  [0] = centre.
__constant__ double GK_SIGMA2[9] = {
    0x1.98862a07ae7b4p-3, 0x1.68856f9ab1982p-3, 0x1.ef9093fc46e5ap-4, 0x1.0941b71ceef37p-4,
    0x1.ba4d4125ffd29p-6, 0x1.1f30504e20207p-7, 0x1.227362b5fc92ep-9, 0x1.c98b8c5d0dda5p-12,
    0x1.18aad19e4159bp-14};
__constant__ double GK_SIGMA1[5] = {
    0x1.9884a307594fbp-2, 0x1.ef8eb9ad499bap-3, 0x1.ba4b99d1799a9p-5, 0x1.22724cb7eb26ap-8,
    0x1.18a9c4fd536c6p-13};

// Order-preserving float32 -> uint32 (total order of non-NaN floats).
__device__ __forceinline__ uint32_t ford(float f)
{
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float funord(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

// Fold a thread's ordered-uint minimum / maximum into the global pair mm[0] / mm[1] (initialised to 0xFFFFFFFF / 0): wave reduction,
// the block's waves through LDS, then ONE thread touches the global pair -- and issues an atomic only when the block still improves on
// what is there (a relaxed read first: hundreds of thousands of blocks end on the same two words, and same-address atomics serialise in
// L2; a stale read only costs a redundant atomic; per wave instead of per block the reads alone cost aff_pad_kernel 0.35 ms).  For
// producers that already hold every value of a tensor in registers: the separate minmax_kernel pass over it (0.4 ms per 0.5 GB at
// 512^3) goes away.  Every thread of the block must call it (it has a barrier); blocks of at most 16 waves.
__device__ __forceinline__ void mm_fold2(uint32_t lo, uint32_t hi, uint32_t *pmin, uint32_t *pmax)
{
    __shared__ uint32_t mm_s[32];
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t l2 = __shfl_down(lo, o), h2 = __shfl_down(hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) { mm_s[2 * wave] = lo; mm_s[2 * wave + 1] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; ++w) { const uint32_t l2 = mm_s[2 * w], h2 = mm_s[2 * w + 1]; lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; }
        if (pmin && lo < __hip_atomic_load(pmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(pmin, lo);
        if (pmax && hi > __hip_atomic_load(pmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(pmax, hi);
    }
}
__device__ __forceinline__ void mm_fold(uint32_t lo, uint32_t hi, uint32_t *mm) { mm_fold2(lo, hi, &mm[0], &mm[1]); }

// ------------------------------------------------------------------------------------
// W1/W2: separable Gaussian pass (scipy NI_Correlate1D symmetric branch, 'nearest').
//   tmp = c*w0; for k = R..1: tmp += (in[-k] + in[+k]) * w[k];   all in double; store f32.
// ------------------------------------------------------------------------------------
template <int R>
__global__ void gauss_pass_kernel(const float *__restrict__ in, float *__restrict__ out,
                                  int64_t Z, int64_t Y, int64_t X, int axis)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t n = Z * Y * X;
    if (i >= n) return;
    const double *w = (R == 8) ? GK_SIGMA2 : GK_SIGMA1;
    int64_t x = i % X, y = (i / X) % Y, z = i / (X * Y);
    int64_t len, pos, stride;
    if (axis == 0) { len = Z; pos = z; stride = Y * X; }
    else if (axis == 1) { len = Y; pos = y; stride = X; }
    else { len = X; pos = x; stride = 1; }
    const float *base = in + (i - pos * stride);
    double tmp = (double)base[pos * stride] * w[0];
#pragma unroll
    for (int k = R; k >= 1; --k) {
        int64_t lo = pos - k < 0 ? 0 : pos - k;
        int64_t hi = pos + k > len - 1 ? len - 1 : pos + k;
        double s = (double)base[lo * stride] + (double)base[hi * stride];
        double p = s * w[k];
        tmp = tmp + p;
    }
    out[i] = (float)tmp;
}

// The z pass as a march: a thread owns one (y, x) column of a z segment and keeps the 2 R + 1 planes an output needs (+ 3 planes of
// prefetch) in a register ring, so every input plane is read ONCE per segment, coalesced along x, instead of 2 R + 1 times through the
// caches by a thread per voxel (17 x 0.5 GB of L2 traffic at 512^3: 0.95 ms for a pass whose compulsory bytes stream in 0.25).  Same
// operands in the same order as gauss_pass_kernel -- bit-identical.
template <int R>
__global__ __launch_bounds__(256) void gauss_z_kernel(const float *__restrict__ in, float *__restrict__ out, int Z, int Y, int X, int ZS, int nxb)
{
#pragma clang fp contract(off)
    constexpr int P = 3, NSL = 2 * R + 1 + P;
    const double *w = (R == 8) ? GK_SIGMA2 : GK_SIGMA1;
    const int xb = blockIdx.x % nxb, y = (blockIdx.x / nxb) % Y, seg = blockIdx.x / (nxb * Y);
    const int x = xb * 256 + (int)threadIdx.x;
    if (x >= X) return;
    const int zs = seg * ZS, ze = zs + ZS < Z ? zs + ZS : Z;
    const int64_t plane = (int64_t)Y * X;
    const float *col = in + (int64_t)y * X + x;
    float *ocol = out + (int64_t)y * X + x;
    auto ld = [&](int p) __attribute__((always_inline)) -> double {
        const int q = p < 0 ? 0 : (p > Z - 1 ? Z - 1 : p);                 // mode='nearest'
        return (double)col[(int64_t)q * plane];
    };
    double ring[NSL];                                   // plane p sits in slot (p - (zs - R)) % NSL
#pragma unroll
    for (int i = 0; i < NSL - 1; ++i) ring[i] = ld(zs - R + i);
    for (int t0 = 0; zs + t0 < ze; t0 += NSL) {
#pragma unroll
        for (int j = 0; j < NSL; ++j) {
            const int z = zs + t0 + j;
            if (z < ze) {                                                  // (uniform)
                ring[(j + 2 * R + P) % NSL] = ld(z + R + P);               // the slot of plane z - 1, dead since the last output
                double tmp = ring[(j + R) % NSL] * w[0];
#pragma unroll
                for (int k = R; k >= 1; --k) {
                    const double s2 = ring[(j + R - k) % NSL] + ring[(j + R + k) % NSL];
                    const double p2 = s2 * w[k];
                    tmp = tmp + p2;
                }
                ocol[(int64_t)z * plane] = (float)tmp;
            }
        }
    }
}

// The y pass and the x pass of one plane in ONE kernel (scipy applies the axes in order, every pass stores float32: the
// intermediate lives in LDS as float32, so the result is the two-pass one bit for bit).  A block owns 16 x 64 outputs of a plane:
// the (16 + 2R) x (64 + 2R) input tile is loaded once with clamped coordinates ('nearest'), the y pass runs on 16 x (64 + 2R)
// positions -- the halo columns are the y pass of the clamped columns, exactly what the x pass of the separate kernels reads at
// an edge --, the x pass on 16 x 64.  One read and one write of the volume instead of two of each, and 2 x 17 LDS reads per
// output instead of 2 x 17 cached global loads with 64-bit index arithmetic.  Same operation order as gauss_pass_kernel, products
// and sums not contracted (scipy's C loop has no fused multiply-add).
template <int R>
__global__ __launch_bounds__(256) void gauss_yx_kernel(const float *__restrict__ in, float *__restrict__ out, int Z, int Y, int X, int nty, int ntx, uint32_t *mm)
{
#pragma clang fp contract(off)
    constexpr int TY = 16, TX = 64, IW = TX + 2 * R, IH = TY + 2 * R;
    __shared__ float tin[IH * IW];
    __shared__ float mid[TY * IW];
    const double *w = (R == 8) ? GK_SIGMA2 : GK_SIGMA1;
    const int tile = blockIdx.x;
    const int xt = tile % ntx, yt = (tile / ntx) % nty, z = tile / (ntx * nty);
    const int y0 = yt * TY, x0 = xt * TX;
    const float *plane = in + (int64_t)z * Y * X;
    for (int i = threadIdx.x; i < IH * IW; i += 256) {
        const int r = i / IW, c = i % IW;
        int yy = y0 + r - R, xx = x0 + c - R;
        yy = yy < 0 ? 0 : (yy > Y - 1 ? Y - 1 : yy);
        xx = xx < 0 ? 0 : (xx > X - 1 ? X - 1 : xx);
        tin[i] = plane[(int64_t)yy * X + xx];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TY * IW; i += 256) {
        const int r = i / IW, c = i % IW;
        const float *col = tin + (r + R) * IW + c;
        double tmp = (double)col[0] * w[0];
#pragma unroll
        for (int k = R; k >= 1; --k) {
            const double s2 = (double)col[-k * IW] + (double)col[k * IW];
            const double p2 = s2 * w[k];
            tmp = tmp + p2;
        }
        mid[i] = (float)tmp;
    }
    __syncthreads();
    float *oplane = out + (int64_t)z * Y * X;
    uint32_t mlo = 0xFFFFFFFFu, mhi = 0u;
    for (int i = threadIdx.x; i < TY * TX; i += 256) {
        const int r = i / TX, c = i % TX;
        const int y = y0 + r, x = x0 + c;
        const float *row = mid + r * IW + c + R;
        double tmp = (double)row[0] * w[0];
#pragma unroll
        for (int k = R; k >= 1; --k) {
            const double s2 = (double)row[-k] + (double)row[k];
            const double p2 = s2 * w[k];
            tmp = tmp + p2;
        }
        if (y < Y && x < X) {
            oplane[(int64_t)y * X + x] = (float)tmp;
            const uint32_t u = ford((float)tmp);
            mlo = u < mlo ? u : mlo;
            mhi = u > mhi ? u : mhi;
        }
    }
    if (mm) mm_fold(mlo, mhi, mm);                     // (otsu_impl's min / max of the smoothed image, without its own pass)
}

// mm (optional, device, initialised to {0xFFFFFFFF, 0}): receives the ordered-uint min / max of `out` when the last pass is the
// fused y+x kernel; *mm_done says whether it did
static int gaussian_impl(isx_ctx *ctx, const float *img, float *out, const int64_t shape[3],
                         const int32_t sigma[3], float *tmpbuf, uint32_t *mm = nullptr, bool *mm_done = nullptr)
{
    if (mm_done) *mm_done = false;
    int64_t n = shape[0] * shape[1] * shape[2];
    int npass = 0;
    for (int a = 0; a < 3; ++a) {
        ISX_REQUIRE(sigma[a] >= 0 && sigma[a] <= 2, "sigma must be 0, 1 or 2 (reference uses 2 and (0,1,1))");
        npass += sigma[a] > 0;
    }
    if (npass == 0) {
        ISX_HIP(hipMemcpyAsync(out, img, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        return ISX_OK;
    }
    // ping-pong so that the last pass lands in `out`
    const float *src = img;
    int done = 0;
    const bool fuse_yx = sigma[1] > 0 && sigma[1] == sigma[2] && ctx->ftune.gauss_fused &&
                         shape[0] * ((shape[1] + 15) / 16) * ((shape[2] + 63) / 64) < (int64_t)0x7FFFFFFF;
    if (fuse_yx) npass -= 1;
    for (int a = 0; a < 3; ++a) {
        if (sigma[a] == 0) continue;
        if (fuse_yx && a == 1) {
            const int nty = (int)((shape[1] + 15) / 16), ntx = (int)((shape[2] + 63) / 64);
            const unsigned grid = (unsigned)(shape[0] * nty * ntx);
            float *dst = out;                          // the fused pass is the last one
            if (sigma[a] == 2) gauss_yx_kernel<8><<<grid, 256, 0, ctx->stream>>>(src, dst, (int)shape[0], (int)shape[1], (int)shape[2], nty, ntx, mm);
            else gauss_yx_kernel<4><<<grid, 256, 0, ctx->stream>>>(src, dst, (int)shape[0], (int)shape[1], (int)shape[2], nty, ntx, mm);
            if (mm && mm_done) *mm_done = true;
            ISX_CHECK_LAUNCH();
            break;
        }
        int remaining = npass - done - 1;
        float *dst = (remaining % 2 == 0) ? out : tmpbuf;
        const int64_t nxb = (shape[2] + 255) / 256, zseg = 64, nseg = (shape[0] + zseg - 1) / zseg;
        if (a == 0 && ctx->ftune.gauss_fused && shape[0] < (1 << 24) && shape[1] < (1 << 24) && shape[2] < (1 << 24) &&
            nxb * shape[1] * nseg < (int64_t)0x7FFFFFFF) {
            if (sigma[a] == 2) gauss_z_kernel<8><<<(unsigned)(nxb * shape[1] * nseg), 256, 0, ctx->stream>>>(src, dst, (int)shape[0], (int)shape[1], (int)shape[2], (int)zseg, (int)nxb);
            else gauss_z_kernel<4><<<(unsigned)(nxb * shape[1] * nseg), 256, 0, ctx->stream>>>(src, dst, (int)shape[0], (int)shape[1], (int)shape[2], (int)zseg, (int)nxb);
        } else
        if (sigma[a] == 2)
            gauss_pass_kernel<8><<<cdiv(n, TPB), TPB, 0, ctx->stream>>>(src, dst, shape[0], shape[1], shape[2], a);
        else
            gauss_pass_kernel<4><<<cdiv(n, TPB), TPB, 0, ctx->stream>>>(src, dst, shape[0], shape[1], shape[2], a);
        ISX_CHECK_LAUNCH();
        src = dst;
        ++done;
    }
    return ISX_OK;
}

// ------------------------------------------------------------------------------------
// Otsu: min/max -> float32 edges (float64 linspace cast) -> 256-bin histogram with numpy's
// index recipe -> double between-class variance.  skimage thresholding.py:326-350,
// numpy/lib/histograms.py uniform-bin path.
// ------------------------------------------------------------------------------------
__global__ void minmax_kernel(const float *__restrict__ a, int64_t n, uint32_t *mm)
{
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    // 16 bytes per lane and load (a thread per float ran at 1.1 TB/s: too few bytes in flight); the head up to the first
    // 16-byte boundary and the tail are taken one by one
    const int64_t head = (int64_t)((16 - ((uintptr_t)a & 15)) & 15) / 4;
    const int64_t h = head < n ? head : n, n4 = (n - h) / 4;
    const float4 *a4 = (const float4 *)(a + h);
    const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid0; i < n4; i += stride) {
        const float4 v = a4[i];
        const uint32_t u0 = ford(v.x), u1 = ford(v.y), u2 = ford(v.z), u3 = ford(v.w);
        const uint32_t l01 = u0 < u1 ? u0 : u1, l23 = u2 < u3 ? u2 : u3, h01 = u0 > u1 ? u0 : u1, h23 = u2 > u3 ? u2 : u3;
        const uint32_t l4 = l01 < l23 ? l01 : l23, h4 = h01 > h23 ? h01 : h23;
        lo = l4 < lo ? l4 : lo;
        hi = h4 > hi ? h4 : hi;
    }
    for (int64_t i = tid0; i < h + (n - h - 4 * n4); i += stride) {
        const uint32_t u = ford(a[i < h ? i : 4 * n4 + i]);          // i >= h: h + 4 n4 + (i - h)
        lo = u < lo ? u : lo;
        hi = u > hi ? u : hi;
    }
    mm_fold(lo, hi, mm);          // (one atomic pair per block, and only when it improves: the per-wave pairs -- 16 k of them on one cache
                                  //  line -- were a third of this kernel's time)
}

// strided channel max for W0 (3 channels at once): cm[c] = ordered max of fm[c*n ...]
// (cm[c] = max, cm[4 + c] = min: the minimum costs nothing in the same pass, and with both the value range of the padded, normalised
//  affinities follows without reading them again -- aff_range_kernel)
__global__ void chanmax_kernel(const float *__restrict__ fm, int64_t n, uint32_t *cm)
{
    int c = blockIdx.y;
    const float *a = fm + (int64_t)c * n;
    uint32_t hi = 0u, lo = 0xFFFFFFFFu;
    const int64_t head = (int64_t)((16 - ((uintptr_t)a & 15)) & 15) / 4;          // (as minmax_kernel: 16-byte loads)
    const int64_t h = head < n ? head : n, n4 = (n - h) / 4;
    const float4 *a4 = (const float4 *)(a + h);
    const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    // (two 16-byte loads in flight per lane: with the minimum beside the maximum one load per iteration ran at 2.4 TB/s)
    for (int64_t i = tid0; i < n4; i += 2 * stride) {
        const float4 v = a4[i], w = a4[i + stride < n4 ? i + stride : i];
        const uint32_t u0 = ford(v.x), u1 = ford(v.y), u2 = ford(v.z), u3 = ford(v.w);
        const uint32_t u4 = ford(w.x), u5 = ford(w.y), u6 = ford(w.z), u7 = ford(w.w);
        const uint32_t h01 = u0 > u1 ? u0 : u1, h23 = u2 > u3 ? u2 : u3, h45 = u4 > u5 ? u4 : u5, h67 = u6 > u7 ? u6 : u7;
        const uint32_t l01 = u0 < u1 ? u0 : u1, l23 = u2 < u3 ? u2 : u3, l45 = u4 < u5 ? u4 : u5, l67 = u6 < u7 ? u6 : u7;
        const uint32_t h4 = h01 > h23 ? h01 : h23, h8 = h45 > h67 ? h45 : h67, l4 = l01 < l23 ? l01 : l23, l8 = l45 < l67 ? l45 : l67;
        const uint32_t hh = h4 > h8 ? h4 : h8, ll = l4 < l8 ? l4 : l8;
        hi = hh > hi ? hh : hi;
        lo = ll < lo ? ll : lo;
    }
    for (int64_t i = tid0; i < h + (n - h - 4 * n4); i += stride) {
        const uint32_t u = ford(a[i < h ? i : 4 * n4 + i]);
        hi = u > hi ? u : hi;
        lo = u < lo ? u : lo;
    }
    mm_fold2(lo, hi, &cm[4 + c], &cm[c]);
}

// Value range of what aff_pad_kernel writes, from the channel extrema alone: x -> x / mx is monotone for mx > 0 (correctly rounded
// division), so min / max of a channel's quotients are the quotients of its min / max, and the padding adds 0.  amm = {ordered min,
// ordered max, valid}; not valid (a channel maximum that is not a positive finite number): the flood measures the range itself.
__global__ void aff_range_kernel(const uint32_t *cm, uint32_t *amm)
{
    float lo = 0.0f, hi = 0.0f;
    bool ok = true;
    for (int c = 0; c < 3; ++c) {
        const float mx = funord(cm[c]), mn = funord(cm[4 + c]);
        ok = ok && mx > 0.0f && mx < __builtin_inff() && mn == mn;
        const float ql = mn / mx, qh = mx / mx;
        lo = ql < lo ? ql : lo;
        hi = qh > hi ? qh : hi;
    }
    amm[0] = ford(lo); amm[1] = ford(hi); amm[2] = ok ? 1u : 0u;
}

__global__ void otsu_edges_kernel(const uint32_t *mm, float *edges)
{
    int i = threadIdx.x;
    if (i > 256) return;
    float mn = funord(mm[0]), mx = funord(mm[1]);
    double start = (double)mn, stop = (double)mx;
    double step = (stop - start) / 256.0;
    double y = (double)i * step;
    y = y + start;
    edges[i] = (i == 256) ? mx : (float)y;
}

__global__ void hist_kernel(const float *__restrict__ a, int64_t n, const uint32_t *mm,
                            const float *__restrict__ edges_g, unsigned long long *counts)
{
    __shared__ unsigned int h[256];
    __shared__ float edges[257];
    for (int i = threadIdx.x; i < 257; i += blockDim.x) edges[i] = edges_g[i];
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    float mn = funord(mm[0]), mx = funord(mm[1]);
    if (mn != mx) {
        float denom = mx - mn;
        auto bin = [&](float v) __attribute__((always_inline)) {
            float f = (v - mn) / denom;
            f = f * 256.0f;
            int idx = (int)f;
            if (idx == 256) idx = 255;
            if (v < edges[idx]) idx -= 1;
            if (v >= edges[idx + 1] && idx != 255) idx += 1;
            atomicAdd(&h[idx], 1u);
        };
        // 16 bytes per lane and load, as minmax_kernel (head up to the first 16-byte boundary and tail one by one)
        const int64_t head = (int64_t)((16 - ((uintptr_t)a & 15)) & 15) / 4;
        const int64_t hd = head < n ? head : n, n4 = (n - hd) / 4;
        const float4 *a4 = (const float4 *)(a + hd);
        const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
        for (int64_t i = tid0; i < n4; i += stride) {
            const float4 v = a4[i];
            bin(v.x); bin(v.y); bin(v.z); bin(v.w);
        }
        for (int64_t i = tid0; i < hd + (n - hd - 4 * n4); i += stride) bin(a[i < hd ? i : 4 * n4 + i]);
    }
    __syncthreads();
    if (threadIdx.x < 256 && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// One lane: 256-bin double statistics, sequential like np.cumsum.
__global__ void otsu_final_kernel(const unsigned long long *counts, const float *edges,
                                  const uint32_t *mm, const float *a, float *thresh)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float mn = funord(mm[0]), mx = funord(mm[1]);
    if (mn == mx) { *thresh = a[0]; return; }
    // suffix sums first (they are consumed at i+1 while walking forward)
    double w2[256], t2[256];
    double acc = 0.0, acc2 = 0.0;
    for (int i = 255; i >= 0; --i) {
        float s = edges[i] + edges[i + 1];
        float c = s / 2.0f;
        double cnt = (double)counts[i];
        acc = acc + cnt;
        acc2 = acc2 + cnt * (double)c;
        w2[i] = acc;
        t2[i] = acc2;
    }
    double w1 = 0.0, t1 = 0.0, bestv = 0.0;
    int best = 0;
    for (int i = 0; i < 255; ++i) {
        float s = edges[i] + edges[i + 1];
        float c = s / 2.0f;
        double cnt = (double)counts[i];
        w1 = w1 + cnt;
        t1 = t1 + cnt * (double)c;
        double m1 = t1 / w1;
        double m2 = t2[i + 1] / w2[i + 1];
        double d = m1 - m2;
        double v = (w1 * w2[i + 1]) * (d * d);
        if (i == 0 || v > bestv) { bestv = v; best = i; }
    }
    float s = edges[best] + edges[best + 1];
    *thresh = s / 2.0f;
}

// mm_ready: the ordered-uint min / max of img, already on the device (gaussian_impl's fused last pass), else nullptr
static int otsu_impl(isx_ctx *ctx, const float *img, int64_t n, float *thresh_dev, uint32_t *mm_ready = nullptr)
{
    Scratch<uint32_t> mm_own(ctx, 2);
    Scratch<float> edges(ctx, 257);
    Scratch<unsigned long long> counts(ctx, 256);
    if (!mm_own.ok() || !edges.ok() || !counts.ok()) return ISX_ERR_NOMEM;
    uint32_t *mm = mm_ready ? mm_ready : mm_own.p;
    ISX_HIP(hipMemsetAsync(counts.p, 0, 256 * sizeof(unsigned long long), ctx->stream));
    unsigned grid = cdiv(n, (int64_t)TPB * 8);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    if (!mm_ready) {
        const uint32_t init[2] = {0xFFFFFFFFu, 0u};
        ISX_HIP(hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
        minmax_kernel<<<grid, TPB, 0, ctx->stream>>>(img, n, mm);
        ISX_CHECK_LAUNCH();
    }
    otsu_edges_kernel<<<1, 320, 0, ctx->stream>>>(mm, edges.p);
    ISX_CHECK_LAUNCH();
    hist_kernel<<<grid, TPB, 0, ctx->stream>>>(img, n, mm, edges.p, counts.p);
    ISX_CHECK_LAUNCH();
    otsu_final_kernel<<<1, 64, 0, ctx->stream>>>(counts.p, edges.p, mm, img, thresh_dev);
    ISX_CHECK_LAUNCH();
    return ISX_OK;
}

// _get_mask's threshold (watershed.py:226-233): Otsu of the sigma-2 Gaussian of img; the Gaussian's last pass hands the min / max over
static int smooth_otsu(isx_ctx *ctx, const float *img, const int64_t shape[3], float *sm, float *tmp, float *thresh_dev)
{
    const int64_t n = shape[0] * shape[1] * shape[2];
    const int32_t sig[3] = {2, 2, 2};
    Scratch<uint32_t> mm(ctx, 2);
    if (!mm.ok()) return ISX_ERR_NOMEM;
    const uint32_t init[2] = {0xFFFFFFFFu, 0u};
    ISX_HIP(hipMemcpyAsync(mm.p, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    bool mm_done = false;
    ISX_TRY(gaussian_impl(ctx, img, sm, shape, sig, tmp, mm.p, &mm_done));
    return otsu_impl(ctx, sm, n, thresh_dev, mm_done ? mm.p : nullptr);
}

// mask (optionally zero-padded by `pad` voxels per side) = img > *thresh
__global__ void mask_kernel(const float *__restrict__ img, const float *thresh, int64_t Z, int64_t Y,
                            int64_t X, int pad, uint8_t *__restrict__ mask)
{
    int64_t PZ = Z + 2 * pad, PY = Y + 2 * pad, PX = X + 2 * pad;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= PZ * PY * PX) return;
    int64_t x, y, z;
    if (PZ * PY * PX < (int64_t)0xFFFFFFFFll) {              // (uniform) 32-bit divisions for every volume but the largest
        const uint32_t i32 = (uint32_t)i, r = i32 / (uint32_t)PX;
        x = (int64_t)(i32 % (uint32_t)PX) - pad; y = (int64_t)(r % (uint32_t)PY) - pad; z = (int64_t)(r / (uint32_t)PY) - pad;
    } else { x = i % PX - pad; y = (i / PX) % PY - pad; z = i / (PX * PY) - pad; }
    uint8_t m = 0;
    if (x >= 0 && x < X && y >= 0 && y < Y && z >= 0 && z < Z) m = img[(z * Y + y) * X + x] > *thresh;
    mask[i] = m;
}

// the same, four consecutive (padded) voxels per thread: two divisions and one 4-byte store for the four (volumes below 2^32 voxels
// whose mask pointer is 4-byte aligned; the tail of a volume that is not a multiple of four is the last thread's)
__global__ void mask4_kernel(const float *__restrict__ img, const float *thresh, int Z, int Y, int X, int pad, uint8_t *__restrict__ mask, uint32_t PN)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, base = t << 2;
    if (base >= PN) return;
    const uint32_t PX = (uint32_t)(X + 2 * pad), PY = (uint32_t)(Y + 2 * pad);
    const float thr = *thresh;
    const uint32_t r = base / PX;
    int x = (int)(base - r * PX) - pad, y = (int)(r % PY) - pad, z = (int)(r / PY) - pad;
    uint32_t m = 0u;
    const int cnt = PN - base < 4u ? (int)(PN - base) : 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < cnt && x >= 0 && x < X && y >= 0 && y < Y && z >= 0 && z < Z)
            m |= (uint32_t)(img[((int64_t)z * Y + y) * X + x] > thr) << (8 * j);
        if (++x == X + pad) { x = -pad; if (++y == Y + pad) { y = -pad; ++z; } }
    }
    if (cnt == 4) *(uint32_t *)(mask + base) = m;
    else for (int j = 0; j < cnt; ++j) mask[base + j] = (uint8_t)((m >> (8 * j)) & 0xFFu);
}

static void launch_mask(isx_ctx *ctx, const float *img, const float *thresh, int64_t Z, int64_t Y, int64_t X, int pad, uint8_t *mask)
{
    const int64_t PN = (Z + 2 * pad) * (Y + 2 * pad) * (X + 2 * pad);
    if (PN < (int64_t)0xFFFFFFF0ll && ((uintptr_t)mask & 3) == 0 && X + 2 * pad >= 4)
        mask4_kernel<<<cdiv((PN + 3) / 4, TPB), TPB, 0, ctx->stream>>>(img, thresh, (int)Z, (int)Y, (int)X, pad, mask, (uint32_t)PN);
    else
        mask_kernel<<<cdiv(PN, TPB), TPB, 0, ctx->stream>>>(img, thresh, Z, Y, X, pad, mask);
}

// ------------------------------------------------------------------------------------
// W2: peak_local_max (skimage 0.18.3 feature/peak.py): 3x3x3 maximum filter with zero
// padding, equality test, all-equal ("trivial") image -> no peaks, > threshold, 1-voxel
// border excluded.  Candidates are emitted as sortable keys:
//   key = (~ford(intensity) << 32) | raveled index   (descending intensity, ascending index)
// ------------------------------------------------------------------------------------
// One block = 4 x 8 x 64 voxels.  The halo tile (6 x 10 x 66, zeros outside the volume) is loaded once, coalesced,
// and the 3x3x3 maximum is taken separably in LDS (x, then y, then z): 9 LDS reads per voxel instead of 27 global
// loads (the direct version ran at 14-24 ms per 512^3 volume whatever the data; this one is bound by reading the
// volume).  "Some voxel is not its own neighbourhood maximum" is written once per block into one of 64 flag words
// on separate cache lines (a single flag word serialises in L2) and OR-ed by peak_flag_kernel.
static constexpr int PEAK_FLAGS = 64, PEAK_FLAG_PITCH = 32;
__global__ __launch_bounds__(256) void peak_kernel(const float *__restrict__ g, int Z, int Y, int X, float thr,
                                                   uint32_t *flagv, unsigned long long *keys, uint32_t *count,
                                                   uint32_t cap, int nty, int ntx)
{
    __shared__ float T[6 * 10 * 66];
    __shared__ float U[6 * 10 * 64];
    __shared__ float V[6 * 8 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int xt = blockIdx.x % ntx, yt = (blockIdx.x / ntx) % nty, zt = blockIdx.x / (ntx * nty);
    const int z0 = zt * 4, y0 = yt * 8, x0 = xt * 64;
    for (int s = tid; s < 6 * 10 * 66; s += 256) {
        const int hx = s % 66, hy = (s / 66) % 10, hz = s / 660;
        const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
        const bool in = z >= 0 && z < Z && y >= 0 && y < Y && x >= 0 && x < X;
        T[s] = in ? g[((uint32_t)z * Y + y) * X + x] : 0.0f;              // mode='constant', cval=0
    }
    __syncthreads();
    for (int s = tid; s < 6 * 10 * 64; s += 256) {
        const int x = s & 63, r = s >> 6;
        const float *t = T + r * 66 + x;
        U[s] = fmaxf(t[0], fmaxf(t[1], t[2]));
    }
    __syncthreads();
    for (int s = tid; s < 6 * 8 * 64; s += 256) {
        const int x = s & 63, y = (s >> 6) & 7, hz = s >> 9;
        const float *u = U + (hz * 10 + y) * 64 + x;
        V[s] = fmaxf(u[0], fmaxf(u[64], u[128]));
    }
    __syncthreads();
    // candidates are emitted with ONE global atomic per block (a returning atomic per wave on the single counter
    // serialises in L2: 2 M of them cost 13 ms at 512^3): ballots per (wave, k), block prefix in LDS
    __shared__ uint32_t wcnt[33];
    bool nonpeak = false;
    unsigned long long ems[8];
    float cs[8];
    const int wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int s = tid + 256 * k;
        const int xx = s & 63, yy = (s >> 6) & 7, zz = s >> 9;
        const int z = z0 + zz, y = y0 + yy, x = x0 + xx;
        const bool in = z < Z && y < Y && x < X;
        const float *v = V + (zz * 8 + yy) * 64 + xx;
        const float m = fmaxf(v[0], fmaxf(v[512], v[1024]));
        const float c = T[((zz + 1) * 10 + (yy + 1)) * 66 + xx + 1];
        const bool interior = z >= 1 && z < Z - 1 && y >= 1 && y < Y - 1 && x >= 1 && x < X - 1;
        if (in && !(c == m)) nonpeak = true;
        const bool emit = in && interior && c == m && c > thr;
        ems[k] = __ballot(emit);
        cs[k] = c;
        if (lane == 0) wcnt[wave * 8 + k] = (uint32_t)__popcll(ems[k]);
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 32; ++i) { const uint32_t c2 = wcnt[i]; wcnt[i] = run; run += c2; }
        wcnt[32] = run ? atomicAdd(count, run) : 0u;          // (spread over 64 cache lines as a timing probe: 1.76 -> 1.65 ms for the whole seed stage -- not the bound)
    }
    __syncthreads();
    const uint32_t gbase = wcnt[32];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (ems[k] & (1ull << lane)) {
            const int s = tid + 256 * k;
            const int xx = s & 63, yy = (s >> 6) & 7, zz = s >> 9;
            const uint32_t slot = gbase + wcnt[wave * 8 + k] + (uint32_t)__popcll(ems[k] & ((1ull << lane) - 1ull));
            if (slot < cap)
                keys[slot] = ((unsigned long long)(~ford(cs[k])) << 32) |
                             (unsigned long long)(((uint32_t)(z0 + zz) * Y + (y0 + yy)) * X + (x0 + xx));
        }
    }
    if (__syncthreads_or(nonpeak) && tid == 0) flagv[(blockIdx.x % PEAK_FLAGS) * PEAK_FLAG_PITCH] = 1u;
}

__global__ void peak_flag_kernel(const uint32_t *flagv, uint32_t *flag)
{
    const uint32_t v = threadIdx.x < PEAK_FLAGS ? flagv[threadIdx.x * PEAK_FLAG_PITCH] : 0u;
    if (__syncthreads_or(v != 0u) && threadIdx.x == 0) *flag = 1u;
}

// keys (sorted) -> (z,y,x) coordinates (+pad) as int64, and/or raveled padded index
__global__ void seeds_decode_kernel(const unsigned long long *keys, uint32_t n, int64_t Y, int64_t X,
                                    int pad, int64_t *coords)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t idx = (int64_t)(keys[i] & 0xFFFFFFFFull);
    coords[3 * i + 0] = idx / (Y * X) + pad;
    coords[3 * i + 1] = (idx / X) % Y + pad;
    coords[3 * i + 2] = idx % X + pad;
}

// ------------------------------------------------------------------------------------
// W3: connected components (6-connectivity) by lock-free union-find over the padded mask
// (root = smallest raveled index of the component), component sizes, size filter.
// ------------------------------------------------------------------------------------
// Every access to the union-find array inside the merge/flatten kernels is an agent-scope
// access (sc1 load / atomic): the per-XCD L2s are not coherent with each other within a launch,
// and a plain load or a plain path-compression store next to another XCD's atomicMin loses
// links (observed on MI355X: a component shed 1-10 voxel pieces, differently on every run).
__device__ __forceinline__ uint32_t uf_ld(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t a)
{
    uint32_t p = uf_ld(parent + a);
    while (p != a) {
        uint32_t gp = uf_ld(parent + p);
        if (gp != p) atomicMin(&parent[a], gp);      // path halving; values only ever decrease
        a = p;
        p = gp;
    }
    return a;
}

__device__ __forceinline__ void uf_union(uint32_t *parent, uint32_t a, uint32_t b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { uint32_t t = a; a = b; b = t; }     // a > b: hang a under b
        uint32_t old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ void ccl_flatten_kernel(uint32_t *parent, int64_t N)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (uf_ld(parent + i) == NO_PARENT) return;
    uint32_t r = uf_find(parent, (uint32_t)i);
    atomicMin(&parent[i], r);
}

// sizes[root] += voxel count.  One wave walks CCL_RUN x 64 contiguous voxels with coalesced loads and
// carries ONE wave-uniform (root, count) run across them: lanes whose root equals the run root are
// counted with a ballot, holes (NO_PARENT) never break the run, and a different root either replaces
// the run (when it is the bigger one so far) or is added directly.  A giant, porous component thus costs
// about one same-address atomic per 2048 voxels (same-address atomics serialise in L2: the first version
// flushed at every hole and spent 350 ms on a 512^3 frame).
// FLATTEN: the pass that makes every voxel point at its root is this same walk (the root a lane counts is the root it has just found
// and stored): one read of the pointers instead of two and one launch less (ccl_flatten_kernel 0.77 + this kernel 0.83 ms at 512^3).
static constexpr int CCL_RUN = 32;
template <bool FLATTEN>
__global__ void ccl_sizes_kernel(uint32_t *parent, uint32_t *sizes, int64_t N)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t base = wave * (int64_t)(64 * CCL_RUN);
    if (base >= N) return;
    uint32_t run_root = NO_PARENT, run_cnt = 0;                  // wave-uniform
    for (int k = 0; k < CCL_RUN; ++k) {
        const int64_t i = base + (int64_t)k * 64 + lane;
        uint32_t p = NO_PARENT;
        if (i < N) {
            if constexpr (FLATTEN) {
                if (uf_ld(parent + i) != NO_PARENT) {
                    p = uf_find(parent, (uint32_t)i);
                    atomicMin(&parent[i], p);
                }
            } else p = parent[i];
        }
        bool todo = p != NO_PARENT;
        if (run_root != NO_PARENT) {
            run_cnt += (uint32_t)__popcll(__ballot(todo && p == run_root));
            todo = todo && p != run_root;
        }
        while (__any(todo)) {
            const int leader = __ffsll((long long)__ballot(todo)) - 1;
            const uint32_t lr = __shfl(p, leader);
            const uint32_t c = (uint32_t)__popcll(__ballot(todo && p == lr));
            if (c > run_cnt) {                                   // adopt the new root, flush the old run
                if (lane == 0 && run_root != NO_PARENT && run_cnt) atomicAdd(&sizes[run_root], run_cnt);
                run_root = lr;
                run_cnt = c;
            } else if (lane == 0) {
                atomicAdd(&sizes[lr], c);
            }
            todo = todo && p != lr;
        }
    }
    if (lane == 0 && run_root != NO_PARENT && run_cnt) atomicAdd(&sizes[run_root], run_cnt);
}

// (four voxels per thread: one 16-byte read of the roots, one 4-byte write of the mask)
__global__ void ccl_apply_kernel(const uint32_t *__restrict__ parent, const uint32_t *__restrict__ sizes,
                                 uint32_t min_area, uint32_t max_area, uint8_t *__restrict__ mask_out,
                                 int64_t N, int vec)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto keep = [&](uint32_t r) __attribute__((always_inline)) -> uint32_t {
        if (r == NO_PARENT) return 0u;
        const uint32_t sz = sizes[r];
        return (sz >= min_area && sz < max_area) ? 1u : 0u;
    };
    if (!vec) {
        if (t < N) mask_out[t] = (uint8_t)keep(parent[t]);
        return;
    }
    const int64_t n4 = N >> 2;
    if (t < n4) {
        const uint4 r = ((const uint4 *)parent)[t];
        ((uint32_t *)mask_out)[t] = keep(r.x) | (keep(r.y) << 8) | (keep(r.z) << 16) | (keep(r.w) << 24);
    } else if (t == n4) {
        for (int64_t i = 4 * n4; i < N; ++i) mask_out[i] = (uint8_t)keep(parent[i]);
    }
}

// Two-level labelling: a block first resolves its own 8 x 16 x 32 tile with a union-find in LDS (no global traffic
// besides one mask read and one parent write per voxel), roots being the smallest raveled index inside the tile; then
// only the voxels on the three low faces of every tile are merged across tiles with the global agent-scope union-find.
// (The one-level version did three global unions per voxel: 10 ms of the 21 ms watershed stage at 512^3.)
static constexpr int CT_Z = 8, CT_Y = 16, CT_X = 32, CT_N = CT_Z * CT_Y * CT_X;     // 4096 voxels, 16 per thread (16 x 16 x 32: this kernel 0.9 -> 1.57 ms for 0.14 ms less in the seam pass)

__device__ __forceinline__ uint32_t lds_find(uint32_t *lab, uint32_t a)
{
    uint32_t p = lab[a];
    while (p != a) { const uint32_t gp = lab[p]; if (gp != p) atomicMin(&lab[a], gp); a = p; p = gp; }
    return a;
}

__device__ __forceinline__ void lds_union(uint32_t *lab, uint32_t a, uint32_t b)
{
    for (;;) {
        a = lds_find(lab, a);
        b = lds_find(lab, b);
        if (a == b) return;
        if (a < b) { const uint32_t t = a; a = b; b = t; }       // a > b: hang a under b
        const uint32_t old = atomicMin(&lab[a], b);
        if (old == a) return;
        a = old;
    }
}

__global__ __launch_bounds__(256) void ccl_tile_kernel(const uint8_t *__restrict__ mask, uint32_t *__restrict__ parent,
                                                       int64_t PZ, int64_t PY, int64_t PX, int ntx, int nty)
{
    __shared__ uint32_t lab[CT_N];
    __shared__ uint8_t mk[CT_N];
    const int tx = blockIdx.x % ntx, ty = (blockIdx.x / ntx) % nty, tz = blockIdx.x / (ntx * nty);
    const int64_t z0 = (int64_t)tz * CT_Z, y0 = (int64_t)ty * CT_Y, x0 = (int64_t)tx * CT_X;
    for (int i = threadIdx.x; i < CT_N; i += 256) {
        const int lx = i % CT_X, ly = (i / CT_X) % CT_Y, lz = i / (CT_X * CT_Y);
        const int64_t z = z0 + lz, y = y0 + ly, x = x0 + lx;
        const bool in = z < PZ && y < PY && x < PX;
        mk[i] = in ? mask[(z * PY + y) * PX + x] : 0;
        lab[i] = (uint32_t)i;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CT_N; i += 256) {
        if (!mk[i]) continue;
        const int lx = i % CT_X, ly = (i / CT_X) % CT_Y, lz = i / (CT_X * CT_Y);
        if (lx > 0 && mk[i - 1]) lds_union(lab, (uint32_t)i, (uint32_t)(i - 1));
        if (ly > 0 && mk[i - CT_X]) lds_union(lab, (uint32_t)i, (uint32_t)(i - CT_X));
        if (lz > 0 && mk[i - CT_X * CT_Y]) lds_union(lab, (uint32_t)i, (uint32_t)(i - CT_X * CT_Y));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CT_N; i += 256) {
        const int lx = i % CT_X, ly = (i / CT_X) % CT_Y, lz = i / (CT_X * CT_Y);
        const int64_t z = z0 + lz, y = y0 + ly, x = x0 + lx;
        if (!(z < PZ && y < PY && x < PX)) continue;
        uint32_t out = NO_PARENT;
        if (mk[i]) {
            const uint32_t r = lds_find(lab, (uint32_t)i);
            const int rx = r % CT_X, ry = (r / CT_X) % CT_Y, rz = r / (CT_X * CT_Y);
            out = (uint32_t)(((z0 + rz) * PY + (y0 + ry)) * PX + (x0 + rx));
        }
        parent[(z * PY + y) * PX + x] = out;
    }
}

// cross-tile links: voxels on the low x / y / z face of their tile
__device__ __forceinline__ void ccl_seam_one(const uint8_t *__restrict__ mask, uint32_t *parent, uint32_t i, uint32_t x, uint32_t y, uint32_t z,
                                             uint32_t px, uint32_t py)
{
    if (x > 0 && x % CT_X == 0 && mask[i - 1]) uf_union(parent, i, i - 1);
    if (y > 0 && y % CT_Y == 0 && mask[i - px]) uf_union(parent, i, i - px);
    if (z > 0 && z % CT_Z == 0 && mask[i - px * py]) uf_union(parent, i, i - px * py);
}

// (32-bit index arithmetic: the launcher guarantees N < 2^32.)  A thread takes 16 consecutive voxels with ONE 16-byte mask load and
// two divisions, then walks the coordinates: a thread per voxel -- 135 M threads at 512^3, most of which read one byte and leave --
// was 1 ms, a quarter of the stage.  `vec` = the mask pointer is 16-byte aligned and rows are at least 16 voxels long.
__global__ void ccl_seam_kernel(const uint8_t *__restrict__ mask, uint32_t *parent, int64_t PZ, int64_t PY, int64_t PX, int vec)
{
    const uint32_t px = (uint32_t)PX, py = (uint32_t)PY, N = (uint32_t)(PZ * PY * PX);
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (!vec) {
        const uint32_t i = t;
        if (i >= N || !mask[i]) return;
        const uint32_t x = i % px, r = i / px, y = r % py, z = r / py;
        ccl_seam_one(mask, parent, i, x, y, z, px, py);
        return;
    }
    const uint32_t n16 = N >> 4;
    if (t > n16) return;
    const uint32_t base = t << 4;
    uint32_t m[4] = {0u, 0u, 0u, 0u};
    int cntv = 16;
    if (t < n16) { const uint4 v = ((const uint4 *)mask)[t]; m[0] = v.x; m[1] = v.y; m[2] = v.z; m[3] = v.w; }
    else { cntv = (int)(N - base); for (int j = 0; j < cntv; ++j) m[j >> 2] |= (uint32_t)(mask[base + j] != 0) << (8 * (j & 3)); }
    if (!(m[0] | m[1] | m[2] | m[3])) return;
    uint32_t x = base % px, r = base / px, y = r % py, z = r / py;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (j < cntv && ((m[j >> 2] >> (8 * (j & 3))) & 0xFFu)) ccl_seam_one(mask, parent, base + j, x, y, z, px, py);
        if (++x == px) { x = 0; if (++y == py) { y = 0; ++z; } }
    }
}

static int ccl_impl(isx_ctx *ctx, const uint8_t *mask, const int64_t pshape[3], uint32_t *parent,
                    uint32_t *sizes)
{
    int64_t N = pshape[0] * pshape[1] * pshape[2];
    ISX_REQUIRE(N < (int64_t)0xFFFFFFF0ll, "volume too large for 32-bit voxel indices");
    unsigned grid = cdiv(N, TPB);
    const int ntz = (int)((pshape[0] + CT_Z - 1) / CT_Z), nty = (int)((pshape[1] + CT_Y - 1) / CT_Y),
              ntx = (int)((pshape[2] + CT_X - 1) / CT_X);
    ccl_tile_kernel<<<(unsigned)(ntz * nty * ntx), 256, 0, ctx->stream>>>(mask, parent, pshape[0], pshape[1], pshape[2], ntx, nty);
    ISX_CHECK_LAUNCH();
    const int vec = ((uintptr_t)mask & 15) == 0 && pshape[2] >= 16;
    ccl_seam_kernel<<<vec ? cdiv((N >> 4) + 1, TPB) : grid, TPB, 0, ctx->stream>>>(mask, parent, pshape[0], pshape[1], pshape[2], vec);
    ISX_CHECK_LAUNCH();
    // (tried in round 5 and dropped: four voxels per thread in the flatten pass -- 1.13 instead of 0.77 ms, its time is the pointer
    //  chases, not the launch -- and per-tile counts from the tile kernel folded into the roots by the flatten pass instead of
    //  ccl_sizes_kernel: the tile kernel 0.86 -> 1.8 ms, the flatten pass 11 ms)
    if (sizes && ctx->ftune.ccl_fused) {
        ISX_HIP(hipMemsetAsync(sizes, 0, N * sizeof(uint32_t), ctx->stream));
        ccl_sizes_kernel<true><<<cdiv(cdiv(N, (int64_t)64 * CCL_RUN) * 64, TPB), TPB, 0, ctx->stream>>>(parent, sizes, N);
        ISX_CHECK_LAUNCH();
        return ISX_OK;
    }
    ccl_flatten_kernel<<<grid, TPB, 0, ctx->stream>>>(parent, N);
    ISX_CHECK_LAUNCH();
    if (sizes) {
        ISX_HIP(hipMemsetAsync(sizes, 0, N * sizeof(uint32_t), ctx->stream));
        ccl_sizes_kernel<false><<<cdiv(cdiv(N, (int64_t)64 * CCL_RUN) * 64, TPB), TPB, 0, ctx->stream>>>(parent, sizes, N);
        ISX_CHECK_LAUNCH();
    }
    return ISX_OK;
}

// seed bookkeeping ---------------------------------------------------------------
__global__ void seed_ravel_kernel(const int64_t *coords, uint32_t n, int64_t PY, int64_t PX,
                                  uint32_t *ridx)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ridx[i] = (uint32_t)((coords[3 * i] * PY + coords[3 * i + 1]) * PX + coords[3 * i + 2]);
}

// Standalone W4 entry: seeds must lie strictly inside the padded box (the flood reads all six
// neighbours unchecked, like the reference); a seed on a non-mask voxel still spreads into the
// mask voxels around it (the reference never tests mask[seed]), so it is added to the mask used
// for component analysis -- it is pre-labelled, hence never entered, and the flood is unchanged.
__global__ void seed_check_mark_kernel(const int64_t *coords, uint32_t n, int64_t PZ, int64_t PY, int64_t PX,
                                       uint8_t *mask, uint32_t *bad)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t z = coords[3 * i], y = coords[3 * i + 1], x = coords[3 * i + 2];
    if (z < 1 || z > PZ - 2 || y < 1 || y > PY - 2 || x < 1 || x > PX - 2) { *bad = 1u; return; }
    mask[(z * PY + y) * PX + x] = 1;
}

__global__ void seed_flag_kernel(const uint32_t *ridx, uint32_t n, const uint8_t *mask, uint32_t *flag)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = mask[ridx[i]] ? 1u : 0u;
}

__global__ void seed_copy_kernel(const int64_t *coords, uint32_t n, int64_t *out, int sub)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[3 * i + 0] = coords[3 * i + 0] - sub;
    out[3 * i + 1] = coords[3 * i + 1] - sub;
    out[3 * i + 2] = coords[3 * i + 2] - sub;
}

__global__ void seed_compact_kernel(const int64_t *coords, const uint32_t *flag, const uint32_t *pos,
                                    uint32_t n, int64_t *out, int sub)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    uint32_t o = pos[i];
    out[3 * o + 0] = coords[3 * i + 0] - sub;
    out[3 * o + 1] = coords[3 * i + 1] - sub;
    out[3 * o + 2] = coords[3 * i + 2] - sub;
}

// ------------------------------------------------------------------------------------
// W0: affinities / per-channel max, zero-padded by one voxel (watershed.py:194-201).
// ------------------------------------------------------------------------------------
// One block per padded (channel, z, y) row: no per-voxel index divisions (a thread per padded voxel with three 64-bit
// divisions ran at 2.3 TB/s of its 3.2 GB).
__global__ __launch_bounds__(256) void aff_pad_kernel(const float *__restrict__ fm, const uint32_t *cm, int64_t Z, int64_t Y,
                                                      int64_t X, float *__restrict__ aff)
{
    const int64_t PZ = Z + 2, PY = Y + 2, PX = X + 2;
    // 1-D grid over (channel, z, y): two 32-bit divisions per BLOCK (a 2-D grid capped the depth at 21 843 planes)
    const unsigned cz = blockIdx.x / (unsigned)PY;
    const int y = (int)(blockIdx.x % (unsigned)PY) - 1, c = (int)(cz / (unsigned)PZ), z = (int)(cz % (unsigned)PZ) - 1;
    float *row = aff + (((int64_t)c * PZ + (z + 1)) * PY + (y + 1)) * PX;
    const bool inside = z >= 0 && z < Z && y >= 0 && y < Y;
    const float *src = fm + (((int64_t)c * Z + (inside ? z : 0)) * Y + (inside ? y : 0)) * X;
    const float mx = funord(cm[c]);
    for (int px = (int)threadIdx.x; px < PX; px += 256) {
        const int x = px - 1;
        float v = 0.0f;
        if (inside && x >= 0 && x < X) v = src[x] / mx;
        row[px] = v;
    }
}

// ------------------------------------------------------------------------------------
// W4: the flood.  Exactly the reference's sequential process per mask component
// (SURVEY.md F3/F4): min-heap on (value, age, index); a voxel is labelled when first
// discovered; neighbours visited in the order [-z, -y, -x, +x, +y, +z]; the edge value is
// aff[axis][idx] for the negative directions and aff[axis][neighbour] for the positive.
// Components are independent, so each one is flooded by its own lane with a private
// binary heap in HBM; ages are per-component push counters (only their order matters).
// Heap entry: 16 bytes {key = ford(value) << 32 | age, idx}.
// ------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) HeapEnt {
    unsigned long long key;
    uint32_t idx;
    uint32_t pad;
};

__device__ __forceinline__ bool ent_less(const HeapEnt &a, const HeapEnt &b)
{
    return a.key < b.key || (a.key == b.key && a.idx < b.idx);
}

__device__ __forceinline__ void heap_push(HeapEnt *h, uint32_t &n, HeapEnt e)
{
    uint32_t c = n++;
    while (c > 0) {
        uint32_t p = (c - 1) >> 1;
        HeapEnt pe = h[p];
        if (!ent_less(e, pe)) break;
        h[c] = pe;
        c = p;
    }
    h[c] = e;
}

__device__ __forceinline__ HeapEnt heap_pop(HeapEnt *h, uint32_t &n)
{
    HeapEnt top = h[0];
    HeapEnt last = h[--n];
    if (n > 0) {
        uint32_t c = 0;
        for (;;) {
            uint32_t l = 2 * c + 1;
            if (l >= n) break;
            HeapEnt le = h[l];
            uint32_t m = l;
            if (l + 1 < n) {
                HeapEnt re = h[l + 1];
                if (ent_less(re, le)) { le = re; m = l + 1; }
            }
            if (!ent_less(le, last)) break;
            h[c] = le;
            c = m;
        }
        h[c] = last;
    }
    return top;
}

__global__ void seed_label_kernel(const uint32_t *ridx, uint32_t n, const uint32_t *parent,
                                  uint32_t *out, unsigned long long *keys)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t p = ridx[i];
    out[p] = i + 1;                                        // _prep_data, watershed.py:61-62
    uint32_t root = parent[p];                             // NO_PARENT if the seed is off-mask
    keys[i] = ((unsigned long long)root << 32) | p;
}

// after sorting by (root, idx): head flags of each component run
__global__ void comp_head_kernel(const unsigned long long *keys, uint32_t n, uint32_t *head)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = (uint32_t)(keys[i] >> 32);
    head[i] = (i == 0 || r != (uint32_t)(keys[i - 1] >> 32)) ? 1u : 0u;
}

// comp_first[c] = index of the first seed of component c; comp_cap[c] = heap capacity
__global__ void comp_table_kernel(const unsigned long long *keys, const uint32_t *head,
                                  const uint32_t *cid, uint32_t n, const uint32_t *sizes,
                                  uint32_t *comp_first, unsigned long long *comp_cap)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    uint32_t c = cid[i];
    uint32_t root = (uint32_t)(keys[i] >> 32);
    comp_first[c] = i;
    comp_cap[c] = (root == NO_PARENT) ? 1ull : (unsigned long long)sizes[root];
}

// The reference's loop on one lane: pop the minimum, label + push the unlabelled mask neighbours.
__device__ __forceinline__ void serial_flood(HeapEnt *h, uint32_t hn, uint32_t age, const float *__restrict__ aff,
                                             const uint8_t *__restrict__ mask, uint32_t *__restrict__ out,
                                             int64_t PY, int64_t PX, int64_t PN, float s0, float s1, float s2)
{
    const int64_t SZ = PY * PX, SY = PX;
    const int64_t noff[6] = {-SZ, -SY, -1, 1, SY, SZ};
    const int ax[6] = {0, 1, 2, 2, 1, 0};
    const float sc[3] = {s0, s1, s2};
    while (hn > 0) {
        HeapEnt top = heap_pop(h, hn);
        int64_t idx = top.idx;
        uint32_t lab = out[idx];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            int64_t nb = idx + noff[k];
            if (!mask[nb] || out[nb]) continue;
            out[nb] = lab;
            float v = aff[(int64_t)ax[k] * PN + (k < 3 ? idx : nb)];
            v = v * sc[ax[k]];
            v = v + 0.0f;                                   // -0.0 == +0.0 in the heap order
            HeapEnt e;
            e.key = ((unsigned long long)ford(v) << 32) | (unsigned long long)(++age);
            e.idx = (uint32_t)nb;
            e.pad = 0;
            heap_push(h, hn, e);
        }
    }
}

// Small components: one lane per component (thousands of them run side by side).
// Component -> rank for the multi-GPU flood (W4 shards exactly by mask component, F4): contiguous
// ranges of the (root-sorted) component list balanced by voxel count; identical on every rank.
__device__ __forceinline__ bool comp_owned(const unsigned long long *comp_off, uint32_t c, uint32_t n_comp, int rank, int world)
{
    if (world <= 1) return true;
    const unsigned long long total = comp_off[n_comp];
    const int owner = total ? (int)((comp_off[c] * (unsigned long long)world) / total) : 0;
    return owner == rank;
}

__global__ void flood_kernel(const float *__restrict__ aff, const uint8_t *__restrict__ mask,
                             uint32_t *__restrict__ out, const unsigned long long *__restrict__ keys,
                             uint32_t n_seeds, const uint32_t *__restrict__ comp_first,
                             const unsigned long long *__restrict__ comp_off, uint32_t n_comp,
                             HeapEnt *__restrict__ heap_pool, int64_t PY, int64_t PX, int64_t PN,
                             float s0, float s1, float s2, unsigned long long coop_min_size, int shard_rank,
                             int shard_world)
{
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_comp) return;
    if (!comp_owned(comp_off, c, n_comp, shard_rank, shard_world)) return;
    uint32_t first = comp_first[c];
    uint32_t end = (c + 1 < n_comp) ? comp_first[c + 1] : n_seeds;
    if ((uint32_t)(keys[first] >> 32) == NO_PARENT) return;   // seeds outside the mask never spread
    if (comp_off[c + 1] - comp_off[c] >= coop_min_size) return;   // flooded by flood_wave_kernel
    HeapEnt *h = heap_pool + comp_off[c];
    uint32_t hn = 0;
    // seeds are sorted by index inside the component; age = rank keeps (value 0, age 0, index) order
    for (uint32_t s = first; s < end; ++s) {
        HeapEnt e;
        e.key = ((unsigned long long)0x80000000u << 32) | (unsigned long long)(s - first);
        e.idx = (uint32_t)(keys[s] & 0xFFFFFFFFull);
        e.pad = 0;
        heap_push(h, hn, e);
    }
    serial_flood(h, hn, end - first - 1, aff, mask, out, PY, PX, PN, s0, s1, s2);
}

// ------------------------------------------------------------------------------------------------
// Large components: one WAVE per component, exact speculative batches.
//
// The 64 smallest queued elements sit sorted across the lanes (lane r = rank r); a sorted run in LDS
// holds what follows.  Per round every lane treats its element as popped: it loads its six
// neighbours and edge values at once, competes for unlabelled neighbours through claim[] (atomicMax of
// epoch*64 + 63-rank: the lowest rank of the newest round wins), and the round commits the longest
// prefix of ranks for which no lower rank pushed a STRICTLY smaller value (an equal value is younger,
// hence later) -- exactly the pops the sequential heap would have made next, in the same order.
// Ages are handed out in (rank, direction) order, i.e. the reference's push order.  On smooth fields
// the prefix is short (chains) but still saves the heap traffic; on tie-heavy maps it is the whole
// wave.  If the LDS run overflows, the component continues on the exact serial heap.
// ------------------------------------------------------------------------------------------------
static constexpr int COOP_SCAP = 2048;                      // LDS run capacity (entries of 16 B)
static constexpr unsigned long long COOP_MIN_SIZE = 512;    // components below this stay on flood_kernel
static constexpr unsigned long long COLD_MIN_SIZE = 4096;   // components from this size get a cold arena (raised by the host until the arenas fit COLD_ARENA_BUDGET)
static constexpr unsigned long long COLD_ARENA_BUDGET = 8ull << 30;   // bytes; one arena is >= 7.4 MB (bucket tables), ISX_FLOOD_ARENA_MB overrides
static constexpr uint32_t COLD_UNSORTED = 0x80000000u;
static constexpr int COLD_NBK = 65536;                      // buckets: top 16 bits of the order-preserving key
static constexpr int COLD_DRAIN = 1024;                     // elements moved from a bucket to the LDS run at once
static constexpr uint32_t COLD_NONE = 0xFFFFFFFFu;
// arena layout in 16-byte units: bhead | btail | bcount | bmeta | bfill (u32[NBK] each), blast (u64[NBK]), bm0 (u64[NBK/64]), bm1 (u64[NBK/4096]), pool
static constexpr uint32_t COLD_META_ENTS = (5u * COLD_NBK * 4u + COLD_NBK * 8u + COLD_NBK / 64 * 8u + COLD_NBK / 4096 * 8u) / 16u;

__host__ __device__ inline unsigned long long cold_arena_entries(unsigned long long size)
{
    return size >= COLD_MIN_SIZE ? (unsigned long long)COLD_META_ENTS + 4ull * size + 5ull * COLD_NBK + 4096ull : 0ull;
}

__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src)
{
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl(lo, src);
    hi = __shfl(hi, src);
    return ((unsigned long long)hi << 32) | lo;
}

// Broadcast from a wave-uniform lane: v_readlane (scalar result, a few cycles) instead of ds_bpermute.
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ unsigned long long rdlane64(unsigned long long v, int l)
{
    return ((unsigned long long)rdlane((uint32_t)(v >> 32), l) << 32) | rdlane((uint32_t)v, l);
}

// 16-byte queue entries are moved as vector values (a conditionally assigned struct temporary across an asm
// barrier was kept in scratch memory by the compiler: a private-memory round trip per 64 entries shifted)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct HeapEnt;
__device__ __forceinline__ u32x4 run_ld(const HeapEnt *run, int i) { return ((const u32x4 *)run)[i]; }
__device__ __forceinline__ void run_st(HeapEnt *run, int i, u32x4 v) { ((u32x4 *)run)[i] = v; }

// Cross-lane moves on the DPP path (a few cycles) instead of ds_bpermute (an LDS-crossbar round trip each).
// wave_shr:1 -- lane i receives lane i - 1, lane 0 receives `fill`.
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t fill)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x138, 0xf, 0xf, false);
}
// inclusive prefix minimum over the 64 lanes: row_shr 1/2/4/8 inside each row of 16, then row_bcast 15 and 31
__device__ __forceinline__ uint32_t wave_prefix_min(uint32_t v)
{
    const int id = -1;       // 0xFFFFFFFF: the identity of an unsigned minimum
#define ISX_SCAN_STEP(ctrl, rmask) { const uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)v, ctrl, rmask, 0xf, false); v = t < v ? t : v; }
    ISX_SCAN_STEP(0x111, 0xf)
    ISX_SCAN_STEP(0x112, 0xf)
    ISX_SCAN_STEP(0x114, 0xf)
    ISX_SCAN_STEP(0x118, 0xf)
    ISX_SCAN_STEP(0x142, 0xa)
    ISX_SCAN_STEP(0x143, 0xc)
#undef ISX_SCAN_STEP
    return v;
}

// Every read of the cold structures bypasses the (non-coherent, possibly stale) vector L1: they are rewritten
// in place by this wave between reads.
__device__ __forceinline__ uint32_t cld32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long cld64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ HeapEnt cld_ent(const HeapEnt *p)
{
    HeapEnt e;
    e.key = cld64(&p->key);
    const unsigned long long t = cld64((const unsigned long long *)&p->idx);
    e.idx = (uint32_t)t;
    e.pad = (uint32_t)(t >> 32);
    return e;
}

struct Cold {
    // arena of one component; its views are computed from the base on use (eight live pointers cost 16 SGPRs in kernels
    // that are short of them).  Layout in 16-byte units: bhead | btail | bcount | bmeta | bfill (u32[NBK] each),
    // blast (u64[NBK]), bm0 (u64[NBK/64]), bm1 (u64[NBK/4096]), pool
    uint32_t *base;
    __device__ __forceinline__ uint32_t *bhead() const { return base; }
    __device__ __forceinline__ uint32_t *btail() const { return base + COLD_NBK; }
    __device__ __forceinline__ uint32_t *bcount() const { return base + 2 * COLD_NBK; }
    // bits 0..30: consumed prefix of the head segment; bit 31: appended out of key order
    __device__ __forceinline__ uint32_t *bmeta() const { return base + 3 * COLD_NBK; }
    // tail segment: entries used | log2(capacity) << 24 (a copy of its header: one load trip per append)
    __device__ __forceinline__ uint32_t *bfill() const { return base + 4 * COLD_NBK; }
    // key of the last appended element
    __device__ __forceinline__ unsigned long long *blast() const { return (unsigned long long *)(base + 5 * COLD_NBK); }
    __device__ __forceinline__ unsigned long long *bm0() const { return blast() + COLD_NBK; }
    __device__ __forceinline__ unsigned long long *bm1() const { return bm0() + COLD_NBK / 64; }
    // a segment = header entry {key = next | n << 32, idx = cap} + cap entries
    __device__ __forceinline__ HeapEnt *pool() const { return (HeapEnt *)base + COLD_META_ENTS; }
    uint32_t pool_cap, bump, count;
    uint32_t fifo_drains;      // statistics: drains served by the prefix of an ascending list
    // write-through copy of one bucket's append state (plateaus append to the same bucket round after round):
    // saves the three dependent loads of the leader.  Wave-uniform.  Dropped whenever a drain touches the bucket.
    uint32_t cb, cb_tail, cb_tn, cb_tcap, cb_cnt;
    unsigned long long cb_last;
    bool cb_unsorted;
    uint32_t low;              // lowest non-empty bucket (COLD_NBK when empty): decides which new elements may stay hot
    float vlo, vinv;           // bucket = (value - vlo) * vinv, clamped
    bool on, failed;
#ifdef ISX_FLOOD_PROF
    unsigned long long tk[8], tk_prev, n_groups;
#endif
};

#ifdef ISX_FLOOD_PROF
#define COLD_TICK0(c) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); (c).tk_prev = __builtin_readcyclecounter(); } while (0)
#define COLD_TICK(c, i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_readcyclecounter(); (c).tk[i] += t_ - (c).tk_prev; (c).tk_prev = t_; } while (0)
#else
#define COLD_TICK0(c) do { } while (0)
#define COLD_TICK(c, i) do { } while (0)
#endif

// Bucket of a key: a monotone (non-strict) map of the edge VALUE onto [0, COLD_NBK): linear between the smallest and the
// largest scaled affinity of the volume (host: flood_impl), so that the buckets in use are evenly filled whatever binade
// the values live in (the top 16 bits of the ordered float gave 128 buckets per binade: ~250 buckets for [0.25, 1)).
__device__ __forceinline__ uint32_t cold_bucket(const Cold &c, unsigned long long key)
{
    const float t = (funord((uint32_t)(key >> 32)) - c.vlo) * c.vinv;
    return t >= (float)(COLD_NBK - 1) ? (uint32_t)(COLD_NBK - 1) : (t > 0.0f ? (uint32_t)t : 0u);
}

// Append up to one element per lane to the buckets.  Lanes with the same bucket form a group (leader = lowest
// lane); leaders do the bookkeeping in parallel, members get their slot from the leader.
__device__ __forceinline__ void cold_append(Cold &c, bool has, unsigned long long key, uint32_t idx, uint32_t lab, int lane)
{
    COLD_TICK0(c);
    const uint32_t b = cold_bucket(c, key);
    // earlier arena stores of this wave must be visible to the leaders' loads; a round whose elements all go to the
    // cached bucket loads nothing and lets its stores (and the round's label stores) drain under the next round
    const bool uncached = has && b != c.cb;
    if (__any(uncached)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // every member loads its bucket's append state now (five independent loads, same addresses within a group): the
    // round trip runs under the grouping below instead of after it
    uint32_t m_tail = COLD_NONE, m_cnt = 0, m_fill = 0, m_meta = 0;
    unsigned long long m_last = 0ull;
    if (uncached) {
        m_tail = cld32(&c.btail()[b]);
        m_cnt = cld32(&c.bcount()[b]);
        m_fill = cld32(&c.bfill()[b]);
        m_meta = cld32(&c.bmeta()[b]);
        m_last = cld64(&c.blast()[b]);
    }
    COLD_TICK(c, 0);
    int gleader = 0, grank = 0, gsize = 0;
    unsigned long long glast = 0ull, ggrp = 0ull;
    uint32_t minb = (uint32_t)COLD_NBK;            // lowest bucket of this batch (wave-uniform)
    bool gviol = false;
    {
        // lanes with the same bucket ("match any"): one ballot per bucket-id bit, each lane keeps the lanes that agree with it
        // on every bit.  (A loop over the distinct buckets cost ~16 k cycles per batch of 64 residuals spread over ~50
        // buckets: 12 % of the config-2 flood.)
        unsigned long long same = __ballot(has);
        static_assert(COLD_NBK <= 65536, "bucket ids are 16 bits");
#pragma unroll
        for (int bit = 0; bit < 16; ++bit) {
            const bool one = (b >> bit) & 1u;
            const unsigned long long bal = __ballot(has && one);
            same &= one ? bal : ~bal;
        }
        if (has) {
            ggrp = same;
            gleader = __ffsll((long long)same) - 1;
            grank = __popcll(same & ((1ull << lane) - 1ull));
            gsize = __popcll(same);
        }
        uint32_t mb = has ? b : (uint32_t)COLD_NBK;
        for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)mb, o); mb = t < mb ? t : mb; }
        minb = mb;
#ifdef ISX_FLOOD_PROF
        ++c.n_groups;
#endif
    }
    {
        // is each group's placement order (lane order) also its key order?  what is its last key?
        const unsigned long long before = ggrp & ((1ull << lane) - 1ull);
        const unsigned long long pk = shfl64(key, before ? 63 - __clzll((long long)before) : lane);
        const unsigned long long vb = __ballot(has && before && key < pk);
        gviol = (vb & ggrp) != 0ull;
        glast = shfl64(key, has ? 63 - __clzll((long long)ggrp) : lane);
    }
    COLD_TICK(c, 1);
    const bool leader = has && gleader == lane;
    uint32_t tail = COLD_NONE, cnt = 0, tn = 0, tcap = 0;
    bool unsorted = false;
    if (leader) {
        // a bucket stays FIFO-drainable while every append continues its ascending key order
        if (b == c.cb) {
            tail = c.cb_tail; cnt = c.cb_cnt; tn = c.cb_tn; tcap = c.cb_tcap;
            unsorted = c.cb_unsorted;
            if (!unsorted && (gviol || c.cb_last > key)) { unsorted = true; atomicOr(&c.bmeta()[b], COLD_UNSORTED); }
        } else {
            tail = m_tail;
            cnt = m_cnt;
            if (tail != COLD_NONE) {
                tn = m_fill & 0xFFFFFFu;
                tcap = 1u << (m_fill >> 24);
                if (cnt > 0 && m_last > key) gviol = true;
            }
            unsorted = (m_meta & COLD_UNSORTED) != 0u;
            if (gviol && !unsorted) { unsorted = true; atomicOr(&c.bmeta()[b], COLD_UNSORTED); }
        }
    }
#ifdef ISX_FLOOD_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    COLD_TICK(c, 2);
    const uint32_t free_tail = leader ? tcap - tn : 0u;
    const uint32_t need_new = (leader && (uint32_t)gsize > free_tail) ? (uint32_t)gsize - free_tail : 0u;
    uint32_t newcap = 0;
    if (need_new) {
        newcap = tcap ? tcap * 2u : 4u;
        if (newcap > 65536u) newcap = 65536u;
        while (newcap < need_new) newcap *= 2u;
    }
    const uint32_t alloc = need_new ? newcap + 1u : 0u;
    // bump allocation for the (few) leaders that open a segment: a scalar loop over them
    uint32_t my_off = 0, total = 0;
    for (unsigned long long am = __ballot(alloc != 0u); am; am &= am - 1ull) {
        const int l = __ffsll((long long)am) - 1;
        if (lane == l) my_off = c.bump + total;
        total += rdlane(alloc, l);
    }
    if ((unsigned long long)c.bump + total > c.pool_cap) { c.failed = true; return; }
    c.bump += total;
    if (leader) {
        if (need_new) {
            HeapEnt h;
            h.key = (unsigned long long)COLD_NONE | ((unsigned long long)need_new << 32);
            h.idx = newcap;
            h.pad = 0;
            c.pool()[my_off] = h;
            if (tail != COLD_NONE) {
                HeapEnt ht;
                ht.key = (unsigned long long)my_off | ((unsigned long long)tcap << 32);    // old tail is now full
                ht.idx = tcap;
                ht.pad = 0;
                c.pool()[tail] = ht;
            } else {
                c.bhead()[b] = my_off;
            }
            c.btail()[b] = my_off;
        } else {
            HeapEnt ht;
            ht.key = (unsigned long long)COLD_NONE | ((unsigned long long)(tn + (uint32_t)gsize) << 32);   // a tail has no successor
            ht.idx = tcap;
            ht.pad = 0;
            c.pool()[tail] = ht;
        }
        c.bcount()[b] = cnt + (uint32_t)gsize;
        {
            const uint32_t f_tn = need_new ? need_new : tn + (uint32_t)gsize, f_cap = need_new ? newcap : tcap;
            c.bfill()[b] = f_tn | ((uint32_t)(31 - __clz((int)f_cap)) << 24);
            c.blast()[b] = glast;
        }
        if (cnt == 0) {
            atomicOr(&c.bm0()[b >> 6], 1ull << (b & 63));
            atomicOr(&c.bm1()[b >> 12], 1ull << ((b >> 6) & 63));
        }
    }
    {
        // the first group's bucket becomes the cached one
        const int l0 = __ffsll((long long)__ballot(has)) - 1;
        if (l0 >= 0) {
            const uint32_t n_tail = need_new ? my_off : tail;
            const uint32_t n_tn = need_new ? need_new : tn + (uint32_t)gsize;
            const uint32_t n_tcap = need_new ? newcap : tcap;
            c.cb = rdlane(b, l0);
            c.cb_tail = rdlane(n_tail, l0);
            c.cb_tn = rdlane(n_tn, l0);
            c.cb_tcap = rdlane(n_tcap, l0);
            c.cb_cnt = rdlane(cnt + (uint32_t)gsize, l0);
            c.cb_unsorted = rdlane((uint32_t)unsorted, l0) != 0u;
            c.cb_last = rdlane64(glast, l0);
        }
    }
    const int gl = gleader & 63;
    const uint32_t l_tail = __shfl(tail, gl), l_tn = __shfl(tn, gl), l_free = __shfl(free_tail, gl), l_new = __shfl(my_off, gl);
    if (has) {
        const uint32_t slot = ((uint32_t)grank < l_free) ? l_tail + 1u + l_tn + (uint32_t)grank : l_new + 1u + ((uint32_t)grank - l_free);
        HeapEnt e;
        e.key = key;
        e.idx = idx;
        e.pad = lab;
        c.pool()[slot] = e;
    }
    COLD_TICK(c, 3);
    c.count += (uint32_t)__popcll(__ballot(has));
    c.low = minb < c.low ? minb : c.low;
    // no wait here: every reader of the arena (cold_append, cold_drain, cold_lowest, the hand-over) starts with
    // s_waitcnt vmcnt(0), so these stores complete under the LDS work that follows
}

// Append src[0, n) (LDS, list order = src order) to ONE bucket: the wave-uniform form of cold_append for a batch whose elements
// all map to bucket b (plateaus: every residual of a commit burst has the same value).  One state update and at most one new
// segment for the whole batch instead of one cold_append per 64 elements (the work-group kernel's flush of ~350 residuals per
// sub-round was 6 calls of ~5 k cycles each on one wave while seven waited).  `ascending`: src is in key order.
__device__ __forceinline__ void cold_append_bucket(Cold &c, const HeapEnt *src, int n, uint32_t b, bool ascending,
                                                   unsigned long long first_key, unsigned long long last_key, int lane)
{
    uint32_t tail, cnt, tn = 0, tcap = 0;
    bool unsorted;
    unsigned long long last;
    if (b == c.cb) {
        tail = c.cb_tail; cnt = c.cb_cnt; tn = c.cb_tn; tcap = c.cb_tcap; unsorted = c.cb_unsorted; last = c.cb_last;
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tail = cld32(&c.btail()[b]);
        cnt = cld32(&c.bcount()[b]);
        const uint32_t fill = cld32(&c.bfill()[b]), meta = cld32(&c.bmeta()[b]);
        last = cld64(&c.blast()[b]);
        if (tail != COLD_NONE) { tn = fill & 0xFFFFFFu; tcap = 1u << (fill >> 24); }
        unsorted = (meta & COLD_UNSORTED) != 0u;
    }
    const bool viol = !ascending || (tail != COLD_NONE && cnt > 0 && last > first_key);
    if (viol && !unsorted) { unsorted = true; if (lane == 0) atomicOr(&c.bmeta()[b], COLD_UNSORTED); }
    const uint32_t free_tail = tcap - tn;
    const uint32_t need_new = (uint32_t)n > free_tail ? (uint32_t)n - free_tail : 0u;
    uint32_t newcap = 0;
    if (need_new) {
        newcap = tcap ? tcap * 2u : 4u;
        if (newcap > 65536u) newcap = 65536u;
        while (newcap < need_new) newcap *= 2u;
    }
    const uint32_t alloc = need_new ? newcap + 1u : 0u;
    if ((unsigned long long)c.bump + alloc > c.pool_cap) { c.failed = true; return; }
    const uint32_t my_off = c.bump;
    c.bump += alloc;
    if (lane == 0) {
        if (need_new) {
            HeapEnt h;
            h.key = (unsigned long long)COLD_NONE | ((unsigned long long)need_new << 32);
            h.idx = newcap;
            h.pad = 0;
            c.pool()[my_off] = h;
            if (tail != COLD_NONE) {
                HeapEnt ht;
                ht.key = (unsigned long long)my_off | ((unsigned long long)tcap << 32);    // old tail is now full
                ht.idx = tcap;
                ht.pad = 0;
                c.pool()[tail] = ht;
            } else {
                c.bhead()[b] = my_off;
            }
            c.btail()[b] = my_off;
        } else {
            HeapEnt ht;
            ht.key = (unsigned long long)COLD_NONE | ((unsigned long long)(tn + (uint32_t)n) << 32);
            ht.idx = tcap;
            ht.pad = 0;
            c.pool()[tail] = ht;
        }
        c.bcount()[b] = cnt + (uint32_t)n;
        const uint32_t f_tn = need_new ? need_new : tn + (uint32_t)n, f_cap = need_new ? newcap : tcap;
        c.bfill()[b] = f_tn | ((uint32_t)(31 - __clz((int)f_cap)) << 24);
        c.blast()[b] = last_key;
        if (cnt == 0) {
            atomicOr(&c.bm0()[b >> 6], 1ull << (b & 63));
            atomicOr(&c.bm1()[b >> 12], 1ull << ((b >> 6) & 63));
        }
    }
    for (uint32_t j = (uint32_t)lane; j < (uint32_t)n; j += 64u) {
        const uint32_t slot = j < free_tail ? tail + 1u + tn + j : my_off + 1u + (j - free_tail);
        c.pool()[slot] = src[j];
    }
    c.cb = b;
    c.cb_tail = need_new ? my_off : tail;
    c.cb_tn = need_new ? need_new : tn + (uint32_t)n;
    c.cb_tcap = need_new ? newcap : tcap;
    c.cb_cnt = cnt + (uint32_t)n;
    c.cb_unsorted = unsorted;
    c.cb_last = last_key;
    c.count += (uint32_t)n;
    c.low = b < c.low ? b : c.low;
}

// lowest non-empty bucket (c.count > 0)
__device__ __forceinline__ uint32_t cold_lowest(const Cold &c, int lane)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long w1 = lane < COLD_NBK / 4096 ? cld64(&c.bm1()[lane]) : 0ull;
    const unsigned long long nz = __ballot(w1 != 0ull);
    const int wi = __ffsll((long long)nz) - 1;
    const unsigned long long word1 = shfl64(w1, wi);
    const int i0 = wi * 64 + (__ffsll((long long)word1) - 1);
    const unsigned long long word0 = cld64(&c.bm0()[i0]);
    return (uint32_t)i0 * 64u + (uint32_t)(__ffsll((long long)word0) - 1);
}

__device__ __forceinline__ void cold_mark_empty(Cold &c, uint32_t b, int lane)
{
    if (b == c.cb) c.cb = COLD_NONE;
    if (lane == 0) {
        c.bhead()[b] = COLD_NONE;
        c.btail()[b] = COLD_NONE;
        c.bcount()[b] = 0;
        c.bmeta()[b] = 0;
        const unsigned long long old = atomicAnd(&c.bm0()[b >> 6], ~(1ull << (b & 63)));
        if ((old & ~(1ull << (b & 63))) == 0ull) atomicAnd(&c.bm1()[b >> 12], ~(1ull << ((b >> 6) & 63)));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// in-LDS bitonic sort of srun[0, n) by key (n <= COLD_DRAIN); one wave
__device__ __forceinline__ void lds_sort(HeapEnt *srun, int n, int lane)
{
    int np = 64;
    while (np < n) np <<= 1;
    for (int j = n + lane; j < np; j += 64) { HeapEnt e; e.key = ~0ull; e.idx = 0; e.pad = 0; srun[j] = e; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // j is a power of two: the pair index is shifts and masks (the division by a run-time j cost most of the 59 k cycles a
    // 256-element sort took -- 12 % of the config-2 flood); all pairs of a pass are read before the first is written back
    for (int k = 2; k <= np; k <<= 1)
        for (int j = k >> 1, lj = 31 - __clz(j); j > 0; j >>= 1, --lj) {
            for (int t0 = 0; t0 < np / 2; t0 += 128) {
                HeapEnt a[2], bb[2];
                int ii[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t0 + 64 * u + lane;
                    ii[u] = ((t >> lj) << (lj + 1)) | (t & (j - 1));         // lower index of the pair
                    if (t < np / 2) { a[u] = srun[ii[u]]; bb[u] = srun[ii[u] + j]; }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t0 + 64 * u + lane;
                    if (t < np / 2) {
                        const bool up = ((ii[u] & k) == 0);
                        if ((a[u].key > bb[u].key) == up) { srun[ii[u]] = bb[u]; srun[ii[u] + j] = a[u]; }
                    }
                }
            }
            asm volatile("" ::: "memory");          // one wave: LDS operations execute in order, the next pass sees these stores
        }
}

// Move the <= COLD_DRAIN smallest elements of the lowest bucket into srun[0, taken), sorted.  srun must be empty.
// sorted_out (optional): the caller sorts -- *sorted_out tells whether srun[0, taken) still needs it (the work-group kernel
// sorts with all of its waves)
__device__ __forceinline__ int cold_drain(Cold &c, HeapEnt *srun, int *hist, int lane, int drain_n, bool *sorted_out = nullptr)
{
    const uint32_t b = cold_lowest(c, lane);
    if (b == c.cb) c.cb = COLD_NONE;                    // the drain rewrites this bucket's state
    const uint32_t cnt = cld32(&c.bcount()[b]);
    const uint32_t head = cld32(&c.bhead()[b]);
    const uint32_t meta = cld32(&c.bmeta()[b]);
    const uint32_t off0 = meta & ~COLD_UNSORTED;        // consumed prefix of the head segment
    const bool fifo = (meta & COLD_UNSORTED) == 0u;     // the bucket's list order is its key order
    int taken = 0;
    if (cnt <= (uint32_t)drain_n) {
        uint32_t seg = head, off = off0;
        while (seg != COLD_NONE) {
            const HeapEnt h = cld_ent(&c.pool()[seg]);
            const uint32_t n = (uint32_t)(h.key >> 32) - off;
            for (uint32_t j = lane; j < n; j += 64) srun[taken + j] = cld_ent(&c.pool()[seg + 1u + off + j]);
            taken += (int)n;
            seg = (uint32_t)h.key;
            off = 0;
        }
        cold_mark_empty(c, b, lane);
    } else if (fifo) {
        // ---- ascending list (plateaus: equal values, growing ages): the smallest elements are its prefix ----
        uint32_t seg = head, off = off0;
        while (taken < drain_n) {                       // cnt > drain_n: the list cannot run out
            const HeapEnt h = cld_ent(&c.pool()[seg]);
            const uint32_t n = (uint32_t)(h.key >> 32);
            uint32_t m = n - off;
            if (m > (uint32_t)(drain_n - taken)) m = (uint32_t)(drain_n - taken);
            for (uint32_t j = lane; j < m; j += 64) srun[taken + j] = cld_ent(&c.pool()[seg + 1u + off + j]);
            taken += (int)m;
            off += m;
            if (off == n) { seg = (uint32_t)h.key; off = 0; }
        }
        ++c.fifo_drains;
        if (lane == 0) {
            c.bhead()[b] = seg;
            c.bmeta()[b] = off;
            c.bcount()[b] = cnt - (uint32_t)taken;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // ---- select by key: min / max, then 64-bin histograms until a non-empty prefix of <= COLD_DRAIN exists ----
        // Bins are (key - lo) >> sh: any monotone binning selects a prefix in key order, and a shift costs nothing
        // (the proportional binning it replaces needed a 64-bit division per element and pass).  Every pass keeps
        // four independent loads per lane in flight instead of waiting for each 64-element row.
        unsigned long long lo = ~0ull, hi = 0ull;
        for (uint32_t seg = head; seg != COLD_NONE;) {
            const HeapEnt h = cld_ent(&c.pool()[seg]);
            const uint32_t n = (uint32_t)(h.key >> 32);
            const HeapEnt *el = &c.pool()[seg + 1u];
            uint32_t j = (seg == head ? off0 : 0u) + lane;
            for (; j + 192u < n; j += 256u) {
                const unsigned long long k0 = cld64(&el[j].key), k1 = cld64(&el[j + 64u].key), k2 = cld64(&el[j + 128u].key),
                                         k3 = cld64(&el[j + 192u].key);
                const unsigned long long a0 = k0 < k1 ? k0 : k1, a1 = k2 < k3 ? k2 : k3, b0 = k0 > k1 ? k0 : k1, b1 = k2 > k3 ? k2 : k3;
                const unsigned long long mn = a0 < a1 ? a0 : a1, mx = b0 > b1 ? b0 : b1;
                lo = mn < lo ? mn : lo;
                hi = mx > hi ? mx : hi;
            }
            for (; j < n; j += 64u) {
                const unsigned long long k = cld64(&el[j].key);
                lo = k < lo ? k : lo;
                hi = k > hi ? k : hi;
            }
            seg = (uint32_t)h.key;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long l2 = shfl64(lo, lane ^ o), h2 = shfl64(hi, lane ^ o);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        int sh = (hi - lo) ? 64 - __clzll((long long)(hi - lo)) - 6 : 0;       // (hi - lo) >> sh < 64
        if (sh < 0) sh = 0;
        int B = 0;
#define ISX_BIN_ADD(k) { const unsigned long long d_ = (k) - lo; if ((k) >= lo && (d_ >> sh) < 64ull) atomicAdd(&hist[(int)(d_ >> sh)], 1); }
        for (;;) {
            hist[lane] = 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            for (uint32_t seg = head; seg != COLD_NONE;) {
                const HeapEnt h = cld_ent(&c.pool()[seg]);
                const uint32_t n = (uint32_t)(h.key >> 32);
                const HeapEnt *el = &c.pool()[seg + 1u];
                uint32_t j = (seg == head ? off0 : 0u) + lane;
                for (; j + 192u < n; j += 256u) {
                    const unsigned long long k0 = cld64(&el[j].key), k1 = cld64(&el[j + 64u].key), k2 = cld64(&el[j + 128u].key),
                                             k3 = cld64(&el[j + 192u].key);
                    ISX_BIN_ADD(k0) ISX_BIN_ADD(k1) ISX_BIN_ADD(k2) ISX_BIN_ADD(k3)
                }
                for (; j < n; j += 64u) {
                    const unsigned long long k = cld64(&el[j].key);
                    ISX_BIN_ADD(k)
                }
                seg = (uint32_t)h.key;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            int mine = hist[lane], incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (lane >= o) incl += t;
            }
            // B = number of leading bins whose cumulative count stays <= drain_n
            const unsigned long long fits = __ballot(incl <= drain_n);
            B = (~fits) ? (__ffsll((long long)~fits) - 1) : 64;
            const int cumB = B > 0 ? __shfl(incl, B - 1) : 0;
            if (cumB > 0) break;
            // the first non-empty bin is bin B and it alone is too large: zoom into it (keys are unique, so a bin of
            // width 1 holds one element and this cannot happen with sh == 0)
            lo += (unsigned long long)B << sh;
            sh = sh > 6 ? sh - 6 : 0;
        }
#undef ISX_BIN_ADD
        // ---- extract bins < B into the run; compact the rest in place (write cursor trails the read cursor) ----
        // A row's loads are issued before any store of that row and later rows lie at higher addresses than anything
        // stored so far, so no load ever has to wait for a store: two rows are loaded ahead.
        // While the rest streams by, check whether it is in key order again: a list that went out of order through a few
        // small keys (a run eviction, an early element) loses exactly those to this drain, and everything that follows can
        // take the prefix path.  (Without this one stray append made every later drain of a growing plateau bucket a
        // select over the whole bucket: 2.5 M cycles per drain on a 5 M-voxel component.)
        bool rest_sorted = true;
        unsigned long long prev_last = 0ull;
        uint32_t wseg = head, wn = 0, wcap = cld_ent(&c.pool()[head]).idx, kept = 0;
        for (uint32_t seg = head; seg != COLD_NONE;) {
            const HeapEnt h = cld_ent(&c.pool()[seg]);
            const uint32_t n = (uint32_t)(h.key >> 32), nxt = (uint32_t)h.key;
            for (uint32_t base0 = (seg == head ? off0 : 0u); base0 < n; base0 += 128u) {
                HeapEnt e2[2];
                bool valid2[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t j = base0 + 64u * (uint32_t)u + lane;
                    valid2[u] = j < n;
                    e2[u].key = 0; e2[u].idx = 0; e2[u].pad = 0;
                    if (valid2[u]) e2[u] = cld_ent(&c.pool()[seg + 1u + j]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const HeapEnt e = e2[u];
                    const bool valid = valid2[u];
                    const bool take = valid && e.key >= lo && ((e.key - lo) >> sh) < (unsigned long long)B;
                    const unsigned long long tb = __ballot(take), kb = __ballot(valid && !take);
                    if (rest_sorted && kb) {
                        const unsigned long long below = kb & ((1ull << lane) - 1ull);
                        const unsigned long long pk = shfl64(e.key, below ? 63 - __clzll((long long)below) : lane);
                        if (__any(valid && !take && e.key < (below ? pk : prev_last))) rest_sorted = false;
                        prev_last = rdlane64(e.key, 63 - __clzll((long long)kb));
                    }
                    if (take) srun[taken + __popcll(tb & ((1ull << lane) - 1ull))] = e;
                    taken += __popcll(tb);
                    int krank = __popcll(kb & ((1ull << lane) - 1ull));
                    int remaining = __popcll(kb), done = 0;
                    while (remaining > 0) {
                        const int room = (int)(wcap - wn);
                        const int m = room < remaining ? room : remaining;
                        if (valid && !take && krank >= done && krank < done + m) c.pool()[wseg + 1u + wn + (uint32_t)(krank - done)] = e;
                        wn += (uint32_t)m;
                        done += m;
                        remaining -= m;
                        if (wn == wcap && remaining > 0) {
                            const HeapEnt wh = cld_ent(&c.pool()[wseg]);
                            if (lane == 0) { HeapEnt t2 = wh; t2.key = (wh.key & 0xFFFFFFFFull) | ((unsigned long long)wcap << 32); c.pool()[wseg] = t2; }
                            wseg = (uint32_t)wh.key;
                            wn = 0;
                            wcap = cld_ent(&c.pool()[wseg]).idx;
                        }
                    }
                    kept += (uint32_t)__popcll(kb);
                }
            }
            seg = nxt;
        }
        if (lane == 0) {
            HeapEnt t2;
            t2.key = (unsigned long long)COLD_NONE | ((unsigned long long)wn << 32);     // truncate the list here
            t2.idx = wcap;
            t2.pad = 0;
            c.pool()[wseg] = t2;
            c.btail()[b] = wseg;
            c.bfill()[b] = wn | ((uint32_t)(31 - __clz((int)wcap)) << 24);
            c.bcount()[b] = kept;
            c.bmeta()[b] = rest_sorted ? 0u : COLD_UNSORTED;     // compacted from the start of the head segment
            if (rest_sorted) c.blast()[b] = prev_last;           // the list's last element: what the next append is checked against
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (sorted_out) *sorted_out = fifo;
    else if (!fifo) lds_sort(srun, taken, lane);
    c.count -= (uint32_t)taken;
    c.low = c.count > 0 ? cold_lowest(c, lane) : (uint32_t)COLD_NBK;
    return taken;
}

__global__ void arena_need_kernel(const unsigned long long *comp_cap, uint32_t n_comp, unsigned long long coop_min,
                                  unsigned long long cold_min, unsigned long long *need)
{
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_comp) return;
    unsigned long long sz = c < n_comp ? comp_cap[c] : 0ull;
    need[c] = (sz >= coop_min && sz >= cold_min) ? (unsigned long long)COLD_META_ENTS + 4ull * sz + 5ull * COLD_NBK + 4096ull : 0ull;
}

// launch order of the window kernel: largest component first (work-groups are dispatched in block order and a big component
// keeps its CU for the whole flood -- started last, the largest one alone set the time of a structured volume)
__global__ void comp_order_key_kernel(const unsigned long long *comp_off, uint32_t n_comp, unsigned long long coop_min,
                                      unsigned long long *keys)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_comp) return;
    const unsigned long long sz = comp_off[c + 1] - comp_off[c];
    const unsigned long long inv = 0xFFFFFFFFull - (sz < 0xFFFFFFFFull ? sz : 0xFFFFFFFFull);
    keys[c] = sz >= coop_min ? (inv << 32) | c : ~0ull;            // components of the one-lane kernel sort last and are skipped
}

#include "ws_flood_window.h"
#include "ws_flood_group.h"

// Flood given raveled seed indices (device u32) in list order (label = position + 1).
// parent/sizes: CCL of `mask` if the caller already has it, else nullptr.
static int flood_impl(isx_ctx *ctx, const float *aff, const uint32_t *ridx, uint32_t n_seeds,
                      const uint8_t *mask, uint32_t *out, const int64_t pshape[3],
                      const uint32_t *parent_in, const uint32_t *sizes_in, const float *scale, const uint32_t *aff_mm = nullptr)
{
    // aff_mm: ordered-uint {min, max, valid} of aff on the device when the caller can derive them (aff_range_kernel), else nullptr
    if (n_seeds == 0) {
        set_error("zero seeds: the reference raises here (np.apply_along_axis, watershed.py:50)");
        return ISX_ERR_NO_SEEDS;
    }
    int64_t PN = pshape[0] * pshape[1] * pshape[2];
    Scratch<uint32_t> parent_own(ctx, parent_in ? 1 : PN), sizes_own(ctx, sizes_in ? 1 : PN);
    if (!parent_own.ok() || !sizes_own.ok()) return ISX_ERR_NOMEM;
    const uint32_t *parent = parent_in, *sizes = sizes_in;
    if (!parent_in) {
        ProfScope ps(ctx, ISX_PROF_WS_CCL);
        ISX_TRY(ccl_impl(ctx, mask, pshape, parent_own.p, sizes_own.p));
        parent = parent_own.p;
        sizes = sizes_own.p;
    }
    ProfScope ps(ctx, ISX_PROF_WS_FLOOD);
    Scratch<unsigned long long> keys(ctx, n_seeds), keys_sorted(ctx, n_seeds), comp_cap(ctx, n_seeds + 1),
        comp_off(ctx, n_seeds + 1);
    Scratch<uint32_t> head(ctx, n_seeds), cid(ctx, n_seeds), comp_first(ctx, n_seeds);
    if (!keys.ok() || !keys_sorted.ok() || !comp_cap.ok() || !comp_off.ok() || !head.ok() || !cid.ok() ||
        !comp_first.ok())
        return ISX_ERR_NOMEM;
    unsigned sg = cdiv(n_seeds, TPB);
    seed_label_kernel<<<sg, TPB, 0, ctx->stream>>>(ridx, n_seeds, parent, out, keys.p);
    ISX_CHECK_LAUNCH();
    ISX_TRY(sort_u64(ctx, (const uint64_t *)keys.p, (uint64_t *)keys_sorted.p, n_seeds));
    comp_head_kernel<<<sg, TPB, 0, ctx->stream>>>(keys_sorted.p, n_seeds, head.p);
    ISX_CHECK_LAUNCH();
    ISX_TRY(exclusive_scan_u32(ctx, head.p, cid.p, n_seeds));
    // number of components = cid[last] + head[last]
    uint32_t last_cid = 0, last_head = 0;
    ISX_HIP(hipMemcpyAsync(&last_cid, cid.p + (n_seeds - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipMemcpyAsync(&last_head, head.p + (n_seeds - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    uint32_t n_comp = last_cid + last_head;
    ISX_HIP(hipMemsetAsync(comp_cap.p, 0, (size_t)(n_comp + 1) * 8, ctx->stream));
    comp_table_kernel<<<sg, TPB, 0, ctx->stream>>>(keys_sorted.p, head.p, cid.p, n_seeds, sizes,
                                                     comp_first.p, comp_cap.p);
    ISX_CHECK_LAUNCH();
    ISX_TRY(exclusive_scan_u64(ctx, (const uint64_t *)comp_cap.p, (uint64_t *)comp_off.p, n_comp + 1));
    unsigned long long total = 0;
    ISX_HIP(hipMemcpyAsync(&total, comp_off.p + n_comp, 8, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    // cold arenas for the components large enough to need one (sizes are the heap capacities)
    Scratch<unsigned long long> arena_need(ctx, n_comp + 1), arena_off(ctx, n_comp + 1);
    if (!arena_need.ok() || !arena_off.ok()) return ISX_ERR_NOMEM;
    // ISX_FLOOD_* knobs (tests' stress modes, tuning runs): read once into the context (isx_ctx_create / isx_ctx_reload_knobs);
    // a negative field = the library default
    const isx_flood_tune &ft = ctx->ftune;
    auto knob = [](long long v, long long dflt) { return v >= 0 ? v : dflt; };
    unsigned long long coop_min = (unsigned long long)knob(ft.coop_min, COOP_MIN_SIZE);         // tests: force a path
    unsigned long long cold_min = (unsigned long long)knob(ft.cold_min, COLD_MIN_SIZE);
    int run_limit = (int)knob(ft.run_limit, COOP_SCAP - 6 * 64), drain_n = (int)knob(ft.drain, COLD_DRAIN);      // tests shrink these to stress the cold tier
    ISX_REQUIRE(run_limit >= 8 && run_limit <= COOP_SCAP - 6 * 64 && drain_n >= 4 && drain_n <= COLD_DRAIN && drain_n <= run_limit,
                "ISX_FLOOD_RUN_LIMIT / ISX_FLOOD_DRAIN out of range");
    // A component without an arena whose LDS run overflows finishes on the one-lane heap; on structured volumes those
    // mid-size components were the critical path (tiled 512^3 blobs: 263 ms with arenas from 32 768 voxels, 103 ms from
    // 4 096).  An arena costs 7.4 MB of bucket tables whatever the component's size, so the threshold doubles until the
    // arenas fit the budget (a volume of many thousand mid-size components keeps the larger ones only).
    const unsigned long long arena_budget = ft.arena_mb >= 0 ? (unsigned long long)ft.arena_mb << 20 : COLD_ARENA_BUDGET;
    unsigned long long arena_total = 0;
    for (;;) {
        arena_need_kernel<<<cdiv(n_comp + 1, TPB), TPB, 0, ctx->stream>>>(comp_cap.p, n_comp, coop_min, cold_min, arena_need.p);
        ISX_CHECK_LAUNCH();
        ISX_TRY(exclusive_scan_u64(ctx, (const uint64_t *)arena_need.p, (uint64_t *)arena_off.p, n_comp + 1));
        ISX_HIP(hipMemcpyAsync(&arena_total, arena_off.p + n_comp, 8, hipMemcpyDeviceToHost, ctx->stream));
        ISX_HIP(hipStreamSynchronize(ctx->stream));
        // never above the round-1 threshold: the large components keep their arenas whatever they cost
        if (arena_total * sizeof(HeapEnt) <= arena_budget || ft.cold_min >= 0 || cold_min >= 32768ull) break;
        cold_min *= 2;
    }
    Scratch<HeapEnt> pool(ctx, (size_t)total + 1);
    Scratch<HeapEnt> arena(ctx, (size_t)arena_total + 1);
    // large components: the ordered speculative window (ws_flood_window.h); ISX_FLOOD_WIN / ISX_FLOOD_LIST: tiny windows / lists (tests)
    const int wcap = (int)knob(ft.win, WIN_W), tcap = (int)knob(ft.list, WIN_T);
    const int max_bursts = (int)knob(ft.bursts, 3);
    // + (ISX_FLOOD_GWATCHDOG << 16): sub-round limit of the work-group kernel (tests of its hand-over)
    const int group_bursts = (max_bursts & 0x7FF) | ((ft.gwatchdog & 0x7FFF) << 16);
    // the largest components: one WORK-GROUP each (ws_flood_group.h; window of 512 episodes, lists of up to 192 discoveries).
    // ISX_FLOOD_GROUP_MIN: smallest component for it (0 = never: everything >= coop_min on the single-wave window kernel);
    // ISX_FLOOD_GWIN / ISX_FLOOD_GLIST shrink its window / lists (tests)
    unsigned long long group_min = (unsigned long long)knob(ft.group_min, GROUP_MIN_SIZE);
    if (group_min == 0) group_min = ~0ull;
    if (group_min != ~0ull && group_min < coop_min) group_min = coop_min;
    // ISX_FLOOD_GSTEPS: pops a wave may make between two work-group barriers (1 = one); ISX_FLOOD_GMINSTEPS: steps every wave
    // makes before the front's "ready" stops it
    const int gwcap = (int)knob(ft.gwin, GRP_W), gtcap = (int)knob(ft.glist, GRP_T), gsteps = (int)knob(ft.gsteps, 64), gminsteps = (int)knob(ft.gminsteps, 4);
    ISX_REQUIRE(wcap >= 1 && wcap <= WIN_W && tcap >= 6 && tcap <= WIN_T, "ISX_FLOOD_WIN / ISX_FLOOD_LIST out of range");
    ISX_REQUIRE(gwcap >= 1 && gwcap <= GRP_W && gtcap >= 6 && gtcap <= GRP_T && gsteps >= 1 && gsteps <= 255 && gminsteps >= 1 && gminsteps <= 255, "ISX_FLOOD_GWIN / ISX_FLOOD_GLIST / ISX_FLOOD_GSTEPS out of range");
    const bool trace_on = !ft.trace.empty();        // development: commit order of every component's roots (ISX_FLOOD_TRACE=<file>)
    Scratch<uint32_t> claim(ctx, trace_on ? (size_t)(3 * (total + 1)) : 1);
    Scratch<unsigned long long> vstate(ctx, (size_t)PN);
    Scratch<unsigned long long> stats(ctx, 64);
    if (!pool.ok() || !claim.ok() || !arena.ok() || !stats.ok() || !vstate.ok()) return ISX_ERR_NOMEM;
    ISX_HIP(hipMemsetAsync(stats.p, 0, 64 * 8, ctx->stream));
    float s[3] = {1.f, 1.f, 1.f};
    if (scale) for (int a = 0; a < 3; ++a) s[a] = fabsf(scale[a]);
    // value range of the (scaled) edge values -> linear bucket map of the cold tier (any monotone map is correct; this
    // one fills the buckets evenly).  ISX_FLOOD_BUCKETS: number of buckets spread over the range (tests / tuning).
    float bk_lo = 0.0f, bk_inv = 0.0f;
    {
        Scratch<uint32_t> mm(ctx, 2);
        if (!mm.ok()) return ISX_ERR_NOMEM;
        uint32_t hmm[3] = {0u, 0u, 0u};
        if (aff_mm) {
            ISX_HIP(hipMemcpyAsync(hmm, aff_mm, 12, hipMemcpyDeviceToHost, ctx->stream));
            ISX_HIP(hipStreamSynchronize(ctx->stream));
        }
        const bool derived = hmm[2] != 0;
        const uint32_t d0 = hmm[0], d1 = hmm[1];
        if (!derived || ctx->ftune.ws_check) {
            const uint32_t init[2] = {0xFFFFFFFFu, 0u};
            ISX_HIP(hipMemcpyAsync(mm.p, init, 8, hipMemcpyHostToDevice, ctx->stream));
            minmax_kernel<<<1024, TPB, 0, ctx->stream>>>(aff, 3 * PN, mm.p);
            ISX_CHECK_LAUNCH();
            ISX_HIP(hipMemcpyAsync(hmm, mm.p, 8, hipMemcpyDeviceToHost, ctx->stream));
            ISX_HIP(hipStreamSynchronize(ctx->stream));
            ISX_REQUIRE(!derived || (hmm[0] == d0 && hmm[1] == d1), "internal: derived affinity range %08x..%08x, measured %08x..%08x", d0, d1, hmm[0], hmm[1]);
        }
        auto unord = [](uint32_t u) { uint32_t b = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u; float f; memcpy(&f, &b, 4); return f; };
        const float rlo = unord(hmm[0]), rhi = unord(hmm[1]);
        const float smin = std::min(s[0], std::min(s[1], s[2])), smax = std::max(s[0], std::max(s[1], s[2]));
        const float vlo = std::min(std::min(rlo * smin, rlo * smax), 0.0f), vhi = std::max(std::max(rhi * smin, rhi * smax), vlo);
        const int nbuckets = (int)knob(ft.buckets, 4096);      // (16384 until round 3: fewer, larger drains are cheaper for both kernels; config 2: 2.55 -> 2.43 s)
        ISX_REQUIRE(nbuckets >= 1 && nbuckets <= COLD_NBK, "ISX_FLOOD_BUCKETS out of range");
        bk_lo = vlo;
        bk_inv = (vhi > vlo && std::isfinite(vhi - vlo)) ? (float)nbuckets / (vhi - vlo) : 0.0f;
    }
    // small components: one lane each; large ones: one wave or one work-group each (every kernel is launched over all components and
    // picks its own by size -- no host round trip to split the list).  Components are disjoint, so the kernels are independent: the
    // one-lane kernel runs on the main stream beside the side streams' tiers (in front of them it was 4.4 ms of a structured 512^3
    // volume's 56, on the critical path for nothing)
    {
        flood_state_init_kernel<<<cdiv(PN, TPB), TPB, 0, ctx->stream>>>(mask, out, vstate.p, PN);
        ISX_CHECK_LAUNCH();
        Scratch<unsigned long long> okeys(ctx, n_comp), order(ctx, n_comp);
        if (!okeys.ok() || !order.ok()) return ISX_ERR_NOMEM;
        comp_order_key_kernel<<<cdiv(n_comp, TPB), TPB, 0, ctx->stream>>>(comp_off.p, n_comp, coop_min, okeys.p);
        ISX_CHECK_LAUNCH();
        ISX_TRY(sort_u64(ctx, (const uint64_t *)okeys.p, (uint64_t *)order.p, n_comp));
        const bool side = ft.group_stream != 0;
        // Between a fork and its join an early return (a failed launch, a failed second fork) would hand the Scratch buffers
        // below back to the context's cache while kernels queued on a side stream may still use them: on that path the guard
        // waits for the side streams first.  The normal path joins by events and disarms it.
        struct SideGuard {
            isx_ctx *c; bool armed[2] = {false, false};
            ~SideGuard() { for (int i = 0; i < 2; ++i) if (armed[i] && c->aux_streams[i]) (void)hipStreamSynchronize(c->aux_streams[i]); }
        } guard{ctx};
        if (group_min != ~0ull) {
            // the giants run beside the mid-size components (other CUs): a side stream, joined below
            // (ISX_FLOOD_GROUP_STREAM=0: same stream, one kernel after the other -- structured 512^3 seed 3, one 345 k-voxel
            // giant among 2 900 mid-size components: 150 ms against 104 ms side by side)
            if (side) { ISX_TRY(ctx->aux_fork()); guard.armed[0] = true; }
            flood_group_kernel<<<n_comp, GRP_W, 0, side ? ctx->aux_streams[0] : ctx->stream>>>(order.p, aff, mask, out, vstate.p, keys_sorted.p, n_seeds, comp_first.p,
                                                                   comp_off.p, n_comp, pool.p, arena_off.p, arena.p, pshape[1], pshape[2], PN,
                                                                   s[0], s[1], s[2], group_min, run_limit, drain_n, gwcap, gtcap, group_bursts,
                                                                   bk_lo, bk_inv, stats.p, ctx->ws_shard_rank, ctx->ws_shard_world, gsteps | (gminsteps << 8));
            ISX_CHECK_LAUNCH();
        }
        // two tiers of the single-wave kernel: components below small_max voxels on the small instantiation (~30 KB of LDS:
        // five per CU instead of one), the others on the large one, side by side on their own streams
        unsigned long long small_max = (unsigned long long)knob(ft.small_max, WIN_SMALL_MAX);      // 0: no small tier
        if (small_max < coop_min) small_max = coop_min;
        if (group_min != ~0ull && small_max > group_min) small_max = group_min;
        uint32_t *trace_p = trace_on ? claim.p : nullptr;
        const bool two = small_max > coop_min && side;
        if (two) { ISX_TRY(ctx->aux_fork(1)); guard.armed[1] = true; }
        flood_window_kernel<WIN_W, WIN_T, WIN_RCAP, WIN_PEND><<<n_comp, 64, 0, two ? ctx->aux_streams[1] : ctx->stream>>>(
            order.p, aff, mask, out, vstate.p, keys_sorted.p, n_seeds, comp_first.p, comp_off.p, n_comp, pool.p, arena_off.p, arena.p, pshape[1],
            pshape[2], PN, s[0], s[1], s[2], small_max, run_limit, drain_n, wcap, tcap, max_bursts, bk_lo, bk_inv, stats.p, ctx->ws_shard_rank,
            ctx->ws_shard_world, trace_p, group_min);
        ISX_CHECK_LAUNCH();
        flood_kernel<<<cdiv(n_comp, 64), 64, 0, ctx->stream>>>(aff, mask, out, keys_sorted.p, n_seeds,
                                                              comp_first.p, comp_off.p, n_comp, pool.p,
                                                              pshape[1], pshape[2], PN, s[0], s[1], s[2], coop_min,
                                                              ctx->ws_shard_rank, ctx->ws_shard_world);
        ISX_CHECK_LAUNCH();
        if (small_max > coop_min) {
            const int s_wcap = wcap < WIN_SW ? wcap : WIN_SW, s_tcap = tcap < WIN_ST ? tcap : WIN_ST;
            const int s_run = run_limit < WIN_SRCAP - 64 ? run_limit : WIN_SRCAP - 64, s_drain = drain_n < 256 ? (drain_n < s_run ? drain_n : s_run) : 256;
            flood_window_kernel<WIN_SW, WIN_ST, WIN_SRCAP, WIN_SPEND><<<n_comp, 64, 0, ctx->stream>>>(
                order.p, aff, mask, out, vstate.p, keys_sorted.p, n_seeds, comp_first.p, comp_off.p, n_comp, pool.p, arena_off.p, arena.p, pshape[1],
                pshape[2], PN, s[0], s[1], s[2], coop_min, s_run, s_drain, s_wcap, s_tcap, max_bursts, bk_lo, bk_inv, stats.p, ctx->ws_shard_rank,
                ctx->ws_shard_world, trace_p, small_max);
            ISX_CHECK_LAUNCH();
        }
        if (two) { ISX_TRY(ctx->aux_join(1)); guard.armed[1] = false; }
        if (group_min != ~0ull && side) { ISX_TRY(ctx->aux_join(0)); guard.armed[0] = false; }

        if (trace_on) {
            const char *tp = ft.trace.c_str();
            ISX_HIP(hipStreamSynchronize(ctx->stream));
            std::vector<uint32_t> hl(3 * ((size_t)total + 1)), hf(n_comp);
            std::vector<unsigned long long> ho(n_comp + 1), hk(n_seeds);
            ISX_HIP(hipMemcpy(hl.data(), claim.p, 3 * ((size_t)total + 1) * 4, hipMemcpyDeviceToHost));
            ISX_HIP(hipMemcpy(hf.data(), comp_first.p, (size_t)n_comp * 4, hipMemcpyDeviceToHost));
            ISX_HIP(hipMemcpy(ho.data(), comp_off.p, (size_t)(n_comp + 1) * 8, hipMemcpyDeviceToHost));
            ISX_HIP(hipMemcpy(hk.data(), keys_sorted.p, (size_t)n_seeds * 8, hipMemcpyDeviceToHost));
            FILE *f = fopen(tp, "wb");
            if (f) {
                unsigned long long hdr[3] = {n_comp, n_seeds, total};
                fwrite(hdr, 8, 3, f); fwrite(ho.data(), 8, n_comp + 1, f); fwrite(hf.data(), 4, n_comp, f); fwrite(hk.data(), 8, n_seeds, f);
                fwrite(hl.data(), 4, 3 * ((size_t)total + 1), f); fclose(f);
            }
        }
    }
    {
        unsigned long long hs[64];
        ISX_HIP(hipMemcpyAsync(hs, stats.p, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
        ISX_HIP(hipStreamSynchronize(ctx->stream));
#ifdef ISX_FLOOD_CHECK          // these slots carry the phase timers in the ISX_FLOOD_PROF build
        if (hs[40]) fprintf(stderr, "[isx flood/window] ORDER CHECK failed first at tag %llu sub-round %llu nw %llu run %llu: key %016llx !< next %016llx at p %llu (1 window, 2 window/run, 3 run: %llu)\n   window tail (last first): %016llx %016llx %016llx %016llx | run head: %016llx %016llx %016llx %016llx | cold %llu fifo drains %llu seeds %llu staged %llu\n   [tier check] last hot insert key %016llx hot_max then %016llx reason %llu (1 below hot max, 2 cold free) nw %llu run %llu\n", hs[40], hs[41], hs[42], hs[43], hs[44], hs[45], hs[46], hs[47], hs[20], hs[21], hs[22], hs[23], hs[24], hs[25], hs[26], hs[27], hs[28], hs[29], hs[30], hs[31], hs[20], hs[21], hs[22], hs[23], hs[24]);
#endif
        // the watchdog dump first: a tripped kernel has overwritten slots 20..31 with its front state (the guard / skip reports
        // below live in some of the same slots and would be misread)
        if (hs[9] && group_min != ~0ull)
            fprintf(stderr, "[isx flood/group] watchdog state (if it was the work-group kernel): nw %llu run %llu | free chunks %llu reserve %llu free slots %llu dirty %llu forced %llu | front: info %08llx (np %llu npop %llu started %llu idle %llu) mx %08llx resmin %08llx slot %llu key %016llx | next key %016llx qhead %016llx | episodes %llu started %llu idle %llu with-spare %llu chunks-held %llu entries %llu | pops %llu cold %llu\n",
                    hs[24], hs[25], hs[26] & 0xFFFF, (hs[26] >> 16) & 0xFFFF, (hs[26] >> 32) & 0xFFFF, (hs[26] >> 48) & 0xFF, hs[26] >> 56, hs[27] & 0xFFFFFFFF,
                    hs[27] & 0xFF, (hs[27] >> 8) & 0xFF, (hs[27] >> 16) & 1, (hs[27] >> 17) & 1, hs[27] >> 32, hs[10] & 0xFFFFFFFF, (hs[10] >> 32) & 0xFFF, hs[28], hs[23], hs[29],
                    hs[30] & 0xFFFF, (hs[30] >> 16) & 0xFFFF, (hs[30] >> 32) & 0xFFFF, hs[30] >> 48, (hs[31] >> 32) & 0xFFFF, hs[31] >> 48, hs[31] & 0xFFFFFFFF, hs[10] >> 44);
        if (hs[9] && group_min != ~0ull)
            fprintf(stderr, "[isx flood/group] window head keys %016llx %016llx %016llx %016llx | run head keys %016llx %016llx %016llx %016llx\n", hs[40], hs[41], hs[42], hs[43], hs[44], hs[45], hs[46], hs[47]);
        if (hs[9]) {
            fprintf(stderr, "[isx flood/window] WATCHDOG: %llu component(s) exceeded the sub-round limit; front: nw %llu run %llu cold %llu cur|meta %016llx key %016llx "
                            "qhead %016llx maxnew|nstage %016llx pops %llu | pos1 key %016llx cur|meta %016llx wh %llu npfree %llu\n", hs[9], hs[24], hs[25], hs[26], hs[27], hs[28], hs[29], hs[30], hs[31], hs[20], hs[21], hs[22], hs[23]);
            fprintf(stderr, "[isx flood/window] sub-rounds %llu pops %llu executed %llu rows %llu hot_ins %llu mid_ins %llu cold_app %llu drains %llu evicted %llu squashes %llu (bound %llu) "
                            "squashed eps %llu forced %llu\n", hs[0], hs[1], hs[2], hs[13], hs[3], hs[7], hs[4], hs[5], hs[6], hs[11], hs[14], hs[12], hs[15]);
            set_error("flood watchdog tripped (internal error)");
            return ISX_ERR_INTERNAL;
        }
        if (hs[22]) fprintf(stderr, "[isx flood/group] serial continuation requested %llu time(s) and SKIPPED (development flag); last: mode %llu sub-round %llu component size %llu\n",
                            hs[22], hs[23] & 0xFF, (hs[23] >> 8) & 0xFFFFFF, hs[23] >> 32);
        if (hs[20]) {
            fprintf(stderr, "[isx flood/group] GUARD: %llu report(s), site %llu (1-3: voxel index outside the volume, 100+: hand-over sanity, 200: run empty with cold elements, 3xx: order check tag) "
                            "sub-round %llu where %llu lane %llu keys %016llx %016llx nw|run %016llx\n", hs[20], hs[21], hs[10] & 0xFFFFFFFF, (hs[10] >> 32) & 0xFF, hs[10] >> 40, hs[40], hs[41], hs[42]);
            set_error("flood group kernel produced an invalid voxel index (internal error)");
            return ISX_ERR_INTERNAL;
        }
        if (hs[16]) {
            fprintf(stderr, "[isx flood] TIER INVARIANT: %llu run eviction(s) left the hot maximum above the lowest cold bucket\n", hs[16]);
            set_error("flood tier invariant violated (internal error)");
            return ISX_ERR_INTERNAL;
        }
        if (!ft.stats) return ISX_OK;
        fprintf(stderr, "[isx flood/window] sub-rounds %llu pops %llu (%.2f/sub-round) executed %llu (useful %.2f) rows %llu hot_ins %llu mid_ins %llu cold_app %llu drains %llu (prefix %llu) "
                        "evicted %llu squashes %llu (bound %llu) squashed eps %llu forced %llu serial_handover %llu\n",
                hs[0], hs[1], hs[0] ? (double)hs[1] / hs[0] : 0.0, hs[2], hs[2] ? (double)hs[1] / hs[2] : 0.0, hs[13], hs[3], hs[7], hs[4], hs[5], hs[10],
                hs[6], hs[11], hs[14], hs[12], hs[15], hs[8]);
        if (group_min != ~0ull)
            fprintf(stderr, "[isx flood/group] (both kernels share the counters above) commit bursts %llu list reloads %llu; of the work-group kernel alone: rounds %llu pops %llu drains %llu (%llu as the prefix of a list in key order)\n", hs[17], hs[18], hs[63], hs[62], hs[60], hs[61]);
#ifdef ISX_FLOOD_PROF
        // (slots 32-40 are shared with the window kernel's timers: on a volume that uses both kernels build with -DISX_FLOOD_PROF_GROUP_ONLY;
        //  per round of the work-group kernel, summed over its work-groups)
        const double gr = hs[63] ? (double)hs[63] : 1.0;
        if (group_min != ~0ull)
            fprintf(stderr, "[isx flood/group] cycles/sub-round of lane 0: (S0)->C %.0f | C loop %.0f | wait (S5) %.0f | squash %.0f | to (E0) %.0f | order (E1-E3a) %.0f | verify %.0f | labels %.0f | enqueue %.0f | "
                            "burst-loop exit %.0f | drain call %.0f | drain sort %.0f | refill %.0f | loop end %.0f || C: claims+loads wait %.0f ; flush %.0f\n",
                    (double)hs[32] / gr, (double)hs[43] / gr, (double)hs[34] / gr, (double)hs[35] / gr, (double)hs[44] / gr, (double)hs[33] / gr, (double)hs[36] / gr,
                    (double)hs[37] / gr, (double)hs[38] / gr, (double)hs[45] / gr, (double)hs[41] / gr, (double)hs[42] / gr, (double)hs[39] / gr, (double)hs[40] / gr, (double)hs[46] / gr, (double)hs[47] / gr);
        if (group_min != ~0ull)
            fprintf(stderr, "[isx flood/group] bursts cut by: front not ready (rounds) %llu | not started %llu | running %llu | bound %llu | pend full %llu | whole window %llu ; window fill at first burst %.1f ; flushes %llu (one-bucket %llu, 64-element appends %llu)\n",
                    hs[48], hs[49], hs[50], hs[51], hs[52], hs[53], hs[63] ? (double)hs[54] / hs[63] : 0.0, hs[57], hs[55], hs[56]);
        fprintf(stderr, "[isx flood/window] cycles/sub-round: loop+evict %.0f | scan %.0f | trip %.0f | lists %.0f | squash %.0f | commit: decide %.0f labels %.0f enqueue %.0f | refill %.0f\n",
                (double)hs[32] / hs[0], (double)hs[33] / hs[0], (double)hs[34] / hs[0], (double)hs[35] / hs[0], (double)hs[36] / hs[0],
                (double)hs[37] / hs[0], (double)hs[38] / hs[0], (double)hs[39] / hs[0], (double)hs[40] / hs[0]);
#endif
        return ISX_OK;
    }
}

// W2 shared by the public entry and the fused path.  keys_sorted must hold `cap` entries.
static int seeds_impl(isx_ctx *ctx, const float *cent, const int64_t shape[3], int smooth, float thr,
                      unsigned long long *keys_sorted, uint32_t cap, uint32_t *count_host)
{
    int64_t n = shape[0] * shape[1] * shape[2];
    ISX_REQUIRE(n < (int64_t)0xFFFFFFF0ll, "volume too large for 32-bit voxel indices");
    Scratch<float> g(ctx, smooth ? n : 1), tmp(ctx, smooth ? n : 1);
    Scratch<uint32_t> flags(ctx, 2 + PEAK_FLAGS * PEAK_FLAG_PITCH);          // [nonpeak, count, spread flag words]
    Scratch<unsigned long long> keys(ctx, cap);
    if (!g.ok() || !tmp.ok() || !flags.ok() || !keys.ok()) return ISX_ERR_NOMEM;
    const float *src = cent;
    if (smooth) {
        const int32_t sig[3] = {0, 1, 1};
        ISX_TRY(gaussian_impl(ctx, cent, g.p, shape, sig, tmp.p));
        src = g.p;
    }
    ISX_HIP(hipMemsetAsync(flags.p, 0, (2 + PEAK_FLAGS * PEAK_FLAG_PITCH) * sizeof(uint32_t), ctx->stream));
    const int ntz = (int)((shape[0] + 3) / 4), nty = (int)((shape[1] + 7) / 8), ntx = (int)((shape[2] + 63) / 64);
    peak_kernel<<<(unsigned)(ntz * nty * ntx), 256, 0, ctx->stream>>>(src, (int)shape[0], (int)shape[1], (int)shape[2], thr,
                                                                    flags.p + 2, keys.p, flags.p + 1, cap, nty, ntx);
    ISX_CHECK_LAUNCH();
    peak_flag_kernel<<<1, 64, 0, ctx->stream>>>(flags.p + 2, flags.p);
    ISX_CHECK_LAUNCH();
    uint32_t fl[2];
    ISX_HIP(hipMemcpyAsync(fl, flags.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    uint32_t cnt = fl[0] ? fl[1] : 0u;          // all voxels "peaks" => trivial image => none
    *count_host = cnt;
    uint32_t m = cnt < cap ? cnt : cap;
    ISX_TRY(sort_u64(ctx, (const uint64_t *)keys.p, (uint64_t *)keys_sorted, m));
    return ISX_OK;
}

}  // namespace isx

using namespace isx;

extern "C" {

int isx_ws_set_shard(isx_ctx *ctx, int rank, int world)
{
    ISX_REQUIRE(ctx && world >= 1 && rank >= 0 && rank < world, "shard (%d of %d)", rank, world);
    ctx->ws_shard_rank = rank;
    ctx->ws_shard_world = world;
    return ISX_OK;
}

int isx_ws_gaussian(isx_ctx *ctx, const float *img, float *out, const int64_t shape[3],
                    const int32_t sigma[3])
{
    ISX_REQUIRE(ctx && img && out && shape && sigma, "null argument");
    // every pass reads neighbours that other work-groups of the same pass write: never in place (itersegx.h)
    ISX_REQUIRE((const float *)out != img, "isx_ws_gaussian: img and out must be different buffers");
    ISX_HIP(hipSetDevice(ctx->device));
    ProfScope ps(ctx, ISX_PROF_WS_FILTERS);
    int64_t n = shape[0] * shape[1] * shape[2];
    Scratch<float> tmp(ctx, n);
    if (!tmp.ok()) return ISX_ERR_NOMEM;
    return gaussian_impl(ctx, img, out, shape, sigma, tmp.p);
}

int isx_ws_threshold_otsu(isx_ctx *ctx, const float *img, int64_t n, float *thresh)
{
    ISX_REQUIRE(ctx && img && thresh && n > 0, "bad argument");
    ISX_HIP(hipSetDevice(ctx->device));
    Scratch<float> t(ctx, 1);
    if (!t.ok()) return ISX_ERR_NOMEM;
    ISX_TRY(otsu_impl(ctx, img, n, t.p));
    ISX_HIP(hipMemcpyAsync(thresh, t.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    return ISX_OK;
}

int isx_ws_mask(isx_ctx *ctx, const float *img, const int64_t shape[3], uint8_t *mask, float *thresh)
{
    ISX_REQUIRE(ctx && img && shape && mask, "null argument");
    ISX_HIP(hipSetDevice(ctx->device));
    ProfScope ps(ctx, ISX_PROF_WS_FILTERS);
    int64_t n = shape[0] * shape[1] * shape[2];
    Scratch<float> sm(ctx, n), tmp(ctx, n), t(ctx, 1);
    if (!sm.ok() || !tmp.ok() || !t.ok()) return ISX_ERR_NOMEM;
    ISX_TRY(smooth_otsu(ctx, img, shape, sm.p, tmp.p, t.p));
    launch_mask(ctx, img, t.p, shape[0], shape[1], shape[2], 0, mask);
    ISX_CHECK_LAUNCH();
    if (thresh) {
        ISX_HIP(hipMemcpyAsync(thresh, t.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        ISX_HIP(hipStreamSynchronize(ctx->stream));
    }
    return ISX_OK;
}

int isx_ws_seeds(isx_ctx *ctx, const float *cent, const int64_t shape[3], int smooth,
                 float threshold_abs, int64_t *coords, int64_t cap, int64_t *count)
{
    ISX_REQUIRE(ctx && cent && shape && count && cap >= 0 && (coords || cap == 0), "bad argument");
    ISX_HIP(hipSetDevice(ctx->device));
    ProfScope ps(ctx, ISX_PROF_WS_FILTERS);
    uint32_t c32 = (uint32_t)(cap > 0x7FFFFFF0ll ? 0x7FFFFFF0ll : cap);
    Scratch<unsigned long long> ks(ctx, c32 ? c32 : 1);
    if (!ks.ok()) return ISX_ERR_NOMEM;
    uint32_t cnt = 0;
    ISX_TRY(seeds_impl(ctx, cent, shape, smooth, threshold_abs, ks.p, c32, &cnt));
    *count = cnt;
    uint32_t m = cnt < c32 ? cnt : c32;
    if (m) {
        seeds_decode_kernel<<<cdiv(m, TPB), TPB, 0, ctx->stream>>>(ks.p, m, shape[1], shape[2], 0, coords);
        ISX_CHECK_LAUNCH();
    }
    return ISX_OK;
}

int isx_ws_filter(isx_ctx *ctx, const uint8_t *mask, const int64_t pshape[3], int64_t min_area,
                  int64_t max_area, uint8_t *mask_out, const int64_t *seeds, int64_t n_seeds,
                  int64_t *seeds_out, int64_t *n_out)
{
    ISX_REQUIRE(ctx && mask && pshape && mask_out, "null argument");
    ISX_REQUIRE(n_seeds == 0 || (seeds && seeds_out && n_out), "seed buffers missing");
    ISX_HIP(hipSetDevice(ctx->device));
    ProfScope ps(ctx, ISX_PROF_WS_CCL);
    int64_t N = pshape[0] * pshape[1] * pshape[2];
    Scratch<uint32_t> parent(ctx, N), sizes(ctx, N);
    if (!parent.ok() || !sizes.ok()) return ISX_ERR_NOMEM;
    ISX_TRY(ccl_impl(ctx, mask, pshape, parent.p, sizes.p));
    uint32_t lo = (uint32_t)(min_area < 0 ? 0 : (min_area > 0xFFFFFFFFll ? 0xFFFFFFFFll : min_area));
    uint32_t hi = (uint32_t)(max_area < 0 ? 0 : (max_area > 0xFFFFFFFFll ? 0xFFFFFFFFll : max_area));
    { const int vec = ((uintptr_t)mask_out & 3) == 0;
      ccl_apply_kernel<<<vec ? cdiv((N >> 2) + 1, TPB) : cdiv(N, TPB), TPB, 0, ctx->stream>>>(parent.p, sizes.p, lo, hi, mask_out, N, vec); }
    ISX_CHECK_LAUNCH();
    if (n_out) *n_out = 0;
    if (n_seeds > 0) {
        uint32_t n = (uint32_t)n_seeds;
        Scratch<uint32_t> ridx(ctx, n), flag(ctx, n), pos(ctx, n);
        if (!ridx.ok() || !flag.ok() || !pos.ok()) return ISX_ERR_NOMEM;
        unsigned sg = cdiv(n, TPB);
        seed_ravel_kernel<<<sg, TPB, 0, ctx->stream>>>(seeds, n, pshape[1], pshape[2], ridx.p);
        seed_flag_kernel<<<sg, TPB, 0, ctx->stream>>>(ridx.p, n, mask_out, flag.p);
        ISX_CHECK_LAUNCH();
        ISX_TRY(exclusive_scan_u32(ctx, flag.p, pos.p, n));
        seed_compact_kernel<<<sg, TPB, 0, ctx->stream>>>(seeds, flag.p, pos.p, n, seeds_out, 0);
        ISX_CHECK_LAUNCH();
        uint32_t lp = 0, lf = 0;
        ISX_HIP(hipMemcpyAsync(&lp, pos.p + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
        ISX_HIP(hipMemcpyAsync(&lf, flag.p + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
        ISX_HIP(hipStreamSynchronize(ctx->stream));
        *n_out = (int64_t)lp + lf;
    }
    return ISX_OK;
}

int isx_ws_components(isx_ctx *ctx, const uint8_t *mask, const int64_t pshape[3], uint32_t *roots,
                      uint32_t *sizes)
{
    ISX_REQUIRE(ctx && mask && pshape && roots, "null argument");
    ISX_HIP(hipSetDevice(ctx->device));
    ProfScope ps(ctx, ISX_PROF_WS_CCL);
    return ccl_impl(ctx, mask, pshape, roots, sizes);
}

int isx_affinity_watershed(isx_ctx *ctx, const float *aff, const int64_t *seeds, int64_t n_seeds,
                           const uint8_t *mask, uint32_t *out, const int64_t pshape[3],
                           const float *scale)
{
    ISX_REQUIRE(ctx && aff && mask && out && pshape, "null argument");
    ISX_HIP(hipSetDevice(ctx->device));
    if (n_seeds <= 0 || !seeds) {
        set_error("zero seeds: the reference raises here (np.apply_along_axis, watershed.py:50)");
        return ISX_ERR_NO_SEEDS;
    }
    uint32_t n = (uint32_t)n_seeds;
    int64_t PN = pshape[0] * pshape[1] * pshape[2];
    Scratch<uint32_t> ridx(ctx, n), bad(ctx, 1);
    Scratch<uint8_t> mask2(ctx, PN);
    if (!ridx.ok() || !bad.ok() || !mask2.ok()) return ISX_ERR_NOMEM;
    ISX_HIP(hipMemcpyAsync(mask2.p, mask, PN, hipMemcpyDeviceToDevice, ctx->stream));
    ISX_HIP(hipMemsetAsync(bad.p, 0, 4, ctx->stream));
    seed_check_mark_kernel<<<cdiv(n, TPB), TPB, 0, ctx->stream>>>(seeds, n, pshape[0], pshape[1], pshape[2],
                                                                 mask2.p, bad.p);
    ISX_CHECK_LAUNCH();
    uint32_t hbad = 0;
    ISX_HIP(hipMemcpyAsync(&hbad, bad.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    ISX_REQUIRE(!hbad, "a seed lies on or outside the zero border of the padded volume");
    seed_ravel_kernel<<<cdiv(n, TPB), TPB, 0, ctx->stream>>>(seeds, n, pshape[1], pshape[2], ridx.p);
    ISX_CHECK_LAUNCH();
    return flood_impl(ctx, aff, ridx.p, n, mask2.p, out, pshape, nullptr, nullptr, scale);
}

int isx_segment_output_image(isx_ctx *ctx, const float *fm, const int64_t shape[3], uint32_t *labels,
                             int64_t min_area, int64_t max_area, const float *abs_thresh,
                             int64_t *seeds_out, int64_t seeds_cap, int64_t *n_seeds_out,
                             uint8_t *mask_out, float *thresh_out)
{
    return isx_segment_output_image_scaled(ctx, fm, shape, labels, min_area, max_area, abs_thresh, seeds_out, seeds_cap,
                                           n_seeds_out, mask_out, thresh_out, nullptr);
}

int isx_segment_output_image_scaled(isx_ctx *ctx, const float *fm, const int64_t shape[3], uint32_t *labels,
                                    int64_t min_area, int64_t max_area, const float *abs_thresh,
                                    int64_t *seeds_out, int64_t seeds_cap, int64_t *n_seeds_out,
                                    uint8_t *mask_out, float *thresh_out, const float *scale)
{
    ISX_REQUIRE(ctx && fm && shape && labels, "null argument");
    ISX_HIP(hipSetDevice(ctx->device));
    const int64_t Z = shape[0], Y = shape[1], X = shape[2], n = Z * Y * X;
    const int64_t pshape[3] = {Z + 2, Y + 2, X + 2};
    const int64_t PN = pshape[0] * pshape[1] * pshape[2];
    ISX_REQUIRE(PN < (int64_t)0xFFFFFFF0ll, "volume too large for 32-bit voxel indices");
    if (n_seeds_out) *n_seeds_out = 0;

    // ---- W2 seeds (centre map = channel 4) ------------------------------------------
    uint32_t cap = (uint32_t)(n < 0x7FFFFFF0ll ? n : 0x7FFFFFF0ll);
    // local maxima of a 3x3x3 filter are at most every other voxel per axis in practice;
    // start with n/8 and retry with the true count if it ever overflows
    uint32_t try_cap = (uint32_t)(n / 8 + 1024 < cap ? n / 8 + 1024 : cap);
    uint32_t n_raw = 0;
    std::unique_ptr<Scratch<unsigned long long>> ks;
    {
        ProfScope ps(ctx, ISX_PROF_WS_FILTERS);
        for (;;) {
            ks.reset(new Scratch<unsigned long long>(ctx, try_cap ? try_cap : 1));
            if (!ks->ok()) return ISX_ERR_NOMEM;
            ISX_TRY(seeds_impl(ctx, fm + 4 * n, shape, 1, 0.04f, ks->p, try_cap, &n_raw));
            if (n_raw <= try_cap) break;
            try_cap = n_raw;
        }
    }

    // ---- W1 mask (channel 3), written zero-padded ---------------------------------------
    Scratch<uint8_t> mask_p(ctx, PN);
    Scratch<float> thr(ctx, 1);
    if (!mask_p.ok() || !thr.ok()) return ISX_ERR_NOMEM;
    {
        ProfScope ps(ctx, ISX_PROF_WS_FILTERS);
        if (abs_thresh) {
            ISX_HIP(hipMemcpyAsync(thr.p, abs_thresh, 4, hipMemcpyHostToDevice, ctx->stream));
        } else {
            Scratch<float> sm(ctx, n), tmp(ctx, n);
            if (!sm.ok() || !tmp.ok()) return ISX_ERR_NOMEM;
            ISX_TRY(smooth_otsu(ctx, fm + 3 * n, shape, sm.p, tmp.p, thr.p));
        }
        launch_mask(ctx, fm + 3 * n, thr.p, Z, Y, X, 1, mask_p.p);
        ISX_CHECK_LAUNCH();
        if (thresh_out) ISX_HIP(hipMemcpyAsync(thresh_out, thr.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    }

    // ---- W3 components + size filter; seeds restricted to survivors ----------------------
    Scratch<uint32_t> parent(ctx, PN), sizes(ctx, PN);
    Scratch<uint8_t> mask_f_own(ctx, mask_out ? 1 : PN);
    if (!parent.ok() || !sizes.ok() || !mask_f_own.ok()) return ISX_ERR_NOMEM;
    uint8_t *mask_f = mask_out ? mask_out : mask_f_own.p;
    uint32_t n_keep = 0;
    Scratch<uint32_t> ridx_keep(ctx, n_raw ? n_raw : 1);
    if (!ridx_keep.ok()) return ISX_ERR_NOMEM;
    {
        ProfScope ps(ctx, ISX_PROF_WS_CCL);
        ISX_TRY(ccl_impl(ctx, mask_p.p, pshape, parent.p, sizes.p));
        uint32_t lo = (uint32_t)(min_area < 0 ? 0 : (min_area > 0xFFFFFFFFll ? 0xFFFFFFFFll : min_area));
        uint32_t hi = (uint32_t)(max_area < 0 ? 0 : (max_area > 0xFFFFFFFFll ? 0xFFFFFFFFll : max_area));
        { const int vec = ((uintptr_t)mask_f & 3) == 0;
          ccl_apply_kernel<<<vec ? cdiv((PN >> 2) + 1, TPB) : cdiv(PN, TPB), TPB, 0, ctx->stream>>>(parent.p, sizes.p, lo, hi, mask_f, PN, vec); }
        ISX_CHECK_LAUNCH();
        if (n_raw > 0) {
            Scratch<int64_t> coords(ctx, (size_t)n_raw * 3), coords_keep(ctx, (size_t)n_raw * 3);
            Scratch<uint32_t> ridx(ctx, n_raw), flag(ctx, n_raw), pos(ctx, n_raw);
            if (!coords.ok() || !coords_keep.ok() || !ridx.ok() || !flag.ok() || !pos.ok()) return ISX_ERR_NOMEM;
            unsigned sg = cdiv(n_raw, TPB);
            seeds_decode_kernel<<<sg, TPB, 0, ctx->stream>>>(ks->p, n_raw, Y, X, 1, coords.p);
            seed_ravel_kernel<<<sg, TPB, 0, ctx->stream>>>(coords.p, n_raw, pshape[1], pshape[2], ridx.p);
            seed_flag_kernel<<<sg, TPB, 0, ctx->stream>>>(ridx.p, n_raw, mask_f, flag.p);
            ISX_CHECK_LAUNCH();
            ISX_TRY(exclusive_scan_u32(ctx, flag.p, pos.p, n_raw));
            seed_compact_kernel<<<sg, TPB, 0, ctx->stream>>>(coords.p, flag.p, pos.p, n_raw, coords_keep.p, 0);
            ISX_CHECK_LAUNCH();
            uint32_t lp = 0, lf = 0;
            ISX_HIP(hipMemcpyAsync(&lp, pos.p + (n_raw - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
            ISX_HIP(hipMemcpyAsync(&lf, flag.p + (n_raw - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
            ISX_HIP(hipStreamSynchronize(ctx->stream));
            n_keep = lp + lf;
            if (n_keep) {
                seed_ravel_kernel<<<cdiv(n_keep, TPB), TPB, 0, ctx->stream>>>(coords_keep.p, n_keep, pshape[1],
                                                                            pshape[2], ridx_keep.p);
                ISX_CHECK_LAUNCH();
                if (seeds_out) {
                    uint32_t m = n_keep < (uint32_t)seeds_cap ? n_keep : (uint32_t)seeds_cap;
                    // unpadded coordinates, like the `seeds` the reference returns (watershed.py:222)
                    seed_copy_kernel<<<cdiv(m, TPB), TPB, 0, ctx->stream>>>(coords_keep.p, m, seeds_out, 1);
                    ISX_CHECK_LAUNCH();
                    ISX_HIP(hipStreamSynchronize(ctx->stream));
                }
            }
        }
    }
    if (n_seeds_out) *n_seeds_out = n_keep;
    if (n_keep == 0) {
        ISX_HIP(hipStreamSynchronize(ctx->stream));
        set_error("zero seeds survive: the reference raises here (np.apply_along_axis, watershed.py:50)");
        return ISX_ERR_NO_SEEDS;
    }

    // ---- W0 affinities + W4 flood -----------------------------------------------------
    Scratch<float> aff(ctx, (size_t)3 * PN);
    Scratch<uint32_t> cm(ctx, 8), amm(ctx, 4);
    if (!aff.ok() || !cm.ok() || !amm.ok()) return ISX_ERR_NOMEM;
    {
        ProfScope ps(ctx, ISX_PROF_WS_FILTERS);
        ISX_HIP(hipMemsetAsync(cm.p, 0, 16, ctx->stream));
        ISX_HIP(hipMemsetAsync(cm.p + 4, 0xFF, 16, ctx->stream));
        unsigned g = cdiv(n, (int64_t)TPB * 8);
        if (g > 2048) g = 2048;
        chanmax_kernel<<<dim3(g, 3), TPB, 0, ctx->stream>>>(fm, n, cm.p);
        ISX_CHECK_LAUNCH();
        ISX_REQUIRE(3 * (Z + 2) * (Y + 2) < (int64_t)0x7FFFFFFF, "volume too large for the padding kernel's grid");
        aff_pad_kernel<<<(unsigned)(3 * (Z + 2) * (Y + 2)), 256, 0, ctx->stream>>>(fm, cm.p, Z, Y, X, aff.p);
        ISX_CHECK_LAUNCH();
        aff_range_kernel<<<1, 1, 0, ctx->stream>>>(cm.p, amm.p);
        ISX_CHECK_LAUNCH();
    }
    // the filtered mask's components are a subset of the raw mask's: reuse roots and sizes
    ISX_TRY(flood_impl(ctx, aff.p, ridx_keep.p, n_keep, mask_f, labels, pshape, parent.p, sizes.p, scale, amm.p));
    return ISX_OK;
}

}  // extern "C"
