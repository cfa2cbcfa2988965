// conv55.hip -- the last convolution of the network, 5 -> 5 channels (c8_0.conv1; nn.Conv3d inside ConvModule.forward,
// src/iterseg/unet.py:63-79), on the vector ALU.
//
// A translation unit of its own because it is compiled with -fno-slp-vectorize (iterseg_amd/_build.py): left to itself the compiler
// packs the 675 scalar FMAs of a voxel into v_pk_fma_f32 with op_sel modifiers, some of them on a destination pair that overlaps
// the source whose halves they cross.  A hand-packed version of exactly that shape (round 4: `v_pk_fma_f32 v[10:11], v[6:7],
// s[24:25], v[10:11] op_sel:[0,1,1]`) lost the addend of its low result on whole 16-lane rows when a second process time-sliced
// the GPU, and never in a single process (LOG.md round 4).  The cause could not be established beyond "the
// low half was computed as if op_sel on src2 were 0"; this kernel therefore contains no packed fp32 instruction at all, and
// tests/test_abi.py::test_no_packed_fp32_instruction_crosses_halves_of_its_own_destination scans every shipped kernel for the shape.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "unet_state.h"

namespace isx {

// ------------------------------------------------------------------------------------------
// Kernel: the last convolution of the network, 5 -> 5 channels (c8_0.conv1).  2 x 8 x 16 voxels per block, one per
// thread; the halo tile (4 x 10 x 18 voxels x 5 channels, BN+ReLU of the producer applied) sits in LDS channel-major
// so that the 27 x 5 reads of a wave are conflict-free, and the 675 weights are read with uniform addresses, i.e.
// through the scalar cache into SGPRs (the generic thin kernel re-read them from LDS for every FMA group and
// bounds-checked every tap: 3.8 ms per 64 chunks; this one is bound by its 675 FMAs per voxel).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv55_kernel(const float *__restrict__ src, const float *__restrict__ scale,
                                                     const float *__restrict__ shift, const float *__restrict__ w,
                                                     const float *__restrict__ bias, int D, int H, int W,
                                                     float *__restrict__ out, float *__restrict__ psum,
                                                     float *__restrict__ psq, int tiles_per_chunk, int nty, int ntx)
{
    constexpr int C = 5, PS = 180, RS = 18, S = 720, CP = 724;      // CP: channel plane pitch (floats)
    __shared__ float As[C * CP];
    __shared__ float red[4 * 8 * 2];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / tiles_per_chunk, tile = blockIdx.x % tiles_per_chunk;
    const int nzt = (D + 1) >> 1;
    const int zt = tile % nzt, xt = (tile / nzt) % ntx, yt = tile / (nzt * ntx);
    const int z0 = zt * 2, y0 = yt * 8, x0 = xt * 16;
    const int64_t M = (int64_t)D * H * W;
    float sc[C], sh[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { sc[c] = scale[(int64_t)b * C + c]; sh[c] = shift[(int64_t)b * C + c]; }
    for (int s = tid; s < S; s += 256) {
        const int hz = s / PS, hy = (s / RS) % 10, hx = s % RS;
        const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
        const bool ok = z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W;
        const float *p = src + ((int64_t)b * M + (ok ? ((int64_t)z * H + y) * W + x : 0)) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) As[c * CP + s] = ok ? fmaxf(fmaf(p[c], sc[c], sh[c]), 0.f) : 0.f;
    }
    __syncthreads();
    const int zz = tid >> 7, yy = (tid >> 4) & 7, xx = tid & 15;
    const int z = z0 + zz, y = y0 + yy, x = x0 + xx;
    const bool live = z < D && y < H && x < W;
    const float *ap = As + (zz + 1) * PS + (yy + 1) * RS + (xx + 1);
    float acc[C];
#pragma unroll
    for (int o = 0; o < C; ++o) acc[o] = bias[o];
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int tap = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1);
                const int toff = dz * PS + dy * RS + dx;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float v = ap[c * CP + toff];
                    const float *wp = w + (tap * C + c) * 8;          // uniform address: scalar loads
#pragma unroll
                    for (int o = 0; o < C; ++o) acc[o] = fmaf(v, wp[o], acc[o]);
                }
            }
    if (live) {
        float *q = out + ((int64_t)b * M + ((int64_t)z * H + y) * W + x) * C;
#pragma unroll
        for (int o = 0; o < C; ++o) q[o] = acc[o];
    }
    const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int o = 0; o < C; ++o) {
        float sm = live ? acc[o] : 0.f, q2 = live ? acc[o] * acc[o] : 0.f;
        for (int off = 32; off > 0; off >>= 1) { sm += __shfl_xor(sm, off); q2 += __shfl_xor(q2, off); }
        if (lane == 0) { red[(wave * 8 + o) * 2] = sm; red[(wave * 8 + o) * 2 + 1] = q2; }
    }
    __syncthreads();
    if (tid < C) {
        float sm = 0.f, q2 = 0.f;
        for (int wv = 0; wv < 4; ++wv) { sm += red[(wv * 8 + tid) * 2]; q2 += red[(wv * 8 + tid) * 2 + 1]; }
        const int64_t o = ((int64_t)b * tiles_per_chunk + tile) * C + tid;
        psum[o] = sm;
        psq[o] = q2;
    }
}


int launch_conv55(isx_ctx *ctx, const Level &l, int B, Src sa, const float *w, const float *bias, float *out, float *psum, float *psq,
                  int tiles)
{
    conv55_kernel<<<(unsigned)(B * tiles), 256, 0, ctx->stream>>>(sa.p, sa.scale, sa.shift, w, bias, l.D, l.H, l.W, out, psum, psq, tiles,
                                                                 (l.H + 7) / 8, (l.W + 15) / 16);
    ISX_CHECK_LAUNCH();
    return ISX_OK;
}

}  // namespace isx
