// unet.hip -- the affinity U-Net forward of libitersegx.so on gfx950 (MI355X).
//
// Replaces, for inference, iterseg's UNet / ConvModule (src/iterseg/unet.py:25-106, :126-364),
// load_unet (src/iterseg/predict.py:25-35) and the chunk loop process_chunks /
// predict_chunk_feature_map (predict.py:64-126).
//
// Design (MI355X-first, not a translation of torch.nn):
//   * activations are channels-last [chunk][z][y][x][C] fp32 in HBM; a batch of chunks is
//     pushed through every layer together (train-mode BatchNorm statistics stay per chunk,
//     exactly like the reference's N=1 forward per chunk);
//   * the 3x3x3 convolutions with >= 32 input channels are implicit GEMMs on the fp32 matrix
//     cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate): a work-group owns
//     128 output voxels x 32/64 output channels, stages the input halo tile (once per 4 input
//     channels, reused by all 27 taps) and the 27x4xBN weight slab in LDS, and prefetches the
//     next stage into registers while the matrix cores run;
//   * BatchNorm + ReLU never make a pass of their own: every consumer (next conv's halo staging,
//     max-pool, depthwise transposed conv, final sigmoid) applies the producer's per-chunk
//     scale/shift on load, and every conv epilogue emits deterministic per-tile partial sums
//     for the statistics; the crop+concat of the decoder is a two-source read in the conv;
//   * the thin layers (1->32, 64->5, 5->5) are HBM-bound and run as direct VALU kernels.
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "unet_state.h"
#include "h2_split.h"

namespace isx {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// streaming (non-temporal) 16-byte load.  Tried for the activation tiles of the MFMA conv (r01b): it
// RAISED L2-miss traffic 93 -> 131 GB per launch set and cost 20 % -- halo overlap needs the L2.
__device__ __forceinline__ float4 ld_stream(const float *p)
{
    f32x4 v = __builtin_nontemporal_load((const f32x4 *)p);
    return make_float4(v.x, v.y, v.z, v.w);
}


void unet_destroy(UNetState *u)
{
    if (!u) return;
    if (u->train) train_destroy(u->train);
    for (void *p : u->blocks) (void)hipFree(p);
    for (void *p : u->wblocks) (void)hipFree(p);
    delete u;
}

const char *MODS[9] = {"c0", "c1", "c2", "c3", "c4", "c5_0", "c6_0", "c7_0", "c8_0"};
const int MOD_CIN[9] = {1, 32, 64, 128, 256, 512, 256, 128, 64};
const int MOD_COUT[9] = {32, 64, 128, 256, 256, 128, 64, 32, -1};
const int MOD_LEVEL[9] = {0, 1, 2, 3, 4, 3, 2, 1, 0};
const char *UPS[4] = {"up0", "up1", "up2", "up3"};
const int UP_C[4] = {256, 128, 64, 32};
const int UP_KZ[4] = {2, 1, 1, 1};

// ------------------------------------------------------------------------------------------
// Kernels: chunk gather
// ------------------------------------------------------------------------------------------
__global__ void gather_chunks_kernel(const float *__restrict__ vol, int64_t Y, int64_t X,
                                     const int32_t *__restrict__ table, int cz, int cy, int cx,
                                     float *__restrict__ x0, int64_t total)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)total) return;
    const uint32_t M = (uint32_t)cz * cy * cx;
    const int b = (int)(i / M);
    const uint32_t m = i % M;
    const int x = (int)(m % (uint32_t)cx), y = (int)((m / (uint32_t)cx) % (uint32_t)cy), z = (int)(m / ((uint32_t)cx * cy));
    const int32_t *r = table + 9 * b;
    x0[i] = vol[((int64_t)(r[0] + z) * Y + (r[1] + y)) * X + (r[2] + x)];
}

// ------------------------------------------------------------------------------------------
// Kernel: first convolution, 1 -> 32 channels, as a 32 x 32 x 28 GEMM per wave on the fp32 matrix cores
// (the VALU version spent 6.7 ms per 64 chunks on 864 FMAs per voxel; the layer is bound by its 84 MB of
// output per chunk).  Row = voxel, column = output channel, K = tap (27, padded to 28 with a zero weight):
// MFMA p multiplies taps 2p (lanes 0-31) and 2p+1 (lanes 32-63).  The 14 weight operands live in
// registers for the whole kernel, the 14 voxel operands are plain cached loads of the 1-channel chunk.
// Block = 256 threads = 4 waves x 32 voxels; emits per-block BN partial sums.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_first_kernel(const float *__restrict__ x0, const float *__restrict__ w,
                                                         const float *__restrict__ bias, int D, int H, int W,
                                                         float *__restrict__ out, float *__restrict__ psum,
                                                         float *__restrict__ psq, int blocks_per_chunk, int nt_store, int planar)
{
    __shared__ float red[4][32][2];
    const int b = blockIdx.x / blocks_per_chunk, blk = blockIdx.x % blocks_per_chunk;
    const int64_t M = (int64_t)D * H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hh = lane >> 5, nn = lane & 31;
    const int64_t m0 = (int64_t)blk * 128 + wave * 32;
    const int64_t m = m0 + nn;                              // this lane's voxel (A rows)
    const bool mv = m < M;
    const int64_t mc = mv ? m : 0;
    const uint32_t mc32 = (uint32_t)mc;                     // one chunk has far fewer than 2^32 voxels
    const int x = (int)(mc32 % (uint32_t)W), y = (int)((mc32 / (uint32_t)W) % (uint32_t)H), z = (int)(mc32 / ((uint32_t)W * H));
    const float *src = x0 + (int64_t)b * M + mc;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // The voxel's six "has a neighbour on this side" flags once, and per tap a wave-uniform 32-bit offset: the two taps of an
    // MFMA (lanes 0-31 / 32-63) are compile-time constants selected by the half.  (With the tap index computed per lane the
    // kernel spent 640 vector instructions per wave -- 1.56 ms of its 1.84 ms per launch set -- on 64-bit index arithmetic and
    // 14 x 6 bounds comparisons; SQ_INSTS_VALU of the r04e SQ pass.)
    const bool has[3][3] = {{mv && z > 0, mv, mv && z < D - 1}, {y > 0, true, y < H - 1}, {x > 0, true, x < W - 1}};
    const int HW = H * W;
#pragma unroll
    for (int p = 0; p < 14; ++p) {
        float v = 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int t = 2 * p + half;                     // compile-time: 0..27
            if (t < 27) {
                const int dz = t / 9 - 1, dy = (t / 3) % 3 - 1, dx = t % 3 - 1;
                const bool ok = has[0][dz + 1] && has[1][dy + 1] && has[2][dx + 1];
                const int off = dz * HW + dy * W + dx;      // wave-uniform
                if (hh == half && ok) v = src[off];
            }
        }
        const int t = 2 * p + hh;
        const float wv = t < 27 ? w[t * 32 + nn] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v, wv, acc, 0, 0, 0);
    }
    // C/D layout of 32x32: column = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5).  The tile goes through
    // LDS so that the wave writes it as four fully contiguous 1 KB stores (float4 per lane) instead of sixteen
    // 2 x 128-byte ones: this layer is bound by its 84 MB of output per chunk.
    __shared__ __attribute__((aligned(16))) float T[4][32 * 36];
    const float bsv = bias[nn];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
        const float v = acc[i] + bsv;
        T[wave][row * 36 + nn] = v;
        if (m0 + row < M) { s += v; q += v * v; }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // row-major: store k = rows 8 k .. 8 k + 7, all 32 channels; planar ([chunk][4 planes][voxel][8], UNetState::planar):
        // store k = plane k, all 32 rows -- 1 KB contiguous either way
        const int f = lane + 64 * k;
        const int row = planar ? lane >> 1 : f >> 3, c4 = planar ? 2 * k + (lane & 1) : f & 7;
        if (m0 + row < M) {
            const float4 tv = *(const float4 *)&T[wave][row * 36 + 4 * c4];
            f32x4 tn = {tv.x, tv.y, tv.z, tv.w};
            float *op = planar ? &out[(((int64_t)b * 4 + k) * M + m0 + row) * 8 + 4 * (lane & 1)] : &out[((int64_t)b * M + m0 + row) * 32 + 4 * c4];
            if (nt_store) __builtin_nontemporal_store(tn, (f32x4 *)op);
            else *(float4 *)op = tv;
        }
    }
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    if (hh == 0) { red[wave][nn][0] = s; red[wave][nn][1] = q; }
    __syncthreads();
    if (threadIdx.x < 32) {
        float ss = 0.f, qq = 0.f;
        for (int wv = 0; wv < 4; ++wv) { ss += red[wv][threadIdx.x][0]; qq += red[wv][threadIdx.x][1]; }
        int64_t o = ((int64_t)b * 32 + threadIdx.x) * blocks_per_chunk + blk;          // [chunk][channel][tile]: see bn_finalize_kernel
        psum[o] = ss;
        psq[o] = qq;
    }
}

// ------------------------------------------------------------------------------------------
// Kernel: implicit-GEMM 3x3x3 convolution on the fp32 matrix cores.
// ------------------------------------------------------------------------------------------

static constexpr int A_STRIDE = 6;     // floats per halo slot (4 channels + 2 pad: conflict-free ds_read_b64)

// TX = 3: all 27 taps.  TX = 1: the 9 taps with dx = 0 only (the folded tail layer, where the three
// x-taps live in the output columns instead).
//
// MT = 32-row GEMM tiles per wave: a workgroup covers 128 * MT output voxels.  With MT = 2 one weight
// slab in LDS (the larger half of every stage) feeds twice the voxels, and each B fragment read from
// LDS is used by two MFMAs.
template <int BN, int TX, int MT>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = BN / 32;
    constexpr int NTAPS = 9 * TX;
    constexpr int BFLOATS = NTAPS * 4 * BN;
    constexpr int B4 = BFLOATS / 4;                       // float4 per weight slab
    constexpr int NB = (B4 + 255) / 256;
    float *Bs = smem;
    float *As = smem + BFLOATS;
    constexpr int ROWS = 128 * MT;
    constexpr int TXW = 8 * MT;                          // mode 0 tile: 2 x 8 x TXW voxels
    int *rowinfo = (int *)(As + a.slots_max * A_STRIDE);
    float *red = (float *)(rowinfo + ROWS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, nn = lane & 31;
    // XCD-aware order: consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own
    // L2), so give XCD k the k-th contiguous eighth of the tile list -- the work-groups resident on one XCD
    // are then spatial neighbours and share their halo lines in that XCD's L2.
    int lin = blockIdx.x;
    if (a.xcd_remap) {
        const int per = gridDim.x >> 3, main = per << 3;
        if (lin < main) lin = (lin & 7) * per + (lin >> 3);
    }
    // the N tiles of one voxel tile are adjacent in this order: they read the same halo lines
    const int ntile = lin % a.ntiles;
    lin /= a.ntiles;
    const int b = lin / a.tiles_per_chunk, tile = lin % a.tiles_per_chunk;
    const int D = a.D, H = a.H, W = a.W;
    const int64_t M = (int64_t)D * H * W;

    // ---- tile geometry ---------------------------------------------------------------
    int PS, RS, S;                   // plane stride / row stride in halo slots, slot count
    int z0, y0 = 0, x0 = 0, p0 = 0, qlo = 0, TZ, TP = 0;
    const int WP = W + 2;
    if (a.mode == 0) {
        // z fastest: the tiles in flight on one XCD then cover whole z columns (no z halo re-fetch)
        const int nzt = (D + 1) >> 1;
        int zt = tile % nzt, xt = (tile / nzt) % a.ntx, yt = tile / (nzt * a.ntx);
        z0 = zt * 2; y0 = yt * 8; x0 = xt * TXW; TZ = 2;
        RS = TXW + 2; PS = 10 * RS; S = 4 * PS;
    } else {
        TZ = (a.mode == 1) ? 2 : 1;
        TP = ROWS / TZ;
        const int nzt = (D + TZ - 1) / TZ;
        int zt = tile % nzt, pt = tile / nzt;
        z0 = zt * TZ; p0 = pt * TP;
        int plast = p0 + TP - 1;
        if (plast > H * W - 1) plast = H * W - 1;
        int qfirst = (p0 / W) * WP + (p0 % W) + 1;
        int qlast = (plast / W) * WP + (plast % W) + 1;
        qlo = qfirst - WP - 1;
        RS = WP;
        PS = qlast + WP + 1 - qlo + 1;
        S = (TZ + 2) * PS;
    }

    // ---- per-row table (output voxel of each GEMM row) and this lane's A rows ---------------
    if (tid < ROWS) {
        int r = tid, g = -1;
        if (a.mode == 0) {
            int zz = r / (8 * TXW), yy = (r / TXW) & 7, xx = r % TXW;
            int z = z0 + zz, y = y0 + yy, x = x0 + xx;
            if (z < D && y < H && x < W) g = (z * H + y) * W + x;
        } else {
            int zz = r / TP, p = p0 + r % TP;
            int z = z0 + zz;
            if (z < D && p < H * W) g = z * H * W + p;
        }
        rowinfo[r] = g;
    }
    auto row_slot = [&](int r) __attribute__((always_inline)) -> int {
        if (a.mode == 0) {
            int zz = r / (8 * TXW), yy = (r / TXW) & 7, xx = r % TXW;
            return (zz + 1) * PS + (yy + 1) * RS + (xx + 1);
        }
        int zz = r / TP, p = p0 + r % TP;
        if (p > H * W - 1) p = H * W - 1;                // invalid rows read some in-tile slot
        int q = (p / W) * WP + (p % W) + 1;
        return (zz + 1) * PS + (q - qlo);
    };
    const int aslot0 = row_slot((wave * MT) * 32 + nn);
    const int aslot1 = (MT > 1) ? row_slot((wave * MT + 1) * 32 + nn) : 0;

    // ---- halo slots staged by this thread: global voxel index, -1 (zero padding) or -2 (no slot).
    // Named scalars, not arrays: hipcc leaves array forms of these and of the prefetch registers in
    // scratch memory when they live across the K loop (CDNA guide rule 20).
    auto slot_voxel = [&](int s) __attribute__((always_inline)) -> int {
        if (s >= S) return -2;
        int z, y, x;
        if (a.mode == 0) {
            int hz = s / PS, hy = (s / RS) % 10, hx = s % RS;
            z = z0 + hz - 1; y = y0 + hy - 1; x = x0 + hx - 1;
        } else {
            int hz = s / PS, qq = qlo + s % PS + WP;      // + WP keeps the division non-negative
            z = z0 + hz - 1;
            y = qq / WP - 1;
            x = qq % WP - 1;
        }
        return (z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W) ? (z * H + y) * W + x : -1;
    };
    const int gv0 = slot_voxel(tid), gv1 = slot_voxel(tid + 256), gv2 = slot_voxel(tid + 512),
              gv3 = slot_voxel(tid + 768), gv4 = (MT > 1) ? slot_voxel(tid + 1024) : -2;

    const int nsteps = (a.CA + a.CB) >> 2;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][j][i] = 0.f;

    // Register prefetch of the next stage.  Loads are unconditional (out-of-tile slots read a
    // clamped address) and stay untouched in registers while the matrix cores run; the producer's
    // BN scale/shift + ReLU and the zero padding are applied when the stage is written to LDS, so
    // nothing between the loads and the MFMA phase waits on vmcnt.
    static_assert(NB <= 7, "prefetch register plan covers at most 7 float4 of weights per thread");
    float4 ra0, ra1, ra2, ra3, ra4, rb0, rb1, rb2, rb3, rb4, rb5, rb6, vs, vh;
    bool has_affine = false;
#define ISX_LOAD_A(k) if constexpr (k < 4 || MT > 1) ra##k = *(const float4 *)&src[((int64_t)b * M + (gv##k > 0 ? gv##k : 0)) * C + cc];
// (guarded: the slab is not a multiple of 256 float4 -- 864 / 1728 -- and the last stage of the last tile ends the
// allocation; the unguarded tail read 2.5 KB past it, harmless only while the next page happens to be mapped)
#define ISX_LOAD_B(k) if constexpr (NB > k) { if (B4 % 256 == 0 || tid + 256 * k < B4) rb##k = wsrc[tid + 256 * k]; }
#define ISX_LOAD_STAGE(step_)                                                                     \
    {                                                                                             \
        const int c0 = (step_) << 2;                                                              \
        const float *src, *sc, *sh;                                                               \
        int C, cc;                                                                                \
        if (c0 < a.CA) { src = a.srcA; sc = a.scaleA; sh = a.shiftA; C = a.CA; cc = c0; }         \
        else { src = a.srcB; sc = a.scaleB; sh = a.shiftB; C = a.CB; cc = c0 - a.CA; }            \
        has_affine = sc != nullptr;                                                               \
        const float *scp = has_affine ? sc : a.bias, *shp = has_affine ? sh : a.bias;             \
        const int64_t so = has_affine ? (int64_t)b * C + cc : 0;                                  \
        vs = *(const float4 *)&scp[so];                                                           \
        vh = *(const float4 *)&shp[so];                                                           \
        ISX_LOAD_A(0) ISX_LOAD_A(1) ISX_LOAD_A(2) ISX_LOAD_A(3) ISX_LOAD_A(4)                     \
        const float4 *wsrc = (const float4 *)(a.w + ((int64_t)ntile * nsteps + (step_)) * BFLOATS); \
        ISX_LOAD_B(0) ISX_LOAD_B(1) ISX_LOAD_B(2) ISX_LOAD_B(3) ISX_LOAD_B(4) ISX_LOAD_B(5) ISX_LOAD_B(6) \
    }
#define ISX_STORE_A(k)                                                                            \
    if ((k < 4 || MT > 1) && gv##k != -2) {                                                       \
        float4 v = ra##k;                                                                         \
        if (has_affine) {                                                                         \
            v.x = fmaxf(fmaf(v.x, vs.x, vh.x), 0.f);                                              \
            v.y = fmaxf(fmaf(v.y, vs.y, vh.y), 0.f);                                              \
            v.z = fmaxf(fmaf(v.z, vs.z, vh.z), 0.f);                                              \
            v.w = fmaxf(fmaf(v.w, vs.w, vh.w), 0.f);                                              \
        }                                                                                         \
        if (gv##k < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);                                       \
        float *dst = As + (tid + 256 * k) * A_STRIDE;                                             \
        *(float2 *)dst = make_float2(v.x, v.y);                                                   \
        *(float2 *)(dst + 2) = make_float2(v.z, v.w);                                             \
    }
#define ISX_STORE_B(k) if constexpr (NB > k) { if (tid + 256 * k < B4) ((float4 *)Bs)[tid + 256 * k] = rb##k; }
#define ISX_STORE_STAGE()                                                                         \
    {                                                                                             \
        ISX_STORE_A(0) ISX_STORE_A(1) ISX_STORE_A(2) ISX_STORE_A(3) ISX_STORE_A(4)                \
        ISX_STORE_B(0) ISX_STORE_B(1) ISX_STORE_B(2) ISX_STORE_B(3) ISX_STORE_B(4) ISX_STORE_B(5) ISX_STORE_B(6) \
    }

    ISX_LOAD_STAGE(0)
    for (int step = 0; step < nsteps; ++step) {
        __syncthreads();                 // everyone is done reading the previous stage
        ISX_STORE_STAGE()
        __syncthreads();
        if (step + 1 < nsteps) ISX_LOAD_STAGE(step + 1)   // in flight while the matrix cores run
        __builtin_amdgcn_sched_barrier(0);
        const float *ap0 = As + aslot0 * A_STRIDE + 2 * hh;
        const float *ap1 = As + aslot1 * A_STRIDE + 2 * hh;
        const float *bp = Bs + (2 * hh) * BN + nn;
#pragma unroll
        for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dxi = 0; dxi < TX; ++dxi) {
                    const int dx = (TX == 3) ? dxi - 1 : 0;
                    const int tap = ((dz + 1) * 3 + (dy + 1)) * TX + dxi;
                    const int toff = dz * PS + dy * RS + dx;
                    const float2 av0 = *(const float2 *)(ap0 + toff * A_STRIDE);
                    float2 av1 = av0;
                    if constexpr (MT > 1) av1 = *(const float2 *)(ap1 + toff * A_STRIDE);
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const float b0 = bp[(tap * 4) * BN + 32 * j];
                        const float b1 = bp[(tap * 4 + 1) * BN + 32 * j];
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.x, b0, acc[0][j], 0, 0, 0);
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0.y, b1, acc[0][j], 0, 0, 0);
                        if constexpr (MT > 1) {
                            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.x, b0, acc[1][j], 0, 0, 0);
                            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1.y, b1, acc[1][j], 0, 0, 0);
                        }
                    }
                }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef ISX_LOAD_A
#undef ISX_LOAD_B
#undef ISX_LOAD_STAGE
#undef ISX_STORE_A
#undef ISX_STORE_B
#undef ISX_STORE_STAGE

    // ---- epilogue: + bias, store, per-tile BN partial sums (valid rows only) ----------------
    // C/D layout of 32x32: column = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int ch = ntile * BN + 32 * j + nn;
        const float bsv = a.bias[ch];
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (wave * MT + m) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                const int g = rowinfo[row];
                if (g >= 0) {
                    const float v = acc[m][j][i] + bsv;
                    if (ch < a.store_cols) a.out[((int64_t)b * M + g) * a.store_cols + ch] = v;
                    s += v;
                    q += v * v;
                }
            }
        s += __shfl_xor(s, 32);
        q += __shfl_xor(q, 32);
        if (hh == 0) { red[(wave * BN + 32 * j + nn) * 2] = s; red[(wave * BN + 32 * j + nn) * 2 + 1] = q; }
    }
    __syncthreads();
    if (tid < BN && !a.no_stats) {
        float s = 0.f, q = 0.f;
        for (int wv = 0; wv < 4; ++wv) { s += red[(wv * BN + tid) * 2]; q += red[(wv * BN + tid) * 2 + 1]; }
        int64_t o = ((int64_t)b * a.tiles_per_chunk + tile) * a.Cout + ntile * BN + tid;
        a.psum[o] = s;
        a.psq[o] = q;
    }
}

// ------------------------------------------------------------------------------------------
// Kernels: BatchNorm statistics (train mode, per chunk).  Deterministic: fixed-order double sums.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float *__restrict__ psum, const float *__restrict__ psq,
                                                          int tiles, int C, double count, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ scale,
                                                          float *__restrict__ shift, float *__restrict__ bmean,
                                                          float *__restrict__ bvar, float *__restrict__ invstd_out,
                                                          float *__restrict__ rmean, float *__restrict__ rvar, int tile_fastest)
{
    // tile_fastest: the partial sums are [chunk][channel][tile] (conv_first_kernel, the conv_h2 kernels: consecutive lanes read
    // consecutive floats) instead of [chunk][tile][channel]; same lanes, same order, same sums
    __shared__ double rs[256], rq[256];
    const int c = blockIdx.x, b = blockIdx.y;
    double s = 0.0, q = 0.0;
    for (int t = threadIdx.x; t < tiles; t += 256) {
        int64_t o = tile_fastest ? ((int64_t)b * C + c) * tiles + t : ((int64_t)b * tiles + t) * C + c;
        s += (double)psum[o];
        q += (double)psq[o];
    }
    rs[threadIdx.x] = s;
    rq[threadIdx.x] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rq[threadIdx.x] += rq[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double mean = rs[0] / count;
        double var = rq[0] / count - mean * mean;           // biased, as BatchNorm normalises with
        if (var < 0.0) var = 0.0;
        float invstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
        float sc = gamma[c] * invstd;
        scale[b * C + c] = sc;
        shift[b * C + c] = beta[c] - (float)mean * sc;
        bmean[b * C + c] = (float)mean;
        invstd_out[b * C + c] = invstd;
        const float ub = (float)(count > 1.0 ? var * count / (count - 1.0) : var);
        bvar[b * C + c] = ub;
        if (rmean) {          // a single chunk (gridDim.y == 1, the training step): bn_running_kernel's update, done here
            rmean[c] = (1.f - BN_MOMENTUM) * rmean[c] + BN_MOMENTUM * (float)mean;
            rvar[c] = (1.f - BN_MOMENTUM) * rvar[c] + BN_MOMENTUM * ub;
        }
    }
}

// running statistics: chunks update them one after the other, like the reference's chunk loop
__global__ void bn_running_kernel(const float *bmean, const float *bvar, int B, int C, float *rmean, float *rvar)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float m = rmean[c], v = rvar[c];
    for (int b = 0; b < B; ++b) {
        m = (1.f - BN_MOMENTUM) * m + BN_MOMENTUM * bmean[b * C + c];
        v = (1.f - BN_MOMENTUM) * v + BN_MOMENTUM * bvar[b * C + c];
    }
    rmean[c] = m;
    rvar[c] = v;
}

// ... of every layer of a forward pass in ONE launch at its end (grid.y = layer): nothing in the pass reads the running statistics,
// and a launch per layer put 17 us of a sequential 64-step chain between each layer's statistics and the next convolution
// (324 launches, 2.8 ms per 512^3 frame).  Same chain per channel: bit-identical.
struct BnRunAll {
    const float *bmean[ISX_MAX_LAYERS], *bvar[ISX_MAX_LAYERS];
    float *rmean[ISX_MAX_LAYERS], *rvar[ISX_MAX_LAYERS];
    int C[ISX_MAX_LAYERS];
};
__global__ void bn_running_all_kernel(const BnRunAll t, int B)
{
    const int k = blockIdx.y, C = t.C[k];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float *bmean = t.bmean[k], *bvar = t.bvar[k];
    float m = t.rmean[k][c], v = t.rvar[k][c];
    for (int b = 0; b < B; ++b) {
        m = (1.f - BN_MOMENTUM) * m + BN_MOMENTUM * bmean[b * C + c];
        v = (1.f - BN_MOMENTUM) * v + BN_MOMENTUM * bvar[b * C + c];
    }
    t.rmean[k][c] = m;
    t.rvar[k][c] = v;
}

// ------------------------------------------------------------------------------------------
// Kernels: max-pool (stride = kernel, padding (0,1,1)) and depthwise transposed convolution
// (stride = kernel) + crop, both with the producer's BN+ReLU applied on load.  4 channels/thread.
// ------------------------------------------------------------------------------------------
constexpr int G = 1;        // float4 groups per thread
__device__ __forceinline__ void put_channels(float *dst, const float4 (&v)[G], bool nt)
{
    if (nt) __builtin_nontemporal_store(f32x4{v[0].x, v[0].y, v[0].z, v[0].w}, (f32x4 *)dst);
    else *(float4 *)dst = v[0];
}

__global__ void pool_kernel(const float *__restrict__ src, const float *__restrict__ scale,
                            const float *__restrict__ shift, int C, int D, int H, int W, int kz, int Do, int Ho,
                            int Wo, float *__restrict__ dst, int64_t total)
{
    // 32-bit index arithmetic throughout (total < 2^32 is checked by the launcher): the 64-bit divisions this kernel
    // started with cost more than its memory traffic
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)total) return;
    const uint32_t CG = (uint32_t)C / (4 * G);
    const int c0 = (int)(i % CG) * 4 * G;
    const uint32_t v = i / CG;
    const uint32_t Mo32 = (uint32_t)Do * Ho * Wo;
    const int64_t Mo = Mo32, M = (int64_t)D * H * W;
    const int b = (int)(v / Mo32);
    const uint32_t m = v % Mo32;
    const int x = (int)(m % (uint32_t)Wo), y = (int)((m / (uint32_t)Wo) % (uint32_t)Ho), z = (int)(m / ((uint32_t)Wo * Ho));
    float4 vs[G], vh[G], best[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        vs[g] = *(const float4 *)&scale[(int64_t)b * C + c0 + 4 * g];
        vh[g] = *(const float4 *)&shift[(int64_t)b * C + c0 + 4 * g];
        best[g] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    }
    for (int i0 = 0; i0 < kz; ++i0)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                int zz = z * kz + i0, yy = 2 * y - 1 + j, xx = 2 * x - 1 + k;
                if (zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    float4 t = *(const float4 *)&src[((int64_t)b * M + ((int64_t)zz * H + yy) * W + xx) * C + c0 + 4 * g];
                    t.x = fmaxf(fmaf(t.x, vs[g].x, vh[g].x), 0.f);
                    t.y = fmaxf(fmaf(t.y, vs[g].y, vh[g].y), 0.f);
                    t.z = fmaxf(fmaf(t.z, vs[g].z, vh[g].z), 0.f);
                    t.w = fmaxf(fmaf(t.w, vs[g].w, vh[g].w), 0.f);
                    best[g].x = fmaxf(best[g].x, t.x);
                    best[g].y = fmaxf(best[g].y, t.y);
                    best[g].z = fmaxf(best[g].z, t.z);
                    best[g].w = fmaxf(best[g].w, t.w);
                }
            }
    put_channels(&dst[((int64_t)b * Mo + m) * C + c0], best, false);
}

// The same pooling over PLANAR tensors, [chunk][C/8][voxel][8] in and out (UNetState::planar): thread = (chunk, 8-channel plane,
// output voxel) reads the 2 x 32 contiguous bytes of its two source columns per (z, y) row and writes 32 -- consecutive lanes
// read and write consecutive memory.
__device__ __forceinline__ void bn_relu4(float4 &t, const float4 &s, const float4 &h)
{
    t.x = fmaxf(fmaf(t.x, s.x, h.x), 0.f); t.y = fmaxf(fmaf(t.y, s.y, h.y), 0.f);
    t.z = fmaxf(fmaf(t.z, s.z, h.z), 0.f); t.w = fmaxf(fmaf(t.w, s.w, h.w), 0.f);
}
// presplit: the pooled values leave already split for the split-f16 convolution that reads them -- [8 hi halves][8 residual halves]
// of value * H2_SA in the 32 bytes of a voxel's plane, exactly what that kernel's staging computes from the float values
// (split8r: the 512-thread kernels' leaner form is the same function of its input) -- so that its staging of this source is a copy.
__global__ void pool_planar_kernel(const float *__restrict__ src, const float *__restrict__ scale, const float *__restrict__ shift,
                                   int C, int D, int H, int W, int kz, int Do, int Ho, int Wo, float *__restrict__ dst, int64_t total,
                                   int presplit)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)total) return;
    const uint32_t Mo32 = (uint32_t)Do * Ho * Wo, NP = (uint32_t)C >> 3;
    const int64_t M = (int64_t)D * H * W;
    const uint32_t m = i % Mo32, pl = (i / Mo32) % NP;
    const int b = (int)(i / (Mo32 * NP));
    const int x = (int)(m % (uint32_t)Wo), y = (int)((m / (uint32_t)Wo) % (uint32_t)Ho), z = (int)(m / ((uint32_t)Wo * Ho));
    const float4 s0 = *(const float4 *)&scale[(int64_t)b * C + 8 * pl], s1 = *(const float4 *)&scale[(int64_t)b * C + 8 * pl + 4];
    const float4 h0 = *(const float4 *)&shift[(int64_t)b * C + 8 * pl], h1 = *(const float4 *)&shift[(int64_t)b * C + 8 * pl + 4];
    float4 b0 = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), b1 = b0;
    const float *sp = src + ((int64_t)b * NP + pl) * M * 8;
    for (int i0 = 0; i0 < kz; ++i0)
        for (int j = 0; j < 2; ++j) {
            const int zz = z * kz + i0, yy = 2 * y - 1 + j;
            if (zz >= D || yy < 0 || yy >= H) continue;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int xx = 2 * x - 1 + k;
                if (xx < 0 || xx >= W) continue;
                const float *pp = &sp[(((int64_t)zz * H + yy) * W + xx) * 8];
                float4 t0 = *(const float4 *)pp, t1 = *(const float4 *)(pp + 4);
                bn_relu4(t0, s0, h0);
                bn_relu4(t1, s1, h1);
                b0.x = fmaxf(b0.x, t0.x); b0.y = fmaxf(b0.y, t0.y); b0.z = fmaxf(b0.z, t0.z); b0.w = fmaxf(b0.w, t0.w);
                b1.x = fmaxf(b1.x, t1.x); b1.y = fmaxf(b1.y, t1.y); b1.z = fmaxf(b1.z, t1.z); b1.w = fmaxf(b1.w, t1.w);
            }
        }
    float *dp = &dst[(((int64_t)b * NP + pl) * Mo32 + m) * 8];
    if (presplit) {
        uint4 hi, lo;
        split8r(make_float4(b0.x * H2_SA, b0.y * H2_SA, b0.z * H2_SA, b0.w * H2_SA),
                make_float4(b1.x * H2_SA, b1.y * H2_SA, b1.z * H2_SA, b1.w * H2_SA), hi, lo);
        *(uint4 *)dp = hi;
        *(uint4 *)(dp + 4) = lo;
        return;
    }
    *(float4 *)dp = b0;
    *(float4 *)(dp + 4) = b1;
}

// out[z, y, x, c] = act(in[z/kz, (y+off)/2, (x+off)/2, c]) * w[(z%kz, (y+off)%2, (x+off)%2)][c] + bias[c]
// for the cropped output box (Do, Ho, Wo); off = 0 for the [:-1] crops, 1 for the [1:-1] crop.
__global__ void upsample_kernel(const float *__restrict__ src, const float *__restrict__ scale,
                                const float *__restrict__ shift, int C, int D, int H, int W, int kz, int off,
                                const float *__restrict__ w, const float *__restrict__ bias, int Do, int Ho, int Wo,
                                float *__restrict__ dst, int64_t total, int nt_store, int src_planar, int dst_planar)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;       // 32-bit index arithmetic (see pool_kernel)
    if (i >= (uint32_t)total) return;
    const uint32_t CG = (uint32_t)C / (4 * G);
    const uint32_t Mo32 = (uint32_t)Do * Ho * Wo;
    const int64_t Mo = Mo32, M = (int64_t)D * H * W;
    int c0, b;
    uint32_t m;
    if (dst_planar) {                  // thread = (chunk, 8-channel plane, output voxel, half plane): consecutive lanes, consecutive memory
        const uint32_t v = i >> 1, pl = (v / Mo32) % (CG >> 1);
        c0 = (int)(pl * 8 + (i & 1) * 4);
        m = v % Mo32;
        b = (int)(v / (Mo32 * (CG >> 1)));
    } else {
        c0 = (int)(i % CG) * 4 * G;
        const uint32_t v = i / CG;
        b = (int)(v / Mo32);
        m = v % Mo32;
    }
    const int x = (int)(m % (uint32_t)Wo), y = (int)((m / (uint32_t)Wo) % (uint32_t)Ho), z = (int)(m / ((uint32_t)Wo * Ho));
    int zs = z / kz, ys = (y + off) >> 1, xs = (x + off) >> 1;
    int kidx = ((z % kz) * 2 + ((y + off) & 1)) * 2 + ((x + off) & 1);
    float4 o[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float4 vs = *(const float4 *)&scale[(int64_t)b * C + c0 + 4 * g];
        const float4 vh = *(const float4 *)&shift[(int64_t)b * C + c0 + 4 * g];
        const int64_t ms = ((int64_t)zs * H + ys) * W + xs;
        float4 t = src_planar ? *(const float4 *)&src[(((int64_t)b * (C >> 3) + (c0 >> 3)) * M + ms) * 8 + (c0 & 7) + 4 * g]
                              : *(const float4 *)&src[((int64_t)b * M + ms) * C + c0 + 4 * g];
        const float4 wv = *(const float4 *)&w[(int64_t)kidx * C + c0 + 4 * g];
        const float4 bv = *(const float4 *)&bias[c0 + 4 * g];
        t.x = fmaf(fmaxf(fmaf(t.x, vs.x, vh.x), 0.f), wv.x, bv.x);
        t.y = fmaf(fmaxf(fmaf(t.y, vs.y, vh.y), 0.f), wv.y, bv.y);
        t.z = fmaf(fmaxf(fmaf(t.z, vs.z, vh.z), 0.f), wv.z, bv.z);
        t.w = fmaf(fmaxf(fmaf(t.w, vs.w, vh.w), 0.f), wv.w, bv.w);
        o[g] = t;
    }
    float *dp = dst_planar ? &dst[(((int64_t)b * (C >> 3) + (c0 >> 3)) * Mo + m) * 8 + (c0 & 7)] : &dst[((int64_t)b * Mo + m) * C + c0];
    put_channels(dp, o, nt_store != 0);
}

// The same operator per SOURCE voxel (kz = 1): one load and one BatchNorm + ReLU feed the four outputs a coarse voxel
// up-samples to -- a quarter of the loads, address arithmetic and activation math of the per-output kernel above (which ran at
// 3.7 TB/s against 5.6 TB/s for pool_kernel).  thread = (coarse voxel, 4 G channels).
__global__ void upsample_src_kernel(const float *__restrict__ src, const float *__restrict__ scale, const float *__restrict__ shift,
                                    int C, int D, int H, int W, int off, const float *__restrict__ w, const float *__restrict__ bias,
                                    int Ho, int Wo, float *__restrict__ dst, int64_t total)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)total) return;
    const uint32_t CG = (uint32_t)C / (4 * G);
    const int c0 = (int)(i % CG) * 4 * G;
    const uint32_t v = i / CG;
    const uint32_t M32 = (uint32_t)D * H * W;
    const int64_t M = M32, Mo = (int64_t)D * Ho * Wo;
    const int b = (int)(v / M32);
    const uint32_t m = v % M32;
    const int xs = (int)(m % (uint32_t)W), ys = (int)((m / (uint32_t)W) % (uint32_t)H), z = (int)(m / ((uint32_t)W * H));
    float4 t[G], bv[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const float4 vs = *(const float4 *)&scale[(int64_t)b * C + c0 + 4 * g];
        const float4 vh = *(const float4 *)&shift[(int64_t)b * C + c0 + 4 * g];
        t[g] = *(const float4 *)&src[((int64_t)b * M + m) * C + c0 + 4 * g];
        t[g].x = fmaxf(fmaf(t[g].x, vs.x, vh.x), 0.f); t[g].y = fmaxf(fmaf(t[g].y, vs.y, vh.y), 0.f);
        t[g].z = fmaxf(fmaf(t[g].z, vs.z, vh.z), 0.f); t[g].w = fmaxf(fmaf(t[g].w, vs.w, vh.w), 0.f);
        bv[g] = *(const float4 *)&bias[c0 + 4 * g];
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int y = 2 * ys + j - off, x = 2 * xs + k - off;             // (y + off) >> 1 == ys, parity j
            if (y < 0 || y >= Ho || x < 0 || x >= Wo) continue;
            float4 o[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float4 wv = *(const float4 *)&w[(int64_t)(j * 2 + k) * C + c0 + 4 * g];
                o[g] = make_float4(fmaf(t[g].x, wv.x, bv[g].x), fmaf(t[g].y, wv.y, bv[g].y), fmaf(t[g].z, wv.z, bv[g].z),
                                   fmaf(t[g].w, wv.w, bv[g].w));
            }
            put_channels(&dst[((int64_t)b * Mo + ((int64_t)z * Ho + y) * Wo + x) * C + c0], o, true);
        }
}

// ... and over PLANAR tensors in and out: thread = (chunk, 8-channel plane, z, coarse row, FINE column, half plane) loads 16
// bytes (the two fine columns of a coarse voxel read the same bytes: L1), applies BatchNorm + ReLU once and writes its two
// output rows -- consecutive lanes store consecutive 16-byte pieces, so every store instruction writes whole lines and can
// stream.  (One thread per coarse voxel writing 32-byte pieces with 32-byte gaps took 1.8 x the row-major kernel's time, and
// 3.4 x with streaming stores.)
__global__ void upsample_src_planar_kernel(const float *__restrict__ src, const float *__restrict__ scale, const float *__restrict__ shift,
                                           int C, int D, int H, int W, int off, const float *__restrict__ w, const float *__restrict__ bias,
                                           int Ho, int Wo, float *__restrict__ dst, int64_t total)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)total) return;
    const uint32_t NP = (uint32_t)C >> 3, R = (uint32_t)D * H;      // R: (z, coarse row) pairs
    const int64_t M = (int64_t)D * H * W, Mo = (int64_t)D * Ho * Wo;
    const int half = (int)(i & 1u);
    const uint32_t v = i >> 1;
    const int xf = (int)(v % (uint32_t)Wo);
    const uint32_t r = (v / (uint32_t)Wo) % R, pl = (v / ((uint32_t)Wo * R)) % NP;
    const int b = (int)(v / ((uint32_t)Wo * R * NP));
    const int ys = (int)(r % (uint32_t)H), z = (int)(r / (uint32_t)H);
    const int xs = (xf + off) >> 1, k = (xf + off) & 1;
    const int c0 = (int)pl * 8 + 4 * half;
    float4 t = *(const float4 *)&src[(((int64_t)b * NP + pl) * M + ((int64_t)z * H + ys) * W + xs) * 8 + 4 * half];
    bn_relu4(t, *(const float4 *)&scale[(int64_t)b * C + c0], *(const float4 *)&shift[(int64_t)b * C + c0]);
    const float4 bv = *(const float4 *)&bias[c0];
    float *dpl = dst + ((int64_t)b * NP + pl) * Mo * 8 + 4 * half;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int y = 2 * ys + j - off;
        if (y < 0 || y >= Ho) continue;
        const float4 wv = *(const float4 *)&w[(int64_t)(j * 2 + k) * C + c0];
        __builtin_nontemporal_store(f32x4{fmaf(t.x, wv.x, bv.x), fmaf(t.y, wv.y, bv.y), fmaf(t.z, wv.z, bv.z), fmaf(t.w, wv.w, bv.w)},
                                    (f32x4 *)&dpl[(((int64_t)z * Ho + y) * Wo + xf) * 8]);
    }
}

// ------------------------------------------------------------------------------------------
// Kernels: the thin tail, Cin -> NOUT (<= 8) channels, direct.  One voxel per thread; weights
// [27][Cin][8] broadcast from LDS.  Two sources like the MFMA kernel (concat of the decoder).
// ------------------------------------------------------------------------------------------
template <int NOUT>
__global__ __launch_bounds__(256) void conv_thin_kernel(const float *__restrict__ srcA, const float *__restrict__ scaleA,
                                                        const float *__restrict__ shiftA, int CA,
                                                        const float *__restrict__ srcB, const float *__restrict__ scaleB,
                                                        const float *__restrict__ shiftB, int CB,
                                                        const float *__restrict__ w, const float *__restrict__ bias,
                                                        int D, int H, int W, int nout, float *__restrict__ out,
                                                        float *__restrict__ psum, float *__restrict__ psq,
                                                        int blocks_per_chunk)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Cin = CA + CB;
    float *ws = smem;                                   // [27][Cin][8]
    float *red = smem + 27 * Cin * 8;                   // [4][8][2]
    for (int i = threadIdx.x; i < 27 * Cin * 8; i += 256) ws[i] = w[i];
    __syncthreads();
    const int b = blockIdx.x / blocks_per_chunk, blk = blockIdx.x % blocks_per_chunk;
    const int64_t M = (int64_t)D * H * W;
    const int64_t m = (int64_t)blk * 256 + threadIdx.x;
    float acc[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) acc[o] = bias[o];
    const bool live = m < M;
    if (live) {
        int x = (int)(m % W), y = (int)((m / W) % H), z = (int)(m / ((int64_t)W * H));
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int zz = z + dz, yy = y + dy, xx = x + dx;
                    if (zz < 0 || zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    const int tap = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1);
                    const int64_t g = (int64_t)b * M + ((int64_t)zz * H + yy) * W + xx;
                    for (int part = 0; part < 2; ++part) {
                        const float *src = part ? srcB : srcA;
                        const float *sc = part ? scaleB : scaleA, *sh = part ? shiftB : shiftA;
                        const int C = part ? CB : CA, cbase = part ? CA : 0;
                        if (C == 0) continue;
                        if ((C & 3) == 0) {
                            for (int c = 0; c < C; c += 4) {
                                float4 v = *(const float4 *)&src[g * C + c];
                                if (sc) {
                                    const float4 vs = *(const float4 *)&sc[(int64_t)b * C + c];
                                    const float4 vh = *(const float4 *)&sh[(int64_t)b * C + c];
                                    v.x = fmaxf(fmaf(v.x, vs.x, vh.x), 0.f);
                                    v.y = fmaxf(fmaf(v.y, vs.y, vh.y), 0.f);
                                    v.z = fmaxf(fmaf(v.z, vs.z, vh.z), 0.f);
                                    v.w = fmaxf(fmaf(v.w, vs.w, vh.w), 0.f);
                                }
                                const float *wp = ws + ((int64_t)tap * Cin + cbase + c) * 8;
#pragma unroll
                                for (int o = 0; o < NOUT; ++o) {
                                    acc[o] = fmaf(v.x, wp[o], acc[o]);
                                    acc[o] = fmaf(v.y, wp[8 + o], acc[o]);
                                    acc[o] = fmaf(v.z, wp[16 + o], acc[o]);
                                    acc[o] = fmaf(v.w, wp[24 + o], acc[o]);
                                }
                            }
                        } else {
                            for (int c = 0; c < C; ++c) {
                                float v = src[g * C + c];
                                if (sc) v = fmaxf(fmaf(v, sc[(int64_t)b * C + c], sh[(int64_t)b * C + c]), 0.f);
                                const float *wp = ws + ((int64_t)tap * Cin + cbase + c) * 8;
#pragma unroll
                                for (int o = 0; o < NOUT; ++o) acc[o] = fmaf(v, wp[o], acc[o]);
                            }
                        }
                    }
                }
        for (int o = 0; o < nout; ++o) out[((int64_t)b * M + m) * nout + o] = acc[o];
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        float s = live ? acc[o] : 0.f, q = live ? acc[o] * acc[o] : 0.f;
        for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
        if (lane == 0) { red[(wave * 8 + o) * 2] = s; red[(wave * 8 + o) * 2 + 1] = q; }
    }
    __syncthreads();
    if ((int)threadIdx.x < nout) {
        float s = 0.f, q = 0.f;
        for (int wv = 0; wv < 4; ++wv) { s += red[(wv * 8 + threadIdx.x) * 2]; q += red[(wv * 8 + threadIdx.x) * 2 + 1]; }
        int64_t o = ((int64_t)b * blocks_per_chunk + blk) * nout + threadIdx.x;
        psum[o] = s;
        psq[o] = q;
    }
}

// ------------------------------------------------------------------------------------------
// Kernel: second half of the folded tail layer (c8.conv0, Cin -> <= 5 channels).  The MFMA pass
// (TX = 1) produced P[v][dx*nout + o] = sum over (dz, dy, c) of in[v + (dz,dy,0)][c] * w[o][c][dz][dy][dx];
// here out[v][o] = bias[o] + sum over dx of P[v + dx*ex][dx*nout + o] (x-neighbours outside the chunk
// contribute zero = the convolution's zero padding), plus the per-block BN partial sums.
// ------------------------------------------------------------------------------------------
template <int NOUT>
__global__ __launch_bounds__(256) void tail_combine_kernel(const float *__restrict__ P, const float *__restrict__ bias,
                                                           int W, int64_t M, float *__restrict__ out,
                                                           float *__restrict__ psum, float *__restrict__ psq,
                                                           int blocks_per_chunk)
{
    // Each lane loads its own 16-float row of P with four coalesced float4 loads; the x-neighbours' partial sums come
    // from the adjacent lanes by shuffle (only lanes 0 and 63 load their outer neighbour's five values).  The scalar
    // version issued 15 loads of 4 bytes at a 64-byte lane stride: 1.8 ms per 64 chunks.
    __shared__ float red[4 * 8 * 2];
    const int b = blockIdx.x / blocks_per_chunk, blk = blockIdx.x % blocks_per_chunk;
    const int64_t m = (int64_t)blk * 256 + threadIdx.x;
    const bool live = m < M;
    const int lane = threadIdx.x & 63;
    const int64_t mc = live ? m : M - 1;
    const float4 *row = (const float4 *)(P + ((int64_t)b * M + mc) * 16);
    const float4 q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3];
    const float p[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    const int x = (int)(mc % W);
    constexpr int nout = NOUT;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
        // left neighbour's column o (its dx = +1 ... stored at [o]) and right neighbour's column 2*nout + o
        float l = __shfl_up(p[o], 1), r = __shfl_down(p[2 * nout + o], 1);
        if (lane == 0 && live && x > 0) l = P[((int64_t)b * M + m - 1) * 16 + o];
        if (lane == 63 && live && x < W - 1 && m + 1 < M) r = P[((int64_t)b * M + m + 1) * 16 + 2 * nout + o];
        float v = bias[o] + p[nout + o];
        if (x > 0) v += l;
        if (x < W - 1) v += r;
        acc[o] = v;
    }
    if (live)
        for (int o = 0; o < nout; ++o) out[((int64_t)b * M + m) * nout + o] = acc[o];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 0; o < 5; ++o) {
        float sm = live ? acc[o] : 0.f, q = live ? acc[o] * acc[o] : 0.f;
        for (int off = 32; off > 0; off >>= 1) { sm += __shfl_xor(sm, off); q += __shfl_xor(q, off); }
        if (lane == 0) { red[(wave * 8 + o) * 2] = sm; red[(wave * 8 + o) * 2 + 1] = q; }
    }
    __syncthreads();
    if ((int)threadIdx.x < nout) {
        float sm = 0.f, q = 0.f;
        for (int wv = 0; wv < 4; ++wv) { sm += red[(wv * 8 + threadIdx.x) * 2]; q += red[(wv * 8 + threadIdx.x) * 2 + 1]; }
        int64_t o = ((int64_t)b * blocks_per_chunk + blk) * nout + threadIdx.x;
        psum[o] = sm;
        psq[o] = q;
    }
}

// ------------------------------------------------------------------------------------------
// Kernel: final BatchNorm + activation (sigmoid unless chan_final_activations says otherwise); writes the chunk output as
// NCDHW (optional) and/or its centre crop into the (C, Z, Y, X) feature map (process_chunks, predict.py:82-95).
// ------------------------------------------------------------------------------------------
__global__ void final_kernel(const float *__restrict__ raw, const float *__restrict__ scale,
                             const float *__restrict__ shift, int nout, int act, int c_off, int c_total, int cz, int cy, int cx,
                             const int32_t *__restrict__ table, float *__restrict__ out_ncdhw,
                             float *__restrict__ fm, int64_t Z, int64_t Y, int64_t X, int64_t total)
{
    const uint32_t i32 = blockIdx.x * blockDim.x + threadIdx.x;     // 32-bit index arithmetic (see pool_kernel)
    if (i32 >= (uint32_t)total) return;
    const int64_t i = i32;
    const uint32_t M32 = (uint32_t)cz * cy * cx;
    const int64_t M = M32;
    const int b = (int)(i32 / M32);
    const uint32_t m = i32 % M32;
    const int x = (int)(m % (uint32_t)cx), y = (int)((m / (uint32_t)cx) % (uint32_t)cy), z = (int)(m / ((uint32_t)cx * cy));
    const int32_t *r = table + 9 * b;
    bool in_crop = fm && z >= r[3] && z < r[4] && y >= r[5] && y < r[6] && x >= r[7] && x < r[8];
    float a[8];
    if (act != ISX_FINAL_SIGMOID) {                     // (the default keeps its own loop: one channel live at a time)
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = c < nout ? fmaf(raw[i * nout + c], scale[b * nout + c], shift[b * nout + c]) : 0.f;
        final_act_apply(a, nout, act);
    }
    for (int c = 0; c < nout; ++c) {
        float v;
        if (act == ISX_FINAL_SIGMOID) {
            v = fmaf(raw[i * nout + c], scale[b * nout + c], shift[b * nout + c]);
            v = 1.f / (1.f + expf(-v));
        } else {
            v = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) if (k == c) v = a[k];
        }
        if (out_ncdhw) out_ncdhw[((int64_t)b * c_total + c_off + c) * M + m] = v;        // forks are concatenated (unet.py:322)
        if (in_crop) fm[(((int64_t)(c_off + c) * Z + (r[0] + z)) * Y + (r[1] + y)) * X + (r[2] + x)] = v;
    }
}

// materialise an intermediate as NCDHW for isx_unet_tap_host (tests): mode 0 raw, 1 affine,
// 2 affine+relu, 3 affine+final activation `act` (row-major source, C <= 8)
__global__ void tap_kernel(const float *__restrict__ src, const float *__restrict__ scale,
                           const float *__restrict__ shift, int C, int64_t M, int mode, int act, float *__restrict__ dst, int planar)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * C) return;
    int c = (int)(i % C);
    int64_t m = i / C;
    float v = planar ? src[((int64_t)(c >> 3) * M + m) * 8 + (c & 7)] : src[i];      // (UNetState::planar)
    if (mode >= 1) v = fmaf(v, scale[c], shift[c]);
    if (mode == 2) v = fmaxf(v, 0.f);
    if (mode == 3) {
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = k < C ? fmaf(src[m * C + k], scale[k], shift[k]) : 0.f;
        final_act_apply(a, C, act);
#pragma unroll
        for (int k = 0; k < 8; ++k) if (k == c) v = a[k];
    }
    dst[(int64_t)c * M + m] = v;
}

__global__ void absmax_kernel(const float *__restrict__ a, int64_t n, uint32_t *mx)
{
    uint32_t hi = 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t bits = __float_as_uint(a[i]);
        uint32_t u = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
        hi = u > hi ? u : hi;
    }
    for (int o = 32; o > 0; o >>= 1) { uint32_t h2 = __shfl_down(hi, o); hi = h2 > hi ? h2 : hi; }
    if ((threadIdx.x & 63) == 0) atomicMax(mx, hi);
}

__global__ void divide_kernel(float *__restrict__ a, int64_t n, const uint32_t *mx)
{
    uint32_t u = *mx;
    float m = __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        a[i] = a[i] / m;
}

// ------------------------------------------------------------------------------------------
// Host side: weights
// ------------------------------------------------------------------------------------------
static int dev_upload(UNetState *u, const std::vector<float> &h, float **out)
{
    void *p = nullptr;
    ISX_HIP(hipMalloc(&p, (h.size() ? h.size() : 1) * sizeof(float)));
    u->wblocks.push_back(p);
    if (!h.empty()) ISX_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = (float *)p;
    return ISX_OK;
}

int dev_alloc(UNetState *u, size_t floats, float **out)
{
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, (floats ? floats : 1) * sizeof(float));
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu floats) failed: %s", floats, hipGetErrorString(e));
        return ISX_ERR_NOMEM;
    }
    u->blocks.push_back(p);
    *out = (float *)p;
    return ISX_OK;
}

static const HostTensor *need(UNetState *u, const std::string &key, std::vector<int64_t> shape)
{
    auto it = u->host.find(key);
    if (it == u->host.end()) { set_error("state_dict: missing key %s", key.c_str()); return nullptr; }
    if (it->second.shape != shape) { set_error("state_dict: size mismatch for %s", key.c_str()); return nullptr; }
    return &it->second;
}

bool use_mfma(int cin, int cout) { return cin >= 32 && (cin % 4) == 0 && (cout % 32) == 0; }
// thin layer with many inputs: x-taps folded into the MFMA N tile (3 * cout <= 16 columns)
bool use_fold(int cin, int cout) { return cin >= 32 && (cin % 4) == 0 && cout <= 5; }

static int commit_weights(isx_ctx *ctx, const int *fork_channels, int nforks)
{
    UNetState *u = ctx->unet;
    ISX_REQUIRE(nforks >= 1 && nforks <= ISX_MAX_FORKS, "1 or 2 decoders (the reference's decoder() handles forks 0 and 1, unet.py:330-362)");
    int out_channels = 0;
    for (int f = 0; f < nforks; ++f) {
        ISX_REQUIRE(fork_channels[f] >= 1 && fork_channels[f] <= 8, "decoder %d: out_channels must be 1..8", f);
        u->fork_channels[f] = fork_channels[f];
        out_channels += fork_channels[f];
    }
    ISX_REQUIRE(out_channels <= 8, "at most 8 output channels over all decoders");
    u->out_channels = out_channels;
    u->nforks = nforks;
    u->nlayers = 18 + 8 * (nforks - 1);
    size_t expected = 0;
    for (int mx = 0; mx < 9 + 4 * (nforks - 1); ++mx) {
        const int fork = mx < 9 ? 0 : 1, mi = mx < 9 ? mx : mx - 4;              // mi: table index (c0..c8)
        const std::string mname = mi < 5 ? std::string(MODS[mi]) : std::string(MODS[mi]).substr(0, 2) + "_" + std::to_string(fork);
        int cin = MOD_CIN[mi], cout = MOD_COUT[mi] < 0 ? fork_channels[fork] : MOD_COUT[mi];
        for (int j = 0; j < 2; ++j) {
            ConvBN &L = u->cv[(mx < 9 ? 2 * mx : 18 + 2 * (mx - 9)) + j];
            L.conv = mname + ".conv" + std::to_string(j);
            L.bn = mname + ".batch" + std::to_string(j);
            L.cin = j == 0 ? cin : cout;
            L.cout = cout;
            const HostTensor *w = need(u, L.conv + ".weight", {cout, L.cin, 3, 3, 3});
            const HostTensor *bs = need(u, L.conv + ".bias", {cout});
            const HostTensor *g = need(u, L.bn + ".weight", {cout});
            const HostTensor *be = need(u, L.bn + ".bias", {cout});
            const HostTensor *rm = need(u, L.bn + ".running_mean", {cout});
            const HostTensor *rv = need(u, L.bn + ".running_var", {cout});
            if (!w || !bs || !g || !be || !rm || !rv) return ISX_ERR_STATE;
            auto nb = u->host.find(L.bn + ".num_batches_tracked");
            if (nb == u->host.end()) { set_error("state_dict: missing key %s.num_batches_tracked", L.bn.c_str()); return ISX_ERR_STATE; }
            L.nbt = nb->second.i64;
            expected += 7;
            // ---- repack -------------------------------------------------------------------
            std::vector<float> pk;
            const int ci_n = L.cin;
            if (use_mfma(L.cin, cout)) {
                const int BN = (cout % 64 == 0) ? 64 : 32;
                const int nsteps = ci_n / 4;
                pk.assign((size_t)cout * ci_n * 27, 0.f);
                for (int co = 0; co < cout; ++co)
                    for (int ci = 0; ci < ci_n; ++ci)
                        for (int t = 0; t < 27; ++t) {
                            size_t dst = ((((size_t)(co / BN) * nsteps + ci / 4) * 27 + t) * 4 + ci % 4) * BN + co % BN;
                            pk[dst] = w->f[((size_t)co * ci_n + ci) * 27 + t];
                        }
            } else if (use_fold(L.cin, cout)) {   // folded tail: [Cin/4][9 (dz,dy)][4][32], column = dx*cout + o
                const int nsteps = ci_n / 4;
                pk.assign((size_t)nsteps * 9 * 4 * 32 + 2048, 0.f);
                for (int co = 0; co < cout; ++co)
                    for (int ci = 0; ci < ci_n; ++ci)
                        for (int t9 = 0; t9 < 9; ++t9)
                            for (int dxi = 0; dxi < 3; ++dxi)
                                pk[((((size_t)(ci / 4)) * 9 + t9) * 4 + ci % 4) * 32 + dxi * cout + co] =
                                    w->f[((size_t)co * ci_n + ci) * 27 + t9 * 3 + dxi];
            } else if (L.cin == 1) {          // conv_first: [27][32]
                pk.assign((size_t)27 * cout, 0.f);
                for (int co = 0; co < cout; ++co)
                    for (int t = 0; t < 27; ++t) pk[(size_t)t * cout + co] = w->f[(size_t)co * 27 + t];
            } else {                          // conv_thin: [27][Cin][8]
                ISX_REQUIRE(cout <= 8, "layer %s: no kernel for %d -> %d channels", L.conv.c_str(), L.cin, cout);
                pk.assign((size_t)27 * ci_n * 8, 0.f);
                for (int co = 0; co < cout; ++co)
                    for (int ci = 0; ci < ci_n; ++ci)
                        for (int t = 0; t < 27; ++t)
                            pk[((size_t)t * ci_n + ci) * 8 + co] = w->f[((size_t)co * ci_n + ci) * 27 + t];
            }
            ISX_TRY(dev_upload(u, pk, &L.w));
            float wabs = 0.f;
            for (float v : w->f) wabs = std::max(wabs, fabsf(v));
            const bool w_in_range = wabs < 6.0e4f;              // otherwise this layer stays on the fp32 kernel
            if (use_fold(L.cin, cout) && L.cin % 8 == 0 && w_in_range) {
                std::vector<uint16_t> hp;
                for (int pm = 0; pm <= 2; pm += 2) {              // split halves (their hi half serves plain f16), bf16
                    pack_h2_fold_weights(w->f.data(), cout, L.cin, hp, pm);
                    void *d = nullptr;
                    ISX_HIP(hipMalloc(&d, hp.size() * 2));
                    u->wblocks.push_back(d);
                    ISX_HIP(hipMemcpy(d, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
                    (pm == 0 ? L.w_h2_32 : L.w_bf_32) = d;
                }
            }
            if (use_mfma(L.cin, cout) && L.cin % 8 == 0 && wabs < H2_W_LIMIT) {       // split-f16 packs (both tile widths: the tile
                std::vector<uint16_t> hp;                       // geometry is only known once a chunk arrives)
                for (int bn = 32; bn <= 64; bn += 32) {
                    if (cout % bn) continue;
                    for (int pm = 0; pm <= 2; pm += 2) {
                        pack_h2_weights(w->f.data(), cout, L.cin, bn, hp, pm);
                        void *d = nullptr;
                        ISX_HIP(hipMalloc(&d, hp.size() * 2));
                        u->wblocks.push_back(d);
                        ISX_HIP(hipMemcpy(d, hp.data(), hp.size() * 2, hipMemcpyHostToDevice));
                        if (pm == 0) (bn == 64 ? L.w_h2_64 : L.w_h2_32) = d;
                        else (bn == 64 ? L.w_bf_64 : L.w_bf_32) = d;
                    }
                }
            }
            std::vector<float> b8(std::max(cout, 8), 0.f);
            std::copy(bs->f.begin(), bs->f.end(), b8.begin());
            ISX_TRY(dev_upload(u, b8, &L.bias));
            ISX_TRY(dev_upload(u, g->f, &L.gamma));
            ISX_TRY(dev_upload(u, be->f, &L.beta));
            L.gmax = L.bmax = 0.f;
            for (float v : g->f) L.gmax = std::max(L.gmax, fabsf(v));
            for (float v : be->f) L.bmax = std::max(L.bmax, fabsf(v));
            ISX_TRY(dev_upload(u, rm->f, &L.rmean));
            ISX_TRY(dev_upload(u, rv->f, &L.rvar));
        }
    }
    ISX_TRY(dev_upload(u, std::vector<float>(32, 0.f), &u->zeros32));
    for (int k = 0; k < 4; ++k) {
        UpConv &U = u->up[k];
        U.name = UPS[k];
        U.C = UP_C[k];
        U.kz = UP_KZ[k];
        const HostTensor *w = need(u, U.name + ".weight", {U.C, 1, U.kz, 2, 2});
        const HostTensor *bs = need(u, U.name + ".bias", {U.C});
        if (!w || !bs) return ISX_ERR_STATE;
        expected += 2;
        std::vector<float> pk((size_t)U.kz * 4 * U.C);
        for (int c = 0; c < U.C; ++c)
            for (int t = 0; t < U.kz * 4; ++t) pk[(size_t)t * U.C + c] = w->f[(size_t)c * U.kz * 4 + t];
        ISX_TRY(dev_upload(u, pk, &U.w));
        ISX_TRY(dev_upload(u, bs->f, &U.bias));
        U.wmax = U.bmax = 0.f;
        for (float v : w->f) U.wmax = std::max(U.wmax, fabsf(v));
        for (float v : bs->f) U.bmax = std::max(U.bmax, fabsf(v));
    }
    if (u->host.size() != expected) {
        set_error("state_dict: %zu unexpected key(s)", u->host.size() - expected);
        return ISX_ERR_STATE;
    }
    u->committed = true;
    return ISX_OK;
}

// ------------------------------------------------------------------------------------------
// Host side: workspace and the forward pass
// ------------------------------------------------------------------------------------------
int conv_mode(const Level &l)
{
    if (l.W > 88 || (l.W % 8 == 0 && l.H % 8 == 0)) return 0;
    return (l.D % 2 == 0) ? 1 : 2;
}

int conv_tiles(const Level &l, int mode, int mt, int *nty, int *ntx, int *ntp, int *slots)
{
    if (mode == 0) {
        const int txw = 8 * mt;
        *nty = (l.H + 7) / 8;
        *ntx = (l.W + txw - 1) / txw;
        *ntp = 0;
        *slots = 4 * 10 * (txw + 2);
        return ((l.D + 1) / 2) * *nty * *ntx;
    }
    int TZ = mode == 1 ? 2 : 1, TP = 128 * mt / TZ;
    *nty = *ntx = 0;
    *ntp = (l.H * l.W + TP - 1) / TP;
    int Lh = TP + 2 * ((TP + l.W - 1) / l.W + 1) + 2 * (l.W + 2) + 2;
    *slots = (TZ + 2) * Lh;
    return ((l.D + TZ - 1) / TZ) * *ntp;
}

static int layer_tiles(const UNetState *u, int li, const Level &l, int batch)
{
    const ConvBN &L = u->cv[li];
    if (use_mfma(L.cin, L.cout)) {
        int a, b, c, d;
        return conv_tiles(l, conv_mode(l), 1, &a, &b, &c, &d);
    }
    if (L.cin == 1) return (int)((l.M + 127) / 128);
    if (L.cin == 5 && L.cout == 5) return ((l.D + 1) / 2) * ((l.H + 7) / 8) * ((l.W + 15) / 16);     // conv55 tiles
    const int blocks = (int)((l.M + 255) / 256);      // thin kernel blocks, or combine blocks of the folded layer
    return use_fold(L.cin, L.cout) ? std::max(blocks, conv_tail_tiles(l)) : blocks;        // (or the tiles of the one-pass tail kernel)
}

int prepare_workspace(isx_ctx *ctx, int cz, int cy, int cx, int batch)
{
    UNetState *u = ctx->unet;
    if (u->wz == cz && u->wy == cy && u->wx == cx && u->wbatch >= batch) return ISX_OK;
    ISX_REQUIRE(cz >= 2 && cz % 2 == 0 && cy >= 16 && cx >= 16 && cy % 16 == 0 && cx % 16 == 0,
                "chunk (%d,%d,%d): the U-Net needs an even z and y, x multiples of 16 (unet.py:328-345)", cz, cy, cx);
    if (u->raw[0]) {                 // new geometry: drop the old workspace
        ISX_HIP(hipStreamSynchronize(ctx->stream));
        for (void *p : u->blocks) (void)hipFree(p);
        u->blocks.clear();
        u->raw[0] = nullptr;
        u->last_n = 0;
    }
    Level *lv = u->lv;
    lv[0] = {cz, cy, cx, 0};
    for (int k = 1; k <= 4; ++k) {
        lv[k].D = (k == 4) ? lv[k - 1].D / 2 : lv[k - 1].D;
        lv[k].H = lv[k - 1].H / 2 + 1;
        lv[k].W = lv[k - 1].W / 2 + 1;
    }
    for (int k = 0; k <= 4; ++k) lv[k].M = (int64_t)lv[k].D * lv[k].H * lv[k].W;
    const int B = batch;
    ISX_TRY(dev_alloc(u, (size_t)B * lv[0].M, &u->x0));
    for (int li = 0; li < u->nlayers; ++li) {
        ConvBN &L = u->cv[li];
        const Level &l = lv[MOD_LEVEL[layer_module(li)]];
        ISX_TRY(dev_alloc(u, (size_t)B * l.M * L.cout, &u->raw[li]));
        L.tiles = layer_tiles(u, li, l, B);
        ISX_TRY(dev_alloc(u, (size_t)B * L.tiles * L.cout, &L.psum));
        ISX_TRY(dev_alloc(u, (size_t)B * L.tiles * L.cout, &L.psq));
        ISX_TRY(dev_alloc(u, (size_t)B * L.cout, &L.scale));
        ISX_TRY(dev_alloc(u, (size_t)B * L.cout, &L.shift));
        ISX_TRY(dev_alloc(u, (size_t)B * L.cout, &L.bmean));
        ISX_TRY(dev_alloc(u, (size_t)B * L.cout, &L.bvar));
        ISX_TRY(dev_alloc(u, (size_t)B * L.cout, &L.invstd));
    }
    ISX_TRY(dev_alloc(u, (size_t)B * lv[0].M * 16 + 64, &u->ptail));
    for (int k = 0; k < 4; ++k) {
        ISX_TRY(dev_alloc(u, (size_t)B * lv[k + 1].M * MOD_COUT[k], &u->pool[k]));       // pooled module k output
        ISX_TRY(dev_alloc(u, (size_t)B * lv[3 - k].M * UP_C[k], &u->ups[k]));            // up_k output at level 3-k
        if (u->nforks > 1 && k > 0) ISX_TRY(dev_alloc(u, (size_t)B * lv[3 - k].M * UP_C[k], &u->ups[ups_index(1, k)]));
    }
    u->wz = cz; u->wy = cy; u->wx = cx; u->wbatch = B;
    return ISX_OK;
}


int launch_conv_mfma(isx_ctx *ctx, const Level &l, int B, Src sa, Src sb, const float *w_packed, const float *bias,
                     int cout, float *out, float *psum, float *psq, int no_stats)
{
    ConvArgs a;
    a.srcA = sa.p; a.scaleA = sa.scale; a.shiftA = sa.shift; a.CA = sa.C;
    a.srcB = sb.p; a.scaleB = sb.scale; a.shiftB = sb.shift; a.CB = sb.C;
    a.w = w_packed; a.bias = bias; a.out = out; a.psum = psum; a.psq = psq;
    a.D = l.D; a.H = l.H; a.W = l.W; a.Cout = cout;
    a.mode = conv_mode(l);
    a.tiles_per_chunk = conv_tiles(l, a.mode, 1, &a.nty, &a.ntx, &a.ntp, &a.slots_max);
    a.store_cols = cout;
    a.no_stats = no_stats;
    a.xcd_remap = ctx->tune.xcd_remap;
    ISX_REQUIRE(a.slots_max <= 1024, "internal: halo tile of %d slots does not fit the staging plan", a.slots_max);
    ISX_REQUIRE(cout % 32 == 0 && (sa.C + sb.C) % 4 == 0, "internal: MFMA conv needs Cout %% 32 == 0, Cin %% 4 == 0");
    const int BN = (cout % 64 == 0) ? 64 : 32;
    size_t lds = (size_t)(27 * 4 * BN + a.slots_max * A_STRIDE + 128 + 4 * BN * 2) * sizeof(float);
    a.ntiles = cout / BN;
    const unsigned grid = (unsigned)(B * a.tiles_per_chunk * a.ntiles);
    if (BN == 64) conv_mfma_kernel<64, 3, 1><<<grid, 256, lds, ctx->stream>>>(a);
    else conv_mfma_kernel<32, 3, 1><<<grid, 256, lds, ctx->stream>>>(a);
    ISX_CHECK_LAUNCH();
    return ISX_OK;
}

// Will run_conv give layer li (sources bounded by `bound`) to the reduced-precision kernel (conv_h2.hip)?  The conditions are
// run_conv's own.  Only that kernel reads a planar source, so a producer may write one only if this says yes for its reader.
static bool conv_takes_h2(isx_ctx *ctx, int li, const Level &l, int B, float bound)
{
    UNetState *u = ctx->unet;
    const ConvBN &L = u->cv[li];
    const bool bf = u->precision == ISX_PRECISION_BF16;
    if (u->precision == ISX_PRECISION_FP32 || !(bf || bound < 6.0e4f)) return false;
    if (use_mfma(L.cin, L.cout)) {
        if (!bf && !(bound < H2_ACT_LIMIT)) return false;         // the 27-tap kernels scale their operands (h2_split.h)
        int t0, t1, t2, slots;
        conv_tiles(l, conv_mode(l), 1, &t0, &t1, &t2, &slots);
        const bool b64 = conv_h2_bn(ctx->tune, slots, L.cout) == 64 || conv_h2_nw8(ctx->tune, l, L.cout, B);
        const void *wh = bf ? (b64 ? L.w_bf_64 : L.w_bf_32) : (b64 ? L.w_h2_64 : L.w_h2_32);
        return (!u->train || bf) && wh;
    }
    if (use_fold(L.cin, L.cout)) return !u->train && (bf ? L.w_bf_32 : L.w_h2_32);
    return false;
}

// want_planar: write raw[li] as [chunk][C/8][voxel][8] if this layer runs on the reduced-precision kernel (the caller has
// checked that its only reader does: conv_takes_h2); UNetState::planar[li] records what was written.
static int run_conv(isx_ctx *ctx, int li, const Level &l, int B, Src sa, Src sb, const H2Extra *ex = nullptr, bool want_planar = false)
{
    UNetState *u = ctx->unet;
    ConvBN &L = u->cv[li];
    const int cin = sa.C + sb.C;
    int stat_tiles = L.tiles;                 // tiles per chunk in the partial-sum layout the conv kernel wrote
    int stat_tile_fastest = 0;                // ... which is [chunk][channel][tile] for conv_first_kernel and the conv_h2 kernels
    ISX_REQUIRE(cin == L.cin, "internal: layer %s fed %d channels, expects %d", L.conv.c_str(), cin, L.cin);
    u->planar[li] = false;
    if (use_mfma(L.cin, L.cout)) {
        ProfScope ps(ctx, ISX_PROF_CONV_MFMA);
        int t0, t1, t2, slots;
        conv_tiles(l, conv_mode(l), 1, &t0, &t1, &t2, &slots);
        const bool bf = u->precision == ISX_PRECISION_BF16;
        const bool b64 = conv_h2_bn(ctx->tune, slots, L.cout) == 64 || conv_h2_nw8(ctx->tune, l, L.cout, B);
        const void *wh = bf ? (b64 ? L.w_bf_64 : L.w_bf_32) : (b64 ? L.w_h2_64 : L.w_h2_32);
        const bool in_range = bf || std::max(sa.bound, sb.bound) < H2_ACT_LIMIT;    // else the scaled operands could leave the f16 range
        // training keeps fp32 products except in the bf16 mixed-precision mode (its packs are refreshed every step)
        const bool reduced = u->precision != ISX_PRECISION_FP32 && (!u->train || bf);
        const int pm = u->precision == ISX_PRECISION_SPLIT_F16 ? 0 : (bf ? 2 : 1);
        const bool h2 = reduced && wh && in_range;
        u->arith[li] = h2 ? ISX_ARITH_REDUCED : (reduced && wh && !in_range ? ISX_ARITH_FP32_RANGE_FALLBACK : ISX_ARITH_FP32);
        ISX_REQUIRE(h2 || !(sa.planar || sb.planar), "internal: a planar source reached a convolution that does not read it");
        if (h2) {
            const int out_planar = (want_planar && L.cout % 8 == 0) ? 1 : 0;
            u->planar[li] = out_planar != 0;
            ISX_TRY(launch_conv_h2(ctx, l, B, sa, sb, wh, L.bias, L.cout, u->raw[li], L.psum, L.psq, 0, ctx->tune.xcd_remap, &stat_tiles, pm, out_planar, ex));
            stat_tile_fastest = 1;
        } else {
            ISX_REQUIRE(!ex, "internal: up-sampling on read needs the reduced-precision convolution");
            ISX_TRY(launch_conv_mfma(ctx, l, B, sa, sb, L.w, L.bias, L.cout, u->raw[li], L.psum, L.psq, 0));
        }
    } else if (use_fold(L.cin, L.cout)) {
        // the x-taps folded into 3*cout <= 16 output columns, 9 (dz,dy) taps on the matrix cores
        const bool bf = u->precision == ISX_PRECISION_BF16;
        const void *wf = bf ? L.w_bf_32 : L.w_h2_32;
        const int pm = u->precision == ISX_PRECISION_SPLIT_F16 ? 0 : (bf ? 2 : 1);
        const bool fold_range = bf || std::max(sa.bound, sb.bound) < 6.0e4f;
        const bool fold_h2 = u->precision != ISX_PRECISION_FP32 && !u->train && wf && fold_range;
        u->arith[li] = fold_h2 ? ISX_ARITH_REDUCED
                               : (u->precision != ISX_PRECISION_FP32 && !u->train && wf && !fold_range ? ISX_ARITH_FP32_RANGE_FALLBACK : ISX_ARITH_FP32);
        ISX_REQUIRE(fold_h2 || !ex, "internal: up-sampling on read needs the reduced-precision tail kernel");
        ISX_REQUIRE(fold_h2 || !(sa.planar || sb.planar), "internal: a planar source reached a convolution that does not read it");
        // reduced precisions: one pass on the f16 / bf16 matrix cores, x-shift in LDS, no partial-sum tensor (conv_tail.hip)
        const bool one_pass = fold_h2;
        if (one_pass) {
            ProfScope ps(ctx, ISX_PROF_CONV_MFMA);
            ISX_TRY(launch_conv_tail(ctx, l, B, sa, sb, wf, L.bias, L.cout, u->raw[li], L.psum, L.psq, pm, &stat_tiles, ex));
        } else {
            ProfScope ps(ctx, ISX_PROF_CONV_MFMA);
            ConvArgs a;
            a.srcA = sa.p; a.scaleA = sa.scale; a.shiftA = sa.shift; a.CA = sa.C;
            a.srcB = sb.p; a.scaleB = sb.scale; a.shiftB = sb.shift; a.CB = sb.C;
            a.w = L.w; a.bias = u->zeros32; a.out = u->ptail; a.psum = nullptr; a.psq = nullptr;
            a.D = l.D; a.H = l.H; a.W = l.W; a.Cout = 32;
            a.mode = conv_mode(l);
            a.tiles_per_chunk = conv_tiles(l, a.mode, 1, &a.nty, &a.ntx, &a.ntp, &a.slots_max);
            a.store_cols = 16;
            a.no_stats = 1;
            a.xcd_remap = ctx->tune.xcd_remap;
            ISX_REQUIRE(a.slots_max <= 1024, "internal: halo tile of %d slots does not fit the staging plan", a.slots_max);
            size_t lds = (size_t)(9 * 4 * 32 + a.slots_max * A_STRIDE + 128 + 4 * 32 * 2) * sizeof(float);
            a.ntiles = 1;
            const unsigned grid = (unsigned)(B * a.tiles_per_chunk);
            conv_mfma_kernel<32, 1, 1><<<grid, 256, lds, ctx->stream>>>(a);
            ISX_CHECK_LAUNCH();
        }
        // pass 2: shifted combine + bias + BN partial sums
        if (!one_pass) {
            ProfScope ps(ctx, ISX_PROF_CONV_DIRECT);
#define ISX_TAIL(N_) tail_combine_kernel<N_><<<(unsigned)(B * L.tiles), 256, 0, ctx->stream>>>(u->ptail, L.bias, l.W, l.M, \
                                                                                         u->raw[li], L.psum, L.psq, L.tiles)
            switch (L.cout) {
                case 1: ISX_TAIL(1); break;
                case 2: ISX_TAIL(2); break;
                case 3: ISX_TAIL(3); break;
                case 4: ISX_TAIL(4); break;
                default: ISX_TAIL(5); break;
            }
#undef ISX_TAIL
            ISX_CHECK_LAUNCH();
        }
    } else if (L.cin == 1) {
        ProfScope ps(ctx, ISX_PROF_CONV_DIRECT);
        ISX_REQUIRE(L.cout == 32 && !sa.scale, "internal: conv_first shape");
        u->planar[li] = want_planar;                   // (its only reader is the module's second convolution: planar_ok)
        conv_first_kernel<<<(unsigned)(B * L.tiles), 256, 0, ctx->stream>>>(sa.p, L.w, L.bias, l.D, l.H, l.W, u->raw[li],
                                                                           L.psum, L.psq, L.tiles, ctx->tune.conv_first_nt, want_planar ? 1 : 0);
        ISX_CHECK_LAUNCH();
        stat_tile_fastest = 1;
    } else {
        ProfScope ps(ctx, ISX_PROF_CONV_DIRECT);
        if (L.cin == 5 && L.cout == 5 && sa.C == 5 && sb.C == 0 && sa.scale) {
            ISX_TRY(launch_conv55(ctx, l, B, sa, L.w, L.bias, u->raw[li], L.psum, L.psq, L.tiles));
        } else {
        size_t lds = (size_t)(27 * L.cin * 8 + 64) * sizeof(float);
        ISX_REQUIRE(lds <= 160 * 1024, "internal: thin conv weights do not fit LDS");
        if (L.cout == 5)
            conv_thin_kernel<5><<<(unsigned)(B * L.tiles), 256, lds, ctx->stream>>>(
                sa.p, sa.scale, sa.shift, sa.C, sb.p, sb.scale, sb.shift, sb.C, L.w, L.bias, l.D, l.H, l.W, L.cout,
                u->raw[li], L.psum, L.psq, L.tiles);
        else
            conv_thin_kernel<8><<<(unsigned)(B * L.tiles), 256, lds, ctx->stream>>>(
                sa.p, sa.scale, sa.shift, sa.C, sb.p, sb.scale, sb.shift, sb.C, L.w, L.bias, l.D, l.H, l.W, L.cout,
                u->raw[li], L.psum, L.psq, L.tiles);
        ISX_CHECK_LAUNCH();
        }
    }
    {
        ProfScope ps(ctx, ISX_PROF_ELEMENTWISE);
        bn_finalize_kernel<<<dim3((unsigned)L.cout, (unsigned)B), 256, 0, ctx->stream>>>(
            L.psum, L.psq, stat_tiles, L.cout, (double)l.M, L.gamma, L.beta, L.scale, L.shift, L.bmean, L.bvar, L.invstd,
            B == 1 ? L.rmean : nullptr, B == 1 ? L.rvar : nullptr, stat_tile_fastest);
        ISX_CHECK_LAUNCH();
        if (B > 1) {
            if (ctx->tune.bn_run_fused) u->bn_pending.push_back(li);           // forward_batch updates them all at its end
            else {
                bn_running_kernel<<<cdiv(L.cout, 64), 64, 0, ctx->stream>>>(L.bmean, L.bvar, B, L.cout, L.rmean, L.rvar);
                ISX_CHECK_LAUNCH();
            }
        }
        L.nbt += B;
    }
    return ISX_OK;
}

// Forward B chunks whose rows are already in u->d_table and whose voxels are in u->x0.
int forward_batch(isx_ctx *ctx, int B, float *out_ncdhw, float *fm, const int64_t shape[3])
{
    UNetState *u = ctx->unet;
    const Level *lv = u->lv;
    const Src none{nullptr, nullptr, nullptr, 0};
    u->bn_pending.clear();
    // |relu(BN(x))| <= |gamma| * sqrt(M - 1) + |beta| for batch statistics over M values: the range check of the
    // split-f16 convolution (run_conv) uses these bounds, pooling keeps them, the up-convolution scales them
    auto bn_bound = [&](int li) { const ConvBN &L = u->cv[li]; return L.gmax * sqrtf((float)lv[MOD_LEVEL[layer_module(li)]].M) + L.bmax; };
    auto act = [&](int li) { const ConvBN &L = u->cv[li]; return Src{u->raw[li], L.scale, L.shift, L.cout, bn_bound(li), u->planar[li]}; };
    // the first convolution of a module has one reader, the second one: if that reader runs on the reduced-precision kernel the
    // tensor between them is written planar (whole-line halo loads for every K stage of the reader)
    auto planar_ok = [&](int li_reader, const Level &l, int li_writer) {
        return ctx->tune.planar && !u->train && u->cv[li_writer].cout % 8 == 0 && conv_takes_h2(ctx, li_reader, l, B, bn_bound(li_writer));
    };
    // ISX_PLANAR >= 2: the skip tensors (second convolution of c0..c3: read by the pooling and by the decoder's first convolution
    // of that level) and the tensors that feed / leave up1..up3 are planar as well, whenever every reader can take one -- the
    // tail layer reads both its sources that way (conv_tail.hip: on row-major sources it re-fetched every 128-byte line of its
    // halo once per K stage from beyond L2)
    auto skip_planar_ok = [&](int mi) {
        if (ctx->tune.planar < 2 || u->train || mi > 3 || u->cv[2 * mi + 1].cout % 8) return false;
        const int k = 3 - mi;                                     // the decoder module that reads this skip tensor beside up_k
        for (int f = 0; f < u->nforks; ++f) {
            const int li_in = (k == 0) ? 9 : dec_layer(f, k - 1, 1);
            const float up_bound = bn_bound(li_in) * u->up[k].wmax + u->up[k].bmax;
            if (!conv_takes_h2(ctx, dec_layer(f, k, 0), lv[mi], B, std::max(up_bound, bn_bound(2 * mi + 1)))) return false;
        }
        return conv_takes_h2(ctx, 2 * (mi + 1), lv[mi + 1], B, bn_bound(2 * mi + 1));       // the pooled tensor's reader
    };
    // ---- encoder ----------------------------------------------------------------------------
    for (int mi = 0; mi <= 4; ++mi) {
        const Level &l = lv[mi];
        Src in = (mi == 0) ? Src{u->x0, nullptr, nullptr, 1}
                           : Src{u->pool[mi - 1], nullptr, nullptr, MOD_COUT[mi - 1], bn_bound(2 * (mi - 1) + 1)};
        if (mi > 0) { in.planar = u->pool_planar[mi - 1]; in.presplit = u->pool_presplit[mi - 1]; }
        ISX_TRY(run_conv(ctx, 2 * mi, l, B, in, none, nullptr, planar_ok(2 * mi + 1, l, 2 * mi)));
        ISX_TRY(run_conv(ctx, 2 * mi + 1, l, B, act(2 * mi), none, nullptr, skip_planar_ok(mi)));
        if (mi < 4) {
            ProfScope ps(ctx, ISX_PROF_ELEMENTWISE);
            const Level &lo = lv[mi + 1];
            const ConvBN &L = u->cv[2 * mi + 1];
            int64_t total = (int64_t)B * lo.M * (L.cout / 4);
            ISX_REQUIRE(total < (int64_t)0xFFFFFFFFll, "internal: launch set too large for 32-bit element indices");
            // a planar skip tensor is pooled plane by plane into a planar tensor (skip_planar_ok has checked its reader)
            const bool sp = u->planar[2 * mi + 1];
            u->pool_planar[mi] = sp;
            // pre-split for its reader (skip_planar_ok has checked that the reader runs on the reduced-precision kernel): only in
            // the split-f16 arithmetic, whose staging would compute exactly these halves
            u->pool_presplit[mi] = sp && ctx->tune.h2_presplit && u->precision == ISX_PRECISION_SPLIT_F16;
            if (sp) {
                total = (int64_t)B * lo.M * (L.cout / 8);
                pool_planar_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(u->raw[2 * mi + 1], L.scale, L.shift, L.cout, l.D, l.H, l.W,
                                                                           mi == 3 ? 2 : 1, lo.D, lo.H, lo.W, u->pool[mi], total,
                                                                           u->pool_presplit[mi] ? 1 : 0);
            } else
            pool_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(u->raw[2 * mi + 1], L.scale, L.shift, L.cout, l.D, l.H,
                                                                l.W, mi == 3 ? 2 : 1, lo.D, lo.H, lo.W, u->pool[mi],
                                                                total);
            ISX_CHECK_LAUNCH();
        }
    }
    // ---- decoder(s): every fork runs c5..c8 of its own on the shared encoder outputs and transposed convolutions; up0 of
    //      the bottleneck is the same tensor for both, the outputs are concatenated (unet.py:313-325) ---------------------
    int c_off = 0;
    for (int f = 0; f < u->nforks; ++f) {
        for (int k = 0; k < 4; ++k) {
            const int li_in = (k == 0) ? 9 : dec_layer(f, k - 1, 1);   // layer feeding up_k: c4, c5, c6, c7 (second conv)
            const Level &ls = lv[MOD_LEVEL[layer_module(li_in)]];   // source level
            const Level &lo = lv[3 - k];                          // output level (after crop)
            const UpConv &U = u->up[k];
            const ConvBN &Lin = u->cv[li_in];
            float *upo = u->ups[ups_index(f, k)];
            const int l0f = dec_layer(f, k, 0), skipf = 2 * (3 - k) + 1;
            ISX_REQUIRE(ls.D == lo.D || k == 0, "internal: up%d changes the depth", k);
            u->up_folded[ups_index(f, k)] = false;
            {
                // the tail layer (64 -> n, conv_tail.hip) up-samples its first source while staging it: up3 is the largest
                // transposed convolution (5.4 GB written and read back per 64-chunk launch set at 256^2)
                const ConvBN &L0 = u->cv[l0f];
                const bool bf = u->precision == ISX_PRECISION_BF16;
                const float bnd = bn_bound(li_in) * U.wmax + U.bmax;
                const bool on_read = ctx->tune.tail_up && U.kz == 1 && U.C % 8 == 0 && U.C <= 64 &&
                                     use_fold(L0.cin, L0.cout) && !u->train &&
                                     u->precision != ISX_PRECISION_FP32 && (bf ? L0.w_bf_32 : L0.w_h2_32) &&
                                     (bf || std::max(bnd, bn_bound(skipf)) < 6.0e4f);
                if (on_read) {
                    H2Extra ex;
                    ex.upA = 1; ex.uH = ls.H; ex.uW = ls.W; ex.uoff = k == 3 ? 1 : 0; ex.upw = U.w; ex.upb = U.bias;
                    u->up_folded[ups_index(f, k)] = true;
                    u->ups_planar[ups_index(f, k)] = false;
                    ISX_TRY(run_conv(ctx, l0f, lo, B, Src{u->raw[li_in], Lin.scale, Lin.shift, U.C, bnd, u->planar[li_in]}, act(skipf), &ex));
                    ISX_TRY(run_conv(ctx, dec_layer(f, k, 1), lo, B, act(l0f), none));
                    continue;
                }
            }
            const int l0 = dec_layer(f, k, 0), l1 = dec_layer(f, k, 1);   // c5..c8 of this fork
            const int skip_li = 2 * (3 - k) + 1;                   // c3, c2, c1, c0 second conv
            const float up_bound = bn_bound(li_in) * U.wmax + U.bmax;
            // up1 / up2 (kernel (1, 2, 2)) are up-sampled by their consumer while it stages its halo, where that consumer's launch has
            // the form for it (conv_h2_can_up: the 512-voxel kernel and the two-row-tile 32-column kernel): the up-sampled tensors
            // (21.6 + 42.6 MB per chunk) are never written or read, bit-identical operands (conv_h2.hip, UP)
            const int pm_k = u->precision == ISX_PRECISION_SPLIT_F16 ? 0 : (u->precision == ISX_PRECISION_BF16 ? 2 : 1);
            const bool on_read_h2 = ctx->tune.h2_up && (U.kz == 1 || (U.kz == 2 && ctx->tune.h2_up >= 2 && lo.D % 2 == 0)) && U.C % 8 == 0 && !u->train && use_mfma(u->cv[l0].cin, u->cv[l0].cout) &&
                                    conv_takes_h2(ctx, l0, lo, B, std::max(up_bound, bn_bound(skip_li))) &&
                                    conv_h2_can_up(ctx, lo, B, u->cv[l0].cout, pm_k, U.C, U.kz);
            if (on_read_h2) {
                u->up_folded[ups_index(f, k)] = true;
                u->ups_planar[ups_index(f, k)] = false;
            } else
            if (!(f > 0 && k == 0)) {
                ProfScope ps(ctx, ISX_PROF_ELEMENTWISE);
                int64_t total = (int64_t)B * lo.M * (U.C / 4);
                ISX_REQUIRE(total < (int64_t)0xFFFFFFFFll, "internal: launch set too large for 32-bit element indices");
                const int64_t total_src = (int64_t)B * ls.M * (U.C / 4);
                const bool per_src = U.kz == 1 && ctx->tune.up_per_source && total_src < (int64_t)0xFFFFFFFFll;
                // up1..up3 (per source voxel): planar out iff planar in (the producing convolution checked the reader of the
                // up-sampled tensor); up0 (kz = 2, per output voxel): planar out whenever its reader takes one
                bool up0_planar = !per_src && ctx->tune.planar >= 2 && !u->train && U.C % 8 == 0;
                for (int ff = 0; ff < u->nforks && up0_planar; ++ff)         // (up0's tensor is shared by the forks)
                    up0_planar = conv_takes_h2(ctx, dec_layer(ff, k, 0), lo, B, std::max(up_bound, bn_bound(skip_li)));
                u->ups_planar[ups_index(f, k)] = per_src ? u->planar[li_in] : up0_planar;
                if (per_src && u->planar[li_in]) {
                    const int64_t total_pl = (int64_t)B * (U.C / 8) * ls.D * ls.H * lo.W * 2;
                    ISX_REQUIRE(total_pl < (int64_t)0xFFFFFFFFll, "internal: launch set too large for 32-bit element indices");
                    upsample_src_planar_kernel<<<cdiv(total_pl, 256), 256, 0, ctx->stream>>>(u->raw[li_in], Lin.scale, Lin.shift, U.C, ls.D, ls.H, ls.W,
                                                                                          k == 3 ? 1 : 0, U.w, U.bias, lo.H, lo.W, upo, total_pl);
                } else if (per_src)
                    upsample_src_kernel<<<cdiv(total_src, 256), 256, 0, ctx->stream>>>(u->raw[li_in], Lin.scale, Lin.shift, U.C, ls.D, ls.H, ls.W,
                                                                                    k == 3 ? 1 : 0, U.w, U.bias, lo.H, lo.W, upo, total_src);
                else
                upsample_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(u->raw[li_in], Lin.scale, Lin.shift, U.C, ls.D,
                                                                        ls.H, ls.W, U.kz, k == 3 ? 1 : 0, U.w, U.bias, lo.D,
                                                                        lo.H, lo.W, upo, total, 1 /* streaming stores: the 5.4 GB output is far larger than L2 */,
                                                                        u->planar[li_in] ? 1 : 0, u->ups_planar[ups_index(f, k)] ? 1 : 0);
                ISX_CHECK_LAUNCH();
            }
            if (on_read_h2) {
                H2Extra ex;
                ex.upA = 1; ex.uH = ls.H; ex.uW = ls.W; ex.uoff = k == 3 ? 1 : 0; ex.ukz = U.kz; ex.upw = U.w; ex.upb = U.bias;
                ISX_TRY(run_conv(ctx, l0, lo, B, Src{u->raw[li_in], Lin.scale, Lin.shift, U.C, up_bound, u->planar[li_in]}, act(skip_li), &ex, planar_ok(l1, lo, l0)));
            } else
            ISX_TRY(run_conv(ctx, l0, lo, B, Src{upo, nullptr, nullptr, U.C, up_bound, u->ups_planar[ups_index(f, k)]}, act(skip_li), nullptr, planar_ok(l1, lo, l0)));
            // its output feeds up_{k+1} only (k < 3): planar if that transposed convolution runs per source voxel and the
            // convolution behind it reads planar sources
            bool next_per_src = k < 3 && ctx->tune.planar >= 2 && !u->train &&
                                u->up[k + 1].kz == 1 && ctx->tune.up_per_source && u->cv[l1].cout % 8 == 0 &&
                                (int64_t)B * lo.M * (u->cv[l1].cout / 4) < (int64_t)0xFFFFFFFFll;
            if (next_per_src) {
                const float nb = bn_bound(l1) * u->up[k + 1].wmax + u->up[k + 1].bmax;
                next_per_src = conv_takes_h2(ctx, dec_layer(f, k + 1, 0), lv[3 - (k + 1)], B, std::max(nb, bn_bound(2 * (3 - (k + 1)) + 1)));
            }
            ISX_TRY(run_conv(ctx, l1, lo, B, act(l0), none, nullptr, next_per_src));
        }
        {
            ProfScope ps(ctx, ISX_PROF_ELEMENTWISE);
            const int lf = dec_layer(f, 3, 1);
            const ConvBN &L = u->cv[lf];
            int64_t total = (int64_t)B * lv[0].M;
            ISX_REQUIRE(total < (int64_t)0xFFFFFFFFll, "internal: launch set too large for 32-bit element indices");
            final_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(u->raw[lf], L.scale, L.shift, L.cout, u->final_act[f], c_off, u->out_channels,
                                                                 lv[0].D, lv[0].H, lv[0].W, u->d_table, out_ncdhw, fm,
                                                                 shape ? shape[0] : 0, shape ? shape[1] : 0, shape ? shape[2] : 0, total);
            ISX_CHECK_LAUNCH();
            c_off += L.cout;
        }
    }
    if (!u->bn_pending.empty()) {                       // the running statistics of every layer of this pass (run_conv deferred them)
        ProfScope ps(ctx, ISX_PROF_ELEMENTWISE);
        BnRunAll t;
        int n = 0, maxc = 0;
        for (int li : u->bn_pending) {
            const ConvBN &L = u->cv[li];
            t.bmean[n] = L.bmean; t.bvar[n] = L.bvar; t.rmean[n] = L.rmean; t.rvar[n] = L.rvar; t.C[n] = L.cout;
            maxc = std::max(maxc, L.cout);
            ++n;
        }
        u->bn_pending.clear();
        bn_running_all_kernel<<<dim3(cdiv(maxc, 64), (unsigned)n), 64, 0, ctx->stream>>>(t, B);
        ISX_CHECK_LAUNCH();
    }
    u->last_n = B;
    return ISX_OK;
}

static int check_ready(isx_ctx *ctx)
{
    ISX_REQUIRE(ctx, "null context");
    if (!ctx->unet || !ctx->unet->committed) {
        set_error("U-Net weights not loaded: call isx_load_weight_host for every key, then isx_weights_commit");
        return ISX_ERR_STATE;
    }
    return ISX_OK;
}

// rows: host int32 [n][9]; processes them in launch sets of `batch` chunks
static int run_chunks(isx_ctx *ctx, const float *volume, const int64_t shape[3], const int32_t *rows, int64_t n,
                      const int64_t chunk[3], float *out_ncdhw, float *fm)
{
    UNetState *u = ctx->unet;
    const int cz = (int)chunk[0], cy = (int)chunk[1], cx = (int)chunk[2];
    for (int64_t i = 0; i < n; ++i) {
        const int32_t *r = rows + 9 * i;
        ISX_REQUIRE(r[0] >= 0 && r[1] >= 0 && r[2] >= 0 && r[0] + cz <= shape[0] && r[1] + cy <= shape[1] &&
                        r[2] + cx <= shape[2],
                    "chunk %ld starts at (%d,%d,%d): outside the volume", (long)i, r[0], r[1], r[2]);
    }
    if (u->train && u->precision == ISX_PRECISION_BF16) ISX_TRY(train_ensure_bf16_packs(ctx));
    // workspace for min(batch, n) chunks: ~0.6 GB per (10,256,256) chunk, grown on demand
    ISX_TRY(prepare_workspace(ctx, cz, cy, cx, (int)std::min<int64_t>(u->batch, n)));
    const int64_t M0 = u->lv[0].M;
    Scratch<int32_t> dtab(ctx, (size_t)n * 9);
    if (!dtab.ok()) return ISX_ERR_NOMEM;
    ISX_HIP(hipMemcpyAsync(dtab.p, rows, (size_t)n * 9 * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    for (int64_t first = 0; first < n; first += u->wbatch) {
        int B = (int)std::min<int64_t>(u->wbatch, n - first);
        u->d_table = dtab.p + 9 * first;
        {
            ProfScope ps(ctx, ISX_PROF_ELEMENTWISE);
            int64_t total = (int64_t)B * M0;
            gather_chunks_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(volume, shape[1], shape[2], u->d_table, cz, cy,
                                                                         cx, u->x0, total);
            ISX_CHECK_LAUNCH();
        }
        ISX_TRY(forward_batch(ctx, B, out_ncdhw ? out_ncdhw + first * u->out_channels * M0 : nullptr, fm, shape));
    }
    return ISX_OK;
}

}  // namespace isx

using namespace isx;

extern "C" {

int isx_load_weight_host(isx_ctx *ctx, const char *key, const void *data, const int64_t *shape, int ndim)
{
    ISX_REQUIRE(ctx && key && data && ndim >= 0 && ndim <= 5 && (shape || ndim == 0), "bad argument");
    if (!ctx->unet) { ctx->unet = new UNetState(); ctx->unet->tune = ctx->tune; }
    UNetState *u = ctx->unet;
    if (u->committed) { set_error("weights already committed; create a new context to load another network"); return ISX_ERR_STATE; }
    HostTensor t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= shape[i]; }
    std::string k(key);
    if (k.size() > 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0) {
        t.is_i64 = true;
        t.i64 = *(const int64_t *)data;
    } else {
        t.f.assign((const float *)data, (const float *)data + n);
    }
    u->host[k] = std::move(t);
    return ISX_OK;
}

int isx_weights_commit(isx_ctx *ctx, int out_channels)
{
    ISX_REQUIRE(ctx, "null context");
    if (!ctx->unet) { set_error("no weights loaded"); return ISX_ERR_STATE; }
    ISX_HIP(hipSetDevice(ctx->device));
    return commit_weights(ctx, &out_channels, 1);
}

int isx_weights_commit_forked(isx_ctx *ctx, const int32_t *fork_channels, int nforks)
{
    ISX_REQUIRE(ctx && fork_channels, "null argument");
    if (!ctx->unet) { set_error("no weights loaded"); return ISX_ERR_STATE; }
    ISX_REQUIRE(nforks >= 1 && nforks <= ISX_MAX_FORKS, "1 or 2 decoders");
    ISX_HIP(hipSetDevice(ctx->device));
    int fc[ISX_MAX_FORKS] = {0, 0};
    for (int f = 0; f < nforks; ++f) fc[f] = fork_channels[f];
    return commit_weights(ctx, fc, nforks);
}

int isx_get_weight_host(isx_ctx *ctx, const char *key, void *data, int64_t nbytes)
{
    ISX_TRY(check_ready(ctx));
    ISX_REQUIRE(key && data, "null argument");
    ISX_HIP(hipSetDevice(ctx->device));
    UNetState *u = ctx->unet;
    std::string k(key);
    auto it = u->host.find(k);
    ISX_REQUIRE(it != u->host.end(), "unknown key %s", key);
    for (int li = 0; li < u->nlayers; ++li) {
        ConvBN &L = u->cv[li];
        const float *src = nullptr;
        if (k == L.bn + ".running_mean") src = L.rmean;
        else if (k == L.bn + ".running_var") src = L.rvar;
        else if (k == L.bn + ".num_batches_tracked") {
            ISX_REQUIRE(nbytes >= 8, "buffer too small");
            *(int64_t *)data = L.nbt;
            return ISX_OK;
        }
        if (src) {
            ISX_REQUIRE(nbytes >= (int64_t)L.cout * 4, "buffer too small");
            ISX_HIP(hipStreamSynchronize(ctx->stream));
            ISX_HIP(hipMemcpy(data, src, (size_t)L.cout * 4, hipMemcpyDeviceToHost));
            return ISX_OK;
        }
    }
    // parameters are not changed by inference: return the loaded values
    ISX_REQUIRE(!it->second.is_i64 && nbytes >= (int64_t)it->second.f.size() * 4, "buffer too small");
    memcpy(data, it->second.f.data(), it->second.f.size() * 4);
    return ISX_OK;
}

int isx_unet_set_precision(isx_ctx *ctx, int mode)
{
    ISX_REQUIRE(ctx && mode >= ISX_PRECISION_FP32 && mode <= ISX_PRECISION_BF16, "precision mode %d", mode);
    if (!ctx->unet) { ctx->unet = new UNetState(); ctx->unet->tune = ctx->tune; }
    ctx->unet->precision = mode;
    return ISX_OK;
}

int isx_unet_set_final_activations(isx_ctx *ctx, const int32_t *acts, int n_forks)
{
    ISX_REQUIRE(ctx && acts && n_forks >= 1 && n_forks <= ISX_MAX_FORKS, "final activations: %d decoders", n_forks);
    for (int f = 0; f < n_forks; ++f)
        ISX_REQUIRE(acts[f] >= ISX_FINAL_SIGMOID && acts[f] <= ISX_FINAL_SOFTMAX, "final activation code %d", acts[f]);
    if (!ctx->unet) { ctx->unet = new UNetState(); ctx->unet->tune = ctx->tune; }
    ISX_REQUIRE(!ctx->unet->committed || n_forks == ctx->unet->nforks, "final activations for %d decoders, the network has %d", n_forks,
                ctx->unet->nforks);
    for (int f = 0; f < ISX_MAX_FORKS; ++f) ctx->unet->final_act[f] = f < n_forks ? acts[f] : ISX_FINAL_SIGMOID;
    return ISX_OK;
}

int isx_unet_precision_report(isx_ctx *ctx, int32_t *modes, int cap, int32_t *n_fallback)
{
    ISX_TRY(check_ready(ctx));
    UNetState *u = ctx->unet;
    ISX_REQUIRE(u->last_n >= 1, "no forward has run yet");
    int nf = 0;
    for (int li = 0; li < u->nlayers; ++li) {
        if (modes && li < cap) modes[li] = u->arith[li];
        if (u->arith[li] == ISX_ARITH_FP32_RANGE_FALLBACK) ++nf;
    }
    if (n_fallback) *n_fallback = nf;
    return u->nlayers;
}

int isx_unet_set_batch(isx_ctx *ctx, int batch)
{
    ISX_REQUIRE(ctx && batch >= 1 && batch <= 64, "batch must be 1..64");
    if (!ctx->unet) { ctx->unet = new UNetState(); ctx->unet->tune = ctx->tune; }
    ctx->unet->batch = batch;
    if (batch > ctx->unet->wbatch) ctx->unet->wz = 0;      // forces a re-allocation at the next forward
    return ISX_OK;
}

int isx_unet_forward_chunks(isx_ctx *ctx, const float *volume, const int64_t shape[3], const int32_t *starts,
                            int64_t n, const int64_t chunk[3], float *out)
{
    ISX_TRY(check_ready(ctx));
    ISX_REQUIRE(volume && shape && starts && chunk && out && n >= 1, "bad argument");
    ISX_HIP(hipSetDevice(ctx->device));
    std::vector<int32_t> rows((size_t)n * 9, 0);
    for (int64_t i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) rows[9 * i + d] = starts[3 * i + d];
    return run_chunks(ctx, volume, shape, rows.data(), n, chunk, out, nullptr);
}

int isx_predict_volume(isx_ctx *ctx, const float *volume, const int64_t shape[3], const int32_t *table,
                       int64_t first, int64_t count, const int64_t chunk[3], float *feature_map)
{
    ISX_TRY(check_ready(ctx));
    ISX_REQUIRE(volume && shape && table && chunk && feature_map && first >= 0 && count >= 0, "bad argument");
    ISX_HIP(hipSetDevice(ctx->device));
    if (count == 0) return ISX_OK;
    return run_chunks(ctx, volume, shape, table + 9 * first, count, chunk, nullptr, feature_map);
}

int isx_unet_tap_host(isx_ctx *ctx, const char *name, float *out, int64_t cap_elems, int64_t shape_out[4])
{
    ISX_TRY(check_ready(ctx));
    ISX_REQUIRE(name && out && shape_out, "null argument");
    ISX_HIP(hipSetDevice(ctx->device));
    UNetState *u = ctx->unet;
    ISX_REQUIRE(u->last_n >= 1, "no forward has run yet");
    std::string nm(name);
    const float *src = nullptr, *sc = nullptr, *sh = nullptr;
    int C = 0, mode = 0, level = 0, tap_act = ISX_FINAL_SIGMOID;
    bool tap_planar = false;
    for (int li = 0; li < u->nlayers && !src; li += 2) {        // module names follow the state dict: c0..c4, c5_0..c8_0, c5_1..c8_1
        const int mi = layer_module(li);
        const std::string m = u->cv[li].conv.substr(0, u->cv[li].conv.find('.'));
        for (int j = 0; j < 2; ++j) {
            const ConvBN &L = u->cv[li + j];
            if (nm == L.conv) { src = u->raw[li + j]; C = L.cout; mode = 0; level = MOD_LEVEL[mi]; tap_planar = u->planar[li + j]; }
            else if (nm == L.bn) { src = u->raw[li + j]; sc = L.scale; sh = L.shift; C = L.cout; mode = 1; level = MOD_LEVEL[mi]; tap_planar = u->planar[li + j]; }
        }
        if (nm == m) {
            const ConvBN &L = u->cv[li + 1];
            src = u->raw[li + 1]; sc = L.scale; sh = L.shift; C = L.cout; mode = (mi == 8) ? 3 : 2; level = MOD_LEVEL[mi]; tap_planar = u->planar[li + 1];
            tap_act = u->final_act[li >= 18 ? 1 : 0];
        }
    }
    for (int k = 0; k < 4 && !src; ++k) {
        if (nm == std::string("d") + std::to_string(k)) {
            src = u->pool[k]; C = MOD_COUT[k]; level = k + 1; tap_planar = u->pool_planar[k];
            if (u->pool_presplit[k]) {        // the tensor holds split halves: the float values are pooled again from the resident raw tensor
                const Level &lf = u->lv[k], &lc = u->lv[k + 1];
                const ConvBN &Lp = u->cv[2 * k + 1];
                const int64_t total_p = (int64_t)u->last_n * lc.M * (Lp.cout / 8);
                pool_planar_kernel<<<cdiv(total_p, 256), 256, 0, ctx->stream>>>(u->raw[2 * k + 1], Lp.scale, Lp.shift, Lp.cout, lf.D, lf.H, lf.W,
                                                                             k == 3 ? 2 : 1, lc.D, lc.H, lc.W, u->pool[k], total_p, 0);
                ISX_CHECK_LAUNCH();
                u->pool_presplit[k] = false;
            }
        }
        else if (nm == std::string("up") + std::to_string(k) + ".cropped") {
            src = u->ups[k]; C = UP_C[k]; level = 3 - k; tap_planar = u->ups_planar[k];
            if (u->up_folded[k]) {            // not materialised by the folded forward: computed here from the resident coarse tensor
                const int li_in = (k == 0) ? 9 : dec_layer(0, k - 1, 1);
                const Level &ls = u->lv[MOD_LEVEL[layer_module(li_in)]], &lo = u->lv[3 - k];
                const UpConv &U = u->up[k];
                const ConvBN &Lin = u->cv[li_in];
                const int64_t total = (int64_t)u->last_n * lo.M * (U.C / 4);
                if (u->planar[li_in] && U.kz == 1) {            // (a planar coarse tensor: the planar kernel, into a planar tap)
                    const int64_t total_pl = (int64_t)u->last_n * (U.C / 8) * ls.D * ls.H * lo.W * 2;
                    upsample_src_planar_kernel<<<cdiv(total_pl, 256), 256, 0, ctx->stream>>>(u->raw[li_in], Lin.scale, Lin.shift, U.C, ls.D, ls.H, ls.W,
                                                                                          k == 3 ? 1 : 0, U.w, U.bias, lo.H, lo.W, u->ups[k], total_pl);
                    tap_planar = true;
                } else
                upsample_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(u->raw[li_in], Lin.scale, Lin.shift, U.C, ls.D, ls.H, ls.W, U.kz,
                                                                           k == 3 ? 1 : 0, U.w, U.bias, lo.D, lo.H, lo.W, u->ups[k], total, 0, u->planar[li_in] ? 1 : 0, 0);
                ISX_CHECK_LAUNCH();
            }
        }
    }
    ISX_REQUIRE(src, "unknown tap '%s'", name);
    const Level &l = u->lv[level];
    ISX_REQUIRE(cap_elems >= l.M * C, "tap buffer too small: need %ld floats", (long)(l.M * C));
    Scratch<float> tmp(ctx, (size_t)l.M * C);
    if (!tmp.ok()) return ISX_ERR_NOMEM;
    tap_kernel<<<cdiv(l.M * C, 256), 256, 0, ctx->stream>>>(src, sc, sh, C, l.M, mode, tap_act, tmp.p, tap_planar ? 1 : 0);     // chunk 0 of the set
    ISX_CHECK_LAUNCH();
    ISX_HIP(hipMemcpyAsync(out, tmp.p, (size_t)l.M * C * 4, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    shape_out[0] = C; shape_out[1] = l.D; shape_out[2] = l.H; shape_out[3] = l.W;
    return ISX_OK;
}

int isx_normalise_volume(isx_ctx *ctx, float *volume, int64_t n)
{
    ISX_REQUIRE(ctx && volume && n > 0, "bad argument");
    ISX_HIP(hipSetDevice(ctx->device));
    ProfScope ps(ctx, ISX_PROF_ELEMENTWISE);
    Scratch<uint32_t> mx(ctx, 1);
    if (!mx.ok()) return ISX_ERR_NOMEM;
    ISX_HIP(hipMemsetAsync(mx.p, 0, 4, ctx->stream));
    unsigned g = cdiv(n, 256 * 8);
    if (g > 2048) g = 2048;
    absmax_kernel<<<g, 256, 0, ctx->stream>>>(volume, n, mx.p);
    divide_kernel<<<g, 256, 0, ctx->stream>>>(volume, n, mx.p);
    ISX_CHECK_LAUNCH();
    return ISX_OK;
}

int isx_segment_volume(isx_ctx *ctx, const float *volume, const int64_t shape[3], const int64_t chunk[3],
                       const int64_t margin[3], float *feature_map, uint32_t *labels)
{
    ISX_TRY(check_ready(ctx));
    ISX_REQUIRE(volume && shape && chunk && margin && feature_map && labels, "null argument");
    ISX_HIP(hipSetDevice(ctx->device));
    int64_t n = isx_make_chunks(shape, chunk, margin, nullptr, 0);
    if (n < 0) return (int)n;
    std::vector<int32_t> table((size_t)n * 9);
    isx_make_chunks(shape, chunk, margin, table.data(), n);
    ISX_TRY(run_chunks(ctx, volume, shape, table.data(), n, chunk, nullptr, feature_map));
    int rc = isx_segment_output_image(ctx, feature_map, shape, labels, 10, 10000000, nullptr, nullptr, 0, nullptr,
                                      nullptr, nullptr);
    // output_volume[:] = 0 (segmentation.py:195)
    ISX_HIP(hipMemsetAsync(feature_map, 0, (size_t)ctx->unet->out_channels * shape[0] * shape[1] * shape[2] * 4, ctx->stream));
    return rc;
}

}  // extern "C"
