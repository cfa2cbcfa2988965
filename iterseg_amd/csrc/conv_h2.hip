// conv_h2.hip -- 3x3x3 convolution on the fp16 matrix cores with fp32-class accuracy ("split" operands).
//
// Replaces the same reference operator as conv_mfma_kernel (nn.Conv3d inside ConvModule.forward,
// src/iterseg/unet.py:63-79,91-106) for the wide layers when the context's precision is ISX_PRECISION_SPLIT_F16.
//
// Arithmetic.  Every fp32 operand x is split as  x = x_h + x_l / 2^11 (+ residual <= 2^-22 |x|):
//     x_h = half(x)                     (0 if |x| < 2^-14, so no operand is subnormal)
//     x_l = half((x - x_h) * 2^11)      (0 if that is below 2^-14: absolute error <= 3e-8)
// and a product a*b is accumulated as  a_h*b_h  (accumulator HI)  +  (a_h*b_l + a_l*b_h)  (accumulator LO, carrying
// the 2^11 scale), the a_l*b_l term (2^-22 relative) is dropped.  All three partial products are exact in fp32
// (11 x 11 significant bits) and both accumulators are fp32, so the result differs from an fp32 convolution by
// about 2^-21 per product -- below the rounding noise of the fp32 accumulation over K >= 864 terms.  The epilogue
// returns HI + LO * 2^-11 + bias.  Operands are clamped to +-65504 (post-BatchNorm activations are nowhere near).
//
// Precision modes (template parameter PM; isx_unet_set_precision): 0 = the split arithmetic above (the default, fp32-class);
// 1 = plain f16 operands (round to nearest, one MFMA per product, ~5e-4 relative per operand: the "fp16 U-Net" of BASELINE
// config 5); 2 = plain bf16 operands (v_mfma_f32_16x16x32_bf16, fp32 range, ~4e-3 per operand: the mixed-precision mode of
// BASELINE config 4's training).  The single-product modes use only the hi half of each weight stage and of the halo tile
// (half the LDS traffic and a third of the MFMAs); accumulation, bias and the BatchNorm statistics stay fp32 in every mode.
//
// Structure: a work-group owns 128 MT or 256 output voxels x BN (x NJ) output columns; the halo of its voxels for a K stage of
// 8 input channels is staged in LDS (BatchNorm + ReLU of the producer, zero padding and the operand split applied at the
// store) and reused by all 27 taps together with the stage's weight slab; v_mfma_f32_16x16x32_f16 takes k = 8 * (lane / 16)
// + channel, so the four 16-lane groups of a wave feed four taps: 28 tap slots = 7 tap quads (the 28th tap has zero weights).
// The matrix phase is software-pipelined by hand, the next stage's halo loads go out between its matrix instructions, the
// output tile leaves through LDS as whole rows, the grid is 1-D and XCD-aware (LOG.md has the measurements
// behind each of these).  Three kernels share the body:
//   conv_h2c16_kernel<BN, MT, PM>   4 waves, weight slab through registers; BN = 64 (two work-groups per CU where they fit) or
//                                   32 columns with MT = 2 row tiles per wave (the Cout = 32 layers);
//   conv_h2c16nw8_kernel<PM>        8 waves on ONE 64-column weight slab over 256 voxels, slabs by LDS-DMA into two buffers;
//   conv_h2c16nw8x2_kernel<PM>      the same with two N tiles per work-group (Cout % 128 == 0).
// The thin tail layer has its own kernel (conv_tail.hip); training in fp32 and the exact-fp32 mode use conv_mfma_kernel (unet.hip).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "unet_state.h"
#include "h2_split.h"

namespace isx {

// development: -DISX_H2_STAMP accumulates, per launch, the cycles wave 0 of every work-group spends in each part of a K stage
// (h2_stamp[]: barrier 1, wait for the prefetched loads, staging stores, barrier 2, load issue, matrix phase, prologue, epilogue,
// stages, work-groups) and the launcher prints them (it synchronises: timing experiments only).
// -DISX_H2_STAMP=2: only the work-group's span in shader-clock cycles and at 100 MHz (the clock the chip holds), no stamps in the loop
#ifdef ISX_H2_STAMP
__device__ unsigned long long h2_stamp[12];
#endif
#if defined(ISX_H2_STAMP) && ISX_H2_STAMP != 2
#define H2_STAGE_STAMPS 1
#define H2_T(var) const unsigned long long var = __builtin_readcyclecounter();
#define H2_ACC(i, d) tk[i] += (d);
#else
#define H2_T(var)
#define H2_ACC(i, d)
#endif
#ifndef ISX_H2_ABLATE
#define ISX_H2_ABLATE 0          // development (timing experiments, wrong results): 2 = no matrix phase, 3 = no halo loads, 4 = no slab DMA,
                                 // 6 = matrix instructions on the first fragments only (no fragment reads), 7 = 3 + 4 (no vector-memory
                                 // instruction in the K loop), 8 = 7 and no staging arithmetic / LDS stores (barriers + matrix phase only)
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 8 fp32 -> 8 f16 (round to nearest even, clamped) / 8 bf16 (round to nearest even), packed in a uint4
__device__ __forceinline__ uint32_t pack_f16_rn(float x, float y)
{
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    union { h2v v; uint32_t u; } H;
    H.v[0] = (_Float16)__builtin_amdgcn_fmed3f(x, -H2_MAX, H2_MAX);
    H.v[1] = (_Float16)__builtin_amdgcn_fmed3f(y, -H2_MAX, H2_MAX);
    return H.u;
}
__host__ __device__ inline uint16_t bf16_rne_bits(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ uint32_t pack_bf16_rn(float x, float y)
{
    uint32_t a = __float_as_uint(x), b = __float_as_uint(y);
    a += 0x7FFFu + ((a >> 16) & 1u);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xFFFF0000u);
}
template <int PM>
__device__ __forceinline__ void round8(const float4 &a, const float4 &b, uint4 &hi)
{
    if constexpr (PM == 1) {
        hi.x = pack_f16_rn(a.x, a.y); hi.y = pack_f16_rn(a.z, a.w); hi.z = pack_f16_rn(b.x, b.y); hi.w = pack_f16_rn(b.z, b.w);
    } else {
        hi.x = pack_bf16_rn(a.x, a.y); hi.y = pack_bf16_rn(a.z, a.w); hi.z = pack_bf16_rn(b.x, b.y); hi.w = pack_bf16_rn(b.z, b.w);
    }
}

// One 1 KiB piece of a weight slab straight into LDS (global_load_lds_dwordx4: lane i's 16 bytes land at lds_wave_base + 16 i,
// no register round trip).  Inline assembly on purpose: through the builtin the compiler waits for vmcnt(0) before EVERY later
// LDS read of the work-group's one shared array (it cannot tell the two slab buffers apart), which would put the DMA's latency
// in front of the matrix phase it is meant to hide behind.  The wave's vmcnt still counts it, in issue order, and the body
// waits for vmcnt(0) before the barrier in front of the slab's first use.
__device__ __forceinline__ void lds_dma16(const uint4 *g, uint4 *lds_wave_base)
{
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(la), "v"(g) : "m0", "memory");
}

// n / d for 0 <= n < 2^21 by one multiplication with inv = 1 / d (a 1-ulp reciprocal will do): (n + 0.5) / d lies at least 0.5 / d from
// every integer, the three roundings move it by less than (n + 0.5) / d * 1.8e-7.  The prologue's row / slot tables are ~30 integer
// divisions by run-time tile sizes otherwise, ~35 vector instructions each -- as many as the whole K loop of a four-stage layer
// (c0.conv1, c1.conv0, c7.conv1) issues, on a vector pipe that those layers keep busier than the matrix pipe.
__device__ __forceinline__ float qinv(int d) { return __builtin_amdgcn_rcpf((float)d); }
__device__ __forceinline__ int qdiv(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }

// MT = 32-row tiles per wave (1: 128 voxels per 4-wave work-group, 2: 256).  With 32 columns, MT = 2 gives the wave a 64 x 32
// tile whose fragment reads feed as many MFMAs as the 32 x 64 tile of the 64-column variant.
// NW = waves per work-group (4, or 8: a 512-thread work-group over 256 voxels whose eight waves share ONE weight slab).
// NJ (NW = 8 only): N tiles of BN columns per work-group.  NJ = 2: the halo tile of a K stage is staged ONCE for 2 x 64
// output columns -- the layers with Cout >= 128 (47 % of the network's FLOPs) otherwise load, normalise, split and store
// every halo tile once per 64 columns --, the two weight slabs of a stage alternate between the two DMA buffers (the slab
// sequence g = step * NJ + jj simply continues: slab g + 1 lands while the matrix phase reads slab g), one more barrier per
// stage for twice the matrix work per stage, 128 accumulator registers.
// UP: source A is the COARSE output of the previous decoder module and is up-sampled while it is staged (the depthwise transposed
// convolution of unet.py:216-242 with kernel = stride = (1, 2, 2): fine voxel (z, y, x) reads coarse ((y + off) >> 1, (x + off) >> 1) and
// takes the weight of its parity): relu(bn(x)) * w_up[parity][c] + b_up[c], the same fmaf the materialising kernel applies -- bit for
// bit the same operand --, so the up-sampled tensor (42.6 + 21.6 MB per chunk for up2 / up1) is never written or read.
template <int BN, int MT, int PM, int NW, int NJ = 1, bool UP = false>
__device__ __forceinline__ void conv_h2_body(const ConvArgs &a)
{
    static_assert(NJ == 1 || NW == 8, "N-tile pairs: the LDS-DMA kernel only");
    // M512 (NW = 8, MT = 2): 512 output voxels per work-group, a 64 x 64 (x NJ) wave tile.  Per matrix instruction half the weight-slab
    // bytes of the 256-voxel forms and 2/3 (NJ = 1) or half (NJ = 2) of their fragment reads: the round-5 ablations (LOG.md) put 27 % of
    // the 27-tap layers' time on the K loop's vector-memory traffic and 12 % on fragment reads, and nothing on how they are scheduled.
    // Possible since the split mode needs one accumulator per output (128 registers for 64 x 128).
    constexpr bool M512 = NW == 8 && MT == 2;
    constexpr bool SPLIT = PM == 0;
    // DMA: the 512-thread kernel takes its weight slabs by LDS-DMA into two buffers used alternately: the slab of stage s + 1 is
    // requested during the matrix phase of stage s (no prefetch registers, no LDS stores by the waves: 7 global loads + 7
    // ds_write_b128 per lane and stage gone) and is complete before the staging of s + 1 ends
    constexpr bool DMA = NW == 8;
    constexpr int THREADS = 64 * NW;
    constexpr int ROWS = 32 * NW * MT, TXW = 2 * NW * MT;
    constexpr int NA = NW == 8 ? 3 : (MT > 1 ? 5 : 4);       // halo slots per thread (1536 / 1280 / 1024 slots at most: launch_conv_h2)
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];
    constexpr int NTAPS = 27, NSLOT = 28, NQUAD = 7;          // tap slots of a weight stage: 7 tap quads, the 28th tap has zero weights
    constexpr int BU = NSLOT * BN;                        // uint4 per half of the weight slab (hi | lo)
    constexpr int SLAB = SPLIT ? 2 * BU : BU;             // uint4 of weights staged per K step (single-product modes: hi only)
    constexpr int NB = (SLAB + THREADS - 1) / THREADS;    // uint4 of weights per thread per stage (14 / 7; halved without lo)
    constexpr bool BEXACT = SLAB % THREADS == 0;
    uint4 *Bs = smem4;                                    // [hi: 28][BN] | [lo: 28][BN]  (DMA: two of them)
    uint4 *AsH = smem4 + (DMA ? 2 : 1) * SLAB;
    uint4 *AsL = AsH + a.slots_max;                       // (SPLIT only)
    int *rowinfo = (int *)(AsH + (SPLIT ? 2 : 1) * a.slots_max);
    // per-wave BatchNorm partial sums of the epilogue; M512: behind the epilogue's output tile inside the staging buffers (free by then)
    float *red = M512 ? (float *)smem4 + ROWS * (BN + 4) : (float *)(rowinfo + ROWS);
    // (UP) [4 ukz parities + bias][CA] floats, pre-scaled by the activation scale of the f16 modes
    float *upws = M512 ? (float *)(rowinfo + ROWS) : red + NW * BN * 2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int lin = blockIdx.x;
    if (a.xcd_remap) {
        const int per = gridDim.x >> 3, main = per << 3;
        if (lin < main) lin = (lin & 7) * per + (lin >> 3);
    }
    const int ntile = (lin % a.ntiles) * NJ;              // first N tile of this work-group (a.ntiles = Cout / (BN * NJ))
    lin /= a.ntiles;
    const int b = lin / a.tiles_per_chunk, tile = lin % a.tiles_per_chunk;
    const int D = a.D, H = a.H, W = a.W;
    const int64_t M = (int64_t)D * H * W;

    // ---- tile geometry: identical to conv_mfma_kernel with one row tile per wave ---------------------------
    int PS, RS, S;
    int z0, y0 = 0, x0 = 0, p0 = 0, qlo = 0, TZ, TP = 0;
    const int WP = W + 2;
    // mode 0: a box of 2 x TYb x TXb voxels (2 TYb TXb <= ROWS) on a halo of 4 x (TYb + 2) x (TXb + 2) slots: the interior boxes are
    // 8 x TXW, the edge strips (ConvArgs::sr_*, sb_*) tall and narrow / flat and wide
    // (the N-tile-pair kernel sits at the register limit -- run-time box sizes cost it 35 more spilled registers and 6 % -- and its
    // layers, Cout % 128 == 0, are below the box-tiled levels for every chunk shape iterseg uses: it keeps the padded boxes)
    constexpr bool STRIPS = NJ == 1 || M512;
    int TYb = 8, TXb = TXW, xlim = W;                            // xlim: the last bottom-strip box stops where the right strip begins
    float iA, iB, iC, iD;                                        // qdiv reciprocals -- boxes: 1 / TXb, TYb, RS, TYb + 2; linearised: 1 / TP, W, PS, WP
    if (a.mode == 0) {
        const int nzt = (D + 1) >> 1;
        const int zt = tile % nzt, rest = tile / nzt;
        z0 = zt * 2; TZ = 2;
        if (!STRIPS || rest < a.nI) { y0 = (rest / a.ntx) * 8; x0 = (rest % a.ntx) * TXW; }
        else if (rest < a.nI + a.sr_n) { TYb = a.sr_ty; TXb = a.sr_tx; y0 = (rest - a.nI) * a.sr_ty; x0 = a.sr_x0; }
        else { TYb = a.sb_ty; TXb = a.sb_tx; y0 = a.sb_y0; x0 = (rest - a.nI - a.sr_n) * a.sb_tx; if (a.sr_n) xlim = a.sr_x0; }
        if constexpr (STRIPS) { RS = TXb + 2; PS = (TYb + 2) * RS; } else { RS = TXW + 2; PS = 10 * (TXW + 2); }
        S = 4 * PS;
        iA = qinv(TXb); iB = qinv(TYb); iC = qinv(RS); iD = qinv(TYb + 2);
    } else {
        TZ = a.tz;
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
        iA = qinv(TP); iB = qinv(W); iC = qinv(PS); iD = qinv(WP);
    }
    if (tid < ROWS) {
        int r = tid, g = -1;
        if (a.mode == 0) {
            int zz, yy, xx;
            if constexpr (STRIPS) { const int t = qdiv(r, iA); xx = r - t * TXb; zz = qdiv(t, iB); yy = t - zz * TYb; }
            else { zz = r / (8 * TXW); yy = (r / TXW) & 7; xx = r % TXW; }
            int z = z0 + zz, y = y0 + yy, x = x0 + xx;
            if (zz < 2 && z < D && y < H && x < xlim) g = (z * H + y) * W + x;
        } else {
            int zz = qdiv(r, iA), p = p0 + r - zz * TP;
            int z = z0 + zz;
            if (z < D && p < H * W) g = z * H * W + p;
        }
        rowinfo[r] = g;
    }
    auto row_slot = [&](int r) __attribute__((always_inline)) -> int {
        if (a.mode == 0) {
            if constexpr (!STRIPS) {
                const int zz = r / (8 * TXW), yy = (r / TXW) & 7, xx = r % TXW;
                return (zz + 1) * (10 * (TXW + 2)) + (yy + 1) * (TXW + 2) + (xx + 1);
            }
            const int t = qdiv(r, iA);
            int xx = r - t * TXb, zz = qdiv(t, iB), yy = t - zz * TYb;
            if (zz > 1) { zz = 1; yy = TYb - 1; xx = TXb - 1; }          // rows beyond the box (strips: 2 TYb TXb < ROWS): any valid slot
            return (zz + 1) * PS + (yy + 1) * RS + (xx + 1);
        }
        int zz = qdiv(r, iA), p = p0 + r - zz * TP;
        if (p > H * W - 1) p = H * W - 1;
        const int py = qdiv(p, iB);
        int q = py * WP + (p - py * W) + 1;
        return (zz + 1) * PS + (q - qlo);
    };
    // lane (r = lane & 15, h = lane >> 4) holds row r of a 16-row tile and tap 4 q + h
    const int l16 = lane & 15, h4 = lane >> 4;
    constexpr int MT16 = 2 * MT, NT16 = BN / 16;
    int aslot16[MT16];
#pragma unroll
    for (int t = 0; t < MT16; ++t) aslot16[t] = row_slot(wave * MT * 32 + 16 * t + l16);
    // a halo slot's voxel (-1: outside the volume = zero padding, -2: beyond the tile's slots) and, for the up-sampling kernels, its
    // coarse voxel | parity << 28 (upsample_kernel: zs = z / kz, kidx = ((z % kz) * 2 + py) * 2 + px)
    auto slot_voxel = [&](int s, int &gu) __attribute__((always_inline)) -> int {
        gu = 0;
        if (s >= S) return -2;
        int z, y, x;
        if (a.mode == 0) {
            int hz, hy, hx;
            if constexpr (STRIPS) { const int t = qdiv(s, iC); hx = s - t * RS; hz = qdiv(t, iD); hy = t - hz * (TYb + 2); }
            else { hz = s / (10 * (TXW + 2)); hy = (s / (TXW + 2)) % 10; hx = s % (TXW + 2); }
            z = z0 + hz - 1; y = y0 + hy - 1; x = x0 + hx - 1;
        } else {
            const int hz = qdiv(s, iC), qq = qlo + (s - hz * PS) + WP, yq = qdiv(qq, iD);
            z = z0 + hz - 1;
            y = yq - 1;
            x = qq - yq * WP - 1;
        }
        if (!(z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W)) return -1;
        if constexpr (UP) {
            const int yo = y + a.uoff, xo = x + a.uoff;
            const int zc = a.ukz == 2 ? z >> 1 : z, pz = a.ukz == 2 ? z & 1 : 0;
            gu = ((zc * a.uH + (yo >> 1)) * a.uW + (xo >> 1)) | (((pz << 2) | ((yo & 1) << 1) | (xo & 1)) << 28);
        }
        return (z * H + y) * W + x;
    };
    int gu0, gu1, gu2, gu3 = 0, gu4 = 0;
    const int gv0 = slot_voxel(tid, gu0), gv1 = slot_voxel(tid + THREADS, gu1), gv2 = slot_voxel(tid + 2 * THREADS, gu2),
              gv3 = NA > 3 ? slot_voxel(tid + 3 * THREADS, gu3) : -2, gv4 = NA > 4 ? slot_voxel(tid + 4 * THREADS, gu4) : -2;
    bool ld_up = false;                                       // the stage in the prefetch registers is an up-sampled one
    bool ld_pre = false;                                      // ... comes from a pre-split source
    int ld_cc = 0;

    const int nsteps = (a.CA + a.CB) >> 3;
    typedef float f32x4c __attribute__((ext_vector_type(4)));
    // one accumulator per output in every mode: the split mode adds its three partial products into the same register (the
    // operands are scaled so that the residual halves are normal numbers: h2_split.h)
    f32x4c c_hiN[NJ][MT16][NT16];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
#pragma unroll
        for (int m = 0; m < MT16; ++m)
#pragma unroll
            for (int j = 0; j < NT16; ++j) c_hiN[jj][m][j] = f32x4c{0.f, 0.f, 0.f, 0.f};
    // operand scales of the f16 modes (powers of two; bf16 has the fp32 range and takes none)
    constexpr float ASC = PM == 2 ? 1.f : H2_SA, OSC = PM == 2 ? 1.f : 1.f / (H2_SA * H2_SW);

    // named prefetch registers (arrays that live across the K loop end up in scratch)
    float4 ra0, rc0, ra1, rc1, ra2, rc2, ra3, rc3, ra4, rc4, vs0, vs1, vh0, vh1;
    if (ISX_H2_ABLATE == 3 || ISX_H2_ABLATE >= 7) { ra0 = rc0 = ra1 = rc1 = ra2 = rc2 = ra3 = rc3 = ra4 = rc4 = make_float4(0.5f, 0.25f, 0.125f, 1.f); }   // (no halo loads)
    uint4 rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7, rb8, rb9, rb10, rb11, rb12, rb13;
    float relu_lo = 0.f;                                      // ReLU floor of the stage: 0 behind a BatchNorm, -inf for a source without one
    // the stage being loaded: source pointer (chunk b, channel offset included), its row length, the weight slab
    const float *ld_src = nullptr;
    int ld_C = 0;
    const uint4 *ld_wsrc = nullptr;
    // a wave whose 64 slots of pass k all lie beyond the tile's S slots skips the load: the tile has 400-720 slots, the plan
    // covers 1024-1536, and a dead load (every lane clamped to voxel 0) costs the load path as much as a live one -- they
    // were 40-56 % of the halo load instructions
    const int wbase = __builtin_amdgcn_readfirstlane(tid) & ~63;
#define H2_LOAD_A(k)                                                                              \
    if constexpr (k < NA && ISX_H2_ABLATE != 3 && ISX_H2_ABLATE < 7) if (THREADS * k + wbase < S) {                    \
        /* (a 32-bit element offset here -- one multiply, scalar-base load -- costs the N-tile-pair kernel 37 more spilled registers) */ \
        const int gl_ = (UP && ld_up) ? (gu##k & 0x0FFFFFFF) : (gv##k > 0 ? gv##k : 0);          \
        const float *pp = &ld_src[(int64_t)gl_ * ld_C];                                           \
        ra##k = *(const float4 *)pp;                                                              \
        rc##k = *(const float4 *)(pp + 4);                                                        \
    }
#define H2_LOAD_B(k) if constexpr (NB > k) { if (BEXACT || tid + THREADS * k < SLAB) rb##k = ld_wsrc[tid + THREADS * k]; }
// a source without BatchNorm gets scale 1, shift 0 and a ReLU floor of -inf (x * 1 + 0 and max(x, -inf) are exact): the staging
// below has no branches
#define H2_LOAD_HEAD(step_)                                                                       \
    {                                                                                             \
        const int c0 = (step_) << 3;                                                              \
        const float *src, *sc, *sh;                                                               \
        int C, cc;                                                                                \
        if (c0 < a.CA) { src = a.srcA; sc = a.scaleA; sh = a.shiftA; C = a.CA; cc = c0; }         \
        else { src = a.srcB; sc = a.scaleB; sh = a.shiftB; C = a.CB; cc = c0 - a.CA; }            \
        const bool has_affine = sc != nullptr;                                                    \
        const float *scp = has_affine ? sc : a.bias, *shp = has_affine ? sh : a.bias;             \
        const int64_t so = has_affine ? (int64_t)b * C + cc : 0;                                  \
        vs0 = *(const float4 *)&scp[so]; vs1 = *(const float4 *)&scp[so + 4];                     \
        vh0 = *(const float4 *)&shp[so]; vh1 = *(const float4 *)&shp[so + 4];                     \
        if (!has_affine) {                                                                        \
            vs0 = make_float4(1.f, 1.f, 1.f, 1.f); vs1 = vs0;                                     \
            vh0 = make_float4(0.f, 0.f, 0.f, 0.f); vh1 = vh0;                                     \
        }                                                                                         \
        ld_up = UP && a.upA && c0 < a.CA;                                                         \
        ld_pre = SPLIT && a.presplitA && c0 < a.CA;                                               \
        ld_cc = cc;                                                                               \
        if (ASC != 1.f && !ld_up) {                  /* (an up stage is scaled through its up weights) */ \
            vs0.x *= ASC; vs0.y *= ASC; vs0.z *= ASC; vs0.w *= ASC; vs1.x *= ASC; vs1.y *= ASC; vs1.z *= ASC; vs1.w *= ASC; \
            vh0.x *= ASC; vh0.y *= ASC; vh0.z *= ASC; vh0.w *= ASC; vh1.x *= ASC; vh1.y *= ASC; vh1.z *= ASC; vh1.w *= ASC; \
        }                                                                                         \
        relu_lo = has_affine ? 0.f : -__builtin_inff();                                           \
        const int64_t Ms = ld_up ? (int64_t)(D / a.ukz) * a.uH * a.uW : M;                                  \
        ld_src = src + (int64_t)b * Ms * C + cc;                                                  \
        ld_C = C;                                                                                 \
        if (c0 < a.CA ? a.planarA : a.planarB) {               /* [chunk][C/8][voxel][8]: this stage's plane */ \
            ld_src = src + ((int64_t)b * (C >> 3) + (cc >> 3)) * Ms * 8;                          \
            ld_C = 8;                                                                             \
        }                                                                                         \
    }
// the stage's halo loads one at a time, between the matrix instructions of step s_ of the matrix phase
#define H2_LOAD_SLOT(s_)                                                                          \
    switch (s_) {                                                                                 \
        case 0: H2_LOAD_A(0) break; case 1: H2_LOAD_A(1) break; case 2: H2_LOAD_A(2) break;        \
        case 3: H2_LOAD_A(3) break; case 4: H2_LOAD_A(4) break; default: break;                   \
    }
// piece k_ of slab g_ of the work-group's slab sequence (g = step * NJ + jj: N tile ntile + g % NJ, K stage g / NJ) -> buffer g & 1
#define H2_DMA_B1(g_, k_)                                                                         \
    if (ISX_H2_ABLATE != 4 && ISX_H2_ABLATE < 7 && (k_) < NB && (BEXACT || THREADS * (k_) + wbase < SLAB))                                   \
        lds_dma16((const uint4 *)a.w + ((int64_t)(ntile + (g_) % NJ) * nsteps + (g_) / NJ) * (2 * BU) + tid + THREADS * (k_), \
                  Bs + ((g_) & 1) * SLAB + wbase + THREADS * (k_));
#define H2_DMA_B(g_) { _Pragma("unroll") for (int k_ = 0; k_ < NB; ++k_) H2_DMA_B1(g_, k_) }
// (register path) the stage's weight slab, loaded at the top of its own iteration: its registers are then free during the matrix
// phase, which keeps its fragments two steps ahead in them, and it lands behind the barrier and the halo arithmetic
#define H2_LOAD_B_ALL(step_)                                                                      \
    {                                                                                             \
        ld_wsrc = (const uint4 *)a.w + ((int64_t)ntile * nsteps + (step_)) * (2 * BU);  /* packs keep the hi|lo stride in every mode */ \
        H2_LOAD_B(0) H2_LOAD_B(1) H2_LOAD_B(2) H2_LOAD_B(3) H2_LOAD_B(4) H2_LOAD_B(5) H2_LOAD_B(6) \
        H2_LOAD_B(7) H2_LOAD_B(8) H2_LOAD_B(9) H2_LOAD_B(10) H2_LOAD_B(11) H2_LOAD_B(12) H2_LOAD_B(13) \
    }
// BatchNorm + ReLU, the operand split (or rounding), zero padding by select, one predicated store: no branches (the branchy
// form compiled to 19 exec-mask regions and 128 register moves per stage of the 64-column kernel)
#ifndef ISX_H2_LEAN
#define ISX_H2_LEAN 1            // split mode, 512-thread kernels: ReLU folded into the split's clamp, residual by v_fma_mix_f32 (2 of ~6.5 vector
                                 // instructions per staged element less; bit-identical; round 5: -2 ... -7 % on the 512-voxel kernels' layers)
#endif
#define H2_AFF(v, s_, h_)                                                                         \
    v.x = fmaf(v.x, s_.x, h_.x); v.y = fmaf(v.y, s_.y, h_.y); v.z = fmaf(v.z, s_.z, h_.z); v.w = fmaf(v.w, s_.w, h_.w);
#define H2_RELU(v) v.x = fmaxf(v.x, relu_lo); v.y = fmaxf(v.y, relu_lo); v.z = fmaxf(v.z, relu_lo); v.w = fmaxf(v.w, relu_lo);
#define H2_AFFL(v, s_, h_) H2_AFF(v, s_, h_) H2_RELU(v)
#define H2_STORE_A(k)                                                                             \
    if constexpr (k < NA && ISX_H2_ABLATE < 8) {                                                                       \
        float4 v = ra##k, w = rc##k;                                                              \
        if (SPLIT && ld_pre) {          /* (a pre-split source: the two 16-byte loads ARE the hi and the residual halves) */ \
            if (gv##k >= 0) {                                                                     \
                AsH[tid + THREADS * k] = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)); \
                AsL[tid + THREADS * k] = make_uint4(__float_as_uint(w.x), __float_as_uint(w.y), __float_as_uint(w.z), __float_as_uint(w.w)); \
            }                                                                                     \
        } else {                                                                                  \
        constexpr bool LEAN = SPLIT && ISX_H2_LEAN && NW == 8;   /* (the four-wave kernels measure 2 % slower with it) */ \
        float lo_clamp = -H2_MAX;                                                                 \
        if (LEAN && !(UP && ld_up)) { H2_AFF(v, vs0, vh0) H2_AFF(w, vs1, vh1) lo_clamp = relu_lo == 0.f ? 0.f : -H2_MAX; } \
        else { H2_AFFL(v, vs0, vh0) H2_AFFL(w, vs1, vh1) }                                        \
        if (UP && ld_up) {                                                                        \
            const float *uw = &upws[((unsigned)gu##k >> 28) * a.CA + ld_cc], *ub = &upws[4 * a.ukz * a.CA + ld_cc]; \
            const float4 u0 = *(const float4 *)uw, u1 = *(const float4 *)(uw + 4);                \
            const float4 b0 = *(const float4 *)ub, b1 = *(const float4 *)(ub + 4);                \
            v.x = fmaf(v.x, u0.x, b0.x); v.y = fmaf(v.y, u0.y, b0.y); v.z = fmaf(v.z, u0.z, b0.z); v.w = fmaf(v.w, u0.w, b0.w); \
            w.x = fmaf(w.x, u1.x, b1.x); w.y = fmaf(w.y, u1.y, b1.y); w.z = fmaf(w.z, u1.z, b1.z); w.w = fmaf(w.w, u1.w, b1.w); \
        }                                                                                         \
        uint4 hi, lo = make_uint4(0u, 0u, 0u, 0u);                                                \
        if constexpr (LEAN) split8r_lean(v, w, lo_clamp, hi, lo);                                 \
        else if constexpr (SPLIT) split8r(v, w, hi, lo); else round8<PM>(v, w, hi);               \
        /* (a slot outside the volume -- the convolution's zero padding -- is the same slot in every K stage: zeroed once in the  */ \
        /*  prologue and never written again, instead of eight selects per slot and stage)                                       */ \
        if (gv##k >= 0) {                                                                         \
            AsH[tid + THREADS * k] = hi;                                                          \
            if constexpr (SPLIT) AsL[tid + THREADS * k] = lo;                                     \
        }                                                                                         \
        }                                                                                         \
    }
#define H2_ZERO_A(k)                                                                              \
    if constexpr (k < NA) if (gv##k == -1) {                                                      \
        AsH[tid + THREADS * k] = make_uint4(0u, 0u, 0u, 0u);                                      \
        if constexpr (SPLIT) AsL[tid + THREADS * k] = make_uint4(0u, 0u, 0u, 0u);                 \
    }
#define H2_STORE_B(k) if constexpr (NB > k) { if (BEXACT || tid + THREADS * k < SLAB) Bs[tid + THREADS * k] = rb##k; }

    // LDS offset of tap t relative to the output voxel's slot (tap 27 = padding of tap 26: zero weights)
    auto tap_off = [&](int t) __attribute__((always_inline)) -> int {
        if (t > NTAPS - 1) t = NTAPS - 1;
        return (t / 9 - 1) * PS + ((t / 3) % 3 - 1) * RS + (t % 3 - 1);
    };
    // the lane's tap offset of every tap quad: the same in every K stage, selected once -- selected inside the matrix phase, the
    // four-way choice by lane group compiled to divergent branches, one basic block per quad, and nothing (fragment reads least
    // of all) was scheduled across them
    int toq[NQUAD];
#pragma unroll
    for (int q = 0; q < NQUAD; ++q) {
        const int t0 = tap_off(4 * q), t1 = tap_off(4 * q + 1), t2 = tap_off(4 * q + 2), t3 = tap_off(4 * q + 3);
        toq[q] = (h4 & 2) ? ((h4 & 1) ? t3 : t2) : ((h4 & 1) ? t1 : t0);
    }
#ifdef ISX_H2_STAMP
    unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tk_begin = __builtin_readcyclecounter();
    const unsigned long long tr_begin = __builtin_amdgcn_s_memrealtime();
#endif
    H2_ZERO_A(0) H2_ZERO_A(1) H2_ZERO_A(2) H2_ZERO_A(3) H2_ZERO_A(4)
    if constexpr (UP) if (a.upA)
        for (int i = tid; i < (4 * a.ukz + 1) * a.CA; i += THREADS) upws[i] = (i < 4 * a.ukz * a.CA ? a.upw[i] : a.upb[i - 4 * a.ukz * a.CA]) * ASC;
    if constexpr (DMA) H2_DMA_B(0)
    H2_LOAD_HEAD(0) H2_LOAD_A(0) H2_LOAD_A(1) H2_LOAD_A(2) H2_LOAD_A(3) H2_LOAD_A(4)
    H2_T(tk_loop)
    for (int step = 0; step < nsteps; ++step) {
        H2_T(t0)
        if constexpr (!DMA) H2_LOAD_B_ALL(step)
        __syncthreads();
        H2_T(t1)
#ifdef H2_STAGE_STAMPS
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        H2_T(t1b)
        H2_STORE_A(0) H2_STORE_A(1) H2_STORE_A(2) H2_STORE_A(3) H2_STORE_A(4)
        if constexpr (!DMA) {
        H2_STORE_B(0) H2_STORE_B(1) H2_STORE_B(2) H2_STORE_B(3) H2_STORE_B(4) H2_STORE_B(5) H2_STORE_B(6)
        H2_STORE_B(7) H2_STORE_B(8) H2_STORE_B(9) H2_STORE_B(10) H2_STORE_B(11) H2_STORE_B(12) H2_STORE_B(13)
        } else __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of the stage's slab have landed (nothing else is in flight here)
        H2_T(t2)
        __syncthreads();
        H2_T(t3)
        const bool more = step + 1 < nsteps;
        if (more) H2_LOAD_HEAD(step + 1)                                   // its loads go out between the matrix instructions below
        __builtin_amdgcn_sched_barrier(0);
        H2_T(t4)
        if (ISX_H2_ABLATE == 2) { c_hiN[0][0][0][0] += (float)step; } else {
            // Software-pipelined by hand, one step = one 16-column tile j of one tap quad q (3 MT16 matrix instructions): the
            // B fragments of step s + 2 and the A fragments of the next quad are read from LDS before the matrix instructions
            // of step s are issued, and one of the next stage's global loads goes out per step.  The compiler's own schedule
            // read every fragment right before its use (its register budget is spent on the prefetched stage) and issued the
            // 24 global loads back to back before the first matrix instruction: 2 100-2 800 cycles per stage waiting for room
            // in the load queue with the matrix pipe idle (ISX_H2_STAMP build, LOG.md round 3).
            union FR { uint4 u; f16x8 v; bf16x8 w; };
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
            const int g = step * NJ + jj;                           // position in the slab sequence (= step unless NJ > 1)
            const bool more_slab = g + 1 < nsteps * NJ;
            if (jj > 0) {              // the slab of this N tile has landed (every wave's pieces), and every wave is done with the previous one
                __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            auto &c_hi = c_hiN[jj];
            const uint4 *bqh = Bs + (DMA ? (g & 1) * SLAB : 0) + h4 * BN + l16, *bql = bqh + BU;
            constexpr int NS = NQUAD * NT16;
            // two row tiles per wave (MT16 = 4: 12 matrix instructions = 192 cycles per step): B one step ahead and the lo halves
            // of A read at the start of their own quad (first used 8 matrix instructions later) -- the registers do not go further
            // (the N-tile pair holds 128 accumulator registers: the shallower pipeline of the two-row-tile kernel as well)
            constexpr int BR = (MT16 < 4 && NJ == 1) ? 3 : 2, BD = BR - 1;
            constexpr bool AL2 = MT16 < 4 && NJ == 1;
            FR ah[2][MT16], al[(SPLIT && AL2) ? 2 : 1][SPLIT ? MT16 : 1], bh[BR], bl[SPLIT ? BR : 1];
#define H2_FRAG_AH(q_)                                                                            \
            if (ISX_H2_ABLATE != 6 || (q_) < 2) { _Pragma("unroll") for (int m = 0; m < MT16; ++m) ah[(q_) & 1][m].u = AsH[aslot16[m] + toq[q_]]; }
#define H2_FRAG_AL(q_)                                                                            \
            if constexpr (SPLIT) if (ISX_H2_ABLATE != 6 || (q_) < 2) { _Pragma("unroll") for (int m = 0; m < MT16; ++m) al[AL2 ? ((q_) & 1) : 0][m].u = AsL[aslot16[m] + toq[q_]]; }
#define H2_FRAG_B(s_)                                                                             \
            if (ISX_H2_ABLATE != 6 || (s_) < BR) {                                                \
                bh[(s_) % BR].u = bqh[(4 * ((s_) / NT16)) * BN + 16 * ((s_) % NT16)];             \
                if constexpr (SPLIT) bl[(s_) % BR].u = bql[(4 * ((s_) / NT16)) * BN + 16 * ((s_) % NT16)]; \
            }
            // (the fragment addresses aslot16[m] + toq[q] are loop invariants: hoisted out of the K loop they cost 14-28 registers
            // and spill; an empty asm that "changes" toq keeps each one add next to its read)
#pragma unroll
            for (int q = 0; q < NQUAD; ++q) __asm__ volatile("" : "+v"(toq[q]));
            H2_FRAG_AH(0)
            H2_FRAG_B(0)
            if constexpr (BD > 1) H2_FRAG_B(1)
            if constexpr (AL2) H2_FRAG_AL(0)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int q = s / NT16, j = s % NT16, qs = q & 1, r = s % BR, ql = AL2 ? qs : 0;
                if (s + BD < NS) H2_FRAG_B(s + BD)
                if (!AL2 && j == 0) H2_FRAG_AL(q)
                if (j == (NT16 >= 2 ? NT16 - 2 : 0) && q + 1 < NQUAD) { H2_FRAG_AH(q + 1) if constexpr (AL2) H2_FRAG_AL(q + 1) }
                if (more && jj == 0) { H2_LOAD_SLOT(s) }
                // (DMA) the next slab of the sequence, one 1 KiB piece every third step, into the buffer the previous matrix phase read
                if constexpr (DMA) { if (more_slab && s >= 3 && (s - 3) % 3 == 0) H2_DMA_B1(g + 1, (s - 3) / 3) }
                if constexpr (SPLIT) {
#pragma unroll
                    for (int m = 0; m < MT16; ++m)
                        c_hi[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[qs][m].v, bh[r].v, c_hi[m][j], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MT16; ++m)
                        c_hi[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[qs][m].v, bl[r].v, c_hi[m][j], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MT16; ++m)
                        c_hi[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ql][m].v, bh[r].v, c_hi[m][j], 0, 0, 0);
                } else if constexpr (PM == 1) {
#pragma unroll
                    for (int m = 0; m < MT16; ++m)
                        c_hi[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[qs][m].v, bh[r].v, c_hi[m][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int m = 0; m < MT16; ++m)
                        c_hi[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[qs][m].w, bh[r].w, c_hi[m][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            static_assert(NS >= 5, "one halo load per step");
            static_assert(!DMA || 3 + 3 * (NB - 1) < NS, "one slab piece every third step");
            }   // jj
#undef H2_FRAG_AH
#undef H2_FRAG_AL
#undef H2_FRAG_B
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef H2_STAGE_STAMPS
        { H2_T(t5) H2_ACC(0, t1 - t0) H2_ACC(1, t1b - t1) H2_ACC(2, t2 - t1b) H2_ACC(3, t3 - t2) H2_ACC(4, t4 - t3) H2_ACC(5, t5 - t4) }
#endif
    }
    H2_T(tk_loop_end)
    H2_ACC(6, tk_loop - tk_begin)
#undef H2_LOAD_A
#undef H2_LOAD_B
#undef H2_LOAD_HEAD
#undef H2_LOAD_SLOT
#undef H2_LOAD_B_ALL
#undef H2_DMA_B
#undef H2_DMA_B1
#undef H2_AFFL
#undef H2_AFF
#undef H2_RELU
#undef H2_STORE_A
#undef H2_ZERO_A
#undef H2_STORE_B

    // element (voxel g, channel ch) of the output tensor, row-major or planar (ConvArgs::out_planar; store_cols = Cout then)
    auto out_at = [&](int g, int ch) __attribute__((always_inline)) -> float * {
        if (a.out_planar) return &a.out[(((int64_t)b * (a.store_cols >> 3) + (ch >> 3)) * M + g) * 8 + (ch & 7)];
        return &a.out[((int64_t)b * M + g) * a.store_cols + ch];
    };
    // ---- epilogue: HI + LO / 2^11 + bias, store, per-tile BN partial sums (valid rows only) ------------------
    // The tile goes through LDS (free after the last stage) and leaves as whole rows -- BN consecutive channels of a voxel, one
    // global_store_dwordx4 per lane, four rows per wave instruction -- instead of one dword per lane in 64-byte pieces with the
    // address arithmetic of 32 stores per lane (15-24 k cycles per work-group in the stamped build: as much as two K stages).
    // C/D of 16x16: column = lane & 15, row = 4 * (lane >> 4) + register.
    constexpr int TRS = BN + 4;                                    // floats per row of the LDS tile (16-byte rows, conflict-free both ways)
    // (the launcher pads slots_max so that the staging buffers -- slabs + halo tile -- hold the ROWS x TRS tile)
    float *T = (float *)smem4;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        const int ntile_j = ntile + jj;
        __syncthreads();                                           // every wave has read its last fragments (jj > 0: and stored tile jj - 1)
#pragma unroll
        for (int j = 0; j < NT16; ++j) {
            const int ch = ntile_j * BN + 16 * j + l16;
            const float bsv = a.bias[ch];
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int m = 0; m < MT16; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = wave * MT * 32 + 16 * m + 4 * h4 + i;
                    const float v = OSC == 1.f ? c_hiN[jj][m][j][i] + bsv : fmaf(c_hiN[jj][m][j][i], OSC, bsv);
                    T[row * TRS + 16 * j + l16] = v;
                    if (rowinfo[row] >= 0) { s += v; q += v * v; }
                }
            s += __shfl_xor(s, 16); q += __shfl_xor(q, 16);
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
            if (h4 == 0) { red[(wave * BN + 16 * j + l16) * 2] = s; red[(wave * BN + 16 * j + l16) * 2 + 1] = q; }
        }
        __syncthreads();
        constexpr int C4 = BN / 4;
#pragma unroll
        for (int e0 = 0; e0 < ROWS * C4; e0 += THREADS) {
            const int e = e0 + tid;
            // row-major: 16 lanes per row (its BN channels); planar: two lanes per (row, plane), rows consecutive
            const int row = a.out_planar ? (e >> 1) % ROWS : e / C4;
            const int c4 = a.out_planar ? 2 * (e / (2 * ROWS)) + (e & 1) : e % C4;
            const int g = rowinfo[row];
            const int ch = ntile_j * BN + 4 * c4;
            if (g >= 0) {
                const f32x4c o = *(const f32x4c *)&T[row * TRS + 4 * c4];
                f32x4c *op = (f32x4c *)out_at(g, ch);
                if (a.nt_out) __builtin_nontemporal_store(o, op); else *op = o;
            }
        }
        if (tid < BN && !a.no_stats) {
            float s = 0.f, q = 0.f;
            for (int wv = 0; wv < NW; ++wv) { s += red[(wv * BN + tid) * 2]; q += red[(wv * BN + tid) * 2 + 1]; }
            // [chunk][channel][tile] (tile fastest): bn_finalize_kernel's lanes walk a channel's tiles -- with the channels fastest
            // every lane of it touched its own 128-byte line (57 us for c0.conv1's 2 560 tiles x 64 chunks)
            int64_t o = ((int64_t)b * a.Cout + ntile_j * BN + tid) * a.tiles_per_chunk + tile;
            a.psum[o] = s;
            a.psq[o] = q;
        }
    }
#if defined(ISX_H2_STAMP) && ISX_H2_STAMP == 2
    if (tid == 0 && (blockIdx.x & 63) == 0) {          // one work-group in 64: two atomics that contend with nothing
        atomicAdd(&h2_stamp[10], (unsigned long long)(__builtin_readcyclecounter() - tk_begin));
        atomicAdd(&h2_stamp[11], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - tr_begin));
    }
#elif defined(ISX_H2_STAMP)
    if (tid == 0) {
#ifdef H2_STAGE_STAMPS
        tk[7] = __builtin_readcyclecounter() - tk_loop_end;
#endif
        for (int i = 0; i < 8; ++i) atomicAdd(&h2_stamp[i], tk[i]);
        atomicAdd(&h2_stamp[8], (unsigned long long)nsteps);
        atomicAdd(&h2_stamp[9], 1ull);
        atomicAdd(&h2_stamp[10], (unsigned long long)(__builtin_readcyclecounter() - tk_begin));        // shader-clock cycles of the work-group
        atomicAdd(&h2_stamp[11], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - tr_begin));     // the same span at 100 MHz
    }
#endif
}

// four waves, weight slab through registers.  (The accumulators of the 64-column variant come to 258 registers unless the
// compiler is told that two waves per SIMD are wanted.)
template <int BN, int MT, int PM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_h2c16_kernel(const ConvArgs a)
{
    conv_h2_body<BN, MT, PM, 4>(a);
}

// 512-thread work-groups (eight waves share one weight slab over 256 voxels), slabs by LDS-DMA
template <int PM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void conv_h2c16nw8_kernel(const ConvArgs a)
{
    conv_h2_body<64, 1, PM, 8>(a);
}

// 512 voxels per work-group: a 64 x 64 wave tile (conv_h2c16nw8m_kernel), or 64 x 128 with two N tiles (conv_h2c16nw8mx2_kernel)
template <int PM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void conv_h2c16nw8m_kernel(const ConvArgs a)
{
    conv_h2_body<64, 2, PM, 8>(a);
}
template <int PM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void conv_h2c16nw8mx2_kernel(const ConvArgs a)
{
    conv_h2_body<64, 2, PM, 8, 2>(a);
}

// ... the two forms that can up-sample their first source while staging it (UP)
template <int PM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void conv_h2c16nw8mu_kernel(const ConvArgs a)
{
    conv_h2_body<64, 2, PM, 8, 1, true>(a);
}
template <int PM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_h2c16u_kernel(const ConvArgs a)
{
    conv_h2_body<32, 2, PM, 4, 1, true>(a);
}
template <int PM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void conv_h2c16nw8x2u_kernel(const ConvArgs a)
{
    conv_h2_body<64, 1, PM, 8, 2, true>(a);
}

// ... with two N tiles per work-group (Cout % 128 == 0: the halo tile of a K stage staged once for 128 output columns)
template <int PM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void conv_h2c16nw8x2_kernel(const ConvArgs a)
{
    conv_h2_body<64, 1, PM, 8, 2>(a);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// two row tiles per wave for the 32-column tiles: only for launches that still fill the chip with 256-voxel tiles
int h2_mt(const isx_tune &tune, const Level &l, int batch)
{
    const int on = tune.h2_mt2;                            // 0: never, 1 (default): big launches, 2: always (tests)
    return on == 2 || (on && (int64_t)batch * l.M >= (1 << 21));
}

// 512-thread work-groups (NW = 8: eight waves on one 64-column weight slab, 256 voxels): every layer with Cout % 64 == 0 whose
// 256-voxel halo fits the staging plan and whose launch still fills the chip
bool conv_h2_nw8(const isx_tune &tune, const Level &l, int cout, int batch)
{
    if (!tune.h2_nw8 || cout % 64) return false;
    int ty, tx, tp, sl;
    const int tiles2 = conv_tiles(l, conv_mode(l), 2, &ty, &tx, &tp, &sl);
    return sl <= 1536 && (tune.h2_nw8 == 2 || (int64_t)batch * tiles2 * (cout / 64) >= 512);        // 2: always (tests)
}

// 64-column tiles (4-wave kernel) whenever two work-groups still fit into the 160 KB of LDS (weight slab 57 KB + 32 B per halo slot)
int conv_h2_bn(const isx_tune &tune, int slots, int cout)
{
    const size_t lds64 = (size_t)(2 * 28 * 64 + 2 * slots) * 16 + (128 + 4 * 64 * 2) * sizeof(float);
    const int force64 = tune.h2_force_bn64;
    return (cout % 64 == 0 && (force64 || 2 * lds64 <= 160 * 1024)) ? 64 : 32;
}

// w: reference layout [Cout][Cin][27] fp32 -> [Cout/BN][Cin/8][hi|lo][28 taps][BN][8 channels] halves
void pack_h2_weights(const float *w, int cout, int cin, int bn, std::vector<uint16_t> &out, int pm)
{
    const int ntiles = cout / bn, nsteps = cin / 8;
    const size_t half_slab = (size_t)28 * bn * 8;
    out.assign((size_t)ntiles * nsteps * 2 * half_slab + 4096, 0);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < 27; ++t) {
                _Float16 h, l;
                split_h2r(w[((size_t)co * cin + ci) * 27 + t] * H2_SW, h, l);       // scaled, residual as it is (h2_split.h)
                const size_t base = ((size_t)(co / bn) * nsteps + ci / 8) * 2 * half_slab;
                const size_t e = ((size_t)t * bn + co % bn) * 8 + ci % 8;
                uint16_t hb, lb;
                memcpy(&hb, &h, 2);
                memcpy(&lb, &l, 2);
                if (pm == 2) { hb = bf16_rne_bits(w[((size_t)co * cin + ci) * 27 + t]); lb = 0; }
                out[base + e] = hb;
                out[base + half_slab + e] = lb;
            }
}

// folded tail layer (cout <= 5): [Cin/8][hi|lo][12 slots for the 9 (dz,dy) taps][32 columns = dx*cout + o][8 channels] halves
void pack_h2_fold_weights(const float *w, int cout, int cin, std::vector<uint16_t> &out, int pm)
{
    const int nsteps = cin / 8;
    const size_t half_slab = (size_t)12 * 32 * 8;       // 9 taps in 12 slots: five tap pairs of the 32x32x16 core or three tap quads of the 16x16x32 one
    out.assign((size_t)nsteps * 2 * half_slab + 4096, 0);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < 27; ++t) {
                _Float16 h, l;
                split_h2(w[((size_t)co * cin + ci) * 27 + t], h, l);
                const int t9 = t / 3, dxi = t % 3;
                const size_t base = (size_t)(ci / 8) * 2 * half_slab;
                const size_t e = ((size_t)t9 * 32 + dxi * cout + co) * 8 + ci % 8;
                uint16_t hb, lb;
                memcpy(&hb, &h, 2);
                memcpy(&lb, &l, 2);
                if (pm == 2) { hb = bf16_rne_bits(w[((size_t)co * cin + ci) * 27 + t]); lb = 0; }
                out[base + e] = hb;
                out[base + half_slab + e] = lb;
            }
}

// Device version of pack_h2_weights for training (the weights change every step): wref = [Cout][Cin][27] fp32.
// transpose != 0: the dgrad operator (Cin <-> Cout, taps mirrored), packed as a convolution with `cin_p` = Cout input
// channels (zero-padded to a multiple of 8) and `cout_p` = Cin output channels.  bf16 in the hi half, lo left zero.
__global__ void repack_bf16_kernel(const float *__restrict__ wref, int cout, int cin, int transpose, int bn, int cout_p, int cin_p,
                                   uint16_t *__restrict__ dst)
{
    const int nsteps = cin_p / 8;
    const size_t half_slab = (size_t)28 * bn * 8;
    const int64_t n = (int64_t)cout_p * cin_p * 27;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % 27), ci = (int)((i / 27) % cin_p), co = (int)(i / (27 * (int64_t)cin_p));
        float v = 0.f;
        if (!transpose) { if (co < cout && ci < cin) v = wref[((size_t)co * cin + ci) * 27 + t]; }
        else if (ci < cout && co < cin) v = wref[((size_t)ci * cin + co) * 27 + (26 - t)];
        const size_t base = ((size_t)(co / bn) * nsteps + ci / 8) * 2 * half_slab;
        dst[base + ((size_t)t * bn + co % bn) * 8 + ci % 8] = bf16_rne_bits(v);
    }
}

int repack_bf16(isx_ctx *ctx, const float *wref, int cout, int cin, int transpose, int bn, int cout_p, int cin_p, void *dst)
{
    const int64_t n = (int64_t)cout_p * cin_p * 27;
    unsigned grid = (unsigned)((n + 255) / 256);
    if (grid > 4096) grid = 4096;
    repack_bf16_kernel<<<grid, 256, 0, ctx->stream>>>(wref, cout, cin, transpose, bn, cout_p, cin_p, (uint16_t *)dst);
    ISX_CHECK_LAUNCH();
    return ISX_OK;
}

size_t h2_pack_halves(int cout_p, int cin_p, int bn) { return (size_t)(cout_p / bn) * (cin_p / 8) * 2 * 28 * bn * 8 + 4096; }

// Which kernel form a layer's launch takes (the part of launch_conv_h2's decisions that does not depend on the strip / tz geometry):
// nw8 = 512-thread work-groups, m512 = 512 voxels per work-group (then one N tile), BN / mt of the four-wave kernels.
struct H2Choice { bool nw8, m512, pair_sure; int BN, mt, mode5, tiles5, slots5; };
static H2Choice h2_choice(isx_ctx *ctx, const Level &l, int B, int cout, int pm)
{
    H2Choice c;
    int t0, t1, t2, slots1;
    const int mode = conv_mode(l);
    conv_tiles(l, mode, 1, &t0, &t1, &t2, &slots1);
    c.nw8 = conv_h2_nw8(ctx->tune, l, cout, B);
    c.BN = c.nw8 ? 64 : conv_h2_bn(ctx->tune, slots1, cout);
    // 32-column tiles get two row tiles per wave (256-voxel work-groups) when the launch is big enough to fill
    // the chip that way; the partial-sum layout then follows that tiling (mt is fixed per workspace: h2_mt)
    c.mt = (c.BN == 32 && h2_mt(ctx->tune, l, ctx->unet->wbatch)) ? 2 : 1;
    if (c.mt == 2) {                    // very narrow levels: the 256-position halo does not fit the staging plan
        int ty, tx, tp, sl;
        conv_tiles(l, mode, 2, &ty, &tx, &tp, &sl);
        if (sl > 1280) c.mt = 1;
    }
    // 512-voxel work-groups (conv_h2c16nw8m*_kernel): where the launch still fills the chip and the halo of 512 voxels fits the
    // staging plan and, beside the two slab buffers, the LDS.  Levels whose planes the linearised tiles cannot hold that way
    // (65^2: 1 608 slots) take box tiles with edge strips instead.
    // (measured, round 5: -9 ... -14 % on the Cout = 64 layers, -2 % on c2 at 65^2; the Cout % 128 == 0 layers of the linearised
    // levels (33^2) are 2-3 % faster on the 256-voxel N-tile pair; eight waves over 512 voxels x 32 columns lose 6 % to the
    // four-wave kernel of the Cout = 32 layers)
    c.m512 = false;
    c.mode5 = (mode != 0 && l.W >= 48 && l.H >= 16 && l.D % 2 == 0) ? 0 : mode;
    c.tiles5 = c.slots5 = 0;
    if (c.nw8 && ctx->tune.h2_m512) {
        int ty, tx, tp, sl;
        c.tiles5 = conv_tiles(l, c.mode5, 4, &ty, &tx, &tp, &sl);
        c.slots5 = sl;
        const size_t lds5 = (size_t)(pm == 0 ? 2 : 1) * (2 * 2 * 14 * c.BN + sl) * 16 + 512 * sizeof(int);
        const bool pair_wins = cout % 128 == 0 && c.mode5 != 0 && ctx->tune.h2_nj2 && ctx->tune.h2_m512 == 1;
        c.m512 = !pair_wins && sl <= 1536 && lds5 <= 160 * 1024 && (ctx->tune.h2_m512 == 2 || (int64_t)B * c.tiles5 * (cout / 64) >= 2048);
    }
    // the N-tile pair, where no later geometry decision (strips, the all-planes tiles of an odd level) can change the tile count its
    // "still fills the chip" test looks at: box tiles stay padded for it, linearised even levels have no variants
    {
        int ty, tx, tp, sl;
        const int tiles2 = conv_tiles(l, mode, 2, &ty, &tx, &tp, &sl);
        c.pair_sure = c.nw8 && !c.m512 && ctx->tune.h2_nj2 && cout % 128 == 0 && mode != 2 &&
                      (ctx->tune.h2_nj2 == 2 || (int64_t)B * tiles2 * (cout / 128) >= 512);
    }
    return c;
}

// Can this layer's launch up-sample its first source while staging it (H2Extra::upA)?  The 512-voxel kernel on one N tile, the
// four-wave 32-column kernel with two row tiles per wave and the 256-voxel N-tile pair have that form (conv_h2c16nw8mu_kernel /
// conv_h2c16u_kernel / conv_h2c16nw8x2u_kernel).
bool conv_h2_can_up(isx_ctx *ctx, const Level &l, int B, int cout, int pm, int ca, int ukz)
{
    const H2Choice c = h2_choice(ctx, l, B, cout, pm);
    const size_t table = (size_t)(4 * ukz + 1) * ca * sizeof(float), per = (pm == 0 ? 2 : 1) * 16;
    int ty, tx, tp, slots2;
    conv_tiles(l, conv_mode(l), 2, &ty, &tx, &tp, &slots2);
    if (c.m512 && ctx->tune.h2_m512 != 3)
        return (size_t)(pm == 0 ? 2 : 1) * (2 * 2 * 14 * 64) * 16 + (size_t)c.slots5 * per + 512 * sizeof(int) + table <= 160 * 1024;
    if (!c.nw8 && c.BN == 32 && c.mt == 2)
        return (size_t)(pm == 0 ? 2 : 1) * (2 * 14 * 32) * 16 + (size_t)slots2 * per + (256 + 4 * 32 * 2) * sizeof(float) + table <= 160 * 1024;
    if (c.pair_sure)
        return (size_t)(pm == 0 ? 2 : 1) * (2 * 2 * 14 * 64) * 16 + (size_t)slots2 * per + (256 + 8 * 64 * 2) * sizeof(float) + table <= 160 * 1024;
    return false;
}

// The 27-tap convolution of one layer over B chunks.  out_planar: write [chunk][Cout/8][voxel][8] (UNetState::planar).
int launch_conv_h2(isx_ctx *ctx, const Level &l, int B, Src sa, Src sb, const void *w_h2, const float *bias, int cout,
                   float *out, float *psum, float *psq, int no_stats, int xcd_remap, int *tiles_used, int pm, int out_planar,
                   const H2Extra *ex)
{
    ConvArgs a;
    const bool up = ex && ex->upA;
    a.upA = up ? 1 : 0; a.uH = up ? ex->uH : 0; a.uW = up ? ex->uW : 0; a.uoff = up ? ex->uoff : 0; a.ukz = up ? ex->ukz : 1;
    a.upw = up ? ex->upw : nullptr; a.upb = up ? ex->upb : nullptr;
    a.nt_out = ctx->tune.h2_nt_out;
    a.planarA = sa.planar ? 1 : 0; a.planarB = sb.planar ? 1 : 0; a.out_planar = out_planar;
    a.presplitA = sa.presplit ? 1 : 0;
    ISX_REQUIRE(!sa.presplit || (pm == 0 && sa.planar && !sa.scale && !(ex && ex->upA)), "internal: a pre-split source reached a launch that cannot copy it");
    ISX_REQUIRE(!a.out_planar || cout % 8 == 0, "internal: planar output needs Cout %% 8 == 0");
    ISX_REQUIRE(!(a.planarA && sa.C % 8) && !(a.planarB && sb.C % 8), "internal: planar source with %d / %d channels", sa.C, sb.C);
    a.srcA = sa.p; a.scaleA = sa.scale; a.shiftA = sa.shift; a.CA = sa.C;
    a.srcB = sb.p; a.scaleB = sb.scale; a.shiftB = sb.shift; a.CB = sb.C;
    a.w = (const float *)w_h2; a.bias = bias; a.out = out; a.psum = psum; a.psq = psq;
    a.D = l.D; a.H = l.H; a.W = l.W; a.Cout = cout;
    a.mode = conv_mode(l);
    a.tiles_per_chunk = conv_tiles(l, a.mode, 1, &a.nty, &a.ntx, &a.ntp, &a.slots_max);
    a.store_cols = cout;
    a.no_stats = no_stats;
    a.xcd_remap = xcd_remap;
    ISX_REQUIRE(cout % 32 == 0 && sa.C % 8 == 0 && sb.C % 8 == 0, "internal: split-f16 conv needs Cout %% 32 == 0, Cin %% 8 == 0");
    const H2Choice ch = h2_choice(ctx, l, B, cout, pm);
    const bool nw8 = ch.nw8, m512 = ch.m512;
    const int BN = ch.BN, mt = ch.mt;
    if (nw8) {                          // the tile of MT = 2, one row tile per wave
        int ty, tx, tp, sl;
        a.tiles_per_chunk = conv_tiles(l, a.mode, 2, &ty, &tx, &tp, &sl);
        a.nty = ty; a.ntx = tx; a.ntp = tp; a.slots_max = sl;
    }
    if (m512) {
        int ty, tx, tp, sl;
        a.mode = ch.mode5;
        a.tiles_per_chunk = conv_tiles(l, a.mode, 4, &ty, &tx, &tp, &sl);
        a.nty = ty; a.ntx = tx; a.ntp = tp; a.slots_max = sl;
    }
    const int rows_wg = m512 ? 512 : (nw8 ? 256 : 128 * mt);
    if (mt == 2) {
        int ty, tx, tp, sl;
        a.tiles_per_chunk = conv_tiles(l, a.mode, 2, &ty, &tx, &tp, &sl);
        a.nty = ty; a.ntx = tx; a.ntp = tp; a.slots_max = sl;
    }
    if (a.mode == 0) {
        // box tiles: interior boxes of 8 x TXW voxels, and edge strips where the remainders of H / W make them cheaper than padded
        // boxes (129 = 128 + 1: 25 padded boxes per plane pair against 3 strips)
        const int rows = rows_wg, txw = rows / 16, cap = a.slots_max;      // strips never grow the LDS footprint of the launch
        const int ny = l.H / 8, ry = l.H % 8, nx = l.W / txw, rx = l.W % txw;
        const int nyp = (l.H + 7) / 8, nxp = (l.W + txw - 1) / txw;
        a.ntx = nxp; a.nI = nyp * nxp; a.sr_n = a.sb_n = 0;
        a.sr_ty = a.sr_tx = a.sr_x0 = a.sb_ty = a.sb_tx = a.sb_y0 = 0;
        const bool may_pair = nw8 && !m512 && ctx->tune.h2_nj2 && cout % 128 == 0;       // conv_h2c16nw8x2_kernel: padded boxes only (STRIPS)
        if ((ry || rx) && ny > 0 && nx > 0 && ctx->tune.h2_strips && !may_pair) {
            // right strip: rx columns, all H rows; bottom strip: ry rows, the interior columns
            int sr_ty = 0, sr_n = 0, sb_tx = 0, sb_n = 0;
            if (rx) { sr_ty = std::min(rows / (2 * rx), cap / (4 * (rx + 2)) - 2); sr_ty = std::min(sr_ty, l.H); if (sr_ty >= 1) sr_n = (l.H + sr_ty - 1) / sr_ty; }
            if (ry) { sb_tx = std::min(rows / (2 * ry), cap / (4 * (ry + 2)) - 2); sb_tx = std::min(sb_tx, nx * txw); if (sb_tx >= 1) sb_n = (nx * txw + sb_tx - 1) / sb_tx; }
            const bool ok = (!rx || sr_n > 0) && (!ry || sb_n > 0);
            // without a right strip the bottom strip spans the padded interior columns; without a bottom strip the interior rows are padded
            const int ny_i = ry && sb_n ? ny : nyp, nx_i = rx && sr_n ? nx : nxp;
            if (ry && sb_n && !(rx && sr_n)) { sb_tx = std::min(sb_tx, l.W); sb_n = (l.W + sb_tx - 1) / sb_tx; }
            if (ok && ny_i * nx_i + sr_n + sb_n < nyp * nxp) {
                a.ntx = nx_i; a.nI = ny_i * nx_i;
                if (rx && sr_n) { a.sr_n = sr_n; a.sr_ty = sr_ty; a.sr_tx = rx; a.sr_x0 = nx * txw; }
                if (ry && sb_n) { a.sb_n = sb_n; a.sb_ty = ry; a.sb_tx = sb_tx; a.sb_y0 = ny * 8; }
                a.slots_max = std::max(a.slots_max, std::max(a.sr_n ? 4 * (a.sr_ty + 2) * (a.sr_tx + 2) : 0, a.sb_n ? 4 * (a.sb_ty + 2) * (a.sb_tx + 2) : 0));
            }
        }
        a.tiles_per_chunk = ((l.D + 1) / 2) * (a.nI + a.sr_n + a.sb_n);
    }
    a.tz = a.mode == 1 ? 2 : 1;
    if (a.mode == 2 && l.D > 1 && l.D <= 8 && ctx->tune.h2_tz) {
        // odd, shallow levels: all planes in one tile when that needs fewer tiles (the rows beyond tz * (rows / tz) stay empty)
        const int rows = rows_wg, cap = nw8 ? 1536 : (mt == 2 ? 1280 : 1024);
        const int tp = rows / l.D, ntp = (l.H * l.W + tp - 1) / tp;
        const int lh = tp + 2 * ((tp + l.W - 1) / l.W + 1) + 2 * (l.W + 2) + 2, slots = (l.D + 2) * lh;
        if (tp >= 8 && ntp < a.tiles_per_chunk && slots <= cap) { a.tz = l.D; a.ntp = ntp; a.tiles_per_chunk = ntp; a.slots_max = slots; }
    }
    ISX_REQUIRE(a.slots_max <= (nw8 ? 1536 : (mt == 2 ? 1280 : 1024)), "internal: halo tile of %d slots does not fit the staging plan", a.slots_max);
    // N-tile pairs (conv_h2c16nw8x2_kernel): Cout % 128 == 0 and a launch that still fills the chip
    // (512-voxel work-groups take one N tile: with two the 64 x 128 wave tile has 55 spilled registers; ISX_H2_M512=3 runs it anyway)
    const int nj = (m512 && ctx->tune.h2_m512 != 3) ? 1 : ((nw8 && ctx->tune.h2_nj2 && cout % 128 == 0 &&
                    (ctx->tune.h2_nj2 == 2 || (int64_t)B * a.tiles_per_chunk * (cout / 128) >= 512)) ? 2 : 1);
    a.ntiles = cout / (BN * nj);
    {   // the output tile of the epilogue (rows x (BN + 4) floats) aliases the staging buffers: slabs + halo tile; the halo
        // part is padded where the single-product modes' smaller buffers would not hold it
        const size_t rows = rows_wg, tile = rows * (BN + 4) * sizeof(float) + (m512 ? 8 * BN * 2 * sizeof(float) : 0);   // (M512: + the partial sums)
        const size_t per = (pm == 0 ? 2 : 1) * 16, slabs = (size_t)(pm == 0 ? 2 : 1) * (nw8 ? 2 : 1) * 2 * 14 * BN * 16;
        if (slabs + (size_t)a.slots_max * per < tile) a.slots_max = (int)((tile - slabs + per - 1) / per);
    }
    size_t lds = (size_t)((pm == 0 ? 2 : 1) * ((nw8 ? 2 : 1) * 2 * 14 * BN + a.slots_max)) * 16 + (rows_wg + (m512 ? 0 : (nw8 ? 8 : 4) * BN * 2)) * sizeof(float);
    if (up) {
        ISX_REQUIRE((m512 && nj == 1) || (!nw8 && BN == 32 && mt == 2) || (nw8 && !m512 && nj == 2), "internal: this launch has no up-sampling form (conv_h2_can_up)");
        ISX_REQUIRE(a.upw && a.upb && sa.C % 8 == 0 && (a.ukz == 1 || (a.ukz == 2 && l.D % 2 == 0)), "internal: up-sampling on read needs the transposed convolution's weights");
        lds += (size_t)(4 * a.ukz + 1) * sa.C * sizeof(float);
    }
    ISX_REQUIRE(lds <= 160 * 1024, "internal: split-f16 conv tile needs %zu bytes of LDS", lds);
    if (tiles_used) *tiles_used = a.tiles_per_chunk;        // layout of the BN partial sums this launch writes
    const unsigned grid = (unsigned)(B * a.tiles_per_chunk * a.ntiles);
    bool &attr_done = ctx->h2_attr_done;                  // per context (setting it again from another context is harmless)
#define ISX_H2_ATTR(K_) ISX_HIP(hipFuncSetAttribute((const void *)K_, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
#define ISX_H2_ATTR3(K_) ISX_H2_ATTR(K_<0>) ISX_H2_ATTR(K_<1>) ISX_H2_ATTR(K_<2>)
    if (!attr_done) {
        ISX_H2_ATTR((conv_h2c16_kernel<64, 1, 0>)) ISX_H2_ATTR((conv_h2c16_kernel<64, 1, 1>)) ISX_H2_ATTR((conv_h2c16_kernel<64, 1, 2>))
        ISX_H2_ATTR((conv_h2c16_kernel<32, 1, 0>)) ISX_H2_ATTR((conv_h2c16_kernel<32, 1, 1>)) ISX_H2_ATTR((conv_h2c16_kernel<32, 1, 2>))
        ISX_H2_ATTR((conv_h2c16_kernel<32, 2, 0>)) ISX_H2_ATTR((conv_h2c16_kernel<32, 2, 1>)) ISX_H2_ATTR((conv_h2c16_kernel<32, 2, 2>))
        ISX_H2_ATTR3(conv_h2c16nw8_kernel) ISX_H2_ATTR3(conv_h2c16nw8x2_kernel) ISX_H2_ATTR3(conv_h2c16nw8m_kernel) ISX_H2_ATTR3(conv_h2c16nw8mx2_kernel) ISX_H2_ATTR3(conv_h2c16nw8mu_kernel) ISX_H2_ATTR3(conv_h2c16u_kernel) ISX_H2_ATTR3(conv_h2c16nw8x2u_kernel)
        attr_done = true;
    }
#undef ISX_H2_ATTR3
#undef ISX_H2_ATTR
#define ISX_H2_GO16(BN_, MT_) \
    do { if (pm == 0) conv_h2c16_kernel<BN_, MT_, 0><<<grid, 256, lds, ctx->stream>>>(a); \
         else if (pm == 1) conv_h2c16_kernel<BN_, MT_, 1><<<grid, 256, lds, ctx->stream>>>(a); \
         else conv_h2c16_kernel<BN_, MT_, 2><<<grid, 256, lds, ctx->stream>>>(a); } while (0)
    if (up && nw8 && !m512) {
        if (pm == 0) conv_h2c16nw8x2u_kernel<0><<<grid, 512, lds, ctx->stream>>>(a);
        else if (pm == 1) conv_h2c16nw8x2u_kernel<1><<<grid, 512, lds, ctx->stream>>>(a);
        else conv_h2c16nw8x2u_kernel<2><<<grid, 512, lds, ctx->stream>>>(a);
    } else if (up && m512) {
        if (pm == 0) conv_h2c16nw8mu_kernel<0><<<grid, 512, lds, ctx->stream>>>(a);
        else if (pm == 1) conv_h2c16nw8mu_kernel<1><<<grid, 512, lds, ctx->stream>>>(a);
        else conv_h2c16nw8mu_kernel<2><<<grid, 512, lds, ctx->stream>>>(a);
    } else if (up) {
        if (pm == 0) conv_h2c16u_kernel<0><<<grid, 256, lds, ctx->stream>>>(a);
        else if (pm == 1) conv_h2c16u_kernel<1><<<grid, 256, lds, ctx->stream>>>(a);
        else conv_h2c16u_kernel<2><<<grid, 256, lds, ctx->stream>>>(a);
    } else if (m512 && nj == 2) {
        if (pm == 0) conv_h2c16nw8mx2_kernel<0><<<grid, 512, lds, ctx->stream>>>(a);
        else if (pm == 1) conv_h2c16nw8mx2_kernel<1><<<grid, 512, lds, ctx->stream>>>(a);
        else conv_h2c16nw8mx2_kernel<2><<<grid, 512, lds, ctx->stream>>>(a);
    } else if (m512) {
        if (pm == 0) conv_h2c16nw8m_kernel<0><<<grid, 512, lds, ctx->stream>>>(a);
        else if (pm == 1) conv_h2c16nw8m_kernel<1><<<grid, 512, lds, ctx->stream>>>(a);
        else conv_h2c16nw8m_kernel<2><<<grid, 512, lds, ctx->stream>>>(a);
    } else if (nw8 && nj == 2) {
        if (pm == 0) conv_h2c16nw8x2_kernel<0><<<grid, 512, lds, ctx->stream>>>(a);
        else if (pm == 1) conv_h2c16nw8x2_kernel<1><<<grid, 512, lds, ctx->stream>>>(a);
        else conv_h2c16nw8x2_kernel<2><<<grid, 512, lds, ctx->stream>>>(a);
    } else if (nw8) {
        if (pm == 0) conv_h2c16nw8_kernel<0><<<grid, 512, lds, ctx->stream>>>(a);
        else if (pm == 1) conv_h2c16nw8_kernel<1><<<grid, 512, lds, ctx->stream>>>(a);
        else conv_h2c16nw8_kernel<2><<<grid, 512, lds, ctx->stream>>>(a);
    } else if (BN == 64) ISX_H2_GO16(64, 1);
    else if (mt == 2) ISX_H2_GO16(32, 2);
    else ISX_H2_GO16(32, 1);
#undef ISX_H2_GO16
    ISX_CHECK_LAUNCH();
#if defined(ISX_H2_STAMP) && ISX_H2_STAMP == 2
    {
        static int nlaunch = 0;                     // no synchronisation per launch: the clock of the back-to-back sequence
        if (++nlaunch % 288 == 0) {
            unsigned long long hs[12] = {0};
            ISX_HIP(hipStreamSynchronize(ctx->stream));
            ISX_HIP(hipMemcpyFromSymbol(hs, HIP_SYMBOL(h2_stamp), sizeof(hs)));
            if (getenv("ISX_H2_STAMP_PRINT") && hs[11])
                fprintf(stderr, "h2clock after %d launches: %.3f GHz (sampled work-groups: %.3e cycles)\n", nlaunch, (double)hs[10] / (double)hs[11] * 0.1, (double)hs[10]);
            memset(hs, 0, sizeof(hs));
            ISX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(h2_stamp), hs, sizeof(hs)));
        }
    }
#elif defined(ISX_H2_STAMP)
    {
        unsigned long long hs[12] = {0};
        ISX_HIP(hipStreamSynchronize(ctx->stream));
        ISX_HIP(hipMemcpyFromSymbol(hs, HIP_SYMBOL(h2_stamp), sizeof(hs)));
        const double wg = (double)hs[9], st = (double)hs[8];
        if (getenv("ISX_H2_STAMP_PRINT") && wg > 0 && st > 0)
            fprintf(stderr, "h2stamp B=%d D=%d H=%d W=%d cin=%d cout=%d BN=%d mt=%d grid=%u stages/wg=%.1f | per stage: bar1 %.0f loadwait %.0f store %.0f bar2 %.0f issue %.0f matrix %.0f | per wg: prologue %.0f epilogue %.0f | clock %.2f GHz\n",
                    B, l.D, l.H, l.W, sa.C + sb.C, cout, BN, mt, grid, st / wg, hs[0] / st, hs[1] / st, hs[2] / st, hs[3] / st, hs[4] / st,
                    hs[5] / st, hs[6] / wg, hs[7] / wg, hs[11] ? (double)hs[10] / (double)hs[11] * 0.1 : 0.0);
        memset(hs, 0, sizeof(hs));
        ISX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(h2_stamp), hs, sizeof(hs)));
    }
#endif
    return ISX_OK;
}

}  // namespace isx
