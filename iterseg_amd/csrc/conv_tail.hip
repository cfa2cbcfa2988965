// conv_tail.hip -- the thin tail convolution (c8.conv0: Cin >= 32 -> <= 5 channels, 3x3x3) in ONE pass on the f16 matrix cores.
//
// Replaces the same reference operator as conv_h2_kernel<32, 1, ...> + tail_combine_kernel (nn.Conv3d inside
// ConvModule.forward, src/iterseg/unet.py:63-79,91-106, for the module c8_0 built at unet.py:196-210) when the context's
// precision is a reduced one; arithmetic and operand split exactly as conv_h2.hip (x = x_h + x_l / 2^11, three
// v_mfma_f32_16x16x32_f16 per product block, fp32 accumulators; PM = 1 / 2: plain f16 / bf16 operands).
//
// Why its own kernel.  The layer has 3 % of the network's FLOPs and took 10.7 % of the convolution time (r03: 9.1 ms per
// 64-chunk launch set + 0.57 ms for the combine): on the generic 128-voxel tile it is bound by its halo gather (halo = 3.1 x
// the tile, 3 900-5 200 cycles of load issue per K stage against 1 100 of matrix work), and it wrote and
// re-read a [voxel][16] partial-sum tensor (84 MB per chunk) for the x-shift.  Here:
//   * the x-taps stay folded into the N tile (column = dx * nout + o, 15 of 16 columns), so the matrix pass has 9 (dz, dy)
//     taps = 3 tap quads of the 16x16x32 instruction, and the halo has NO y/z-diagonal x-neighbours to fetch;
//   * a work-group (4 waves) owns 5 (z) x 4 (y) x 30 (x) output voxels on a halo of 7 x 6 x 32 slots (2.24 x the tile), the
//     32 x-positions of a halo row being the 30 outputs + the two positions whose partial sums the x-shift needs -- the
//     shift happens in LDS in the epilogue and the partial sums never reach memory;
//   * 52 KB of LDS and four waves per work-group: two work-groups per CU, so one's staging / prologue / epilogue runs beside
//     the other's matrix phase (the first version -- eight waves over all ten z planes with the whole weight tensor resident
//     in LDS, one work-group per CU -- took 7.6 ms per launch set: every wave of the CU in the same phase);
//   * ten 16-row accumulator tiles per wave (80 registers), B fragments of a stage read once (6 reads) for 90 MFMAs, matrix
//     phase software-pipelined by hand (fragments one tile pair ahead, one halo load of the next stage per step).
// BatchNorm partial sums: per work-group, fixed summation order (deterministic), reduced by bn_finalize_kernel like every
// other layer's.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "unet_state.h"
#include "h2_split.h"

namespace isx {

typedef _Float16 t_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 t_bf16x8 __attribute__((ext_vector_type(8)));
typedef float t_f32x4 __attribute__((ext_vector_type(4)));

struct TailArgs {
    const float *srcA, *scaleA, *shiftA;   // channels [0, CA): [B][M][CA] (or planar [B][CA/8][M][8]) + optional BN+ReLU
    const float *srcB, *scaleB, *shiftB;   // channels [CA, CA+CB)
    int CA, CB, planarA, planarB;
    const uint4 *w;        // pack_h2_fold_weights: [Cin/8][hi|lo][12 tap slots][32 columns][8 channels] halves (columns 0..15 read)
    const float *bias;     // [nout]
    float *out;            // [B][M][nout] raw convolution output (+ bias)
    float *psum, *psq;     // [B][tiles][nout]
    int D, H, W, nout;
    int nzt, nyt, nxt;     // tiles per chunk = nzt * nyt * nxt
    int xcd_remap;
    // UP kernels: source A is the COARSE input of the transposed convolution in front of this layer (up3, kz = 1), up-sampled
    // while it is staged -- slot voxel (z, y, x) reads (z, (y + uoff) >> 1, (x + uoff) >> 1) of a (D, uH, uW) tensor, BatchNorm +
    // ReLU as usual, then * upw[parity][c] + upb[c] (UpConv::w / bias; parity = ((y + uoff) & 1) * 2 + ((x + uoff) & 1)): what
    // upsample_src_kernel would have written to a 4 x larger tensor (5.4 GB per 64-chunk launch set) and this kernel read back
    int uH, uW, uoff;
    const float *upw, *upb;
};

static constexpr int TL_TZ = 5, TL_TY = 4, TL_TX = 30, TL_PX = 32;           // output tile; PX = TX + 2 positions per halo row
static constexpr int TL_HY = TL_TY + 2, TL_HZ = TL_TZ + 2;
static constexpr int TL_SLOTS = TL_HZ * TL_HY * TL_PX;                       // 1344 halo slots
static constexpr int TL_THREADS = 256, TL_NW = TL_THREADS / 64;

static constexpr int TL_ROWS = TL_TZ * TL_TY * TL_PX;                        // 640 partial-sum positions = 40 row tiles of 16
static constexpr int TL_MT = TL_ROWS / 16 / TL_NW;                           // 10 row tiles per wave
static constexpr int TL_ZYW = TL_TZ * TL_TY / TL_NW;                         // 5 halo rows (z-local * TY + y-local) per wave
static constexpr int TL_PSTRIDE = 20;                                        // floats per position of the LDS partial-sum tile (conflict-free C stores)
static constexpr int TL_BH = 12 * 16;                                        // uint4 per half of a stage's weights (12 tap slots x 16 columns)
static_assert(TL_MT == 2 * TL_ZYW && TL_MT % 2 == 0, "two 16-row tiles per halo row, tiles processed in pairs");

__device__ __forceinline__ uint32_t tl_pack_f16_rn(float x, float y)
{
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    union { h2v v; uint32_t u; } H;
    H.v[0] = (_Float16)__builtin_amdgcn_fmed3f(x, -H2_MAX, H2_MAX);
    H.v[1] = (_Float16)__builtin_amdgcn_fmed3f(y, -H2_MAX, H2_MAX);
    return H.u;
}
__device__ __forceinline__ uint32_t tl_pack_bf16_rn(float x, float y)
{
    uint32_t a = __float_as_uint(x), b = __float_as_uint(y);
    a += 0x7FFFu + ((a >> 16) & 1u);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xFFFF0000u);
}

// ABL (development builds, -DISX_TAIL_DEV, timing only -- wrong results): 1 no staging arithmetic, 2 no matrix phase, 3 no halo
// loads, 4 no epilogue
// 1 KiB of weights straight into LDS (see lds_dma16 in conv_h2.hip for why this is inline assembly)
__device__ __forceinline__ void tl_lds_dma16(const uint4 *g, uint4 *lds_wave_base)
{
    const uint32_t la = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(la), "v"(g) : "m0", "memory");
}

template <int PM, int ABL = 0, bool UP = false>
__global__ __launch_bounds__(TL_THREADS) __attribute__((amdgpu_waves_per_eu(2))) void conv_tail_kernel(const TailArgs a)
{
    constexpr bool SPLIT = PM == 0;
    extern __shared__ __attribute__((aligned(16))) uint4 tl_smem[];
    const int nsteps = (a.CA + a.CB) >> 3;
    // LDS: [halo hi: SLOTS][halo lo: SLOTS (split only)][this stage's weights: (hi|lo) x 12 x 16]; the epilogue's partial-sum
    // tile (ROWS x PSTRIDE floats) and the reduction scratch alias it after the last stage
    uint4 *AsH = tl_smem;
    uint4 *AsL = AsH + TL_SLOTS;
    uint4 *Bs = AsH + (SPLIT ? 2 : 1) * TL_SLOTS;             // two buffers of (hi|lo) x 12 x 16, filled by LDS-DMA alternately
    float *upws = (float *)(Bs + 4 * TL_BH);                  // (UP) [4 parities][CA] weights + [CA] bias of the transposed convolution
    int *gvs = (int *)(upws + 5 * 64);                        // per halo slot: voxel index (-1: outside the volume = zero padding)
    int *gus = gvs + TL_SLOTS;                                // (UP) per halo slot: coarse voxel | parity << 28

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, h4 = lane >> 4;
    int lin = blockIdx.x;
    if (a.xcd_remap) {
        const int per = gridDim.x >> 3, main = per << 3;
        if (lin < main) lin = (lin & 7) * per + (lin >> 3);
    }
    const int tiles_per_chunk = a.nzt * a.nyt * a.nxt;
    const int b = lin / tiles_per_chunk, tile = lin % tiles_per_chunk;
    // z fastest: the two z tiles of a (y, x) column run side by side and share two halo planes in L2
    const int zt = tile % a.nzt, xt = (tile / a.nzt) % a.nxt, yt = tile / (a.nzt * a.nxt);
    const int z0 = zt * TL_TZ, y0 = yt * TL_TY, x0 = xt * TL_TX;
    const int D = a.D, H = a.H, W = a.W;
    const int64_t M = (int64_t)D * H * W;

    // ---- halo slots of this thread: voxel index, -1 outside the volume (zero padding), -2 beyond the slot list -----------
    // (the coarse voxel | parity << 28 of the up-sampling kernels comes from the slot's coordinates, not from three run-time divisions of
    //  its voxel index: 18 divisions = ~650 vector instructions per thread of a work-group that issues ~5 000 in its K loop)
    auto slot_voxel = [&](int s, int &u) __attribute__((always_inline)) -> int {
        u = 0;
        if (s >= TL_SLOTS) return -2;
        const int hz = s / (TL_HY * TL_PX), hy = (s / TL_PX) % TL_HY, hx = s % TL_PX;
        const int z = z0 + hz - 1, y = y0 + hy - 1, x = x0 + hx - 1;
        if (!(z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W)) return -1;
        if constexpr (UP) {
            const int yo = y + a.uoff, xo = x + a.uoff;
            u = ((z * a.uH + (yo >> 1)) * a.uW + (xo >> 1)) | ((((yo & 1) << 1) | (xo & 1)) << 28);
        }
        return (z * H + y) * W + x;
    };
    // (in LDS, not in registers: twelve registers the split kernel does not have; a thread reads back only what it wrote)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int sl = tid + TL_THREADS * k;
        if (sl < TL_SLOTS) {
            int u;
            const int g = slot_voxel(sl, u);
            gvs[sl] = g;
            if constexpr (UP) gus[sl] = u;
        }
    }
    const int64_t Mup = UP ? (int64_t)D * a.uH * a.uW : 0;
    if constexpr (UP) {
        for (int e = tid; e < 5 * a.CA; e += TL_THREADS) upws[e] = e < 4 * a.CA ? a.upw[e] : a.upb[e - 4 * a.CA];
    }
    bool stage_up = false, st_up = false;
    int up_cc = 0, st_cc = 0;
    // this thread's share of a stage's weights: uint4 e = tid (and 256 + tid for waves 0, 1) of the [hi|lo][12][16] block, read from
    // the 32-column pack  [Cin/8][hi|lo][12 tap slots][32 columns]  straight into LDS (global_load_lds_dwordx4: no registers)
    auto wsrc = [&](int e) __attribute__((always_inline)) -> int { return ((e / TL_BH) * 12 + ((e % TL_BH) >> 4)) * 32 + (e & 15); };
    const int wo0 = wsrc(tid), wo1 = wsrc(tid + TL_THREADS);
    const bool wb1 = SPLIT && wave < (2 * TL_BH - TL_THREADS) / 64;
#define TL_DMA_B(step_)                                                                           \
    {                                                                                             \
        const uint4 *wst = a.w + (int64_t)(step_) * (2 * 12 * 32);                                \
        uint4 *bd = Bs + ((step_) & 1) * (2 * TL_BH) + 64 * wave;                                 \
        tl_lds_dma16(wst + wo0, bd);                                                              \
        if (wb1) tl_lds_dma16(wst + wo1, bd + TL_THREADS);                                        \
    }

    // ---- accumulators: row tile m of this wave = halo row zy = ZYW * wave + m / 2 (z-local * TY + y-local), x half m & 1 ----
    t_f32x4 c_hi[TL_MT], c_lo[SPLIT ? TL_MT : 1];
    int aslot[TL_MT];
#pragma unroll
    for (int m = 0; m < TL_MT; ++m) {
        c_hi[m] = t_f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (SPLIT) c_lo[m] = t_f32x4{0.f, 0.f, 0.f, 0.f};
        const int zy = TL_ZYW * wave + (m >> 1), zl = zy / TL_TY, yl = zy % TL_TY;
        aslot[m] = ((zl + 1) * TL_HY + (yl + 1)) * TL_PX + 16 * (m & 1) + l16;
    }
    // tap quad q, lane group h4 -> tap t = 4 q + h4 (dz = t / 3, dy = t % 3; slots 9..11 carry zero weights: any valid slot)
    int toq[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        int t = 4 * q + h4;
        if (t > 8) t = 8;
        toq[q] = (t / 3 - 1) * (TL_HY * TL_PX) + (t % 3 - 1) * TL_PX;
    }

    float4 ra0, rc0, ra1, rc1, ra2, rc2, ra3, rc3, ra4, rc4, ra5, rc5, vs0, vs1, vh0, vh1;
    float relu_lo = 0.f;
    const float *ld_src = nullptr;
    int ld_C = 0;
#define TL_LOAD_HEAD(step_)                                                                       \
    {                                                                                             \
        const int c0 = (step_) << 3;                                                              \
        const float *src, *sc, *sh;                                                               \
        int C, cc, planar;                                                                        \
        if (c0 < a.CA) { src = a.srcA; sc = a.scaleA; sh = a.shiftA; C = a.CA; cc = c0; planar = a.planarA; } \
        else { src = a.srcB; sc = a.scaleB; sh = a.shiftB; C = a.CB; cc = c0 - a.CA; planar = a.planarB; }    \
        if (sc) {                                                                                 \
            const int64_t so = (int64_t)b * C + cc;                                               \
            vs0 = *(const float4 *)&sc[so]; vs1 = *(const float4 *)&sc[so + 4];                   \
            vh0 = *(const float4 *)&sh[so]; vh1 = *(const float4 *)&sh[so + 4];                   \
            relu_lo = 0.f;                                                                        \
        } else {                      /* a source without BatchNorm: x * 1 + 0, max(x, -inf) are exact */ \
            vs0 = make_float4(1.f, 1.f, 1.f, 1.f); vs1 = vs0;                                     \
            vh0 = make_float4(0.f, 0.f, 0.f, 0.f); vh1 = vh0;                                     \
            relu_lo = -__builtin_inff();                                                          \
        }                                                                                         \
        stage_up = UP && c0 < a.CA;                                                               \
        up_cc = cc;                                                                               \
        const int64_t Ms = stage_up ? Mup : M;                                                    \
        if (planar) { ld_src = src + ((int64_t)b * (C >> 3) + (cc >> 3)) * Ms * 8; ld_C = 8; }    \
        else { ld_src = src + (int64_t)b * Ms * C + cc; ld_C = C; }                               \
    }
#define TL_LOAD_A(k)                                                                              \
    if (ABL != 3 && (k < 5 || wave == 0)) {                                                       \
        const int gl_ = (UP && stage_up) ? (gus[tid + TL_THREADS * k] & 0x0FFFFFFF) : gvs[tid + TL_THREADS * k]; \
        const float *pp = ld_src + (uint32_t)(gl_ > 0 ? gl_ : 0) * (uint32_t)ld_C;                \
        ra##k = *(const float4 *)pp;                                                              \
        rc##k = *(const float4 *)(pp + 4);                                                        \
    }
#define TL_AFFL(v, s_, h_)                                                                        \
    v.x = fmaxf(fmaf(v.x, s_.x, h_.x), relu_lo); v.y = fmaxf(fmaf(v.y, s_.y, h_.y), relu_lo);     \
    v.z = fmaxf(fmaf(v.z, s_.z, h_.z), relu_lo); v.w = fmaxf(fmaf(v.w, s_.w, h_.w), relu_lo);
#define TL_STORE_A(k)                                                                             \
    if (k < 5 || wave == 0) {                                                                     \
        float4 v = ra##k, w = rc##k;                                                              \
        if constexpr (ABL != 1) { TL_AFFL(v, vs0, vh0) TL_AFFL(w, vs1, vh1) }                     \
        const int gs_ = gvs[tid + TL_THREADS * k];                                                \
        if (UP && st_up) {                                                                        \
            const float *uw = &upws[((unsigned)gus[tid + TL_THREADS * k] >> 28) * a.CA + st_cc], *ub = &upws[4 * a.CA + st_cc]; \
            const float4 u0 = *(const float4 *)uw, u1 = *(const float4 *)(uw + 4);                \
            const float4 b0 = *(const float4 *)ub, b1 = *(const float4 *)(ub + 4);                \
            v.x = fmaf(v.x, u0.x, b0.x); v.y = fmaf(v.y, u0.y, b0.y); v.z = fmaf(v.z, u0.z, b0.z); v.w = fmaf(v.w, u0.w, b0.w); \
            w.x = fmaf(w.x, u1.x, b1.x); w.y = fmaf(w.y, u1.y, b1.y); w.z = fmaf(w.z, u1.z, b1.z); w.w = fmaf(w.w, u1.w, b1.w); \
        }                                                                                         \
        uint4 hi, lo = make_uint4(0u, 0u, 0u, 0u);                                                \
        if constexpr (ABL == 1) { hi = make_uint4(__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)); \
                                  lo = make_uint4(__float_as_uint(w.x), __float_as_uint(w.y), __float_as_uint(w.z), __float_as_uint(w.w)); } else \
        if constexpr (SPLIT) split8(v, w, hi, lo);                                                \
        else if constexpr (PM == 1) { hi.x = tl_pack_f16_rn(v.x, v.y); hi.y = tl_pack_f16_rn(v.z, v.w); hi.z = tl_pack_f16_rn(w.x, w.y); hi.w = tl_pack_f16_rn(w.z, w.w); } \
        else { hi.x = tl_pack_bf16_rn(v.x, v.y); hi.y = tl_pack_bf16_rn(v.z, v.w); hi.z = tl_pack_bf16_rn(w.x, w.y); hi.w = tl_pack_bf16_rn(w.z, w.w); } \
        const bool dead = gs_ < 0;                                                                \
        hi.x = dead ? 0u : hi.x; hi.y = dead ? 0u : hi.y; hi.z = dead ? 0u : hi.z; hi.w = dead ? 0u : hi.w; \
        if constexpr (SPLIT) { lo.x = dead ? 0u : lo.x; lo.y = dead ? 0u : lo.y; lo.z = dead ? 0u : lo.z; lo.w = dead ? 0u : lo.w; } \
        AsH[tid + TL_THREADS * k] = hi;                                                           \
        if constexpr (SPLIT) AsL[tid + TL_THREADS * k] = lo;                                      \
    }

    if constexpr (ABL == 3) { ra0 = rc0 = ra1 = rc1 = ra2 = rc2 = ra3 = rc3 = ra4 = rc4 = ra5 = rc5 = make_float4(0.5f, 0.25f, 0.125f, 1.f); }
    TL_DMA_B(0)
    TL_LOAD_HEAD(0)
    TL_LOAD_A(0) TL_LOAD_A(1) TL_LOAD_A(2) TL_LOAD_A(3) TL_LOAD_A(4) TL_LOAD_A(5)
    union FR { uint4 u; t_f16x8 v; t_bf16x8 w; };
    for (int step = 0; step < nsteps; ++step) {
        __syncthreads();                                        // every wave has read the previous stage's fragments
        st_up = stage_up; st_cc = up_cc;                        // of the stage being stored (the next TL_LOAD_HEAD overwrites them)
        TL_STORE_A(0) TL_STORE_A(1) TL_STORE_A(2) TL_STORE_A(3) TL_STORE_A(4) TL_STORE_A(5)
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the stage's weights have landed
        __syncthreads();
        const bool more = step + 1 < nsteps;
        if (more) { TL_DMA_B(step + 1) TL_LOAD_HEAD(step + 1) }
        const uint4 *bsh = Bs + (step & 1) * (2 * TL_BH) + h4 * 16 + l16, *bsl = bsh + TL_BH;
        FR bh[3], bl[SPLIT ? 3 : 1];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            bh[q].u = bsh[(4 * q) * 16];
            if constexpr (SPLIT) bl[q].u = bsl[(4 * q) * 16];
        }
        // one step = a PAIR of row tiles of one tap quad (6 matrix instructions, the two dependent ones of a tile three apart);
        // the fragments of the next pair are read before the matrix instructions of this one are issued, and one of the next
        // stage's halo loads goes out per step
        constexpr int NS = ABL == 2 ? 0 : 3 * (TL_MT / 2);
        if constexpr (ABL == 2) { if (more) { TL_LOAD_A(0) TL_LOAD_A(1) TL_LOAD_A(2) TL_LOAD_A(3) TL_LOAD_A(4) TL_LOAD_A(5) } c_hi[0][0] += bh[0].v[0] + bh[1].v[1] + bh[2].v[2]; if constexpr (SPLIT) c_lo[0][0] += bl[0].v[0]; }
        FR ah[2][2], al[SPLIT ? 2 : 1][2];
#define TL_FRAG(s_)                                                                               \
        {                                                                                         \
            const int q_ = (s_) / (TL_MT / 2), m_ = 2 * ((s_) % (TL_MT / 2));                     \
            ah[(s_) & 1][0].u = AsH[aslot[m_] + toq[q_]];                                         \
            ah[(s_) & 1][1].u = AsH[aslot[m_ + 1] + toq[q_]];                                     \
            if constexpr (SPLIT) { al[(s_) & 1][0].u = AsL[aslot[m_] + toq[q_]]; al[(s_) & 1][1].u = AsL[aslot[m_ + 1] + toq[q_]]; } \
        }
        if constexpr (ABL != 2) TL_FRAG(0)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int q = s / (TL_MT / 2), m = 2 * (s % (TL_MT / 2)), r = s & 1;
            if (s + 1 < NS) TL_FRAG(s + 1)
            if (more) {
                if (s == 0) { TL_LOAD_A(0) } else if (s == 2) { TL_LOAD_A(1) } else if (s == 4) { TL_LOAD_A(2) }
                else if (s == 6) { TL_LOAD_A(3) } else if (s == 8) { TL_LOAD_A(4) } else if (s == 10) { TL_LOAD_A(5) }
            }
            if constexpr (SPLIT) {
                c_hi[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r][0].v, bh[q].v, c_hi[m], 0, 0, 0);
                c_lo[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r][0].v, bl[q].v, c_lo[m], 0, 0, 0);
                c_hi[m + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r][1].v, bh[q].v, c_hi[m + 1], 0, 0, 0);
                c_lo[m + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r][1].v, bl[q].v, c_lo[m + 1], 0, 0, 0);
                c_lo[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[r][0].v, bh[q].v, c_lo[m], 0, 0, 0);
                c_lo[m + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[r][1].v, bh[q].v, c_lo[m + 1], 0, 0, 0);
            } else if constexpr (PM == 1) {
                c_hi[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r][0].v, bh[q].v, c_hi[m], 0, 0, 0);
                c_hi[m + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r][1].v, bh[q].v, c_hi[m + 1], 0, 0, 0);
            } else {
                c_hi[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r][0].w, bh[q].w, c_hi[m], 0, 0, 0);
                c_hi[m + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[r][1].w, bh[q].w, c_hi[m + 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef TL_FRAG
    }
#undef TL_DMA_B
#undef TL_LOAD_HEAD
#undef TL_LOAD_A
#undef TL_AFFL
#undef TL_STORE_A

    // ---- epilogue: partial sums -> LDS, x-shift + bias, stores, BatchNorm partial sums -----------------------------------
    // C/D of 16x16: column = lane & 15, row = 4 * (lane >> 4) + register.  Position p = (zy * 32 + px) holds its 16 columns
    // at P[p * PSTRIDE ..]; the four lane groups of a store land 80 floats apart = 16 banks: conflict-free.
    float *P = (float *)tl_smem;
    if constexpr (ABL == 4) { if (c_hi[0][0] == 123.456f) a.out[tid] = c_hi[1][1] + c_hi[9][3]; return; }
    __syncthreads();                                            // the last fragments are read: the halo / weight LDS is free
#pragma unroll
    for (int m = 0; m < TL_MT; ++m) {
        const int zy = TL_ZYW * wave + (m >> 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = zy * TL_PX + 16 * (m & 1) + 4 * h4 + i;
            float v;
            if constexpr (SPLIT) v = fmaf(c_lo[m][i], 1.f / H2_SCALE, c_hi[m][i]);
            else v = c_hi[m][i];
            P[p * TL_PSTRIDE + l16] = v;
        }
    }
    __syncthreads();
    // out[z][y][x][o] = bias[o] + sum over dx of P[(z, y, px = x - x0 + dx)][dx * nout + o]  (px = 0 is x0 - 1: a position
    // outside the volume staged zeros, so its partial sums are the convolution's zero padding).  Thread f < 30 * nout owns
    // element f = (x - x0) * nout + o of every output row of the tile: consecutive lanes store consecutive floats.
    const int nout = a.nout;
    const int rowf = TL_TX * nout;
    float s = 0.f, q = 0.f;
    if (tid < rowf) {
        const int j = tid / nout, o = tid - j * nout, x = x0 + j;
        const float bsv = a.bias[o];
        const float *pr = &P[j * TL_PSTRIDE + o];
        if (x < W) {
#pragma unroll 4
            for (int zy = 0; zy < TL_TZ * TL_TY; ++zy) {
                const int z = z0 + zy / TL_TY, y = y0 + zy % TL_TY;
                if (z < D && y < H) {
                    const float *pz = pr + zy * (TL_PX * TL_PSTRIDE);
                    const float v = bsv + pz[0] + pz[TL_PSTRIDE + nout] + pz[2 * TL_PSTRIDE + 2 * nout];
                    a.out[((int64_t)b * M + ((int64_t)z * H + y) * W + x0) * nout + tid] = v;
                    s += v;
                    q += v * v;
                }
            }
        }
    }
    // deterministic reduction: the 30 threads of a channel in fixed order
    float *red = P + TL_ROWS * TL_PSTRIDE;                       // behind the partial-sum tile
    red[2 * tid] = s;
    red[2 * tid + 1] = q;
    __syncthreads();
    if (tid < nout && a.psum) {
        float ss = 0.f, qq = 0.f;
        for (int t = tid; t < rowf; t += nout) { ss += red[2 * t]; qq += red[2 * t + 1]; }
        const int64_t oi = ((int64_t)b * tiles_per_chunk + tile) * nout + tid;
        a.psum[oi] = ss;
        a.psq[oi] = qq;
    }
}

// tiles per chunk of the one-pass tail kernel (the layout of its BatchNorm partial sums)
int conv_tail_tiles(const Level &l)
{
    return ((l.D + TL_TZ - 1) / TL_TZ) * ((l.H + TL_TY - 1) / TL_TY) * ((l.W + TL_TX - 1) / TL_TX);
}

// LDS bytes of the kernel for `cin` input channels in precision mode pm; 0 if the layer does not fit
size_t conv_tail_lds(int cin, int pm)
{
    (void)cin;
    const size_t stage = (size_t)((pm == 0 ? 2 : 1) * TL_SLOTS + 4 * TL_BH) * 16 + (5 * 64 + 2 * TL_SLOTS) * sizeof(float);   // + up weights, slot tables
    const size_t epi = (size_t)(TL_ROWS * TL_PSTRIDE + 2 * TL_THREADS) * sizeof(float);
    const size_t lds = stage > epi ? stage : epi;
    return lds <= 160 * 1024 ? lds : 0;
}

int launch_conv_tail(isx_ctx *ctx, const Level &l, int B, Src sa, Src sb, const void *w_fold, const float *bias, int nout,
                     float *out, float *psum, float *psq, int pm, int *tiles_used, const H2Extra *ex)
{
    const int cin = sa.C + sb.C;
    ISX_REQUIRE(nout >= 1 && nout <= 5 && sa.C % 8 == 0 && sb.C % 8 == 0 && cin >= 8, "internal: tail convolution %d -> %d", cin, nout);
    ISX_REQUIRE(!(sa.planar && sa.C % 8) && !(sb.planar && sb.C % 8), "internal: planar source with %d / %d channels", sa.C, sb.C);
    const size_t lds = conv_tail_lds(cin, pm);
    ISX_REQUIRE(lds, "internal: tail convolution with %d input channels does not fit LDS", cin);
    TailArgs a;
    a.srcA = sa.p; a.scaleA = sa.scale; a.shiftA = sa.shift; a.CA = sa.C; a.planarA = sa.planar ? 1 : 0;
    a.srcB = sb.p; a.scaleB = sb.scale; a.shiftB = sb.shift; a.CB = sb.C; a.planarB = sb.planar ? 1 : 0;
    a.w = (const uint4 *)w_fold; a.bias = bias; a.out = out; a.psum = psum; a.psq = psq;
    a.D = l.D; a.H = l.H; a.W = l.W; a.nout = nout;
    a.nzt = (l.D + TL_TZ - 1) / TL_TZ; a.nyt = (l.H + TL_TY - 1) / TL_TY; a.nxt = (l.W + TL_TX - 1) / TL_TX;
    a.xcd_remap = ctx->tune.xcd_remap;
    const bool up = ex && ex->upA;
    a.uH = up ? ex->uH : 0; a.uW = up ? ex->uW : 0; a.uoff = up ? ex->uoff : 0; a.upw = up ? ex->upw : nullptr; a.upb = up ? ex->upb : nullptr;
    ISX_REQUIRE(!up || (a.upw && a.upb && sa.C <= 64), "internal: up-sampling on read needs the transposed convolution's weights (<= 64 channels)");
    const int tiles = a.nzt * a.nyt * a.nxt;
    if (tiles_used) *tiles_used = tiles;
    const unsigned grid = (unsigned)(B * tiles);
    if (!ctx->tail_attr_done) {
        ISX_HIP(hipFuncSetAttribute((const void *)conv_tail_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ISX_HIP(hipFuncSetAttribute((const void *)conv_tail_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ISX_HIP(hipFuncSetAttribute((const void *)conv_tail_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ISX_HIP(hipFuncSetAttribute((const void *)conv_tail_kernel<0, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ISX_HIP(hipFuncSetAttribute((const void *)conv_tail_kernel<1, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ISX_HIP(hipFuncSetAttribute((const void *)conv_tail_kernel<2, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ctx->tail_attr_done = true;
    }
#ifdef ISX_TAIL_DEV
    const int abl = getenv("ISX_TAIL_ABLATE") ? atoi(getenv("ISX_TAIL_ABLATE")) : 0;
    if (pm == 0 && abl == 1) conv_tail_kernel<0, 1><<<grid, TL_THREADS, lds, ctx->stream>>>(a);
    else if (pm == 0 && abl == 2) conv_tail_kernel<0, 2><<<grid, TL_THREADS, lds, ctx->stream>>>(a);
    else if (pm == 0 && abl == 3) conv_tail_kernel<0, 3><<<grid, TL_THREADS, lds, ctx->stream>>>(a);
    else if (pm == 0 && abl == 4) conv_tail_kernel<0, 4><<<grid, TL_THREADS, lds, ctx->stream>>>(a);
    else
#endif
    if (up) {
        if (pm == 0) conv_tail_kernel<0, 0, true><<<grid, TL_THREADS, lds, ctx->stream>>>(a);
        else if (pm == 1) conv_tail_kernel<1, 0, true><<<grid, TL_THREADS, lds, ctx->stream>>>(a);
        else conv_tail_kernel<2, 0, true><<<grid, TL_THREADS, lds, ctx->stream>>>(a);
    } else
    if (pm == 0) conv_tail_kernel<0><<<grid, TL_THREADS, lds, ctx->stream>>>(a);
    else if (pm == 1) conv_tail_kernel<1><<<grid, TL_THREADS, lds, ctx->stream>>>(a);
    else conv_tail_kernel<2><<<grid, TL_THREADS, lds, ctx->stream>>>(a);
    ISX_CHECK_LAUNCH();
    return ISX_OK;
}

}  // namespace isx
