// unet_state.h -- state shared by the U-Net forward (unet.hip) and the training path (train.hip).
#pragma once

#include <map>
#include <string>
#include <vector>

#include "isx_common.h"

namespace isx {

static constexpr float BN_EPS = 1e-5f;
static constexpr float BN_MOMENTUM = 0.1f;

struct HostTensor {
    std::vector<float> f;
    std::vector<int64_t> shape;
    int64_t i64 = 0;
    bool is_i64 = false;
};

struct ConvBN {                 // one Conv3d followed by its BatchNorm3d
    std::string conv, bn;       // state-dict prefixes, e.g. "c0.conv0", "c0.batch0"
    int cin = 0, cout = 0;
    float *w = nullptr;         // packed weights (layout depends on the kernel that reads them)
    void *w_h2_64 = nullptr, *w_h2_32 = nullptr;   // split-f16 packs for 64- / 32-column tiles (conv_h2.hip); their hi halves serve the plain-f16 mode
    void *w_bf_64 = nullptr, *w_bf_32 = nullptr;   // bf16 packs, same layout (hi half bf16, lo unused)
    float *bias = nullptr, *gamma = nullptr, *beta = nullptr, *rmean = nullptr, *rvar = nullptr;
    int64_t nbt = 0;
    float *scale = nullptr, *shift = nullptr;       // [B][cout], valid after bn_finalize
    float *bmean = nullptr, *bvar = nullptr;        // [B][cout] batch mean / unbiased var
    float *invstd = nullptr;                        // [B][cout] 1/sqrt(biased var + eps) (backward)
    float *psum = nullptr, *psq = nullptr;          // partial sums: [B][cout][tiles] from conv_first_kernel and the conv_h2 kernels (tile fastest: bn_finalize_kernel reads them coalesced), [B][tiles][cout] from the others
    int tiles = 0;
    float gmax = 0.f, bmax = 0.f;                   // max |gamma|, max |beta|: |BN output| <= gmax * sqrt(M - 1) + bmax
};

struct UpConv {
    std::string name;
    int C = 0, kz = 1;
    float *w = nullptr;         // [kz*4][C]
    float *bias = nullptr;
    float wmax = 0.f, bmax = 0.f;
};

struct Level { int D = 0, H = 0, W = 0; int64_t M = 0; };

// Layer numbering.  0..9: encoder (c0.conv0 .. c4.conv1); 10..17: decoder of fork 0 (c5_0.conv0 .. c8_0.conv1); 18..25: decoder
// of fork 1 (c5_1 .. c8_1) when the network is forked (reference: unet.py:152-157,192-210,313-325 -- the forks share
// the encoder and the transposed convolutions, each has its own c5..c8; the reference's decoder() handles i = 0, 1 only).
static constexpr int ISX_MAX_LAYERS = 26, ISX_MAX_MODULES = 13, ISX_MAX_UPS = 7, ISX_MAX_FORKS = 2;
inline int layer_module(int li) { return li < 18 ? li / 2 : 5 + (li - 18) / 2; }       // 0..8: c0..c8 (table index of MOD_*)
inline int dec_layer(int fork, int k, int j) { return (fork == 0 ? 10 : 18) + 2 * k + j; }     // k = 0..3: c5..c8
inline int ups_index(int fork, int k) { return (k == 0 || fork == 0) ? k : 3 + k; }        // up0's output is shared by the forks

struct UNetState {
    isx_tune tune;                   // copy of the owning context's knobs (isx_ctx_create)
    std::map<std::string, HostTensor> host;
    bool committed = false;
    int out_channels = 5;            // all forks together (the channel count of the output tensor)
    int nforks = 1, fork_channels[ISX_MAX_FORKS] = {5, 0};
    int final_act[ISX_MAX_FORKS] = {0, 0};       // ISX_FINAL_* per decoder (isx_unet_set_final_activations)
    int nlayers = 18;                // 18, or 26 with two forks
    int batch = 4;
    int precision = ISX_PRECISION_SPLIT_F16;     // arithmetic of the wide inference convolutions (isx_unet_set_precision)
    // layers: index 0..17 = c0.conv0, c0.conv1, c1.conv0, ... c4.conv1, c5.conv0 ... c8.conv1
    ConvBN cv[ISX_MAX_LAYERS];
    UpConv up[4];
    // workspace for (cz, cy, cx) x batch
    int wz = 0, wy = 0, wx = 0, wbatch = 0;
    Level lv[5];
    std::vector<void *> blocks;      // workspace (re-allocated when the chunk geometry changes)
    std::vector<void *> wblocks;     // weights
    float *x0 = nullptr;
    float *raw[ISX_MAX_LAYERS] = {nullptr};
    float *pool[4] = {nullptr};
    float *ups[ISX_MAX_UPS] = {nullptr};      // ups_index(fork, k)
    float *ptail = nullptr;           // [B][M0][16] partial sums of the folded tail layer
    float *wtail = nullptr, *zeros32 = nullptr;
    const int32_t *d_table = nullptr;   // device rows of the launch set in flight
    int last_n = 0;
    bool up_folded[ISX_MAX_UPS] = {false};     // last forward: ups[i] was NOT materialised (the tail kernel up-sampled it on read)
    bool pool_planar[4] = {false}, ups_planar[ISX_MAX_UPS] = {false};      // last forward: pool[i] / ups[i] are stored planar
    std::vector<int> bn_pending;      // layers of the forward pass in progress whose running statistics are updated at its end (bn_running_all_kernel)
    bool pool_presplit[4] = {false};  // last forward: pool[i] holds its values already split for the split-f16 kernels ([8 hi halves][8 residual halves] per voxel and plane)
    bool planar[ISX_MAX_LAYERS] = {false};   // last forward: raw[li] is stored planar (first convolution of a module whose only reader is the second)
    int arith[ISX_MAX_LAYERS] = {0};  // how each convolution ran in the last forward (ISX_ARITH_*, isx_unet_precision_report)
    struct TrainState *train = nullptr;
};


struct TrainState;
// train.hip: on a context with a bound trainer the bf16 weight packs are derived from the trainer's fp32 master weights
// and only refreshed while the precision is BF16; every bf16 forward (training or inference) calls this first so that a
// precision switch after fp32 steps never convolves with stale or zero packs.
int train_ensure_bf16_packs(isx_ctx *ctx);

struct Src { const float *p; const float *scale; const float *shift; int C; float bound = 0.f; bool planar = false; bool presplit = false; };   // bound: max |activation| (split-f16 range check)

// implicit-GEMM conv on the fp32 matrix cores (unet.hip); see ConvArgs there
struct ConvArgs {
    const float *srcA, *scaleA, *shiftA;   // channels [0, CA): tensor [B][M][CA] (+ optional BN+ReLU)
    const float *srcB, *scaleB, *shiftB;   // channels [CA, CA+CB): the skip tensor of the decoder
    int CA, CB;
    const float *w;        // packed [Cout/BN][Cin/4][taps][4][BN]
    const float *bias;     // [Cout]
    float *out;            // [B][M][store_cols] raw conv output (+bias)
    float *psum, *psq;     // [B][Cout][tiles] (conv_h2 kernels: tile fastest) / [B][tiles][Cout] (conv_mfma_kernel)
    int D, H, W, Cout;
    int mode;              // 0: 2x8x(8 MT) voxel tiles; 1: 2 planes x 64 MT row-wrapped positions; 2: 1 plane x 128 MT
    int tiles_per_chunk, nty, ntx, ntp;
    int slots_max;         // LDS halo slots reserved
    int store_cols;        // columns of the N tile actually stored (Cout, or 16 for the folded tail layer)
    int no_stats;          // 1: skip BN partial sums
    int xcd_remap;         // 1: XCD-contiguous tile order (see the kernel)
    int ntiles;            // Cout / BN (the grid is 1-D: [chunk][tile][ntile])
    int nt_out;            // 1: streaming stores for the output tensor
    // planar tensors: [chunk][C / 8][voxel][8 channels] instead of [chunk][voxel][C] -- a K stage's 32 bytes of consecutive voxels
    // are contiguous, so a halo load instruction reads whole lines instead of 64 pieces of 16 bytes (UNetState::planar)
    int planarA, planarB, out_planar;
    int presplitA;         // source A (planar, no BatchNorm) holds split halves already: its staging is a copy (pool_planar_kernel, UNetState::pool_presplit)
    // conv_h2 kernels, mode 0: box tiles.  A plane pair is covered by nI interior boxes of 8 x TXW voxels (ntx per row) plus,
    // where H or W leave a remainder the launcher found cheaper to cover this way than by padded interior boxes, sr_n boxes of
    // sr_ty x sr_tx voxels down the right edge (x0 = sr_x0, all y) and sb_n boxes of sb_ty x sb_tx along the bottom edge
    // (y0 = sb_y0, x < sr_x0): at 129^2 a padded tiling spends 17.7 % of its matrix work on the "+1" row and column
    int nI, sr_n, sr_ty, sr_tx, sr_x0, sb_n, sb_ty, sb_tx, sb_y0;
    // conv_h2 kernels, modes 1 / 2 (linearised planes): a tile is tz planes x (rows / tz) positions.  2 / 1 by default; a level with
    // few, small planes takes all of them (17 x 17 x 5: 6 tiles of 5 x 51 rows instead of 10 of 1 x 256 -- 289 positions fill 1.13 tiles)
    int tz;
    // conv_h2 kernels with UP: source A is up-sampled while it is staged (H2Extra): coarse plane uH x uW, crop offset, the transposed
    // convolution's weights [4 parities][CA] and bias [CA]
    int upA, uH, uW, uoff, ukz;            // ukz: the transposed convolution's kernel depth (1; 2 for up0, which up-samples z as well)
    const float *upw, *upb;                // [4 ukz parities][CA], [CA]
};

struct H2Extra {           // the tail kernel's optional source: up3 up-sampled on read (conv_tail.hip, TailArgs)
    int upA = 0, uH = 0, uW = 0, uoff = 0, ukz = 1;
    const float *upw = nullptr, *upb = nullptr;
};

extern const char *MODS[9];
extern const int MOD_CIN[9], MOD_COUT[9], MOD_LEVEL[9], UP_C[4], UP_KZ[4];
extern const char *UPS[4];

bool use_mfma(int cin, int cout);
bool use_fold(int cin, int cout);
int conv_mode(const Level &l);
int conv_tiles(const Level &l, int mode, int mt, int *nty, int *ntx, int *ntp, int *slots);
// raw launch of the MFMA conv (no BN finalize): used by the forward and by dgrad
int launch_conv_mfma(isx_ctx *ctx, const Level &l, int B, Src sa, Src sb, const float *w_packed, const float *bias,
                     int cout, float *out, float *psum, float *psq, int no_stats);
// split-f16 convolution (conv_h2.hip)
int conv_h2_bn(const isx_tune &tune, int slots, int cout);
bool conv_h2_nw8(const isx_tune &tune, const Level &l, int cout, int batch);
int h2_mt(const isx_tune &tune, const Level &l, int batch);
void pack_h2_weights(const float *w, int cout, int cin, int bn, std::vector<uint16_t> &out, int pm);
void pack_h2_fold_weights(const float *w, int cout, int cin, std::vector<uint16_t> &out, int pm);
int repack_bf16(isx_ctx *ctx, const float *wref, int cout, int cin, int transpose, int bn, int cout_p, int cin_p, void *dst);
size_t h2_pack_halves(int cout_p, int cin_p, int bn);
struct H2Extra;
int launch_conv_h2(isx_ctx *ctx, const Level &l, int B, Src sa, Src sb, const void *w_h2, const float *bias, int cout,
                   float *out, float *psum, float *psq, int no_stats, int xcd_remap, int *tiles_used, int pm = 0, int out_planar = 0,
                   const H2Extra *ex = nullptr);
bool conv_h2_can_up(isx_ctx *ctx, const Level &l, int B, int cout, int pm, int ca, int ukz);
// one-pass thin tail convolution (conv_tail.hip)
int conv_tail_tiles(const Level &l);
size_t conv_tail_lds(int cin, int pm);
int launch_conv_tail(isx_ctx *ctx, const Level &l, int B, Src sa, Src sb, const void *w_fold, const float *bias, int nout,
                     float *out, float *psum, float *psq, int pm, int *tiles_used, const H2Extra *ex = nullptr);
// the 5 -> 5 convolution on the vector ALU (conv55.hip: compiled without SLP vectorisation)
int launch_conv55(isx_ctx *ctx, const Level &l, int B, Src sa, const float *w, const float *bias, float *out, float *psum, float *psq,
                  int tiles);
int prepare_workspace(isx_ctx *ctx, int cz, int cy, int cx, int batch);
int forward_batch(isx_ctx *ctx, int B, float *out_ncdhw, float *fm, const int64_t shape[3]);
int dev_alloc(UNetState *u, size_t floats, float **out);
void train_destroy(TrainState *t);

// ---- final activation of a decoder (ConvModule.forward, unet.py:91-106; codes ISX_FINAL_*) ---------------------------------
// v[0, C): the BatchNorm outputs of one voxel -> the activations, in place.  C <= 8.
__device__ __forceinline__ void final_act_apply(float (&v)[8], int C, int act)
{
    if (act == ISX_FINAL_SOFTMAX) {
        // nn.Softmax() with dim=None: dim 1 of the (N, C, D, H, W) tensor (torch.nn.functional._get_softmax_dim)
        float mx = v[0], s = 0.f;
#pragma unroll
        for (int c = 1; c < 8; ++c) if (c < C) mx = fmaxf(mx, v[c]);
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c < C) { v[c] = expf(v[c] - mx); s += v[c]; }
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c < C) v[c] = v[c] / s;
        return;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) if (c < C) {
        const float z = v[c];
        v[c] = act == ISX_FINAL_SIGMOID ? 1.f / (1.f + expf(-z)) : act == ISX_FINAL_RELU ? fmaxf(z, 0.f) : act == ISX_FINAL_TANH ? tanhf(z) : z;
    }
}
// g[0, C): dL/dy of one voxel -> dL/dz, in place; y: the activations final_act_apply produced
__device__ __forceinline__ void final_act_grad(float (&g)[8], const float (&y)[8], int C, int act)
{
    float dot = 0.f;
    if (act == ISX_FINAL_SOFTMAX) {
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c < C) dot += g[c] * y[c];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) if (c < C) {
        const float yy = y[c];
        g[c] = act == ISX_FINAL_SIGMOID ? g[c] * (1.f - yy) * yy : act == ISX_FINAL_RELU ? (yy > 0.f ? g[c] : 0.f)
             : act == ISX_FINAL_TANH ? g[c] * (1.f - yy * yy) : act == ISX_FINAL_SOFTMAX ? yy * (g[c] - dot) : g[c];
    }
}

}  // namespace isx
