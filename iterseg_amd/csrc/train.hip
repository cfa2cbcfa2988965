// train.hip -- backward pass, losses and Adam for the affinity U-Net on gfx950 (MI355X).
//
// Replaces, for one (1,1,D,H,W) patch per step (the reference trains with batch size 1),
// _train_step / the duplicated optimiser.step() of src/iterseg/train.py:344-353 and :310, nn.BCELoss /
// nn.MSELoss / WeightedBCELoss (train.py:197-217, custom_loss.py:56-84,161-185) and torch.optim.Adam
// (train.py:145).  torch.autograd is not used: the forward of unet.hip keeps every raw convolution
// output of the patch, and this file walks the graph backwards with hand-written kernels:
//
//   * BatchNorm(train) + ReLU/sigmoid backward: two passes per layer (deterministic block partial sums of
//     dz and dz*xhat, then d(raw) = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat))); the second pass
//     also emits the bias gradient partials;
//   * dgrad: the forward implicit-GEMM MFMA kernel run on d(raw) with channel-transposed, tap-flipped
//     weights (a 3x3x3 correlation backwards is the same correlation with mirrored taps);
//   * wgrad: an MFMA GEMM with the voxels as the K dimension -- dW[tap] (Cout x Cin) = dY^T . X_shifted --
//     64x64 tiles, one tap and one voxel slab per work-group, split-K partials reduced in a fixed order;
//     the operand X gets the producer's BN+ReLU and the decoder concat on load, like the forward;
//   * max-pool / depthwise transposed-conv backward, loss + d(loss), Adam on flat buffers.
//
// Parameters, gradients and Adam moments live in four flat float buffers OWNED BY THE CALLER (PyTorch
// tensors), in state-dict order of the trainable keys, so the data-parallel gradient all-reduce is one
// RCCL call on the gradient buffer.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "unet_state.h"

namespace isx {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct TrainState {
    int cz = 0, cy = 0, cx = 0;
    float *P = nullptr, *G = nullptr, *Mo = nullptr, *Vo = nullptr;
    int64_t nparam = 0, adam_t = 0;
    int64_t off_w[ISX_MAX_LAYERS], off_b[ISX_MAX_LAYERS], off_gamma[ISX_MAX_LAYERS], off_beta[ISX_MAX_LAYERS], off_upw[4], off_upb[4];
    int dstride[ISX_MAX_LAYERS];                 // channel stride of draw[li] (cout, or 8 for a thin layer whose dgrad runs on MFMA)
    std::vector<void *> blocks;
    float *draw[ISX_MAX_LAYERS] = {nullptr}, *gact[ISX_MAX_LAYERS] = {nullptr}, *w_dgrad[ISX_MAX_LAYERS] = {nullptr};
    // bf16 mixed precision (ISX_PRECISION_BF16): device-refreshed packs of the forward (both tile widths) and dgrad operators
    void *wbf_fwd64[ISX_MAX_LAYERS] = {nullptr}, *wbf_fwd32[ISX_MAX_LAYERS] = {nullptr}, *wbf_dg[ISX_MAX_LAYERS] = {nullptr};
    int wbf_dg_bn[ISX_MAX_LAYERS] = {0};
    bool bf_fresh = false;           // the bf16 packs match P
    float *dcat[ISX_MAX_FORKS][4] = {{nullptr}}, *dpool[4] = {nullptr};      // dcat[fork][k]: gradient of cat(up_k, skip) of that decoder
    float *part = nullptr, *coef = nullptr, *wpart = nullptr, *lossbuf = nullptr, *zeros = nullptr;
    size_t part_floats = 0, wpart_floats = 0;
};

void train_destroy(TrainState *t)
{
    if (!t) return;
    for (void *p : t->blocks) (void)hipFree(p);
    delete t;
}

static int talloc(TrainState *t, size_t floats, float **out)
{
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, (floats ? floats : 1) * sizeof(float));
    if (e != hipSuccess) { set_error("hipMalloc(%zu floats) failed: %s", floats, hipGetErrorString(e)); return ISX_ERR_NOMEM; }
    t->blocks.push_back(p);
    *out = (float *)p;
    return ISX_OK;
}

// ------------------------------------------------------------------------------------------------
// weight repacking on the device (after every optimiser step): reference layouts in P -> kernel layouts
// ------------------------------------------------------------------------------------------------
// forward MFMA layout [Cout/BN][Cin/4][27][4][BN]; transpose = 1 builds the dgrad weights: the roles of the
// channel axes are swapped and the taps mirrored (t -> 26 - t), i.e. w'[ci][co][t] = w[co][ci][26 - t].
__global__ void repack_mfma_kernel(const float *__restrict__ wref, int cout, int cin, int transpose, float *__restrict__ pk,
                                   int ci_pad)
{
    // ci_pad > 0: the packed conv has ci_pad input channels of which only the first CI are real (zero rows after)
    const int CO = transpose ? cin : cout, CI = transpose ? cout : cin;     // shapes of the conv being packed
    const int BN = (CO % 64 == 0) ? 64 : 32;
    const int nsteps = (ci_pad > 0 ? ci_pad : CI) / 4;
    const int64_t n = (int64_t)CO * CI * 27;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int t = (int)(i % 27);
        int ci = (int)((i / 27) % CI), co = (int)(i / (27 * (int64_t)CI));
        float v = transpose ? wref[((int64_t)ci * cin + co) * 27 + (26 - t)] : wref[((int64_t)co * cin + ci) * 27 + t];
        int64_t dst = ((((int64_t)(co / BN) * nsteps + ci / 4) * 27 + t) * 4 + ci % 4) * BN + co % BN;
        pk[dst] = v;
    }
}

// both packs of a wide layer from one read of the reference weights: the forward one and the dgrad one (above, transpose = 1)
__global__ void repack_mfma_both_kernel(const float *__restrict__ wref, int cout, int cin, float *__restrict__ pk_fwd,
                                        float *__restrict__ pk_dg)
{
    const int BNF = (cout % 64 == 0) ? 64 : 32, BND = (cin % 64 == 0) ? 64 : 32;
    const int nf = cin / 4, nd = cout / 4;
    const int64_t n = (int64_t)cout * cin * 27;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % 27), ci = (int)((i / 27) % cin), co = (int)(i / (27 * (int64_t)cin));
        const float v = wref[i];
        pk_fwd[((((int64_t)(co / BNF) * nf + ci / 4) * 27 + t) * 4 + ci % 4) * BNF + co % BNF] = v;
        pk_dg[((((int64_t)(ci / BND) * nd + co / 4) * 27 + (26 - t)) * 4 + co % 4) * BND + ci % BND] = v;      // w'[ci][co][26 - t]
    }
}

__global__ void repack_fold_kernel(const float *__restrict__ wref, int cout, int cin, float *__restrict__ pk)
{
    const int64_t n = (int64_t)cout * cin * 27;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int t = (int)(i % 27), ci = (int)((i / 27) % cin), co = (int)(i / (27 * (int64_t)cin));
        int t9 = t / 3, dxi = t % 3;
        pk[((((int64_t)(ci / 4)) * 9 + t9) * 4 + ci % 4) * 32 + dxi * cout + co] = wref[i];
    }
}

// conv_first [27][32] (cin == 1) and conv_thin [27][Cin][8]
__global__ void repack_direct_kernel(const float *__restrict__ wref, int cout, int cin, int first, float *__restrict__ pk)
{
    const int64_t n = (int64_t)cout * cin * 27;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int t = (int)(i % 27), ci = (int)((i / 27) % cin), co = (int)(i / (27 * (int64_t)cin));
        if (first) pk[(int64_t)t * cout + co] = wref[i];
        else pk[((int64_t)t * cin + ci) * 8 + co] = wref[i];
    }
}

__global__ void repack_up_kernel(const float *__restrict__ wref, int C, int kt, float *__restrict__ pk)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * kt) return;
    int t = i % kt, c = i / kt;
    pk[(int64_t)t * C + c] = wref[i];
}

// ------------------------------------------------------------------------------------------------
// loss + d(loss)/d(y_hat).  kind 0: BCE (mean over all elements, logs clamped at -100 like torch);
// 1: MSE; 2: weighted BCE (per-channel mean * weight, mean over channels, custom_loss.py:161-185).
// y_hat is recomputed from raw17 (act(BN(raw)), act = the decoder's final activation, ISX_FINAL_*); target is (C, D, H, W)
// planar.  lossbuf partials: [blocks][C] ; gout: [M][C] channels-last -- d(loss)/d(y_hat) for the default sigmoid (its
// derivative is fused into the BatchNorm backward, act_grad), d(loss)/d(BN output) for every other activation.
// ------------------------------------------------------------------------------------------------
struct LossSpec {          // loss = sum_c w[c] * sum_m elem_loss_kind[c](yh[m,c], y[c,m])
    int kind[8];           // 0 BCE (log clamped at -100 like torch), 1 squared error
    float w[8];            // includes every mean: e.g. 1/(M*C) for nn.BCELoss
};

__global__ __launch_bounds__(256) void loss_kernel(const float *__restrict__ raw, const float *__restrict__ scale,
                                                   const float *__restrict__ shift, const float *__restrict__ target,
                                                   int C, int64_t M, const LossSpec spec, int act,
                                                   float *__restrict__ gout, float *__restrict__ part, int *__restrict__ bad)
{
    __shared__ float red[4][8];
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float l[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) l[c] = 0.f;
    if (m < M && act != ISX_FINAL_SIGMOID) {
        float yv[8], gv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) yv[c] = c < C ? fmaf(raw[m * C + c], scale[c], shift[c]) : 0.f;
        final_act_apply(yv, C, act);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            gv[c] = 0.f;
            if (c < C) {
                const float yh = yv[c], y = target[(int64_t)c * M + m];
                if (spec.kind[c] == 1) {
                    const float d = yh - y;
                    l[c] = d * d;
                    gv[c] = 2.f * d * spec.w[c];
                } else {
                    if (!(yh >= 0.f && yh <= 1.f)) *bad = 1;            // nn.BCELoss refuses such an input (relu / tanh / no activation)
                    const float lg = fmaxf(logf(yh), -100.f), lg1 = fmaxf(logf(1.f - yh), -100.f);
                    l[c] = -(y * lg + (1.f - y) * lg1);
                    gv[c] = spec.w[c] * (yh - y) / fmaxf((1.f - yh) * yh, 1e-12f);
                }
            }
        }
        final_act_grad(gv, yv, C, act);
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c < C) gout[m * C + c] = gv[c];
    } else if (m < M)
        for (int c = 0; c < C; ++c) {
            float z = fmaf(raw[m * C + c], scale[c], shift[c]);
            float yh = 1.f / (1.f + expf(-z));
            float y = target[(int64_t)c * M + m];
            float g;
            if (spec.kind[c] == 1) {
                float d = yh - y;
                l[c] = d * d;
                g = 2.f * d * spec.w[c];
            } else {
                float lg = fmaxf(logf(yh), -100.f), lg1 = fmaxf(logf(1.f - yh), -100.f);
                l[c] = -(y * lg + (1.f - y) * lg1);
                float denom = fmaxf((1.f - yh) * yh, 1e-12f);
                g = spec.w[c] * (yh - y) / denom;
            }
            gout[m * C + c] = g;
        }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c = 0; c < C; ++c) {
        float v = l[c];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wave][c] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < C) part[(int64_t)blockIdx.x * C + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// DiceLoss (custom_loss.py:24-53), pass 1: per-block partials of (sum yh*y, sum yh, sum y) per channel ->
// part[blocks][3C].  H*W is a multiple of 256 (chunk y, x are multiples of 16), so a block never straddles
// two z planes and the per-plane sums the reference's per-channel log needs fall out of the same partials.
__global__ __launch_bounds__(256) void dice_sums_kernel(const float *__restrict__ raw, const float *__restrict__ scale,
                                                        const float *__restrict__ shift, const float *__restrict__ target,
                                                        int C, int64_t M, int act, float *__restrict__ part)
{
    __shared__ float red[4][24];
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float yv[8];
    if (act != ISX_FINAL_SIGMOID) {
#pragma unroll
        for (int c = 0; c < 8; ++c) yv[c] = (c < C && m < M) ? fmaf(raw[m * C + c], scale[c], shift[c]) : 0.f;
        final_act_apply(yv, C, act);
    }
    for (int c = 0; c < C; ++c) {
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (m < M) {
            float yh;
            if (act == ISX_FINAL_SIGMOID) {
                float z = fmaf(raw[m * C + c], scale[c], shift[c]);
                yh = 1.f / (1.f + expf(-z));
            } else {
                yh = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k == c) yh = yv[k];
            }
            float y = target[(int64_t)c * M + m];
            v0 = yh * y; v1 = yh; v2 = y;
        }
        for (int o = 32; o > 0; o >>= 1) { v0 += __shfl_xor(v0, o); v1 += __shfl_xor(v1, o); v2 += __shfl_xor(v2, o); }
        if (lane == 0) { red[wave][3 * c] = v0; red[wave][3 * c + 1] = v1; red[wave][3 * c + 2] = v2; }
    }
    __syncthreads();
    if ((int)threadIdx.x < 3 * C) part[(int64_t)blockIdx.x * 3 * C + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// part[blocks][K] -> out[group][K]: fixed-order double sums over `per` consecutive blocks per group
__global__ void colsum_group_kernel(const float *__restrict__ part, int per, int K, float *__restrict__ out)
{
    __shared__ double red[256];
    const int k = blockIdx.x, grp = blockIdx.y;
    double s = 0.0;
    for (int b = threadIdx.x; b < per; b += 256) s += (double)part[((int64_t)grp * per + b) * K + k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(int64_t)grp * K + k] = (float)red[0];
}

struct DiceCoef { float ka[8], kb[8]; };
// DiceLoss, pass 2: d loss / d yh[m, c] = y * ka[c] + kb[c]; with a final activation other than the sigmoid the kernel goes on
// to d loss / d (BN output) like loss_kernel does
__global__ void dice_grad_kernel(const float *__restrict__ target, int C, int64_t M, const DiceCoef k,
                                 const float *__restrict__ raw, const float *__restrict__ scale, const float *__restrict__ shift, int act,
                                 float *__restrict__ gout)
{
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    if (act == ISX_FINAL_SIGMOID) {
        for (int c = 0; c < C; ++c) gout[m * C + c] = fmaf(target[(int64_t)c * M + m], k.ka[c], k.kb[c]);
        return;
    }
    float yv[8], gv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        yv[c] = c < C ? fmaf(raw[m * C + c], scale[c], shift[c]) : 0.f;
        gv[c] = c < C ? fmaf(target[(int64_t)c * M + m], k.ka[c], k.kb[c]) : 0.f;
    }
    final_act_apply(yv, C, act);
    final_act_grad(gv, yv, C, act);
#pragma unroll
    for (int c = 0; c < 8; ++c) if (c < C) gout[m * C + c] = gv[c];
}

// sums the [blocks][C] partials in double, fixed order: out[c] = per-channel sum
__global__ void colsum_kernel(const float *__restrict__ part, int blocks, int C, float *__restrict__ out)
{
    __shared__ double red[256];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < blocks; b += 256) s += (double)part[(int64_t)b * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = (float)red[0];
}

// ------------------------------------------------------------------------------------------------
// BN(train) + activation backward.  act: 0 relu, 1 sigmoid, 2 none (g already is the gradient at the BatchNorm output).  g may be strided (a channel slice of a wider
// tensor): element (m, c) at g[m * gstride + c].
// pass 1: partials of sum(dz), sum(dz * xhat) per channel -> part[blocks][2C]
// pass 2: draw = gamma*invstd*(dz - c1 - xhat*c2); partials of sum(draw) -> part[blocks][C]
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_grad(float g, float z, int act)
{
    if (act == 0) return z > 0.f ? g : 0.f;
    if (act == 2) return g;
    float s = 1.f / (1.f + expf(-z));
    return g * (1.f - s) * s;
}

// The two BatchNorm-backward passes with four channels per lane (16-byte loads and stores): C % 4 == 0 and 16-byte aligned rows.
// The one-channel-per-lane kernels below them moved 4 bytes per lane and load and ran at a quarter of the HBM rate (round 5:
// 1.5 ms of the 9.7 ms bf16 step for 2.1 GB).  Per channel the rows are summed in another grouping than in the scalar kernels
// (a lane owns rows m = rslot (mod nslot) with nslot = 256 / (C / 4)); the sums stay in double and in a fixed order.
__global__ __launch_bounds__(256) void bn_bwd_reduce4_kernel(const float *__restrict__ g, int gstride,
                                                             const float *__restrict__ raw, const float *__restrict__ scale,
                                                             const float *__restrict__ shift, const float *__restrict__ mean,
                                                             const float *__restrict__ invstd, int C, int64_t M, int act,
                                                             int rows_per_block, float *__restrict__ part)
{
    extern __shared__ double smd[];                     // [256][4][2]
    const int L4 = C >> 2, l4 = threadIdx.x % L4, c = 4 * l4;
    const int rslot = threadIdx.x / L4, nslot = 256 / L4;
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    if (rslot < nslot) {
        const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
        int64_t r1 = r0 + rows_per_block;
        if (r1 > M) r1 = M;
        const float4 sc = *(const float4 *)&scale[c], sh = *(const float4 *)&shift[c], mu = *(const float4 *)&mean[c], is = *(const float4 *)&invstd[c];
        for (int64_t m = r0 + rslot; m < r1; m += 2 * nslot) {
            float4 xv[2], gv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t mm = m + (int64_t)u * nslot;
                xv[u] = mm < r1 ? *(const float4 *)&raw[mm * C + c] : make_float4(0.f, 0.f, 0.f, 0.f);
                gv[u] = mm < r1 ? *(const float4 *)&g[mm * gstride + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (m + (int64_t)u * nslot >= r1) break;
                const float x[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, gg[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
                const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w}, muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float z = fmaf(x[k], scv[k], shv[k]);
                    const float dz = act_grad(gg[k], z, act);
                    s1[k] += (double)dz;
                    s2[k] += (double)dz * (double)((x[k] - muv[k]) * isv[k]);
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { smd[(threadIdx.x * 4 + k) * 2] = s1[k]; smd[(threadIdx.x * 4 + k) * 2 + 1] = s2[k]; }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        const int q = threadIdx.x >> 2, k = threadIdx.x & 3;
        double a = 0.0, b = 0.0;
        for (int r = 0; r < nslot; ++r) { a += smd[((r * L4 + q) * 4 + k) * 2]; b += smd[((r * L4 + q) * 4 + k) * 2 + 1]; }
        part[((int64_t)blockIdx.x * C + threadIdx.x) * 2] = (float)a;
        part[((int64_t)blockIdx.x * C + threadIdx.x) * 2 + 1] = (float)b;
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const float *__restrict__ g, int gstride,
                                                            const float *__restrict__ raw, const float *__restrict__ scale,
                                                            const float *__restrict__ shift, const float *__restrict__ mean,
                                                            const float *__restrict__ invstd, const float *__restrict__ coef,
                                                            int C, int64_t M, int act, int rows_per_block,
                                                            float *__restrict__ draw, int dstride, float *__restrict__ part)
{
    extern __shared__ float sm[];                        // [256][4]
    const int L4 = C >> 2, l4 = threadIdx.x % L4, c = 4 * l4;
    const int rslot = threadIdx.x / L4, nslot = 256 / L4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (rslot < nslot) {
        const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
        int64_t r1 = r0 + rows_per_block;
        if (r1 > M) r1 = M;
        const float4 sc4 = *(const float4 *)&scale[c], sh4 = *(const float4 *)&shift[c], mu4 = *(const float4 *)&mean[c], is4 = *(const float4 *)&invstd[c];
        const float4 ca = *(const float4 *)&coef[2 * c], cb = *(const float4 *)&coef[2 * c + 4];       // {c1, c2} of channels c, c + 1 | c + 2, c + 3
        const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w}, muv[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, isv[4] = {is4.x, is4.y, is4.z, is4.w};
        const float c1v[4] = {ca.x, ca.z, cb.x, cb.z}, c2v[4] = {ca.y, ca.w, cb.y, cb.w};
        for (int64_t m = r0 + rslot; m < r1; m += 2 * nslot) {
            float4 xv[2], gv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t mm = m + (int64_t)u * nslot;
                xv[u] = mm < r1 ? *(const float4 *)&raw[mm * C + c] : make_float4(0.f, 0.f, 0.f, 0.f);
                gv[u] = mm < r1 ? *(const float4 *)&g[mm * gstride + c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t mm = m + (int64_t)u * nslot;
                if (mm >= r1) break;
                const float x[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w}, gg[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
                float d[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float z = fmaf(x[k], scv[k], shv[k]);
                    const float dz = act_grad(gg[k], z, act);
                    const float xh = (x[k] - muv[k]) * isv[k];
                    d[k] = scv[k] * (dz - c1v[k] - xh * c2v[k]);          // sc = gamma * invstd
                    s1[k] += d[k];
                }
                *(float4 *)&draw[mm * dstride + c] = make_float4(d[0], d[1], d[2], d[3]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) sm[threadIdx.x * 4 + k] = s1[k];
    __syncthreads();
    if ((int)threadIdx.x < C) {
        const int q = threadIdx.x >> 2, k = threadIdx.x & 3;
        float a = 0.f;
        for (int r = 0; r < nslot; ++r) a += sm[(r * L4 + q) * 4 + k];
        part[(int64_t)blockIdx.x * C + threadIdx.x] = a;
    }
}


__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float *__restrict__ g, int gstride,
                                                            const float *__restrict__ raw, const float *__restrict__ scale,
                                                            const float *__restrict__ shift, const float *__restrict__ mean,
                                                            const float *__restrict__ invstd, int C, int64_t M, int act,
                                                            int rows_per_block, float *__restrict__ part)
{
    // thread layout: channel = threadIdx.x % C4 groups... generic: each thread walks rows for a fixed channel
    extern __shared__ double smd[];                     // [256][2]; double like ATen's CPU accumulation
    const int nth = 256;
    const int c = threadIdx.x % C;                      // requires C <= 256; threads >= (256/C)*C idle
    const int lanes_per_row = C;
    const int rslot = threadIdx.x / lanes_per_row, nslot = nth / lanes_per_row;
    double s1 = 0.0, s2 = 0.0;
    if (rslot < nslot) {
        const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
        int64_t r1 = r0 + rows_per_block;
        if (r1 > M) r1 = M;
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
        // four rows' loads in flight per thread; the sums keep their order (bit-identical to the one-row loop, which ran at
        // 1.1 TB/s: one dependent pair of 4-byte loads per iteration)
        for (int64_t m = r0 + rslot; m < r1; m += 4 * nslot) {
            float xv[4], gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t mm = m + (int64_t)u * nslot;
                xv[u] = mm < r1 ? raw[mm * C + c] : 0.f;
                gv[u] = mm < r1 ? g[mm * gstride + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (m + (int64_t)u * nslot >= r1) break;
                const float x = xv[u];
                const float z = fmaf(x, sc, sh);
                const float dz = act_grad(gv[u], z, act);
                s1 += (double)dz;
                s2 += (double)dz * (double)((x - mu) * is);
            }
        }
    }
    smd[threadIdx.x * 2] = s1;
    smd[threadIdx.x * 2 + 1] = s2;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < nslot; ++k) { a += smd[(k * lanes_per_row + threadIdx.x) * 2]; b += smd[(k * lanes_per_row + threadIdx.x) * 2 + 1]; }
        part[((int64_t)blockIdx.x * C + threadIdx.x) * 2] = (float)a;
        part[((int64_t)blockIdx.x * C + threadIdx.x) * 2 + 1] = (float)b;
    }
}

// reduce [blocks][C][2] -> dgamma (sum dz*xhat), dbeta (sum dz), coef[c] = {mean(dz), mean(dz*xhat)}
__global__ void bn_bwd_finalize_kernel(const float *__restrict__ part, int blocks, int C, double count,
                                       float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ coef)
{
    __shared__ double r1[256], r2[256];
    const int c = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < blocks; k += 256) { a += (double)part[((int64_t)k * C + c) * 2]; b += (double)part[((int64_t)k * C + c) * 2 + 1]; }
    r1[threadIdx.x] = a;
    r2[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        dbeta[c] = (float)r1[0];
        dgamma[c] = (float)r2[0];
        coef[2 * c] = (float)(r1[0] / count);
        coef[2 * c + 1] = (float)(r2[0] / count);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *__restrict__ g, int gstride,
                                                           const float *__restrict__ raw, const float *__restrict__ scale,
                                                           const float *__restrict__ shift, const float *__restrict__ mean,
                                                           const float *__restrict__ invstd, const float *__restrict__ coef,
                                                           int C, int64_t M, int act, int rows_per_block,
                                                           float *__restrict__ draw, int dstride, float *__restrict__ part)
{
    extern __shared__ float sm[];
    const int c = threadIdx.x % C;
    const int rslot = threadIdx.x / C, nslot = 256 / C;
    float s1 = 0.f;
    if (rslot < nslot) {
        const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
        int64_t r1 = r0 + rows_per_block;
        if (r1 > M) r1 = M;
        const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c], c1 = coef[2 * c], c2 = coef[2 * c + 1];
        for (int64_t m = r0 + rslot; m < r1; m += 4 * nslot) {      // (four rows in flight, sums in the same order: see the reduce kernel)
            float xv[4], gv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t mm = m + (int64_t)u * nslot;
                xv[u] = mm < r1 ? raw[mm * C + c] : 0.f;
                gv[u] = mm < r1 ? g[mm * gstride + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t mm = m + (int64_t)u * nslot;
                if (mm >= r1) break;
                const float x = xv[u];
                const float z = fmaf(x, sc, sh);
                const float dz = act_grad(gv[u], z, act);
                const float xh = (x - mu) * is;
                const float d = sc * (dz - c1 - xh * c2);          // sc = gamma * invstd
                draw[mm * dstride + c] = d;
                s1 += d;
            }
        }
    }
    sm[threadIdx.x] = s1;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float a = 0.f;
        for (int k = 0; k < nslot; ++k) a += sm[k * C + threadIdx.x];
        part[(int64_t)blockIdx.x * C + threadIdx.x] = a;
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad on the fp32 matrix cores.  grid = (co_tiles * ci_tiles, 27 taps, splits); block = 4 waves in a
// 2x2 arrangement over a 64(co) x 64(ci) tile.  K = voxels, staged 64 at a time in LDS:
//   dY tile [64 vox][64 co], X tile [64 vox][64 ci] (X = concat of two sources, BN+ReLU on load,
//   shifted by the tap, zero outside the patch).  D[co][ci] += sum_v dY[v][co] * X[v + tap][ci].
// partial [split][27][CoP][CiP] (CoP/CiP = channel counts padded to 64).
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    const float *dy;                       // [M][dy_stride], first Cout channels
    int dy_stride;
    const float *srcA, *scaleA, *shiftA;   // conv input, channels [0, CA)
    const float *srcB, *scaleB, *shiftB;   // channels [CA, CA+CB)
    int CA, CB, Cout;
    int D, H, W;
    int co_tiles, ci_tiles, splits;
    int64_t vox_per_split;                 // multiple of 64
    float *partial;
};

// TCO x TCI output tile (32 or 64 each) = NSUB 32x32 sub-tiles; with fewer than 4 sub-tiles the waves also split the
// 64 voxels of a K block between them (KG groups) and their accumulators are summed through LDS at the end, so a
// 32-channel layer no longer computes a zero-padded 64x64 tile.  The next block is prefetched into registers
// while the matrix cores work on the current one.
// One work-group computes the THREE x-taps (dx = -1, 0, +1) of one (dz, dy): the x-shift of a tap is a shift by one row of
// the staged input tile (66 rows for 64 voxels), so dY and the input are fetched once for three taps instead of once per tap
// (round 2: the kernel was bound by L2 traffic at 16 FLOP per byte, 0.29 of the fp32 matrix peak).  A voxel at the end of an
// x-row has no +1 neighbour (the next linear voxel belongs to the next row): its dY value is zeroed for that tap.
template <int TCO, int TCI>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const WgradArgs a)
{
    constexpr int NSUB = (TCO / 32) * (TCI / 32), KG = 4 / NSUB, KSTEPS = 32 / KG;
    constexpr int NY = TCO / 16;                             // float4 of dY per thread per block (64 voxels x TCO / 4 / 256)
    constexpr int XROWS = 66, XQ = TCI / 4, NX = (XROWS * XQ + 255) / 256;      // input rows u = v0 - 1 .. v0 + 64
    __shared__ __attribute__((aligned(16))) float Ys[64 * TCO];
    __shared__ __attribute__((aligned(16))) float Xs[XROWS * TCI];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, nn = lane & 31;
    const int co_t = blockIdx.x / a.ci_tiles, ci_t = blockIdx.x % a.ci_tiles;
    const int tap9 = blockIdx.y, split = blockIdx.z;
    const int dz = tap9 / 3 - 1, dy = tap9 % 3 - 1;
    const int D = a.D, H = a.H, W = a.W, Cin = a.CA + a.CB;
    const int64_t M = (int64_t)D * H * W;
    const int64_t v_begin = (int64_t)split * a.vox_per_split;
    int64_t v_end = v_begin + a.vox_per_split;
    if (v_end > M) v_end = M;
    const int sub = wave % NSUB, kg = wave / NSUB;
    const int wr = sub / (TCI / 32), wc = sub % (TCI / 32);
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int co0 = co_t * TCO, ci0 = ci_t * TCI;
    // staging roles: TCO/4 (TCI/4) lanes cover the channels of one voxel with a float4 each
    constexpr int YQ = TCO / 4, YV = 256 / YQ;               // voxels per pass
    const int yq = tid % YQ, yv = tid / YQ;
    float4 ry[NY], rx[NX];

    auto load_block = [&](int64_t v0) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NY; ++p) {
            const int64_t v = v0 + yv + YV * p;
            float4 yvv = make_float4(0.f, 0.f, 0.f, 0.f);
            const int co = co0 + 4 * yq;
            if (v < v_end) {
                if (co + 3 < a.Cout) yvv = *(const float4 *)&a.dy[v * a.dy_stride + co];
                else
                    for (int j = 0; j < 4; ++j)
                        if (co + j < a.Cout) (&yvv.x)[j] = a.dy[v * a.dy_stride + co + j];
            }
            ry[p] = yvv;
        }
#pragma unroll
        for (int p = 0; p < NX; ++p) {
            const int idx = tid + 256 * p;
            const int row = idx / XQ, xq = idx % XQ;
            const int64_t u = v0 - 1 + row;                  // the voxel whose (dz, dy)-shifted input sits in this row
            float4 xvv = make_float4(0.f, 0.f, 0.f, 0.f);
            const int ci = ci0 + 4 * xq;
            if (row < XROWS && u >= 0 && u < M && ci < Cin) {
                // (32-bit: a chunk has far fewer than 2^32 voxels; three 64-bit divisions per loaded row were most of this kernel's vector instructions)
                const uint32_t uu = (uint32_t)u, hw = (uint32_t)W * (uint32_t)H, zq = uu / hw, rem = uu - zq * hw, yq = rem / (uint32_t)W;
                const int x = (int)(rem - yq * (uint32_t)W), y = (int)yq + dy, z = (int)zq + dz;
                if (z >= 0 && z < D && y >= 0 && y < H) {
                    const int64_t g = ((int64_t)z * H + y) * W + x;
                    const bool fromB = ci >= a.CA;
                    const float *src = fromB ? a.srcB : a.srcA, *sc = fromB ? a.scaleB : a.scaleA, *sh = fromB ? a.shiftB : a.shiftA;
                    const int C = fromB ? a.CB : a.CA, cc = fromB ? ci - a.CA : ci;
                    if (cc + 3 < C) {
                        xvv = *(const float4 *)&src[g * C + cc];
                        if (sc) {
                            const float4 s4 = *(const float4 *)&sc[cc], h4 = *(const float4 *)&sh[cc];
                            xvv.x = fmaxf(fmaf(xvv.x, s4.x, h4.x), 0.f);
                            xvv.y = fmaxf(fmaf(xvv.y, s4.y, h4.y), 0.f);
                            xvv.z = fmaxf(fmaf(xvv.z, s4.z, h4.z), 0.f);
                            xvv.w = fmaxf(fmaf(xvv.w, s4.w, h4.w), 0.f);
                        }
                    } else {
                        for (int j = 0; j < 4; ++j)
                            if (cc + j < C) {
                                float t = src[g * C + cc + j];
                                if (sc) t = fmaxf(fmaf(t, sc[cc + j], sh[cc + j]), 0.f);
                                (&xvv.x)[j] = t;
                            }
                    }
                }
            }
            rx[p] = xvv;
        }
    };

    if (v_begin < v_end) load_block(v_begin);
    for (int64_t v0 = v_begin; v0 < v_end; v0 += 64) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NY; ++p) *(float4 *)&Ys[(yv + YV * p) * TCO + 4 * yq] = ry[p];
#pragma unroll
        for (int p = 0; p < NX; ++p) {
            const int idx = tid + 256 * p;
            if (idx < XROWS * XQ) *(float4 *)&Xs[(idx / XQ) * TCI + 4 * (idx % XQ)] = rx[p];
        }
        __syncthreads();
        if (v0 + 64 < v_end) load_block(v0 + 64);          // in flight while the matrix cores run
        const int kk0 = kg * 2 * KSTEPS + hh;               // this lane's first voxel row of the block
        const float *yp = Ys + kk0 * TCO + wr * 32 + nn;
        const float *xp = Xs + kk0 * TCI + wc * 32 + nn;   // row kk + t holds the input of voxel v + (t - 1)
        int xm = (int)((v0 + kk0) % W);                     // x coordinate of that voxel; rows advance by 2
#pragma unroll
        for (int k = 0; k < KSTEPS; ++k) {
            const float yv_ = yp[k * 2 * TCO];
            const float y0 = xm > 0 ? yv_ : 0.f, y2 = xm < W - 1 ? yv_ : 0.f;      // no -1 / +1 neighbour in this x-row
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y0, xp[(k * 2) * TCI], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(yv_, xp[(k * 2 + 1) * TCI], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(y2, xp[(k * 2 + 2) * TCI], acc[2], 0, 0, 0);
            xm += 2;
            if (xm >= W) xm -= W;
        }
    }
    // sum the K groups (fixed order), then store.  D layout: column (ci) = lane & 31,
    // row (co) = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
    if (KG > 1) {
        __shared__ float red[(KG > 1 ? (KG - 1) * NSUB : 1) * 1024];       // [KG-1][NSUB][16][64], one tap at a time
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            __syncthreads();
            if (kg > 0)
#pragma unroll
                for (int i = 0; i < 16; ++i) red[(((kg - 1) * NSUB + sub) * 16 + i) * 64 + lane] = acc[t][i];
            __syncthreads();
            if (kg == 0)
                for (int g = 1; g < KG; ++g)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[t][i] += red[(((g - 1) * NSUB + sub) * 16 + i) * 64 + lane];
        }
    }
    if (kg == 0) {
        const int CoP = a.co_tiles * TCO, CiP = a.ci_tiles * TCI;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            float *dst = a.partial + (((int64_t)split * 27 + (tap9 * 3 + t)) * CoP) * CiP;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = co0 + wr * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                const int ci = ci0 + wc * 32 + nn;
                dst[(int64_t)co * CiP + ci] = acc[t][i];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad on the bf16 matrix cores (precision "bf16": BASELINE config 4's mixed arithmetic).  Same decomposition as
// wgrad_mfma_kernel -- grid (co_tiles * ci_tiles, 9 (dz, dy) taps, splits), three x-taps per work-group from one staged
// input tile, K = voxels in blocks of 64 -- with both operands rounded to bfloat16 (round to nearest even) when they are
// staged and v_mfma_f32_32x32x16_bf16 (K = 16 voxels per instruction; fp32 accumulation, fp32 partials, the same fixed-order
// reduction).  The operands of that instruction are K-major per lane (8 consecutive voxels of one channel) while the tensors
// are channel-major per voxel: the tiles stay [voxel][channel] in LDS and the fragments are read with the transposing
// ds_read_b64_tr_b16 (four rows of 16 channels per 16-lane group, scratch/tr_read_test.hip checks the lane map).  Row
// strides of 192 B (64-channel tiles; 64 B for 32-channel ones) keep the four rows of a read on different banks.  The dY
// tile is staged three times -- as is, with the voxels at x = 0 zeroed (tap dx = -1 has no left neighbour there) and with
// those at x = W - 1 zeroed (dx = +1) -- because a per-element mask on packed fragments would cost more than the MFMAs.
typedef short v4s16 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t wg_pack_bf16(float x, float y)
{
    uint32_t a = __float_as_uint(x), b = __float_as_uint(y);
    a += 0x7FFFu + ((a >> 16) & 1u);
    b += 0x7FFFu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xFFFF0000u);
}

template <int TCO, int TCI>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const WgradArgs a)
{
    constexpr int NSUB = (TCO / 32) * (TCI / 32), KG = 4 / NSUB, KSTEPS = 4 / KG;     // 16-voxel K steps of a block per wave
    constexpr int NY = TCO / 16;
    constexpr int XROWS = 66, XQ = TCI / 4, NX = (XROWS * XQ + 255) / 256;
    constexpr int YSTR = TCO == 64 ? 192 : 64, XSTR = TCI == 64 ? 192 : 64;           // bytes per LDS row
    __shared__ __attribute__((aligned(16))) unsigned char Ys[3][64 * YSTR];
    __shared__ __attribute__((aligned(16))) unsigned char Xs[(XROWS + 2) * XSTR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, nn = lane & 31;
    const int co_t = blockIdx.x / a.ci_tiles, ci_t = blockIdx.x % a.ci_tiles;
    const int tap9 = blockIdx.y, split = blockIdx.z;
    const int dz = tap9 / 3 - 1, dy = tap9 % 3 - 1;
    const int D = a.D, H = a.H, W = a.W, Cin = a.CA + a.CB;
    const int64_t M = (int64_t)D * H * W;
    const int64_t v_begin = (int64_t)split * a.vox_per_split;
    int64_t v_end = v_begin + a.vox_per_split;
    if (v_end > M) v_end = M;
    const int sub = wave % NSUB, kg = wave / NSUB;
    const int wr = sub / (TCI / 32), wc = sub % (TCI / 32);
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int co0 = co_t * TCO, ci0 = ci_t * TCI;
    constexpr int YQ = TCO / 4, YV = 256 / YQ;
    const int yq = tid % YQ, yv = tid / YQ;
    float4 ry[NY], rx[NX];
    int rxm[NY];                                              // x coordinate of the dY voxel (-1: outside the split)

    auto load_block = [&](int64_t v0) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NY; ++p) {
            const int64_t v = v0 + yv + YV * p;
            float4 yvv = make_float4(0.f, 0.f, 0.f, 0.f);
            const int co = co0 + 4 * yq;
            rxm[p] = -1;
            if (v < v_end) {
                rxm[p] = (int)((uint32_t)v % (uint32_t)W);
                if (co + 3 < a.Cout) yvv = *(const float4 *)&a.dy[v * a.dy_stride + co];
                else
                    for (int j = 0; j < 4; ++j)
                        if (co + j < a.Cout) (&yvv.x)[j] = a.dy[v * a.dy_stride + co + j];
            }
            ry[p] = yvv;
        }
#pragma unroll
        for (int p = 0; p < NX; ++p) {
            const int idx = tid + 256 * p;
            const int row = idx / XQ, xq = idx % XQ;
            const int64_t u = v0 - 1 + row;
            float4 xvv = make_float4(0.f, 0.f, 0.f, 0.f);
            const int ci = ci0 + 4 * xq;
            if (row < XROWS && u >= 0 && u < M && ci < Cin) {
                // (32-bit: a chunk has far fewer than 2^32 voxels; three 64-bit divisions per loaded row were most of this kernel's vector instructions)
                const uint32_t uu = (uint32_t)u, hw = (uint32_t)W * (uint32_t)H, zq = uu / hw, rem = uu - zq * hw, yq = rem / (uint32_t)W;
                const int x = (int)(rem - yq * (uint32_t)W), y = (int)yq + dy, z = (int)zq + dz;
                if (z >= 0 && z < D && y >= 0 && y < H) {
                    const int64_t g = ((int64_t)z * H + y) * W + x;
                    const bool fromB = ci >= a.CA;
                    const float *src = fromB ? a.srcB : a.srcA, *sc = fromB ? a.scaleB : a.scaleA, *sh = fromB ? a.shiftB : a.shiftA;
                    const int C = fromB ? a.CB : a.CA, cc = fromB ? ci - a.CA : ci;
                    if (cc + 3 < C) {
                        xvv = *(const float4 *)&src[g * C + cc];
                        if (sc) {
                            const float4 s4 = *(const float4 *)&sc[cc], h4 = *(const float4 *)&sh[cc];
                            xvv.x = fmaxf(fmaf(xvv.x, s4.x, h4.x), 0.f);
                            xvv.y = fmaxf(fmaf(xvv.y, s4.y, h4.y), 0.f);
                            xvv.z = fmaxf(fmaf(xvv.z, s4.z, h4.z), 0.f);
                            xvv.w = fmaxf(fmaf(xvv.w, s4.w, h4.w), 0.f);
                        }
                    } else {
                        for (int j = 0; j < 4; ++j)
                            if (cc + j < C) {
                                float t = src[g * C + cc + j];
                                if (sc) t = fmaxf(fmaf(t, sc[cc + j], sh[cc + j]), 0.f);
                                (&xvv.x)[j] = t;
                            }
                    }
                }
            }
            rx[p] = xvv;
        }
    };

    // fragment addresses: lane (group g of 16, i = 4q + p) supplies row q, channels 16g + 4p .. +3 of its wave's 32-channel half
    const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, gg = (lane >> 4) & 1;
    const int yoff = (8 * hh + gq) * YSTR + (wr * 32 + 16 * gg + 4 * gp) * 2;
    const int xoff = (8 * hh + gq) * XSTR + (wc * 32 + 16 * gg + 4 * gp) * 2;
    typedef __attribute__((address_space(3))) v4s16 *lds_v4;
    union Frag { v4s16 h[2]; bf16x8w v; };

    if (v_begin < v_end) load_block(v_begin);
    for (int64_t v0 = v_begin; v0 < v_end; v0 += 64) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NY; ++p) {
            const uint2 pk = make_uint2(wg_pack_bf16(ry[p].x, ry[p].y), wg_pack_bf16(ry[p].z, ry[p].w));
            const uint2 zero = make_uint2(0u, 0u);
            const int o = (yv + YV * p) * YSTR + 8 * yq;
            *(uint2 *)&Ys[0][o] = rxm[p] > 0 ? pk : zero;                    // dx = -1: no left neighbour at x = 0
            *(uint2 *)&Ys[1][o] = pk;
            *(uint2 *)&Ys[2][o] = (rxm[p] >= 0 && rxm[p] < W - 1) ? pk : zero;     // dx = +1: none at x = W - 1
        }
#pragma unroll
        for (int p = 0; p < NX; ++p) {
            const int idx = tid + 256 * p;
            if (idx < XROWS * XQ) *(uint2 *)&Xs[(idx / XQ) * XSTR + 8 * (idx % XQ)] = make_uint2(wg_pack_bf16(rx[p].x, rx[p].y), wg_pack_bf16(rx[p].z, rx[p].w));
        }
        __syncthreads();
        if (v0 + 64 < v_end) load_block(v0 + 64);          // in flight while the matrix cores run
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int kb = (kg * KSTEPS + ks) * 16;
            Frag ya[3], xb[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                ya[t].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(&Ys[t][0] + yoff + kb * YSTR));
                ya[t].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(&Ys[t][0] + yoff + (kb + 4) * YSTR));
                xb[t].h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(&Xs[0] + xoff + (kb + t) * XSTR));
                xb[t].h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(&Xs[0] + xoff + (kb + t + 4) * XSTR));
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[t].v, xb[t].v, acc[t], 0, 0, 0);
        }
    }
    if (KG > 1) {
        __shared__ float red[(KG > 1 ? (KG - 1) * NSUB : 1) * 1024];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            __syncthreads();
            if (kg > 0)
#pragma unroll
                for (int i = 0; i < 16; ++i) red[(((kg - 1) * NSUB + sub) * 16 + i) * 64 + lane] = acc[t][i];
            __syncthreads();
            if (kg == 0)
                for (int g = 1; g < KG; ++g)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[t][i] += red[(((g - 1) * NSUB + sub) * 16 + i) * 64 + lane];
        }
    }
    if (kg == 0) {
        const int CoP = a.co_tiles * TCO, CiP = a.ci_tiles * TCI;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            float *dst = a.partial + (((int64_t)split * 27 + (tap9 * 3 + t)) * CoP) * CiP;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int co = co0 + wr * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                const int ci = ci0 + wc * 32 + nn;
                dst[(int64_t)co * CiP + ci] = acc[t][i];
            }
        }
    }
}

// dW[co][ci][tap] = sum over splits (fixed order).  Threads are laid over the PARTIAL layout ([tap][co][ci], ci fastest), so
// that every split is read in whole cache lines (indexing by the output layout, tap fastest, touched 27 lines per wave for
// 2-3 floats each: 1.2-1.6 TB/s on the 100 MB of level-0 partials); the scattered side is the 0.4 MB output.
__global__ void wgrad_reduce_kernel(const float *__restrict__ partial, int splits, int Cout, int Cin, int CoP, int CiP,
                                    float *__restrict__ dw)
{
    const int64_t per = (int64_t)27 * CoP * CiP;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < per; j += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(j % CiP), co = (int)((j / CiP) % CoP), t = (int)(j / ((int64_t)CiP * CoP));
        if (ci >= Cin || co >= Cout) continue;
        // the split-K partials (fp32 MFMA accumulations over one K block each) are of either sign and cancel: the final
        // sum runs in double, in a fixed order, so that it adds no rounding of its own
        double s = 0.0;
#pragma unroll 8
        for (int k = 0; k < splits; ++k) s += (double)partial[(int64_t)k * per + j];
        dw[((int64_t)co * Cin + ci) * 27 + t] = (float)s;
    }
}

// ------------------------------------------------------------------------------------------------
// small dgrad (few gradient channels in, up to 64 out): dX[v][ci] = sum_tap sum_co dY[v - tap][co] w[co][ci][tap]
// thread = (voxel, 4 output channels); weights read straight from the reference layout in P (L1-resident)
// ------------------------------------------------------------------------------------------------
__global__ void dgrad_small_kernel(const float *__restrict__ dy, int dstride, const float *__restrict__ wref, int Cout, int Cin,
                                   int D, int H, int W, float *__restrict__ dx, int64_t total)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int C4 = (Cin + 3) / 4;
    const int c4 = (int)(i % C4);
    const int64_t v = i / C4;
    const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((int64_t)W * H));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < 27; ++t) {
        const int zz = z - (t / 9 - 1), yy = y - ((t / 3) % 3 - 1), xx = x - (t % 3 - 1);
        if (zz < 0 || zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const float *gy = dy + (((int64_t)zz * H + yy) * W + xx) * dstride;
        for (int co = 0; co < Cout; ++co) {
            const float gv = gy[co];
            for (int j = 0; j < 4; ++j) {
                const int ci = 4 * c4 + j;
                if (ci < Cin) acc[j] = fmaf(gv, wref[((int64_t)co * Cin + ci) * 27 + t], acc[j]);
            }
        }
    }
    for (int j = 0; j < 4; ++j)
        if (4 * c4 + j < Cin) dx[v * Cin + 4 * c4 + j] = acc[j];
}

// ------------------------------------------------------------------------------------------------
// max-pool backward (windows do not overlap): thread = (pooled voxel, 4 channels) recomputes the arg-max of
// its window over act = relu(BN(raw)) (first maximum, like ATen), writes dpool there and zero elsewhere,
// then adds the decoder's skip gradient (channels [skip_off, skip_off+C) of dcat, if any).
// ------------------------------------------------------------------------------------------------
__global__ void pool_bwd_kernel(const float *__restrict__ raw, const float *__restrict__ scale, const float *__restrict__ shift,
                                int C, int D, int H, int W, int kz, int Do, int Ho, int Wo, const float *__restrict__ dpool,
                                const float *__restrict__ dcat, int cat_stride, int skip_off, float *__restrict__ gact,
                                int64_t total)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int C4 = C >> 2;
    const int c4 = (int)(i % C4);
    const int64_t mo = i / C4;
    const int x = (int)(mo % Wo), y = (int)((mo / Wo) % Ho), z = (int)(mo / ((int64_t)Wo * Ho));
    const float4 vs = *(const float4 *)&scale[4 * c4], vh = *(const float4 *)&shift[4 * c4];
    const float4 gp = *(const float4 *)&dpool[mo * C + 4 * c4];
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int arg[4] = {-1, -1, -1, -1};
    for (int i0 = 0; i0 < kz; ++i0)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                const int zz = z * kz + i0, yy = 2 * y - 1 + j, xx = 2 * x - 1 + k;
                if (zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const float4 t = *(const float4 *)&raw[(((int64_t)zz * H + yy) * W + xx) * C + 4 * c4];
                const float a4[4] = {fmaxf(fmaf(t.x, vs.x, vh.x), 0.f), fmaxf(fmaf(t.y, vs.y, vh.y), 0.f),
                                     fmaxf(fmaf(t.z, vs.z, vh.z), 0.f), fmaxf(fmaf(t.w, vs.w, vh.w), 0.f)};
                const int code = (i0 * 2 + j) * 2 + k;
                for (int e = 0; e < 4; ++e)
                    if (a4[e] > best[e]) { best[e] = a4[e]; arg[e] = code; }
            }
    for (int i0 = 0; i0 < kz; ++i0)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                const int zz = z * kz + i0, yy = 2 * y - 1 + j, xx = 2 * x - 1 + k;
                if (zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const int code = (i0 * 2 + j) * 2 + k;
                const int64_t g = ((int64_t)zz * H + yy) * W + xx;
                float4 o = make_float4(arg[0] == code ? gp.x : 0.f, arg[1] == code ? gp.y : 0.f,
                                       arg[2] == code ? gp.z : 0.f, arg[3] == code ? gp.w : 0.f);
                if (dcat) {
                    const float4 s = *(const float4 *)&dcat[g * cat_stride + skip_off + 4 * c4];
                    o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
                }
                *(float4 *)&gact[g * C + 4 * c4] = o;
            }
}

// ------------------------------------------------------------------------------------------------
// depthwise transposed conv backward.  Forward: u[zo,yo,xo,c] = act[zs,ys,xs,c] * w[kidx][c] + b[c] on the
// cropped box.  thread = (source voxel, 4 channels): gathers its <= kz*4 outputs:
//   g_act[src][c] = sum dU * w ;  partial sums of dU * act (per kidx) and dU (bias) per block.
// part layout [blocks][kt + 1][C]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float *__restrict__ raw, const float *__restrict__ scale,
                                                           const float *__restrict__ shift, int C, int D, int H, int W,
                                                           int kz, int off, const float *__restrict__ w,
                                                           const float *__restrict__ dcat, int cat_stride, int Do, int Ho,
                                                           int Wo, float *__restrict__ gact, float *__restrict__ part,
                                                           int vox_per_block)
{
    // A thread keeps ONE channel quad for the whole block (256 % C4 == 0), so the per-(kidx, channel)
    // parameter-gradient sums stay in registers; threads sharing a quad are combined through LDS in a
    // fixed order -> deterministic.
    extern __shared__ float sm[];                      // [256 / C4][(kt+1) * 4] per quad ... laid out [slot][9*4][C4]
    const int kt = kz * 4;
    const int C4 = C >> 2;
    const int c4 = threadIdx.x % C4, slot = threadIdx.x / C4, nslot = 256 / C4;
    const int64_t M = (int64_t)D * H * W;
    const int64_t v0 = (int64_t)blockIdx.x * vox_per_block;
    int64_t v1 = v0 + vox_per_block;
    if (v1 > M) v1 = M;
    float aw[8][4], ab[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) aw[r][e] = 0.f;
    const float4 vsx = *(const float4 *)&scale[4 * c4], vhx = *(const float4 *)&shift[4 * c4];
    if (slot < nslot)
        for (int64_t v = v0 + slot; v < v1; v += nslot) {
            const int xs = (int)(v % W), ys = (int)((v / W) % H), zs = (int)(v / ((int64_t)W * H));
            const float4 t = *(const float4 *)&raw[v * C + 4 * c4];
            const float a4[4] = {fmaxf(fmaf(t.x, vsx.x, vhx.x), 0.f), fmaxf(fmaf(t.y, vsx.y, vhx.y), 0.f),
                                 fmaxf(fmaf(t.z, vsx.z, vhx.z), 0.f), fmaxf(fmaf(t.w, vsx.w, vhx.w), 0.f)};
            float gsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        if (a >= kz) continue;
                        const int zo = zs * kz + a, yo = 2 * ys + j - off, xo = 2 * xs + k - off;
                        if (zo >= Do || yo < 0 || yo >= Ho || xo < 0 || xo >= Wo) continue;
                        const int kidx = (a * 2 + j) * 2 + k;
                        const float4 du = *(const float4 *)&dcat[(((int64_t)zo * Ho + yo) * Wo + xo) * cat_stride + 4 * c4];
                        const float4 wv = *(const float4 *)&w[(int64_t)kidx * C + 4 * c4];
                        const float d4[4] = {du.x, du.y, du.z, du.w}, w4[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            gsum[e] = fmaf(d4[e], w4[e], gsum[e]);
                            aw[kidx][e] = fmaf(d4[e], a4[e], aw[kidx][e]);
                            ab[e] += d4[e];
                        }
                    }
            *(float4 *)&gact[v * C + 4 * c4] = make_float4(gsum[0], gsum[1], gsum[2], gsum[3]);
        }
    // sm[slot][r][c4*4+e], r = 0..kt (kt = bias row)
    if (slot < nslot) {
        for (int r = 0; r < kt; ++r)
            for (int e = 0; e < 4; ++e) sm[((size_t)slot * (kt + 1) + r) * C + 4 * c4 + e] = aw[r][e];
        for (int e = 0; e < 4; ++e) sm[((size_t)slot * (kt + 1) + kt) * C + 4 * c4 + e] = ab[e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (kt + 1) * C; i += 256) {
        float acc = 0.f;
        for (int sl = 0; sl < nslot; ++sl) acc += sm[(size_t)sl * (kt + 1) * C + i];
        part[(int64_t)blockIdx.x * (kt + 1) * C + i] = acc;
    }
}

// up-conv parameter gradients: reference layout dW[c][kidx], db[c]
// one wave per output: lane l sums the blocks l, l + 64, ... (double), then a fixed xor tree -- deterministic, and 1 300 blocks
// (up3 at 128 source voxels per block) no longer cost a serial walk per output
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const long long b = __double_as_longlong(v);
        const int lo = __shfl_xor((int)(b & 0xFFFFFFFFll), o), hi = __shfl_xor((int)(b >> 32), o);
        v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    return v;
}

__global__ __launch_bounds__(256) void up_grad_finalize_kernel(const float *__restrict__ part, int blocks, int C, int kt,
                                                               float *__restrict__ dw, float *__restrict__ db, int accumulate)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= (kt + 1) * C) return;                          // (whole waves leave together)
    const int r = i / C, c = i % C;
    double s = 0.0;
    for (int b = lane; b < blocks; b += 64) s += (double)part[(int64_t)b * (kt + 1) * C + i];
    s = wave_sum_f64(s);
    if (lane != 0) return;
    // accumulate: the transposed convolutions are shared by the decoders of a forked network (unet.py:213-242)
    if (r < kt) dw[(int64_t)c * kt + r] = (float)s + (accumulate ? dw[(int64_t)c * kt + r] : 0.f);
    else db[c] = (float)s + (accumulate ? db[c] : 0.f);
}

// dst[m][c0 + c] += src[m][c0 + c] for c < n (row stride `stride`): gradients that two decoders send to the same tensor
__global__ void add_cols_kernel(float *__restrict__ dst, const float *__restrict__ src, int stride, int c0, int n, int64_t M)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * n) return;
    const int64_t o = (i / n) * stride + c0 + (int)(i % n);
    dst[o] += src[o];
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad)
// ------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, int64_t n, float lr, float bc1, float bc2)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = 0.9f * m[i] + 0.1f * gi;
        const float vi = 0.999f * v[i] + 0.001f * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + 1e-8f;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}

// ------------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------------
static int64_t param_layout(UNetState *u, TrainState *t)
{
    int64_t off = 0;
    for (int l0 = 0; l0 < u->nlayers; l0 += 2) {          // module by module: c0..c4, c5_0..c8_0, then c5_1..c8_1 of a second fork
        for (int j = 0; j < 2; ++j) {
            const ConvBN &L = u->cv[l0 + j];
            t->off_w[l0 + j] = off; off += (int64_t)L.cout * L.cin * 27;
            t->off_b[l0 + j] = off; off += L.cout;
        }
        for (int j = 0; j < 2; ++j) {
            const ConvBN &L = u->cv[l0 + j];
            t->off_gamma[l0 + j] = off; off += L.cout;
            t->off_beta[l0 + j] = off; off += L.cout;
        }
    }
    for (int k = 0; k < 4; ++k) {
        t->off_upw[k] = off; off += (int64_t)u->up[k].C * u->up[k].kz * 4;
        t->off_upb[k] = off; off += u->up[k].C;
    }
    return off;
}

static int repack_all(isx_ctx *ctx)
{
    UNetState *u = ctx->unet;
    TrainState *t = u->train;
    t->bf_fresh = u->precision == ISX_PRECISION_BF16;
    for (int li = 0; li < u->nlayers; ++li) {
        ConvBN &L = u->cv[li];
        const float *wref = t->P + t->off_w[li];
        const int64_t n = (int64_t)L.cout * L.cin * 27;
        unsigned grid = cdiv(n, 256);
        if (grid > 4096) grid = 4096;
        if (use_mfma(L.cin, L.cout)) {
            if (t->w_dgrad[li]) repack_mfma_both_kernel<<<grid, 256, 0, ctx->stream>>>(wref, L.cout, L.cin, L.w, t->w_dgrad[li]);
            else repack_mfma_kernel<<<grid, 256, 0, ctx->stream>>>(wref, L.cout, L.cin, 0, L.w, 0);
            if (u->precision == ISX_PRECISION_BF16) {
                if (t->wbf_fwd64[li]) ISX_TRY(repack_bf16(ctx, wref, L.cout, L.cin, 0, 64, L.cout, L.cin, t->wbf_fwd64[li]));
                if (t->wbf_fwd32[li]) ISX_TRY(repack_bf16(ctx, wref, L.cout, L.cin, 0, 32, L.cout, L.cin, t->wbf_fwd32[li]));
                if (t->wbf_dg[li]) ISX_TRY(repack_bf16(ctx, wref, L.cout, L.cin, 1, t->wbf_dg_bn[li], L.cin, L.cout, t->wbf_dg[li]));
            }
        } else if (use_fold(L.cin, L.cout)) {
            repack_fold_kernel<<<grid, 256, 0, ctx->stream>>>(wref, L.cout, L.cin, L.w);
            if (t->w_dgrad[li]) repack_mfma_kernel<<<grid, 256, 0, ctx->stream>>>(wref, L.cout, L.cin, 1, t->w_dgrad[li], 8);
        } else {
            repack_direct_kernel<<<grid, 256, 0, ctx->stream>>>(wref, L.cout, L.cin, L.cin == 1 ? 1 : 0, L.w);
        }
        ISX_CHECK_LAUNCH();
        if (L.cout < 8)      // thin layers read an 8-padded private copy of the bias
            ISX_HIP(hipMemcpyAsync(L.bias, t->P + t->off_b[li], L.cout * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    }
    for (int k = 0; k < 4; ++k) {
        UpConv &U = u->up[k];
        repack_up_kernel<<<cdiv(U.C * U.kz * 4, 256), 256, 0, ctx->stream>>>(t->P + t->off_upw[k], U.C, U.kz * 4, U.w);
        ISX_CHECK_LAUNCH();
    }
    return ISX_OK;
}

int train_ensure_bf16_packs(isx_ctx *ctx)
{
    UNetState *u = ctx->unet;
    if (!u || !u->train || u->precision != ISX_PRECISION_BF16 || u->train->bf_fresh) return ISX_OK;
    return repack_all(ctx);
}

// gradient of the loss w.r.t. the post-activation output of layer li arrives in g (channel stride gstride);
// produces draw[li] and the BN / bias gradients.
static int bn_backward(isx_ctx *ctx, int li, const float *g, int gstride, int act)
{
    UNetState *u = ctx->unet;
    TrainState *t = u->train;
    const ConvBN &L = u->cv[li];
    const Level &l = u->lv[MOD_LEVEL[layer_module(li)]];
    const int C = L.cout;
    const int rows = 512;
    const int blocks = (int)((l.M + rows - 1) / rows);
    ISX_REQUIRE((size_t)blocks * C * 2 <= t->part_floats, "internal: partial buffer too small");
    // four channels per lane where the rows allow 16-byte accesses (every layer but the 5-channel ones)
    const bool v4 = C % 4 == 0 && C <= 1024 && gstride % 4 == 0 && t->dstride[li] % 4 == 0 && ((uintptr_t)g & 15) == 0 &&
                    ((uintptr_t)t->draw[li] & 15) == 0 && ((uintptr_t)u->raw[li] & 15) == 0;
    if (v4)
        bn_bwd_reduce4_kernel<<<blocks, 256, 256 * 8 * sizeof(double), ctx->stream>>>(g, gstride, u->raw[li], L.scale, L.shift, L.bmean,
                                                                                  L.invstd, C, l.M, act, rows, t->part);
    else
    bn_bwd_reduce_kernel<<<blocks, 256, 256 * 2 * sizeof(double), ctx->stream>>>(g, gstride, u->raw[li], L.scale, L.shift, L.bmean,
                                                                             L.invstd, C, l.M, act, rows, t->part);
    ISX_CHECK_LAUNCH();
    bn_bwd_finalize_kernel<<<C, 256, 0, ctx->stream>>>(t->part, blocks, C, (double)l.M, t->G + t->off_gamma[li],
                                                     t->G + t->off_beta[li], t->coef);
    ISX_CHECK_LAUNCH();
    if (v4)
        bn_bwd_apply4_kernel<<<blocks, 256, 256 * 4 * sizeof(float), ctx->stream>>>(g, gstride, u->raw[li], L.scale, L.shift, L.bmean,
                                                                                L.invstd, t->coef, C, l.M, act, rows, t->draw[li],
                                                                                t->dstride[li], t->part);
    else
    bn_bwd_apply_kernel<<<blocks, 256, 256 * sizeof(float), ctx->stream>>>(g, gstride, u->raw[li], L.scale, L.shift, L.bmean,
                                                                        L.invstd, t->coef, C, l.M, act, rows, t->draw[li],
                                                                        t->dstride[li], t->part);
    ISX_CHECK_LAUNCH();
    colsum_kernel<<<C, 256, 0, ctx->stream>>>(t->part, blocks, C, t->G + t->off_b[li]);      // conv bias gradient
    ISX_CHECK_LAUNCH();
    return ISX_OK;
}

// ------------------------------------------------------------------------------------------------
// wgrad of the thin layers at the full-resolution level (round 3).  On the matrix-core kernel a 1 -> 32 or 5 -> 5 layer
// computed zero-padded 32 x 32 tiles for nine (dz, dy) rows of work-groups, each re-reading dY: 0.6 ms per layer for
// 0.5 GFLOP.  Here every voxel's dY row is read once and the 27 shifted inputs come from L1.
//   wgrad_cin1_kernel: Cin = 1.  thread = (output channel, voxel lane); 27 accumulators per thread.
//   wgrad_few_kernel:  Cin, Cout <= 8.  thread = (tap, voxel lane); Cout x Cin accumulators per thread.
// Partials [block][27][Cout][Cin] (fp32 sums over <= 2048 voxels), reduced in block order by wgrad_thin_reduce_kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad_cin1_kernel(const float *__restrict__ dy, int dy_stride, const float *__restrict__ x, int Cout,
                                                         int D, int H, int W, int rows_per_block, int blocks_per_plane, float *__restrict__ part)
{
    // a block = `rows_per_block` consecutive voxels of ONE z-plane; the input around them (three planes, one image row and
    // one voxel either side) is staged in LDS once, so a tap is an LDS broadcast read instead of a global load per lane
    extern __shared__ float sm1[];                            // xs[3][rows + 2 W + 2] | sred[nslot][27][Cout]
    const int span = rows_per_block + 2 * W + 2;
    float *xs = sm1, *sred = sm1 + 3 * span;
    const int co = threadIdx.x % Cout, vl = threadIdx.x / Cout, nslot = 256 / Cout;
    const int z = blockIdx.x / blocks_per_plane, p0 = (blockIdx.x % blocks_per_plane) * rows_per_block;
    const int HW = H * W;
    int p1 = p0 + rows_per_block;
    if (p1 > HW) p1 = HW;
    for (int i = threadIdx.x; i < 3 * span; i += 256) {
        const int pl = i / span, q = p0 - W - 1 + i % span, zz = z + pl - 1;
        xs[i] = (zz >= 0 && zz < D && q >= 0 && q < HW) ? x[(int64_t)zz * HW + q] : 0.f;
    }
    __syncthreads();
    float acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = 0.f;
    if (vl < nslot)
        for (int p = p0 + vl; p < p1; p += nslot) {
            const float g = dy[((int64_t)z * HW + p) * dy_stride + co];
            const int xx = p % W, yy = p / W;
            const float *c0 = xs + (p - p0) + W + 1;          // this voxel in plane z - 1
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const int dz = t / 9 - 1, dyy = (t / 3) % 3 - 1, dx = t % 3 - 1;
                // planes outside the volume and positions outside the plane were staged as zeros; what is left are the
                // wrap-arounds of the linear index: x +- 1 across a row end, y +- 1 across the plane's first / last row
                const bool in = (unsigned)(xx + dx) < (unsigned)W && (unsigned)(yy + dyy) < (unsigned)H;
                const float xv = in ? c0[(dz + 1) * span + dyy * W + dx] : 0.f;
                acc[t] = fmaf(g, xv, acc[t]);
            }
        }
    if (vl < nslot)
#pragma unroll
        for (int t = 0; t < 27; ++t) sred[(vl * 27 + t) * Cout + co] = acc[t];
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * Cout; i += 256) {
        float a = 0.f;
        for (int k = 0; k < nslot; ++k) a += sred[k * 27 * Cout + i];
        part[(int64_t)blockIdx.x * 27 * Cout + i] = a;         // [t][co] (Cin = 1)
    }
}

template <int CO, int CI>
__global__ __launch_bounds__(256) void wgrad_few_kernel(const float *__restrict__ dy, int dy_stride, const float *__restrict__ src,
                                                        const float *__restrict__ scale, const float *__restrict__ shift, int D, int H,
                                                        int W, int rows_per_block, float *__restrict__ part)
{
    constexpr int NL = 9;                                     // voxel lanes: 27 taps x 9 = 243 threads
    __shared__ float sred[NL * 27 * CO * CI];
    const int t = threadIdx.x % 27, vl = threadIdx.x / 27;
    const int64_t M = (int64_t)D * H * W;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float acc[CO][CI];
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int c = 0; c < CI; ++c) acc[o][c] = 0.f;
    float sc[CI], sh[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) { sc[c] = scale ? scale[c] : 1.f; sh[c] = scale ? shift[c] : 0.f; }
    const int dz = t / 9 - 1, dyy = (t / 3) % 3 - 1, dx = t % 3 - 1;
    if (vl < NL)
        for (int64_t m = r0 + vl; m < r1; m += NL) {
            const int xx = (int)(m % W) + dx, yy = (int)((m / W) % H) + dyy, zz = (int)(m / ((int64_t)W * H)) + dz;
            if (zz < 0 || zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const int64_t nb = ((int64_t)zz * H + yy) * W + xx;
            float xv[CI], g[CO];
#pragma unroll
            for (int c = 0; c < CI; ++c) {
                const float v = src[nb * CI + c];
                xv[c] = scale ? fmaxf(fmaf(v, sc[c], sh[c]), 0.f) : v;
            }
#pragma unroll
            for (int o = 0; o < CO; ++o) g[o] = dy[m * dy_stride + o];
#pragma unroll
            for (int o = 0; o < CO; ++o)
#pragma unroll
                for (int c = 0; c < CI; ++c) acc[o][c] = fmaf(g[o], xv[c], acc[o][c]);
        }
    if (vl < NL)
#pragma unroll
        for (int o = 0; o < CO; ++o)
#pragma unroll
            for (int c = 0; c < CI; ++c) sred[((vl * 27 + t) * CO + o) * CI + c] = acc[o][c];
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * CO * CI; i += 256) {
        float a = 0.f;
        for (int k = 0; k < NL; ++k) a += sred[k * 27 * CO * CI + i];
        part[(int64_t)blockIdx.x * 27 * CO * CI + i] = a;      // [t][co][ci]
    }
}

// dW[co][ci][t] = sum over blocks of partial[block][t][co][ci]: one wave per output (lane l: blocks l, l + 64, ...; fixed xor tree)
__global__ __launch_bounds__(256) void wgrad_thin_reduce_kernel(const float *__restrict__ part, int blocks, int Cout, int Cin, float *__restrict__ dw)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = 27 * Cout * Cin;
    if (i >= n) return;
    double a = 0.0;
    for (int b = lane; b < blocks; b += 64) a += (double)part[(int64_t)b * n + i];
    a = wave_sum_f64(a);
    if (lane != 0) return;
    const int ci = i % Cin, co = (i / Cin) % Cout, t = i / (Cin * Cout);
    dw[((int64_t)co * Cin + ci) * 27 + t] = (float)a;
}

static int wgrad(isx_ctx *ctx, int li, Src sa, Src sb)
{
    UNetState *u = ctx->unet;
    TrainState *t = u->train;
    const ConvBN &L = u->cv[li];
    const Level &l = u->lv[MOD_LEVEL[layer_module(li)]];
    ISX_REQUIRE(l.M < (int64_t)0x7FFFFFFF, "internal: weight-gradient kernels index a chunk's voxels in 32 bits");
    if (ctx->tune.thin_wgrad && sb.C == 0 && sa.C == L.cin) {
        // the thin layers at the big levels: dedicated kernels (see above)
        const int rows = 2048;
        int blocks = (int)((l.M + rows - 1) / rows);
        const int rows1 = 1024, bpp = (int)(((int64_t)l.H * l.W + rows1 - 1) / rows1);       // Cin = 1: blocks stay inside a z-plane
        const size_t lds1 = ((size_t)3 * (rows1 + 2 * l.W + 2) + (size_t)(L.cout ? 256 / L.cout : 1) * 27 * L.cout) * sizeof(float);
        const bool cin1 = L.cin == 1 && !sa.scale && L.cout >= 8 && 256 % L.cout == 0 && lds1 <= 64 * 1024;
        if (cin1) blocks = l.D * bpp;
        const size_t need = (size_t)blocks * 27 * L.cout * L.cin;
        bool done = false;
        if (need <= t->wpart_floats) {
            if (cin1) {
                wgrad_cin1_kernel<<<blocks, 256, lds1, ctx->stream>>>(t->draw[li], t->dstride[li], sa.p, L.cout, l.D, l.H, l.W, rows1, bpp, t->wpart);
                done = true;
            } else if (L.cin == 5 && L.cout == 5) {
                wgrad_few_kernel<5, 5><<<blocks, 256, 0, ctx->stream>>>(t->draw[li], t->dstride[li], sa.p, sa.scale, sa.shift, l.D, l.H, l.W, rows, t->wpart);
                done = true;
            }
        }
        if (done) {
            ISX_CHECK_LAUNCH();
            wgrad_thin_reduce_kernel<<<cdiv(27 * L.cout * L.cin, 4), 256, 0, ctx->stream>>>(t->wpart, blocks, L.cout, L.cin, t->G + t->off_w[li]);
            ISX_CHECK_LAUNCH();
            return ISX_OK;
        }
    }
    WgradArgs a;
    a.dy = t->draw[li];
    a.dy_stride = t->dstride[li];
    a.srcA = sa.p; a.scaleA = sa.scale; a.shiftA = sa.shift; a.CA = sa.C;
    a.srcB = sb.p; a.scaleB = sb.scale; a.shiftB = sb.shift; a.CB = sb.C;
    a.Cout = L.cout;
    a.D = l.D; a.H = l.H; a.W = l.W;
    const int tco = L.cout <= 32 ? 32 : 64, tci = L.cin <= 32 ? 32 : 64;
    a.co_tiles = (L.cout + tco - 1) / tco;
    a.ci_tiles = (L.cin + tci - 1) / tci;
    const int64_t chunks = (l.M + 63) / 64;
    int want = ctx->tune.wgrad_wgs / (a.co_tiles * a.ci_tiles * 18) + 1;          // enough work-groups to fill the chip (9 per tile, three x-taps each; twice the K range of round 1's split: the partials are 3x larger per work-group)
    if (want > chunks / 4) want = (int)std::max<int64_t>(1, chunks / 4);     // at least 4 K blocks per work-group
    if (want > 256) want = 256;
    a.vox_per_split = ((chunks + want - 1) / want) * 64;
    a.splits = (int)((l.M + a.vox_per_split - 1) / a.vox_per_split);
    a.partial = t->wpart;
    const size_t need = (size_t)a.splits * 27 * a.co_tiles * tco * a.ci_tiles * tci;
    ISX_REQUIRE(need <= t->wpart_floats, "internal: wgrad partial buffer too small (%zu > %zu)", need, t->wpart_floats);
    const dim3 wgrid(a.co_tiles * a.ci_tiles, 9, a.splits);
    // precision "bf16": every layer with Cin % 32 == 0 (the wide layers, whose forward runs on the bf16 matrix cores, and the
    // 64 -> n tail layer) takes its weight gradient there too; c0.conv0, the n -> n layer and every other precision keep fp32 products
    const bool bfw = u->precision == ISX_PRECISION_BF16 && L.cin % 32 == 0 && ctx->tune.bf16_wgrad;      // (also the 64 -> n tail layer)
    if (bfw) {
        if (tco == 64 && tci == 64) wgrad_bf16_kernel<64, 64><<<wgrid, 256, 0, ctx->stream>>>(a);
        else if (tco == 32 && tci == 64) wgrad_bf16_kernel<32, 64><<<wgrid, 256, 0, ctx->stream>>>(a);
        else if (tco == 64 && tci == 32) wgrad_bf16_kernel<64, 32><<<wgrid, 256, 0, ctx->stream>>>(a);
        else wgrad_bf16_kernel<32, 32><<<wgrid, 256, 0, ctx->stream>>>(a);
    } else if (tco == 64 && tci == 64) wgrad_mfma_kernel<64, 64><<<wgrid, 256, 0, ctx->stream>>>(a);
    else if (tco == 32 && tci == 64) wgrad_mfma_kernel<32, 64><<<wgrid, 256, 0, ctx->stream>>>(a);
    else if (tco == 64 && tci == 32) wgrad_mfma_kernel<64, 32><<<wgrid, 256, 0, ctx->stream>>>(a);
    else wgrad_mfma_kernel<32, 32><<<wgrid, 256, 0, ctx->stream>>>(a);
    ISX_CHECK_LAUNCH();
    const int64_t n = (int64_t)27 * a.co_tiles * tco * a.ci_tiles * tci;
    unsigned grid = cdiv(n, 256);
    if (grid > 8192) grid = 8192;
    wgrad_reduce_kernel<<<grid, 256, 0, ctx->stream>>>(t->wpart, a.splits, L.cout, L.cin, a.co_tiles * tco, a.ci_tiles * tci,
                                                    t->G + t->off_w[li]);
    ISX_CHECK_LAUNCH();
    return ISX_OK;
}

// gradient w.r.t. the input of conv li (all Cin channels, stride Cin) into dst
static int dgrad(isx_ctx *ctx, int li, float *dst)
{
    UNetState *u = ctx->unet;
    TrainState *t = u->train;
    const ConvBN &L = u->cv[li];
    const Level &l = u->lv[MOD_LEVEL[layer_module(li)]];
    if (u->precision == ISX_PRECISION_BF16 && t->wbf_dg[li] && t->dstride[li] == L.cout) {
        // bf16 mixed precision: the same operator on the bf16 matrix cores (d(raw) is rounded to bf16 at the LDS store)
        return launch_conv_h2(ctx, l, 1, Src{t->draw[li], nullptr, nullptr, L.cout}, Src{nullptr, nullptr, nullptr, 0}, t->wbf_dg[li],
                              t->zeros, L.cin, dst, nullptr, nullptr, 1, ctx->tune.xcd_remap, nullptr, 2);
    }
    if (t->w_dgrad[li]) {
        // the forward kernel with Cin <-> Cout: "input" = d(raw) (Cout channels), "output" = dX (Cin channels)
        return launch_conv_mfma(ctx, l, 1, Src{t->draw[li], nullptr, nullptr, t->dstride[li]}, Src{nullptr, nullptr, nullptr, 0},
                                t->w_dgrad[li], t->zeros, L.cin, dst, nullptr, nullptr, 1);
    }
    const int64_t total = l.M * ((L.cin + 3) / 4);
    dgrad_small_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(t->draw[li], t->dstride[li], t->P + t->off_w[li], L.cout, L.cin,
                                                              l.D, l.H, l.W, dst, total);
    ISX_CHECK_LAUNCH();
    return ISX_OK;
}

static int backward(isx_ctx *ctx)
{
    UNetState *u = ctx->unet;
    TrainState *t = u->train;
    const Level *lv = u->lv;
    const Src none{nullptr, nullptr, nullptr, 0};
    auto act = [&](int li) { const ConvBN &L = u->cv[li]; return Src{u->raw[li], L.scale, L.shift, L.cout}; };
    // ---- decoder(s), last module first.  A forked network (unet.py:313-325) sends two gradients to every skip tensor, to
    //      up0's output and to the shared transposed-convolution weights: fork 1 runs first, fork 0 adds its share. -----
    for (int f = u->nforks - 1; f >= 0; --f) {
        for (int k = 3; k >= 0; --k) {
            const int l1 = dec_layer(f, k, 1), l0 = dec_layer(f, k, 0);
            const int skip_li = 2 * (3 - k) + 1;
            const UpConv &U = u->up[k];
            const Level &lo = lv[3 - k];
            // conv1 of the module: g arrives in gact[l1] (the loss wrote it for the last module)
            ISX_TRY(bn_backward(ctx, l1, t->gact[l1], u->cv[l1].cout, k == 3 ? (u->final_act[f] == ISX_FINAL_SIGMOID ? 1 : 2) : 0));
            ISX_TRY(wgrad(ctx, l1, act(l0), none));
            ISX_TRY(dgrad(ctx, l1, t->gact[l0]));
            // conv0 of the module: input = cat(up_k, act(skip))
            ISX_TRY(bn_backward(ctx, l0, t->gact[l0], u->cv[l0].cout, 0));
            ISX_TRY(wgrad(ctx, l0, Src{u->ups[ups_index(f, k)], nullptr, nullptr, U.C}, act(skip_li)));
            ISX_TRY(dgrad(ctx, l0, t->dcat[f][k]));                    // [M][2C]: first C -> up_k, last C -> skip
            if (u->nforks > 1 && f == 0) {
                // the other decoder's share: the skip half always; for k = 0 also the up half (up0(c4) is one tensor)
                const int c0 = k == 0 ? 0 : U.C, n = k == 0 ? 2 * U.C : U.C;
                add_cols_kernel<<<cdiv(lo.M * n, 256), 256, 0, ctx->stream>>>(t->dcat[0][k], t->dcat[1][k], 2 * U.C, c0, n, lo.M);
                ISX_CHECK_LAUNCH();
            }
            if (k == 0 && f > 0) continue;                           // up0's backward runs once, on the summed gradient
            // transposed conv backward -> gradient of the module that fed it (c4, c5, c6, c7 second conv)
            const int li_in = (k == 0) ? 9 : dec_layer(f, k - 1, 1);
            const Level &ls = lv[MOD_LEVEL[layer_module(li_in)]];
            const ConvBN &Lin = u->cv[li_in];
            const int kt = U.kz * 4;
            const int vpb = 128;                                      // (1024 until round 3: 163 work-groups for up3's 166 k source voxels, 1 ms)
            const int blocks = (int)((ls.M + vpb - 1) / vpb);
            ISX_REQUIRE((size_t)blocks * (kt + 1) * U.C <= t->part_floats, "internal: partial buffer too small");
            upsample_bwd_kernel<<<blocks, 256, (size_t)(256 / (U.C / 4)) * (kt + 1) * U.C * sizeof(float), ctx->stream>>>(
                u->raw[li_in], Lin.scale, Lin.shift, U.C, ls.D, ls.H, ls.W, U.kz, k == 3 ? 1 : 0, U.w, t->dcat[f][k], 2 * U.C, lo.D, lo.H,
                lo.W, t->gact[li_in], t->part, vpb);
            ISX_CHECK_LAUNCH();
            up_grad_finalize_kernel<<<cdiv((kt + 1) * U.C, 4), 256, 0, ctx->stream>>>(t->part, blocks, U.C, kt, t->G + t->off_upw[k],
                                                                                     t->G + t->off_upb[k],
                                                                                     (u->nforks > 1 && f == 0 && k > 0) ? 1 : 0);
            ISX_CHECK_LAUNCH();
        }
    }
    // ---- encoder, bottom first ---------------------------------------------------------------------
    for (int mi = 4; mi >= 0; --mi) {
        const int l1 = 2 * mi + 1, l0 = 2 * mi;
        const Level &l = lv[mi];
        ISX_TRY(bn_backward(ctx, l1, t->gact[l1], u->cv[l1].cout, 0));
        ISX_TRY(wgrad(ctx, l1, act(l0), none));
        ISX_TRY(dgrad(ctx, l1, t->gact[l0]));
        ISX_TRY(bn_backward(ctx, l0, t->gact[l0], u->cv[l0].cout, 0));
        if (mi == 0) {
            ISX_TRY(wgrad(ctx, l0, Src{u->x0, nullptr, nullptr, 1}, none));
            break;
        }
        ISX_TRY(wgrad(ctx, l0, Src{u->pool[mi - 1], nullptr, nullptr, MOD_COUT[mi - 1]}, none));
        ISX_TRY(dgrad(ctx, l0, t->dpool[mi - 1]));
        // pool backward into the previous module's output gradient, plus its share of the decoder concat
        const int lp = 2 * (mi - 1) + 1;
        const ConvBN &Lp = u->cv[lp];
        const Level &lprev = lv[mi - 1];
        const int kdec = 3 - (mi - 1);                              // decoder stage that consumed this skip
        const int64_t total = l.M * (Lp.cout / 4);
        pool_bwd_kernel<<<cdiv(total, 256), 256, 0, ctx->stream>>>(u->raw[lp], Lp.scale, Lp.shift, Lp.cout, lprev.D, lprev.H, lprev.W,
                                                                mi == 4 ? 2 : 1, l.D, l.H, l.W, t->dpool[mi - 1], t->dcat[0][kdec],
                                                                2 * Lp.cout, Lp.cout, t->gact[lp], total);
        ISX_CHECK_LAUNCH();
    }
    return ISX_OK;
}

}  // namespace isx

using namespace isx;

extern "C" {

int64_t isx_train_param_count(isx_ctx *ctx)
{
    if (!ctx || !ctx->unet || !ctx->unet->committed) { set_error("weights not committed"); return ISX_ERR_STATE; }
    TrainState tmp;
    return param_layout(ctx->unet, &tmp);
}

int isx_train_bind(isx_ctx *ctx, const int64_t chunk[3], float *params, float *grads, float *adam_m, float *adam_v,
                   int64_t n)
{
    ISX_REQUIRE(ctx && chunk && params && grads && adam_m && adam_v, "null argument");
    if (!ctx->unet || !ctx->unet->committed) { set_error("weights not committed"); return ISX_ERR_STATE; }
    ISX_HIP(hipSetDevice(ctx->device));
    UNetState *u = ctx->unet;
    ISX_REQUIRE(!u->train, "training state already bound to this context");
    ISX_REQUIRE(u->out_channels <= 8, "out_channels");
    TrainState *t = new TrainState();
    const int64_t np = param_layout(u, t);
    if (n != np) { delete t; set_error("parameter buffer has %ld floats, expected %ld", (long)n, (long)np); return ISX_ERR_ARG; }
    t->P = params; t->G = grads; t->Mo = adam_m; t->Vo = adam_v; t->nparam = np;
    t->cz = (int)chunk[0]; t->cy = (int)chunk[1]; t->cx = (int)chunk[2];
    u->train = t;
    u->batch = 1;
    u->wz = 0;                                        // force a B = 1 workspace
    ISX_TRY(prepare_workspace(ctx, t->cz, t->cy, t->cx, 1));
    // current weights (host copies in reference layout) -> P; BN / bias pointers now alias P
    for (int li = 0; li < u->nlayers; ++li) {
        ConvBN &L = u->cv[li];
        const HostTensor &w = u->host[L.conv + ".weight"], &b = u->host[L.conv + ".bias"];
        const HostTensor &g = u->host[L.bn + ".weight"], &be = u->host[L.bn + ".bias"];
        ISX_HIP(hipMemcpyAsync(t->P + t->off_w[li], w.f.data(), w.f.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        ISX_HIP(hipMemcpyAsync(t->P + t->off_b[li], b.f.data(), b.f.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        ISX_HIP(hipMemcpyAsync(t->P + t->off_gamma[li], g.f.data(), g.f.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        ISX_HIP(hipMemcpyAsync(t->P + t->off_beta[li], be.f.data(), be.f.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        if (L.cout >= 8) L.bias = t->P + t->off_b[li];     // thin layers keep their 8-padded bias copy (refreshed below)
        L.gamma = t->P + t->off_gamma[li];
        L.beta = t->P + t->off_beta[li];
    }
    for (int k = 0; k < 4; ++k) {
        UpConv &U = u->up[k];
        const HostTensor &w = u->host[U.name + ".weight"], &b = u->host[U.name + ".bias"];
        ISX_HIP(hipMemcpyAsync(t->P + t->off_upw[k], w.f.data(), w.f.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        ISX_HIP(hipMemcpyAsync(t->P + t->off_upb[k], b.f.data(), b.f.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        U.bias = t->P + t->off_upb[k];
    }
    // backward buffers
    size_t max_rawc = 0;
    for (int li = 0; li < u->nlayers; ++li) {
        const ConvBN &L = u->cv[li];
        const Level &l = u->lv[MOD_LEVEL[layer_module(li)]];
        t->dstride[li] = L.cout;
        if (use_fold(L.cin, L.cout) && L.cin % 32 == 0) {
            // thin layer fed by a wide one: its d(raw) is stored 8 channels wide (zeros after cout) so that the
            // dgrad into the wide input runs on the MFMA kernel (8 -> Cin) instead of a scalar loop
            t->dstride[li] = 8;
            ISX_TRY(talloc(t, (size_t)l.M * 8, &t->draw[li]));
            ISX_HIP(hipMemsetAsync(t->draw[li], 0, (size_t)l.M * 8 * sizeof(float), ctx->stream));
            ISX_TRY(talloc(t, (size_t)8 * L.cin * 27 + 2048, &t->w_dgrad[li]));
            ISX_HIP(hipMemsetAsync(t->w_dgrad[li], 0, ((size_t)8 * L.cin * 27 + 2048) * sizeof(float), ctx->stream));
        } else {
            ISX_TRY(talloc(t, (size_t)l.M * L.cout, &t->draw[li]));
            if (li != 0 && use_mfma(L.cout, L.cin)) ISX_TRY(talloc(t, (size_t)L.cout * L.cin * 27 + 2048, &t->w_dgrad[li]));
        }
        ISX_TRY(talloc(t, (size_t)l.M * L.cout, &t->gact[li]));
        max_rawc = std::max(max_rawc, (size_t)L.cout);
    }
    for (int k = 0; k < 4; ++k) {
        for (int f = 0; f < u->nforks; ++f) ISX_TRY(talloc(t, (size_t)u->lv[3 - k].M * 2 * u->up[k].C, &t->dcat[f][k]));
        ISX_TRY(talloc(t, (size_t)u->lv[k + 1].M * MOD_COUT[k], &t->dpool[k]));
    }
    // bf16 packs (refreshed on the device after every optimiser step while the context is in ISX_PRECISION_BF16)
    for (int li = 0; li < u->nlayers; ++li) {
        ConvBN &L = u->cv[li];
        if (!use_mfma(L.cin, L.cout) || L.cin % 8) continue;
        const Level &l = u->lv[MOD_LEVEL[layer_module(li)]];
        int t0, t1, t2, slots;
        conv_tiles(l, conv_mode(l), 1, &t0, &t1, &t2, &slots);
        for (int bn = 32; bn <= 64; bn += 32) {
            if (L.cout % bn) continue;
            float *p = nullptr;
            const size_t halves = h2_pack_halves(L.cout, L.cin, bn);
            ISX_TRY(talloc(t, (halves + 1) / 2, &p));
            ISX_HIP(hipMemsetAsync(p, 0, halves * 2, ctx->stream));
            (bn == 64 ? t->wbf_fwd64[li] : t->wbf_fwd32[li]) = p;
            (bn == 64 ? L.w_bf_64 : L.w_bf_32) = p;              // the forward reads the refreshed packs from now on
        }
        if (li != 0 && L.cin % 32 == 0 && L.cout % 8 == 0) {     // dgrad operator: Cout -> Cin
            // the tile width launch_conv_h2 will pick for this operator (batch 1, L.cin output columns): the 512-thread kernel
            // reads 64-column packs wherever it runs.  (Round 3 packed by conv_h2_bn alone: at patch (10, 256, 256) the
            // 256 -> 64 layer's data gradient ran on the 512-thread kernel over a 32-column pack -- garbage that no test saw,
            // the layer-by-layer comparison ran at 64^2 only.)
            const int bn = conv_h2_nw8(ctx->tune, l, L.cin, 1) ? 64 : conv_h2_bn(ctx->tune, slots, L.cin);
            float *p = nullptr;
            const size_t halves = h2_pack_halves(L.cin, L.cout, bn);
            ISX_TRY(talloc(t, (halves + 1) / 2, &p));
            ISX_HIP(hipMemsetAsync(p, 0, halves * 2, ctx->stream));
            t->wbf_dg[li] = p;
            t->wbf_dg_bn[li] = bn;
        }
    }
    t->part_floats = (size_t)4 << 20;
    ISX_TRY(talloc(t, t->part_floats, &t->part));
    ISX_TRY(talloc(t, 1024, &t->coef));
    ISX_TRY(talloc(t, 4096, &t->lossbuf));
    ISX_TRY(talloc(t, 1024, &t->zeros));
    ISX_HIP(hipMemsetAsync(t->zeros, 0, 1024 * 4, ctx->stream));
    size_t wp = 0;
    for (int li = 0; li < u->nlayers; ++li) {
        const ConvBN &L = u->cv[li];
        const size_t tco = L.cout <= 32 ? 32 : 64, tci = L.cin <= 32 ? 32 : 64;       // as in wgrad()
        size_t tiles = ((L.cout + tco - 1) / tco) * ((L.cin + tci - 1) / tci);
        size_t splits = (size_t)ctx->tune.wgrad_wgs / (tiles * 18) + 1;
        if (splits > 256) splits = 256;
        wp = std::max(wp, (splits + 1) * 27 * tiles * tco * tci);
    }
    t->wpart_floats = wp;
    ISX_TRY(talloc(t, wp, &t->wpart));
    ISX_TRY(repack_all(ctx));
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    return ISX_OK;
}

// forward of one patch + loss (+ backward + n_adam optimiser steps).  x: device (D,H,W); y: device (C,D,H,W).
// loss_out (host): [0] = loss, [1..C] = per-channel mean of the elementwise loss term.
// shared body: loss_kind < 0 -> the per-channel (kind, weight) spec in `spec`; 3 -> DiceLoss
static int train_step_impl(isx_ctx *ctx, const float *x, const float *y, int dice, LossSpec spec, float lr,
                           int n_adam, int do_backward, float *loss_out)
{
    ISX_HIP(hipSetDevice(ctx->device));
    UNetState *u = ctx->unet;
    TrainState *t = u->train;
    // the context's U-Net workspace may have been re-shaped by an inference call with another chunk geometry since the
    // last step (the trained UNet object is handed back to the caller): the backward buffers are sized for the patch
    // bound at isx_train_bind, so the forward workspace is brought back to it first (a no-op when nothing changed)
    ISX_TRY(prepare_workspace(ctx, t->cz, t->cy, t->cx, 1));
    ISX_TRY(train_ensure_bf16_packs(ctx));      // precision switched since the last update
    const Level &l0 = u->lv[0];
    const int C = u->out_channels;
    ISX_REQUIRE(C <= 8, "at most 8 output channels");
    // forward (train-mode BN, keeps every raw tensor)
    ISX_HIP(hipMemcpyAsync(u->x0, x, (size_t)l0.M * 4, hipMemcpyDeviceToDevice, ctx->stream));
    Scratch<int32_t> tab(ctx, 9);
    if (!tab.ok()) return ISX_ERR_NOMEM;
    ISX_HIP(hipMemsetAsync(tab.p, 0, 36, ctx->stream));
    u->d_table = tab.p;
    ISX_TRY(forward_batch(ctx, 1, nullptr, nullptr, nullptr));
    // loss and its gradient with respect to the sigmoid output
    const int blocks = (int)((l0.M + 255) / 256);
    float hs[8] = {0};
    double total = 0.0;
    if (!dice) {
        ISX_REQUIRE((size_t)blocks * C <= t->part_floats, "internal: partial buffer too small");
        // every decoder's channels against their slice of the target (the reference concatenates the forks' outputs and
        // applies one loss to the whole tensor, unet.py:322 + train.py:344-363: the means in spec.w are over all channels)
        int c_off = 0;
        int *bad = (int *)(t->lossbuf + 2048);
        bool range_check = false;                         // a BCE term on a decoder whose activation can leave [0, 1]
        for (int f = 0, c0 = 0; f < u->nforks; c0 += u->fork_channels[f], ++f)
            if (u->final_act[f] != ISX_FINAL_SIGMOID && u->final_act[f] != ISX_FINAL_SOFTMAX)
                for (int c = 0; c < u->fork_channels[f]; ++c) range_check = range_check || spec.kind[c0 + c] == 0;
        if (range_check) ISX_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
        for (int f = 0; f < u->nforks; ++f) {
            const int lf = dec_layer(f, 3, 1);
            const ConvBN &Lf = u->cv[lf];
            LossSpec sp;
            for (int c = 0; c < 8; ++c) { sp.kind[c] = c_off + c < 8 ? spec.kind[c_off + c] : 0; sp.w[c] = c_off + c < 8 ? spec.w[c_off + c] : 0.f; }
            loss_kernel<<<blocks, 256, 0, ctx->stream>>>(u->raw[lf], Lf.scale, Lf.shift, y + (int64_t)c_off * l0.M, Lf.cout, l0.M, sp,
                                                         u->final_act[f], t->gact[lf], t->part, bad);
            ISX_CHECK_LAUNCH();
            colsum_kernel<<<Lf.cout, 256, 0, ctx->stream>>>(t->part, blocks, Lf.cout, t->lossbuf + c_off);
            ISX_CHECK_LAUNCH();
            c_off += Lf.cout;
        }
        ISX_HIP(hipMemcpyAsync(hs, t->lossbuf, C * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (range_check) {
            // torch raises inside nn.BCELoss, before backward: parameters and optimiser state stay as they were
            int hb = 0;
            ISX_HIP(hipMemcpyAsync(&hb, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            ISX_HIP(hipStreamSynchronize(ctx->stream));
            ISX_REQUIRE(!hb, "all elements of input should be between 0 and 1 (BCELoss on a decoder whose final activation is not a "
                             "sigmoid / softmax)");
        }
    } else {
        // every decoder's channels against their slice of the target, like the other losses; the reference applies ONE DiceLoss
        // to the concatenated output (unet.py:322, custom_loss.py:24-53), so the mean runs over all C channels
        const int64_t plane = (int64_t)l0.H * l0.W;
        ISX_REQUIRE(plane % 256 == 0 && l0.D * 3 * C <= 2048, "internal: DiceLoss needs H*W %% 256 == 0");
        int c_off = 0;
        for (int f = 0; f < u->nforks; ++f) {
            const int lf = dec_layer(f, 3, 1);
            const ConvBN &Lf = u->cv[lf];
            const int Cf = Lf.cout;
            ISX_REQUIRE((size_t)blocks * Cf * 3 <= t->part_floats, "internal: partial buffer too small");
            dice_sums_kernel<<<blocks, 256, 0, ctx->stream>>>(u->raw[lf], Lf.scale, Lf.shift, y + (int64_t)c_off * l0.M, Cf, l0.M, u->final_act[f], t->part);
            ISX_CHECK_LAUNCH();
            colsum_group_kernel<<<dim3(3 * Cf, l0.D), 256, 0, ctx->stream>>>(t->part, (int)(plane / 256), 3 * Cf, t->lossbuf + 128);
            ISX_CHECK_LAUNCH();
            std::vector<float> sums((size_t)l0.D * 3 * Cf);
            ISX_HIP(hipMemcpyAsync(sums.data(), t->lossbuf + 128, sums.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
            ISX_HIP(hipStreamSynchronize(ctx->stream));
            DiceCoef k;
            for (int c = 0; c < Cf; ++c) {
                double I = 0, A = 0, Bq = 0, zlog = 0;
                for (int z = 0; z < l0.D; ++z) {
                    const double i = sums[(size_t)z * 3 * Cf + 3 * c], a2 = sums[(size_t)z * 3 * Cf + 3 * c + 1],
                                 b2 = sums[(size_t)z * 3 * Cf + 3 * c + 2];
                    I += i; A += a2; Bq += b2;
                    zlog += 1.0 - (2.0 * i + 1.0) / (a2 + b2 + 1.0);
                }
                const double S = A + Bq + 1.0;
                total += (1.0 - (2.0 * I + 1.0) / S) / C;
                k.ka[c] = (float)(-2.0 / (C * S));
                k.kb[c] = (float)((2.0 * I + 1.0) / (C * S * S));
                // the reference's per-channel log applies DiceLoss to (1, Z, Y, X): the z planes act as channels
                hs[c_off + c] = (float)(zlog / l0.D);
            }
            dice_grad_kernel<<<blocks, 256, 0, ctx->stream>>>(y + (int64_t)c_off * l0.M, Cf, l0.M, k, u->raw[lf], Lf.scale, Lf.shift, u->final_act[f],
                                                              t->gact[lf]);
            ISX_CHECK_LAUNCH();
            c_off += Cf;
        }
    }
    if (do_backward) {
        ISX_TRY(backward(ctx));
        for (int s = 0; s < n_adam; ++s) {
            ++t->adam_t;
            const float bc1 = 1.f - powf(0.9f, (float)t->adam_t), bc2 = 1.f - powf(0.999f, (float)t->adam_t);
            adam_kernel<<<2048, 256, 0, ctx->stream>>>(t->P, t->G, t->Mo, t->Vo, t->nparam, lr, bc1, bc2);
            ISX_CHECK_LAUNCH();
        }
        if (n_adam > 0) ISX_TRY(repack_all(ctx));
    }
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    if (!dice) {
        for (int c = 0; c < C; ++c) {
            loss_out[1 + c] = hs[c] / (float)l0.M;
            total += (double)spec.w[c] * (double)hs[c];
        }
    } else {
        for (int c = 0; c < C; ++c) loss_out[1 + c] = hs[c];
    }
    loss_out[0] = (float)total;
    return ISX_OK;
}

int isx_train_step(isx_ctx *ctx, const float *x, const float *y, int loss_kind, const float *chan_weights, float lr,
                   int n_adam, int do_backward, float *loss_out)
{
    ISX_REQUIRE(ctx && x && y && loss_out, "null argument");
    if (!ctx->unet || !ctx->unet->train) { set_error("call isx_train_bind first"); return ISX_ERR_STATE; }
    ISX_REQUIRE(loss_kind >= 0 && loss_kind <= 3 && (loss_kind != 2 || chan_weights), "loss kind");
    const int C = ctx->unet->out_channels;
    const TrainState *tb = ctx->unet->train;                   // the bound patch, not whatever the workspace currently holds
    const double MC = (double)tb->cz * tb->cy * tb->cx * C;
    LossSpec spec;
    for (int c = 0; c < 8; ++c) {
        spec.kind[c] = loss_kind == 1 ? 1 : 0;
        spec.w[c] = (float)((loss_kind == 2 && c < C ? (double)chan_weights[c] : 1.0) / MC);
    }
    return train_step_impl(ctx, x, y, loss_kind == 3, spec, lr, n_adam, do_backward, loss_out);
}

int isx_train_step_channelwise(isx_ctx *ctx, const float *x, const float *y, const int32_t *chan_kind,
                               const float *chan_weight, float lr, int n_adam, int do_backward, float *loss_out)
{
    ISX_REQUIRE(ctx && x && y && loss_out && chan_kind && chan_weight, "null argument");
    if (!ctx->unet || !ctx->unet->train) { set_error("call isx_train_bind first"); return ISX_ERR_STATE; }
    const int C = ctx->unet->out_channels;
    const TrainState *tb = ctx->unet->train;                   // the bound patch, not whatever the workspace currently holds
    const double M = (double)tb->cz * tb->cy * tb->cx;
    LossSpec spec;
    for (int c = 0; c < 8; ++c) {
        spec.kind[c] = 0;
        spec.w[c] = 0.f;
        if (c < C) {
            ISX_REQUIRE(chan_kind[c] == ISX_LOSS_BCE || chan_kind[c] == ISX_LOSS_MSE, "channel %d: loss kind %d", c, chan_kind[c]);
            spec.kind[c] = chan_kind[c];
            spec.w[c] = (float)((double)chan_weight[c] / M);
        }
    }
    return train_step_impl(ctx, x, y, 0, spec, lr, n_adam, do_backward, loss_out);
}

// Gradient tensors of the LAST backward pass, channels-last [M][C] float32, copied to the host (tests: the per-layer pins
// of the bf16 data-gradient convolutions).  name: "draw:<li>" = d(loss)/d(raw conv output) of layer li (C = its stored
// channel stride), "gact:<li>" = d(loss)/d(activation output of layer li) as the layer above delivered it,
// "dpool:<k>" = gradient of the pooled input of encoder module k + 1, "dcat:<k>" = gradient of decoder stage k's
// concatenated input (fork 0).  shape_out = {M, C}.
int isx_train_tap_host(isx_ctx *ctx, const char *name, float *out, int64_t cap_elems, int64_t shape_out[2])
{
    if (!ctx || !ctx->unet || !ctx->unet->train) { set_error("call isx_train_bind first"); return ISX_ERR_STATE; }
    ISX_REQUIRE(name && out && shape_out, "null argument");
    ISX_HIP(hipSetDevice(ctx->device));
    UNetState *u = ctx->unet;
    TrainState *t = u->train;
    const std::string nm(name);
    const size_t colon = nm.find(':');
    ISX_REQUIRE(colon != std::string::npos, "tap name '%s': expected <kind>:<index>", name);
    const std::string kind = nm.substr(0, colon);
    const int idx = atoi(nm.c_str() + colon + 1);
    const float *src = nullptr;
    int64_t M = 0, C = 0;
    if ((kind == "draw" || kind == "gact") && idx >= 0 && idx < u->nlayers) {
        const Level &l = u->lv[MOD_LEVEL[layer_module(idx)]];
        M = l.M;
        C = kind == "draw" ? t->dstride[idx] : u->cv[idx].cout;
        src = kind == "draw" ? t->draw[idx] : t->gact[idx];
    } else if (kind == "dpool" && idx >= 0 && idx < 4) {
        M = u->lv[idx + 1].M; C = MOD_COUT[idx]; src = t->dpool[idx];
    } else if (kind == "dcat" && idx >= 0 && idx < 4) {
        M = u->lv[3 - idx].M; C = 2 * u->up[idx].C; src = t->dcat[0][idx];
    }
    ISX_REQUIRE(src, "unknown gradient tap '%s'", name);
    ISX_REQUIRE(cap_elems >= M * C, "tap buffer too small: need %ld floats", (long)(M * C));
    ISX_HIP(hipMemcpyAsync(out, src, (size_t)(M * C) * 4, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    shape_out[0] = M; shape_out[1] = C;
    return ISX_OK;
}

// after an external update of the bound parameter buffer (e.g. an all-reduce + optimiser step done elsewhere)
int isx_train_refresh_weights(isx_ctx *ctx)
{
    if (!ctx || !ctx->unet || !ctx->unet->train) { set_error("call isx_train_bind first"); return ISX_ERR_STATE; }
    ISX_HIP(hipSetDevice(ctx->device));
    return repack_all(ctx);
}

// one Adam step on the bound buffers (used when gradients were all-reduced between backward and update)
int isx_adam_step(isx_ctx *ctx, float lr, int n_steps)
{
    if (!ctx || !ctx->unet || !ctx->unet->train) { set_error("call isx_train_bind first"); return ISX_ERR_STATE; }
    ISX_HIP(hipSetDevice(ctx->device));
    TrainState *t = ctx->unet->train;
    for (int s = 0; s < n_steps; ++s) {
        ++t->adam_t;
        const float bc1 = 1.f - powf(0.9f, (float)t->adam_t), bc2 = 1.f - powf(0.999f, (float)t->adam_t);
        adam_kernel<<<2048, 256, 0, ctx->stream>>>(t->P, t->G, t->Mo, t->Vo, t->nparam, lr, bc1, bc2);
        ISX_CHECK_LAUNCH();
    }
    return repack_all(ctx);
}

}  // extern "C"
