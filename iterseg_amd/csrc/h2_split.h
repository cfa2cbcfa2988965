// h2_split.h -- the operand split of the split-f16 convolutions (conv_h2.hip): x = x_h + x_l / 2^11.  Shared with the kernels that
// can write an activation tensor already split (unet.hip: pool / transposed convolution), so that its consumer stages it by copy.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace isx {

static constexpr float H2_SCALE = 2048.f, H2_MIN = 6.103515625e-05f, H2_MAX = 65504.f;
// The 27-tap kernels (conv_h2.hip) keep ONE accumulator per output: x = x_h + x_r with the residual x_r = x - x_h stored as it is,
// not scaled by 2^11, so that a_h b_h + a_h b_r + a_r b_h add up in one register.  For the residuals to stay normal halves the
// operands are scaled by powers of two first (exact): activations by H2_SA when they are staged (folded into the BatchNorm scale /
// shift), weights by H2_SW when they are packed; the epilogue multiplies by 1 / (H2_SA H2_SW).  A residual is below 2^-10 of its
// operand, so it is a normal half while the scaled operand is above 2^-4 (activations above 2^-7, weights above 2^-16); smaller
// operands lose nothing that matters (absolute error 2^-24 / scale).  Range: |activation| H2_SA and |weight| H2_SW must stay below
// 65504 -- the per-layer bound check of run_conv (unet.hip) uses H2_ACT_LIMIT / H2_W_LIMIT.
static constexpr float H2_SA = 8.f, H2_SW = 4096.f, H2_ACT_LIMIT = 6.0e4f / H2_SA, H2_W_LIMIT = 6.0e4f / H2_SW;

__host__ __device__ inline void split_h2(float x, _Float16 &h, _Float16 &l)
{
    x = x < -H2_MAX ? -H2_MAX : (x > H2_MAX ? H2_MAX : x);
    h = (x < H2_MIN && x > -H2_MIN) ? (_Float16)0.f : (_Float16)x;
    const float r = (x - (float)h) * H2_SCALE;
    l = (r < H2_MIN && r > -H2_MIN) ? (_Float16)0.f : (_Float16)r;
}

// 8 fp32 -> 8 halves (hi) + 8 halves (lo), each packed in a uint4.  Device staging version of split_h2: the high
// part is rounded toward zero (v_cvt_pkrtz_f16_f32, two values per instruction) -- any rounding works, the low part
// absorbs the remainder -- and subnormal halves are kept (the f16 matrix cores honour them; checked on MI355X with
// scratch/h2/mfma_layout.hip), so only the clamp remains of the host routine's special cases.
__device__ __forceinline__ void split2(float x, float y, uint32_t &hi, uint32_t &lo)
{
    typedef __fp16 h2v __attribute__((ext_vector_type(2)));
    union { h2v v; uint32_t u; } H, L;
    x = __builtin_amdgcn_fmed3f(x, -H2_MAX, H2_MAX);
    y = __builtin_amdgcn_fmed3f(y, -H2_MAX, H2_MAX);
    H.v = __builtin_amdgcn_cvt_pkrtz(x, y);
    const float rx = (x - (float)H.v[0]) * H2_SCALE, ry = (y - (float)H.v[1]) * H2_SCALE;
    L.v = __builtin_amdgcn_cvt_pkrtz(rx, ry);
    hi = H.u;
    lo = L.u;
}

// ... with the residual as it is (single-accumulator kernels; the operands arrive scaled by H2_SA)
__host__ __device__ inline void split_h2r(float x, _Float16 &h, _Float16 &r)
{
    x = x < -H2_MAX ? -H2_MAX : (x > H2_MAX ? H2_MAX : x);
    h = (_Float16)x;
    r = (_Float16)(x - (float)h);
}
__device__ __forceinline__ void split2r(float x, float y, uint32_t &hi, uint32_t &lo)
{
    typedef __fp16 h2v __attribute__((ext_vector_type(2)));
    union { h2v v; uint32_t u; } H, L;
    x = __builtin_amdgcn_fmed3f(x, -H2_MAX, H2_MAX);
    y = __builtin_amdgcn_fmed3f(y, -H2_MAX, H2_MAX);
    H.v = __builtin_amdgcn_cvt_pkrtz(x, y);
    L.v = __builtin_amdgcn_cvt_pkrtz(x - (float)H.v[0], y - (float)H.v[1]);
    hi = H.u;
    lo = L.u;
}
// ... the staging form: the clamp doubles as the ReLU (lo_clamp = 0 behind a BatchNorm, -H2_MAX for a source without one), the
// residual is one mixed-precision FMA, fma(half h, -1, x) = x - h rounded once like the subtraction it replaces (v_fma_mix_f32)
__device__ __forceinline__ void split2r_lean(float x, float y, float lo_clamp, uint32_t &hi, uint32_t &lo)
{
    typedef __fp16 h2v __attribute__((ext_vector_type(2)));
    union { h2v v; uint32_t u; } H, L;
    x = __builtin_amdgcn_fmed3f(x, lo_clamp, H2_MAX);
    y = __builtin_amdgcn_fmed3f(y, lo_clamp, H2_MAX);
    H.v = __builtin_amdgcn_cvt_pkrtz(x, y);
    float rx, ry;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(rx) : "v"(H.u), "v"(x));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(ry) : "v"(H.u), "v"(y));
    L.v = __builtin_amdgcn_cvt_pkrtz(rx, ry);
    hi = H.u;
    lo = L.u;
}
__device__ __forceinline__ void split8r_lean(const float4 &a, const float4 &b, float lo_clamp, uint4 &hi, uint4 &lo)
{
    split2r_lean(a.x, a.y, lo_clamp, hi.x, lo.x);
    split2r_lean(a.z, a.w, lo_clamp, hi.y, lo.y);
    split2r_lean(b.x, b.y, lo_clamp, hi.z, lo.z);
    split2r_lean(b.z, b.w, lo_clamp, hi.w, lo.w);
}

__device__ __forceinline__ void split8r(const float4 &a, const float4 &b, uint4 &hi, uint4 &lo)
{
    split2r(a.x, a.y, hi.x, lo.x);
    split2r(a.z, a.w, hi.y, lo.y);
    split2r(b.x, b.y, hi.z, lo.z);
    split2r(b.z, b.w, hi.w, lo.w);
}

__device__ __forceinline__ void split8(const float4 &a, const float4 &b, uint4 &hi, uint4 &lo)
{
    split2(a.x, a.y, hi.x, lo.x);
    split2(a.z, a.w, hi.y, lo.y);
    split2(b.x, b.y, hi.z, lo.z);
    split2(b.z, b.w, hi.w, lo.w);
}

}  // namespace isx
