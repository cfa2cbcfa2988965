#include <hip/hip_runtime.h>
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const short *in, float *out)
{
    __shared__ __attribute__((aligned(16))) short s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = in[i];
    __syncthreads();
    const int lane = threadIdx.x, i = lane & 15, q = i >> 2, p = i & 3, g = (lane >> 4) & 1, hh = lane >> 5;
    // rows of 64 shorts (128 B)
    v4s a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s *)(s + (8 * hh + q) * 64 + 16 * g + 4 * p));
    out[lane * 4 + 0] = a[0]; out[lane * 4 + 1] = a[1]; out[lane * 4 + 2] = a[2]; out[lane * 4 + 3] = a[3];
}
int main()
{
    short h[4096]; for (int i = 0; i < 4096; ++i) h[i] = (short)i;     // element (row r, col c) = r * 64 + c
    short *d; float *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 4 * 4); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o); float r[256]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) printf("lane %2d: %4.0f %4.0f %4.0f %4.0f\n", l, r[4 * l], r[4 * l + 1], r[4 * l + 2], r[4 * l + 3]);
    return 0;
}
