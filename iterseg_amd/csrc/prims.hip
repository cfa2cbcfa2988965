// prims.hip -- device-wide sort / scan helpers (rocPRIM) used by the watershed stage for
// the small bookkeeping arrays (seed lists, component tables).  Kept in their own
// translation unit because the rocPRIM headers dominate compile time.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "isx_common.h"
#include "prims.h"

namespace isx {

int sort_u64(isx_ctx *ctx, const uint64_t *in, uint64_t *out, size_t n)
{
    if (n == 0) return ISX_OK;
    size_t tmp_bytes = 0;
    ISX_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, in, out, n, 0, 64, ctx->stream));
    Scratch<char> tmp(ctx, tmp_bytes);
    if (!tmp.ok()) return ISX_ERR_NOMEM;
    ISX_HIP(rocprim::radix_sort_keys(tmp.p, tmp_bytes, in, out, n, 0, 64, ctx->stream));
    return ISX_OK;
}

template <typename T>
static int excl_scan(isx_ctx *ctx, const T *in, T *out, size_t n)
{
    if (n == 0) return ISX_OK;
    size_t tmp_bytes = 0;
    ISX_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, in, out, T(0), n, rocprim::plus<T>(),
                                    ctx->stream));
    Scratch<char> tmp(ctx, tmp_bytes);
    if (!tmp.ok()) return ISX_ERR_NOMEM;
    ISX_HIP(rocprim::exclusive_scan(tmp.p, tmp_bytes, in, out, T(0), n, rocprim::plus<T>(),
                                    ctx->stream));
    return ISX_OK;
}

int exclusive_scan_u32(isx_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n)
{
    return excl_scan<uint32_t>(ctx, in, out, n);
}

int exclusive_scan_u64(isx_ctx *ctx, const uint64_t *in, uint64_t *out, size_t n)
{
    return excl_scan<uint64_t>(ctx, in, out, n);
}

}  // namespace isx
