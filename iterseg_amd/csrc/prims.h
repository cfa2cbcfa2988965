// prims.h -- device-wide sort / scan helpers (implemented in prims.hip on rocPRIM).
#pragma once
#include <cstddef>
#include <cstdint>
struct isx_ctx;
namespace isx {
int sort_u64(isx_ctx *ctx, const uint64_t *in, uint64_t *out, size_t n);
int exclusive_scan_u32(isx_ctx *ctx, const uint32_t *in, uint32_t *out, size_t n);
int exclusive_scan_u64(isx_ctx *ctx, const uint64_t *in, uint64_t *out, size_t n);
}  // namespace isx
