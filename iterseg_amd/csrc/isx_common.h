// isx_common.h -- context, error plumbing and the device workspace cache shared by all
// translation units of libitersegx.so.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/itersegx.h"

namespace isx {

void set_error(const char *fmt, ...);

#define ISX_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            isx::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                   \
            return ISX_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

#define ISX_CHECK_LAUNCH()                                                              \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            isx::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_),   \
                           __FILE__, __LINE__);                                         \
            return ISX_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

#define ISX_TRY(expr)                \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != ISX_OK) return rc_; \
    } while (0)

#define ISX_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            isx::set_error(__VA_ARGS__); \
            return ISX_ERR_ARG;         \
        }                               \
    } while (0)

struct UNetState;   // unet.hip

}  // namespace isx

// Size-bucketed cache of hipMalloc'd blocks: the hot path allocates its scratch once and
// then reuses it for every chunk / frame (no hipMalloc inside steady-state steps).
// Development knobs (environment, read ONCE per context at isx_ctx_create): the library keeps no process-global mutable
// state -- everything lives in the handle, so contexts can be driven from different host threads.
struct isx_tune {
    int bn_run_fused = 1;           // ISX_BN_RUN_FUSED: 0 = the running statistics by a launch per layer (rounds 1-4) instead of one per forward pass
    int h2_mt2 = 1;                 // ISX_H2_MT2: 0 never, 1 big launches, 2 always -- two row tiles per wave in conv_h2
    int h2_force_bn64 = 0;          // ISX_H2_FORCE_BN64
    int xcd_remap = 1;              // ISX_CONV_XCD_REMAP
    int h2_nt_out = 1;              // ISX_H2_NT_OUT: streaming stores in the epilogue of the split-f16 convolutions (conv stage -2 ms per frame)
    int conv_first_nt = 1;          // ISX_CONV_FIRST_NT: streaming stores for c0.conv0's 5.4 GB output per launch set (34.7 -> 31.5 ms per frame)
    int up_per_source = 1;          // ISX_UP_PER_SOURCE: 0 = the transposed convolutions one thread per OUTPUT voxel (rounds 1-2)
    int planar = 2;                 // ISX_PLANAR (2: + the skip / pooled / up-sampled tensors, r4): the first convolution of a module writes [C/8][voxel][8] for the second one (inference, reduced-precision kernels)
    int h2_nw8 = 1;                 // ISX_H2_NW8: 1 = 512-thread work-groups (eight waves on ONE weight slab, 256 voxels, slabs by LDS-DMA) for the
                                    // layers with Cout % 64 == 0 whose launch fills the chip; 0 = the 4-wave kernel everywhere; 2 = always (tests)
    int wgrad_wgs = 4096;           // ISX_WGRAD_WGS: work-groups a matrix-core weight gradient aims for (its split count: fewer = fewer partials to write and reduce)
    int thin_wgrad = 1;             // ISX_THIN_WGRAD: 0 = the 1 -> 32 and 5 -> 5 layers' weight gradients on the padded matrix-core tiles (round 2)
    int bf16_wgrad = 1;             // ISX_BF16_WGRAD: 0 = precision "bf16" keeps fp32 products in the weight gradients (round 2's form)
    int h2_nj2 = 1;                 // ISX_H2_NJ2: 1 = two 64-column N tiles per work-group of the LDS-DMA kernel for Cout % 128 == 0 (2: always, tests)
    int h2_m512 = 1;                // ISX_H2_M512: 1 = 512-voxel work-groups (64 x 64 / 64 x 128 wave tiles) for the 512-thread kernels where the launch fills the chip (2: always, tests; 0: off)
    int h2_tz = 1;                  // ISX_H2_TZ: 0 = one plane per tile on odd levels (rounds 1-3) instead of all planes of a shallow level
    int h2_strips = 1;              // ISX_H2_STRIPS: 0 = padded 8 x TXW boxes only (rounds 1-3) instead of edge strips where H / W leave a remainder
    int h2_up = 2;                  // ISX_H2_UP: 0 = up0 .. up2 materialised instead of up-sampled while their consumer stages its halo (conv_h2.hip, UP); 1 = up1 / up2 only
    int h2_presplit = 1;            // ISX_H2_PRESPLIT: 0 = pooled tensors as float values (their reader splits them while staging) instead of pre-split halves
    int tail_up = 1;                // ISX_TAIL_UP: 0 = up3 materialised by upsample_src_kernel instead of up-sampled while the tail kernel stages it
};

// ISX_FLOOD_* (the watershed tests' stress modes, tuning runs): a negative field = the library default (csrc/ws.hip).  Read at
// isx_ctx_create and again by isx_ctx_reload_knobs -- never per call.
struct isx_flood_tune {
    long long coop_min = -1, cold_min = -1, group_min = -1, small_max = -1, arena_mb = -1;
    int run_limit = -1, drain = -1, win = -1, list = -1, bursts = -1, gsteps = -1, gminsteps = -1, gwin = -1, glist = -1, buckets = -1;
    int gwatchdog = 0, group_stream = 1, stats = 0;
    int gauss_fused = 1;            // ISX_GAUSS_FUSED: 0 = one kernel per axis (rounds 1-3) instead of the marching z pass + the fused y + x pass (which also hands Otsu its min / max)
    int ccl_fused = 1;              // ISX_CCL_FUSED: 0 = ccl_flatten_kernel + ccl_sizes_kernel as two passes (rounds 1-4)
    int ws_check = 0;               // ISX_WS_CHECK: 1 = derived quantities (the affinities' value range from the channel extrema) are also measured the slow way and must agree (tests)
    std::string trace;              // ISX_FLOOD_TRACE=<file>: development dump of every component's commit order
};

struct isx_ctx {
    int device = 0;
    isx_tune tune;
    isx_flood_tune ftune;
    bool tail_attr_done = false;    // ... and of conv_tail_kernel
    bool h2_attr_done = false;      // dynamic-LDS attribute of the conv_h2 kernels set for this context's device
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    std::multimap<size_t, void *> free_blocks;
    std::unordered_map<void *, size_t> live_blocks;
    size_t cached_bytes = 0;
    // profiling by kernel class
    bool prof_on = false;
    struct ProfSpan { int cls; hipEvent_t a, b; };
    std::vector<ProfSpan> prof_spans;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[ISX_PROF_NCLASS] = {0};
    int64_t prof_n[ISX_PROF_NCLASS] = {0};
    isx::UNetState *unet = nullptr;
    int ws_shard_rank = 0, ws_shard_world = 1;     // W4 component sharding (isx_ws_set_shard)
    void *comm = nullptr;                          // RCCL communicator (isx_comm_init), ncclComm_t
    int comm_rank = 0, comm_world = 1;
    // side streams of the flood (forked from / joined to `stream`): 0 = the work-group kernel of the giants, 1 = the large
    // tier of the window kernel, so that giants, large and mid-size components flood side by side
    hipStream_t aux_streams[2] = {nullptr, nullptr};
    hipEvent_t aux_ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int aux_fork(int i = 0);
    int aux_join(int i = 0);
    void *pinned_buf[2] = {nullptr, nullptr};      // host staging of isx_segment_volume_host (0: frame in, 1: labels out)
    size_t pinned_bytes[2] = {0, 0};

    void *alloc(size_t bytes);
    void release(void *p);
    void trim();
    void *pinned(int which, size_t bytes);
    void pinned_free();
    void prof_begin(int cls);
    void prof_end();
    void prof_collect();
};

namespace isx {

// RAII device scratch tied to a context.
template <typename T>
struct Scratch {
    isx_ctx *ctx;
    T *p;
    size_t n;
    Scratch(isx_ctx *c, size_t count) : ctx(c), p((T *)c->alloc(count * sizeof(T))), n(count) {}
    ~Scratch() { if (p) ctx->release(p); }
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    operator T *() const { return p; }
    bool ok() const { return p != nullptr; }
    void resize(size_t count)          // contents are not kept
    {
        if (p) ctx->release(p);
        p = (T *)ctx->alloc(count * sizeof(T));
        n = count;
    }
};

struct ProfScope {
    isx_ctx *ctx;
    ProfScope(isx_ctx *c, int cls) : ctx(c) { if (c->prof_on) c->prof_begin(cls); }
    ~ProfScope() { if (ctx->prof_on) ctx->prof_end(); }
};

inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

void unet_destroy(UNetState *);
void comm_release(isx_ctx *c);            // comm.hip

}  // namespace isx
