// comm.hip -- the three exchanges of the sharded path as C entry points on RCCL (xGMI): gradient all-reduce (data-parallel
// training, reference: train.py:344-363 run once per rank), all-gather of per-rank tiles (feature-map z-slabs of one volume,
// label tiles of a stack: segmentation.py:873-882 sharded over frames) and the max-reduce that merges component-sharded
// label volumes.  For hosts that do not already hold a communicator; the Python host of this repository keeps using
// torch.distributed (same RCCL underneath) -- INTEGRATION.md "Collectives".
//
// RCCL is loaded lazily (dlopen) when the first communicator is created: a process that never calls isx_comm_* does not
// load it, and a process that already has it (torch.distributed, backend "nccl") shares that copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>

#include "isx_common.h"

namespace isx {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

static Rccl *rccl()
{
    static Rccl r;                         // C++11: initialised once, thread-safe
    static const bool ok = [] {
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) return false;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
        r.AllGather = (decltype(r.AllGather))dlsym(r.handle, "ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        return r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather && r.GetErrorString;
    }();
    return ok ? &r : nullptr;
}

// dlerror() is NULL when dlopen succeeded and only a dlsym came back empty (or when it was already consumed)
static const char *dl_reason()
{
    const char *e = dlerror();
    return e ? e : "a required nccl* symbol is missing from the library";
}

#define ISX_NCCL(call)                                                                           \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess) { set_error("%s failed: %s", #call, rccl()->GetErrorString(r_)); return ISX_ERR_HIP; } \
    } while (0)

__global__ void scale_kernel(float *p, int64_t n, float f)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] *= f;
}

void comm_release(isx_ctx *c)
{
    if (c->comm && rccl()) (void)rccl()->CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
    c->comm_world = 1;
    c->comm_rank = 0;
}

}  // namespace isx

using namespace isx;

extern "C" {

int isx_comm_unique_id(uint8_t *id)
{
    ISX_REQUIRE(id, "null argument");
    static_assert(sizeof(ncclUniqueId) == ISX_COMM_ID_BYTES, "ncclUniqueId size");
    if (!rccl()) { set_error("RCCL (librccl.so) could not be loaded: %s", dl_reason()); return ISX_ERR_STATE; }
    ncclUniqueId u;
    ISX_NCCL(rccl()->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return ISX_OK;
}

int isx_comm_init(isx_ctx *ctx, const uint8_t *id, int rank, int world)
{
    ISX_REQUIRE(ctx && id, "null argument");
    ISX_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d of %d", rank, world);
    ISX_REQUIRE(!ctx->comm, "this context already has a communicator");
    if (!rccl()) { set_error("RCCL (librccl.so) could not be loaded: %s", dl_reason()); return ISX_ERR_STATE; }
    ISX_HIP(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    ISX_NCCL(rccl()->CommInitRank(&comm, world, u, rank));       // collective over all ranks: one process per GPU
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return ISX_OK;
}

int isx_comm_destroy(isx_ctx *ctx)
{
    ISX_REQUIRE(ctx, "null context");
    ISX_HIP(hipSetDevice(ctx->device));
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    comm_release(ctx);
    return ISX_OK;
}

int isx_allreduce_grads(isx_ctx *ctx, float *grads, int64_t n)
{
    ISX_REQUIRE(ctx && grads && n > 0, "bad argument");
    if (!ctx->comm) { set_error("call isx_comm_init first"); return ISX_ERR_STATE; }
    ISX_HIP(hipSetDevice(ctx->device));
    // one bucket (the flat gradient buffer of isx_train_bind: 39.9 MB): on the fully connected xGMI mesh RCCL runs it as a
    // reduce-scatter + all-gather over all links; then the mean, as DistributedDataParallel does
    ISX_NCCL(rccl()->AllReduce(grads, grads, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    if (ctx->comm_world > 1) {
        scale_kernel<<<1024, 256, 0, ctx->stream>>>(grads, n, 1.0f / (float)ctx->comm_world);
        ISX_CHECK_LAUNCH();
    }
    return ISX_OK;
}

int isx_gather_tiles(isx_ctx *ctx, const void *send, void *recv, int64_t bytes_per_rank)
{
    ISX_REQUIRE(ctx && send && recv && bytes_per_rank > 0, "bad argument");
    if (!ctx->comm) { set_error("call isx_comm_init first"); return ISX_ERR_STATE; }
    ISX_HIP(hipSetDevice(ctx->device));
    ISX_NCCL(rccl()->AllGather(send, recv, (size_t)bytes_per_rank, ncclUint8, (ncclComm_t)ctx->comm, ctx->stream));
    return ISX_OK;
}

int isx_allreduce_max_labels(isx_ctx *ctx, int32_t *labels, int64_t n)
{
    ISX_REQUIRE(ctx && labels && n > 0, "bad argument");
    if (!ctx->comm) { set_error("call isx_comm_init first"); return ISX_ERR_STATE; }
    ISX_HIP(hipSetDevice(ctx->device));
    ISX_NCCL(rccl()->AllReduce(labels, labels, (size_t)n, ncclInt32, ncclMax, (ncclComm_t)ctx->comm, ctx->stream));
    return ISX_OK;
}

}  // extern "C"
