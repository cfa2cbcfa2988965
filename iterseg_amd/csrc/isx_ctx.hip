// isx_ctx.hip -- context lifetime, workspace cache, error string, profiling spans and the
// chunk-geometry entry point of libitersegx.so.
#include <cstdarg>
#include <cstdlib>
#include <cstring>

#include "isx_common.h"
#include "unet_state.h"

namespace isx {

static thread_local std::string g_err;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

}  // namespace isx

void *isx_ctx::alloc(size_t bytes)
{
    if (bytes == 0) bytes = 256;
    // round to 1 MiB above 1 MiB, else to 4 KiB: few distinct sizes -> good reuse
    size_t gran = bytes >= (1u << 20) ? (1u << 20) : 4096;
    bytes = (bytes + gran - 1) / gran * gran;
    auto it = free_blocks.lower_bound(bytes);
    if (it != free_blocks.end() && it->first <= bytes + bytes / 4) {
        void *p = it->second;
        size_t sz = it->first;
        free_blocks.erase(it);
        live_blocks[p] = sz;
        return p;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        trim();
        e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            isx::set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
            return nullptr;
        }
    }
    cached_bytes += bytes;
    live_blocks[p] = bytes;
    return p;
}

void isx_ctx::release(void *p)
{
    auto it = live_blocks.find(p);
    if (it == live_blocks.end()) return;
    free_blocks.emplace(it->second, p);
    live_blocks.erase(it);
}

void isx_ctx::trim()
{
    for (auto &kv : free_blocks) {
        (void)hipFree(kv.second);
        cached_bytes -= kv.first;
    }
    free_blocks.clear();
}

int isx_ctx::aux_fork(int i)
{
    if (!aux_streams[i]) {
        ISX_HIP(hipStreamCreateWithFlags(&aux_streams[i], hipStreamNonBlocking));
        ISX_HIP(hipEventCreateWithFlags(&aux_ev[i][0], hipEventDisableTiming));
        ISX_HIP(hipEventCreateWithFlags(&aux_ev[i][1], hipEventDisableTiming));
    }
    ISX_HIP(hipEventRecord(aux_ev[i][0], stream));
    ISX_HIP(hipStreamWaitEvent(aux_streams[i], aux_ev[i][0], 0));
    return ISX_OK;
}

int isx_ctx::aux_join(int i)
{
    ISX_HIP(hipEventRecord(aux_ev[i][1], aux_streams[i]));
    ISX_HIP(hipStreamWaitEvent(stream, aux_ev[i][1], 0));
    return ISX_OK;
}

void isx_ctx::prof_begin(int cls)
{
    auto get = [&]() {
        hipEvent_t e;
        if (!prof_pool.empty()) { e = prof_pool.back(); prof_pool.pop_back(); }
        else (void)hipEventCreate(&e);
        return e;
    };
    ProfSpan s{cls, get(), get()};
    (void)hipEventRecord(s.a, stream);
    prof_spans.push_back(s);
}

void isx_ctx::prof_end()
{
    (void)hipEventRecord(prof_spans.back().b, stream);
    if (prof_spans.size() >= 4096) prof_collect();
}

void isx_ctx::prof_collect()
{
    if (prof_spans.empty()) return;
    (void)hipStreamSynchronize(stream);
    for (auto &s : prof_spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            prof_ms[s.cls] += ms;
            prof_n[s.cls] += 1;
        }
        prof_pool.push_back(s.a);
        prof_pool.push_back(s.b);
    }
    prof_spans.clear();
}

extern "C" {

int isx_version(void) { return ISX_VERSION; }

const char *isx_last_error(void) { return isx::g_err.c_str(); }

int isx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// Development knobs: the environment is read here -- at isx_ctx_create and when a test asks (isx_ctx_reload_knobs) -- and never
// inside a call.
static void read_knobs(isx_ctx *c)
{
    isx_tune t;
    if (const char *e = getenv("ISX_H2_MT2")) t.h2_mt2 = atoi(e);
    if (const char *e = getenv("ISX_BN_RUN_FUSED")) t.bn_run_fused = atoi(e);
    if (const char *e = getenv("ISX_H2_FORCE_BN64")) t.h2_force_bn64 = atoi(e);
    if (const char *e = getenv("ISX_CONV_XCD_REMAP")) t.xcd_remap = atoi(e);
    if (const char *e = getenv("ISX_H2_NJ2")) t.h2_nj2 = atoi(e);
    if (const char *e = getenv("ISX_H2_STRIPS")) t.h2_strips = atoi(e);
    if (const char *e = getenv("ISX_H2_TZ")) t.h2_tz = atoi(e);
    if (const char *e = getenv("ISX_H2_M512")) t.h2_m512 = atoi(e);
    if (const char *e = getenv("ISX_TAIL_UP")) t.tail_up = atoi(e);
    if (const char *e = getenv("ISX_H2_PRESPLIT")) t.h2_presplit = atoi(e);
    if (const char *e = getenv("ISX_H2_UP")) t.h2_up = atoi(e);
    if (const char *e = getenv("ISX_BF16_WGRAD")) t.bf16_wgrad = atoi(e);
    if (const char *e = getenv("ISX_WGRAD_WGS")) t.wgrad_wgs = std::max(18, atoi(e));
    if (const char *e = getenv("ISX_THIN_WGRAD")) t.thin_wgrad = atoi(e);
    if (const char *e = getenv("ISX_PLANAR")) t.planar = atoi(e);
    if (const char *e = getenv("ISX_H2_NW8")) t.h2_nw8 = atoi(e);
    if (const char *e = getenv("ISX_UP_PER_SOURCE")) t.up_per_source = atoi(e);
    if (const char *e = getenv("ISX_CONV_FIRST_NT")) t.conv_first_nt = atoi(e);
    if (const char *e = getenv("ISX_H2_NT_OUT")) t.h2_nt_out = atoi(e);
    c->tune = t;
    isx_flood_tune f;
    auto ll = [](const char *name, long long &dst) { if (const char *e = getenv(name)) dst = (long long)strtoull(e, nullptr, 10); };
    auto in = [](const char *name, int &dst) { if (const char *e = getenv(name)) dst = atoi(e); };
    ll("ISX_FLOOD_COOP_MIN", f.coop_min); ll("ISX_FLOOD_COLD_MIN", f.cold_min); ll("ISX_FLOOD_GROUP_MIN", f.group_min);
    ll("ISX_FLOOD_SMALL_MAX", f.small_max); ll("ISX_FLOOD_ARENA_MB", f.arena_mb);
    in("ISX_FLOOD_RUN_LIMIT", f.run_limit); in("ISX_FLOOD_DRAIN", f.drain); in("ISX_FLOOD_WIN", f.win); in("ISX_FLOOD_LIST", f.list);
    in("ISX_FLOOD_BURSTS", f.bursts); in("ISX_FLOOD_GSTEPS", f.gsteps); in("ISX_FLOOD_GMINSTEPS", f.gminsteps); in("ISX_FLOOD_GWIN", f.gwin);
    in("ISX_FLOOD_GLIST", f.glist); in("ISX_FLOOD_BUCKETS", f.buckets); in("ISX_FLOOD_GWATCHDOG", f.gwatchdog);
    in("ISX_FLOOD_GROUP_STREAM", f.group_stream);
    in("ISX_GAUSS_FUSED", f.gauss_fused);
    in("ISX_WS_CHECK", f.ws_check);
    in("ISX_CCL_FUSED", f.ccl_fused);
    f.stats = getenv("ISX_FLOOD_STATS") != nullptr;
    if (const char *e = getenv("ISX_FLOOD_TRACE")) f.trace = e;
    c->ftune = f;
}

isx_ctx *isx_ctx_create(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        isx::set_error("no HIP device available (%s): libitersegx has no CPU path",
                       e != hipSuccess ? hipGetErrorString(e) : "0 devices");
        return nullptr;
    }
    if (device < 0 || device >= n) {
        isx::set_error("device %d out of range (0..%d)", device, n - 1);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        isx::set_error("hipSetDevice(%d) failed", device);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            isx::set_error("device %d is %s; libitersegx is built for gfx950 only", device,
                           prop.gcnArchName);
            return nullptr;
        }
    }
    isx_ctx *c = new isx_ctx();
    read_knobs(c);
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        isx::set_error("hipStreamCreate failed");
        delete c;
        return nullptr;
    }
    c->stream = c->own_stream;
    return c;
}

int isx_ctx_reload_knobs(isx_ctx *c)
{
    ISX_REQUIRE(c, "null context");
    read_knobs(c);
    if (c->unet) c->unet->tune = c->tune;
    return ISX_OK;
}

void isx_ctx_destroy(isx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) isx::comm_release(c);
    if (c->unet) isx::unet_destroy(c->unet);
    for (auto &kv : c->live_blocks) (void)hipFree(kv.first);
    c->live_blocks.clear();
    c->trim();
    c->pinned_free();
    for (int i = 0; i < 2; ++i) {
        if (c->aux_streams[i]) { (void)hipStreamSynchronize(c->aux_streams[i]); (void)hipStreamDestroy(c->aux_streams[i]); }
        for (auto e : c->aux_ev[i]) if (e) (void)hipEventDestroy(e);
    }
    for (auto &s : c->prof_spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto e : c->prof_pool) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int isx_ctx_set_stream(isx_ctx *c, void *s)
{
    ISX_REQUIRE(c, "null context");
    c->stream = (hipStream_t)s;      // NULL is the HIP null (legacy default) stream
    return ISX_OK;
}

int isx_ctx_synchronize(isx_ctx *c)
{
    ISX_REQUIRE(c, "null context");
    ISX_HIP(hipStreamSynchronize(c->stream));
    return ISX_OK;
}

int isx_ctx_device(const isx_ctx *c) { return c ? c->device : -1; }

int64_t isx_ctx_workspace_bytes(const isx_ctx *c) { return c ? (int64_t)c->cached_bytes : 0; }

void isx_ctx_release_workspace(isx_ctx *c)
{
    if (!c) return;
    (void)hipStreamSynchronize(c->stream);
    c->trim();
}

int isx_prof_enable(isx_ctx *c, int enable)
{
    ISX_REQUIRE(c, "null context");
    c->prof_collect();
    c->prof_on = enable != 0;
    if (!enable) {
        memset(c->prof_ms, 0, sizeof(c->prof_ms));
        memset(c->prof_n, 0, sizeof(c->prof_n));
    }
    return ISX_OK;
}

int isx_prof_get(isx_ctx *c, int cls, double *total_ms, int64_t *launches)
{
    ISX_REQUIRE(c && cls >= 0 && cls < ISX_PROF_NCLASS, "bad profile class");
    c->prof_collect();
    if (total_ms) *total_ms = c->prof_ms[cls];
    if (launches) *launches = c->prof_n[cls];
    return ISX_OK;
}

// make_chunks, src/iterseg/predict.py:38-61 -- same per-axis arithmetic as
// iterseg_amd/chunking.py (kept in C so non-Python hosts get the identical table).
int64_t isx_make_chunks(const int64_t shape[3], const int64_t chunk[3], const int64_t margin[3],
                        int32_t *table, int64_t cap)
{
    std::vector<int64_t> st[3];
    std::vector<std::pair<int64_t, int64_t>> cr[3];
    for (int d = 0; d < 3; ++d) {
        int64_t arr = shape[d], chk = chunk[d], mrg = margin[d], step = chk - 2 * mrg;
        if (arr < chk) { isx::set_error("axis %d: volume %ld smaller than chunk %ld", d, (long)arr, (long)chk); return ISX_ERR_ARG; }
        if (step <= 0 || arr - 2 * mrg <= 0) { isx::set_error("axis %d: margin too large", d); return ISX_ERR_ARG; }
        for (int64_t s = 0; s < arr - 2 * mrg; s += step) st[d].push_back(s);
        st[d].back() = arr - chk;
        if (st[d].size() > 1 && st[d][st[d].size() - 1] == st[d][st[d].size() - 2]) st[d].pop_back();
        cr[d].assign(st[d].size(), {mrg, chk - mrg});
        cr[d][0].first = 0;
        int64_t taken = 0;
        for (size_t i = 0; i + 1 < cr[d].size(); ++i) taken += cr[d][i].second - cr[d][i].first;
        cr[d].back().first = chk - (arr - taken);
        cr[d].back().second = chk;
    }
    int64_t n = (int64_t)(st[0].size() * st[1].size() * st[2].size());
    int64_t k = 0;
    for (size_t a = 0; a < st[0].size(); ++a)
        for (size_t b = 0; b < st[1].size(); ++b)
            for (size_t c = 0; c < st[2].size(); ++c, ++k) {
                if (!table || k >= cap) continue;
                int32_t *r = table + 9 * k;
                r[0] = (int32_t)st[0][a]; r[1] = (int32_t)st[1][b]; r[2] = (int32_t)st[2][c];
                r[3] = (int32_t)cr[0][a].first; r[4] = (int32_t)cr[0][a].second;
                r[5] = (int32_t)cr[1][b].first; r[6] = (int32_t)cr[1][b].second;
                r[7] = (int32_t)cr[2][c].first; r[8] = (int32_t)cr[2][c].second;
            }
    return n;
}

}  // extern "C"
