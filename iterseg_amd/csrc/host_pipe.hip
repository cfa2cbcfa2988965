// host_pipe.hip -- the host-buffer side of the path: what the reference's per-frame loop does around the processing
// function (segmentation.py:833-900: one float32 (Z,Y,X) frame in, one uint32 (Z+2,Y+2,X+2) label block out, frame after
// frame of a (t,z,y,x) stack) without leaving the GPU idle while frames cross PCIe.
//
//   * isx_segment_volume_host   one frame, synchronous (the plugin's processing function, segmentation.py:147-195): the
//                               frame goes through a context-owned PINNED staging buffer filled by several host threads
//                               (pageable hipMemcpy is a single-threaded bounce copy), current_output is checked for
//                               "all zero" by the same threads (it is, on the plugin path: segmentation.py:890) and then
//                               never uploaded; labels come back through the second pinned buffer.
//   * isx_pipe_*                a stack: frame t+1 is staged and uploaded, and the labels of frame t-1 are downloaded and
//                               copied out, on a copy stream and an I/O thread while the compute thread runs frame t.
//
// Threads: the pipe owns one compute thread (it alone drives the context between isx_pipe_create and
// isx_pipe_destroy) and one I/O thread; staging copies fan out over short-lived helper threads.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <thread>

#include "unet_state.h"

namespace isx {

static int host_threads()
{
    static const int n = [] {
        int v = 8;
        if (const char *e = getenv("ISX_HOST_THREADS")) v = atoi(e);
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && v > hw) v = hw;
        return v < 1 ? 1 : v;
    }();
    return n;
}

// fn(begin, end) over [0, n) in contiguous pieces on host threads (the caller runs the first piece)
static void parallel_ranges(int64_t n, int64_t min_piece, const std::function<void(int64_t, int64_t)> &fn)
{
    int nt = host_threads();
    if (n / std::max<int64_t>(min_piece, 1) < nt) nt = (int)std::max<int64_t>(1, n / std::max<int64_t>(min_piece, 1));
    if (nt <= 1) { fn(0, n); return; }
    const int64_t per = (n + nt - 1) / nt;
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    for (int t = 1; t < nt; ++t) {
        const int64_t a = std::min<int64_t>(n, t * per), b = std::min<int64_t>(n, a + per);
        if (a < b) th.emplace_back([&fn, a, b] { fn(a, b); });
    }
    fn(0, std::min<int64_t>(n, per));
    for (auto &t : th) t.join();
}

void host_copy(void *dst, const void *src, size_t bytes)
{
    parallel_ranges((int64_t)bytes, 8 << 20, [&](int64_t a, int64_t b) { memcpy((char *)dst + a, (const char *)src + a, (size_t)(b - a)); });
}

bool host_all_zero(const void *p, size_t bytes)
{
    std::atomic<bool> nz(false);
    parallel_ranges((int64_t)(bytes / 8), 1 << 20, [&](int64_t a, int64_t b) {
        const uint64_t *q = (const uint64_t *)p;
        for (int64_t i = a; i < b && !nz.load(std::memory_order_relaxed); i += 4096) {
            uint64_t acc = 0;
            const int64_t e = std::min<int64_t>(b, i + 4096);
            for (int64_t j = i; j < e; ++j) acc |= q[j];
            if (acc) { nz.store(true); return; }
        }
    });
    if (nz) return false;
    const unsigned char *c = (const unsigned char *)p;
    for (size_t i = bytes / 8 * 8; i < bytes; ++i) if (c[i]) return false;
    return true;
}

// interior (Z,Y,X) of a padded (Z+2,Y+2,X+2) 32-bit block -> contiguous (Z,Y,X)
static void host_copy_cropped(uint32_t *dst, const uint32_t *src_padded, const int64_t shape[3])
{
    const int64_t Z = shape[0], Y = shape[1], X = shape[2], PY = Y + 2, PX = X + 2;
    parallel_ranges(Z * Y, 4096, [&](int64_t a, int64_t b) {
        for (int64_t r = a; r < b; ++r) {
            const int64_t z = r / Y, y = r - z * Y;
            memcpy(dst + r * X, src_padded + ((z + 1) * PY + (y + 1)) * PX + 1, (size_t)X * 4);
        }
    });
}

}  // namespace isx

// pinned staging owned by the context (grown on demand, freed with the context)
void *isx_ctx::pinned(int which, size_t bytes)
{
    if (pinned_bytes[which] >= bytes) return pinned_buf[which];
    if (pinned_buf[which]) { (void)hipHostFree(pinned_buf[which]); pinned_buf[which] = nullptr; pinned_bytes[which] = 0; }
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        isx::set_error("hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    pinned_buf[which] = p;
    pinned_bytes[which] = bytes;
    return p;
}

void isx_ctx::pinned_free()
{
    for (int i = 0; i < 2; ++i) {
        if (pinned_buf[i]) (void)hipHostFree(pinned_buf[i]);
        pinned_buf[i] = nullptr;
        pinned_bytes[i] = 0;
    }
}

struct isx_pipe {
    isx_ctx *ctx = nullptr;
    int64_t shape[3], chunk[3], margin[3];
    int64_t n = 0, pn = 0;
    int depth = 2;
    hipStream_t copy_stream = nullptr;
    struct Slot {
        float *dvol = nullptr;
        uint32_t *dlab = nullptr;
        float *hin = nullptr;             // pinned
        uint32_t *hout = nullptr;         // pinned
        hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;
        // 0 free, 1 submitted (staging/upload queued), 2 uploaded (enqueued), 3 computed + download enqueued, 4 downloaded
        int state = 0;
        int rc = ISX_OK;
        int flags = 0;
        const float *src = nullptr;
        std::string err;
    };
    std::vector<Slot> slots;
    float *fm = nullptr;
    int64_t n_submitted = 0, n_waited = 0;
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    std::deque<std::function<void()>> io_q;
    std::deque<int> compute_q;
    std::thread io_thread, compute_thread;
    int fatal = ISX_OK;
    std::string fatal_msg;
};

namespace isx {

static void pipe_io_loop(isx_pipe *p)
{
    (void)hipSetDevice(p->ctx->device);
    for (;;) {
        std::function<void()> job;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->stop || !p->io_q.empty(); });
            if (p->io_q.empty()) return;
            job = std::move(p->io_q.front());
            p->io_q.pop_front();
        }
        job();
    }
}

static void pipe_compute_loop(isx_pipe *p)
{
    (void)hipSetDevice(p->ctx->device);
    for (;;) {
        int si;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->stop || (!p->compute_q.empty() && p->slots[p->compute_q.front()].state == 2); });
            if (p->stop && (p->compute_q.empty() || p->slots[p->compute_q.front()].state != 2)) return;
            si = p->compute_q.front();
            p->compute_q.pop_front();
        }
        isx_pipe::Slot &s = p->slots[si];
        isx_ctx *ctx = p->ctx;
        int rc = s.rc;                      // an upload that failed ends the frame here
        std::string err = s.err;
        auto fail = [&](const char *what, hipError_t e) { rc = ISX_ERR_HIP; err = std::string(what) + ": " + hipGetErrorString(e); };
        hipError_t e = hipSuccess;
        if (rc == ISX_OK && (e = hipStreamWaitEvent(ctx->stream, s.ev_in, 0)) != hipSuccess) fail("hipStreamWaitEvent", e);
        if (rc == ISX_OK && (s.flags & ISX_PIPE_NORMALISE)) {
            rc = isx_normalise_volume(ctx, s.dvol, p->n);
            if (rc != ISX_OK) err = isx_last_error();
        }
        if (rc == ISX_OK) {
            e = hipMemsetAsync(s.dlab, 0, (size_t)p->pn * 4, ctx->stream);
            if (e != hipSuccess) fail("hipMemsetAsync", e);
        }
        if (rc == ISX_OK) {
            rc = isx_segment_volume(ctx, s.dvol, p->shape, p->chunk, p->margin, p->fm, s.dlab);
            if (rc != ISX_OK) err = isx_last_error();
        }
        if (rc == ISX_OK || rc == ISX_ERR_NO_SEEDS) {
            // the download runs on the copy stream behind this frame's last kernel, under the next frame's compute
            if ((e = hipEventRecord(s.ev_done, ctx->stream)) != hipSuccess) fail("hipEventRecord", e);
            else if ((e = hipStreamWaitEvent(p->copy_stream, s.ev_done, 0)) != hipSuccess) fail("hipStreamWaitEvent", e);
            else if ((e = hipMemcpyAsync(s.hout, s.dlab, (size_t)p->pn * 4, hipMemcpyDeviceToHost, p->copy_stream)) != hipSuccess) fail("hipMemcpyAsync", e);
            else if ((e = hipEventRecord(s.ev_out, p->copy_stream)) != hipSuccess) fail("hipEventRecord", e);
        }
        {
            std::lock_guard<std::mutex> lk(p->mu);
            s.rc = rc;
            s.err = err;
            s.state = 3;
        }
        p->cv.notify_all();
    }
}

}  // namespace isx

extern "C" {

int isx_segment_volume_host(isx_ctx *ctx, const float *input_volume, const int64_t shape[3], const int64_t chunk[3],
                            const int64_t margin[3], uint32_t *current_output)
{
    ISX_REQUIRE(ctx && ctx->unet && ctx->unet->committed, "no network loaded");
    ISX_REQUIRE(input_volume && shape && chunk && margin && current_output, "null argument");
    ISX_HIP(hipSetDevice(ctx->device));
    const int64_t n = shape[0] * shape[1] * shape[2];
    const int64_t pn = (shape[0] + 2) * (shape[1] + 2) * (shape[2] + 2);
    ISX_REQUIRE(n > 0, "empty volume");
    isx::Scratch<float> vol(ctx, n), fm(ctx, (size_t)ctx->unet->out_channels * n);
    isx::Scratch<uint32_t> lab(ctx, pn);
    if (!vol.ok() || !fm.ok() || !lab.ok()) return ISX_ERR_NOMEM;
    float *hin = (float *)ctx->pinned(0, (size_t)n * 4);
    uint32_t *hout = (uint32_t *)ctx->pinned(1, (size_t)pn * 4);
    if (!hin || !hout) return ISX_ERR_NOMEM;
    isx::host_copy(hin, input_volume, (size_t)n * 4);
    ISX_HIP(hipMemcpyAsync(vol.p, hin, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    // current_output is all zero on the plugin path (segment_single_volume allocates it, segmentation.py:890); labels a
    // caller put there are kept (isx_affinity_watershed's contract), which costs the upload
    if (isx::host_all_zero(current_output, (size_t)pn * 4)) {
        ISX_HIP(hipMemsetAsync(lab.p, 0, (size_t)pn * 4, ctx->stream));
    } else {
        isx::host_copy(hout, current_output, (size_t)pn * 4);
        ISX_HIP(hipMemcpyAsync(lab.p, hout, (size_t)pn * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    int rc = isx_segment_volume(ctx, vol.p, shape, chunk, margin, fm.p, lab.p);
    if (rc != ISX_OK && rc != ISX_ERR_NO_SEEDS) return rc;
    ISX_HIP(hipMemcpyAsync(hout, lab.p, (size_t)pn * 4, hipMemcpyDeviceToHost, ctx->stream));
    ISX_HIP(hipStreamSynchronize(ctx->stream));
    isx::host_copy(current_output, hout, (size_t)pn * 4);
    return rc;
}

int isx_pipe_create(isx_ctx *ctx, const int64_t shape[3], const int64_t chunk[3], const int64_t margin[3], int depth,
                    isx_pipe **out)
{
    ISX_REQUIRE(ctx && ctx->unet && ctx->unet->committed, "no network loaded");
    ISX_REQUIRE(shape && chunk && margin && out, "null argument");
    ISX_REQUIRE(depth >= 1 && depth <= 4, "depth must be 1..4");
    ISX_REQUIRE(isx_make_chunks(shape, chunk, margin, nullptr, 0) > 0, "bad chunk geometry");
    ISX_HIP(hipSetDevice(ctx->device));
    isx_pipe *p = new isx_pipe();
    p->ctx = ctx;
    p->depth = depth;
    for (int d = 0; d < 3; ++d) { p->shape[d] = shape[d]; p->chunk[d] = chunk[d]; p->margin[d] = margin[d]; }
    p->n = shape[0] * shape[1] * shape[2];
    p->pn = (shape[0] + 2) * (shape[1] + 2) * (shape[2] + 2);
    p->slots.resize(depth);
    bool ok = hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMalloc((void **)&p->fm, (size_t)ctx->unet->out_channels * p->n * 4) == hipSuccess;
    for (auto &s : p->slots) {
        ok = ok && hipMalloc((void **)&s.dvol, (size_t)p->n * 4) == hipSuccess;
        ok = ok && hipMalloc((void **)&s.dlab, (size_t)p->pn * 4) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&s.hin, (size_t)p->n * 4, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&s.hout, (size_t)p->pn * 4, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.ev_in, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&s.ev_out, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        isx::set_error("isx_pipe_create: allocation of the frame slots failed (%d x %.2f GB device, %.2f GB pinned)", depth,
                       (p->n + p->pn) * 4e-9, (p->n + p->pn) * 4e-9);
        isx_pipe_destroy(p);
        return ISX_ERR_NOMEM;
    }
    p->io_thread = std::thread(isx::pipe_io_loop, p);
    p->compute_thread = std::thread(isx::pipe_compute_loop, p);
    *out = p;
    return ISX_OK;
}

int isx_pipe_submit(isx_pipe *p, const float *frame, int flags)
{
    ISX_REQUIRE(p && frame, "null argument");
    std::unique_lock<std::mutex> lk(p->mu);
    ISX_REQUIRE(p->n_submitted - p->n_waited < p->depth, "pipe full: %d frames in flight, call isx_pipe_wait first", p->depth);
    const int si = (int)(p->n_submitted % p->depth);
    isx_pipe::Slot &s = p->slots[si];
    s.state = 1;
    s.rc = ISX_OK;
    s.flags = flags;
    s.src = frame;
    s.err.clear();
    ++p->n_submitted;
    p->compute_q.push_back(si);
    p->io_q.push_back([p, si] {
        isx_pipe::Slot &s = p->slots[si];
        isx::host_copy(s.hin, s.src, (size_t)p->n * 4);              // the caller keeps `frame` alive until isx_pipe_wait
        hipError_t e = hipMemcpyAsync(s.dvol, s.hin, (size_t)p->n * 4, hipMemcpyHostToDevice, p->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(s.ev_in, p->copy_stream);
        {
            std::lock_guard<std::mutex> lk(p->mu);
            if (e != hipSuccess) { s.rc = ISX_ERR_HIP; s.err = std::string("upload: ") + hipGetErrorString(e); }
            s.state = 2;
        }
        p->cv.notify_all();
    });
    lk.unlock();
    p->cv.notify_all();
    return ISX_OK;
}

int isx_pipe_wait(isx_pipe *p, void *labels, int flags)
{
    ISX_REQUIRE(p && labels, "null argument");
    std::unique_lock<std::mutex> lk(p->mu);
    ISX_REQUIRE(p->n_waited < p->n_submitted, "isx_pipe_wait without a frame in flight");
    const int si = (int)(p->n_waited % p->depth);
    isx_pipe::Slot &s = p->slots[si];
    p->cv.wait(lk, [&] { return s.state >= 3; });
    int rc = s.rc;
    std::string err = s.err;
    lk.unlock();
    if (rc == ISX_OK || rc == ISX_ERR_NO_SEEDS) {
        hipError_t e = hipEventSynchronize(s.ev_out);
        if (e != hipSuccess) { rc = ISX_ERR_HIP; err = std::string("download: ") + hipGetErrorString(e); }
        else if (flags & ISX_PIPE_CROPPED) isx::host_copy_cropped((uint32_t *)labels, s.hout, p->shape);
        else isx::host_copy(labels, s.hout, (size_t)p->pn * 4);
    }
    lk.lock();
    s.state = 0;
    ++p->n_waited;
    lk.unlock();
    if (rc != ISX_OK) isx::set_error("%s", err.empty() ? "frame failed" : err.c_str());
    return rc;
}

void isx_pipe_destroy(isx_pipe *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv.notify_all();
    if (p->io_thread.joinable()) p->io_thread.join();
    if (p->compute_thread.joinable()) p->compute_thread.join();
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    if (p->copy_stream) (void)hipStreamSynchronize(p->copy_stream);
    for (auto &s : p->slots) {
        if (s.dvol) (void)hipFree(s.dvol);
        if (s.dlab) (void)hipFree(s.dlab);
        if (s.hin) (void)hipHostFree(s.hin);
        if (s.hout) (void)hipHostFree(s.hout);
        if (s.ev_in) (void)hipEventDestroy(s.ev_in);
        if (s.ev_done) (void)hipEventDestroy(s.ev_done);
        if (s.ev_out) (void)hipEventDestroy(s.ev_out);
    }
    if (p->fm) (void)hipFree(p->fm);
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    delete p;
}

}  // extern "C"
