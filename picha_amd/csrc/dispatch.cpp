// dispatch.cpp — asynchronous, batching front end of the host-pointer entry points
// (picha_hip_resize_async / picha_hip_color_convert_async, include/picha_hip.h).
//
// Replaces the reference's "one image per libuv worker" dispatch (src/resize.cc:362-364,
// src/colorconvert.cc:252-254).  Per device there is one dispatcher:
//
//   callers ──enqueue──► queue ──► stager thread ──► slot k: [pinned host | device arena | stream | event]
//                                     │  memcpy sources into pinned, ONE H2D, ONE batched device call,
//                                     │  ONE D2H, record event                      (3 slots in flight)
//                                     ▼
//                              completer thread: wait event, copy payload rows out, call done()
//
// A batch = the run of queued requests at the head of the queue that share the operation and its
// parameters (geometry may differ: the device entry point groups equal geometries itself), bounded by
// kMaxBatch requests and kMaxBytes of staging.  Nothing here computes pixels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/picha_hip.h"

namespace {

constexpr int kSlots = 3;
constexpr size_t kMaxBatch = 64;
constexpr size_t kMaxBytes = 96u << 20;  // staging per batch: small enough that three slots pipeline, large enough to amortise launches

const int kBytes[8] = {3, 4, 1, 2, 2, 4, 6, 8};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
size_t image_span(const picha_image &im) {
    return (size_t)(im.height - 1) * (size_t)im.stride + (size_t)im.width * kBytes[im.pixel];
}

struct Request {
    int kind;   // 0 colorConvert, 1 resize, 2 both chained on the device (picha_hip_convert_resize_device)
    int order;  // chained: picha_chain_order
    picha_image src, dst;
    int filter;
    float fscale;   // resize / chained
    float r, g, b;  // colorConvert / chained
    int numerics;   // resize: -1 = the process-wide option when the batch is launched, 0 / 1 = this request's own (picha_hip_resize_async_ex)
    picha_hip_done_fn done;
    void *user;
    bool compatible(const Request &o) const {
        return kind == o.kind && order == o.order && filter == o.filter && fscale == o.fscale && r == o.r && g == o.g && b == o.b && numerics == o.numerics;
    }
};

// Staging copies (user memory <-> pinned staging) are the slowest stage of the host path for large images: one
// thread moves ~30 GB/s while the PCIe link takes ~50.  Copies of 2 MB or more are split across a few helper threads.
class CopyPool {
public:
    CopyPool() {
        unsigned hw = std::thread::hardware_concurrency();
        int n = hw >= 16 ? 3 : (hw >= 8 ? 2 : (hw >= 4 ? 1 : 0));
        if (const char *e = getenv("PICHA_HIP_COPY_THREADS")) n = std::max(0, std::min(15, atoi(e)));  // helper threads; 0 = none
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void copy(void *dst, const void *src, size_t bytes) {
        const size_t parts = workers_.size() + 1;
        if (bytes < (2u << 20) || parts == 1) { memcpy(dst, src, bytes); return; }
        const size_t chunk = align_up((bytes + parts - 1) / parts, 4096);
        Job job;
        size_t off = chunk;  // the caller copies the first chunk itself
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (; off < bytes; off += chunk) {
                tasks_.push_back({static_cast<uint8_t *>(dst) + off, static_cast<const uint8_t *>(src) + off,
                                  std::min(chunk, bytes - off), &job});
                ++job.pending;
            }
        }
        cv_.notify_all();
        memcpy(dst, src, std::min(chunk, bytes));
        std::unique_lock<std::mutex> lk(mu_);
        job.cv.wait(lk, [&] { return job.pending == 0; });
    }

private:
    struct Job { int pending = 0; std::condition_variable cv; };
    struct Task { uint8_t *dst; const uint8_t *src; size_t bytes; Job *job; };
    void loop() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !tasks_.empty(); });
                if (tasks_.empty()) return;
                t = tasks_.front();
                tasks_.pop_front();
            }
            memcpy(t.dst, t.src, t.bytes);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--t.job->pending == 0) t.job->cv.notify_all();
            }
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Task> tasks_;
    std::vector<std::thread> workers_;
    bool stop_ = false;
};
CopyPool &copy_pool() {
    static CopyPool *pool = new CopyPool;  // shared by every device's stager and completer; lives until the process ends
    return *pool;
}

struct Slot {
    uint8_t *pinned = nullptr, *device = nullptr;
    size_t cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t event = nullptr;
    std::vector<Request> reqs;
    std::vector<size_t> doff;  // offset of each request's destination image in the arenas
    int status = PICHA_OK;
    std::string error;
};

class Dispatcher {
public:
    // `device` is the physical HIP device; several dispatchers may share one (option "async_virtual_devices")
    explicit Dispatcher(int device) : device_(device) {
        stager_ = std::thread([this] { stage_loop(); });
        completer_ = std::thread([this] { complete_loop(); });
    }
    ~Dispatcher() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_queue_.notify_all();
        cv_ready_.notify_all();
        cv_free_.notify_all();
        stager_.join();
        completer_.join();
    }
    void enqueue(const Request &r) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            queue_.push_back(r);
            ++outstanding_;
        }
        cv_queue_.notify_one();
    }
    void drain() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_idle_.wait(lk, [this] { return outstanding_ == 0; });
    }
    std::atomic<uint64_t> batches{0}, completed{0};

private:
    bool ensure(Slot &s, size_t bytes) {
        if (!s.stream && hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return false;
        if (!s.event && hipEventCreateWithFlags(&s.event, hipEventDisableTiming) != hipSuccess) return false;
        if (s.cap >= bytes) return true;
        if (s.pinned) (void)hipHostFree(s.pinned);
        if (s.device) (void)hipFree(s.device);
        s.pinned = s.device = nullptr;
        s.cap = 0;
        const size_t cap = align_up(bytes + bytes / 4, 1 << 20);
        if (hipHostMalloc((void **)&s.pinned, cap, hipHostMallocDefault) != hipSuccess) return false;
        if (hipMalloc((void **)&s.device, cap) != hipSuccess) return false;
        s.cap = cap;
        return true;
    }

    void stage_loop() {
        (void)hipSetDevice(device_);
        int next = 0;
        for (;;) {
            std::vector<Request> batch;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_queue_.wait(lk, [this] { return stop_ || !queue_.empty(); });
                if (stop_ && queue_.empty()) return;
                size_t bytes = 0;
                while (!queue_.empty() && batch.size() < kMaxBatch) {
                    const Request &r = queue_.front();
                    if (!batch.empty() && !batch[0].compatible(r)) break;
                    const size_t b = align_up(image_span(r.src), 256) + align_up(image_span(r.dst), 256);
                    if (!batch.empty() && bytes + b > kMaxBytes) break;
                    bytes += b;
                    batch.push_back(r);
                    queue_.pop_front();
                }
            }
            Slot &s = slots_[next];
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_free_.wait(lk, [&] { return stop_ || !busy_[next]; });
                if (stop_ && busy_[next]) return;
                busy_[next] = true;
            }
            run_batch(s, batch);
            {
                std::lock_guard<std::mutex> lk(mu_);
                ready_.push_back(next);
            }
            cv_ready_.notify_one();
            next = (next + 1) % kSlots;
        }
    }

    void run_batch(Slot &s, std::vector<Request> &batch) {
        const int n = (int)batch.size();
        s.reqs = std::move(batch);
        s.doff.assign(n, 0);
        s.status = PICHA_OK;
        s.error.clear();
        std::vector<size_t> soff(n);
        size_t total = 0;
        for (int i = 0; i < n; ++i) { soff[i] = total; total += align_up(image_span(s.reqs[i].src), 256); }
        const size_t dst_begin = total;
        for (int i = 0; i < n; ++i) { s.doff[i] = total; total += align_up(image_span(s.reqs[i].dst), 256); }
        if (!ensure(s, total + 256)) { s.status = PICHA_ERR_HIP; s.error = "dispatcher: staging allocation failed"; return; }
        std::vector<picha_image> dsrc(n), ddst(n);
        for (int i = 0; i < n; ++i) {
            dsrc[i] = s.reqs[i].src; dsrc[i].data = s.device + soff[i];
            ddst[i] = s.reqs[i].dst; ddst[i].data = s.device + s.doff[i];
            copy_pool().copy(s.pinned + soff[i], s.reqs[i].src.data, image_span(s.reqs[i].src));
        }
        hipError_t e = hipMemcpyAsync(s.device, s.pinned, dst_begin, hipMemcpyHostToDevice, s.stream);
        int rc = PICHA_OK;
        if (e == hipSuccess) {
            const Request &r0 = s.reqs[0];
            const picha_call_opts co{(int32_t)sizeof(picha_call_opts), r0.numerics};
            rc = r0.kind == 1   ? picha_hip_resize_device_ex(n, dsrc.data(), ddst.data(), r0.filter, r0.fscale, r0.numerics >= 0 ? &co : nullptr, s.stream)
                 : r0.kind == 0 ? picha_hip_color_convert_device(n, dsrc.data(), ddst.data(), r0.r, r0.g, r0.b, s.stream)
                                : picha_hip_convert_resize_device(n, dsrc.data(), ddst.data(), r0.order, r0.r, r0.g, r0.b,
                                                                  r0.filter, r0.fscale, s.stream);
            if (rc != PICHA_OK) { s.status = rc; s.error = picha_hip_last_error(); }
        }
        if (e == hipSuccess && rc == PICHA_OK)
            e = hipMemcpyAsync(s.pinned + dst_begin, s.device + dst_begin, total - dst_begin, hipMemcpyDeviceToHost, s.stream);
        if (e == hipSuccess) e = hipEventRecord(s.event, s.stream);
        if (e != hipSuccess && s.status == PICHA_OK) { s.status = PICHA_ERR_HIP; s.error = hipGetErrorString(e); }
        ++batches;
    }

    void complete_loop() {
        (void)hipSetDevice(device_);
        for (;;) {
            int k;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_ready_.wait(lk, [this] { return stop_ || !ready_.empty(); });
                if (ready_.empty()) return;
                k = ready_.front();
                ready_.pop_front();
            }
            Slot &s = slots_[k];
            if (s.status == PICHA_OK) {
                hipError_t e = hipEventSynchronize(s.event);
                if (e != hipSuccess) { s.status = PICHA_ERR_HIP; s.error = hipGetErrorString(e); }
            } else {
                (void)hipStreamSynchronize(s.stream);
            }
            const size_t n = s.reqs.size();
            for (size_t i = 0; i < n; ++i) {
                const Request &r = s.reqs[i];
                if (s.status == PICHA_OK) {
                    // payload bytes only: destination row padding stays untouched (src/resize.cc:122-132)
                    const size_t rw = (size_t)r.dst.width * kBytes[r.dst.pixel];
                    const uint8_t *from = s.pinned + s.doff[i];
                    uint8_t *to = static_cast<uint8_t *>(r.dst.data);
                    if ((size_t)r.dst.stride == rw) copy_pool().copy(to, from, rw * r.dst.height);
                    else for (int y = 0; y < r.dst.height; ++y) memcpy(to + (size_t)y * r.dst.stride, from + (size_t)y * r.dst.stride, rw);
                }
                if (r.done) r.done(r.user, s.status, s.status == PICHA_OK ? "" : s.error.c_str());
            }
            completed += n;
            s.reqs.clear();
            {
                std::lock_guard<std::mutex> lk(mu_);
                busy_[k] = false;
                outstanding_ -= n;
                if (outstanding_ == 0) cv_idle_.notify_all();
            }
            cv_free_.notify_all();
        }
    }

    const int device_;
    std::mutex mu_;
    std::condition_variable cv_queue_, cv_ready_, cv_free_, cv_idle_;
    std::deque<Request> queue_;
    std::deque<int> ready_;
    Slot slots_[kSlots];
    bool busy_[kSlots] = {false, false, false};
    size_t outstanding_ = 0;
    bool stop_ = false;
    std::thread stager_, completer_;
};

std::mutex g_mu;
std::map<int, std::unique_ptr<Dispatcher>> g_disp;

Dispatcher *dispatcher_for(int slot, int physical_device) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto &p = g_disp[slot];
    if (!p) p.reset(new Dispatcher(physical_device));
    return p.get();
}

// validation shared with the synchronous entry points lives in capi.cpp
extern "C" int picha_hip_validate_host_op(int kind, const picha_image *src, const picha_image *dst, int filter,
                                          float filter_scale, int *device);

int submit(int kind, int order, const picha_image *src, const picha_image *dst, int filter, float fscale, float r, float g,
           float b, picha_hip_done_fn done, void *user, int numerics = -1) {
    int dev = 0;
    int rc = picha_hip_validate_host_op(kind, src, dst, filter, fscale, &dev);
    if (rc != PICHA_OK) return rc;
    // "async_devices" = N > 0: requests are dealt round-robin to devices 0 .. min(N, visible) - 1, each with its own
    // dispatcher (stager, completer, streams, staging): one host process feeds every GPU of the node with independent
    // images, no exchange between devices.  Default (0): the caller's current device only.
    // "async_virtual_devices" = 1 lets N exceed the visible device count: dispatcher k then runs on physical device
    // k mod visible — on a one-GPU box this exercises the spread itself (several dispatchers, each with its own streams
    // and staging, fed round-robin) although they share the card.  At most 8 dispatchers per visible device.
    const long want = picha_hip_get_option("async_devices");
    int slot = dev;  // dispatcher index; physical device = slot mod visible
    int phys = dev;
    if (want > 0) {
        static std::atomic<unsigned> next{0};
        int ndev = 1;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) ndev = 1;
        // virtual dispatchers are bounded: each one owns two threads, three streams and pinned staging
        const long n = picha_hip_get_option("async_virtual_devices") > 0 ? std::min<long>(want, 8L * ndev) : std::min<long>(want, ndev);
        slot = (int)(next.fetch_add(1) % (unsigned)n);
        phys = slot % ndev;
    }
    Request q;
    q.kind = kind;
    q.order = order;
    q.src = *src;
    q.dst = *dst;
    q.filter = filter;
    q.fscale = fscale;
    q.r = r; q.g = g; q.b = b;
    q.numerics = numerics;
    q.done = done;
    q.user = user;
    dispatcher_for(slot, phys)->enqueue(q);
    return PICHA_OK;
}

}  // namespace

// staging copy shared with the synchronous host entry points (capi.cpp); not part of the public header
extern "C" void picha_hip_staging_copy(void *dst, const void *src, size_t bytes) { copy_pool().copy(dst, src, bytes); }

extern "C" {

int picha_hip_resize_async(const picha_image *src, const picha_image *dst, int filter, float filter_scale,
                           picha_hip_done_fn done, void *user) {
    return submit(1, 0, src, dst, filter, filter_scale, 0.0f, 0.0f, 0.0f, done, user);
}

int picha_hip_resize_async_ex(const picha_image *src, const picha_image *dst, int filter, float filter_scale, const picha_call_opts *opts,
                              picha_hip_done_fn done, void *user) {
    int numerics = -1;
    if (opts) {
        if (opts->size < (int32_t)sizeof(picha_call_opts) || opts->numerics < -1 || opts->numerics > 1) return PICHA_ERR_ARGUMENT;
        numerics = opts->numerics;
    }
    return submit(1, 0, src, dst, filter, filter_scale, 0.0f, 0.0f, 0.0f, done, user, numerics);
}

int picha_hip_color_convert_async(const picha_image *src, const picha_image *dst, float r, float g, float b,
                                  picha_hip_done_fn done, void *user) {
    return submit(0, 0, src, dst, 0, 0.0f, r, g, b, done, user);
}

int picha_hip_convert_resize_async(const picha_image *src, const picha_image *dst, int order, float r, float g, float b,
                                   int filter, float filter_scale, picha_hip_done_fn done, void *user) {
    if (order != PICHA_CONVERT_THEN_RESIZE && order != PICHA_RESIZE_THEN_CONVERT) return PICHA_ERR_ARGUMENT;
    return submit(2, order, src, dst, filter, filter_scale, r, g, b, done, user);
}

int picha_hip_async_drain(void) {
    std::vector<Dispatcher *> all;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto &kv : g_disp) all.push_back(kv.second.get());
    }
    for (Dispatcher *d : all) d->drain();
    return PICHA_OK;
}

int picha_hip_async_dispatcher_stats(int index, uint64_t *batches, uint64_t *requests) {
    std::lock_guard<std::mutex> lk(g_mu);
    int i = 0;
    for (auto &kv : g_disp) {
        if (i++ == index) {
            if (batches) *batches = kv.second->batches;
            if (requests) *requests = kv.second->completed;
            break;
        }
    }
    return (int)g_disp.size();
}

void picha_hip_async_stats(uint64_t *batches, uint64_t *requests) {
    uint64_t b = 0, r = 0;
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_disp) { b += kv.second->batches; r += kv.second->completed; }
    if (batches) *batches = b;
    if (requests) *requests = r;
}

}  // extern "C"
