// capi.cpp — the C ABI of libpicha_hip.so (include/picha_hip.h): validation, per-device tap-table
// cache, launch planning, host staging.  Compiled by hipcc as host code.  No CPU compute path
// lives here: every pixel is produced by the kernels in kernels.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <climits>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/picha_hip.h"
#include "kernels.h"
#include "lane_pool.h"
#include "taps.h"

using namespace picha_hip;

namespace {

// ---- error plumbing ----------------------------------------------------------------------------
thread_local std::string g_last_error;
thread_local const char *g_last_kernel = "";
thread_local char g_chain_name[288];  // "first + second" of a chained convert + resize

int fail(int status, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(PICHA_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));          \
    } while (0)

// ---- format tables (src/picha.h:118-200, src/picha.cc:36-39, src/resize.cc:162-164) -------------
const int kBytes[8] = {3, 4, 1, 2, 2, 4, 6, 8};
const int kChannels[8] = {3, 4, 1, 2, 1, 2, 3, 4};
const char *const kPixelNames[8] = {"rgb", "rgba", "grey", "greya", "r16", "r16g16", "r16g16b16", "r16g16b16a16"};
const char *const kFilterNames[6] = {"cubic", "lanczos", "catmulrom", "mitchel", "box", "triangle"};

bool valid_pixel(int p) { return p >= 0 && p < PICHA_NUM_PIXELS; }

// ---- options -----------------------------------------------------------------------------------
std::mutex g_opt_mu;
long g_opt_path = 0;        // 0 auto, 1 two-pass, 2 fused (error if unsupported), 3 fused but never the streaming kernel
long g_opt_row_chunks = 0;  // 0 auto
// 0 reference arithmetic (bit-exact, the default), 1 fused-multiply-add flavour (<= 1 code value).
// PICHA_HIP_NUMERICS=fma|exact in the environment sets the initial value.
// per-call override of the numerics flavour (picha_hip_*_ex): -1 = none.  Thread-local: the planner runs on the calling thread.
thread_local int tl_numerics = -1;
long g_opt_numerics = [] {
    const char *e = getenv("PICHA_HIP_NUMERICS");
    return (e && strcmp(e, "fma") == 0) ? 1L : 0L;
}();
// intermediate bytes per slice of a chained convert+resize.  Slices small enough to stay in the 256 MB Infinity Cache
// were measured SLOWER (r01, profiles/r01_chain_window.txt: 96 MB 2.91 ms vs unsliced 2.49 ms for cfg3 — the short
// launches do not fill the chip), so the window only bounds memory.
long g_opt_chain_window_mb = 4096;
long g_opt_xcd_remap = 0;   // streaming kernel, XCD-contiguous workgroup order: 0 auto (on for >= 2 outputs per lane), 1 on, 2 off
// async dispatcher: 0 = the caller's current device, N = round-robin over the first N devices (PICHA_HIP_ASYNC_DEVICES)
long g_opt_async_devices = [] {
    const char *e = getenv("PICHA_HIP_ASYNC_DEVICES");
    return e ? std::max(0L, atol(e)) : 0L;
}();
long g_opt_async_virtual = 0;  // 1: "async_devices" may exceed the visible device count (dispatcher k runs on device k mod visible)
long g_opt_acc = 0;         // accumulator kernel: 0 auto (when the span kernel's ring would be large), 1 whenever representable, 2 never
long g_opt_per_lane = 0;    // 0 auto, else outputs per lane for the fused kernel (1, 2, 4)
long g_opt_stream_rot = 0;  // 0 auto, 1 force the rotating-accumulator streaming variant, 2 force the unrolled one
long g_opt_dpp = 0;         // 0 auto, 1 prefer / 2 forbid the cross-lane pixel-sharing streaming kernel
long g_opt_cc_mode = 0;     // colour convert: 0 auto (wide kernel when rows are dword aligned), 1 always the 4-pixel kernel, 2 wide + nt stores
long g_opt_sym = 0;         // symmetric-table streaming kernel (kernels_sym.hip): 0 auto (when available), 2 never
long g_opt_block_waves = 0; // its waves per workgroup (0 auto)
long g_opt_acc_p2 = 0;      // accumulator kernel with two outputs per lane: 0 auto (measured table), 1 whenever possible, 2 never
long g_opt_long_block = 0;  // long-window accumulator kernel: threads per workgroup (0 auto, 64 / 128 / 192 / 256)
long g_opt_sym_sync = 0;    // 1: a barrier per step keeps the waves of a workgroup in lockstep
long g_opt_work = 0;        // sym / accumulator kernels, work items: 0 auto, 1 explicit item list (one workgroup per item, short chunks last),
                            // 2 persistent launch (resident workgroups pull items from per-XCD atomic queues), 3 plain 3-D grid
long g_opt_work_tail = 200; // per mille of every XCD's sub-list that is cut into short row chunks
long g_opt_work_tail_div = 4;  // ... short chunk = long chunk / this
long g_opt_up = 0;          // integer upscales on the accumulator kernel (several outputs per lane, register accumulators): 0 auto (16-bit formats), 1 always, 2 never
long g_opt_xch = 0;         // symmetric-table kernel, neighbour exchange (a lane loads only its own pixels): 0 auto, 1 wherever instantiated, 2 never

// ---- per-device state ---------------------------------------------------------------------------
struct DeviceAxis {
    AxisTaps host;
    void *blob = nullptr;  // one allocation holding every table below
    // generic (two-pass) tables
    int32_t *left = nullptr, *count = nullptr, *eff = nullptr;
    float *weight = nullptr;  // [dst][taps]
    // fused, x role: built on demand per (outputs per lane, step, kernel taps, numerics, sample type)
    struct XTable {
        bool ok = false;        // false: the tap table does not have this uniform structure
        void *blob = nullptr;
        int32_t *left = nullptr;  // [groups]
        float *weight = nullptr;  // [taps][stride]
        int groups = 0, stride = 0;
        int l0 = 0;  // uniform_base tables: the common window offset (window of output x starts at step*x + l0)
        // canonical outputs (resize_sym_kernel): outputs [c0, c1) all carry the weights of the middle output, those weights
        // are symmetric (w[j] == w[taps-1-j] bitwise) and their windows start at step*x + xbase
        bool sym = false;
        int c0 = 0, c1 = 0, xbase = 0;
        int32_t canon[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // float bits of w[0 .. (taps+1)/2) in this table's weight flavour
        std::vector<int32_t> host_left;               // host copy of left[] (row-staged kernel: strip extents)
    };
    std::map<std::tuple<int, int, int, int, int>, XTable> xtables;
    // streaming kernel, y role: uniform-stride view of the table, built on demand per (step, taps)
    struct YStream {
        bool ok = false;
        void *blob = nullptr;
        float *step_weight = nullptr;  // [dst + (taps-1)/step][yw]
        int base = 0;                  // output y reads source rows step*y + base + j
        bool sym = false;              // the interior rows have symmetric canonical weights (flag bit 1 of their step records)
        int32_t canon[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    };
    std::map<std::pair<int, int>, YStream> ystreams;
    // accumulator kernel, y role: per-source-row records for `nacc` round-robin accumulators, built on demand
    struct YRows {
        bool ok = false;
        void *blob = nullptr;
        int32_t *rec = nullptr;    // [src + 8][2 * nacc]: weights (float bits), then completing output row or -1
        int32_t *range = nullptr;  // [dst][2]: first source row needed by this or any later output, last row so far
        // ring aliasing of the one-row kind (the window has one row more than the reference's ring: the FIRST tap reads the
        // window's LAST row, src/resize.cc:52-64,125-129): rec / range then refer to positions in a list of source rows in
        // which that last row appears once more ahead of its window; src[position] = the source row (nullptr: no aliasing)
        int32_t *src = nullptr;    // [list length + 8]
    };
    std::map<int, YRows> yrows;
    // row-staged kernel, x role: per column strip of 256 threads the first source pixel staged into LDS
    struct RowsStrips {
        bool ok = false;
        void *blob = nullptr;
        int32_t *strip_left = nullptr;
        int n_strips = 0, pieces = 0;  // stage pieces per thread and row (1, 2, 4, 8)
    };
    std::map<std::tuple<const void *, int, int>, RowsStrips> rows_strips;
    // run kernel, x role: the period's canonical weights and the columns that do not have them
    struct RunTable {
        bool ok = false;
        void *blob = nullptr;
        float *canon = nullptr;        // [period][taps]
        int c0 = 0, c1 = 0;            // columns [c0, c1) have the canonical weights
        int x_base = 0, taps = 0;
    };
    std::map<std::tuple<int, int, int, int>, RunTable> runtables;
    int32_t *yrange = nullptr;  // y role of resize_up_kernel: [dst][2] {first source row needed by this or any later output row, last row so far}

    DeviceAxis() = default;
    DeviceAxis(const DeviceAxis &) = delete;
    DeviceAxis &operator=(const DeviceAxis &) = delete;
    ~DeviceAxis() {  // hipFree waits for the device, so kernels still reading these tables finish first
        if (blob) (void)hipFree(blob);
        for (auto &kv : xtables) if (kv.second.blob) (void)hipFree(kv.second.blob);
        for (auto &kv : ystreams) if (kv.second.blob) (void)hipFree(kv.second.blob);
        for (auto &kv : yrows) if (kv.second.blob) (void)hipFree(kv.second.blob);
        for (auto &kv : rows_strips) if (kv.second.blob) (void)hipFree(kv.second.blob);
        for (auto &kv : runtables) if (kv.second.blob) (void)hipFree(kv.second.blob);
        if (yrange) (void)hipFree(yrange);
    }
    // fused, y role
    int4 *y_info = nullptr;
    float *y_weight = nullptr;  // [dst][y_pad]
    int32_t *y_slot = nullptr;  // [dst][y_pad]
    int y_pad = 0;
};

using AxisKey = std::tuple<int, uint32_t, int, int>;

// A reusable device (+ optional pinned host) buffer whose successive uses are ordered by an event:
// the next user, on whatever stream, waits for the previous use to finish.  Stream discipline of the
// library (what tests/test_gpu_parity.py::test_temporaries_are_ordered_across_streams pins): a
// temporary is only touched by work enqueued between temp_acquire and temp_release on the caller's
// stream; temp_release records the event on that stream; the next temp_acquire, from any stream,
// makes its stream wait for the event (or the host, when the pinned half is rewritten).
// hipMallocAsync is not used: grow-only buffers need no per-call allocation.  (Round 1 blamed a
// first-session failure — 8 of 192 conversions wrong — on stale pool memory; the record of it,
// every pair with an 8-byte destination pixel wrong from the first pixel on, is what a kernel bug
// of that uncommitted code looks like, not what stale memory looks like: the claim is withdrawn.)
struct TempBuf {
    void *dev = nullptr;
    size_t dev_cap = 0;
    void *host = nullptr;  // pinned
    size_t host_cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
};

constexpr size_t kMaxCachedAxes = 256;  // distinct (filter, scale, src, dst) axes kept per device

struct DeviceCtx {
    std::mutex mu;
    std::map<AxisKey, std::shared_ptr<DeviceAxis>> axes;
    std::deque<AxisKey> axis_order;  // insertion order, for eviction
    std::mutex temp_mu;  // held from temp_acquire to temp_release
    TempBuf scratch;     // two-pass float scratch
    TempBuf tables;      // per-image pointer tables
    std::mutex chain_mu; // held while a chained convert+resize uses `chain`
    TempBuf chain;       // intermediate images of picha_hip_convert_resize_device
    std::mutex results_mu;
    TempBuf results;     // picha_hip_image_compare_device: per-image sums (device + pinned mirror)
    // work-item lists (kernels.h, WorkList) per (images, output rows, strip blocks, long chunk, short chunk, tail share), and
    // one set of queue counters per stream (launches on a stream are ordered, so they can share it); both under `mu`
    // (r04) both caches hand out shared ownership: an entry evicted while another host thread is between lookup and
    // launch stays alive until that thread lets go of it, and its hipFree — which waits for the kernels already enqueued — runs
    // outside `mu`, from whichever thread drops the last reference.  Eviction takes the least recently used entry, one at a time.
    struct ItemList {
        void *blob = nullptr; const int4 *items = nullptr; int per_xcd = 0;
        unsigned long long used = 0;
        ItemList() = default;
        ItemList(const ItemList &) = delete;
        ItemList &operator=(const ItemList &) = delete;
        ~ItemList() { if (blob) (void)hipFree(blob); }
    };
    struct Queue {
        unsigned *counters = nullptr;
        unsigned long long used = 0;
        Queue() = default;
        Queue(const Queue &) = delete;
        Queue &operator=(const Queue &) = delete;
        ~Queue() { if (counters) (void)hipFree(counters); }
    };
    std::map<std::tuple<int, int, int, int, int, int>, std::shared_ptr<ItemList>> item_lists;
    std::map<hipStream_t, std::shared_ptr<Queue>> queues;
    unsigned long long use_tick = 0;
};
constexpr size_t kMaxItemLists = 64, kMaxQueues = 256;

std::mutex g_ctx_mu;
std::map<int, std::unique_ptr<DeviceCtx>> g_ctx;

int current_device(int *dev) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(PICHA_ERR_NO_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    HIP_TRY(hipGetDevice(dev));
    return PICHA_OK;
}

DeviceCtx *ctx_for(int dev) {
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto &p = g_ctx[dev];
    if (!p) p.reset(new DeviceCtx);
    return p.get();
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Caller holds ctx->temp_mu.  After this returns, work enqueued on `s` may use t.dev[0, dev_bytes)
// and the host may write t.host[0, host_bytes).
int temp_acquire(TempBuf &t, size_t dev_bytes, size_t host_bytes, hipStream_t s) {
    if (!t.ev) HIP_TRY(hipEventCreateWithFlags(&t.ev, hipEventDisableTiming));
    const bool grow = dev_bytes > t.dev_cap || host_bytes > t.host_cap;
    if (t.pending) {
        if (grow || host_bytes > 0) HIP_TRY(hipEventSynchronize(t.ev));  // host side is about to be reused
        else HIP_TRY(hipStreamWaitEvent(s, t.ev, 0));
        if (grow || host_bytes > 0) t.pending = false;
    }
    if (dev_bytes > t.dev_cap) {
        if (t.dev) HIP_TRY(hipFree(t.dev));
        t.dev = nullptr; t.dev_cap = 0;
        const size_t cap = align_up(dev_bytes + dev_bytes / 4, 1 << 20);
        HIP_TRY(hipMalloc(&t.dev, cap));
        t.dev_cap = cap;
    }
    if (host_bytes > t.host_cap) {
        if (t.host) HIP_TRY(hipHostFree(t.host));
        t.host = nullptr; t.host_cap = 0;
        const size_t cap = align_up(host_bytes * 2, 1 << 16);
        HIP_TRY(hipHostMalloc(&t.host, cap, hipHostMallocDefault));
        t.host_cap = cap;
    }
    return PICHA_OK;
}
int temp_release(TempBuf &t, hipStream_t s) {
    HIP_TRY(hipEventRecord(t.ev, s));
    t.pending = true;
    return PICHA_OK;
}

// Builds (once) and uploads the tables for one axis.  Synchronous on first use of a geometry;
// afterwards a pure lookup.
int get_axis(DeviceCtx *ctx, int filter, float fscale, int src, int dst, std::shared_ptr<DeviceAxis> *out) {
    uint32_t bits;
    memcpy(&bits, &fscale, 4);
    AxisKey key(filter, bits, src, dst);
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->axes.find(key);
    if (it != ctx->axes.end()) {
        *out = it->second;
        return PICHA_OK;
    }
    std::shared_ptr<DeviceAxis> ax(new DeviceAxis);
    int rc = build_axis_taps(filter, fscale, src, dst, ax->host);
    if (rc != PICHA_OK) return fail(rc, "%s", picha_hip_strerror(rc));
    const AxisTaps &t = ax->host;
    const int T = t.taps;
    const int YP = fused_y_pad(T);
    ax->y_pad = YP;

    // host-side images of the device tables
    std::vector<int4> y_info(dst);
    std::vector<float> y_weight((size_t)dst * YP, 0.0f);
    std::vector<int32_t> y_slot((size_t)dst * YP, 0);
    const int ring = std::max(t.span, 1);
    for (int i = 0; i < dst; ++i) {
        y_info[i] = make_int4(t.left[i], t.need[i], t.count[i], 0);
        for (int j = 0; j < YP; ++j) {
            const int jj = j < T ? j : 0;
            y_weight[(size_t)i * YP + j] = j < T ? t.weight[(size_t)i * T + j] : 0.0f;
            y_slot[(size_t)i * YP + j] = t.eff[(size_t)i * T + jj] % ring;
        }
    }

    // y_info.x = the first source row needed by this output row OR ANY LATER ONE: a row chunk starts resampling there.
    // left[] itself is not monotonic — trimming exact-zero end taps (src/resize.cc:31-34) can move one row's left past
    // its successor's (catmulrom / mitchel / cubic upscales, whose kernels are exactly 0 at integer offsets); a chunk
    // that started at its first row's own left would then never resample a row the next output needs.  need[] is
    // taken before trimming and is monotonic.
    for (int i = dst - 2; i >= 0; --i) y_info[i].x = std::min(y_info[i].x, y_info[i + 1].x);

    struct Part { const void *src; size_t bytes; void **dst; };
    Part parts[] = {
        {t.left.data(), sizeof(int32_t) * dst, (void **)&ax->left},
        {t.count.data(), sizeof(int32_t) * dst, (void **)&ax->count},
        {t.eff.data(), sizeof(int32_t) * (size_t)dst * T, (void **)&ax->eff},
        {t.weight.data(), sizeof(float) * (size_t)dst * T, (void **)&ax->weight},
        {y_info.data(), sizeof(int4) * dst, (void **)&ax->y_info},
        {y_weight.data(), sizeof(float) * (size_t)dst * YP, (void **)&ax->y_weight},
        {y_slot.data(), sizeof(int32_t) * (size_t)dst * YP, (void **)&ax->y_slot},
    };
    size_t total = 0;
    for (auto &p : parts) total += align_up(p.bytes, 256);
    std::vector<uint8_t> image(total, 0);
    HIP_TRY(hipMalloc(&ax->blob, total));
    size_t off = 0;
    for (auto &p : parts) {
        memcpy(image.data() + off, p.src, p.bytes);
        *p.dst = static_cast<uint8_t *>(ax->blob) + off;
        off += align_up(p.bytes, 256);
    }
    HIP_TRY(hipMemcpy(ax->blob, image.data(), total, hipMemcpyHostToDevice));
    *out = ax;
    ctx->axes[key] = ax;
    ctx->axis_order.push_back(key);
    // bounded cache: a service that sees arbitrary geometries must not grow without limit.  Callers hold
    // shared_ptrs while they enqueue work, so an evicted axis is freed only after its last user is done.
    while (ctx->axis_order.size() > kMaxCachedAxes) {
        ctx->axes.erase(ctx->axis_order.front());
        ctx->axis_order.pop_front();
    }
    return PICHA_OK;
}

// Horizontal tables for the fused kernel: `per_lane` consecutive outputs share one lane; output q of
// a group reads span pixels [step*q, step*q + taps) of the span starting at left[group].  For
// per_lane > 1 this needs the reference's tap table to have that uniform stride (true for integer
// scale factors); the check below decides, the result is cached.  Taps are zero padded to `taps`;
// pixels a padded tap addresses are either inside the image (finite, times 0) or outside it (the
// kernel's buffer loads return 0).  fma = 1: weights pre-multiplied by 1/max of the sample type; fma = 2: the
// reference's weights times 2^22 for kernels that unpack samples as scaled denormals (kernels_impl.inc, Sample::kDen).
int get_xtable(DeviceCtx *ctx, DeviceAxis *ax, int per_lane, int step, int taps, int fma, int is16,
               const DeviceAxis::XTable **out, int bytespan_bpp = 0, bool uniform_base = false, int window_align = 0, bool overrun = false) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto key = std::make_tuple(per_lane, step, taps, fma, is16 + 16 * bytespan_bpp + (uniform_base ? 4096 : 0) + (overrun ? 8192 : 0) + 65536 * window_align);
    auto it = ax->xtables.find(key);
    if (it != ax->xtables.end()) { *out = &it->second; return PICHA_OK; }
    DeviceAxis::XTable xt;
    const AxisTaps &t = ax->host;
    const int dst = t.dst_size, T = t.taps;
    const int groups = (dst + per_lane - 1) / per_lane;
    const int stride = groups * per_lane;
    std::vector<int32_t> left(groups), gbase(groups);
    std::vector<float> weight((size_t)taps * stride, 0.0f);
    const float inv = is16 ? 1.0f / 65535.0f : 1.0f / 255.0f;
    bool ok = taps >= T || per_lane > 1;
    // byte-span kernels read whole dwords from the window start: the window (taps*bpp bytes, a
    // multiple of 4) must end inside the row, so windows near the right edge are shifted left and
    // their weights move to the back (zero weights in front leave the sum unchanged)
    // ... and they start on a 4-byte boundary of the row (byte-unaligned dwordx4 loads run at a fraction of
    // the aligned rate): the window start is rounded down to `align_px` pixels, the weights move back by as much
    // window_align (row-staged kernel): window starts are multiples of that many pixels (8-byte LDS reads); such windows may
    // start before pixel 0 and end past the row (the staged span covers them; those taps have weight 0)
    const int align_px = window_align > 0 ? window_align : bytespan_bpp ? ((bytespan_bpp % 2) ? 4 : 2) : 1;
    int max_left = bytespan_bpp ? t.src_size - taps : INT32_MAX;
    // the last window may start on the 4-byte boundary at or ABOVE src - taps: it then runs up to align_px - 1 pixels past the end of
    // the row — the next row's first pixels, or for the last row bytes inside the kernel's dword-rounded descriptor — all under
    // zero weights (r03; rounding down instead lost the row's last pixels, and such images went to the two-pass kernels)
    // `overrun` needs every dword load aligned (image bases and strides multiples of 4: the caller checks), so that no load
    // straddles the descriptor's end; otherwise the last window stays inside the row (rounded down) as before
    if (bytespan_bpp && overrun) max_left += (align_px - ((max_left % align_px) + align_px) % align_px) % align_px;
    else if (bytespan_bpp) max_left -= ((max_left % align_px) + align_px) % align_px;
    if (bytespan_bpp && max_left < 0) ok = false;
    // uniform_base: every window starts at step*x + L0 with one L0 for the whole axis (the cross-lane
    // sharing kernel addresses neighbours' pixels by lane distance); left[] then holds L0 in every entry
    int l0 = INT32_MAX;
    if (uniform_base)
        for (int i = 0; i < dst; ++i) l0 = std::min(l0, t.left[i] - step * i);
    for (int g = 0; g < groups && ok; ++g) {
        int base = INT32_MAX;
        for (int q = 0; q < per_lane && g * per_lane + q < dst; ++q) base = std::min(base, t.left[g * per_lane + q] - step * q);
        if (uniform_base) base = step * g * per_lane + l0;
        base = std::min(base, max_left);
        if (bytespan_bpp || window_align > 1) base -= ((base % align_px) + align_px) % align_px;
        left[g] = uniform_base ? l0 : base;
        gbase[g] = base;
        for (int q = 0; q < per_lane && g * per_lane + q < dst; ++q) {
            const int i = g * per_lane + q;
            const int first = t.left[i] - (base + step * q);  // >= 0 by construction
            if (first + t.count[i] > taps) { ok = false; break; }
            for (int j = 0; j < t.count[i]; ++j) {
                const float w = t.weight[(size_t)i * T + j];
                weight[(size_t)(first + j) * stride + i] = fma == 1 ? w * inv : fma == 2 ? w * 4194304.0f : w;
            }
        }
    }
    xt.ok = ok;
    xt.l0 = uniform_base ? l0 : 0;
    if (ok && !bytespan_bpp && taps <= 15 && dst >= 4) {
        int xb = INT32_MAX;
        for (int i = 0; i < dst; ++i) xb = std::min(xb, t.left[i] - step * i);
        const int mid = dst / 2;
        auto wbits = [&](int i, int j) { int32_t b; memcpy(&b, &weight[(size_t)j * stride + i], 4); return b; };
        auto is_canon = [&](int i) {
            if (gbase[i / per_lane] + step * (i % per_lane) != step * i + xb) return false;
            for (int j = 0; j < taps; ++j) if (wbits(i, j) != wbits(mid, j)) return false;
            return true;
        };
        bool symm = is_canon(mid);
        for (int j = 0; j < taps && symm; ++j) if (wbits(mid, j) != wbits(mid, taps - 1 - j)) symm = false;
        if (symm) {
            int c0 = mid, c1 = mid + 1;
            while (c0 > 0 && is_canon(c0 - 1)) --c0;
            while (c1 < dst && is_canon(c1)) ++c1;
            xt.sym = true; xt.c0 = c0; xt.c1 = c1; xt.xbase = xb;
            for (int j = 0; j < (taps + 1) / 2; ++j) xt.canon[j] = wbits(mid, j);
        }
    }
    if (ok) {
        const size_t lb = align_up(sizeof(int32_t) * groups, 256), wb = sizeof(float) * weight.size();
        std::vector<uint8_t> image(lb + wb, 0);
        memcpy(image.data(), left.data(), sizeof(int32_t) * groups);
        memcpy(image.data() + lb, weight.data(), wb);
        HIP_TRY(hipMalloc(&xt.blob, lb + wb));
        HIP_TRY(hipMemcpy(xt.blob, image.data(), lb + wb, hipMemcpyHostToDevice));
        xt.left = static_cast<int32_t *>(xt.blob);
        xt.weight = reinterpret_cast<float *>(static_cast<uint8_t *>(xt.blob) + lb);
        xt.groups = groups;
        xt.stride = stride;
        xt.host_left = left;
    }
    auto ins = ax->xtables.emplace(key, xt);
    *out = &ins.first->second;
    return PICHA_OK;
}

// Vertical table for the streaming kernel.  Valid when every output row's taps fit the uniform window
// rows [step*y + base, step*y + base + taps) and the reference's ring never aliases (SURVEY quirk 1:
// then the rows read are exactly left + j).  Layout: one record per step m (= SY source rows);
// entry i*step + ph is the weight output m - i applies to source row step*m + base + ph, i.e. its
// tap step*i + ph (0 where the output does not exist or the tap is padding).
int get_ystream(DeviceCtx *ctx, DeviceAxis *ay, int step, int taps, const DeviceAxis::YStream **out) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto key = std::make_pair(step, taps);
    auto it = ay->ystreams.find(key);
    if (it != ay->ystreams.end()) { *out = &it->second; return PICHA_OK; }
    DeviceAxis::YStream ys;
    const AxisTaps &t = ay->host;
    const int dst = t.dst_size, T = t.taps;
    int base = INT32_MAX;
    for (int y = 0; y < dst; ++y) base = std::min(base, t.left[y] - step * y);
    bool ok = true;
    for (int y = 0; y < dst && ok; ++y) {
        if (t.left[y] - (step * y + base) + t.count[y] > taps) ok = false;
        for (int j = 0; j < t.count[y] && ok; ++j)
            if (t.eff[(size_t)y * T + j] != t.left[y] + j) ok = false;  // ring aliasing: not representable
    }
    ys.ok = ok;
    ys.base = base;
    if (ok) {
        const int npend = (taps - 1) / step + 1, iclose = (taps - 1) / step;
        const int yw = (npend * step + 1 + 3) / 4 * 4;  // weights + "same as previous record" flag in the last dword
        const int nsteps = dst + iclose;
        std::vector<float> tab((size_t)nsteps * yw, 0.0f);
        for (int y = 0; y < dst; ++y) {
            const int first = t.left[y] - (step * y + base);
            for (int j = 0; j < t.count[y]; ++j) {
                const int tap = first + j;            // position in the uniform window
                const int i = tap / step, ph = tap % step;
                tab[(size_t)(y + i) * yw + i * step + ph] = t.weight[(size_t)y * T + j];
            }
        }
        // canonical rows: the weights of the middle output row in the uniform window, if symmetric; rows [c0, c1) carry them
        int c0 = 0, c1 = 0;
        if (taps <= 15 && dst >= 4) {
            auto wpad = [&](int y, float *w) {
                for (int j = 0; j < taps; ++j) w[j] = 0.0f;
                const int first = t.left[y] - (step * y + base);
                for (int j = 0; j < t.count[y]; ++j) w[first + j] = t.weight[(size_t)y * T + j];
            };
            float wc[16], wt[16];
            const int mid = dst / 2;
            wpad(mid, wc);
            bool symm = true;
            for (int j = 0; j < taps; ++j) if (memcmp(&wc[j], &wc[taps - 1 - j], 4) != 0) symm = false;
            if (symm) {
                auto is_canon = [&](int y) { wpad(y, wt); return memcmp(wt, wc, sizeof(float) * taps) == 0; };
                c0 = mid; c1 = mid + 1;
                while (c0 > 0 && is_canon(c0 - 1)) --c0;
                while (c1 < dst && is_canon(c1)) ++c1;
                ys.sym = true;
                for (int j = 0; j < (taps + 1) / 2; ++j) memcpy(&ys.canon[j], &wc[j], 4);
            }
        }
        for (int m = 0; m < nsteps; ++m) {
            int32_t flags = 0;
            if (m > 0 && memcmp(&tab[(size_t)m * yw], &tab[(size_t)(m - 1) * yw], sizeof(float) * npend * step) == 0) flags |= 1;
            if (ys.sym && m - (npend - 1) >= c0 && m < c1) flags |= 2;  // outputs m, m-1, .., m-npend+1 all canonical
            memcpy(&tab[(size_t)m * yw + yw - 1], &flags, 4);
        }
        HIP_TRY(hipMalloc(&ys.blob, tab.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(ys.blob, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
        ys.step_weight = static_cast<float *>(ys.blob);
    }
    auto ins = ay->ystreams.emplace(key, ys);
    *out = &ins.first->second;
    return PICHA_OK;
}

// Per-source-row view of the vertical table for the accumulator kernel: output y accumulates in register set
// y % nacc while source rows left[y] .. left[y] + count[y] - 1 go by.  Representable when the vertical windows are
// plain consecutive rows (no ring aliasing, src/resize.cc:52-64 quirk) and outputs y and y + nacc are never pending
// at the same time.
// allow_look: accept tables whose first tap is ring-aliased onto the window's last row (lanczos / box at even integer scales: the
// window is one row longer than the ring).  The reference's sum for such an output row is
//     w0 * h(last) + w1 * h(left + 1) + ... + w[n-1] * h(last)          (row `left` itself is never read)
// in that order, so the accumulator has to START from w0 * h(last).  The march therefore follows a LIST of source rows instead of
// the rows themselves: every row in ascending order, and in front of row `left` of each aliased window that window's last row
// once more, with a record that carries w0 in the output's accumulator (free and zero at that point: 0 + w0 * h is w0 * h) and
// nothing else.  One extra horizontal pass per OUTPUT row; the kernel only needs the indirection position -> source row.
int get_yrows(DeviceCtx *ctx, DeviceAxis *ay, int nacc, const DeviceAxis::YRows **out, bool allow_look = false) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    const int key = nacc + (allow_look ? 1000 : 0);
    auto it = ay->yrows.find(key);
    if (it != ay->yrows.end()) { *out = &it->second; return PICHA_OK; }
    DeviceAxis::YRows yr;
    const AxisTaps &t = ay->host;
    const int dst = t.dst_size, src = t.src_size, T = t.taps;
    bool ok = true, any_look = false;
    std::vector<uint8_t> look(dst, 0);
    for (int y = 0; y < dst && ok; ++y) {
        const int n = t.count[y];
        look[y] = allow_look && n >= 2 && t.eff[(size_t)y * T] == t.left[y] + n - 1;
        for (int j = look[y] ? 1 : 0; j < n && ok; ++j)
            if (t.eff[(size_t)y * T + j] != t.left[y] + j) ok = false;  // any other ring aliasing: not representable
        any_look |= look[y] != 0;
        if (y >= nacc && t.left[y - nacc] + t.count[y - nacc] - 1 >= t.left[y]) ok = false;  // accumulator still busy
        if (look[y] && y > 0 && t.left[y] < t.left[y - 1]) ok = false;                        // (list positions assume ascending windows)
    }
    yr.ok = ok;
    if (ok) {
        // list position of every source row, and of the extra entry in front of each aliased window
        std::vector<int> pos_of_row(src + 1, 0), pos_of_init(dst, -1), src_of;
        {
            std::vector<std::vector<int>> starts(src);
            for (int y = 0; y < dst; ++y)
                if (look[y]) starts[t.left[y]].push_back(y);
            for (int r = 0; r < src; ++r) {
                for (int y : starts[r]) { pos_of_init[y] = (int)src_of.size(); src_of.push_back(t.left[y] + t.count[y] - 1); }
                pos_of_row[r] = (int)src_of.size();
                src_of.push_back(r);
            }
        }
        const int rows = (int)src_of.size() + 8;  // ghost entries past the end: empty records, the last row again
        const size_t rec_words = (size_t)rows * 2 * nacc;
        std::vector<int32_t> tab(rec_words + (size_t)dst * 2 + (any_look ? rows : 0), 0);
        for (int r = 0; r < rows; ++r)
            for (int a = 0; a < nacc; ++a) tab[(size_t)r * 2 * nacc + nacc + a] = -1;
        for (int y = 0; y < dst; ++y) {
            const int a = y % nacc, n = t.count[y];
            for (int j = look[y] ? 1 : 0; j < n; ++j) {
                const float w = t.weight[(size_t)y * T + j];
                memcpy(&tab[(size_t)pos_of_row[t.left[y] + j] * 2 * nacc + a], &w, 4);
            }
            if (look[y]) memcpy(&tab[(size_t)pos_of_init[y] * 2 * nacc + a], &t.weight[(size_t)y * T], 4);
            tab[(size_t)pos_of_row[t.left[y] + n - 1] * 2 * nacc + nacc + a] = y;
        }
        int32_t *range = tab.data() + rec_words;
        for (int y = 0; y < dst; ++y) {
            range[2 * y] = look[y] ? pos_of_init[y] : pos_of_row[t.left[y]];
            range[2 * y + 1] = pos_of_row[t.left[y] + t.count[y] - 1];
        }
        for (int y = dst - 2; y >= 0; --y) range[2 * y] = std::min(range[2 * y], range[2 * (y + 1)]);   // see y_info.x
        for (int y = 1; y < dst; ++y) range[2 * y + 1] = std::max(range[2 * y + 1], range[2 * (y - 1) + 1]);
        if (any_look) {
            int32_t *sr = tab.data() + rec_words + (size_t)dst * 2;
            for (int i = 0; i < rows; ++i) sr[i] = i < (int)src_of.size() ? src_of[i] : src - 1;
        }
        HIP_TRY(hipMalloc(&yr.blob, tab.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(yr.blob, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        yr.rec = static_cast<int32_t *>(yr.blob);
        yr.range = yr.rec + rec_words;
        if (any_look) yr.src = yr.range + (size_t)dst * 2;
    }
    auto ins = ay->yrows.emplace(key, yr);
    *out = &ins.first->second;
    return PICHA_OK;
}

// Run kernel (kernels_run.hip), x role.  The table is periodic with `period` outputs / `period_px` source pixels when, for the
// canonical window start  c(x) = x_base + (x / period) * period_px + ((x % period) * period_px + ceil * (period - 1)) / period  (offsets
// inside a period rounded down or up: whichever needs fewer padding taps),  every output's taps lie
// inside [c(x), c(x) + taps) and its zero-padded weights equal those of the middle period's output x % period bit for bit.  Columns
// that fail (windows clipped by an image edge carry renormalised weights) must be a few at either end of the row — [0, c0) and
// [c1, dst), at most kMaxRunEdge each — which the kernel computes one pixel per lane from the generic tables.  `pad_to` > 0: pad the window to that many taps (an instantiated count).
// wmode: 0 reference weights, 1 times 1/max (fma flavour), 2 times 2^22 (formats that unpack through the denormal form).
constexpr int kMaxRunEdge = 16;
int run_table_taps(const AxisTaps &t, int period, int period_px, int ceil_off, int *x_base_out) {
    const int dst = t.dst_size;
    if (dst < 4 * period) return 0;
    auto offg = [&](int x) { return (x / period) * period_px + ((x % period) * period_px + ceil_off * (period - 1)) / period; };
    const int mid = (dst / 2) / period * period;
    int xb = INT32_MAX, tx = 0;
    for (int q = 0; q < period; ++q) xb = std::min(xb, t.left[mid + q] - offg(mid + q));
    for (int q = 0; q < period; ++q) tx = std::max(tx, t.left[mid + q] - (xb + offg(mid + q)) + t.count[mid + q]);
    *x_base_out = xb;
    return tx;
}
int get_runtable(DeviceCtx *ctx, DeviceAxis *ax, int period, int period_px, int ceil_off, int pad_to, int wmode, int is16, const DeviceAxis::RunTable **out) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto key = std::make_tuple(period, period_px, pad_to, wmode + 4 * is16 + 8 * ceil_off);
    auto it = ax->runtables.find(key);
    if (it != ax->runtables.end()) { *out = &it->second; return PICHA_OK; }
    DeviceAxis::RunTable rt;
    const AxisTaps &t = ax->host;
    const int dst = t.dst_size, T = t.taps;
    int xb = 0;
    const int tx_min = run_table_taps(t, period, period_px, ceil_off, &xb);
    const int TX = std::max(tx_min, pad_to);
    bool ok = tx_min > 0 && TX <= 64;
    std::vector<float> canon;
    int c0 = 0, c1 = dst;
    if (ok) {
        auto offg = [&](int x) { return (x / period) * period_px + ((x % period) * period_px + ceil_off * (period - 1)) / period; };
        const float inv = is16 ? 1.0f / 65535.0f : 1.0f / 255.0f;
        auto padded = [&](int x, float *v) {  // false: the taps do not fit the canonical window
            const int first = t.left[x] - (xb + offg(x));
            if (first < 0 || first + t.count[x] > TX) return false;
            for (int j = 0; j < TX; ++j) v[j] = 0.0f;
            for (int j = 0; j < t.count[x]; ++j) {
                const float w = t.weight[(size_t)x * T + j];
                v[first + j] = wmode == 1 ? w * inv : wmode == 2 ? w * 4194304.0f : w;
            }
            return true;
        };
        const int mid = (dst / 2) / period * period;
        canon.assign((size_t)period * TX, 0.0f);
        for (int q = 0; q < period && ok; ++q) ok = padded(mid + q, &canon[(size_t)q * TX]);
        std::vector<float> v(TX);
        auto is_canon = [&](int x) { return padded(x, v.data()) && memcmp(v.data(), &canon[(size_t)(x % period) * TX], sizeof(float) * TX) == 0; };
        while (ok && c0 < dst && !is_canon(c0)) ++c0;
        while (ok && c1 > c0 && !is_canon(c1 - 1)) --c1;
        for (int x = c0; x < c1 && ok; ++x) ok = is_canon(x);
        if (c0 > kMaxRunEdge || dst - c1 > kMaxRunEdge || c1 - c0 < 2 * period) ok = false;
    }
    rt.ok = ok;
    if (ok) {
        HIP_TRY(hipMalloc(&rt.blob, sizeof(float) * canon.size()));
        HIP_TRY(hipMemcpy(rt.blob, canon.data(), sizeof(float) * canon.size(), hipMemcpyHostToDevice));
        rt.canon = static_cast<float *>(rt.blob);
        rt.c0 = c0; rt.c1 = c1; rt.x_base = xb; rt.taps = TX;
    }
    auto ins = ax->runtables.emplace(key, rt);
    *out = &ins.first->second;
    return PICHA_OK;
}

// ---- validation ----------------------------------------------------------------------------------
int check_src(const picha_image *im) {
    // jsImageToNativeImage (src/picha.cc:61-85): valid pixel, data present, height != 0
    if (!im || !im->data || !valid_pixel(im->pixel) || im->height <= 0 || im->width <= 0 || im->stride < 0)
        return fail(PICHA_ERR_INVALID_IMAGE, "invalid image");
    return PICHA_OK;
}
int check_dst(const picha_image *im) {
    if (!im || !im->data || !valid_pixel(im->pixel)) return fail(PICHA_ERR_INVALID_IMAGE, "invalid image");
    if (im->width <= 0 || im->height <= 0) return fail(PICHA_ERR_INVALID_DIMENSIONS, "invalid dimensions");
    // 64-bit arithmetic: width * bytes-per-pixel must fit the int stride (the reference's int row_stride wraps instead,
    // src/picha.h:214 — a front end must not turn caller-supplied dimensions into a heap overflow)
    const long long rw = (long long)im->width * kBytes[im->pixel];
    if (rw > INT32_MAX - 3) return fail(PICHA_ERR_INVALID_DIMENSIONS, "invalid dimensions");
    if ((long long)im->stride < rw) return fail(PICHA_ERR_INVALID_IMAGE, "invalid image");
    return PICHA_OK;
}
size_t image_span(const picha_image &im) {  // bytes from data to the end of the last row's payload
    return (size_t)(im.height - 1) * (size_t)im.stride + (size_t)im.width * kBytes[im.pixel];
}

bool same_geometry(const picha_image &a, const picha_image &b) {
    return a.stride == b.stride && a.width == b.width && a.height == b.height && a.pixel == b.pixel;
}

// natural alignment the fused kernel's typed loads need
int fused_align(int pixel) {
    switch (pixel) {
    case PICHA_RGB: case PICHA_GREY: return 1;
    case PICHA_GREYA: case PICHA_R16: case PICHA_R16G16B16: return 2;
    case PICHA_RGBA: case PICHA_R16G16: return 4;
    default: return 8;
    }
}

struct TablePair {  // device pointer tables for a group with non-uniform pitch
    const void *const *src = nullptr;
    void *const *dst = nullptr;
    bool used = false;
};

// Uniform pitch?  (data[i] = data[0] + i * pitch)
bool uniform_pitch(const picha_image *im, int n, long long *pitch) {
    *pitch = 0;
    if (n <= 1) return true;
    const long long p = (const char *)im[1].data - (const char *)im[0].data;
    for (int i = 2; i < n; ++i)
        if ((const char *)im[i].data - (const char *)im[i - 1].data != p) return false;
    *pitch = p;
    return true;
}

// Caller holds ctx->temp_mu and calls temp_release(ctx->tables, s) after the launches.
int make_tables(DeviceCtx *ctx, const picha_image *src, const picha_image *dst, int n, hipStream_t s, TablePair *t) {
    const size_t bytes = sizeof(void *) * 2 * (size_t)n;
    int rc = temp_acquire(ctx->tables, bytes, bytes, s);
    if (rc) return rc;
    const void **h = static_cast<const void **>(ctx->tables.host);
    for (int i = 0; i < n; ++i) { h[i] = src[i].data; h[n + i] = dst[i].data; }
    HIP_TRY(hipMemcpyAsync(ctx->tables.dev, h, bytes, hipMemcpyHostToDevice, s));
    t->src = static_cast<const void *const *>(ctx->tables.dev);
    t->dst = reinterpret_cast<void *const *>(static_cast<void **>(ctx->tables.dev) + n);
    t->used = true;
    return PICHA_OK;
}

// Where the accumulator kernel with two outputs per lane was measured faster than with one (filled from the r02 survey,
// profiles/r02_survey_acc_p2.txt): (format, integer step, padded taps, flavour)
bool acc_p2_preferred(int pixel, int step, int taps, int fma) {
    if (fma) return false;  // only the reference-arithmetic flavour was surveyed
    switch (pixel) {
    case PICHA_RGBA: return (step == 4 && (taps == 12 || taps == 16)) || (step == 3 && taps == 12);   // -7.9 %, -6.8 %, -3.3 %
    case PICHA_RGB: return step == 4 && taps == 12;                                                   // -3.5 %
    case PICHA_GREY: return step == 4 && taps == 16;                                                  // -8.4 %
    case PICHA_R16: return step == 2 && taps == 8;                                                    // -7.5 %
    }
    return false;
}

// where the neighbour-exchange form of the symmetric-table kernel is the default (measured, profiles/r03_ab_xch*.txt)
bool sym_xch_preferred(int pixel, int step, int taps, int per_lane, int fma) {
    (void)pixel; (void)step; (void)taps; (void)per_lane; (void)fma;
    return false;
}

// default work mode of a launch with `n` images, `col_waves` waves per output row and `chunks` row chunks (0 = plain grid)
// Measured r03 (profiles/r03_ab_work_items.txt): cfg2 on the symmetric-table kernel -2.3 .. -3.2 % with the explicit list (the
// tail of short chunks), the persistent form +6 % there (an atomic and two barriers per item, no launch-time backfill to save);
// cfg5 on the neighbour-exchange accumulator kernel, whose plain grid is only 2.7 rounds of resident workgroups: list -1.3 %,
// persistent -2 %.  Small launches keep the plain grid: the list only pays once a launch is several rounds long.
// Everything else on the accumulator kernel keeps the plain grid: on short launches (64 x 960x540 upscales, 0.1-0.4 ms) the list is
// 0-5 % and the persistent form 10-100 % slower (profiles/r03_survey_upscale.txt).
int work_default(int n, long long col_waves, int chunks, bool acc_kernel, bool acc_xch) {
    const long long waves = (long long)n * col_waves * chunks;
    if (waves < 6144) return 0;
    if (acc_kernel) return acc_xch ? 2 : 0;
    return 1;
}

// where the neighbour-exchange form of the accumulator kernel is the default (measured r03, profiles/r03_ab_xch.txt): cfg5's
// shape — r16g16b16a16 at 4x with 15 taps, every source byte requested by four lanes in the plain form: -8 % (reference
// arithmetic), -11 % (fma).  rgba at 3x / 4x (8 and 11 taps) and the symmetric-table kernel at 2x lose 1-13 % with it: the
// L1 is not their limit and the DPP moves and the edge lanes' extra load cost more than the narrower loads save.
bool acc_xch_preferred(int pixel, int step, int taps, int per_lane, int fma) {
    (void)fma;
    return pixel == PICHA_R16G16B16A16 && step == 4 && taps == 15 && per_lane == 2;
}

// Outputs per lane for which resize_sym_kernel beats the general streaming / accumulator kernels by >= 2 % (0: it does not):
// interleaved survey of all 8 formats x {2x: 3, 5, 7 taps; 3x: 8, 12; 4x: 11, 15} on 64 x 4K images, both numerics flavours
// (scripts/survey_sym.py, profiles/r02_survey_sym_exact.txt, r02_survey_sym_fma.txt).
int sym_preferred_per_lane(int pixel, int step, int taps, int fma) {
    struct E { int8_t step, taps, exact, fma; };
    static const E rgb[] = {{2, 5, 0, 4}, {3, 8, 4, 0}, {4, 11, 0, 4}, {4, 15, 0, 4}, {0, 0, 0, 0}};
    static const E rgba[] = {{2, 5, 3, 3} /* fma: 3 outputs per lane 1.883 -> 1.858 ms and less fabric traffic (r04, profiles/r04_ab_cfg2_per_lane.txt) */, {2, 7, 0, 2}, {3, 8, 2, 2}, {4, 11, 0, 2} /* exact: the two-output accumulator kernel is ahead */, {4, 15, 0, 2}, {0, 0, 0, 0}};
    static const E grey[] = {{2, 5, 0, 4}, {2, 7, 0, 4}, {0, 0, 0, 0}};
    static const E greya[] = {{2, 3, 0, 2}, {2, 5, 0, 4}, {2, 7, 0, 4}, {3, 8, 2, 2}, {3, 12, 2, 0}, {0, 0, 0, 0}};
    static const E r16[] = {{2, 3, 4, 2}, {2, 5, 4, 4}, {2, 7, 4, 4}, {3, 8, 2, 2}, {3, 12, 2, 2}, {4, 11, 2, 2}, {0, 0, 0, 0}};
    static const E r16g16[] = {{2, 3, 2, 2}, {2, 5, 4, 2}, {2, 7, 3, 4}, {3, 12, 0, 2}, {4, 15, 0, 2}, {0, 0, 0, 0}};
    static const E r16g16b16[] = {{2, 3, 2, 2}, {2, 5, 2, 2}, {2, 7, 2, 2}, {3, 12, 0, 2}, {0, 0, 0, 0}};
    static const E *const tabs[8] = {rgb, rgba, grey, greya, r16, r16g16, r16g16b16, r16g16b16 /* r16g16b16a16: same picks */};
    if (pixel < 0 || pixel > 7) return 0;
    for (const E *e = tabs[pixel]; e->step; ++e)
        if (e->step == step && e->taps == taps) return fma ? e->fma : e->exact;
    return 0;
}

// Work items of a launch (kernels.h, WorkList).  Units = (image, long row chunk) in batch order, dealt to the 8 XCDs in
// contiguous eighths; inside an eighth the last `tail_pm` per mille of the units are cut into short chunks, and every
// (image, rows) piece becomes `nsb` consecutive items, one per column-strip block.  Sub-lists are padded to equal length
// with empty items.  Built once per key, then a lookup.
int get_items(DeviceCtx *ctx, int n, int dst_h, int nsb, int rows_long, int rows_short, int tail_pm, std::shared_ptr<const DeviceCtx::ItemList> *out) {
    std::shared_ptr<DeviceCtx::ItemList> evicted;   // released after `mu`: its hipFree synchronises the device
    std::unique_lock<std::mutex> lk(ctx->mu);
    auto key = std::make_tuple(n, dst_h, nsb, rows_long, rows_short, tail_pm);
    auto it = ctx->item_lists.find(key);
    if (it != ctx->item_lists.end()) { it->second->used = ++ctx->use_tick; *out = it->second; return PICHA_OK; }
    if (ctx->item_lists.size() >= kMaxItemLists) {
        auto lru = ctx->item_lists.begin();
        for (auto j = ctx->item_lists.begin(); j != ctx->item_lists.end(); ++j)
            if (j->second->used < lru->second->used) lru = j;
        evicted = std::move(lru->second);
        ctx->item_lists.erase(lru);
    }
    struct Unit { int img, y0, y1; };
    std::vector<Unit> units;
    const int chunks = (dst_h + rows_long - 1) / rows_long;
    const int rl = (dst_h + chunks - 1) / chunks;
    for (int i = 0; i < n; ++i)
        for (int y = 0; y < dst_h; y += rl) units.push_back({i, y, std::min(dst_h, y + rl)});
    std::vector<int4> lists[8];
    const size_t U = units.size();
    for (int x = 0; x < 8; ++x) {
        const size_t lo = U * x / 8, hi = U * (x + 1) / 8;
        const size_t tail = std::min<size_t>(hi - lo, ((hi - lo) * (size_t)tail_pm + 999) / 1000);
        for (size_t k = lo; k < hi; ++k) {
            const Unit &u = units[k];
            const int step = (k >= hi - tail) ? std::max(1, rows_short) : (u.y1 - u.y0);
            // equal short chunks: ceil split
            const int pieces = (u.y1 - u.y0 + step - 1) / step;
            const int rows = (u.y1 - u.y0 + pieces - 1) / pieces;
            for (int y = u.y0; y < u.y1; y += rows)
                for (int b = 0; b < nsb; ++b) lists[x].push_back(make_int4(b, y, std::min(u.y1, y + rows), u.img));
        }
    }
    size_t per = 1;
    for (auto &l : lists) per = std::max(per, l.size());
    std::vector<int4> flat(per * 8, make_int4(0, 0, 0, 0));
    for (int x = 0; x < 8; ++x) std::copy(lists[x].begin(), lists[x].end(), flat.begin() + (long)(per * x));
    auto il = std::make_shared<DeviceCtx::ItemList>();
    HIP_TRY(hipMalloc(&il->blob, flat.size() * sizeof(int4)));
    HIP_TRY(hipMemcpy(il->blob, flat.data(), flat.size() * sizeof(int4), hipMemcpyHostToDevice));
    il->items = static_cast<const int4 *>(il->blob);
    il->per_xcd = (int)per;
    il->used = ++ctx->use_tick;
    ctx->item_lists.emplace(key, il);
    *out = il;
    lk.unlock();
    evicted.reset();
    return PICHA_OK;
}

// the stream's queue counters, zeroed in stream order for the launch that follows; *hold keeps them alive for the caller's launch
int get_queue(DeviceCtx *ctx, hipStream_t s, unsigned **out, std::shared_ptr<const DeviceCtx::Queue> *hold) {
    std::shared_ptr<DeviceCtx::Queue> q, evicted;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        auto it = ctx->queues.find(s);
        if (it != ctx->queues.end()) q = it->second;
        else {
            if (ctx->queues.size() >= kMaxQueues) {
                auto lru = ctx->queues.begin();
                for (auto j = ctx->queues.begin(); j != ctx->queues.end(); ++j)
                    if (j->second->used < lru->second->used) lru = j;
                evicted = std::move(lru->second);
                ctx->queues.erase(lru);
            }
            q = std::make_shared<DeviceCtx::Queue>();
            HIP_TRY(hipMalloc((void **)&q->counters, 64));
            ctx->queues[s] = q;
        }
        q->used = ++ctx->use_tick;
    }
    evicted.reset();   // outside `mu`
    HIP_TRY(hipMemsetAsync(q->counters, 0, 64, s));
    *out = q->counters;
    *hold = q;
    return PICHA_OK;
}

// Column strips of the row-staged kernel: strip b = groups [256 b, 256 b + 256); it stages the source pixels from
// strip_left[b] (a multiple of `piece_px`, possibly negative) up to the end of its last window, `pieces` x 256 x piece_px
// pixels per row.
int get_rows_strips(DeviceCtx *ctx, DeviceAxis *ax, const DeviceAxis::XTable *xt, int span_px, int piece_px,
                    const DeviceAxis::RowsStrips **out) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    constexpr int tw = 256;   // threads per strip (r05: wave-private strips of 64 threads, no barrier, measured a tie on cfg3 and removed: profiles/r05_ab_rows_wave_strips.txt)
    auto key = std::make_tuple((const void *)xt, span_px, piece_px);
    auto it = ax->rows_strips.find(key);
    if (it != ax->rows_strips.end()) { *out = &it->second; return PICHA_OK; }
    DeviceAxis::RowsStrips rs;
    const int groups = xt->groups;
    const int n_strips = (groups + tw - 1) / tw;
    std::vector<int32_t> sl(n_strips);
    int pieces = 1;
    for (int b = 0; b < n_strips; ++b) {
        int lo = INT32_MAX, hi = INT32_MIN;
        for (int g = b * tw; g < std::min(groups, b * tw + tw); ++g) {
            lo = std::min(lo, xt->host_left[g]);
            hi = std::max(hi, xt->host_left[g] + span_px);
        }
        lo -= ((lo % piece_px) + piece_px) % piece_px;  // floor to a stage piece
        sl[b] = lo;
        const int need = (hi - lo + 3 + tw * piece_px - 1) / (tw * piece_px);  // +3: the last 16-byte read of a span may run past its end
        while (pieces < need) pieces *= 2;
    }
    pieces = std::max(pieces, 2);  // instantiated: 2, 4, 8
    rs.ok = pieces <= 8;
    rs.n_strips = n_strips;
    rs.pieces = pieces;
    if (rs.ok) {
        HIP_TRY(hipMalloc(&rs.blob, sizeof(int32_t) * n_strips));
        HIP_TRY(hipMemcpy(rs.blob, sl.data(), sizeof(int32_t) * n_strips, hipMemcpyHostToDevice));
        rs.strip_left = static_cast<int32_t *>(rs.blob);
    }
    auto ins = ax->rows_strips.emplace(key, rs);
    *out = &ins.first->second;
    return PICHA_OK;
}

long g_opt_fuse = 0;  // chained convert+resize: 0 auto (one fused kernel where it exists and was measured faster), 1 one kernel wherever it exists, 2 always two kernels
long g_opt_rows = 0;  // row-staged (LDS) downscale kernel: 0 auto, 1 whenever it applies, 2 never

// y role of resize_up_kernel (and of resize_edges_kernel behind it): the source rows a block of output rows needs
int get_yrange(DeviceCtx *ctx, DeviceAxis *ay, const int32_t **out) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!ay->yrange) {
        const AxisTaps &t = ay->host;
        const int dst = t.dst_size;
        std::vector<int32_t> range((size_t)dst * 2);
        // (last row: the one the reference has resampled when it produces the output row, need[] >= the window's last row)
        for (int y = 0; y < dst; ++y) { range[2 * y] = t.left[y]; range[2 * y + 1] = std::max(t.need[y], t.left[y] + t.count[y] - 1); }
        for (int y = dst - 2; y >= 0; --y) range[2 * y] = std::min(range[2 * y], range[2 * (y + 1)]);
        for (int y = 1; y < dst; ++y) range[2 * y + 1] = std::max(range[2 * y + 1], range[2 * (y - 1) + 1]);
        HIP_TRY(hipMalloc((void **)&ay->yrange, range.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(ay->yrange, range.data(), range.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    *out = ay->yrange;
    return PICHA_OK;
}

long g_opt_run = 0;   // run kernel (periodic tap tables, kernels_run.hip): 0 auto, 1 whenever it applies, 2 never
long g_opt_run_round = 0;  // its window offsets inside a period: 0 auto (the shorter window), 1 rounded down, 2 rounded up
long g_opt_run_xch = 0;    // its neighbour-exchange form (dword-sized pixels, integer scales; r04): 0 auto, 1 wherever instantiated, 2 never
long g_opt_run_ws = 0;     // weights of one phase of the period in SGPRs (r04): 0 auto, 1 wherever instantiated, 2 never
long g_opt_edge_rows = 0;   // most output rows per block of the edge kernel (0: 64, what its 2048 LDS floats allow): smaller blocks = more, shorter waves
#define PICHA_EFUSE_DEFAULT(o) ((o) == 1)   /* auto = off until measured */
long g_opt_run_efuse = 0;  // (r05, third session) the edge blocks inside the march's grid (RunParams::edge_fused; cfg3b's shared-unpack shape, plain grids): 0 auto, 1 on, 2 off
long g_opt_run_shu = 0;    // shared unpack through LDS (r05; 8-bit grey / greya, kernels_run.hip RunShu): 0 auto (wherever instantiated), 1 the same, 2 never
long g_opt_run_edges = 0;  // where resize_edges_kernel takes the clipped columns' windows from (it follows the march on the same stream either way): 0 auto, 1 a stash the march
                           // writes (r03 form; the image where the form has no stash), 2 the image, 3 = 0.  Auto = the measured winner (profiles/r05_ab_run_edges_shu.txt): the
                           // stash for the plain grey run kernel only (-1.3 .. -1.9 % against the image); greya, the ring / upscale kernels and the shared-unpack form read
                           // the image (greya 7.5x -5 %, rgba 2x upscale -4 %, grey lanczos 2x -2.6 %: their stash rows are longer than the lines the image costs).
                           // (r05, measured and removed: the edge kernel on a second stream BESIDE a march that leaves those columns alone — +5 .. +9 %, the
                           // cross-stream events cost more than the 0.03-0.04 ms the kernel takes; same file)
long g_opt_run_phase = 0;  // time-phased stores of the neighbour-exchange run kernel (r04): 0 auto, -1 off, n > 0: phases of 2^n ticks of the 100 MHz counter (12 = 41 us)

// where the neighbour-exchange run kernel is the default: cfg5's shape (r16g16b16a16 at 4x, 15 taps) — measured r04 against the
// r03 accumulator form (lane-resident weights, 157 VGPRs, 3 waves per SIMD): profiles/r04_ab_run_xch.txt
// (2.03 -> 1.87-1.93 ms, fma 1.94 -> 1.75-1.79), the same format with picha's default 11 taps (64 x 4K -> 960x540: 1.05 -> 0.92 ms, fma
// 1.05 -> 0.90), and rgba at 4x in the fma flavour (11 taps 1.01 -> 0.93, 15 taps 1.10 -> 1.02; the reference arithmetic is 6-20 % slower
// there, and rgba at 3x loses in both): profiles/r04_survey_run_xch.txt
bool run_xch_preferred(int pixel, int period_px, int taps, int per_lane, int fma) {
    if (per_lane != 2 || period_px != 4 || (taps != 11 && taps != 15)) return false;
    return pixel == PICHA_R16G16B16A16 || (pixel == PICHA_RGBA && fma);
}

// PICHA_HIP_TRACE_PLAN=1: why run_group left a geometry to the other kernels (the line of the rule that declined it)
#define RUN_GROUP_NA do { static const bool trace_na = getenv("PICHA_HIP_TRACE_PLAN") != nullptr; if (trace_na) fprintf(stderr, "picha_hip: run_group declines at capi.cpp:%d\n", __LINE__); return PICHA_OK; } while (0)
// Run kernel for one equal-geometry group: downscales whose x scale is a ratio of small integers with a long window.
// *used = false: not applicable, the caller goes on with the other kernels.
// cv != nullptr (r04): src[] are in another pixel format than dst[] — picha.colorConvert to dst[].pixel, then picha.resize, as ONE kernel: integer
// upscales only (the convert runs per window pixel in the ring kernel, kernels_run.hip); the result is the two-call chain's, byte for byte.
struct ChainConvert { float r, g, b; };
int run_group(DeviceCtx *ctx, const picha_image *src, const picha_image *dst, int n, int filter, float fscale, hipStream_t s, bool *used, const ChainConvert *cv = nullptr) {
    *used = false;
    const picha_image &s0 = src[0], &d0 = dst[0];
    const int pixel = d0.pixel, spixel = s0.pixel;   // the format the kernel computes in / the format it loads
    long opt_run, opt_run_round, opt_numerics, opt_chunks, opt_per_lane, opt_work, opt_work_tail, opt_work_tail_div, opt_bw, opt_run_xch, opt_run_ws, opt_run_phase, opt_run_edges, opt_run_shu, opt_run_efuse;
    {
        std::lock_guard<std::mutex> lk(g_opt_mu);
        opt_run_xch = g_opt_run_xch; opt_run_ws = g_opt_run_ws; opt_run_phase = g_opt_run_phase; opt_run_edges = g_opt_run_edges; opt_run_shu = g_opt_run_shu; opt_run_efuse = g_opt_run_efuse;
        if (g_opt_xch != 0 && opt_run_xch == 0) opt_run_xch = 2;   // "xch" = 1 / 2 asks for / rules out the exchange forms of the symmetric-table and accumulator kernels: not this one
        opt_run = g_opt_run; opt_run_round = g_opt_run_round; opt_numerics = tl_numerics >= 0 ? tl_numerics : g_opt_numerics; opt_chunks = g_opt_row_chunks; opt_per_lane = g_opt_per_lane;
        opt_work = g_opt_work; opt_work_tail = std::max(0L, std::min(1000L, g_opt_work_tail)); opt_work_tail_div = std::max(1L, g_opt_work_tail_div);
        opt_bw = g_opt_block_waves;
    }
    if (opt_run == 2 || ((s0.pixel != d0.pixel) != (cv != nullptr))) RUN_GROUP_NA;
    // downscales (the march + per-source-row records), or upscales by an integer factor in x (resize_up_kernel: register ring in y)
    const bool up = d0.width > s0.width && d0.height > s0.height;
    if (!up && (cv || d0.width >= s0.width || d0.height >= s0.height)) RUN_GROUP_NA;
    if (up && (d0.width % s0.width != 0 || d0.width / s0.width > 4)) RUN_GROUP_NA;
    const size_t src_span = image_span(s0), dst_span = image_span(d0);
    if (src_span >= (1u << 30) || dst_span >= (1u << 30)) RUN_GROUP_NA;
    const int al = fused_align(spixel);
    for (int i = 0; i < n; ++i) {
        if (((uintptr_t)src[i].data | (uintptr_t)src[i].stride) % al) RUN_GROUP_NA;
        if (((uintptr_t)dst[i].data | (uintptr_t)dst[i].stride) % 4) RUN_GROUP_NA;  // a lane stores its run as whole dwords
        if (src[i].stride % 4) RUN_GROUP_NA;  // the edge pass fetches a block of rows at one byte phase
    }
    // the period: scale = period_px / period in lowest terms
    int g = s0.width, b = d0.width;
    while (b) { const int t = g % b; g = b; b = t; }
    const int period = d0.width / g, period_px = s0.width / g;
    if (period > 4) RUN_GROUP_NA;
    std::shared_ptr<DeviceAxis> ax_ref, ay_ref;
    int rc = get_axis(ctx, filter, fscale, s0.width, d0.width, &ax_ref);
    if (rc) return rc;
    rc = get_axis(ctx, filter, fscale, s0.height, d0.height, &ay_ref);
    if (rc) return rc;
    DeviceAxis *ax = ax_ref.get(), *ay = ay_ref.get();
    const int fma = opt_numerics == 1 ? 1 : 0;
    const int is16 = pixel >= PICHA_R16;
    const int wmode = fma ? 1 : exact_unpack_scaled(kChannels[pixel], is16 != 0) ? 2 : 0;
    // shape: the smallest instantiated tap count >= the table's, outputs per lane as instantiated ("per_lane" forces one)
    const DeviceAxis::YRows *yr = nullptr;
    const int32_t *up_range = nullptr;
    int nacc = 0, per_lane = 0, ktaps = 0, ceil_off = 0;
    // Three vertical schemes: the march with per-source-row records (downscales without ring aliasing); the ring-in-registers kernel
    // for upscales (ring of four rows) and for ring-aliased downscales whose reference ring is 8 or 16 rows (lanczos at 2x / 4x; at odd factors its windows do not alias).
    // The register ring reproduces the reference's ring whenever it has the reference's length, or the table has no aliasing and
    // no output row needs more rows than the ring holds (taps.cpp: eff[] = the last row <= need[] in the tap's residue class).
    int ring_rows = 0;
    {   //  no instantiation for this format / period at all (rgb, most 16-bit formats): leave before any device table is
        // built — get_yrows below allocates and uploads per axis
        int xb0 = 0;
        bool any = false;
        for (int c = 0; c < 2 && !any; ++c) {
            const int t0 = run_table_taps(ax->host, period, period_px, c, &xb0);
            for (int tx = t0; t0 > 0 && tx <= t0 + 3 && !any; ++tx)
                for (int pl : {2, 4, 8}) {
                    if (cv) any = any || up_cv_kernel_supported(pixel, spixel, period, period_px, tx, pl, c, 4, ay->host.taps);
                    else if (up) { for (int k : {4}) any = any || up_kernel_supported(pixel, period, period_px, tx, pl, c, k, ay->host.taps); }
                    else {
                        for (int na : {4, 6})
                            any = any || run_kernel_supported(pixel, period, period_px, tx, pl, na, c) || run_kernel_supported(pixel, period, period_px, tx, pl, na, c, -1, 1) ||
                                  run_kernel_supported(pixel, period, period_px, tx, pl, na, c, -2, std::min(2, period)) ||
                                  (c == 0 && xb0 <= 0 && run_kernel_supported(pixel, period, period_px, tx, pl, na, 0, -xb0, 0));
                        for (int k : {8, 16}) any = any || up_kernel_supported(pixel, period, period_px, tx, pl, c, k, ay->host.taps);
                    }
                }
        }
        if (!any) RUN_GROUP_NA;
    }
    bool ylist = false;   // (r05) ring-aliased table: the march follows a list of source rows (shared-unpack form only)
    if (!up) {
        for (int na : {4, 6}) {
            const DeviceAxis::YRows *cand = nullptr;
            rc = get_yrows(ctx, ay, na, &cand);
            if (rc) return rc;
            if (cand->ok) { yr = cand; nacc = na; break; }
        }
        bool shu_shape = false;   // (only the shared-unpack shapes follow a row list; every other aliased table goes on to the ring kernel below)
        for (int c = 0; c < 2 && !shu_shape; ++c) {
            int xb0 = 0;
            const int t0 = run_table_taps(ax->host, period, period_px, c, &xb0);
            for (int tx = t0; t0 > 0 && tx <= t0 + 3 && !shu_shape; ++tx)
                for (int pl : {2, 4, 8})
                    for (int na : {4, 6}) shu_shape = shu_shape || run_kernel_supported(pixel, period, period_px, tx, pl, na, c, -2, std::min(2, period));
        }
        if (!yr && !cv && shu_shape && opt_run_shu != 2 && opt_run_ws != 2 && opt_run_edges != 1) {
            bool row0_aliased = false;   // (the kernel's in-loop loads assume source row 0 sits at list position 0 only)
            const AxisTaps &ty = ay->host;
            for (int y = 0; y < d0.height; ++y)
                if (ty.left[y] == 0 && ty.count[y] >= 2 && ty.eff[(size_t)y * ty.taps] == ty.count[y] - 1) row0_aliased = true;
            for (int na : {4, 6}) {
                if (row0_aliased) break;
                const DeviceAxis::YRows *cand = nullptr;
                rc = get_yrows(ctx, ay, na, &cand, true);
                if (rc) return rc;
                if (cand->ok && cand->src) { yr = cand; nacc = na; ylist = true; break; }
            }
        }
    }
    if (!yr) {
        const AxisTaps &ty = ay->host;
        bool aliased = false;
        for (int y = 0; y < d0.height && !aliased; ++y)
            for (int j = 0; j < ty.count[y]; ++j)
                if (ty.eff[(size_t)y * ty.taps + j] != ty.left[y] + j) { aliased = true; break; }
        if (up) ring_rows = 4;
        else if (ty.ref_ring == 8 || ty.ref_ring == 16) ring_rows = ty.ref_ring;
        else RUN_GROUP_NA;
        if (!(ring_rows == ty.ref_ring || (!aliased && ty.span <= ring_rows)) || ty.taps > ring_rows + 1) RUN_GROUP_NA;
        for (int y = 1; y < d0.height; ++y)
            if (ty.need[y] < ty.need[y - 1]) RUN_GROUP_NA;   // rows are emitted in order
        rc = get_yrange(ctx, ay, &up_range);
        if (rc) return rc;
    }
    const bool ring_mode = ring_rows > 0;
    // window offsets inside a period rounded down or up (7.5x, 30 taps: 31 with the offsets 0 / 7, 30 with 0 / 8): the form with the
    // shorter window first; outputs per lane: four for single-channel pixels, two for the others (r03 survey,
    // profiles/r03_survey_run_kernel.txt)
    int tx_of[2], xb_unused = 0;
    for (int c = 0; c < 2; ++c) tx_of[c] = run_table_taps(ax->host, period, period_px, c, &xb_unused);
    if (tx_of[0] <= 0) RUN_GROUP_NA;
    static const int order1[] = {4, 8, 2, 0, 0, 0}, order2[] = {2, 4, 8, 0, 0, 0};   // (0: no candidate)
    // upscales (r03, profiles/r03_ab_up_kernel.txt): the longest run that is instantiated
    // (3x upscales were instantiated in r05 — runs of 3 / 6 / 12 — and the run table declines them: the reference accumulates its tap centres in binary32
    //  (src/resize.cc:23-26, center += scale) and 1/3 is not a binary fraction, so beyond narrow rows output x + 3 does NOT carry output x's weights bit for
    //  bit; only the factors 2 and 4 are periodic.  3x stays on the lane-resident-weight kernels: 0.32-0.40 of the roof at 64 images, profiles/r05_ab_up3.txt)
    static const int order_up[] = {8, 4, 2, 0, 0, 0};
    // neighbour-exchange form (r04): dword-sized pixels at integer scales; the canonical window starts -x_base pixels before the
    // lane's own; the clipped columns are recomputed from the image (no stash)
    int xhl = -1;
    if (!up && !ring_mode && opt_run_xch != 2 && kBytes[pixel] % 4 == 0 && period == 1) {
        int xb = 0;
        const int tx = run_table_taps(ax->host, period, period_px, 0, &xb);
        for (int pl0 : order2) {
            const int pl = opt_per_lane > 0 ? (int)opt_per_lane : pl0;
            if (tx <= 0 || xb > 0 || !run_kernel_supported(pixel, period, period_px, tx, pl, nacc, 0, -xb, 0)) continue;
            if (opt_run_xch == 0 && opt_run != 1 && !run_xch_preferred(pixel, period_px, tx, pl, fma)) continue;
            per_lane = pl; ktaps = tx; ceil_off = 0; xhl = -xb;
            break;
        }
    }
    for (int k = 0; k < 2 && !per_lane; ++k) {
        const int c = (tx_of[1] > 0 && tx_of[1] < tx_of[0]) ? 1 - k : k;   // the shorter window first
        if (tx_of[c] <= 0 || (period == 1 && c == 1) || (opt_run_round != 0 && opt_run_round != c + 1)) continue;
        for (int tx = tx_of[c]; tx <= tx_of[c] + 3 && !per_lane; ++tx)
            for (int pl0 : (up || (period == 2 && period_px == 3) ? order_up : kChannels[pixel] == 1 ? order1 : order2)) {
                const int pl = opt_per_lane > 0 ? (int)opt_per_lane : pl0;   // "per_lane" names the run length outright
                // (r05) 4x upscales of the three- and four-channel formats: runs of four beat runs of eight
                // (rgba -15 %, fma -27 %; rgb -5 % / -12 %; one- and two-channel formats +15 .. +43 %: profiles/r05_ab_up_survey.txt) and the accumulator form (r16g16b16a16 -13 %; its fma flavour stays there: +59 %) — profiles/r05_ab_up4.txt
                if (up && !cv && opt_per_lane <= 0 && period == 4 && pl0 == 8 && kChannels[pixel] >= 3 && up_kernel_supported(pixel, period, period_px, tx, 4, c, ring_rows, ay->host.taps)) continue;
                if (up && !cv && opt_run != 1 && period == 4 && pixel == PICHA_R16G16B16A16 && fma) continue;
                if (pl <= 0 || pl % period || !(cv ? up_cv_kernel_supported(pixel, spixel, period, period_px, tx, pl, c, ring_rows, ay->host.taps) : ring_mode ? up_kernel_supported(pixel, period, period_px, tx, pl, c, ring_rows, ay->host.taps) : (run_kernel_supported(pixel, period, period_px, tx, pl, nacc, c) || (opt_run_ws != 2 && (run_kernel_supported(pixel, period, period_px, tx, pl, nacc, c, -1, 1) || run_kernel_supported(pixel, period, period_px, tx, pl, nacc, c, -2, std::min(2, period))))))) continue;
                per_lane = pl; ktaps = tx; ceil_off = c;
                break;
            }
    }
    if (!per_lane) RUN_GROUP_NA;
    // where it is the default (same survey): windows of 26 taps and more (cfg3b: -3 % reference arithmetic, -10 % fma against the
    // row-staged kernel; greya -35 %); shorter windows only for batches that still fill the chip with a quarter of the
    // accumulator kernel's waves (256 x 4K -> 512x288 grey: -11 %; 128 images: a tie; integer scales 6x / 8x: -9 .. -13 % at 128 images)
    // ... and the other instantiated shapes of one- and two-channel pixels (1.5x: grey -31 %, greya -9 %; 2.5x / 3x / 4x / 5x: -11 .. -26 %,
    // measured on 128-256 images); the 21-tap 7.5x shape was a tie for greya and needs >= 192 images for grey
    const int ch = kChannels[pixel];
    // (r05) rgb exists in the shared-unpack form only (7.5x / 6x / 8x with picha's defaults: the jpeg thumbnail -10 % / fma -9 %, 4K -> 640x360 -13 % / -9 %,
    // -> 480x270 -13 % / -10 %, 8K -> 1024x576 -11 % against the accumulator kernel; 32 4K images -14 %, 16 of them or 64 x 1080p -> 256x144 +7 %: launches of
    // >= 200 000 wave-rows; profiles/r05_ab_shu_rgb.txt, r05_ab_shu_rgb_defaults.txt)
    const bool shu_only = !ring_mode && xhl < 0 && !run_kernel_supported(pixel, period, period_px, ktaps, per_lane, nacc, ceil_off) && !run_kernel_supported(pixel, period, period_px, ktaps, per_lane, nacc, ceil_off, -1, 1);
    const long long wave_cols = (long long)n * (((d0.width + per_lane - 1) / per_lane + 63) / 64);
    // (r05: greya at 7.5x / 21 taps, a tie with the accumulator kernel in its plain form, wins in the shared-unpack form: 128 x 4K -> 512x288 0.579 -> 0.529 ms)
    const bool greya_shu21 = ch == 2 && period_px == 15 && !fma && !ring_mode && opt_run_shu != 2 && opt_run_ws != 2 && opt_run_edges != 1 && n >= 96 &&
                             run_kernel_supported(pixel, period, period_px, ktaps, per_lane, nacc, ceil_off, -2, std::min(2, period));
    const bool short_ok = shu_only ? wave_cols * s0.height >= 200000 : period_px == 15 ? ((ch == 1 && n >= 192) || greya_shu21) : (ch <= 2 && (n >= 96 || wave_cols * s0.height >= 400000));
    // (r05, third session: the count of IMAGES says little about 4K sources — 64 x 4K grey -> 2560x1440 -27 %, -> 1536x864 -28 %, -> 1280x720 -6.5 %, greya 1.5x -1 % against the
    // accumulator kernel the n >= 96 rule left them on; >= 400 000 wave-rows as well: profiles/r05_ab_planner_declines.txt)
    if (!ring_mode && xhl < 0 && opt_run == 0 && ktaps < 26 && !short_ok) RUN_GROUP_NA;
    if (ylist && opt_run == 0 && wave_cols * s0.height < 200000) RUN_GROUP_NA;   // (the row-list shapes were measured on batches: lanczos thumbnails of 64-128 4K images)
    // one phase's weights in SGPRs (r04): where instantiated — the 7.5x shapes, whose 2 x 30 lane-uniform weight registers held the
    // kernel at 3 waves per SIMD (profiles/r04_ab_run_ws.txt)
    const int wsgpr = !ring_mode && xhl < 0 && opt_run_ws != 2 && run_kernel_supported(pixel, period, period_px, ktaps, per_lane, nacc, ceil_off, -1, 1) ? 1 : 0;
    const DeviceAxis::RunTable *rt = nullptr;
    rc = get_runtable(ctx, ax, period, period_px, ceil_off, ktaps, wmode, is16, &rt);
    if (rc) return rc;
    if (!rt->ok) RUN_GROUP_NA;

    // edge columns: recomputed by resize_edges_kernel from a stash of the raw dwords of the two lanes that hold them (kernels_run.hip)
    const int x_groups = (d0.width + per_lane - 1) / per_lane;
    const bool has_edges = rt->c0 > 0 || rt->c1 < d0.width;
    if (xhl >= 0 && rt->x_base != -xhl) RUN_GROUP_NA;
    int stash_words = 0, stash_org[2] = {0, 0};
    // shared unpack (r05, kernels_run.hip RunShu): a lane converts only the pixels its window starts with and takes the rest from lane + 1 through LDS; no stash;
    // both phases' weights in SGPRs, two row buffers, five waves per SIMD.  The default in the reference flavour wherever instantiated (grey 7.5x): cfg3b 0.678 -> 0.652 ms
    // (-3.8 % against the r03 form), 21 taps -2.9 %; the fma flavour sits at its memory pattern (+-0.5 %) and keeps the r03 form (profiles/r05_ab_shu_final.txt)
    const bool shu_ok = !ring_mode && xhl < 0 && !cv && opt_run_shu != 2 && opt_run_ws != 2 && opt_run_edges != 1 &&
                        run_kernel_supported(pixel, period, period_px, ktaps, per_lane, nacc, ceil_off, -2, std::min(2, period)) && (opt_run_shu == 1 || !fma || shu_only);
    if (shu_only && !shu_ok) RUN_GROUP_NA;
    if (ylist && !shu_ok) RUN_GROUP_NA;
    const int32_t *edge_range = nullptr;   // the edge kernel's source-row ranges are ROWS, not list positions
    if (ylist) { rc = get_yrange(ctx, ay, &edge_range); if (rc) return rc; }
    // (r05) the upscale shapes of the ring kernel produce the clipped columns themselves from those columns' own weights, zero-padded into the canonical window
    // (kernels_run.hip EIN): no stash, no edge kernel — where every clipped column's taps lie inside its canonical window
    bool edges_inside = has_edges && (opt_run_edges == 0 || opt_run_edges == 3) && opt_run != 3 &&
                        (ring_mode ? up_edges_inside(period, period_px, ktaps, per_lane) : (!cv && xhl >= 0 && wsgpr == 0 && run_xch_edges_inside(ktaps, per_lane)));
    if (edges_inside) {
        const AxisTaps &tx = ax->host;
        auto offg = [&](int x) { return (x / period) * period_px + ((x % period) * period_px + ceil_off * (period - 1)) / period; };
        for (int x = 0; x < d0.width && edges_inside; ++x) {
            if (x >= rt->c0 && x < rt->c1) continue;
            const int first = tx.left[x] - (rt->x_base + offg(x));
            if (first < 0 || first + tx.count[x] > ktaps) edges_inside = false;
        }
    }
    const bool use_stash = !edges_inside && has_edges && xhl < 0 && !cv && (opt_run_edges == 1 || (opt_run_edges == 0 && pixel == PICHA_GREY && !ring_mode && !shu_ok));
    if (has_edges && xhl < 0 && !cv && !edges_inside) {   // (the stash's applicability rules also hold where the columns now come from the image: the planner's choices are the measured ones)
        const int bpp = kBytes[pixel];
        const int span = ((per_lane - 1) / period) * period_px + (((per_lane - 1) % period) * period_px + ceil_off * (period - 1)) / period + ktaps;  // RunCfg::SPAN
        stash_words = bpp % 4 == 0 ? span * bpp / 4 : (span * bpp + 3) / 4 + 1;                                        // RunCfg::NW
        if (x_groups < 2) RUN_GROUP_NA;
        for (int i = 0; i < n; ++i)
            if ((uintptr_t)src[i].data % 4) RUN_GROUP_NA;  // the stash is laid out for lane windows that start on a dword of the row
        const int g_last = (d0.width - 1) / per_lane;
        for (int side = 0; side < 2; ++side) {
            const int g = side == 0 ? 0 : g_last;
            const int ws = rt->x_base + g * (per_lane / period) * period_px;   // first pixel of the lane's window
            const int org = (ws * bpp) & ~3;
            stash_org[side] = org;
            const int xa = side == 0 ? 0 : rt->c1, xb = side == 0 ? rt->c0 : d0.width;
            if (side == 0 ? rt->c0 > per_lane : rt->c1 < g_last * per_lane) RUN_GROUP_NA;   // the side's columns all belong to that lane
            for (int x = xa; x < xb; ++x) {
                const int bo = ax->host.left[x] * bpp - org;
                if (bo < 0 || bo + ax->host.count[x] * bpp > stash_words * 4) RUN_GROUP_NA;
            }
        }
    }
    int edge_rows = 0;
    std::vector<int> edge_first, edge_last;   // per output row: first source row needed by it or any later row / last row needed so far (the march's range table)
    int edge_fixed_floats = 0, edge_col_floats = 0, edge_ytaps = 0, edge_lds_floats = 0;
    {   // blocks of the edge pass (kernels_run.hip, edge_pass): x weights + y weights + horizontal sums <= 2048 floats (kEdgeFloats)
        const int ncol = std::max(1, std::max(rt->c0, d0.width - rt->c1));   // one side per wave
        const int txe = (ktaps + 3) / 4 * 4;
        const int ch = kChannels[pixel];
        // source rows a block of `rows` output rows spans at most, exactly as the kernel derives them from the march's range table
        // (first row needed by the block's first output row or any later one .. last row needed so far)
        const AxisTaps &ty = ay->host;
        std::vector<int> first(d0.height), last(d0.height);
        for (int y = 0; y < d0.height; ++y) { first[y] = ty.left[y]; last[y] = std::max(ty.need[y], ty.left[y] + ty.count[y] - 1); }
        for (int y = d0.height - 2; y >= 0; --y) first[y] = std::min(first[y], first[y + 1]);
        for (int y = 1; y < d0.height; ++y) last[y] = std::max(last[y], last[y - 1]);
        auto max_rows = [&](int rows) {
            int m = 0;
            for (int ya = 0; ya < d0.height; ya += rows) m = std::max(m, last[std::min(d0.height, ya + rows) - 1] - first[ya] + 1);
            return m;
        };
        int er = std::min(g_opt_edge_rows >= 1 && g_opt_edge_rows <= 64 ? (int)g_opt_edge_rows : 64, d0.height);
        auto floats = [&](int rows) { return ncol * txe + rows * ty.taps + ncol * ch * max_rows(rows); };
        while (er > 1 && floats(er) > 2048) --er;
        if (floats(er) > 2048) RUN_GROUP_NA;
        edge_rows = er;
        edge_lds_floats = floats(er);
        edge_first = first; edge_last = last; edge_fixed_floats = ncol * txe; edge_col_floats = ncol * ch; edge_ytaps = ty.taps;
    }
    RunParams p{};
    long long sp = 0, dp = 0;
    TablePair tab;
    std::unique_lock<std::mutex> temp_lock(ctx->temp_mu, std::defer_lock);
    if (uniform_pitch(src, n, &sp) && uniform_pitch(dst, n, &dp)) {
        p.src = static_cast<const uint8_t *>(s0.data);
        p.dst = static_cast<uint8_t *>(d0.data);
        p.src_pitch = sp; p.dst_pitch = dp;
    } else {
        temp_lock.lock();
        rc = make_tables(ctx, src, dst, n, s, &tab);
        if (rc) return rc;
        p.src_tab = tab.src; p.dst_tab = tab.dst;
    }
    bool stash_used = false;
    auto release_temps = [&]() {
        int r1 = PICHA_OK, r2 = PICHA_OK;
        if (tab.used) r1 = temp_release(ctx->tables, s);
        if (stash_used) r2 = temp_release(ctx->scratch, s);
        return r1 ? r1 : r2;
    };
    if (has_edges && opt_run != 3 && !use_stash && !edges_inside) p.edge_image = 1;
    if (edges_inside) p.edge_in = 1;
    if (cv) { p.cv_on = 1; p.cv_src_pixel = spixel; p.cv_r = cv->r; p.cv_g = cv->g; p.cv_b = cv->b; }
    if (use_stash && opt_run != 3) {
        const size_t bytes = (size_t)n * 2 * s0.height * stash_words * 4;
        if (bytes >= ((size_t)1 << 32)) { (void)release_temps(); RUN_GROUP_NA; }
        if (!temp_lock.owns_lock()) temp_lock.lock();
        rc = temp_acquire(ctx->scratch, bytes, 0, s);
        if (rc) { (void)release_temps(); return rc; }
        stash_used = true;
        p.stash = static_cast<uint32_t *>(ctx->scratch.dev);
        p.stash_words = stash_words; p.stash_org[0] = stash_org[0]; p.stash_org[1] = stash_org[1];
    }
    p.src_stride = s0.stride; p.dst_stride = d0.stride;
    p.src_w = s0.width; p.src_h = s0.height; p.dst_w = d0.width; p.dst_h = d0.height;
    p.src_bytes = (unsigned)src_span; p.dst_bytes = (unsigned)dst_span;
    p.period = period; p.period_px = period_px; p.x_taps = ktaps; p.x_per_lane = per_lane; p.x_base = rt->x_base; p.off_ceil = ceil_off;
    p.x_groups = x_groups;
    p.x_canon = rt->canon;
    p.fma = fma;
    const bool shu = shu_ok;
    p.xch = xhl >= 0 ? 1 : shu ? 2 : 0; p.wsgpr = shu ? std::min(2, period) : wsgpr;   // (the shared-unpack shapes keep both phases' weights in SGPRs)
    // time-phased stores: where the stores are a thin trickle beside the reads (>= 3 source rows per output row) and the launch is long
    // enough for several phases; 2^12 ticks = 41 us measured best on cfg5 (profiles/r04_ab_run_phase.txt)
    if (xhl >= 0 && opt_run_phase >= 0) p.phase_shift = opt_run_phase > 0 ? (int)std::min(20L, opt_run_phase) : (s0.height >= 3 * d0.height && (long long)n * s0.height >= 20000 ? 12 : 0);
    if (ring_mode) { p.yrow_range = up_range; p.y_left = ay->left; p.y_info = ay->y_info; p.ring_rows = ring_rows; }
    else { p.yrow_rec = yr->rec; p.yrow_range = yr->range; p.yrow_src = ylist ? yr->src : nullptr; p.nacc = nacc; }
    p.cx0 = rt->c0; p.cx1 = rt->c1;
    if (opt_run == 3) { p.cx0 = 0; p.cx1 = d0.width; }  // measurement only: no edge pass (the edge columns come out wrong)
    p.edge_rows = edge_rows; p.edge_lds_floats = edge_lds_floats;
    p.x_left = ax->left; p.x_count = ax->count; p.x_weight = ax->weight; p.x_wtaps = ax->host.taps;
    p.y_count = ay->count; p.y_eff = ay->eff; p.y_weight = ay->weight; p.y_taps = ay->host.taps;
    const int col_waves = (p.x_groups + 63) / 64;
    p.block_waves = opt_bw >= 1 && opt_bw <= 4 ? (int)opt_bw : std::min(4, col_waves);
    // (r05, the rgb shared-unpack shapes: 10 KB of exchange slots per wave, so an idle wave of a ragged workgroup costs a resident one: five wave columns run as
    //  2 + 2 + 1 rather than 4 + 1 — 4K -> 640x360: -13 % against +1 %)
    if (!(opt_bw >= 1 && opt_bw <= 4) && shu_only && col_waves > 4 && col_waves % 2 < (4 - col_waves % 4) % 4) p.block_waves = 2;
    const int nsb = (col_waves + p.block_waves - 1) / p.block_waves;
    // row chunks: as for the accumulator kernel (>= ~8 K waves, the re-resampled halo of a chunk under ~6 % of its rows)
    const int ring = std::max(ay->host.span, 1);
    long long chunks = opt_chunks > 0 ? opt_chunks : (32768 + (long long)n * col_waves - 1) / ((long long)n * col_waves);
    if (opt_chunks <= 0) {
        const double src_rows_per_out = std::max(1.0, (double)s0.height / d0.height);
        const long long min_rows = (long long)std::ceil(ring / (0.06 * src_rows_per_out));
        const long long by_halo = std::max<long long>(1, d0.height / std::max<long long>(16, min_rows));
        const long long by_fill = (8192 + (long long)n * col_waves - 1) / ((long long)n * col_waves);
        chunks = std::min(chunks, std::max(by_halo, by_fill));
    }
    chunks = std::max<long long>(1, std::min<long long>(chunks, std::max(1, opt_chunks > 0 ? d0.height : d0.height / 16)));
    if (opt_chunks <= 0 && xhl >= 0 && p.phase_shift > 0) {
        // the neighbour-exchange form with phased stores is bound by its memory pattern (LAB.md 5.0c): every halo row it re-reads is
        // time.  Longer chunks — halo under ~3 % — as long as the launch stays at least two rounds of resident workgroups
        // (cfg5, 32 images: 18 -> 12 chunks 1.922 -> 1.892 ms, fma 1.781 -> 1.732; 9 chunks = 1.4 rounds 2.012; profiles/r04_ab_run_xch.txt)
        const double src_rows_per_out = std::max(1.0, (double)s0.height / d0.height);
        const long long by_halo3 = std::max<long long>(1, d0.height / std::max<long long>(16, (long long)std::ceil(ring / (0.03 * src_rows_per_out))));
        const int resident = run_kernel_resident_blocks(pixel, p);
        const long long two_rounds = resident > 0 ? (2LL * resident + (long long)nsb * n - 1) / ((long long)nsb * n) : chunks;
        chunks = std::max<long long>(1, std::min(chunks, std::max(by_halo3, two_rounds)));
    }
    if (opt_chunks <= 0 && shu) {
        // the shared-unpack form: one whole round of resident workgroups where the halo rule allows it (cfg3b: 5 waves per SIMD = 2560 two-wave workgroups = 10
        // chunks of 256 images: -2 % against 8, +1.5 % at 12; profiles/r05_ab_shu_occupancy.txt)
        const int resident = run_kernel_resident_blocks(pixel, p);
        const long long round = resident > 0 ? resident / ((long long)nsb * n) : 0;
        if (shu_only ? (round >= 2 && round <= chunks) : (round >= chunks && round <= 2 * chunks)) chunks = round;   // (rgb, LDS-bound residency: the jpeg thumbnail 11 -> 4 chunks -5 %)
    }
    // small launches (a single image): chunks of down to one row for about 3000 waves, as for the accumulator kernel
    if (opt_chunks <= 0 && (long long)n * col_waves * chunks < 3000)
        chunks = std::max(chunks, std::min<long long>(d0.height, (3000 + (long long)n * col_waves - 1) / ((long long)n * col_waves)));
    p.rows_per_chunk = (int)((d0.height + chunks - 1) / chunks);
    p.n_chunks = (d0.height + p.rows_per_chunk - 1) / p.rows_per_chunk;
    auto fail_launch = [&](hipError_t e) {
        (void)release_temps();
        return fail(PICHA_ERR_HIP, "run-kernel resize launch failed: %s", hipGetErrorString(e));
    };
    auto slice = [&](int done, int m) {   // images [done, done + m) of the group (grid.z limit of the plain grids)
        RunParams q = p;
        q.n_images = m;
        if (p.src_tab) { q.src_tab = p.src_tab + done; q.dst_tab = p.dst_tab + done; }
        else { q.src = p.src + (long long)done * sp; q.dst = p.dst + (long long)done * dp; }
        q.stash_img0 = done;
        return q;
    };
    auto edges_all = [&](hipStream_t es) {
        hipError_t e = hipSuccess;
        for (int done = 0; done < n && e == hipSuccess; done += 65535) {
            RunParams q = slice(done, std::min(n - done, 65535));
            if (ylist) q.yrow_range = edge_range;
            e = launch_resize_edges_kernel(pixel, q, es);
        }
        return e;
    };
    // launch form: the plain grid (measured r03 on cfg3b: item list +3 %, persistent +30 %; the edge kernel follows either way)
    const int work_mode = !ring_mode && (opt_work == 1 || opt_work == 2) ? (int)opt_work : 0;
    // (r05, third session) cfg3b's shape carries its edge blocks in the march's own grid (kernels_run.hip, run_edges_fused): no second kernel
    if (!work_mode && !ring_mode && shu && p.edge_image && !ylist && !cv && opt_run != 3 && kRunEfuseBuilt && PICHA_EFUSE_DEFAULT(opt_run_efuse) && pixel == PICHA_GREY && period == 2 && period_px == 15 &&
        ktaps == 30 && per_lane == 4 && p.block_waves >= 2 && p.rows_per_chunk >= 1) {
        // the edge blocks start at the march's chunk boundaries, not at multiples of edge_rows: size their sub-blocks for THOSE starts
        auto fits = [&](int er) {
            for (int y0 = 0; y0 < d0.height; y0 += p.rows_per_chunk)
                for (int ya = y0, y1 = std::min(d0.height, y0 + p.rows_per_chunk); ya < y1; ya += er) {
                    const int yb = std::min(y1, ya + er);
                    if (edge_fixed_floats + (yb - ya) * edge_ytaps + edge_col_floats * (edge_last[yb - 1] - edge_first[ya] + 1) > 2048) return false;
                }
            return true;
        };
        int er = edge_rows;
        while (er > 1 && !fits(er)) --er;
        if (fits(er)) { p.edge_fused = 1; p.edge_rows = er; }
    }
    if (work_mode) {
        std::shared_ptr<const DeviceCtx::ItemList> il;
            std::shared_ptr<const DeviceCtx::Queue> qhold;
        rc = get_items(ctx, n, d0.height, nsb, p.rows_per_chunk, std::max(16, p.rows_per_chunk / (int)opt_work_tail_div), (int)opt_work_tail, &il);
        if (rc) { (void)release_temps(); return rc; }
        p.n_images = n;
        p.work.items = il->items; p.work.items_per_xcd = il->per_xcd;
        if (work_mode == 2) {
            const int resident = run_kernel_resident_blocks(pixel, p);
            if (resident > 0 && (long long)il->per_xcd * 8 > 2LL * resident) {
                rc = get_queue(ctx, s, &p.work.queue, &qhold);
                if (rc) { (void)release_temps(); return rc; }
                p.n_chunks = std::max(8, resident / 8 * 8);
            }
        }
        hipError_t e = launch_resize_run_kernel(pixel, p, s, &g_last_kernel);
        if (e == hipSuccess) e = edges_all(s);
        if (e != hipSuccess) return fail_launch(e);
    } else {
        hipError_t e = hipSuccess;
        int done = 0;
        while (done < n) {  // grid.z limit
            const int m = std::min(n - done, 65535);
            const RunParams q = slice(done, m);
            e = ring_mode ? launch_resize_up_kernel(pixel, q, s, &g_last_kernel) : launch_resize_run_kernel(pixel, q, s, &g_last_kernel);
            if (e == hipSuccess && !p.edge_fused) {
                RunParams qe = q;
                if (ylist) qe.yrow_range = edge_range;
                e = launch_resize_edges_kernel(pixel, qe, s);
            }
            if (e != hipSuccess) return fail_launch(e);
            done += m;
        }
    }
    if ((p.stash || p.edge_image) && !p.edge_fused) {
        static thread_local char name[192];
        snprintf(name, sizeof name, "%s + resize_edges<%d columns>", g_last_kernel ? g_last_kernel : "", p.cx0 + (d0.width - p.cx1));
        g_last_kernel = name;
    }
    *used = true;
    return release_temps();
}

// where the one-kernel form of colorConvert -> integer upscale is the default (the instantiated pairs: kernels_run.hip, PICHA_UPCV_LIST; measured on
// 256 x 960x540 / 512 x 480x270 -> 1080p, profiles/r04_ab_convert_up.txt): at 2x every pair, both flavours (-9 .. -29 % against convert + resize_up);
// at 4x the launch is bound by its stores and the gain is smaller: -4 .. -18 % in the reference flavour except grey -> rgb (+-4 %), in the fma flavour
// only the pairs below (grey -> rgb runs 40 % SLOWER there, rgba -> rgb and rgb -> rgba are ties).  "fuse" = 1 takes the one-kernel form wherever it exists.
bool convert_up_preferred(int src_pixel, int dst_pixel, int factor, bool fma) {
    if (factor <= 2) return true;
    if (src_pixel == PICHA_GREY && dst_pixel == PICHA_RGB) return false;
    if (!fma) return true;
    return dst_pixel == PICHA_GREY || (src_pixel == PICHA_GREYA && dst_pixel == PICHA_RGBA);
}

// Row-staged kernel for one equal-geometry group (kernels_rows.hip): plain downscale (src.pixel == dst.pixel) or colour
// convert fused in front (src.pixel != dst.pixel: the result equals colorConvert followed by resize, the intermediate
// quantised in registers).  *used = false: not applicable, the caller goes on with the other kernels.
int rows_group(DeviceCtx *ctx, const picha_image *src, const picha_image *dst, int n, int filter, float fscale, float r, float g,
               float b, hipStream_t s, bool *used) {
    *used = false;
    const picha_image &s0 = src[0], &d0 = dst[0];
    const int variant = rows_variant(s0.pixel, d0.pixel);
    long opt_rows, opt_numerics, opt_chunks, opt_per_lane;
    {
        std::lock_guard<std::mutex> lk(g_opt_mu);
        opt_rows = g_opt_rows; opt_numerics = tl_numerics >= 0 ? tl_numerics : g_opt_numerics; opt_chunks = g_opt_row_chunks; opt_per_lane = g_opt_per_lane;
    }
    if (variant < 0 || opt_rows == 2) return PICHA_OK;
    if (d0.width >= s0.width || d0.height >= s0.height) return PICHA_OK;  // downscales only
    const size_t src_span = image_span(s0), dst_span = image_span(d0);
    if (src_span >= (1u << 31) || dst_span >= (1u << 31)) return PICHA_OK;
    const int dal = fused_align(d0.pixel);
    for (int i = 0; i < n; ++i) {
        if (((uintptr_t)src[i].data | (uintptr_t)src[i].stride) % 4) return PICHA_OK;  // dword loads of whole stage pieces
        if (((uintptr_t)dst[i].data | (uintptr_t)dst[i].stride) % dal) return PICHA_OK;
    }
    std::shared_ptr<DeviceAxis> ax_ref, ay_ref;
    int rc = get_axis(ctx, filter, fscale, s0.width, d0.width, &ax_ref);
    if (rc) return rc;
    rc = get_axis(ctx, filter, fscale, s0.height, d0.height, &ay_ref);
    if (rc) return rc;
    DeviceAxis *ax = ax_ref.get(), *ay = ay_ref.get();
    const int taps = ax->host.taps;
    // where it pays (r02 survey): windows long enough that the per-lane conversions of the accumulator kernel dominate,
    // or the fused colour convert (whose conversion chain must not be repeated per tap)
    const bool conv = s0.pixel != d0.pixel;
    // plain downscales (r02, profiles/r02_ab_rows_*.txt): the kernel is LDS-bound — single-channel formats with two outputs per
    // thread and >= 26 taps gain 11-13 % over the accumulator kernel (cfg3b), shorter windows and 4-channel pixels lose
    const bool plain_auto = !conv && opt_rows == 0;
    if (plain_auto && (taps < 26 || kChannels[d0.pixel] > 2)) return PICHA_OK;
    const int fma = opt_numerics == 1 ? 1 : 0;
    const int is16 = d0.pixel >= PICHA_R16;
    const int align = rows_window_align(variant);
    const int piece_px = rows_stage_pixels(variant);
    // shape: two adjacent outputs per thread where the table has a uniform stride (window of the second output S pixels
    // after the first, zero-padded taps), else one
    const DeviceAxis::YRows *yr = nullptr;
    int nacc = 0;
    for (int na : {4, 6}) {
        const DeviceAxis::YRows *cand = nullptr;
        rc = get_yrows(ctx, ay, na, &cand);
        if (rc) return rc;
        if (cand->ok) { yr = cand; nacc = na; break; }
    }
    if (!yr) return PICHA_OK;
    // candidates, best first: two adjacent outputs per thread at the table's integer step with the smallest tap bucket
    // that holds it, then one output per thread; the first whose table, strips and LDS footprint work is taken
    struct Shape { int per_lane, step, taps; };
    std::vector<Shape> shapes;
    const int st = (int)std::floor((double)s0.width / d0.width);
    static const int p2[][2] = {{3, 12}, {3, 16}, {4, 16}, {4, 20}, {5, 20}, {5, 24}, {6, 24}, {6, 28}, {7, 24}, {7, 32}};
    if (opt_per_lane != 1)
        for (auto &e : p2)
            if (e[0] == st) shapes.push_back({2, e[0], e[1]});
    if (opt_per_lane != 2 && !plain_auto)   // (one output per thread does not beat the accumulator kernel on plain downscales)
        for (int tb : {16, 24, 32, 40})
            if (taps + align - 1 <= tb) { shapes.push_back({1, 1, tb}); break; }
    const int chl = kChannels[d0.pixel] == 3 ? 4 : kChannels[d0.pixel];
    const DeviceAxis::XTable *xt = nullptr;
    const DeviceAxis::RowsStrips *rs = nullptr;
    int per_lane = 0, step = 1, ktaps = 0;
    for (const Shape &sh : shapes) {
        if (!rows_kernel_supported(variant, sh.per_lane, sh.step, sh.taps)) continue;
        const DeviceAxis::XTable *cx = nullptr;
        rc = get_xtable(ctx, ax, sh.per_lane, sh.step, sh.taps, fma, is16, &cx, 0, false, align);
        if (rc) return rc;
        if (!cx->ok) continue;
        const DeviceAxis::RowsStrips *cs = nullptr;
        rc = get_rows_strips(ctx, ax, cx, sh.step * (sh.per_lane - 1) + sh.taps, piece_px, &cs);
        if (rc) return rc;
        if (!cs->ok) continue;
        if ((size_t)2 * cs->pieces * 256 * piece_px * chl * 4 + 32 > 96 * 1024) continue;  // LDS: keep >= 1 workgroup per SIMD pair
        xt = cx; rs = cs; per_lane = sh.per_lane; step = sh.step; ktaps = sh.taps;
        break;
    }
    if (!xt) return PICHA_OK;

    RowsParams p{};
    long long sp = 0, dp = 0;
    TablePair tab;
    std::unique_lock<std::mutex> temp_lock(ctx->temp_mu, std::defer_lock);
    if (uniform_pitch(src, n, &sp) && uniform_pitch(dst, n, &dp)) {
        p.src = static_cast<const uint8_t *>(s0.data);
        p.dst = static_cast<uint8_t *>(d0.data);
        p.src_pitch = sp; p.dst_pitch = dp;
    } else {
        temp_lock.lock();
        rc = make_tables(ctx, src, dst, n, s, &tab);
        if (rc) return rc;
        p.src_tab = tab.src; p.dst_tab = tab.dst;
    }
    p.src_stride = s0.stride; p.dst_stride = d0.stride;
    p.src_w = s0.width; p.src_h = s0.height; p.dst_w = d0.width; p.dst_h = d0.height;
    p.src_bytes = (unsigned)src_span; p.dst_bytes = (unsigned)dst_span;
    p.x_left = xt->left; p.x_weight = xt->weight; p.strip_left = rs->strip_left;
    p.x_taps = ktaps; p.x_per_lane = per_lane; p.x_step = step; p.x_groups = xt->groups; p.xw_stride = xt->stride;
    p.n_strips = rs->n_strips; p.stage_pieces = rs->pieces;
    p.fma = fma;
    p.r = r; p.g = g; p.b = b;
    p.yrow_rec = yr->rec; p.yrow_range = yr->range; p.nacc = nacc;
    // row chunks: >= ~4 workgroups per CU in the launch, but every chunk re-stages the `ring` source rows above its first
    // output: keep that halo under ~8 % of the chunk's own source rows
    const int ring = std::max(ay->host.span, 1);
    const long long blocks_per_chunk = (long long)n * rs->n_strips;
    long long chunks = opt_chunks > 0 ? opt_chunks : (1024 + blocks_per_chunk - 1) / blocks_per_chunk;
    if (opt_chunks <= 0) {
        const double src_rows_per_out = std::max(1.0, (double)s0.height / d0.height);
        const long long min_rows = (long long)std::ceil(ring / (0.08 * src_rows_per_out));
        chunks = std::min(chunks, std::max<long long>(1, d0.height / std::max<long long>(8, min_rows)));
    }
    // small launches (a single image is one or two workgroups per chunk): the halo rule above would leave a handful of
    // workgroups on 256 CUs; chunks of down to one row for about three workgroups per CU (r03, profiles/r03_ab_small_launches.txt)
    if (opt_chunks <= 0 && blocks_per_chunk * chunks < 768) chunks = std::max(chunks, (768 + blocks_per_chunk - 1) / blocks_per_chunk);
    chunks = std::max<long long>(1, std::min<long long>(chunks, d0.height));
    p.rows_per_chunk = (int)((d0.height + chunks - 1) / chunks);
    p.n_chunks = (d0.height + p.rows_per_chunk - 1) / p.rows_per_chunk;
    int done = 0;
    while (done < n) {
        const int m = std::min(n - done, 65535);
        RowsParams q = p;
        q.n_images = m;
        if (p.src_tab) { q.src_tab = p.src_tab + done; q.dst_tab = p.dst_tab + done; }
        else { q.src = p.src + (long long)done * sp; q.dst = p.dst + (long long)done * dp; }
        const hipError_t e = launch_resize_rows_kernel(variant, q, s, &g_last_kernel);
        if (e != hipSuccess) {
            if (tab.used) (void)temp_release(ctx->tables, s);
            return fail(PICHA_ERR_HIP, "row-staged resize launch failed: %s", hipGetErrorString(e));
        }
        done += m;
    }
    *used = true;
    if (tab.used) return temp_release(ctx->tables, s);
    return PICHA_OK;
}

// ---- resize of one equal-geometry group on device ------------------------------------------------
// epi != nullptr: resize followed by a colour convert to dst[].pixel in ONE kernel (accumulator kernel with the convert as a
// store epilogue); *epi_used says whether that was possible — if not, nothing was launched and the caller chains two kernels.
struct ChainEpilogue { float r, g, b; bool force; };
// where the epilogue form is the default (measured r03, profiles/r03_ab_epilogue.txt, 4K sources): 16-bit sources (rgba16 2x down
// -> rgba: -21 % against resize_sym + convert) and rgba at 2x / 3x (-1 .. -3 %: a tie in time, one intermediate image less in
// HBM); rgb sources lose 10 % (their two-kernel chain runs the four-outputs-per-lane streaming kernel) and at 4x it is a tie,
// so those keep the chain.  "fuse" = 1 takes the epilogue wherever it exists.
// r04 (profiles/r04_ab_sym_epilogue.txt): at 2x with 5 taps the epilogue now sits on the symmetric-table kernel (rgba -> grey -8 % / fma -20 %,
// -> rgb -5 %, r16g16b16a16 -> rgba -17 %); rgba with 7 taps stays a chain (the accumulator kernel's epilogue is 5 % behind
// resize_stream + convert there).
bool epilogue_preferred(int pixel, int taps_bucket) {
    return pixel >= PICHA_R16 || (pixel == PICHA_RGBA && taps_bucket <= 8 && taps_bucket != 7);
}
int resize_group(DeviceCtx *ctx, const picha_image *src, const picha_image *dst, int n, int filter,
                 float fscale, hipStream_t s, const ChainEpilogue *epi = nullptr, bool *epi_used = nullptr) {
    const picha_image &s0 = src[0], &d0 = dst[0];
    const int pixel = s0.pixel;
    if (epi_used) *epi_used = false;
    if (!epi) {
        long path, rows_forced;
        { std::lock_guard<std::mutex> lk(g_opt_mu); path = g_opt_path; rows_forced = g_opt_rows == 1; }
        if ((path == 0 && !rows_forced) || path == 6) {  // "path" = 6: the run kernel or an error
            bool used = false;
            int rcr = run_group(ctx, src, dst, n, filter, fscale, s, &used);
            if (rcr || used) return rcr;
            if (path == 6) return fail(PICHA_ERR_ARGUMENT, "run kernel forced but unsupported for this geometry");
        }
        if (path == 0 || path == 5) {  // "path" = 5: the row-staged kernel or an error
            bool used = false;
            int rcr = rows_group(ctx, src, dst, n, filter, fscale, 0.0f, 0.0f, 0.0f, s, &used);
            if (rcr || used) return rcr;
            if (path == 5) return fail(PICHA_ERR_ARGUMENT, "row-staged kernel forced but unsupported for this geometry");
        }
    }
    std::shared_ptr<DeviceAxis> ax_ref, ay_ref;  // keep the tables alive while launches are enqueued
    int rc = get_axis(ctx, filter, fscale, s0.width, d0.width, &ax_ref);
    if (rc) return rc;
    rc = get_axis(ctx, filter, fscale, s0.height, d0.height, &ay_ref);
    if (rc) return rc;
    DeviceAxis *ax = ax_ref.get(), *ay = ay_ref.get();

    long opt_path, opt_chunks, opt_numerics, opt_per_lane, opt_rot, opt_dpp, opt_xcd, opt_acc, opt_sym, opt_bw, opt_sym_sync, opt_acc_p2, opt_xch, opt_long_block;
    long opt_work, opt_work_tail, opt_work_tail_div, opt_up;
    {
        std::lock_guard<std::mutex> lk(g_opt_mu);
        opt_up = g_opt_up;
        opt_work = g_opt_work; opt_work_tail = std::max(0L, std::min(1000L, g_opt_work_tail)); opt_work_tail_div = std::max(1L, g_opt_work_tail_div);
        opt_xch = g_opt_xch;
        opt_sym = g_opt_sym; opt_bw = g_opt_block_waves; opt_sym_sync = g_opt_sym_sync; opt_acc_p2 = g_opt_acc_p2; opt_long_block = g_opt_long_block;
        opt_path = g_opt_path; opt_chunks = g_opt_row_chunks; opt_numerics = tl_numerics >= 0 ? tl_numerics : g_opt_numerics; opt_per_lane = g_opt_per_lane;
        opt_rot = g_opt_stream_rot;
        opt_dpp = g_opt_dpp;
        opt_xcd = g_opt_xcd_remap;
        opt_acc = g_opt_acc;
    }
    const int fma = opt_numerics == 1 ? 1 : 0;
    const int is16 = pixel >= PICHA_R16;
    // x-weight flavour of the span/stream kernels: 1/max folded (fma), scaled by 2^22 (denormal unpack), or plain
    const int xw_mode = fma ? 1 : exact_unpack_scaled(kChannels[pixel], is16 != 0) ? 2 : 0;

    // ---- pick the fused configuration: multi-output lanes where the table is uniform, else 1 per lane
    const int ring = std::max(ay->host.span, 1);
    const size_t src_span = image_span(s0), dst_span = image_span(d0);
    bool fused_ok = src_span < (1u << 31) && dst_span < (1u << 31);
    const int al = fused_align(pixel), dal = fused_align(d0.pixel);
    bool src_rows4 = true;  // every source row starts on a 4-byte boundary: aligned sub-dword windows may overrun their row (get_xtable)
    for (int i = 0; i < n && fused_ok; ++i) {
        if (((uintptr_t)src[i].data | (uintptr_t)src[i].stride) % al) fused_ok = false;
        if (((uintptr_t)dst[i].data | (uintptr_t)dst[i].stride) % dal) fused_ok = false;
        if (((uintptr_t)src[i].data | (uintptr_t)src[i].stride) % 4) src_rows4 = false;
    }
    // resize -> colorConvert as a store epilogue: on the symmetric-table kernel where an instantiation exists for the planner's own
    // outputs-per-lane choice (r04: 8-bit rgba / rgb at 2x — the shapes whose chain used to be resize_sym + a convert kernel), else on
    // the one-output accumulator kernel (r03)
    int epi_sym_pl = 0;
    if (epi) {
        if (!fused_ok || opt_path == 1 || opt_path == 3 || s0.width < d0.width || s0.height < d0.height) return PICHA_OK;
        if (opt_path == 0 && opt_sym != 2 && opt_dpp != 1 && opt_rot == 0 && s0.width % d0.width == 0 && s0.height % d0.height == 0) {
            const int sx = s0.width / d0.width, sy = s0.height / d0.height;
            const int pl = opt_per_lane > 0 ? (int)opt_per_lane : sym_preferred_per_lane(pixel, sx, ax->host.taps, fma);
            if (pl > 0 && sym_epi_supported(pixel, pl, sx, ax->host.taps, sy, ay->host.taps)) epi_sym_pl = pl;
        }
        opt_up = 2; opt_xch = 2;
        if (!epi_sym_pl) {
            if (opt_acc == 2) return PICHA_OK;
            opt_acc_p2 = 2; opt_per_lane = 1; opt_path = 4; opt_work = 3;
        }
    }
    int per_lane = 1, step = 1, ktaps = fused_x_bucket(ax->host.taps, pixel);  // includes alignment room where needed
    const DeviceAxis::XTable *xt = nullptr;
    const DeviceAxis::YRows *yr = nullptr;   // accumulator kernel: per-source-row records ...
    const DeviceAxis::XTable *xa = nullptr;  // ... and its x table
    int nacc = 0, acc_per_lane = 1, acc_x_step = 1;
    bool acc_xch = false;
    int ataps = fused_x_bucket(ax->host.taps, pixel, true);  // the accumulator kernel's own (finer) x bucket
    // integer upscales (r03): the accumulator kernel with several outputs per lane sharing one window and as many register
    // accumulators as output rows hold a source row (kernels_impl.inc, PICHA_ACC_UP_SHAPES) instead of the span kernel's LDS
    // ring.  Ring-aliasing tables (box / lanczos at exact factors) are not representable and stay on the span kernel.
    // Measured r03 (64 x 960x540 -> 1080p, profiles/r03_survey_upscale.txt): the 16-bit formats gain 1.4-3.9x over the one-output
    // accumulator kernel that served them (r16 0.11 -> 0.43 of the roof, r16g16 0.21 -> 0.48, r16g16b16a16 0.37 -> 0.50); the 8-bit
    // formats stay 15-25 % behind their span instantiations (the per-row records and completion tests of this kernel outweigh
    // the LDS ring of that one when a lane has 4-16 values per row), so "up" = 0 takes it for 16-bit samples only, 1 always.
    if (fused_ok && (opt_up == 1 || (opt_up == 0 && is16)) && opt_path != 1 && opt_path != 3 && opt_acc != 2 && d0.width >= 2 * s0.width &&
        (opt_per_lane == 0 || opt_per_lane == 2 || opt_per_lane == 4)) {
        const bool sub = kBytes[pixel] % 4 != 0;
        const int pl = sub ? 4 : 2;
        bool dword_rows = true;
        for (int i = 0; i < n && sub; ++i)
            if (((uintptr_t)dst[i].data | (uintptr_t)dst[i].stride) % 4) dword_rows = false;
        const int kt = ax->host.taps + (pl == 4 ? 2 : 1);
        if (dword_rows && (opt_per_lane == 0 || opt_per_lane == pl)) {
            for (int na : {6, 8, 12}) {
                if (!acc_up_supported(pixel, pl, kt, na)) continue;
                const DeviceAxis::YRows *cand = nullptr;
                rc = get_yrows(ctx, ay, na, &cand);
                if (rc) return rc;
                if (!cand->ok) continue;
                const DeviceAxis::XTable *cx = nullptr;
                rc = get_xtable(ctx, ax, pl, 0, kt, xw_mode, is16, &cx);
                if (rc) return rc;
                if (cx->ok) { yr = cand; nacc = na; xa = cx; acc_per_lane = pl; acc_x_step = 0; ataps = kt; }
                break;
            }
        }
    }
    // upscales by >= 2 of the 8-bit sub-dword formats (rgb, grey, greya): four outputs per lane sharing one window, whole-dword
    // stores (the one-output-per-lane span kernel is bound by its 1-3 byte stores: 0.15-0.26 of the roof, r01 survey)
    if (fused_ok && !yr && !epi && (opt_per_lane == 0 || opt_per_lane == 4) && opt_path != 4 && opt_path != 1 && kBytes[pixel] % 4 != 0 && pixel < PICHA_R16 &&
        d0.width >= 2 * s0.width) {
        bool dword_rows = true;
        for (int i = 0; i < n; ++i)
            if (((uintptr_t)dst[i].data | (uintptr_t)dst[i].stride) % 4) dword_rows = false;
        const int kt = ax->host.taps + 2;
        if (dword_rows && fused_multi_supported(pixel, 4, 0, kt) && fused_supported(pixel, kt, ay->host.taps, ring, 4)) {
            const DeviceAxis::XTable *cx = nullptr;
            rc = get_xtable(ctx, ax, 4, 0, kt, xw_mode, is16, &cx);
            if (rc) return rc;
            if (cx->ok) { xt = cx; per_lane = 4; step = 0; ktaps = kt; }
        }
    }
    if (fused_ok && !xt && !yr && !epi && opt_per_lane != 1 && opt_path != 4) {
        // integer decimation / interpolation step of the tap windows
        const int st = std::max(1, (int)std::lround((double)s0.width / d0.width));
        static const int cand[] = {2, 4};
        for (int pl : cand) {
            if (opt_per_lane > 1 && pl != opt_per_lane) continue;
            if (opt_per_lane == 0 && pl != 2) continue;  // default: 2 outputs per lane
            if (!fused_multi_supported(pixel, pl, st, ax->host.taps)) continue;
            if (!fused_supported(pixel, ax->host.taps, ay->host.taps, ring, pl)) continue;
            const DeviceAxis::XTable *cx = nullptr;
            rc = get_xtable(ctx, ax, pl, st, ax->host.taps, xw_mode, is16, &cx);
            if (rc) return rc;
            if (cx->ok) { xt = cx; per_lane = pl; step = st; ktaps = ax->host.taps; break; }
        }
    }
    // ---- streaming kernel (uniform tables in x and y, no LDS): preferred when available
    if (fused_ok && !yr && opt_path != 1 && opt_path != 3 && opt_path != 4) {
        const int sx = std::max(1, (int)std::lround((double)s0.width / d0.width));
        const int sy = std::max(1, (int)std::lround((double)s0.height / d0.height));
        // lane layouts, measured on cfg2 with the XCD-contiguous workgroup order (profiles/r01_ab_xcd_remap.txt): two
        // outputs per lane are the fastest for both numerics (reference arithmetic 2.24 vs 2.37 ms for one output per
        // lane, fma 2.17 vs 2.23 ms); four outputs cost too many registers for dword-sized pixels
        // sub-dword pixels (grey, rgb, greya, ...): several outputs per lane first — one lane then stores whole
        // dwords and converts each source sample once for all its outputs; needs 4-byte aligned destination rows
        const bool subdword = kBytes[pixel] % 4 != 0;
        bool dst_dword_rows = true;
        for (int i = 0; i < n && subdword; ++i)
            if (((uintptr_t)dst[i].data | (uintptr_t)dst[i].stride) % 4) dst_dword_rows = false;
        const int cand_words[] = {2, 1, 4};
        const int cand_sub[] = {4, 2, 1};
        const int (&cand_s)[3] = subdword ? cand_sub : cand_words;
        // cross-lane sharing variant (rgba8, exact 2x, 5 taps): fewest conversions per output and fully
        // coalesced loads, but measured 2-4 % SLOWER than the plain variant on cfg2 (the kernel is bound by
        // mixed read/write HBM traffic, not by the conversions it saves): opt-in only ("dpp" = 1)
        // fills the parameters, sizes the row chunks and launches (resize_stream_kernel family or resize_sym_kernel)
        auto launch_stream = [&](int pl, const DeviceAxis::XTable *cx, const DeviceAxis::YStream *cy, int rot, bool dpp, bool sym, int bw,
                                 bool xch = false) -> int {
            StreamParams p{};
            long long sp = 0, dp = 0;
            TablePair tab;
            std::unique_lock<std::mutex> temp_lock(ctx->temp_mu, std::defer_lock);
            if (uniform_pitch(src, n, &sp) && uniform_pitch(dst, n, &dp)) {
                p.src = static_cast<const uint8_t *>(s0.data);
                p.dst = static_cast<uint8_t *>(d0.data);
                p.src_pitch = sp; p.dst_pitch = dp;
            } else {
                temp_lock.lock();
                int rc2 = make_tables(ctx, src, dst, n, s, &tab);
                if (rc2) return rc2;
                p.src_tab = tab.src; p.dst_tab = tab.dst;
            }
            p.src_stride = s0.stride; p.dst_stride = d0.stride;
            p.src_w = s0.width; p.src_h = s0.height; p.dst_w = d0.width; p.dst_h = d0.height;
            p.src_bytes = (unsigned)src_span; p.dst_bytes = (unsigned)dst_span;
            p.x_left = cx->left; p.x_weight = cx->weight; p.x_taps = ax->host.taps;
            p.x_per_lane = pl; p.x_step = sx; p.x_groups = cx->groups; p.xw_stride = cx->stride;
            p.fma = fma;
            p.y_step_weight = cy->step_weight; p.y_step = sy; p.y_taps = ay->host.taps; p.y_base = cy->base;
            p.rot = rot;
            p.dpp = dpp ? 1 : 0;
            p.xcd_remap = opt_xcd == 1 || (opt_xcd == 0 && (pl >= 2 || dpp)) ? 1 : 0;
            if (sym) {
                memcpy(p.x_canon, cx->canon, sizeof p.x_canon);
                memcpy(p.y_canon, cy->canon, sizeof p.y_canon);
                p.cx0 = cx->c0; p.cx1 = cx->c1; p.x_base = cx->xbase;
                p.block_waves = bw;
                p.sync_steps = opt_sym_sync == 1 ? 1 : 0;
                p.xch = xch ? 1 : 0;
                if (epi) { p.epi_on = 1; p.epi_pixel = d0.pixel; p.epi_r = epi->r; p.epi_g = epi->g; p.epi_b = epi->b; }
            }
            const long long col_waves = (cx->groups + 63) / 64;
            // row chunks: at least ~32 K waves per launch, chunks of >= 16 output rows.  Every chunk re-reads and
            // re-filters TY - SY halo rows.  The reference arithmetic is VALU-bound and wants few, long chunks; the fma
            // flavour is memory-bound and gains 3 % from ~40-row chunks when the halo is short (in XCD-contiguous order
            // the chunks in flight on an XCD then cover a compact piece of the image): profiles/r01_ab_row_chunks.txt
            long long chunks = opt_chunks > 0 ? opt_chunks : (32768 + n * col_waves - 1) / (n * col_waves);
            if (opt_chunks <= 0 && fma && ay->host.taps - sy <= 5) chunks = std::max<long long>(chunks, (d0.height + 39) / 40);
            if (opt_chunks <= 0 && sym) {
                // resize_sym_kernel (r02, profiles/r02_ab_sym_*.txt): waves that live for ~60 output rows (reference arithmetic)
                // or ~20 (fma flavour, memory-bound) keep the chip's working set compact and its tail short; the TY - SY halo
                // rows a chunk re-filters cost less than that gains while the halo stays under ~8 % of the chunk
                // (r05, one source / one destination: the fma flavour gains another 1.1-1.8 % from 12- to 16-row chunks although they re-read 17-25 % more halo
                // rows — short-lived workgroups in address order are what its memory pattern wants; profiles/r05_ab_cfg2_misc.txt, last section)
                const int halo = std::max(0, ay->host.taps - sy);
                const int rows = fma ? std::max(12, (halo * 8 + sy - 1) / sy) : std::max(60, (halo * 12 + sy - 1) / sy);
                chunks = std::max<long long>(chunks, (d0.height + rows - 1) / rows);
            }
            chunks = std::max<long long>(1, std::min<long long>(chunks, std::max(1, opt_chunks > 0 ? d0.height : d0.height / 16)));
            if (opt_chunks <= 0 && !sym) {
                // (r05) batches of SMALL images on the streaming kernel (cfg1: 256 x 640x480 -> 320x240): one whole round of resident workgroups instead of
                // three or four ragged ones — 15 -> 4 chunks: -8 % (fma -10 %); only where a chunk stays short (<= 128 output rows): profiles/r05_ab_cfg1_chunks.txt
                const long long per_chunk = n * ((cx->groups + 255) / 256);
                const int resident = stream_kernel_resident_blocks(pixel, p);
                const long long round = resident > 0 ? resident / per_chunk : 0;
                if (round >= 2 && round < chunks && (d0.height + round - 1) / round <= 128) chunks = round;
            }
            // small launches (a single image): chunks of down to 4 rows for about 3000 waves (profiles/r03_ab_small_launches.txt)
            if (opt_chunks <= 0 && n * col_waves * chunks < 3000)
                chunks = std::max(chunks, std::min<long long>(std::max(1, d0.height / 4), (3000 + n * col_waves - 1) / (n * col_waves)));
            p.rows_per_chunk = (int)((d0.height + chunks - 1) / chunks);
            p.n_chunks = (d0.height + p.rows_per_chunk - 1) / p.rows_per_chunk;
            // work items (sym kernel): explicit list with short chunks last, optionally pulled by persistent workgroups
            const int work_mode = !sym || p.sync_steps ? 0 : opt_work == 3 ? 0 : opt_work > 0 ? (int)opt_work : work_default(n, col_waves, p.n_chunks, false, false);
            if (work_mode) {
                const int nsb = (int)((cx->groups + 64 * bw - 1) / (64 * bw));
                std::shared_ptr<const DeviceCtx::ItemList> il;
            std::shared_ptr<const DeviceCtx::Queue> qhold;
                // short chunks at the end of every sub-list only where a chunk's halo (TY - SY source rows re-filtered) is short:
                // cfg2 (3 rows) gains 1 % from a 20-25 % tail, cfg5 (11 rows) loses 4 % to it (profiles/r03_ab_work_items.txt)
                const int tail_pm = (ay->host.taps - sy) <= 2 * sy ? (int)opt_work_tail : 0;
                int rc2 = get_items(ctx, n, d0.height, nsb, p.rows_per_chunk, std::max(16, p.rows_per_chunk / (int)opt_work_tail_div), tail_pm, &il);
                if (rc2) return rc2;
                p.n_images = n;
                p.work.items = il->items; p.work.items_per_xcd = il->per_xcd;
                if (work_mode == 2) {
                    const int resident = sym_kernel_resident_blocks(pixel, p);
                    if (resident > 0 && (long long)il->per_xcd * 8 > 2LL * resident) {
                        rc2 = get_queue(ctx, s, &p.work.queue, &qhold);
                        if (rc2) return rc2;
                        p.n_chunks = std::max(8, resident / 8 * 8);
                    }
                }
                const hipError_t e = launch_resize_sym_kernel(pixel, p, s, &g_last_kernel);
                if (e != hipSuccess) {
                    if (tab.used) (void)temp_release(ctx->tables, s);
                    return fail(PICHA_ERR_HIP, "streaming resize launch failed: %s", hipGetErrorString(e));
                }
                if (tab.used) return temp_release(ctx->tables, s);
                return PICHA_OK;
            }
            int done = 0;
            while (done < n) {
                const int m = std::min(n - done, 65535);
                StreamParams q = p;
                q.n_images = m;
                if (p.src_tab) { q.src_tab = p.src_tab + done; q.dst_tab = p.dst_tab + done; }
                else { q.src = p.src + (long long)done * sp; q.dst = p.dst + (long long)done * dp; }
                const hipError_t e = sym ? launch_resize_sym_kernel(pixel, q, s, &g_last_kernel) : launch_resize_stream_kernel(pixel, q, s, &g_last_kernel);
                if (e != hipSuccess) {
                    if (tab.used) (void)temp_release(ctx->tables, s);
                    return fail(PICHA_ERR_HIP, "streaming resize launch failed: %s", hipGetErrorString(e));
                }
                done += m;
            }
            if (tab.used) return temp_release(ctx->tables, s);
            return PICHA_OK;
        };
        // symmetric uniform tables (every picha filter at an integer scale): wave-uniform weights, products shared between
        // outputs (kernels_sym.hip).  Needs at least one interior wave to pay.
        if (opt_sym != 2 && opt_dpp != 1 && opt_rot == 0) {
            // outputs per lane: the measured best for this (format, scale, taps, flavour), or what "per_lane" forces
            const int pref = sym_preferred_per_lane(pixel, sx, ax->host.taps, fma);
            const int cand_sym[5] = {epi ? epi_sym_pl : opt_per_lane > 0 ? (int)opt_per_lane : pref, 0, 0, 0, 0};
            for (int pl : cand_sym) {
                if (pl == 0) break;
                if (opt_per_lane > 0 && pl != opt_per_lane) continue;
                if (subdword && ((pl * kBytes[pixel]) % 4 != 0 || !dst_dword_rows)) continue;
                // neighbour exchange (kernels_sym.hip, XchCfg): dword-sized pixels, uniform-base x table whose windows start
                // (taps - step - 1) / 2 pixels before the lane's own pixels
                bool xch = opt_xch != 2 && !subdword && sym_kernel_supported(pixel, pl, sx, ax->host.taps, sy, ay->host.taps, 1) &&
                           (opt_xch == 1 || sym_xch_preferred(pixel, sx, ax->host.taps, pl, fma));
                if (!xch && !sym_kernel_supported(pixel, pl, sx, ax->host.taps, sy, ay->host.taps)) continue;
                const DeviceAxis::XTable *cx = nullptr;
                const DeviceAxis::YStream *cy = nullptr;
                if (xch) {
                    rc = get_xtable(ctx, ax, pl, sx, ax->host.taps, xw_mode, is16, &cx, 0, true);
                    if (rc) return rc;
                    if (!cx->ok || !cx->sym || cx->xbase != -((ax->host.taps - sx - 1) / 2)) {
                        xch = false;
                        if (!sym_kernel_supported(pixel, pl, sx, ax->host.taps, sy, ay->host.taps)) continue;
                    }
                }
                if (!xch) rc = get_xtable(ctx, ax, pl, sx, ax->host.taps, xw_mode, is16, &cx);
                if (rc) return rc;
                rc = get_ystream(ctx, ay, sy, ay->host.taps, &cy);
                if (rc) return rc;
                if (!cx->ok || !cy->ok || !cx->sym || !cy->sym) continue;
                if (cx->c1 - cx->c0 < 2 * 64 * pl) continue;  // needs interior waves to pay
                // waves per workgroup: small groups backfill the SIMDs sooner (reference arithmetic: VALU + latency bound,
                // -3..5 %); the memory-bound fma flavour prefers 4 (neighbouring strips share their edge lines in one L1)
                // (r03: with the explicit work-item list — every launch of >= 6144 waves — the reference arithmetic is as fast with
                // 4 as with 2, 2.184 vs 2.186 ms on cfg2, and 4 keeps the fabric traffic at 1.09x instead of 1.14x: profiles/r03_ab_work_items.txt)
                // (r05, one source / one destination, three devices: the reference arithmetic with the item list is 1.1-1.6 % faster with 1- or 2-wave groups than with 4 —
                // 2.197 -> 2.163, 2.216 -> 2.191, 2.229 -> 2.213 ms on cfg2 — and the fma flavour 0.4-1.1 % slower: profiles/r05_ab_cfg2_misc.txt.  A 4-wave group needs four
                // free slots on one CU at once; 25 % of the slot time of such a launch is spent waiting for that, profiles/r05_timeline.txt)
                // ... for 8-bit formats: -3 .. -5 % at 3x (greya, rgba); the 16-bit formats at 2x (r16g16, r16g16b16, r16g16b16a16: twice the bytes per output, nearer their
                // memory pattern) keep 4 on launches that go through the item list: -1.5 .. -1.7 % (profiles/r05_ab_cfg2_misc.txt, last section)
                const bool big = (long long)n * ((cx->groups + 63) / 64) * std::max(1, d0.height / 60) >= 6144 && opt_work != 3;
                const int bw = opt_bw > 0 ? (int)std::min<long>(opt_bw, 4) : (fma || (is16 && big) ? 4 : 2);
                if (epi && epi_used) *epi_used = true;
                return launch_stream(pl, cx, cy, 0, false, true, bw, xch);
            }
        }
        if (epi) {   // no symmetric-table launch after all: the accumulator kernel's epilogue, or the caller's two-kernel chain
            if (opt_acc == 2) return PICHA_OK;
            opt_acc_p2 = 2; opt_per_lane = 1; opt_path = 4; opt_work = 3;
        }
        bool want_dpp = opt_dpp == 1 && stream_dpp_supported(pixel, sx, ax->host.taps, sy, ay->host.taps);
        for (int pl : cand_s) {
            if (epi) break;   // the general streaming kernels carry no epilogue
            if (want_dpp && pl != 1) continue;
            if (!want_dpp) {
                if (opt_per_lane > 0 && pl != opt_per_lane) continue;
                if (opt_per_lane == 0 && pl == 4 && !subdword) continue;
                if (subdword && pl > 1 && !dst_dword_rows) continue;
            }
            // two variants of the streaming kernel: rotating accumulators (default) or unrolled roles
            // measured with the XCD order and short row chunks (profiles/r01_ab_row_chunks.txt): for two outputs per
            // lane and <= 5 taps the unrolled variant is 1-2 % ahead of the rotating one; otherwise the rotating one
            const bool prefer_rot = opt_rot == 1 || (opt_rot == 0 && !(pl == 2 && ax->host.taps <= 5 && !subdword));
            int rot = want_dpp ? 1 : -1;
            for (int attempt = 0; attempt < 2 && rot < 0; ++attempt) {
                const int r = (attempt == 0) == prefer_rot ? 1 : 0;
                if ((opt_rot == 1 && r == 0) || (opt_rot == 2 && r == 1)) continue;
                if (stream_kernel_supported(pixel, pl, sx, ax->host.taps, sy, ay->host.taps, r)) rot = r;
            }
            if (rot < 0) continue;
            const DeviceAxis::XTable *cx = nullptr;
            const DeviceAxis::YStream *cy = nullptr;
            rc = get_xtable(ctx, ax, pl, sx, ax->host.taps, want_dpp ? fma : (xw_mode), is16, &cx, 0, want_dpp);
            if (rc) return rc;
            rc = get_ystream(ctx, ay, sy, ay->host.taps, &cy);
            if (rc) return rc;
            if (want_dpp && (!cx->ok || !cy->ok || cx->l0 != -1)) {  // needs windows 2x-1 .. 2x+3
                want_dpp = false;
                rot = -1;
                rc = get_xtable(ctx, ax, pl, sx, ax->host.taps, xw_mode, is16, &cx);
                if (rc) return rc;
                for (int r = 1; r >= 0 && rot < 0; --r)
                    if (stream_kernel_supported(pixel, pl, sx, ax->host.taps, sy, ay->host.taps, r)) rot = r;
                if (rot < 0) continue;
            }
            if (!cx->ok || !cy->ok) continue;
            return launch_stream(pl, cx, cy, rot, want_dpp, false, 0);
        }
    }
    // ---- accumulator kernel (no ring): any downscale whose span-kernel ring would be large.  The ring costs
    // (window rows) x 64 lanes x 4..16 B of LDS per wave and caps the occupancy (a 7.5x thumbnail: 22 KB per wave,
    // 7 waves per CU); the accumulator kernel keeps nacc x channels registers instead.
    constexpr int kAccMinRing = 1;  // measured faster than the span kernel for every ring size it was tried on (profiles/r01_survey_acc_vs_span.txt)
    // no bucket (above 36 taps; 34..36 for odd pixel sizes and 36 for 2- and 6-byte pixels, whose aligned fetch needs room)
    const bool acc_long = !yr && ataps == 0 && ax->host.taps > 32 && ax->host.taps <= kAccLongMaxTaps;
    if (acc_long) {  // long windows: LDS-resident weights, tap count padded to half a load segment
        const int half = acc_long_segment(kBytes[pixel]) / 2;
        ataps = (ax->host.taps + half - 1) / half * half;
    }
    if (fused_ok && !yr && per_lane == 1 && opt_path != 1 && opt_path != 3 && opt_acc != 2 && ataps > 0 &&
        (opt_acc == 1 || opt_path == 4 || ring >= kAccMinRing)) {
        for (int na : {4, 6}) {
            if (!acc_supported(pixel, ataps, na)) break;
            const DeviceAxis::YRows *cand = nullptr;
            // (r03) one-row ring aliasing — lanczos / box at even integer scales — is representable here: the march follows a row
            // list in which the aliased row appears once more ahead of its window (get_yrows); both accumulator kernels do
            rc = get_yrows(ctx, ay, na, &cand, true);
            if (rc) return rc;
            if (cand->ok) { yr = cand; nacc = na; break; }
        }
        // ... where it was measured faster than the LDS-ring span kernel that serves such tables otherwise (r03, 64 x 4K sources,
        // profiles/r03_survey_ring_aliasing.txt): always beyond 32 vertical taps (no span instantiation: lanczos 8x went through
        // the two-pass kernels, 6-17x slower); three or four channels from 16 taps (lanczos 4x: -11 .. -22 %, 6x: 1.5-2.3x);
        // greya from 24, grey from 26 (the ring of one float per lane is cheap: a tie at 6x); box and lanczos 2x stay on the ring
        if (yr && yr->src && opt_acc != 1 && opt_path != 4) {
            const int ty = ay->host.taps, ch = kChannels[pixel];
            // (long x windows have no ring kernel to go back to: the alternative would be the two-pass kernels)
            if (!(acc_long || ty > 32 || (ch >= 3 && ty >= 16) || (ch == 2 && ty >= 24) || (ch == 1 && ty >= 26))) { yr = nullptr; nacc = 0; }
        }
        // two outputs per lane (r02): the union of both windows is loaded and converted once; option "acc_p2": 0 auto (where
        // acc_p2_preferred says it was measured faster), 1 whenever a shape exists, 2 never
        int acc_pl = 1, acc_step = 1;
        if (yr && !acc_long && opt_acc_p2 != 2 && opt_xch != 1 && (opt_per_lane == 0 || opt_per_lane == 2)) {
            const int st = (int)std::floor((double)s0.width / d0.width);
            static const int shapes[][2] = {{2, 8}, {3, 12}, {4, 12}, {4, 16}};
            for (auto &e : shapes) {
                if (e[0] != st || !acc_p2_supported(pixel, e[0], e[1], nacc)) continue;
                if (opt_acc_p2 == 0 && !acc_p2_preferred(pixel, e[0], e[1], fma)) continue;
                const DeviceAxis::XTable *cx = nullptr;
                rc = get_xtable(ctx, ax, 2, e[0], e[1], xw_mode, is16, &cx);
                if (rc) return rc;
                if (cx->ok) { xa = cx; acc_pl = 2; acc_step = e[0]; ataps = e[1]; break; }
            }
        }
        // neighbour-exchange form (r03): integer steps of dword-sized pixels; a lane loads only its own step * P pixels and takes
        // the rest of its window from the two neighbouring lanes (kernels_impl.inc, XchCfg).  Option "xch": 0 auto, 1 wherever
        // an instantiation exists, 2 never
        if (yr && !acc_long && opt_xch != 2 && kBytes[pixel] % 4 == 0 && s0.width % d0.width == 0) {
            const int st = s0.width / d0.width, T = ax->host.taps;
            for (int pl : {2, 4}) {
                if (opt_per_lane > 0 && pl != opt_per_lane) continue;
                if (!acc_xch_supported(pixel, pl, st, T, nacc)) continue;
                if (opt_xch == 0 && !acc_xch_preferred(pixel, st, T, pl, fma)) continue;
                const DeviceAxis::XTable *cx = nullptr;
                rc = get_xtable(ctx, ax, pl, st, T, xw_mode, is16, &cx, 0, true);
                if (rc) return rc;
                if (cx->ok && cx->l0 == -((T - st - 1) / 2)) { xa = cx; acc_pl = pl; acc_step = st; ataps = T; acc_xch = true; break; }
            }
        }
        if (yr && acc_pl == 1) {
            rc = get_xtable(ctx, ax, 1, 1, ataps, xw_mode, is16, &xa, (!acc_long && fused_x_bytespan(ataps, pixel)) ? kBytes[pixel] : 0, false, 0, src_rows4);
            if (rc) return rc;
            if (!xa->ok) yr = nullptr;
        }
        acc_per_lane = acc_pl; acc_x_step = acc_step;
    }
    if (epi && (!yr || acc_long || acc_per_lane != 1 || !acc_epi_supported(pixel, ataps, nacc))) return PICHA_OK;  // caller chains
    if (epi && !epi->force && !epilogue_preferred(pixel, ataps)) return PICHA_OK;
    if (opt_path == 4 && !yr) return fail(PICHA_ERR_ARGUMENT, "accumulator kernel forced but unsupported for this geometry");
    const bool use_acc = yr != nullptr;
    if (fused_ok && !xt && !use_acc) {
        fused_ok = ktaps > 0 && fused_supported(pixel, ktaps, ay->host.taps, ring, 1);
        if (fused_ok) {
            rc = get_xtable(ctx, ax, 1, 1, ktaps, xw_mode, is16, &xt, fused_x_bytespan(ktaps, pixel) ? kBytes[pixel] : 0, false, 0, src_rows4);
            if (rc) return rc;
            fused_ok = xt->ok;
        }
    }
    if ((opt_path == 2 || opt_path == 3) && !fused_ok) return fail(PICHA_ERR_ARGUMENT, "fused kernel forced but unsupported for this geometry");
    const bool use_fused = use_acc || (fused_ok && opt_path != 1);
    if (use_acc) { xt = xa; ktaps = ataps; }

    if (use_fused) {
        FusedParams p{};
        long long sp = 0, dp = 0;
        TablePair tab;
        std::unique_lock<std::mutex> temp_lock(ctx->temp_mu, std::defer_lock);
        if (uniform_pitch(src, n, &sp) && uniform_pitch(dst, n, &dp)) {
            p.src = static_cast<const uint8_t *>(s0.data);
            p.dst = static_cast<uint8_t *>(d0.data);
            p.src_pitch = sp;
            p.dst_pitch = dp;
        } else {
            temp_lock.lock();
            rc = make_tables(ctx, src, dst, n, s, &tab);
            if (rc) return rc;
            p.src_tab = tab.src;
            p.dst_tab = tab.dst;
        }
        p.src_stride = s0.stride; p.dst_stride = d0.stride;
        p.src_w = s0.width; p.src_h = s0.height; p.dst_w = d0.width; p.dst_h = d0.height;
        p.src_bytes = (unsigned)src_span; p.dst_bytes = (unsigned)dst_span;
        p.n_images = n;
        p.x_left = xt->left; p.x_weight = xt->weight; p.x_taps = ktaps;
        p.x_per_lane = per_lane; p.x_step = step; p.x_groups = xt->groups; p.xw_stride = xt->stride;
        p.fma = fma;
        p.y_info = ay->y_info; p.y_weight = ay->y_weight; p.y_slot = ay->y_slot;
        p.y_taps = ay->host.taps; p.y_pad = ay->y_pad;
        p.ring_slots = ring;
        p.xcd_remap = opt_xcd == 1 ? 1 : 0;
        if (use_acc) { p.yrow_rec = yr->rec; p.yrow_range = yr->range; p.yrow_src = yr->src; p.nacc = nacc; p.x_per_lane = acc_per_lane; p.x_step = acc_x_step; p.xch = acc_xch ? 1 : 0; }
        if (acc_long) {
            // workgroup of 1..4 waves: the x weights of a wave take x_taps * 256 B of LDS, which bounds the waves a CU holds, and a
            // thumbnail is only 3-8 waves wide, so the waves of a partly used workgroup exit at once and leave their LDS idle.
            // Score = live resident waves per CU, full four-wave workgroups favoured (their waves share rows: measured 10-25 %
            // ahead of single waves at equal residency for pixels of up to 4 bytes, behind for 8-byte pixels);
            // profiles/r03_ab_acc_long.txt
            const int cw = (d0.width + 63) / 64;
            double best = -1.0;
            int bw = 1;
            for (int k = 1; k <= 4; ++k) {
                const long long lds = (long long)ataps * k * 256;
                const int blocks = (int)std::min<long long>(163840 / lds, 16 / k);
                const double live = (double)cw / ((cw + k - 1) / k * k);
                const double bonus = k == 4 ? (kBytes[pixel] >= 6 ? 1.0 : (kBytes[pixel] >= 3 ? 1.3 : 1.2)) : (k == 3 ? 1.05 : 1.0);
                const double sc = blocks * k * live * bonus;
                if (sc > best + 1e-9) { best = sc; bw = k; }
            }
            p.long_block = opt_long_block >= 64 && opt_long_block <= 256 && opt_long_block % 64 == 0 ? (int)opt_long_block : 64 * bw;
        }
        if (epi) { p.epi_on = 1; p.epi_pixel = d0.pixel; p.epi_r = epi->r; p.epi_g = epi->g; p.epi_b = epi->b; if (epi_used) *epi_used = true; }
        // row chunks: enough waves to fill 256 CUs several times over (~32 K), but every chunk re-resamples the `ring`
        // source rows above its first output: keep that halo under ~6 % of the chunk's own source rows unless the
        // launch would then have fewer than ~8 K waves (4K -> 512x288 thumbnails: 18 -> 6 chunks, -5 %)
        const long long col_waves = (xt->groups + 63) / 64;
        long long chunks = opt_chunks > 0 ? opt_chunks : (32768 + n * col_waves - 1) / (n * col_waves);
        if (opt_chunks <= 0) {
            const double src_rows_per_out = std::max(1.0, (double)s0.height / d0.height);
            const long long min_rows = (long long)std::ceil(ring / (0.06 * src_rows_per_out));
            const long long by_halo = std::max<long long>(1, d0.height / std::max<long long>(16, min_rows));
            const long long by_fill = (8192 + n * col_waves - 1) / (n * col_waves);
            chunks = std::min(chunks, std::max(by_halo, by_fill));
        }
        chunks = std::max<long long>(1, std::min<long long>(chunks, std::max(1, opt_chunks > 0 ? d0.height : d0.height / 16)));  // a forced count may go down to one row
        p.rows_per_chunk = (int)((d0.height + chunks - 1) / chunks);
        p.n_chunks = (d0.height + p.rows_per_chunk - 1) / p.rows_per_chunk;
        if (!acc_long && use_acc && opt_chunks <= 0 && n * col_waves * p.n_chunks < 3000) {
            // a handful of images (the plain picha.resize call is ONE): at 16 rows per chunk the launch is a few hundred waves on
            // 256 CUs and takes as long as one wave needs for its rows.  Chunks of down to one row, for about 3000 waves in all
            // (r03, profiles/r03_sweep_acc_small_batches.txt: one 4K rgba -> 500x281 0.141 -> 0.041 ms, one 1080p rgba -> 300x169
            // 0.092 -> 0.019; the optimum sat at 2-4 K waves for every format and geometry tried; from ~16 images on nothing changes)
            const long long c = std::min<long long>(d0.height, (3000 + n * col_waves - 1) / (n * col_waves));
            if (c > p.n_chunks) {
                p.rows_per_chunk = (int)((d0.height + c - 1) / c);
                p.n_chunks = (d0.height + p.rows_per_chunk - 1) / p.rows_per_chunk;
            }
        }
        if (acc_long && use_acc && opt_chunks <= 0) {
            // long windows, small batches (r03, profiles/r03_sweep_long_chunks.txt): LDS holds only 5-13 waves per CU and a thumbnail
            // row is a handful of waves, so 16 images at 16 rows per chunk leave most of the device idle - and while every
            // workgroup is resident at once the launch takes as long as ONE wave needs for its chunk plus the window above it:
            // shorter chunks are faster although each re-resamples that window (16 x 4K -> 320x180 lanczos: 11 chunks 0.32 ms,
            // 22 chunks 0.20; one image -> 400x225: 28 chunks 0.108 ms, 225 chunks 0.045).  Go down to one row per chunk as long as
            // the launch stays within 80 % of what the occupancy query says is resident.  Beyond one round the time is not monotonic in the chunk count (89-100 % of the queried
            // residency already takes two rounds; measured optima move by format): the chunk rule above stays.
            const int resident = acc_kernel_resident_blocks(pixel, p);
            const long long row_blocks = (long long)n * ((col_waves * 64 + p.long_block - 1) / p.long_block);
            // (r05, third session; profiles/r05_ab_acc_long_chunks.txt) one- and two-byte pixels: the time falls with the chunk count up to ONE FREE SLOT PER CU and jumps by 30-50 %
            // at the queried residency itself (64 x 4K grey -> 384x216, 48 taps: 6 chunks 0.433 ms, 8 chunks = 12 of 13 slots 0.349, 9 chunks 0.513; greya -> 333x187 -17 %); and a
            // launch of several ragged rounds is slower than one round of longer chunks (256 images: 6 chunks = 2.8 rounds 1.52 ms, 2 chunks = one round 1.27)
            const bool narrow = kBytes[pixel] <= 2 && p.long_block == 64;
            int cus = 0, dev_id = 0;
            if (narrow && (hipGetDevice(&dev_id) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess)) cus = 0;
            const double cap = narrow && cus > 0 && resident >= 6 * cus ? (double)(resident - cus) : (p.long_block == 256 ? 1.0 : 0.8) * resident;
            for (int c = d0.height; resident > 0 && c > p.n_chunks; --c) {
                const int rows = (d0.height + c - 1) / c, nc = (d0.height + rows - 1) / rows;
                if (row_blocks * nc > cap) continue;
                p.rows_per_chunk = rows; p.n_chunks = nc;
                break;
            }
            // ... and a launch of one to two such rounds is cut back to one: its second round would be partly idle
            // (31 x 24 Mpixel rgba -> 300x200 lanczos: 12 chunks 3.03 ms, 11 chunks 1.80)
            if (resident > 0 && row_blocks * p.n_chunks > cap && (row_blocks * p.n_chunks <= 2.0 * cap || (narrow && row_blocks * 2 <= cap))) {
                for (int c = p.n_chunks - 1; c >= 1; --c) {
                    const int rows = (d0.height + c - 1) / c, nc = (d0.height + rows - 1) / rows;
                    if (row_blocks * nc > cap) continue;
                    p.rows_per_chunk = rows; p.n_chunks = nc;
                    break;
                }
            }
            static const bool trace = getenv("PICHA_HIP_TRACE_PLAN") != nullptr;
            if (trace)
                fprintf(stderr, "[picha_hip] long-window plan: %d resident workgroups of %d threads, %lld workgroups per chunk row -> %d chunks of %d rows\n",
                        resident, p.long_block, row_blocks, p.n_chunks, p.rows_per_chunk);
        }
        // work items (accumulator kernel, windows up to 36 taps)
        const int work_mode = !use_acc || ktaps > 36 ? 0 : opt_work == 3 ? 0 : opt_work > 0 ? (int)opt_work : work_default(n, col_waves, p.n_chunks, true, acc_xch);
        if (work_mode) {
            const int nsb = (xt->groups + 255) / 256;
            std::shared_ptr<const DeviceCtx::ItemList> il;
            std::shared_ptr<const DeviceCtx::Queue> qhold;
            const double src_per_out = std::max(1.0, (double)s0.height / d0.height);
            const int tail_pm = ring <= 2 * src_per_out ? (int)opt_work_tail : 0;   // as above: only where the halo is short
            rc = get_items(ctx, n, d0.height, nsb, p.rows_per_chunk, std::max(16, p.rows_per_chunk / (int)opt_work_tail_div), tail_pm, &il);
            if (rc) return rc;
            p.work.items = il->items; p.work.items_per_xcd = il->per_xcd;
            if (work_mode == 2) {
                const int resident = acc_kernel_resident_blocks(pixel, p);
                if (resident > 0 && (long long)il->per_xcd * 8 > 2LL * resident) {
                    rc = get_queue(ctx, s, &p.work.queue, &qhold);
                    if (rc) return rc;
                    p.n_chunks = std::max(8, resident / 8 * 8);
                }
            }
            const hipError_t e = launch_resize_acc(pixel, p, s, &g_last_kernel);
            if (e != hipSuccess) {
                if (tab.used) (void)temp_release(ctx->tables, s);
                return fail(PICHA_ERR_HIP, "fused resize launch failed: %s", hipGetErrorString(e));
            }
            if (tab.used) return temp_release(ctx->tables, s);
            return PICHA_OK;
        }
        // grid.z limit
        int done = 0;
        while (done < n) {
            const int m = std::min(n - done, 65535);
            FusedParams q = p;
            q.n_images = m;
            if (p.src_tab) { q.src_tab = p.src_tab + done; q.dst_tab = p.dst_tab + done; }
            else { q.src = p.src + (long long)done * sp; q.dst = p.dst + (long long)done * dp; }
            const hipError_t e = use_acc ? launch_resize_acc(pixel, q, s, &g_last_kernel) : launch_resize_fused(pixel, q, s, &g_last_kernel);
            if (e != hipSuccess) {
                if (tab.used) (void)temp_release(ctx->tables, s);
                return fail(PICHA_ERR_HIP, "fused resize launch failed: %s", hipGetErrorString(e));
            }
            done += m;
        }
        if (tab.used) return temp_release(ctx->tables, s);
        return PICHA_OK;
    }

    // two-pass through a stream-ordered scratch, one image at a time
    const AxisTaps &ty = ay->host;
    const int row0 = ty.first_row;
    const int nrows = ty.need[d0.height - 1] - row0 + 1;
    const size_t scratch_bytes = sizeof(float) * (size_t)nrows * d0.width * kChannels[pixel];
    std::lock_guard<std::mutex> temp_lock(ctx->temp_mu);
    rc = temp_acquire(ctx->scratch, scratch_bytes, 0, s);
    if (rc) return rc;
    float *scratch = static_cast<float *>(ctx->scratch.dev);
    for (int i = 0; i < n; ++i) {
        TwoPassParams p{};
        p.src = static_cast<const uint8_t *>(src[i].data);
        p.dst = static_cast<uint8_t *>(dst[i].data);
        p.src_stride = src[i].stride; p.dst_stride = dst[i].stride;
        p.src_w = s0.width; p.src_h = s0.height; p.dst_w = d0.width; p.dst_h = d0.height;
        p.x_left = ax->left; p.x_count = ax->count; p.x_weight = ax->weight; p.x_taps = ax->host.taps;
        p.y_count = ay->count; p.y_eff = ay->eff; p.y_weight = ay->weight; p.y_taps = ay->host.taps;
        p.row0 = row0; p.nrows = nrows; p.scratch = scratch;
        HIP_TRY(launch_resize_twopass(pixel, p, s));
    }
    g_last_kernel = "twopass";
    return temp_release(ctx->scratch, s);
}

int convert_group(DeviceCtx *ctx, const picha_image *src, const picha_image *dst, int n, float r, float g, float b, hipStream_t s) {
    const picha_image &s0 = src[0], &d0 = dst[0];
    if (s0.pixel == d0.pixel) {
        // doColorConvert same-format branch: NativeImage::copy, payload bytes only (src/picha.cc:27-34)
        const size_t rw = (size_t)s0.width * kBytes[s0.pixel];
        for (int i = 0; i < n; ++i)
            HIP_TRY(hipMemcpy2DAsync(dst[i].data, dst[i].stride, src[i].data, src[i].stride, rw, s0.height,
                                      hipMemcpyDeviceToDevice, s));
        return PICHA_OK;
    }
    ConvertParams p{};
    long long sp = 0, dp = 0;
    TablePair tab;
    std::unique_lock<std::mutex> temp_lock(ctx->temp_mu, std::defer_lock);
    if (uniform_pitch(src, n, &sp) && uniform_pitch(dst, n, &dp)) {
        p.src = static_cast<const uint8_t *>(s0.data);
        p.dst = static_cast<uint8_t *>(d0.data);
        p.src_pitch = sp; p.dst_pitch = dp;
    } else {
        temp_lock.lock();
        int rc = make_tables(ctx, src, dst, n, s, &tab);
        if (rc) return rc;
        p.src_tab = tab.src; p.dst_tab = tab.dst;
    }
    p.src_stride = s0.stride; p.dst_stride = d0.stride;
    p.width = s0.width; p.height = s0.height;
    p.r = r; p.g = g; p.b = b;
    { std::lock_guard<std::mutex> lk(g_opt_mu); p.cc_mode = (int)g_opt_cc_mode; }  // one value for the whole launch
    {   // the wide (16-byte access) kernel needs dword-aligned rows on both sides and 32-bit offsets inside an image
        const size_t sspan = image_span(s0), dspan = image_span(d0);
        bool wide = sspan < (1ull << 32) - 64 && dspan < (1ull << 32) - 64;
        for (int i = 0; i < n && wide; ++i)
            if ((((uintptr_t)src[i].data | (uintptr_t)src[i].stride | (uintptr_t)dst[i].data | (uintptr_t)dst[i].stride) & 3) != 0) wide = false;
        p.wide_ok = wide ? 1 : 0;
        p.src_bytes = (unsigned)sspan; p.dst_bytes = (unsigned)dspan;
    }
    int done = 0;
    while (done < n) {
        const int m = std::min(n - done, 65535);
        ConvertParams q = p;
        q.n_images = m;
        if (p.src_tab) { q.src_tab = p.src_tab + done; q.dst_tab = p.dst_tab + done; }
        else { q.src = p.src + (long long)done * sp; q.dst = p.dst + (long long)done * dp; }
        const hipError_t e = launch_color_convert(s0.pixel, d0.pixel, q, s, &g_last_kernel);
        if (e != hipSuccess) {
            if (tab.used) (void)temp_release(ctx->tables, s);  // the table upload and earlier launches are still in flight
            return fail(PICHA_ERR_HIP, "colour-convert launch failed: %s", hipGetErrorString(e));
        }
        done += m;
    }
    if (tab.used) return temp_release(ctx->tables, s);
    return PICHA_OK;
}

// Host-pointer entry points: stream + pinned staging buffer + device arena travel together as one "lane", leased from
// a per-device pool for the duration of a call.  (They used to be thread_local and were never freed: a service
// that spawns a thread per request leaked ~50 MB pinned + ~50 MB HBM per dead thread.)  The leasing policy — how many idle
// lanes are kept, when and by whom surplus ones are destroyed — is lane_pool.h (run under TSan / ASan by tests/sanitize).
struct HostLane {
    hipStream_t s = nullptr;
    uint8_t *pinned = nullptr; size_t pinned_cap = 0;
    uint8_t *arena = nullptr;  size_t arena_cap = 0;
    std::chrono::steady_clock::time_point idle_since;
    size_t held_bytes() const { return pinned_cap + arena_cap; }   // pinned host staging + device arena
};
using LanePool = LanePoolT<HostLane>;
std::mutex g_lane_mu;
std::map<int, LanePool *> g_lane_pools;  // never destroyed: static destruction order vs. the HIP runtime

LanePool *lane_pool(int dev) {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    auto &p = g_lane_pools[dev];
    if (!p) p = new LanePool;
    return p;
}
void lane_destroy(HostLane *l) {  // errors ignored: also runs at teardown
    if (l->s) (void)hipStreamDestroy(l->s);
    if (l->pinned) (void)hipHostFree(l->pinned);
    if (l->arena) (void)hipFree(l->arena);
    delete l;
}
struct LaneLease {
    int dev = -1;
    HostLane *lane = nullptr;
    ~LaneLease() {
        if (lane) lane_pool(dev)->give(lane);   // nothing is freed on a call's return path
    }
    int acquire(int device) {
        dev = device;
        LanePool *pool = lane_pool(dev);
        std::vector<HostLane *> surplus;
        lane = pool->take(&surplus);
        for (HostLane *l : surplus) lane_destroy(l);  // rare: lanes idle for 10 s beyond what the pool keeps
        if (!lane) {
            lane = new HostLane;
            hipError_t e = hipStreamCreateWithFlags(&lane->s, hipStreamNonBlocking);
            if (e != hipSuccess) {
                delete lane; lane = nullptr;
                pool->cancel();
                return fail(PICHA_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
            }
        }
        return PICHA_OK;
    }
    int reserve(size_t bytes) {  // grow-only within the lane
        if (lane->pinned_cap < bytes) {
            if (lane->pinned) (void)hipHostFree(lane->pinned);
            lane->pinned = nullptr; lane->pinned_cap = 0;
            const size_t cap = align_up(bytes + bytes / 4, 1 << 20);
            HIP_TRY(hipHostMalloc((void **)&lane->pinned, cap, hipHostMallocDefault));
            lane->pinned_cap = cap;
        }
        if (lane->arena_cap < bytes) {
            if (lane->arena) HIP_TRY(hipFree(lane->arena));
            lane->arena = nullptr; lane->arena_cap = 0;
            const size_t cap = align_up(bytes + bytes / 4, 1 << 20);
            HIP_TRY(hipMalloc((void **)&lane->arena, cap));
            lane->arena_cap = cap;
        }
        return PICHA_OK;
    }
};

enum class Op { Resize, Convert, Chain };

extern "C" void picha_hip_staging_copy(void *dst, const void *src, size_t bytes);  // dispatch.cpp

// Shared host-pointer driver: stage in, run the device entry point, stage out payload rows.
int run_host(Op op, int n, const picha_image *src, const picha_image *dst, int filter, float fscale,
             float r, float g, float b, int order = 0) {
    if (n < 0 || (n > 0 && (!src || !dst))) return fail(PICHA_ERR_ARGUMENT, "bad batch arguments");
    if (n == 0) return PICHA_OK;
    for (int i = 0; i < n; ++i) {
        int rc = check_src(&src[i]);
        if (rc) return rc;
        rc = check_dst(&dst[i]);
        if (rc) return rc;
    }
    int dev;
    int rc = current_device(&dev);
    if (rc) return rc;
    LaneLease lease;
    rc = lease.acquire(dev);
    if (rc) return rc;
    hipStream_t s = lease.lane->s;

    // one device arena and one pinned host mirror of it: [src images | dst images], 256-byte aligned.
    // Transfers go through the pinned mirror as two 1-D copies (pageable / 2-D async copies are
    // staged by the runtime through its own bounce buffers anyway: one pinned mirror and 1-D copies keep
    // every byte of the transfer on this lane's stream).
    std::vector<size_t> soff(n), doff(n);
    size_t total = 0;
    for (int i = 0; i < n; ++i) { soff[i] = total; total += align_up(image_span(src[i]), 256); }
    const size_t dst_begin = total;
    for (int i = 0; i < n; ++i) { doff[i] = total; total += align_up(image_span(dst[i]), 256); }
    rc = lease.reserve(total + 256);
    if (rc) return rc;
    uint8_t *pinned = lease.lane->pinned, *arena = lease.lane->arena;
    std::vector<picha_image> dsrc(src, src + n), ddst(dst, dst + n);
    // stage in pieces: while the DMA engine moves one piece, the next one is being copied into the pinned mirror
    // (by a few threads for large pieces, dispatch.cpp) — a 4K image is 33 MB, one memcpy of it takes longer than its H2D
    constexpr size_t kPiece = 8u << 20;
    hipError_t e0 = hipSuccess;
    for (int i = 0; i < n; ++i) {
        dsrc[i].data = arena + soff[i];
        ddst[i].data = arena + doff[i];
        const size_t span = image_span(src[i]);
        for (size_t off = 0; off < span && e0 == hipSuccess; off += kPiece) {
            const size_t len = std::min(kPiece, span - off);
            picha_hip_staging_copy(pinned + soff[i] + off, static_cast<const uint8_t *>(src[i].data) + off, len);
            e0 = hipMemcpyAsync(arena + soff[i] + off, pinned + soff[i] + off, len, hipMemcpyHostToDevice, s);
        }
    }
    if (e0 != hipSuccess) rc = fail(PICHA_ERR_HIP, "H2D failed: %s", hipGetErrorString(e0));
    if (rc == PICHA_OK)
        rc = (op == Op::Resize)    ? picha_hip_resize_device(n, dsrc.data(), ddst.data(), filter, fscale, s)
             : (op == Op::Convert) ? picha_hip_color_convert_device(n, dsrc.data(), ddst.data(), r, g, b, s)
                                   : picha_hip_convert_resize_device(n, dsrc.data(), ddst.data(), order, r, g, b, filter, fscale, s);
    if (rc == PICHA_OK) {
        hipError_t e = hipMemcpyAsync(pinned + dst_begin, arena + dst_begin, total - dst_begin, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) rc = fail(PICHA_ERR_HIP, "D2H failed: %s", hipGetErrorString(e));
    }
    hipError_t e2 = hipStreamSynchronize(s);
    if (rc == PICHA_OK && e2 != hipSuccess)
        rc = fail(PICHA_ERR_HIP, "stream completion failed: %s", hipGetErrorString(e2));
    if (rc == PICHA_OK) {
        for (int i = 0; i < n; ++i) {
            // payload bytes only: dst row padding is never written (src/resize.cc:122-132, src/picha.cc:27-34)
            const size_t rw = (size_t)dst[i].width * kBytes[dst[i].pixel];
            const uint8_t *from = pinned + doff[i];
            uint8_t *to = static_cast<uint8_t *>(dst[i].data);
            if ((size_t)dst[i].stride == rw) picha_hip_staging_copy(to, from, rw * dst[i].height);
            else for (int y = 0; y < dst[i].height; ++y) memcpy(to + (size_t)y * dst[i].stride, from + (size_t)y * dst[i].stride, rw);
        }
    }
    return rc;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int picha_pixel_bytes(int pixel) { return valid_pixel(pixel) ? kBytes[pixel] : 0; }
int picha_pixel_channels(int pixel) { return valid_pixel(pixel) ? kChannels[pixel] : 0; }
int picha_row_stride(int width, int pixel) {  // NativeImage::row_stride (src/picha.h:214) in 64-bit; 0 when it does not fit an int
    const long long st = ((long long)picha_pixel_bytes(pixel) * width + 3) & ~3LL;
    return (width < 0 || st > INT32_MAX) ? 0 : (int)st;
}

int picha_pixel_from_name(const char *name) {
    if (!name) return PICHA_INVALID_PIXEL;
    for (int i = 0; i < 8; ++i)
        if (strcmp(name, kPixelNames[i]) == 0) return i;
    return PICHA_INVALID_PIXEL;
}
const char *picha_pixel_name(int pixel) { return valid_pixel(pixel) ? kPixelNames[pixel] : nullptr; }
int picha_filter_from_name(const char *name) {
    if (!name) return -1;
    for (int i = 0; i < 6; ++i)
        if (strcmp(name, kFilterNames[i]) == 0) return i;
    return -1;
}
const char *picha_filter_name(int f) { return (f >= 0 && f < 6) ? kFilterNames[f] : nullptr; }

int picha_resize_options(int has_filter, int filter, int has_filter_scale, double filter_scale,
                         int *out_filter, float *out_scale) {
    // ResizeOptions() : filter(CubicFilterTag), width(0.70f)   src/resize.cc:173-177
    int f = PICHA_FILTER_CUBIC;
    float w = 0.70f;
    if (has_filter) {  // src/resize.cc:181-188
        w = 1.0f;
        f = filter;
        if (f < 0 || f >= PICHA_NUM_FILTERS) return fail(PICHA_ERR_INVALID_FILTER, "invalid filter mode");
    }
    if (has_filter_scale) {  // src/resize.cc:190-196
        w = (float)filter_scale;
        if (w != w || w <= 0) return fail(PICHA_ERR_INVALID_FILTER_WIDTH, "invalid filter width");
    }
    if (out_filter) *out_filter = f;
    if (out_scale) *out_scale = w;
    return PICHA_OK;
}

void picha_color_settings(double red, double green, double blue, float out[3]) {
    // ColorSettings() src/colorconvert.h:11-14 ; getSettings src/colorconvert.cc:6-22
    float rf = 0.299f, gf = 0.587f, bf = float(0.114);
    if (red == red) rf = (float)red;
    if (green == green) gf = (float)green;
    if (blue == blue) bf = (float)blue;
    const float n = 1.0f / (rf + gf + bf);
    out[0] = rf * n;
    out[1] = gf * n;
    out[2] = bf * n;
}

int picha_hip_make_contribs(int filter, float filter_scale, int src_size, int dst_size, int32_t *left,
                            int32_t *right, float *weights, int max_taps, int *taps_out) {
    if (!left || !right || !weights) return fail(PICHA_ERR_ARGUMENT, "null output");
    AxisTaps t;
    int rc = build_axis_taps(filter, filter_scale, src_size, dst_size, t);
    if (rc) return fail(rc, "%s", picha_hip_strerror(rc));
    if (taps_out) *taps_out = t.taps;
    if (t.taps > max_taps) return fail(PICHA_ERR_ARGUMENT, "max_taps too small: need %d", t.taps);
    for (int i = 0; i < dst_size; ++i) {
        left[i] = t.left[i];
        right[i] = t.left[i] + t.count[i] - 1;
        for (int j = 0; j < max_taps; ++j) weights[(size_t)i * max_taps + j] = j < t.taps ? t.weight[(size_t)i * t.taps + j] : 0.0f;
    }
    return PICHA_OK;
}

const char *picha_hip_last_error(void) { return g_last_error.c_str(); }
const char *picha_hip_last_kernel(void) { return g_last_kernel; }

const char *picha_hip_strerror(int status) {
    switch (status) {
    case PICHA_OK: return "ok";
    case PICHA_ERR_INVALID_IMAGE: return "invalid image";
    case PICHA_ERR_INVALID_DIMENSIONS: return "invalid dimensions";
    case PICHA_ERR_INVALID_FILTER: return "invalid filter mode";
    case PICHA_ERR_INVALID_FILTER_WIDTH: return "invalid filter width";
    case PICHA_ERR_PIXEL_MISMATCH: return "pixel format or size mismatch between source and destination";
    case PICHA_ERR_EXPECTED_PIXEL: return "expected pixel mode";
    case PICHA_ERR_DEGENERATE_TAPS: return "filter taps sum to a non-positive weight";
    case PICHA_ERR_NO_DEVICE: return "no HIP device available";
    case PICHA_ERR_HIP: return "HIP runtime failure";
    case PICHA_ERR_ARGUMENT: return "invalid argument";
    }
    return "unknown status";
}

int picha_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int picha_hip_device_arch(char *buf, size_t len) {
    int dev;
    int rc = current_device(&dev);
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, len, "%s", prop.gcnArchName);
    return PICHA_OK;
}

int picha_hip_set_option(const char *key, long value) {
    if (!key) return fail(PICHA_ERR_ARGUMENT, "null key");
    std::lock_guard<std::mutex> lk(g_opt_mu);
    if (strcmp(key, "path") == 0) { g_opt_path = value; return PICHA_OK; }
    if (strcmp(key, "row_chunks") == 0) { g_opt_row_chunks = value; return PICHA_OK; }
    if (strcmp(key, "numerics") == 0) { g_opt_numerics = value; return PICHA_OK; }
    if (strcmp(key, "per_lane") == 0) { g_opt_per_lane = value; return PICHA_OK; }
    if (strcmp(key, "stream_rot") == 0) { g_opt_stream_rot = value; return PICHA_OK; }
    if (strcmp(key, "dpp") == 0) { g_opt_dpp = value; return PICHA_OK; }
    if (strcmp(key, "sym") == 0) { g_opt_sym = value; return PICHA_OK; }
    if (strcmp(key, "rows") == 0) { g_opt_rows = value; return PICHA_OK; }
    if (strcmp(key, "run") == 0) { g_opt_run = value; return PICHA_OK; }
    if (strcmp(key, "run_round") == 0) { g_opt_run_round = value; return PICHA_OK; }
    if (strcmp(key, "fuse") == 0) { g_opt_fuse = value; return PICHA_OK; }
    if (strcmp(key, "cc_mode") == 0) { g_opt_cc_mode = value; return PICHA_OK; }
    if (strcmp(key, "block_waves") == 0) { g_opt_block_waves = value; return PICHA_OK; }
    if (strcmp(key, "sym_sync") == 0) { g_opt_sym_sync = value; return PICHA_OK; }
    if (strcmp(key, "xch") == 0) { g_opt_xch = value; return PICHA_OK; }
    if (strcmp(key, "run_xch") == 0) { g_opt_run_xch = value; return PICHA_OK; }
    if (strcmp(key, "run_ws") == 0) { g_opt_run_ws = value; return PICHA_OK; }
    if (strcmp(key, "run_phase") == 0) { g_opt_run_phase = value; return PICHA_OK; }
    if (strcmp(key, "run_shu") == 0) { g_opt_run_shu = value; return PICHA_OK; }
    if (strcmp(key, "run_efuse") == 0) { g_opt_run_efuse = value; return PICHA_OK; }
    if (strcmp(key, "edge_rows") == 0) { g_opt_edge_rows = value; return PICHA_OK; }
    if (strcmp(key, "run_edges") == 0) { if (value < 0 || value > 3) return fail(PICHA_ERR_ARGUMENT, "run_edges: 0..3"); g_opt_run_edges = value; return PICHA_OK; }
    if (strcmp(key, "up") == 0) { g_opt_up = value; return PICHA_OK; }
    if (strcmp(key, "work") == 0) { g_opt_work = value; return PICHA_OK; }
    if (strcmp(key, "work_tail") == 0) { g_opt_work_tail = value; return PICHA_OK; }
    if (strcmp(key, "work_tail_div") == 0) { g_opt_work_tail_div = value; return PICHA_OK; }
    if (strcmp(key, "acc_p2") == 0) { g_opt_acc_p2 = value; return PICHA_OK; }
    if (strcmp(key, "long_block") == 0) { g_opt_long_block = value; return PICHA_OK; }
    if (strcmp(key, "xcd_remap") == 0) { g_opt_xcd_remap = value; return PICHA_OK; }
    if (strcmp(key, "acc") == 0) { g_opt_acc = value; return PICHA_OK; }
    if (strcmp(key, "async_devices") == 0) { g_opt_async_devices = value; return PICHA_OK; }
    if (strcmp(key, "async_virtual_devices") == 0) { g_opt_async_virtual = value; return PICHA_OK; }
    if (strcmp(key, "chain_window_mb") == 0) { g_opt_chain_window_mb = value; return PICHA_OK; }
    return fail(PICHA_ERR_ARGUMENT, "unknown option %s", key);
}
long picha_hip_get_option(const char *key) {
    if (!key) return -1;
    std::lock_guard<std::mutex> lk(g_opt_mu);
    if (strcmp(key, "path") == 0) return g_opt_path;
    if (strcmp(key, "row_chunks") == 0) return g_opt_row_chunks;
    if (strcmp(key, "numerics") == 0) return g_opt_numerics;
    if (strcmp(key, "per_lane") == 0) return g_opt_per_lane;
    if (strcmp(key, "stream_rot") == 0) return g_opt_stream_rot;
    if (strcmp(key, "dpp") == 0) return g_opt_dpp;
    if (strcmp(key, "sym") == 0) return g_opt_sym;
    if (strcmp(key, "rows") == 0) return g_opt_rows;
    if (strcmp(key, "run") == 0) return g_opt_run;
    if (strcmp(key, "run_round") == 0) return g_opt_run_round;
    if (strcmp(key, "fuse") == 0) return g_opt_fuse;
    if (strcmp(key, "cc_mode") == 0) return g_opt_cc_mode;
    if (strcmp(key, "block_waves") == 0) return g_opt_block_waves;
    if (strcmp(key, "sym_sync") == 0) return g_opt_sym_sync;
    if (strcmp(key, "xch") == 0) return g_opt_xch;
    if (strcmp(key, "run_xch") == 0) return g_opt_run_xch;
    if (strcmp(key, "run_ws") == 0) return g_opt_run_ws;
    if (strcmp(key, "run_phase") == 0) return g_opt_run_phase;
    if (strcmp(key, "run_shu") == 0) return g_opt_run_shu;
    if (strcmp(key, "run_efuse") == 0) return g_opt_run_efuse;
    if (strcmp(key, "edge_rows") == 0) return g_opt_edge_rows;
    if (strcmp(key, "run_edges") == 0) return g_opt_run_edges;
    if (strcmp(key, "up") == 0) return g_opt_up;
    if (strcmp(key, "work") == 0) return g_opt_work;
    if (strcmp(key, "work_tail") == 0) return g_opt_work_tail;
    if (strcmp(key, "work_tail_div") == 0) return g_opt_work_tail_div;
    if (strcmp(key, "acc_p2") == 0) return g_opt_acc_p2;
    if (strcmp(key, "long_block") == 0) return g_opt_long_block;
    if (strcmp(key, "xcd_remap") == 0) return g_opt_xcd_remap;
    if (strcmp(key, "acc") == 0) return g_opt_acc;
    if (strcmp(key, "async_devices") == 0) return g_opt_async_devices;
    if (strcmp(key, "async_virtual_devices") == 0) return g_opt_async_virtual;
    if (strcmp(key, "chain_window_mb") == 0) return g_opt_chain_window_mb;
    return -1;
}

// ---- device-resident entry points ---------------------------------------------------------------
int picha_hip_resize_device(int n, const picha_image *src, const picha_image *dst, int filter,
                            float filter_scale, void *stream) {
    if (n < 0 || (n > 0 && (!src || !dst))) return fail(PICHA_ERR_ARGUMENT, "bad batch arguments");
    if (filter < 0 || filter >= PICHA_NUM_FILTERS) return fail(PICHA_ERR_INVALID_FILTER, "invalid filter mode");
    if (!(filter_scale > 0)) return fail(PICHA_ERR_INVALID_FILTER_WIDTH, "invalid filter width");
    for (int i = 0; i < n; ++i) {
        int rc = check_src(&src[i]);
        if (rc) return rc;
        rc = check_dst(&dst[i]);
        if (rc) return rc;
        if (src[i].pixel != dst[i].pixel)  // assert(src.pixel == dst.pixel) src/resize.cc:137
            return fail(PICHA_ERR_PIXEL_MISMATCH, "resize: source and destination pixel formats differ");
    }
    if (n == 0) return PICHA_OK;
    int dev;
    int rc = current_device(&dev);
    if (rc) return rc;
    DeviceCtx *ctx = ctx_for(dev);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int i = 0;
    while (i < n) {  // runs of equal geometry -> one launch each
        int j = i + 1;
        while (j < n && same_geometry(src[j], src[i]) && same_geometry(dst[j], dst[i])) ++j;
        rc = resize_group(ctx, src + i, dst + i, j - i, filter, filter_scale, s);
        if (rc) return rc;
        i = j;
    }
    return PICHA_OK;
}

int picha_hip_color_convert_device(int n, const picha_image *src, const picha_image *dst, float r,
                                   float g, float b, void *stream) {
    if (n < 0 || (n > 0 && (!src || !dst))) return fail(PICHA_ERR_ARGUMENT, "bad batch arguments");
    for (int i = 0; i < n; ++i) {
        int rc = check_src(&src[i]);
        if (rc) return rc;
        rc = check_dst(&dst[i]);
        if (rc) return rc;
        if (src[i].width != dst[i].width || src[i].height != dst[i].height)  // asserts src/colorconvert.cc:138-139
            return fail(PICHA_ERR_PIXEL_MISMATCH, "colorConvert: source and destination sizes differ");
    }
    if (n == 0) return PICHA_OK;
    int dev;
    int rc = current_device(&dev);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int i = 0;
    while (i < n) {
        int j = i + 1;
        while (j < n && same_geometry(src[j], src[i]) && same_geometry(dst[j], dst[i])) ++j;
        rc = convert_group(ctx_for(dev), src + i, dst + i, j - i, r, g, b, s);
        if (rc) return rc;
        i = j;
    }
    return PICHA_OK;
}

// ---- chained colorConvert + resize ---------------------------------------------
int picha_hip_convert_resize_device(int n, const picha_image *src, const picha_image *dst, int order, float r,
                                    float g, float b, int filter, float filter_scale, void *stream) {
    if (n < 0 || (n > 0 && (!src || !dst))) return fail(PICHA_ERR_ARGUMENT, "bad batch arguments");
    if (order != PICHA_CONVERT_THEN_RESIZE && order != PICHA_RESIZE_THEN_CONVERT)
        return fail(PICHA_ERR_ARGUMENT, "bad chain order");
    if (filter < 0 || filter >= PICHA_NUM_FILTERS) return fail(PICHA_ERR_INVALID_FILTER, "invalid filter mode");
    if (!(filter_scale > 0)) return fail(PICHA_ERR_INVALID_FILTER_WIDTH, "invalid filter width");
    for (int i = 0; i < n; ++i) {
        int rc = check_src(&src[i]);
        if (rc) return rc;
        rc = check_dst(&dst[i]);
        if (rc) return rc;
    }
    if (n == 0) return PICHA_OK;
    int dev;
    int rc = current_device(&dev);
    if (rc) return rc;
    DeviceCtx *ctx = ctx_for(dev);
    hipStream_t s = static_cast<hipStream_t>(stream);
    long window_mb;
    { std::lock_guard<std::mutex> lk(g_opt_mu); window_mb = g_opt_chain_window_mb; }
    const size_t window = (size_t)std::max(1L, window_mb) << 20;
    int i = 0;
    while (i < n) {  // runs of equal geometry
        int j = i + 1;
        while (j < n && same_geometry(src[j], src[i]) && same_geometry(dst[j], dst[i])) ++j;
        const picha_image &s0 = src[i], &d0 = dst[i];
        if (s0.pixel == d0.pixel) {  // the convert step is the reference's payload copy: nothing to chain
            rc = resize_group(ctx, src + i, dst + i, j - i, filter, filter_scale, s);
        } else if ([&] {  // colour convert fused into the row-staged resize: the intermediate never exists in HBM
                       bool used = false;
                       rc = PICHA_OK;
                       long fuse, path, numerics;
                       { std::lock_guard<std::mutex> lk(g_opt_mu); fuse = g_opt_fuse; path = g_opt_path; numerics = tl_numerics >= 0 ? tl_numerics : g_opt_numerics; }
                       // like resize_group: a forced kernel family ("path" 1..4) is honoured, 5 means this kernel or an error
                       if (order == PICHA_CONVERT_THEN_RESIZE && fuse != 2 && (path == 0 || path == 5)) {
                           rc = rows_group(ctx, src + i, dst + i, j - i, filter, filter_scale, r, g, b, s, &used);
                           if (rc == PICHA_OK && !used && path == 5)
                               rc = fail(PICHA_ERR_ARGUMENT, "row-staged kernel forced but unsupported for this chain");
                       }
                       // colorConvert -> integer upscale (r04): the convert per window pixel inside the ring kernel
                       if (order == PICHA_CONVERT_THEN_RESIZE && fuse != 2 && path == 0 && rc == PICHA_OK && !used &&
                           (fuse == 1 || (s0.width > 0 && convert_up_preferred(s0.pixel, d0.pixel, d0.width / s0.width, numerics == 1)))) {
                           const ChainConvert cv{r, g, b};
                           rc = run_group(ctx, src + i, dst + i, j - i, filter, filter_scale, s, &used, &cv);
                       }
                       // resize -> colorConvert: the convert as a store epilogue of the accumulator kernel (r03), where the
                       // intermediate image would be large (downscales by up to ~4x: tap buckets 5 .. 16)
                       if (order == PICHA_RESIZE_THEN_CONVERT && fuse != 2 && (path == 0 || path == 4)) {
                           const ChainEpilogue epi{r, g, b, fuse == 1};
                           rc = resize_group(ctx, src + i, dst + i, j - i, filter, filter_scale, s, &epi, &used);
                       }
                       return rc != PICHA_OK || used;
                   }()) {
            // done (or failed) in the fused kernel
        } else {  // (equal sizes still resize: a 1:1 resample is not the identity for every filter)
            // the reference's intermediate image: newJsImage(w, h, pixel), stride = row_stride (src/picha.cc:119-133)
            picha_image mid;
            mid.pixel = order == PICHA_CONVERT_THEN_RESIZE ? d0.pixel : s0.pixel;
            mid.width = order == PICHA_CONVERT_THEN_RESIZE ? s0.width : d0.width;
            mid.height = order == PICHA_CONVERT_THEN_RESIZE ? s0.height : d0.height;
            mid.stride = picha_row_stride(mid.width, mid.pixel);
            mid.data = nullptr;
            const size_t mid_bytes = align_up((size_t)mid.stride * mid.height, 256);
            const int slice = (int)std::min<size_t>((size_t)(j - i), std::max<size_t>(1, window / mid_bytes));
            std::lock_guard<std::mutex> lk(ctx->chain_mu);
            rc = temp_acquire(ctx->chain, mid_bytes * slice, 0, s);
            if (rc) return rc;
            std::vector<picha_image> mids(slice, mid);
            for (int k = 0; k < slice; ++k) mids[k].data = static_cast<uint8_t *>(ctx->chain.dev) + mid_bytes * k;
            // slice by slice on one in-order stream; the buffer is reused by the next slice
            for (int k = i; k < j && rc == PICHA_OK; k += slice) {
                const int m = std::min(slice, j - k);
                std::string first;
                if (order == PICHA_CONVERT_THEN_RESIZE) {
                    rc = convert_group(ctx, src + k, mids.data(), m, r, g, b, s);
                    first = g_last_kernel;
                    if (rc == PICHA_OK) rc = resize_group(ctx, mids.data(), dst + k, m, filter, filter_scale, s);
                } else {
                    rc = resize_group(ctx, src + k, mids.data(), m, filter, filter_scale, s);
                    first = g_last_kernel;
                    if (rc == PICHA_OK) rc = convert_group(ctx, mids.data(), dst + k, m, r, g, b, s);
                }
                if (rc == PICHA_OK) {  // two kernels ran: picha_hip_last_kernel names both
                    snprintf(g_chain_name, sizeof g_chain_name, "%s + %s", first.c_str(), g_last_kernel);
                    g_last_kernel = g_chain_name;
                }
            }
            int rc2 = temp_release(ctx->chain, s);
            if (rc == PICHA_OK) rc = rc2;
        }
        if (rc) return rc;
        i = j;
    }
    return PICHA_OK;
}

// ---- Image helpers on device images ----------------------------------------------
int picha_hip_image_compare_device(int n, const picha_image *a, const picha_image *b, uint64_t *sum_abs_diff,
                                   uint64_t *differing_bytes, void *stream) {
    if (n < 0 || (n > 0 && (!a || !b))) return fail(PICHA_ERR_ARGUMENT, "bad batch arguments");
    for (int i = 0; i < n; ++i) {
        int rc = check_src(&a[i]);
        if (rc) return rc;
        rc = check_src(&b[i]);
        if (rc) return rc;
    }
    if (n == 0) return PICHA_OK;
    int dev;
    int rc = current_device(&dev);
    if (rc) return rc;
    DeviceCtx *ctx = ctx_for(dev);
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::lock_guard<std::mutex> lk(ctx->results_mu);
    // device: [n][2] sums, then the two pointer tables; pinned mirror of the same layout
    const size_t sums = align_up(sizeof(uint64_t) * 2 * (size_t)n, 256), tabs = sizeof(void *) * 2 * (size_t)n;
    rc = temp_acquire(ctx->results, sums + tabs, sums + tabs, s);
    if (rc) return rc;
    uint8_t *dbase = static_cast<uint8_t *>(ctx->results.dev), *hbase = static_cast<uint8_t *>(ctx->results.host);
    const void **htab = reinterpret_cast<const void **>(hbase + sums);
    for (int i = 0; i < n; ++i) { htab[i] = a[i].data; htab[n + i] = b[i].data; }
    HIP_TRY(hipMemsetAsync(dbase, 0, sums, s));
    HIP_TRY(hipMemcpyAsync(dbase + sums, htab, tabs, hipMemcpyHostToDevice, s));
    const void *const *dtab = reinterpret_cast<const void *const *>(dbase + sums);
    std::vector<char> comparable(n, 0);
    int i = 0;
    while (i < n) {
        int j = i + 1;
        while (j < n && same_geometry(a[j], a[i]) && same_geometry(b[j], b[i])) ++j;
        const picha_image &a0 = a[i], &b0 = b[i];
        if (a0.width == b0.width && a0.height == b0.height && a0.pixel == b0.pixel) {  // lib/image.js:58, 66
            CompareParams p{};
            p.a_tab = dtab + i; p.b_tab = dtab + n + i;
            p.a_stride = a0.stride; p.b_stride = b0.stride;
            p.row_bytes = a0.width * kBytes[a0.pixel]; p.height = a0.height;
            // >= 64 KB per block where the image allows it, and no more than 65535 row blocks
            p.rows_per_block = std::max({1, (int)((65536 + (size_t)p.row_bytes - 1) / (size_t)p.row_bytes), (a0.height + 65534) / 65535});
            p.n_images = j - i;
            p.out = reinterpret_cast<unsigned long long *>(dbase) + 2 * (size_t)i;
            HIP_TRY(launch_image_compare(p, s));
            for (int k = i; k < j; ++k) comparable[k] = 1;
        }
        i = j;
    }
    HIP_TRY(hipMemcpyAsync(hbase, dbase, sums, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const uint64_t *h = reinterpret_cast<const uint64_t *>(hbase);
    for (int k = 0; k < n; ++k) {
        if (sum_abs_diff) sum_abs_diff[k] = comparable[k] ? h[2 * k] : UINT64_MAX;
        if (differing_bytes) differing_bytes[k] = comparable[k] ? h[2 * k + 1] : UINT64_MAX;
    }
    return PICHA_OK;  // the stream is idle: nothing pending on ctx->results
}

int picha_hip_image_copy_device(int n, const picha_image *src, const picha_image *dst, void *stream) {
    if (n < 0 || (n > 0 && (!src || !dst))) return fail(PICHA_ERR_ARGUMENT, "bad batch arguments");
    for (int i = 0; i < n; ++i) {
        int rc = check_src(&src[i]);
        if (rc) return rc;
        rc = check_src(&dst[i]);
        if (rc) return rc;
        if (src[i].pixel != dst[i].pixel)
            return fail(PICHA_ERR_PIXEL_MISMATCH, "can't copy pixels between different pixel types");
    }
    if (n == 0) return PICHA_OK;
    int dev;
    int rc = current_device(&dev);
    if (rc) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int i = 0; i < n; ++i) {  // lib/image.js:92-95
        const size_t rw = (size_t)kBytes[src[i].pixel] * std::min(src[i].width, dst[i].width);
        const int h = std::min(src[i].height, dst[i].height);
        HIP_TRY(hipMemcpy2DAsync(dst[i].data, dst[i].stride, src[i].data, src[i].stride, rw, h, hipMemcpyDeviceToDevice, s));
    }
    return PICHA_OK;
}

// Validation of one host-pointer request (shared with the asynchronous dispatcher, dispatch.cpp).
// Not part of the public header.
int picha_hip_validate_host_op(int kind, const picha_image *src, const picha_image *dst, int filter,
                               float filter_scale, int *device) {  // kind: 0 colorConvert, 1 resize, 2 chained
    int rc = check_src(src);
    if (rc) return rc;
    rc = check_dst(dst);
    if (rc) return rc;
    if (kind != 0) {
        if (filter < 0 || filter >= PICHA_NUM_FILTERS) return fail(PICHA_ERR_INVALID_FILTER, "invalid filter mode");
        if (!(filter_scale > 0)) return fail(PICHA_ERR_INVALID_FILTER_WIDTH, "invalid filter width");
    }
    if (kind == 1 && src->pixel != dst->pixel)
        return fail(PICHA_ERR_PIXEL_MISMATCH, "resize: source and destination pixel formats differ");
    if (kind == 0 && (src->width != dst->width || src->height != dst->height))
        return fail(PICHA_ERR_PIXEL_MISMATCH, "colorConvert: source and destination sizes differ");
    return current_device(device);
}

// ---- host-pointer entry points --------------------------------------------------------------------
int picha_hip_resize(const picha_image *src, const picha_image *dst, int filter, float filter_scale) {
    return run_host(Op::Resize, 1, src, dst, filter, filter_scale, 0, 0, 0);
}
namespace {
// scope of a per-call numerics override; rc != 0: the options were malformed
struct CallOpts {
    int saved, rc = PICHA_OK;
    explicit CallOpts(const picha_call_opts *o) : saved(tl_numerics) {
        if (!o) return;
        if (o->size < (int32_t)sizeof(picha_call_opts) || o->numerics < -1 || o->numerics > 1) { rc = fail(PICHA_ERR_ARGUMENT, "bad picha_call_opts"); return; }
        if (o->numerics >= 0) tl_numerics = o->numerics;
    }
    ~CallOpts() { tl_numerics = saved; }
};
}  // namespace
int picha_hip_resize_ex(const picha_image *src, const picha_image *dst, int filter, float filter_scale, const picha_call_opts *opts) {
    CallOpts scope(opts);
    if (scope.rc) return scope.rc;
    return run_host(Op::Resize, 1, src, dst, filter, filter_scale, 0, 0, 0);
}
int picha_hip_resize_batch_ex(int n, const picha_image *src, const picha_image *dst, int filter, float filter_scale, const picha_call_opts *opts) {
    CallOpts scope(opts);
    if (scope.rc) return scope.rc;
    return run_host(Op::Resize, n, src, dst, filter, filter_scale, 0, 0, 0);
}
int picha_hip_resize_device_ex(int n, const picha_image *src, const picha_image *dst, int filter, float filter_scale, const picha_call_opts *opts,
                               void *stream) {
    CallOpts scope(opts);
    if (scope.rc) return scope.rc;
    return picha_hip_resize_device(n, src, dst, filter, filter_scale, stream);
}
int picha_hip_color_convert(const picha_image *src, const picha_image *dst, float r, float g, float b) {
    return run_host(Op::Convert, 1, src, dst, 0, 0, r, g, b);
}
int picha_hip_resize_batch(int n, const picha_image *src, const picha_image *dst, int filter, float filter_scale) {
    return run_host(Op::Resize, n, src, dst, filter, filter_scale, 0, 0, 0);
}
int picha_hip_color_convert_batch(int n, const picha_image *src, const picha_image *dst, float r, float g, float b) {
    return run_host(Op::Convert, n, src, dst, 0, 0, r, g, b);
}
int picha_hip_convert_resize(const picha_image *src, const picha_image *dst, int order, float r, float g, float b,
                             int filter, float filter_scale) {
    return run_host(Op::Chain, 1, src, dst, filter, filter_scale, r, g, b, order);
}
int picha_hip_convert_resize_batch(int n, const picha_image *src, const picha_image *dst, int order, float r, float g,
                                   float b, int filter, float filter_scale) {
    return run_host(Op::Chain, n, src, dst, filter, filter_scale, r, g, b, order);
}

// ---- device memory helpers --------------------------------------------------------------------------
int picha_hip_malloc(void **dev_ptr, size_t bytes) {
    if (!dev_ptr) return fail(PICHA_ERR_ARGUMENT, "null out pointer");
    int dev;
    int rc = current_device(&dev);
    if (rc) return rc;
    HIP_TRY(hipMalloc(dev_ptr, bytes));
    return PICHA_OK;
}
int picha_hip_free(void *dev_ptr) {
    HIP_TRY(hipFree(dev_ptr));
    return PICHA_OK;
}
int picha_hip_malloc_placed(void **dev_ptr, size_t bytes, int n, const picha_image *src, picha_image *dst, int filter, float filter_scale, int candidates, float *ms_out) {
    if (!dev_ptr || !src || !dst || n <= 0 || bytes == 0 || candidates < 1 || candidates > 16) return fail(PICHA_ERR_ARGUMENT, "bad arguments");
    *dev_ptr = nullptr;
    int dev;
    int rc = current_device(&dev);
    if (rc) return rc;
    std::vector<void *> base((size_t)candidates, nullptr);
    std::vector<size_t> off((size_t)n);
    for (int i = 0; i < n; ++i) {
        off[(size_t)i] = (size_t)(uintptr_t)dst[i].data;
        const size_t need = off[(size_t)i] + (dst[i].height > 0 ? image_span(dst[i]) : 0);
        if (need > bytes) return fail(PICHA_ERR_ARGUMENT, "destination image outside the arena");
    }
    auto free_all = [&](int keep) { for (int c = 0; c < candidates; ++c) if (c != keep && base[(size_t)c]) (void)hipFree(base[(size_t)c]); };
    for (int c = 0; c < candidates; ++c)   // all alive at once: each lands somewhere else
        if (hipMalloc(&base[(size_t)c], bytes) != hipSuccess) { free_all(-1); return fail(PICHA_ERR_HIP, "arena allocation failed"); }
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); if (s) (void)hipStreamDestroy(s); };
    if (hipStreamCreate(&s) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { cleanup(); free_all(-1); return fail(PICHA_ERR_HIP, "stream / event creation failed"); }
    std::vector<picha_image> d(dst, dst + n);
    std::vector<float> best((size_t)candidates, 3.4e38f);
    const int launches = 6;
    for (int round = 0; round < 3 && rc == PICHA_OK; ++round)   // round 0 builds the tables and raises the clocks; two interleaved timed rounds decide
        for (int c = 0; c < candidates && rc == PICHA_OK; ++c) {
            for (int i = 0; i < n; ++i) d[(size_t)i].data = static_cast<uint8_t *>(base[(size_t)c]) + off[(size_t)i];
            rc = picha_hip_resize_device(n, src, d.data(), filter, filter_scale, s);
            if (rc || round == 0) continue;
            (void)hipEventRecord(e0, s);
            for (int k = 0; k < launches && rc == PICHA_OK; ++k) rc = picha_hip_resize_device(n, src, d.data(), filter, filter_scale, s);
            (void)hipEventRecord(e1, s);
            if (hipEventSynchronize(e1) != hipSuccess) { rc = fail(PICHA_ERR_HIP, "placement timing failed"); break; }
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best[(size_t)c] = std::min(best[(size_t)c], ms / launches);
        }
    (void)hipStreamSynchronize(s);
    cleanup();
    if (rc) { free_all(-1); return rc; }
    int keep = 0;
    for (int c = 1; c < candidates; ++c) if (best[(size_t)c] < best[(size_t)keep]) keep = c;
    free_all(keep);
    for (int i = 0; i < n; ++i) dst[i].data = static_cast<uint8_t *>(base[(size_t)keep]) + off[(size_t)i];
    if (ms_out) for (int c = 0; c < candidates; ++c) ms_out[c] = candidates == 1 ? 0.0f : best[(size_t)c];
    *dev_ptr = base[(size_t)keep];
    return PICHA_OK;
}
int picha_hip_memcpy_h2d(void *d, const void *h, size_t bytes, void *stream) {
    HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    return PICHA_OK;
}
int picha_hip_memcpy_d2h(void *h, const void *d, size_t bytes, void *stream) {
    HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    return PICHA_OK;
}
int picha_hip_stream_synchronize(void *stream) {
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return PICHA_OK;
}
int picha_hip_fill_random(void *dev_ptr, size_t bytes, uint64_t seed, int sample_bytes, void *stream) {
    if (sample_bytes != 1 && sample_bytes != 2) return fail(PICHA_ERR_ARGUMENT, "sample_bytes must be 1 or 2");
    int dev;
    int rc = current_device(&dev);
    if (rc) return rc;
    HIP_TRY(launch_fill_random(dev_ptr, bytes, seed, sample_bytes, static_cast<hipStream_t>(stream)));
    return PICHA_OK;
}
void picha_fill_random_host(void *host_ptr, size_t bytes, uint64_t seed, int sample_bytes) {
    fill_random_host(host_ptr, bytes, seed, sample_bytes);
}

}  // extern "C"
