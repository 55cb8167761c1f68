// TEST INFRASTRUCTURE: drives picha_amd/csrc/dispatch.cpp (compiled against tests/sanitize/mock_hip) from several threads
// under TSan / ASan+UBSan.  The "device" operations are stand-ins that write a checkable function of the source into the
// destination payload; what is under test is the dispatcher: queueing, coalescing of compatible requests, the three
// staging slots, the copy pool, completion callbacks, drain, round-robin over virtual devices.
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/picha_hip.h"

static std::atomic<long> g_async_devices{0}, g_virtual{0};
static const int kB[8] = {3, 4, 1, 2, 2, 4, 6, 8};

extern "C" {
long picha_hip_get_option(const char *key) {
    if (!strcmp(key, "async_devices")) return g_async_devices;
    if (!strcmp(key, "async_virtual_devices")) return g_virtual;
    return 0;
}
int picha_hip_validate_host_op(int, const picha_image *src, const picha_image *dst, int, float, int *device) {
    if (!src || !dst || !src->data || !dst->data || dst->width <= 0) return PICHA_ERR_INVALID_IMAGE;
    *device = 0;
    return PICHA_OK;
}
static int fake_op(int n, const picha_image *src, const picha_image *dst, unsigned char tag) {
    for (int i = 0; i < n; ++i) {
        const unsigned char first = *static_cast<const unsigned char *>(src[i].data);
        for (int y = 0; y < dst[i].height; ++y)
            memset(static_cast<unsigned char *>(dst[i].data) + (size_t)y * dst[i].stride, first ^ tag, (size_t)dst[i].width * kB[dst[i].pixel]);
    }
    return PICHA_OK;
}
int picha_hip_resize_device(int n, const picha_image *s, const picha_image *d, int, float, void *) { return fake_op(n, s, d, 0x11); }
int picha_hip_resize_device_ex(int n, const picha_image *s, const picha_image *d, int, float, const picha_call_opts *, void *) { return fake_op(n, s, d, 0x11); }
int picha_hip_color_convert_device(int n, const picha_image *s, const picha_image *d, float, float, float, void *) { return fake_op(n, s, d, 0x22); }
int picha_hip_convert_resize_device(int n, const picha_image *s, const picha_image *d, int, float, float, float, int, float, void *) { return fake_op(n, s, d, 0x33); }
const char *picha_hip_last_error(void) { return "mock"; }
const char *picha_hip_strerror(int) { return "mock"; }
}

static std::atomic<int> g_done{0}, g_bad{0};
struct Job { std::vector<unsigned char> src, dst; picha_image s, d; unsigned char want; int stride; };
static void on_done(void *user, int status, const char *) {
    Job *j = static_cast<Job *>(user);
    if (status != 0) ++g_bad;
    // payload bytes carry the result, row padding must be untouched (0xEE)
    const int rw = j->d.width * kB[j->d.pixel];
    for (int y = 0; y < j->d.height; ++y)
        for (int x = 0; x < j->stride; ++x) {
            const unsigned char v = j->dst[(size_t)y * j->stride + x];
            if (v != (x < rw ? j->want : 0xEE)) { ++g_bad; return; }
        }
    ++g_done;
}

int main() {
    const int kThreads = 6, kPer = 150;
    for (int round = 0; round < 2; ++round) {
        g_async_devices = round == 0 ? 0 : 3;   // second round: three virtual devices on the one mock device
        g_virtual = round;
        std::vector<std::vector<Job>> jobs(kThreads, std::vector<Job>(kPer));
        std::vector<std::thread> ts;
        for (int t = 0; t < kThreads; ++t)
            ts.emplace_back([&, t] {
                for (int k = 0; k < kPer; ++k) {
                    Job &j = jobs[t][k];
                    const int w = 5 + (k * 7 + t) % 90, h = 1 + (k + t) % 17, pixel = (k + t) % 8, dw = 3 + k % 40, dh = 1 + k % 9;
                    j.src.assign((size_t)h * ((w * kB[pixel] + 3) & ~3), (unsigned char)(k * 13 + t));
                    j.stride = ((dw * kB[pixel] + 3) & ~3) + (k % 3 == 0 ? 4 : 0);
                    j.dst.assign((size_t)dh * j.stride, 0xEE);
                    j.s = {j.src.data(), (w * kB[pixel] + 3) & ~3, w, h, pixel};
                    j.d = {j.dst.data(), j.stride, dw, dh, pixel};
                    const int kind = k % 3;
                    j.want = (unsigned char)(k * 13 + t) ^ (kind == 0 ? 0x11 : kind == 1 ? 0x22 : 0x33);
                    int rc = kind == 0   ? picha_hip_resize_async(&j.s, &j.d, k % 2, 1.0f, on_done, &j)
                             : kind == 1 ? picha_hip_color_convert_async(&j.s, &j.d, 0.3f, 0.6f, 0.1f, on_done, &j)
                                         : picha_hip_convert_resize_async(&j.s, &j.d, PICHA_CONVERT_THEN_RESIZE, 0.3f, 0.6f, 0.1f, 0, 0.7f, on_done, &j);
                    if (rc != PICHA_OK) ++g_bad;
                }
            });
        for (auto &t : ts) t.join();
        picha_hip_async_drain();
        if (g_done != (round + 1) * kThreads * kPer || g_bad != 0) { printf("FAIL round %d: done %d bad %d\n", round, g_done.load(), g_bad.load()); return 1; }
    }
    uint64_t b = 0, r = 0;
    picha_hip_async_stats(&b, &r);
    const int nd = picha_hip_async_dispatcher_stats(0, &b, &r);
    printf("ok: %d requests, %d dispatchers\n", g_done.load(), nd);
    return nd >= 3 ? 0 : 1;
}
