// kernels.hip — launch interface of the gfx950 kernels.  The device code lives in kernels_impl.inc;
// the single-output-per-lane span kernels (hundreds of instantiations) are compiled per pixel format
// in kernels_span.hip so the build parallelises.
#include <array>
#include <map>
#include <mutex>
#include "kernels_impl.inc"

namespace picha_hip {

// defined in kernels_span.hip, one translation unit per pixel format
#define PICHA_DECL_SPAN(N)                                                                           \
    hipError_t launch_span_p1_pixel##N(const FusedParams &p, hipStream_t s, const char **name);      \
    hipError_t launch_acc_p1_pixel##N(const FusedParams &p, hipStream_t s, const char **name);
PICHA_DECL_SPAN(0) PICHA_DECL_SPAN(1) PICHA_DECL_SPAN(2) PICHA_DECL_SPAN(3)
PICHA_DECL_SPAN(4) PICHA_DECL_SPAN(5) PICHA_DECL_SPAN(6) PICHA_DECL_SPAN(7)
#undef PICHA_DECL_SPAN


// defined in kernels_stream.hip, one translation unit per pixel format
#define PICHA_DECL_STREAM(N)                                                                      \
    bool stream_supported_pixel##N(int per_lane, int sx, int tx, int sy, int ty, int rot);        \
    hipError_t launch_stream_pixel##N(const StreamParams &p, hipStream_t s, const char **name);
PICHA_DECL_STREAM(0) PICHA_DECL_STREAM(1) PICHA_DECL_STREAM(2) PICHA_DECL_STREAM(3)
PICHA_DECL_STREAM(4) PICHA_DECL_STREAM(5) PICHA_DECL_STREAM(6) PICHA_DECL_STREAM(7)
#undef PICHA_DECL_STREAM

// defined in kernels_sym.hip, one translation unit per pixel format
#define PICHA_DECL_SYM(N)                                                                \
    bool sym_supported_pixel##N(int per_lane, int sx, int tx, int sy, int ty, int xch);   \
    bool sym_epi_supported_pixel##N(int per_lane, int sx, int tx, int sy, int ty);        \
    hipError_t launch_sym_pixel##N(const StreamParams &p, hipStream_t s, const char **name); \
    int sym_resident_pixel##N(const StreamParams &p);
PICHA_DECL_SYM(0) PICHA_DECL_SYM(1) PICHA_DECL_SYM(2) PICHA_DECL_SYM(3)
PICHA_DECL_SYM(4) PICHA_DECL_SYM(5) PICHA_DECL_SYM(6) PICHA_DECL_SYM(7)
#undef PICHA_DECL_SYM
// defined in kernels_run.hip, one translation unit per pixel format
#define PICHA_DECL_RUN(N)                                                                \
    bool run_supported_pixel##N(int q, int px, int tx, int per_lane, int nacc, int rc, int xhl, int ws);   \
    hipError_t launch_run_pixel##N(const RunParams &p, hipStream_t s, const char **name); \
    hipError_t launch_edges_pixel##N(const RunParams &p, hipStream_t s); \
    bool up_supported_pixel##N(int q, int px, int tx, int per_lane, int rc, int ring, int ty); \
    bool up_cv_supported_pixel##N(int src_pixel, int q, int px, int tx, int per_lane, int rc, int ring, int ty); \
    hipError_t launch_up_pixel##N(const RunParams &p, hipStream_t s, const char **name);
PICHA_DECL_RUN(0) PICHA_DECL_RUN(1) PICHA_DECL_RUN(2) PICHA_DECL_RUN(3)
PICHA_DECL_RUN(4) PICHA_DECL_RUN(5) PICHA_DECL_RUN(6) PICHA_DECL_RUN(7)
#undef PICHA_DECL_RUN
bool run_kernel_supported(int pixel, int period, int period_px, int taps, int per_lane, int nacc, int off_ceil, int xhl, int wsgpr) {
    switch (pixel) {
#define PICHA_CASE(N) case N: return run_supported_pixel##N(period, period_px, taps, per_lane, nacc, off_ceil, xhl, wsgpr);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5) PICHA_CASE(6) PICHA_CASE(7)
#undef PICHA_CASE
    }
    return false;
}
hipError_t launch_resize_run_kernel(int pixel, const RunParams &p, hipStream_t s, const char **name) {
    switch (pixel) {
#define PICHA_CASE(N) case N: return launch_run_pixel##N(p, s, name);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5) PICHA_CASE(6) PICHA_CASE(7)
#undef PICHA_CASE
    }
    return hipErrorInvalidValue;
}
bool up_kernel_supported(int pixel, int period, int period_px, int taps, int per_lane, int off_ceil, int ring_rows, int y_taps) {
    switch (pixel) {
#define PICHA_CASE(N) case N: return up_supported_pixel##N(period, period_px, taps, per_lane, off_ceil, ring_rows, y_taps);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5) PICHA_CASE(6) PICHA_CASE(7)
#undef PICHA_CASE
    }
    return false;
}
bool up_cv_kernel_supported(int pixel, int src_pixel, int period, int period_px, int taps, int per_lane, int off_ceil, int ring_rows, int y_taps) {
    switch (pixel) {
#define PICHA_CASE(N) case N: return up_cv_supported_pixel##N(src_pixel, period, period_px, taps, per_lane, off_ceil, ring_rows, y_taps);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5) PICHA_CASE(6) PICHA_CASE(7)
#undef PICHA_CASE
    }
    return false;
}
hipError_t launch_resize_up_kernel(int pixel, const RunParams &p, hipStream_t s, const char **name) {
    switch (pixel) {
#define PICHA_CASE(N) case N: return launch_up_pixel##N(p, s, name);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5) PICHA_CASE(6) PICHA_CASE(7)
#undef PICHA_CASE
    }
    return hipErrorInvalidValue;
}
hipError_t launch_resize_edges_kernel(int pixel, const RunParams &p, hipStream_t s) {
    switch (pixel) {
#define PICHA_CASE(N) case N: return launch_edges_pixel##N(p, s);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5) PICHA_CASE(6) PICHA_CASE(7)
#undef PICHA_CASE
    }
    return hipErrorInvalidValue;
}
int stream_kernel_resident_blocks(int pixel, const StreamParams &p) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const std::array<int, 14> key = {dev, pixel, p.x_per_lane, p.x_step, p.x_taps, p.y_step, p.y_taps, p.fma, p.rot != 0, -77, 0, 0, 0, 0};
    static std::mutex mu;
    static std::map<std::array<int, 14>, int> cache;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    StreamParams q = p;
    int n = 0;
    q.resident_query = &n;
    if (launch_resize_stream_kernel(pixel, q, nullptr, nullptr) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(mu);
    cache[key] = n;
    return n;
}
int run_kernel_resident_blocks(int pixel, const RunParams &p) {
    // (r05) the answer depends on the kernel instantiation, the workgroup width and the device only: asked once per key (the planner asks on every call of the
    // shapes whose chunk count is a whole round of resident workgroups; the occupancy query costs tens of microseconds of host time)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const std::array<int, 14> key = {dev, pixel, p.period, p.period_px, p.x_taps, p.x_per_lane, p.off_ceil, p.nacc, p.fma, p.xch ? (p.xch == 2 ? -2 : -p.x_base) : -1, p.wsgpr,
                                     p.phase_shift > 0, p.block_waves, p.stash != nullptr};
    static std::mutex mu;
    static std::map<std::array<int, 14>, int> cache;
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(key);
        if (it != cache.end()) return it->second;
    }
    RunParams q = p;
    int n = 0;
    q.resident_query = &n;
    if (launch_resize_run_kernel(pixel, q, nullptr, nullptr) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(mu);
    cache[key] = n;
    return n;
}

bool sym_epi_supported(int pixel, int per_lane, int sx, int tx, int sy, int ty) {
    switch (pixel) {
#define PICHA_CASE(N) case N: return sym_epi_supported_pixel##N(per_lane, sx, tx, sy, ty);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5) PICHA_CASE(6) PICHA_CASE(7)
#undef PICHA_CASE
    }
    return false;
}
bool sym_kernel_supported(int pixel, int per_lane, int sx, int tx, int sy, int ty, int xch) {
    switch (pixel) {
#define PICHA_CASE(N) case N: return sym_supported_pixel##N(per_lane, sx, tx, sy, ty, xch);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5) PICHA_CASE(6) PICHA_CASE(7)
#undef PICHA_CASE
    }
    return false;
}
int sym_kernel_resident_blocks(int pixel, const StreamParams &p) {
    switch (pixel) {
#define PICHA_CASE(N) case N: return sym_resident_pixel##N(p);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5) PICHA_CASE(6) PICHA_CASE(7)
#undef PICHA_CASE
    }
    return 0;
}
hipError_t launch_resize_sym_kernel(int pixel, const StreamParams &p, hipStream_t s, const char **name) {
    switch (pixel) {
#define PICHA_CASE(N) case N: return launch_sym_pixel##N(p, s, name);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5) PICHA_CASE(6) PICHA_CASE(7)
#undef PICHA_CASE
    }
    return hipErrorInvalidValue;
}

// defined in kernels_rows.hip, one translation unit per variant
#define PICHA_DECL_ROWS(N)                                                    \
    bool rows_supported_variant##N(int per_lane, int step, int taps);          \
    hipError_t launch_rows_variant##N(const RowsParams &p, hipStream_t s, const char **name);
PICHA_DECL_ROWS(0) PICHA_DECL_ROWS(1) PICHA_DECL_ROWS(2) PICHA_DECL_ROWS(3) PICHA_DECL_ROWS(4) PICHA_DECL_ROWS(5)
PICHA_DECL_ROWS(6) PICHA_DECL_ROWS(7) PICHA_DECL_ROWS(8) PICHA_DECL_ROWS(9) PICHA_DECL_ROWS(10) PICHA_DECL_ROWS(11)
#undef PICHA_DECL_ROWS
// pixel ids: 0 rgb 1 rgba 2 grey 3 greya 4 r16 5 r16g16 6 r16g16b16 7 r16g16b16a16
int rows_variant(int sp, int dp) {
    if (sp == dp) return sp == 2 ? 0 : sp == 1 ? 1 : sp == 3 ? 4 : sp == 7 ? 5 : -1;
    if (dp == 2) return sp == 0 ? 2 : sp == 1 ? 3 : sp == 6 ? 10 : sp == 7 ? 11 : -1;   // -> grey
    if (dp == 3) return sp == 0 ? 6 : sp == 1 ? 7 : -1;                                  // -> greya
    if (dp == 4) return sp == 6 ? 8 : sp == 7 ? 9 : -1;                                  // -> r16
    return -1;
}
// source pixels per stage piece and window alignment: kernels_rows.hip, rows_stage_px / 8-byte LDS reads of 1-float pixels
int rows_stage_pixels(int v) {
    static const int px[12] = {4, 1, 4, 4, 2, 1, 4, 2, 4, 4, 4, 4};
    return v >= 0 && v < 12 ? px[v] : 1;
}
int rows_window_align(int v) { return (v == 0 || v == 2 || v == 3 || v >= 8) ? 2 : 1; }
bool rows_kernel_supported(int v, int per_lane, int step, int taps) {
    switch (v) {
#define PICHA_CASE(N) case N: return rows_supported_variant##N(per_lane, step, taps);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5)
    PICHA_CASE(6) PICHA_CASE(7) PICHA_CASE(8) PICHA_CASE(9) PICHA_CASE(10) PICHA_CASE(11)
#undef PICHA_CASE
    }
    return false;
}
hipError_t launch_resize_rows_kernel(int v, const RowsParams &p, hipStream_t s, const char **name) {
    switch (v) {
#define PICHA_CASE(N) case N: return launch_rows_variant##N(p, s, name);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5)
    PICHA_CASE(6) PICHA_CASE(7) PICHA_CASE(8) PICHA_CASE(9) PICHA_CASE(10) PICHA_CASE(11)
#undef PICHA_CASE
    }
    return hipErrorInvalidValue;
}

bool stream_kernel_supported(int pixel, int per_lane, int sx, int tx, int sy, int ty, int rot) {
    switch (pixel) {
#define PICHA_CASE(N) case N: return stream_supported_pixel##N(per_lane, sx, tx, sy, ty, rot);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5) PICHA_CASE(6) PICHA_CASE(7)
#undef PICHA_CASE
    }
    return false;
}
hipError_t launch_resize_stream_kernel(int pixel, const StreamParams &p, hipStream_t s, const char **name) {
    if (p.dpp) return pixel == 1 ? launch_stream_dpp(p, s, name) : hipErrorInvalidValue;
    switch (pixel) {
#define PICHA_CASE(N) case N: return launch_stream_pixel##N(p, s, name);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5) PICHA_CASE(6) PICHA_CASE(7)
#undef PICHA_CASE
    }
    return hipErrorInvalidValue;
}
bool stream_dpp_supported(int pixel, int sx, int tx, int sy, int ty) {
    return pixel == 1 && sx == 2 && tx == 5 && sy == 2 && ty == 5;
}


// ---------------------------------------------------------------------------------------------
// Launch interface
// ---------------------------------------------------------------------------------------------
hipError_t launch_color_convert(int sp, int dp, const ConvertParams &p, hipStream_t s, const char **name) {
    if (name) *name = g_kernel_name;  // filled by the launcher that runs
    switch (sp) {
    case 0: return launch_cc_src<0>(dp, p, s);
    case 1: return launch_cc_src<1>(dp, p, s);
    case 2: return launch_cc_src<2>(dp, p, s);
    case 3: return launch_cc_src<3>(dp, p, s);
    case 4: return launch_cc_src<4>(dp, p, s);
    case 5: return launch_cc_src<5>(dp, p, s);
    case 6: return launch_cc_src<6>(dp, p, s);
    case 7: return launch_cc_src<7>(dp, p, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_resize_twopass(int pixel, const TwoPassParams &p, hipStream_t s) {
    switch (pixel) {
    case 0: return launch_twopass_t<3, false>(p, s);
    case 1: return launch_twopass_t<4, false>(p, s);
    case 2: return launch_twopass_t<1, false>(p, s);
    case 3: return launch_twopass_t<2, false>(p, s);
    case 4: return launch_twopass_t<1, true>(p, s);
    case 5: return launch_twopass_t<2, true>(p, s);
    case 6: return launch_twopass_t<3, true>(p, s);
    case 7: return launch_twopass_t<4, true>(p, s);
    }
    return hipErrorInvalidValue;
}

int fused_y_pad(int y_taps) { return y_taps <= 8 ? 8 : (y_taps <= 16 ? 16 : 32); }

// acc: the accumulator kernel has a finer bucket list above 12 taps (a long window's zero-padded slots are pure
// VALU waste in a VALU-bound kernel; the span kernel keeps the coarse list, it has three y variants per bucket)
int fused_x_bucket(int x_taps, int pixel, bool acc) {
    static const int coarse[] = {1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 24, 32, 36};
    static const int fine[] = {4, 5, 6, 7, 8, 10, 12, 14, 16, 18, 20, 22, 24, 28, 32, 36};
    const int *buckets = acc ? fine : coarse;
    const int nb = acc ? int(sizeof fine / sizeof fine[0]) : int(sizeof coarse / sizeof coarse[0]);
    const int bpp = pixel_channels(pixel) * (pixel_is16(pixel) ? 2 : 1);
    if (bpp % 4 != 0) {  // sub-dword pixels: the rotated fetch takes the window as it is (see span_rotate)
        for (int i = 0; i < nb; ++i)
            if (x_taps <= buckets[i]) { if (span_rotate(bpp, buckets[i])) return buckets[i]; break; }
    }
    // aligned fetch: room for moving the window start down to a 4-byte boundary, and a whole-dword window
    const int room = bpp % 4 == 0 ? 0 : (bpp % 2 ? 3 : 1);
    for (int i = 0; i < nb; ++i) {
        const int b = buckets[i];
        if (x_taps + room <= b && (b * bpp) % 4 == 0 && !span_rotate(bpp, b)) return b;
    }
    return 0;
}
bool fused_x_bytespan(int x_taps_bucket, int pixel) {  // aligned fetch: the host must align the windows
    const int bpp = pixel_channels(pixel) * (pixel_is16(pixel) ? 2 : 1);
    return bpp % 4 != 0 && !span_rotate(bpp, x_taps_bucket);
}

bool acc_p2_supported(int pixel, int step, int taps, int nacc) {
    return pixel >= 0 && pixel <= 7 && (nacc == 4 || nacc == 6) && acc_p2_shape(step, taps);
}
bool acc_epi_supported(int pixel, int x_taps_bucket, int nacc) {
    const int t = x_taps_bucket;
    return acc_supported(pixel, t, nacc) && (t == 4 || t == 5 || t == 7 || t == 8 || t == 10 || t == 12 || t == 16);
}
bool acc_up_supported(int pixel, int per_lane, int taps, int nacc) {
    switch (pixel) {
    case 0: return acc_up_shape_ok<3, false>(per_lane, taps, nacc);
    case 1: return acc_up_shape_ok<4, false>(per_lane, taps, nacc);
    case 2: return acc_up_shape_ok<1, false>(per_lane, taps, nacc);
    case 3: return acc_up_shape_ok<2, false>(per_lane, taps, nacc);
    case 4: return acc_up_shape_ok<1, true>(per_lane, taps, nacc);
    case 5: return acc_up_shape_ok<2, true>(per_lane, taps, nacc);
    case 6: return acc_up_shape_ok<3, true>(per_lane, taps, nacc);
    case 7: return acc_up_shape_ok<4, true>(per_lane, taps, nacc);
    }
    return false;
}
bool acc_xch_supported(int pixel, int per_lane, int step, int taps, int nacc) {
    const int bpp = pixel_channels(pixel) * (pixel_is16(pixel) ? 2 : 1);
    return pixel >= 0 && pixel <= 7 && bpp % 4 == 0 && (nacc == 4 || nacc == 6) && acc_xch_shape(per_lane, step, taps);
}
PICHA_PROBE_HOST(picha_probe_kernels)

bool acc_supported(int pixel, int x_taps_bucket, int nacc) {
    if (pixel < 0 || pixel > 7 || (nacc != 4 && nacc != 6)) return false;
    if (x_taps_bucket > 36) {  // resize_acc_long_kernel: multiples of half a segment
        const int half = acc_long_segment(pixel_channels(pixel) * (pixel_is16(pixel) ? 2 : 1)) / 2;
        return x_taps_bucket % half == 0 && x_taps_bucket <= kAccLongMaxTaps + half;
    }
    switch (x_taps_bucket) {
    case 4: case 5: case 6: case 7: case 8: case 10: case 12: case 14: case 16: case 18: case 20: case 22: case 24: case 28: case 32: case 36:
        return true;
    }
    return false;
}
hipError_t launch_resize_acc(int pixel, const FusedParams &p, hipStream_t s, const char **name) {
    switch (pixel) {
#define PICHA_CASE(N) case N: return launch_acc_p1_pixel##N(p, s, name);
    PICHA_CASE(0) PICHA_CASE(1) PICHA_CASE(2) PICHA_CASE(3) PICHA_CASE(4) PICHA_CASE(5) PICHA_CASE(6) PICHA_CASE(7)
#undef PICHA_CASE
    }
    return hipErrorInvalidValue;
}

int acc_kernel_resident_blocks(int pixel, const FusedParams &p) {
    FusedParams q = p;
    int n = 0;
    q.resident_query = &n;
    if (launch_resize_acc(pixel, q, nullptr, nullptr) != hipSuccess) return 0;
    return n;
}

bool fused_multi_supported(int pixel, int per_lane, int step, int x_taps) {
    if ((pixel == 0 || pixel == 2 || pixel == 3) && per_lane == 4 && step == 0) return x_taps >= 5 && x_taps <= 7;  // upscales, 4 outputs per lane
    if (pixel != 1) return false;
    const int key = per_lane * 10000 + step * 100 + x_taps;
    return key == 20205 || key == 20207 || key == 40205 || key == 20103 || key == 20104;
}

bool fused_supported(int pixel, int x_taps, int y_taps, int ring_slots, int per_lane) {
    if (pixel < 0 || pixel > 7) return false;
    if (x_taps < 1 || x_taps > 36) return false;  // x_taps is already a bucket (fused_x_bucket) or an exact multi-output tap count
    if (y_taps < 1 || y_taps > 32) return false;
    const int ch = pixel_channels(pixel);
    const size_t slot = (ch == 1 ? 4 : ch == 2 ? 8 : 16);
    return (size_t)4 * ring_slots * kWave * per_lane * slot <= 160 * 1024;
}

hipError_t launch_resize_fused(int pixel, const FusedParams &p, hipStream_t s, const char **name) {
    if (p.x_per_lane > 1) return launch_span_multi(pixel, p, s, name);
    switch (pixel) {
    case 0: return launch_span_p1_pixel0(p, s, name);
    case 1: return launch_span_p1_pixel1(p, s, name);
    case 2: return launch_span_p1_pixel2(p, s, name);
    case 3: return launch_span_p1_pixel3(p, s, name);
    case 4: return launch_span_p1_pixel4(p, s, name);
    case 5: return launch_span_p1_pixel5(p, s, name);
    case 6: return launch_span_p1_pixel6(p, s, name);
    case 7: return launch_span_p1_pixel7(p, s, name);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_fill_random(void *dev, size_t bytes, uint64_t seed, int sample_bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    size_t blocks = std::min<size_t>((bytes / 4 + 255) / 256, 8192);
    if (blocks == 0) blocks = 1;
    fill_random_kernel<<<dim3((unsigned)blocks), 256, 0, s>>>(static_cast<uint8_t *>(dev), bytes, seed, sample_bytes);
    return hipGetLastError();
}

hipError_t launch_image_compare(const CompareParams &p, hipStream_t s) {
    if (p.n_images <= 0 || p.height <= 0) return hipSuccess;
    const int row_blocks = (p.height + p.rows_per_block - 1) / p.rows_per_block;
    int done = 0;
    while (done < p.n_images) {  // grid.z <= 65535
        CompareParams q = p;
        q.n_images = std::min(p.n_images - done, 65535);
        q.a_tab = p.a_tab + done; q.b_tab = p.b_tab + done; q.out = p.out + 2 * (size_t)done;
        image_compare_kernel<<<dim3(1, row_blocks, q.n_images), 256, 0, s>>>(q);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        done += q.n_images;
    }
    return hipSuccess;
}

// host twin of the generator
void fill_random_host(void *host, size_t bytes, uint64_t seed, int sample_bytes) {
    uint8_t *d = static_cast<uint8_t *>(host);
    for (size_t k = 0; k < bytes; ++k) {
        if (sample_bytes == 1) d[k] = uint8_t(splitmix64(seed ^ k) >> 56);
        else d[k] = uint8_t((splitmix64(seed ^ (k / 2)) >> 48) >> (8 * (k & 1)));
    }
}

}  // namespace picha_hip
