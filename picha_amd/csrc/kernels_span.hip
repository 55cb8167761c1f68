// kernels_span.hip — single-output-per-lane span kernels for ONE pixel format (PICHA_SPAN_PIXEL),
// compiled once per format (see Makefile) so the ~80 instantiations per format build in parallel.
#define PICHA_SPAN_ONLY 1
#include "kernels_impl.inc"

#ifndef PICHA_SPAN_PIXEL
#error "compile with -DPICHA_SPAN_PIXEL=0..7"
#endif

namespace picha_hip {
#define PICHA_CAT2(a, b) a##b
#define PICHA_CAT(a, b) PICHA_CAT2(a, b)
hipError_t PICHA_CAT(launch_span_p1_pixel, PICHA_SPAN_PIXEL)(const FusedParams &p, hipStream_t s, const char **name) {
    return launch_span_p1<pixel_channels(PICHA_SPAN_PIXEL), pixel_is16(PICHA_SPAN_PIXEL)>(p, s, name);
}
PICHA_PROBE_HOST(PICHA_CAT(picha_probe_span_p, PICHA_SPAN_PIXEL))

hipError_t PICHA_CAT(launch_acc_p1_pixel, PICHA_SPAN_PIXEL)(const FusedParams &p, hipStream_t s, const char **name) {
    return launch_acc_p1<pixel_channels(PICHA_SPAN_PIXEL), pixel_is16(PICHA_SPAN_PIXEL)>(p, s, name);
}
}  // namespace picha_hip
