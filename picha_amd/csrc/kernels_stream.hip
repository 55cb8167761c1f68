// kernels_stream.hip — the streaming resize kernels of ONE pixel format (PICHA_STREAM_PIXEL), compiled once per
// format (see Makefile) so that the curated instantiation lists of kernels_impl.inc build in parallel.
#define PICHA_SPAN_ONLY 1
#define PICHA_STREAM_TU 1
#include "kernels_impl.inc"

#ifndef PICHA_STREAM_PIXEL
#error "compile with -DPICHA_STREAM_PIXEL=0..7"
#endif

namespace picha_hip {
#define PICHA_CAT2(a, b) a##b
#define PICHA_CAT(a, b) PICHA_CAT2(a, b)
#define PICHA_STREAM_LIST PICHA_CAT(PICHA_STREAM_LIST_, PICHA_STREAM_PIXEL)

PICHA_PROBE_HOST(PICHA_CAT(picha_probe_stream_p, PICHA_STREAM_PIXEL))

bool PICHA_CAT(stream_supported_pixel, PICHA_STREAM_PIXEL)(int per_lane, int sx, int tx, int sy, int ty, int rot) {
#define X(CH, S16, P, SX, TX, SY, TY, ROT) if (per_lane == P && sx == SX && tx == TX && sy == SY && ty == TY && (rot != 0) == ROT) return true;
    PICHA_STREAM_LIST(X)
#undef X
    return false;
}

hipError_t PICHA_CAT(launch_stream_pixel, PICHA_STREAM_PIXEL)(const StreamParams &p, hipStream_t s, const char **name) {
#define X(CH, S16, P, SX, TX, SY, TY, ROT) \
    if (p.x_per_lane == P && p.x_step == SX && p.x_taps == TX && p.y_step == SY && p.y_taps == TY && (p.rot != 0) == ROT) \
        return launch_stream_t<CH, S16, P, SX, TX, SY, TY, ROT>(p, s, name);
    PICHA_STREAM_LIST(X)
#undef X
    return hipErrorInvalidValue;
}
}  // namespace picha_hip
