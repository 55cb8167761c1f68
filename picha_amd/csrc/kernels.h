// kernels.h — launch interface between the C ABI (capi.cpp) and the gfx950 kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace picha_hip {

// ---- colour convert -------------------------------------------------------------------------
struct ConvertParams {
    const uint8_t *src;  // image 0, row 0
    uint8_t *dst;
    long long src_pitch;  // bytes between consecutive images of the batch (0 for n == 1)
    long long dst_pitch;
    const void *const *src_tab;  // optional per-image pointers (device array); overrides pitch
    void *const *dst_tab;
    int src_stride, dst_stride;
    int width, height;
    int n_images;
    float r, g, b;
};
// src_pixel != dst_pixel.  Returns hipSuccess or the launch error.
hipError_t launch_color_convert(int src_pixel, int dst_pixel, const ConvertParams &p, hipStream_t s);

// ---- resize: two-pass kernels through an HBM scratch (any geometry) ---------------------------
struct TwoPassParams {
    const uint8_t *src;
    uint8_t *dst;
    int src_stride, dst_stride;
    int src_w, src_h, dst_w, dst_h;
    // x axis
    const int32_t *x_left, *x_count;
    const float *x_weight;  // [dst_w][x_taps]
    int x_taps;
    // y axis
    const int32_t *y_count, *y_eff;  // eff: [dst_h][y_taps]
    const float *y_weight;           // [dst_h][y_taps]
    int y_taps;
    int row0, nrows;  // source rows [row0, row0 + nrows) are resampled into scratch
    float *scratch;   // [nrows][dst_w][channels]
};
hipError_t launch_resize_twopass(int pixel, const TwoPassParams &p, hipStream_t s);

// ---- resize: fused single kernel (horizontal pass -> per-wave LDS ring -> vertical pass) -------
struct FusedParams {
    const uint8_t *src;
    uint8_t *dst;
    long long src_pitch, dst_pitch;
    const void *const *src_tab;
    void *const *dst_tab;
    int src_stride, dst_stride;
    int src_w, src_h, dst_w, dst_h;
    int n_images;
    const int32_t *x_left;   // [dst_w], shifted so that x_left + x_taps <= src_w
    const float *x_weight;   // tap-major [x_taps][dst_w], zero padded (front-padded where shifted)
    int x_taps;
    const int4 *y_info;      // [dst_h] {left, need, count, 0}
    const float *y_weight;   // [dst_h][y_pad] zero padded
    const int32_t *y_slot;   // [dst_h][y_pad] ring slot (eff % ring_slots); padded taps repeat tap 0
    int y_taps, y_pad;
    int ring_slots;
    int rows_per_chunk, n_chunks;
};
// True if a fused instantiation exists for this format / tap counts and its LDS ring fits.
bool fused_supported(int pixel, int x_taps, int y_taps, int ring_slots);
// Padded row length of the y tables the fused kernel for y_taps expects (8, 16 or 32).
int fused_y_pad(int y_taps);
hipError_t launch_resize_fused(int pixel, const FusedParams &p, hipStream_t s, const char **kernel_name);

// ---- synthetic pixels ---------------------------------------------------------------------------
hipError_t launch_fill_random(void *dev, size_t bytes, uint64_t seed, int sample_bytes, hipStream_t s);
void fill_random_host(void *host, size_t bytes, uint64_t seed, int sample_bytes);

}  // namespace picha_hip
