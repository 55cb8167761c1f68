/*
 * picha_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the reference's pixel hot path (jhs67/picha v0.5.7,
 * src/resize.cc, src/colorconvert.cc, src/picha.h).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library, and only as the checker or the
 * reported CPU baseline.  The product (libpicha_hip.so, picha_amd/) never links or calls it.
 *
 * Build: see oracle/Makefile (gcc -O3 -ffp-contract=off, baseline x86-64: the same arithmetic
 * as node-gyp's Release build of the reference — IEEE binary32, no FMA contraction, no
 * fast-math).
 */
#ifndef PICHA_ORACLE_H_
#define PICHA_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as picha_image in include/picha_hip.h == NativeImage (src/picha.h:202-218). */
typedef struct oracle_image {
    void *data;
    int32_t stride, width, height, pixel;
} oracle_image;

/* 0 ok; -5 pixel mismatch (assert src.pixel == dst.pixel); -7 totalweight <= 0 (assert); -3 bad filter */
int oracle_resize(const oracle_image *src, const oracle_image *dst, int filter, float filter_scale);
int oracle_color_convert(const oracle_image *src, const oracle_image *dst, float r, float g, float b);
/* getSettings (src/colorconvert.cc:6-22); NaN = weight not supplied */
void oracle_color_settings(double r, double g, double b, float out[3]);
/* makeContribs (src/resize.cc:19-50); weights row i = taps of output i, zero padded to max_taps.
 * returns the largest tap count, or -7 (assert), -3 (bad filter), -10 (max_taps too small) */
int oracle_make_contribs(int filter, float filter_scale, int src_size, int dst_size,
                         int32_t *left, int32_t *right, float *weights, int max_taps);
int oracle_pixel_bytes(int pixel);
int oracle_pixel_channels(int pixel);
/* n images on `threads` pthreads (one image per thread at a time: the libuv-pool model of
 * src/resize.cc:362-364).  Used only for the cpu_baseline timing. */
int oracle_resize_batch_mt(int n, const oracle_image *src, const oracle_image *dst, int filter,
                           float filter_scale, int threads);

#ifdef __cplusplus
}
#endif
#endif
