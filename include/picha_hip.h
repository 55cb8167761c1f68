/*
 * picha_hip.h — C ABI of libpicha_hip.so: the MI355X (gfx950) implementation of picha's
 * pixel hot path (separable filtered resize + pixel-format / colour conversion).
 *
 * This is the drop-in boundary.  Every entry point below is what the reference's native
 * addon would bind for this path; the reference interface each one replaces is cited as
 * file:line into jhs67/picha v0.5.7.  Plain C: no C++ types, no exceptions, no torch types.
 * All functions return 0 (PICHA_OK) or a negative picha_status; the message of the last
 * failure on the calling thread is available from picha_hip_last_error().
 *
 * There is NO CPU fallback behind these entry points: if no gfx950 device is usable the
 * compute calls fail with PICHA_ERR_NO_DEVICE / PICHA_ERR_HIP.
 */
#ifndef PICHA_HIP_H_
#define PICHA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Pixel formats; values are the reference's PixelMode enum (src/picha.h:79-92). */
enum picha_pixel {
    PICHA_INVALID_PIXEL = -1,
    PICHA_RGB = 0,          /* 3 x u8  */
    PICHA_RGBA = 1,         /* 4 x u8  */
    PICHA_GREY = 2,         /* 1 x u8  */
    PICHA_GREYA = 3,        /* 2 x u8  */
    PICHA_R16 = 4,          /* 1 x u16 (native endian) */
    PICHA_R16G16 = 5,       /* 2 x u16 */
    PICHA_R16G16B16 = 6,    /* 3 x u16 */
    PICHA_R16G16B16A16 = 7, /* 4 x u16 */
    PICHA_NUM_PIXELS = 8
};

/* Resize filters; values are the reference's ResizeFilterTag (src/resize.cc:151-160). */
enum picha_filter {
    PICHA_FILTER_CUBIC = 0,
    PICHA_FILTER_LANCZOS = 1,
    PICHA_FILTER_CATMULROM = 2,
    PICHA_FILTER_MITCHEL = 3,
    PICHA_FILTER_BOX = 4,
    PICHA_FILTER_TRIANGLE = 5,
    PICHA_NUM_FILTERS = 6
};

enum picha_status {
    PICHA_OK = 0,
    PICHA_ERR_INVALID_IMAGE = -1,      /* "invalid image"        (src/resize.cc:336-339, src/picha.cc:61-85) */
    PICHA_ERR_INVALID_DIMENSIONS = -2, /* "invalid dimensions"   (src/resize.cc:343-346) */
    PICHA_ERR_INVALID_FILTER = -3,     /* "invalid filter mode"  (src/resize.cc:184-187) */
    PICHA_ERR_INVALID_FILTER_WIDTH = -4, /* "invalid filter width" (src/resize.cc:192-195) */
    PICHA_ERR_PIXEL_MISMATCH = -5,     /* assert(src.pixel == dst.pixel) (src/resize.cc:137); size asserts (src/colorconvert.cc:138-139) */
    PICHA_ERR_EXPECTED_PIXEL = -6,     /* "expected pixel mode"  (src/colorconvert.cc:236-239) */
    PICHA_ERR_DEGENERATE_TAPS = -7,    /* assert(totalweight > 0) (src/resize.cc:45) — the reference aborts */
    PICHA_ERR_NO_DEVICE = -8,          /* no usable gfx950 device */
    PICHA_ERR_HIP = -9,                /* HIP runtime failure; text in picha_hip_last_error() */
    PICHA_ERR_ARGUMENT = -10           /* null pointer / negative count at the C boundary */
};

/*
 * Image descriptor; mirrors NativeImage {data, stride, width, height, pixel}
 * (src/picha.h:202-218).  row(y) = data + y * stride.  `data` is a host pointer for the
 * host entry points and a device pointer for the *_device entry points.
 */
typedef struct picha_image {
    void *data;
    int32_t stride;
    int32_t width;
    int32_t height;
    int32_t pixel; /* enum picha_pixel */
} picha_image;

/* ---- format helpers (src/picha.h:174-200, 212-215) -------------------------------------- */
int picha_pixel_bytes(int pixel);    /* pixelBytes();   0 for an invalid format */
int picha_pixel_channels(int pixel); /* pixelChannels(); 0 for an invalid format */
int picha_row_stride(int width, int pixel); /* NativeImage::row_stride: (bytes*w + 3) & ~3 */
/* pixelSymbolToEnum / pixelEnumToSymbol (src/picha.cc:36-51): "rgb", "rgba", ... */
int picha_pixel_from_name(const char *name); /* PICHA_INVALID_PIXEL if unknown */
const char *picha_pixel_name(int pixel);     /* NULL if invalid */
/* symbolToResizeFilter (src/resize.cc:162-171): "cubic","lanczos","catmulrom","mitchel","box","triangle" */
int picha_filter_from_name(const char *name); /* -1 if unknown */
const char *picha_filter_name(int filter);

/* ---- option marshalling -------------------------------------------------------------------- */
/*
 * getResizeOptions (src/resize.cc:173-198).  Defaults {cubic, 0.70f}; if has_filter the scale
 * resets to 1.0f and `filter` must be valid (else PICHA_ERR_INVALID_FILTER); if has_filter_scale
 * the scale is float(filter_scale) and NaN or <= 0 gives PICHA_ERR_INVALID_FILTER_WIDTH.
 */
int picha_resize_options(int has_filter, int filter, int has_filter_scale, double filter_scale,
                         int *out_filter, float *out_scale);
/*
 * ColorSettings + getSettings (src/colorconvert.h:11-14, src/colorconvert.cc:6-22).  Pass NaN for
 * a weight the caller did not supply.  out[3] = normalised {r, g, b} factors.
 */
void picha_color_settings(double red_weight, double green_weight, double blue_weight, float out[3]);

/*
 * makeContribs (src/resize.cc:19-50) for ScaledFilter<filter>(filter_scale) mapping src_size
 * samples onto dst_size.  Exposed so tests can compare the product's host tap builder with the
 * oracle.  left/right: dst_size ints each.  weights: dst_size * max_taps floats, row i holds the
 * taps of output i (zero padded).  *taps_out receives the largest tap count.  Returns
 * PICHA_ERR_ARGUMENT if max_taps is too small (then *taps_out tells the size needed).
 */
int picha_hip_make_contribs(int filter, float filter_scale, int src_size, int dst_size,
                            int32_t *left, int32_t *right, float *weights, int max_taps,
                            int *taps_out);

/* ---- runtime ------------------------------------------------------------------------------- */
const char *picha_hip_last_error(void); /* thread-local text of the last failure ("" if none) */
const char *picha_hip_strerror(int status); /* the reference's message string for a status */
int picha_hip_device_count(void); /* number of visible HIP devices (0 if none / no driver) */
/* Name of the gcnArch of the current device into buf (e.g. "gfx950"). */
int picha_hip_device_arch(char *buf, size_t len);
/*
 * Tuning / test knobs (outside the drop-in surface; process-global, read once per call under a lock).  Every key
 * picha_hip_set_option accepts:
 *   "numerics"   0 = reference arithmetic, bit-identical results (default); 1 = fused multiply-add
 *                flavour of the resize kernels, at most one code value from the reference
 *                (environment: PICHA_HIP_NUMERICS=fma sets the initial value)
 *   "path"       0 = auto, 1 = force the two-pass scratch kernels, 2 = force a fused kernel (error if
 *                none applies), 3 = the LDS-ring span kernel (never the streaming or accumulator kernels),
 *                4 = the accumulator kernel (error if the tap table is not representable), 5 = the row-staged
 *                LDS kernel (error if it does not apply), 6 = the run kernel (periodic tap tables; error if it does
 *                not apply).  Chained convert + resize calls honour it too: the fused convert+resize kernel is only
 *                taken for 0 and 5
 *   "acc"        accumulator kernel: 0 = auto, 1 = whenever representable, 2 = never
 *   "acc_p2"     accumulator kernel with two outputs per lane: 0 = auto (measured table), 1 = whenever a shape
 *                exists, 2 = never
 *   "long_block" long-window accumulator kernel (33..256 horizontal taps): threads per workgroup, 0 = auto (most live
 *                resident waves per CU), 64 / 128 / 192 / 256
 *   "per_lane"   0 = auto, else output columns per lane for the fused kernels (1, 2, 3, 4; 3 only in the
 *                symmetric-table kernel)
 *   "row_chunks" 0 = auto, else number of row chunks per image for the fused kernels
 *   "stream_rot" 0 = auto, 1 / 2 = force the rotating-accumulator / unrolled variant of the streaming kernel
 *   "dpp"        1 = use the cross-lane pixel-sharing streaming kernel where it applies (rgba, 2x, 5 taps), 2 = never
 *   "sym"        symmetric-table streaming kernel (integer scales): 0 = auto, 2 = never
 *   "xch"        neighbour-exchange form of the symmetric-table / accumulator kernels (a lane loads only the pixels
 *                it owns and takes the rest of its window from the two neighbouring lanes): 0 = auto (measured
 *                table), 1 = wherever an instantiation exists, 2 = never
 *   "work"       symmetric-table / accumulator kernels: 0 = auto (explicit work-item list for launches of >= 6144 waves,
 *                persistent launch for the neighbour-exchange accumulator kernel, else the plain 3-D grid), 1 = the list
 *                (one workgroup per item, short row chunks last), 2 = persistent launch (resident workgroups pull items
 *                from per-XCD atomic queues), 3 = the plain grid
 *   "work_tail"  per mille of every XCD's sub-list that is cut into short row chunks (default 200);
 *   "work_tail_div"  a short chunk = a long chunk / this (default 4)
 *   "up"         integer upscales on the accumulator kernel (several outputs per lane share one window): 0 = auto (16-bit
 *                formats), 1 = wherever a shape exists, 2 = never (LDS-ring span kernel)
 *   "block_waves"  waves per workgroup of the symmetric-table kernel (1..4, 0 = auto)
 *   "sym_sync"   1 = one barrier per step keeps the waves of such a workgroup on the same source rows
 *   "rows"       row-staged LDS downscale kernel: 0 = auto, 1 = whenever it applies, 2 = never
 *   "run"        run kernel (scale = a ratio of small integers, e.g. 7.5: a lane resamples a run of consecutive outputs
 *                with wave-uniform weights; the edge columns come from a second small kernel) and its upscale form
 *                (resize_up: exactly 2x / 4x in x, vertical windows of at most four rows) and ring-aliased form (resize_ring:
 *                lanczos at 2x / 4x): 0 = auto (downscales: grey / greya
 *                with >= 26 taps, their shorter instantiated windows (1.5x .. 8x) for batches of >= 96 images; 2x upscales: every format with an
 *                instantiation), 1 = wherever an instantiation exists, 2 = never.  "per_lane" (2, 4, 8), "row_chunks", "block_waves" and "work" (1, 2; default
 *                plain grid) apply to it too
 *   "run_xch"    (r04) its neighbour-exchange form for dword-sized pixels at integer scales (a lane loads only the pixels it owns, wave-uniform
 *                weights, clipped columns recomputed from the image; cfg5's kernel): 0 = auto, 1 = wherever instantiated, 2 = never
 *   "run_ws"     (r04) the weights of one phase of the period in SGPRs (the 7.5x shapes: four waves per SIMD): 0 = auto, 1 = wherever instantiated, 2 = never
 *   "run_phase"  (r04) time-phased stores of the neighbour-exchange form (finished rows wait in registers until a chip-wide 100 MHz counter crosses a
 *                phase boundary): 0 = auto (launches with >= 3 source rows per output row and >= 20 000 image-rows), -1 = off, n = phases of 2^n ticks
 *   "run_shu"    (r05) its shared-unpack form for 8-bit grey / greya / rgb (a lane converts only the pixels its window starts with; the rest arrive converted
 *                from the next one or two lanes through LDS; also follows the row list of ring-aliased vertical tables): 0 = auto (where measured faster), 1 =
 *                wherever instantiated, 2 = never
 *   "run_edges"  (r05) the columns whose windows an image edge clips: 0 / 3 = auto (produced inside the march where its weights are lane registers — integer
 *                upscales, the neighbour-exchange form — else by a small second kernel reading a stash or the image, whichever measured faster), 1 = second
 *                kernel from the stash the march writes, 2 = second kernel from the image
 *   "edge_rows"  (r05) most output rows per block of that second kernel (1 .. 64; 0 = 64, what 8 KB of LDS hold): 16 .. 64 measured within 1 %
 *   "run_efuse"  (r05) measurement builds only (-DPICHA_RUN_EFUSE): 1 = the edge pass as extra workgroups inside the march's grid (measured slower; ignored otherwise)
 *   "run_round"  its window offsets inside a period: 0 = auto (the rounding with the shorter window), 1 = down, 2 = up
 *                ("run" = 3 is a measurement aid: the run kernel WITHOUT its edge columns — those come out wrong)
 *   "fuse"       picha_hip_convert_resize*: 0 = one fused kernel where it exists and was measured faster (default: every
 *                convert -> downscale pair of the row-staged kernel; resize -> convert for 16-bit and rgba sources; r04: convert ->
 *                integer upscale for the codec-edge pairs), 1 = one kernel wherever an instantiation exists, 2 = always two kernels
 *   "cc_mode"    colour convert layout: 0 = auto, 1 = the 4-pixel kernel, 2 = wide kernel with non-temporal
 *                stores, 3 = store-shaped kernel for expanding pairs
 *   "xcd_remap"  streaming kernels: 0 = auto, 1 = always / 2 = never permute workgroup ids so that each XCD works
 *                through a contiguous range of (column strip, row chunk, image) and neighbours share its L2
 *   "async_devices"  asynchronous entry points: 0 = queue on the caller's current device (default); N > 0 = deal
 *                requests round-robin to the first min(N, visible) devices, one dispatcher each — a single host
 *                process then drives every GPU of a node with independent images (environment:
 *                PICHA_HIP_ASYNC_DEVICES)
 *   "async_virtual_devices"  1 = "async_devices" may exceed the visible device count (dispatcher k runs on device
 *                k mod visible; at most 8 dispatchers per device) — rehearses the spread on a one-GPU box
 *   "chain_window_mb"  size of the intermediate buffer of picha_hip_convert_resize* in MB (default 4096)
 * Returns PICHA_ERR_ARGUMENT for an unknown key.
 */
int picha_hip_set_option(const char *key, long value);
long picha_hip_get_option(const char *key);
/* Name of the kernel instantiation the last device operation on this thread launched (diagnostics, bench JSON):
 *   "resize_sym<...>"        symmetric-table streaming kernel (",xch": neighbour-exchange form; ",epilogue -> fmt": colour convert fused in at the
 *                            store (r04); ",items" / ",persistent": launched through a work-item list / by persistent workgroups)
 *   "resize_stream<...>"     general streaming kernel (",rot": rotating accumulators), "resize_stream_dpp<...>"
 *   "resize_acc<...>"        accumulator kernel ("P2": two outputs per lane, ",xch"; "S0": upscale shape, outputs share one
 *                            window; ",epilogue -> fmt": colour convert fused in at the store), "resize_acc_long<...>" (> 36 taps)
 *   "resize_rows<...>"       row-staged LDS kernel ("src chN -> chM": colour convert fused in)
 *   "resize_up<...> / resize_ring<...> + resize_edges<N columns>"   ring-in-registers kernel: 2x / 4x upscales (F: factor, P outputs
 *                            per lane) / ring-aliased lanczos downscales (K ring rows), and the edge-column kernel; "resize_up<src chNuM,...":
 *                            (r04) a colour convert from that source format fused in front (picha_hip_convert_resize*, convert first)
 *   "resize_run<...> + resize_edges<N columns>"   run kernel (Q outputs / PX source pixels per period, P outputs per lane; r04: ",xchN" neighbour
 *                            exchange with N window pixels in front of the lane's own, ",wsN" N phases' weights in SGPRs, ",phasedM" time-phased
 *                            stores with phases of 2^M x 10 ns) and the kernel that recomputes the N columns whose windows an image edge clips
 *   "resize_span<...>"       LDS-ring span kernel (upscales, ring-aliasing cases)
 *   "twopass"                hpass_generic + vpass_generic through an HBM scratch
 *   "color_convert_wide<a->b,..>", "color_convert_expand<a->b>", "color_convert<a->b,4px>"
 *   "A + B"                  a chained convert + resize that ran as two kernels */
const char *picha_hip_last_kernel(void);

/* ---- the hot path: host-pointer entry points -------------------------------------------------
 * These replace the two functions the reference's NAN methods and libuv workers call:
 *   resizeImage(const ResizeOptions&, NativeImage& src, NativeImage& dst)   src/resize.cc:270-280
 *     (called from UV_resize src/resize.cc:291-294 and NAN_METHOD(resizeSync) src/resize.cc:399)
 *   doColorConvert(const ColorSettings&, NativeImage& src, NativeImage& dst) src/colorconvert.cc:171-188
 *     (called from UV_colorConvert src/colorconvert.cc:199-202 and colorConvertSync :288)
 * `data` pointers are host memory; the call uploads, runs the HIP kernels on the current device,
 * downloads into dst->data and returns when dst is complete.  dst row padding is left untouched
 * (the reference never writes it either, src/resize.cc:122-132).  Thread-safe.
 */
int picha_hip_resize(const picha_image *src, const picha_image *dst, int filter, float filter_scale);
int picha_hip_color_convert(const picha_image *src, const picha_image *dst,
                            float r_factor, float g_factor, float b_factor);
/* n independent images in one call (one H2D / kernel / D2H pipeline); same semantics per image. */
int picha_hip_resize_batch(int n, const picha_image *src, const picha_image *dst, int filter,
                           float filter_scale);
/*
 * Per-call options (r04).  The reference's options are per call (getResizeOptions, src/resize.cc:173-198); the one knob of this
 * library that changes RESULTS — the numerics flavour — can be chosen per call too, so that two callers in one process (one Node
 * process serving two pipelines) do not have to agree on it.  `size` = sizeof(picha_call_opts) (forward compatibility);
 * numerics: -1 = the process-wide "numerics" option, 0 = reference arithmetic (bit-identical), 1 = fused multiply-add flavour
 * (at most one code value from the reference).  opts == NULL behaves like the plain entry point.  picha_hip_resize_async_ex carries
 * the choice with the request (the dispatcher batches requests of equal options together); the plain asynchronous entry points take
 * the process-wide option when their batch is launched.
 */
typedef struct picha_call_opts {
    int32_t size;
    int32_t numerics;
} picha_call_opts;
int picha_hip_resize_ex(const picha_image *src, const picha_image *dst, int filter, float filter_scale,
                        const picha_call_opts *opts);
int picha_hip_resize_batch_ex(int n, const picha_image *src, const picha_image *dst, int filter,
                              float filter_scale, const picha_call_opts *opts);
int picha_hip_color_convert_batch(int n, const picha_image *src, const picha_image *dst,
                                  float r_factor, float g_factor, float b_factor);

/* ---- the hot path: asynchronous host-pointer entry points -----------------------------------------
 * Replacement for the reference's worker dispatch — uv_queue_work(loop, req, UV_resize, V8_resize)
 * (src/resize.cc:362-364) and its colour-convert twin (src/colorconvert.cc:252-254).  Instead of one
 * image per libuv worker, requests go to a per-device dispatcher that coalesces whatever is queued
 * into batches (one H2D / kernel launch / D2H per batch, through pinned staging, three batches in
 * flight so staging, GPU work and copy-out overlap).
 * Argument problems are reported synchronously through the return value, exactly like the reference's
 * synchronous throws (src/resize.cc:336-351); otherwise the call returns PICHA_OK at once and `done`
 * is invoked later, from a library thread, with the outcome.  src->data and dst->data must stay
 * valid until then (the reference pins them the same way, src/resize.cc:355-357).
 */
typedef void (*picha_hip_done_fn)(void *user, int status, const char *error);
int picha_hip_resize_async(const picha_image *src, const picha_image *dst, int filter,
                           float filter_scale, picha_hip_done_fn done, void *user);
int picha_hip_color_convert_async(const picha_image *src, const picha_image *dst, float r_factor,
                                  float g_factor, float b_factor, picha_hip_done_fn done, void *user);
/* picha_hip_resize_async with per-call options (picha_call_opts above): the reference's options are per call, src/resize.cc:179-198 */
int picha_hip_resize_async_ex(const picha_image *src, const picha_image *dst, int filter, float filter_scale,
                              const picha_call_opts *opts, picha_hip_done_fn done, void *user);
/* Blocks until every request queued so far has completed (tests, orderly shutdown). */
int picha_hip_async_drain(void);
/* Number of batches the dispatcher has launched and of requests it has completed (diagnostics). */
void picha_hip_async_stats(uint64_t *batches, uint64_t *requests);
/* The same for the index-th dispatcher (one per device the "async_devices" option spreads requests over); returns the number
 * of dispatchers that exist.  Replaces nothing in the reference (its libuv pool has no per-worker statistics). */
int picha_hip_async_dispatcher_stats(int index, uint64_t *batches, uint64_t *requests);

/* ---- the hot path: device-resident entry points ------------------------------------------------
 * Same operations on images whose `data` already lives in the current device's HBM.  Asynchronous:
 * work is enqueued on `stream` (a hipStream_t passed as void*, NULL = the default stream) and the
 * call returns; no allocation or synchronisation happens once the tap tables for the geometry are
 * cached (first call for a geometry builds and uploads them).  Images of one call may differ in
 * geometry; equal-geometry runs are grouped into single launches.
 */
int picha_hip_resize_device(int n, const picha_image *src, const picha_image *dst, int filter,
                            float filter_scale, void *stream);
int picha_hip_color_convert_device(int n, const picha_image *src, const picha_image *dst,
                                   float r_factor, float g_factor, float b_factor, void *stream);
/* picha_hip_resize_device with per-call options (see picha_call_opts above). */
int picha_hip_resize_device_ex(int n, const picha_image *src, const picha_image *dst, int filter,
                               float filter_scale, const picha_call_opts *opts, void *stream);

/* ---- next to the path: colorConvert and resize chained on the device -----------
 * What a picha user writes as  resize(colorConvert(image, {pixel}), {width, height})  — or the other way
 * round — as ONE call: both kernels run back to back on the device and the intermediate image never crosses
 * PCIe (it lives in a reusable HBM buffer of at most "chain_window_mb" MB, default 4096; larger batches go
 * through it in slices).  Results are bit-identical to the two
 * reference calls in that order (the intermediate is quantised to the destination sample type in between,
 * exactly as the reference's intermediate Buffer is).  src and dst may differ in pixel format AND size:
 *   PICHA_CONVERT_THEN_RESIZE  colorConvert src to dst->pixel at the source size (src/colorconvert.cc:171-188),
 *                              then resize (src/resize.cc:270-280)
 *   PICHA_RESIZE_THEN_CONVERT  resize in src->pixel, then colorConvert to dst->pixel
 * Same error behaviour as the two single operations.
 */
typedef enum picha_chain_order { PICHA_CONVERT_THEN_RESIZE = 0, PICHA_RESIZE_THEN_CONVERT = 1 } picha_chain_order;
int picha_hip_convert_resize_device(int n, const picha_image *src, const picha_image *dst, int order,
                                    float r_factor, float g_factor, float b_factor, int filter,
                                    float filter_scale, void *stream);
int picha_hip_convert_resize(const picha_image *src, const picha_image *dst, int order, float r_factor,
                             float g_factor, float b_factor, int filter, float filter_scale);
int picha_hip_convert_resize_batch(int n, const picha_image *src, const picha_image *dst, int order,
                                   float r_factor, float g_factor, float b_factor, int filter,
                                   float filter_scale);
/* through the batching dispatcher, like picha_hip_resize_async */
int picha_hip_convert_resize_async(const picha_image *src, const picha_image *dst, int order, float r_factor,
                                   float g_factor, float b_factor, int filter, float filter_scale,
                                   picha_hip_done_fn done, void *user);

/* ---- next to the path: Image helpers on device-resident images ----------------
 * Image.prototype.equalPixels / avgChannelDiff / copy (lib/image.js:57-96) without a D2H copy of the pixels.
 * subView needs no call: it is {data + y*stride + x*bytes, same stride, w, h} (lib/image.js:76-87).
 *
 * compare: for each pair, the sum over the payload BYTES (not samples: lib/image.js:68-73 walks row(y)[x])
 * of |a - b| and the number of differing bytes, written to the HOST arrays (either may be NULL); the call
 * synchronises `stream`.  avgChannelDiff = sum / (height * width * bytes); equalPixels = differing == 0.
 * Pairs whose width, height or pixel differ are not compared: sum and count are set to UINT64_MAX
 * (the reference returns 255 / false for them).
 * copy: payload bytes of the common top-left min(width) x min(height) rectangle; PICHA_ERR_PIXEL_MISMATCH if
 * the formats differ ("can't copy pixels between different pixel types", lib/image.js:90-91).  Asynchronous.
 */
int picha_hip_image_compare_device(int n, const picha_image *a, const picha_image *b, uint64_t *sum_abs_diff,
                                   uint64_t *differing_bytes, void *stream);
int picha_hip_image_copy_device(int n, const picha_image *src, const picha_image *dst, void *stream);

/* ---- device memory helpers for hosts without an allocator of their own (Node addon, C tests) --- */
int picha_hip_malloc(void **dev_ptr, size_t bytes);
int picha_hip_free(void *dev_ptr);
/*
 * A destination arena at a MEASURED placement (r05).  How long a resize takes depends on where its destination sits in HBM relative to its source — by 3-7 % between
 * buffers allocated one after the other, the first one after the source being the slowest in every record (profiles/r04_placement.txt, r05_placement.txt) — and nothing in the
 * addresses predicts it.  A service allocates its output arena once: this call allocates `candidates` (1..16) buffers of `bytes`, times picha.resize of the n device images
 * src[] into dst[] laid out in each of them (dst[i].data on entry = the byte OFFSET of image i inside the arena), keeps the fastest, frees the others, and returns its base in
 * *dev_ptr with dst[i].data rewritten to absolute device pointers.  ms_out (may be NULL): candidates floats, the time per launch measured into each (the kept one is the
 * minimum).  Synchronous; release with picha_hip_free.  The reference has no analogue: its destination is whatever `new Buffer` returned (src/resize.cc:293-300).
 */
int picha_hip_malloc_placed(void **dev_ptr, size_t bytes, int n, const picha_image *src, picha_image *dst, int filter, float filter_scale, int candidates, float *ms_out);
int picha_hip_memcpy_h2d(void *dev_dst, const void *host_src, size_t bytes, void *stream);
int picha_hip_memcpy_d2h(void *host_dst, const void *dev_src, size_t bytes, void *stream);
int picha_hip_stream_synchronize(void *stream);
/*
 * Synthetic pixels: sample k of the buffer = top bits of
 * splitmix64(seed ^ k) as u8 (sample_bytes 1) or little-endian u16 (sample_bytes 2).
 */
int picha_hip_fill_random(void *dev_ptr, size_t bytes, uint64_t seed, int sample_bytes, void *stream);
/* Host twin of the generator (same values), for parity checks on sub-batches. */
void picha_fill_random_host(void *host_ptr, size_t bytes, uint64_t seed, int sample_bytes);

#ifdef __cplusplus
}
#endif

#endif /* PICHA_HIP_H_ */
