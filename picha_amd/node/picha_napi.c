/*
 * picha_napi.c — Node.js addon (N-API, plain C) that puts picha's four pixel-path entry points on
 * top of libpicha_hip.so.  It replaces, for this path only, what the reference registers in
 * src/picha.cc:181-185 (resize, resizeSync, colorConvert, colorConvertSync) plus the `catalog`
 * object index.js:8-9 dereferences at load time (empty here: codecs are out of scope).
 *
 * Behaviour mirrored from the reference (file:line into jhs67/picha v0.5.7):
 *   argument arity/type checks and messages      src/resize.cc:322-325,368-371; src/colorconvert.cc:216-219,258-261
 *   image marshalling (jsImageToNativeImage)      src/picha.cc:61-85
 *   width/height via Uint32 -> int, must be > 0   src/resize.cc:341-346
 *   option parsing (getResizeOptions/getSettings) src/resize.cc:179-198; src/colorconvert.cc:6-22  (in the C ABI)
 *   result object (newJsImage)                    src/picha.cc:119-133
 *   async: the request goes to the library's batching dispatcher (picha_hip_*_async, the replacement for
 *   uv_queue_work), cb(undefined, image) runs on the loop thread via a threadsafe function, source buffer /
 *   result / callback pinned for the duration     src/resize.cc:353-364, 296-319
 * Deviations: resizeSync returns after throwing (the reference falls through); a device
 * failure — a class of error the reference does not have — is thrown (sync) or passed as cb(err).
 */
#include <math.h>
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/picha_hip.h"

#define NAPI_OK(call)                                   \
    do {                                                \
        if ((call) != napi_ok) {                        \
            napi_throw_error(env, NULL, "N-API failure: " #call); \
            return NULL;                                \
        }                                               \
    } while (0)

static napi_value throw_msg(napi_env env, const char *msg) {
    napi_throw_error(env, NULL, msg);
    return NULL;
}

static int is_object(napi_env env, napi_value v) {
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    return t == napi_object || t == napi_function;
}
static int is_function(napi_env env, napi_value v) {
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    return t == napi_function;
}
static int is_undefined(napi_env env, napi_value v) {
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return 1;
    return t == napi_undefined;
}

/* obj[name] (undefined if absent) */
static napi_value get_prop(napi_env env, napi_value obj, const char *name) {
    napi_value v;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) napi_get_undefined(env, &v);
    return v;
}

/* JS ToUint32 stored into a C int, as Uint32Value(...).FromMaybe(0) into `int` does (src/resize.cc:341) */
static int32_t to_uint32_as_int(napi_env env, napi_value v) {
    napi_value num;
    uint32_t u = 0;
    if (napi_coerce_to_number(env, v, &num) != napi_ok) {
        napi_value exc;
        napi_get_and_clear_last_exception(env, &exc);
        return 0;
    }
    if (napi_get_value_uint32(env, num, &u) != napi_ok) return 0;
    return (int32_t)u;
}

/* strict-equality against the format names (src/picha.cc:46-51) */
static int pixel_from_value(napi_env env, napi_value v) {
    napi_valuetype t;
    char buf[32];
    size_t len = 0;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_string) return PICHA_INVALID_PIXEL;
    if (napi_get_value_string_utf8(env, v, buf, sizeof buf, &len) != napi_ok) return PICHA_INVALID_PIXEL;
    return picha_pixel_from_name(buf);
}

/* jsImageToNativeImage (src/picha.cc:61-85): data == NULL means "invalid image" */
static picha_image to_native_image(napi_env env, napi_value img, napi_value *data_out) {
    picha_image r;
    memset(&r, 0, sizeof r);
    r.pixel = PICHA_INVALID_PIXEL;
    r.width = to_uint32_as_int(env, get_prop(env, img, "width"));
    r.height = to_uint32_as_int(env, get_prop(env, img, "height"));
    r.stride = to_uint32_as_int(env, get_prop(env, img, "stride"));
    r.pixel = pixel_from_value(env, get_prop(env, img, "pixel"));
    if (r.pixel >= 0 && r.pixel < PICHA_NUM_PIXELS) {
        napi_value data = get_prop(env, img, "data");
        bool isbuf = false;
        if (napi_is_buffer(env, data, &isbuf) == napi_ok && isbuf) {
            void *p = NULL;
            size_t len = 0;
            if (napi_get_buffer_info(env, data, &p, &len) == napi_ok) {
                size_t rw = (size_t)picha_pixel_bytes(r.pixel) * (size_t)(uint32_t)r.width;
                size_t need = (size_t)(uint32_t)r.height * (size_t)(uint32_t)r.stride - (uint32_t)r.stride + rw;
                if (len >= need && r.height != 0) {
                    r.data = p;
                    if (data_out) *data_out = data;
                }
            }
        }
    }
    return r;
}

/* newJsImage (src/picha.cc:119-133): {width, height, stride, pixel, data: uninitialised Buffer} */
static napi_value new_js_image(napi_env env, int w, int h, int pixel, picha_image *native) {
    napi_value obj, v, buf;
    void *p = NULL;
    int stride = picha_row_stride(w, pixel);
    if (stride <= 0 || w <= 0 || h <= 0) { /* width * bytes-per-pixel does not fit an int stride (the reference wraps) */
        napi_throw_error(env, NULL, "invalid dimensions");
        return NULL;
    }
    NAPI_OK(napi_create_object(env, &obj));
    NAPI_OK(napi_create_int32(env, w, &v));
    NAPI_OK(napi_set_named_property(env, obj, "width", v));
    NAPI_OK(napi_create_int32(env, h, &v));
    NAPI_OK(napi_set_named_property(env, obj, "height", v));
    NAPI_OK(napi_create_int32(env, stride, &v));
    NAPI_OK(napi_set_named_property(env, obj, "stride", v));
    NAPI_OK(napi_create_string_utf8(env, picha_pixel_name(pixel), NAPI_AUTO_LENGTH, &v));
    NAPI_OK(napi_set_named_property(env, obj, "pixel", v));
    NAPI_OK(napi_create_buffer(env, (size_t)h * (size_t)stride, &p, &buf));
    NAPI_OK(napi_set_named_property(env, obj, "data", buf));
    native->data = p;
    native->stride = stride;
    native->width = w;
    native->height = h;
    native->pixel = pixel;
    return obj;
}

/* ---- one operation, shared by the sync and async forms --------------------------------------- */
typedef struct op_ctx {
    int is_resize; /* 1 resize, 0 colorConvert, 2 both chained on the device (convertResizeSync, outside the drop-in surface) */
    int order;     /* chained: PICHA_CONVERT_THEN_RESIZE or PICHA_RESIZE_THEN_CONVERT */
    picha_image src, dst;
    int filter;
    float filter_scale;
    float cs[3];
    int numerics;  /* resize only: -1 = the process-wide option, 0 reference arithmetic, 1 fma — opts.numerics ('exact' / 'fma'), this library's per-call extension (picha_call_opts) */
    /* async only */
    napi_threadsafe_function tsfn;
    napi_ref src_ref, dst_ref;
    int status;
    char error[512];
} op_ctx;

static void run_op(op_ctx *c) {
    if (c->is_resize == 2)
        c->status = picha_hip_convert_resize(&c->src, &c->dst, c->order, c->cs[0], c->cs[1], c->cs[2], c->filter, c->filter_scale);
    else if (c->is_resize) {
        picha_call_opts co = { (int32_t)sizeof co, c->numerics };
        c->status = picha_hip_resize_ex(&c->src, &c->dst, c->filter, c->filter_scale, c->numerics >= 0 ? &co : NULL);
    }
    else c->status = picha_hip_color_convert(&c->src, &c->dst, c->cs[0], c->cs[1], c->cs[2]);
    if (c->status != PICHA_OK) {
        const char *m = picha_hip_last_error(); /* thread-local: read it on the thread that failed */
        strncpy(c->error, (m && *m) ? m : picha_hip_strerror(c->status), sizeof c->error - 1);
        c->error[sizeof c->error - 1] = 0;
    }
}

/* completion on the library's completer thread: hop to the loop thread through the threadsafe function */
static void async_done(void *user, int status, const char *error) {
    op_ctx *c = (op_ctx *)user;
    c->status = status;
    if (status != PICHA_OK) {
        strncpy(c->error, (error && *error) ? error : picha_hip_strerror(status), sizeof c->error - 1);
        c->error[sizeof c->error - 1] = 0;
    }
    napi_call_threadsafe_function(c->tsfn, c, napi_tsfn_blocking);
}

/* loop thread: the reference's V8_resize / V8_colorConvert (src/resize.cc:296-319) */
static void async_call_js(napi_env env, napi_value cb, void *context, void *data) {
    op_ctx *c = (op_ctx *)data;
    napi_value dst, global, argv[2], result;
    (void)context;
    if (env == NULL) { free(c); return; } /* environment is shutting down */
    napi_get_reference_value(env, c->dst_ref, &dst);
    napi_get_global(env, &global);
    if (c->status == PICHA_OK) {
        napi_get_undefined(env, &argv[0]);
        argv[1] = dst;
    } else {
        napi_value msg;
        napi_create_string_utf8(env, c->error, NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &argv[0]);
        napi_get_undefined(env, &argv[1]);
    }
    napi_delete_reference(env, c->src_ref);
    napi_delete_reference(env, c->dst_ref);
    napi_threadsafe_function tsfn = c->tsfn;
    free(c);
    /* an exception thrown by the callback propagates as an uncaught exception (reference: FatalException) */
    napi_call_function(env, global, cb, 2, argv, &result);
    napi_release_threadsafe_function(tsfn, napi_tsfn_release);
}

/* Parses (image, opts) for either op.  Returns 0 and leaves a pending JS exception on failure. */
static int prepare(napi_env env, napi_value img, napi_value opts, op_ctx *c, napi_value *src_data, napi_value *dst_obj) {
    c->src = to_native_image(env, img, src_data);
    if (!c->src.data) { throw_msg(env, "invalid image"); return 0; }
    int to_pixel = c->src.pixel, width = c->src.width, height = c->src.height;
    if (c->is_resize != 1) {
        to_pixel = pixel_from_value(env, get_prop(env, opts, "pixel"));
        if (to_pixel == PICHA_INVALID_PIXEL) { throw_msg(env, "expected pixel mode"); return 0; }
        /* getSettings (src/colorconvert.cc:6-22): NumberValue of each weight, NaN keeps the default */
        const char *names[3] = {"redWeight", "greenWeight", "blueWeight"};
        double w[3];
        for (int i = 0; i < 3; ++i) {
            napi_value num;
            w[i] = NAN;
            if (napi_coerce_to_number(env, get_prop(env, opts, names[i]), &num) != napi_ok ||
                napi_get_value_double(env, num, &w[i]) != napi_ok) {
                napi_value exc;
                napi_get_and_clear_last_exception(env, &exc);
                w[i] = NAN;
            }
        }
        picha_color_settings(w[0], w[1], w[2], c->cs);
    }
    if (c->is_resize != 0) {
        width = to_uint32_as_int(env, get_prop(env, opts, "width"));
        height = to_uint32_as_int(env, get_prop(env, opts, "height"));
        if (width <= 0 || height <= 0) { throw_msg(env, "invalid dimensions"); return 0; }
        /* getResizeOptions (src/resize.cc:179-198) */
        napi_value f = get_prop(env, opts, "filter"), fs = get_prop(env, opts, "filterScale");
        int has_filter = !is_undefined(env, f), has_scale = !is_undefined(env, fs), tag = -1;
        double scale = NAN;
        if (has_filter) {
            napi_valuetype t;
            char buf[32];
            size_t len;
            if (napi_typeof(env, f, &t) == napi_ok && t == napi_string &&
                napi_get_value_string_utf8(env, f, buf, sizeof buf, &len) == napi_ok)
                tag = picha_filter_from_name(buf);
        }
        if (has_scale) {
            napi_value num;
            if (napi_coerce_to_number(env, fs, &num) != napi_ok || napi_get_value_double(env, num, &scale) != napi_ok) {
                napi_value exc;
                napi_get_and_clear_last_exception(env, &exc);
                scale = NAN;
            }
        }
        int rc = picha_resize_options(has_filter, tag, has_scale, scale, &c->filter, &c->filter_scale);
        if (rc != PICHA_OK) { throw_msg(env, picha_hip_strerror(rc)); return 0; }
        /* opts.numerics: not a reference option (the reference has one arithmetic); absent = the library's process-wide choice */
        c->numerics = -1;
        napi_value nm = get_prop(env, opts, "numerics");
        if (!is_undefined(env, nm)) {
            napi_valuetype t;
            char buf[16];
            size_t len;
            if (napi_typeof(env, nm, &t) == napi_ok && t == napi_string && napi_get_value_string_utf8(env, nm, buf, sizeof buf, &len) == napi_ok &&
                (strcmp(buf, "exact") == 0 || strcmp(buf, "fma") == 0))
                c->numerics = strcmp(buf, "fma") == 0;
            else { throw_msg(env, "invalid numerics (expected 'exact' or 'fma')"); return 0; }
        }
    }
    if (c->is_resize == 2) { /* opts.order: 'resizeFirst' resizes in the source format, then converts */
        napi_value o = get_prop(env, opts, "order");
        napi_valuetype t;
        char buf[32];
        size_t len;
        c->order = PICHA_CONVERT_THEN_RESIZE;
        if (napi_typeof(env, o, &t) == napi_ok && t == napi_string &&
            napi_get_value_string_utf8(env, o, buf, sizeof buf, &len) == napi_ok && strcmp(buf, "resizeFirst") == 0)
            c->order = PICHA_RESIZE_THEN_CONVERT;
    }
    *dst_obj = new_js_image(env, width, height, to_pixel, &c->dst);
    return *dst_obj != NULL;
}

static napi_value do_sync(napi_env env, napi_callback_info info, int is_resize, const char *usage) {
    size_t argc = 8;
    napi_value argv[8];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc != 2 || !is_object(env, argv[0]) || !is_object(env, argv[1])) return throw_msg(env, usage);
    op_ctx c;
    memset(&c, 0, sizeof c);
    c.is_resize = is_resize;
    napi_value src_data = NULL, dst = NULL;
    if (!prepare(env, argv[0], argv[1], &c, &src_data, &dst)) return NULL;
    run_op(&c);
    if (c.status != PICHA_OK) return throw_msg(env, c.error);
    return dst;
}

static napi_value do_async(napi_env env, napi_callback_info info, int is_resize, const char *usage) {
    size_t argc = 8;
    napi_value argv[8];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (argc != 3 || !is_object(env, argv[0]) || !is_object(env, argv[1]) || !is_function(env, argv[2]))
        return throw_msg(env, usage);
    op_ctx *c = (op_ctx *)calloc(1, sizeof *c);
    if (!c) return throw_msg(env, "out of memory");
    c->is_resize = is_resize;
    napi_value src_data = NULL, dst = NULL, name, undef;
    if (!prepare(env, argv[0], argv[1], c, &src_data, &dst)) { free(c); return NULL; }
    /* pin the source Buffer and the result image until completion (src/resize.cc:355-357); the callback is
       held by the threadsafe function */
    napi_create_reference(env, src_data, 1, &c->src_ref);
    napi_create_reference(env, dst, 1, &c->dst_ref);
    napi_create_string_utf8(env, "picha", NAPI_AUTO_LENGTH, &name); /* Nan::AsyncResource("picha") */
    if (napi_create_threadsafe_function(env, argv[2], NULL, name, 0, 1, NULL, NULL, NULL, async_call_js, &c->tsfn) != napi_ok) {
        napi_delete_reference(env, c->src_ref);
        napi_delete_reference(env, c->dst_ref);
        free(c);
        return throw_msg(env, "could not create threadsafe function");
    }
    /* hand the request to the library's batching dispatcher (replaces uv_queue_work, src/resize.cc:362-364) */
    int rc = c->is_resize == 2 ? picha_hip_convert_resize_async(&c->src, &c->dst, c->order, c->cs[0], c->cs[1], c->cs[2], c->filter,
                                                                c->filter_scale, async_done, c)
             : c->is_resize    ? picha_hip_resize_async_ex(&c->src, &c->dst, c->filter, c->filter_scale, c->numerics >= 0 ? &(picha_call_opts){ (int32_t)sizeof(picha_call_opts), c->numerics } : NULL, async_done, c)
                               : picha_hip_color_convert_async(&c->src, &c->dst, c->cs[0], c->cs[1], c->cs[2], async_done, c);
    if (rc != PICHA_OK) { /* rejected synchronously, e.g. no device: throw like every other argument problem */
        char msg[512];
        const char *m = picha_hip_last_error();
        strncpy(msg, (m && *m) ? m : picha_hip_strerror(rc), sizeof msg - 1);
        msg[sizeof msg - 1] = 0;
        napi_delete_reference(env, c->src_ref);
        napi_delete_reference(env, c->dst_ref);
        napi_release_threadsafe_function(c->tsfn, napi_tsfn_abort);
        free(c);
        return throw_msg(env, msg);
    }
    napi_get_undefined(env, &undef);
    return undef;
}

static napi_value js_resize(napi_env env, napi_callback_info info) { return do_async(env, info, 1, "expected: resize(image, opts, cb)"); }
static napi_value js_resize_sync(napi_env env, napi_callback_info info) { return do_sync(env, info, 1, "expected: resizeSync(image, opts)"); }
static napi_value js_color_convert(napi_env env, napi_callback_info info) { return do_async(env, info, 0, "expected: colorConvert(image, opts, cb)"); }
static napi_value js_color_convert_sync(napi_env env, napi_callback_info info) { return do_sync(env, info, 0, "expected: colorConvertSync(image, opts)"); }

/* outside the drop-in surface: colorConvertSync + resizeSync in one device round trip (include/picha_hip.h, 8f row 2) */
static napi_value js_convert_resize_sync(napi_env env, napi_callback_info info) { return do_sync(env, info, 2, "expected: convertResizeSync(image, opts)"); }
static napi_value js_convert_resize(napi_env env, napi_callback_info info) { return do_async(env, info, 2, "expected: convertResize(image, opts, cb)"); }

/* extras outside the drop-in surface */
static napi_value js_device_count(napi_env env, napi_callback_info info) {
    napi_value v;
    (void)info;
    napi_create_int32(env, picha_hip_device_count(), &v);
    return v;
}
static napi_value js_last_kernel(napi_env env, napi_callback_info info) {
    napi_value v;
    (void)info;
    napi_create_string_utf8(env, picha_hip_last_kernel(), NAPI_AUTO_LENGTH, &v);
    return v;
}

static napi_value init(napi_env env, napi_value exports) {
    napi_value fn, catalog;
    struct { const char *name; napi_callback cb; } methods[] = {
        {"resize", js_resize}, {"resizeSync", js_resize_sync},
        {"colorConvert", js_color_convert}, {"colorConvertSync", js_color_convert_sync},
        {"convertResizeSync", js_convert_resize_sync}, {"convertResize", js_convert_resize},
        {"hipDeviceCount", js_device_count}, {"hipLastKernel", js_last_kernel},
    };
    for (size_t i = 0; i < sizeof methods / sizeof methods[0]; ++i) {
        NAPI_OK(napi_create_function(env, methods[i].name, NAPI_AUTO_LENGTH, methods[i].cb, NULL, &fn));
        NAPI_OK(napi_set_named_property(env, exports, methods[i].name, fn));
    }
    NAPI_OK(napi_create_object(env, &catalog)); /* index.js:8-9 reads Object.keys(picha.catalog) at load */
    NAPI_OK(napi_set_named_property(env, exports, "catalog", catalog));
    return exports;
}

NAPI_MODULE(picha, init)
