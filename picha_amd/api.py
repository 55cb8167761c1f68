"""Host-side mirror of picha's operator API for the pixel hot path.

    resize(image, opts, cb) / resizeSync(image, opts)              reference index.js:13-21, src/resize.cc:321-403
    colorConvert(image, opts, cb) / colorConvertSync(image, opts)  reference index.js:25-33, src/colorconvert.cc:215-291

Same option names and semantics, same error strings; results are `Image` objects.  The async
variants hand the request to the library's batching dispatcher (picha_hip_*_async) and deliver
`cb(None, image)` from a library thread (the reference's callbacks likewise never receive an error
for argument problems — those throw synchronously; a device failure, a class of
error the reference does not have, is delivered as cb(err, None)).
"""
import ctypes
import itertools
import math
import numbers
import threading

import numpy as np

from . import _native as N
from .image import Image

PichaError = N.PichaError

# Async calls go to the library's batching dispatcher (picha_hip_*_async: the replacement for the
# reference's uv_queue_work dispatch, src/resize.cc:362-364).  Until a request completes, its source
# buffer, destination image and callback trampoline are pinned here — the reference pins the same
# three things with Persistent handles (src/resize.cc:355-357).
_pending = {}
_pending_lock = threading.Lock()
_tokens = itertools.count(1)


def _on_done(user, status, err):
    """The ONE trampoline every async request completes through (`user` = the request's token).  A module-level
    CFUNCTYPE object lives as long as the module, so native code never returns through a freed closure."""
    with _pending_lock:
        entry = _pending.pop(int(user or 0), None)
    if entry is None:
        return
    cb, _src, dst = entry
    if status == 0:
        cb(None, dst)
    else:
        cb(PichaError((err or b"").decode() or N.lib().picha_hip_strerror(status).decode(), status), None)


_DONE = N.DONE_FN(_on_done)
_DONE_PTR = ctypes.cast(_DONE, ctypes.c_void_p)


def _submit_async(submit, src, dst, cb):
    """submit(done_ptr, user_ptr) -> status.  Calls cb(None, dst) or cb(err, None) from a library thread."""
    token = next(_tokens)
    with _pending_lock:
        _pending[token] = (cb, src, dst)
    rc = submit(_DONE_PTR, ctypes.c_void_p(token))
    if rc != 0:
        with _pending_lock:
            _pending.pop(token, None)
        N.check(rc)


def drain():
    """Wait for every queued async request (tests, shutdown)."""
    N.check(N.lib().picha_hip_async_drain())


def _get(obj, key):
    if isinstance(obj, dict):
        return obj.get(key)
    return getattr(obj, key, None)


def _uint32(v):
    """JS ToUint32 of a property (undefined / NaN -> 0)."""
    if v is None or not isinstance(v, numbers.Real):
        return 0
    if isinstance(v, float) and (math.isnan(v) or math.isinf(v)):
        return 0
    return int(v) & 0xFFFFFFFF


def _to_int32(u):
    """Uint32Value stored into a C int (src/resize.cc:341-342): values > 2^31-1 wrap negative."""
    return u - (1 << 32) if u >= (1 << 31) else u


def _native_image(img):
    """jsImageToNativeImage (src/picha.cc:61-85).  Returns (CImage, keepalive) or None if invalid."""
    if img is None or isinstance(img, (int, float, str, bytes)):
        return None
    width = _to_int32(_uint32(_get(img, "width")))
    height = _to_int32(_uint32(_get(img, "height")))
    stride = _to_int32(_uint32(_get(img, "stride")))
    pixel = N.pixel_index(_get(img, "pixel")) if isinstance(_get(img, "pixel"), str) else -1
    data = _get(img, "data")
    if pixel < 0 or data is None:
        return None
    if not isinstance(data, np.ndarray):
        try:
            data = np.frombuffer(data, dtype=np.uint8)
        except (TypeError, ValueError):
            return None
    if data.dtype != np.uint8 or data.ndim != 1 or not data.flags["C_CONTIGUOUS"]:
        return None
    rw = N.PIXEL_BYTES[N.PIXELS[pixel]] * width
    if not (data.size >= height * stride - stride + rw and height != 0):
        return None
    if width <= 0 or height < 0 or stride < 0:
        return None
    return N.CImage(data.ctypes.data, stride, width, height, pixel), data


def _new_image(width, height, pixel):
    """newJsImage (src/picha.cc:119-133): stride = row_stride, data uninitialised."""
    stride = (N.PIXEL_BYTES[pixel] * width + 3) & ~3
    if stride > 0x7FFFFFFF:  # the reference's int row_stride wraps here (src/picha.h:214); refuse instead
        raise PichaError("invalid dimensions", N.ERR_INVALID_DIMENSIONS if hasattr(N, "ERR_INVALID_DIMENSIONS") else -2)
    data = np.empty(height * stride, dtype=np.uint8)
    return Image(width=width, height=height, pixel=pixel, stride=stride, data=data)


def _resize_options(opts):
    """getResizeOptions (src/resize.cc:179-198)."""
    L = N.lib()
    f = _get(opts, "filter")
    has_filter = f is not None
    tag = -1
    if has_filter:
        tag = L.picha_filter_from_name(f.encode()) if isinstance(f, str) else -1
    fs = _get(opts, "filterScale")
    has_scale = fs is not None
    scale = float("nan")
    if has_scale:
        try:
            scale = float(fs)
        except (TypeError, ValueError):
            scale = float("nan")
    out_f, out_s = ctypes.c_int(), ctypes.c_float()
    N.check(L.picha_resize_options(int(has_filter), tag, int(has_scale), scale, ctypes.byref(out_f), ctypes.byref(out_s)))
    return out_f.value, out_s.value


def _prepare_resize(image, opts, expected):
    if not _is_object(image) or not _is_object(opts):
        raise PichaError(expected)
    src = _native_image(image)
    if src is None:
        raise PichaError("invalid image")
    width = _to_int32(_uint32(_get(opts, "width")))
    height = _to_int32(_uint32(_get(opts, "height")))
    if width <= 0 or height <= 0:
        raise PichaError("invalid dimensions")
    ftag, fscale = _resize_options(opts)
    dst = _new_image(width, height, N.PIXELS[src[0].pixel])
    return src, dst, ftag, fscale


def _is_object(o):
    return o is not None and not isinstance(o, (int, float, str, bytes, bool))


def _run_resize(src, dst, ftag, fscale, numerics=None):
    cdst = N.CImage(dst.data.ctypes.data, dst.stride, dst.width, dst.height, N.pixel_index(dst.pixel))
    if numerics is None:
        N.check(N.lib().picha_hip_resize(ctypes.byref(src[0]), ctypes.byref(cdst), ftag, fscale))
    else:   # per-call numerics flavour (picha_call_opts): "exact" / "fma", this call only
        co = N.CCallOpts(numerics)
        N.check(N.lib().picha_hip_resize_ex(ctypes.byref(src[0]), ctypes.byref(cdst), ftag, fscale, ctypes.byref(co)))
    return dst


def resizeSync(image, opts, numerics=None):
    """picha.resizeSync.  `numerics` ("exact" / "fma") is this library's per-call extension (include/picha_hip.h, picha_call_opts);
    None = the process-wide option."""
    src, dst, ftag, fscale = _prepare_resize(image, opts, "expected: resizeSync(image, opts)")
    return _run_resize(src, dst, ftag, fscale, numerics)


def resize(image, opts, cb, numerics=None):
    """picha.resize.  `numerics` ("exact" / "fma"): per-call extension as for resizeSync (picha_hip_resize_async_ex: the request carries it)."""
    if not callable(cb):
        raise PichaError("expected: resize(image, opts, cb)")
    src, dst, ftag, fscale = _prepare_resize(image, opts, "expected: resize(image, opts, cb)")
    cdst = N.CImage(dst.data.ctypes.data, dst.stride, dst.width, dst.height, N.pixel_index(dst.pixel))
    L = N.lib()
    if numerics is None:
        _submit_async(lambda done, user: L.picha_hip_resize_async(ctypes.byref(src[0]), ctypes.byref(cdst), ftag, fscale, done, user),
                      src, dst, cb)
    else:
        co = N.CCallOpts(numerics)
        _submit_async(lambda done, user: L.picha_hip_resize_async_ex(ctypes.byref(src[0]), ctypes.byref(cdst), ftag, fscale, ctypes.byref(co), done, user),
                      src, dst, cb)


def _color_settings(opts):
    """getSettings (src/colorconvert.cc:6-22): NumberValue of each weight, NaN keeps the default."""
    vals = []
    for k in ("redWeight", "greenWeight", "blueWeight"):
        v = _get(opts, k)
        try:
            vals.append(float(v) if v is not None else float("nan"))
        except (TypeError, ValueError):
            vals.append(float("nan"))
    out = (ctypes.c_float * 3)()
    N.lib().picha_color_settings(vals[0], vals[1], vals[2], out)
    return out[0], out[1], out[2]


def _prepare_convert(image, opts, expected):
    if not _is_object(image) or not _is_object(opts):
        raise PichaError(expected)
    src = _native_image(image)
    if src is None:
        raise PichaError("invalid image")
    to_pixel = _get(opts, "pixel")
    if not isinstance(to_pixel, str) or N.pixel_index(to_pixel) < 0:
        raise PichaError("expected pixel mode")
    dst = _new_image(src[0].width, src[0].height, to_pixel)
    return src, dst, _color_settings(opts)


def _run_convert(src, dst, cs):
    cdst = N.CImage(dst.data.ctypes.data, dst.stride, dst.width, dst.height, N.pixel_index(dst.pixel))
    N.check(N.lib().picha_hip_color_convert(ctypes.byref(src[0]), ctypes.byref(cdst), cs[0], cs[1], cs[2]))
    return dst


def colorConvertSync(image, opts):
    src, dst, cs = _prepare_convert(image, opts, "expected: colorConvertSync(image, opts)")
    return _run_convert(src, dst, cs)


def colorConvert(image, opts, cb):
    if not callable(cb):
        raise PichaError("expected: colorConvert(image, opts, cb)")
    src, dst, cs = _prepare_convert(image, opts, "expected: colorConvert(image, opts, cb)")
    cdst = N.CImage(dst.data.ctypes.data, dst.stride, dst.width, dst.height, N.pixel_index(dst.pixel))
    L = N.lib()
    _submit_async(lambda done, user: L.picha_hip_color_convert_async(ctypes.byref(src[0]), ctypes.byref(cdst), cs[0], cs[1], cs[2],
                                                                     done, user), src, dst, cb)


# ---- next to the path: both operations in one device round trip ---------------
def _prepare_chain(image, opts, expected):
    src, mid, cs = _prepare_convert(image, opts, expected)
    width = _to_int32(_uint32(_get(opts, "width")))
    height = _to_int32(_uint32(_get(opts, "height")))
    if width <= 0 or height <= 0:
        raise PichaError("invalid dimensions")
    ftag, fscale = _resize_options(opts)
    order = 1 if _get(opts, "order") == "resizeFirst" else 0
    dst = _new_image(width, height, mid.pixel)
    cdst = N.CImage(dst.data.ctypes.data, dst.stride, dst.width, dst.height, N.pixel_index(dst.pixel))
    return src, dst, cdst, order, cs, ftag, fscale


def convertResizeSync(image, opts):
    """resizeSync(colorConvertSync(image, opts), opts) — or, with opts.order == 'resizeFirst',
    colorConvertSync(resizeSync(image, opts), opts) — as one call: the intermediate image stays in HBM.

    opts carries both option sets: pixel (+ redWeight/greenWeight/blueWeight) and width, height (+ filter,
    filterScale).  Results are bit-identical to the two reference calls in that order; errors are theirs.
    """
    src, dst, cdst, order, cs, ftag, fscale = _prepare_chain(image, opts, "expected: convertResizeSync(image, opts)")
    N.check(N.lib().picha_hip_convert_resize(ctypes.byref(src[0]), ctypes.byref(cdst), order, cs[0], cs[1], cs[2], ftag, fscale))
    return dst


def convertResize(image, opts, cb):
    """Async form of convertResizeSync, through the batching dispatcher (cb(None, image) from a library thread)."""
    if not callable(cb):
        raise PichaError("expected: convertResize(image, opts, cb)")
    src, dst, cdst, order, cs, ftag, fscale = _prepare_chain(image, opts, "expected: convertResize(image, opts, cb)")
    L = N.lib()
    _submit_async(lambda done, user: L.picha_hip_convert_resize_async(ctypes.byref(src[0]), ctypes.byref(cdst), order, cs[0], cs[1],
                                                                      cs[2], ftag, fscale, done, user), src, dst, cb)


# ---- helpers exposed for tests / tools ----------------------------------------------------------
def make_contribs(filter, filter_scale, src_size, dst_size):
    """The product's host tap builder (picha_amd/csrc/taps.cpp) -> (left, right, weights[dst, T])."""
    L = N.lib()
    tag = N.FILTERS.index(filter) if isinstance(filter, str) else int(filter)
    taps = ctypes.c_int(0)
    cap = 8
    while True:
        left = np.zeros(dst_size, np.int32)
        right = np.zeros(dst_size, np.int32)
        w = np.zeros((dst_size, cap), np.float32)
        rc = L.picha_hip_make_contribs(tag, filter_scale, src_size, dst_size, left.ctypes.data, right.ctypes.data,
                                       w.ctypes.data, cap, ctypes.byref(taps))
        if rc != 0 and taps.value > cap:
            cap = taps.value
            continue
        N.check(rc)
        return left, right, w[:, : taps.value].copy()


def color_settings(red=None, green=None, blue=None):
    return _color_settings({"redWeight": red, "greenWeight": green, "blueWeight": blue})
