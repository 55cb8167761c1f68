"""TEST INFRASTRUCTURE ONLY — see oracle/__init__.py."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libpicha_oracle.so")

PIXELS = ["rgb", "rgba", "grey", "greya", "r16", "r16g16", "r16g16b16", "r16g16b16a16"]
FILTERS = ["cubic", "lanczos", "catmulrom", "mitchel", "box", "triangle"]
BYTES = [3, 4, 1, 2, 2, 4, 6, 8]
CHANNELS = [3, 4, 1, 2, 1, 2, 3, 4]


class OracleImage(ctypes.Structure):
    _fields_ = [
        ("data", ctypes.c_void_p),
        ("stride", ctypes.c_int32),
        ("width", ctypes.c_int32),
        ("height", ctypes.c_int32),
        ("pixel", ctypes.c_int32),
    ]


_lib = None


def build(force=False):
    src = os.path.join(HERE, "picha_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(SO)
        P = ctypes.POINTER(OracleImage)
        L.oracle_resize.argtypes = [P, P, ctypes.c_int, ctypes.c_float]
        L.oracle_resize.restype = ctypes.c_int
        L.oracle_color_convert.argtypes = [P, P, ctypes.c_float, ctypes.c_float, ctypes.c_float]
        L.oracle_color_convert.restype = ctypes.c_int
        L.oracle_color_settings.argtypes = [ctypes.c_double] * 3 + [ctypes.POINTER(ctypes.c_float)]
        L.oracle_color_settings.restype = None
        L.oracle_make_contribs.argtypes = [
            ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int,
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.oracle_make_contribs.restype = ctypes.c_int
        L.oracle_resize_batch_mt.argtypes = [ctypes.c_int, P, P, ctypes.c_int, ctypes.c_float, ctypes.c_int]
        L.oracle_resize_batch_mt.restype = ctypes.c_int
        _lib = L
    return _lib


def _pix(p):
    return PIXELS.index(p) if isinstance(p, str) else int(p)


def _filt(f):
    return FILTERS.index(f) if isinstance(f, str) else int(f)


def row_stride(w, pixel):
    return (BYTES[_pix(pixel)] * w + 3) & ~3


def _desc(buf, width, height, pixel, stride, offset=0):
    """buf: 1-D uint8 numpy array holding the image bytes."""
    assert buf.dtype == np.uint8 and buf.ndim == 1 and buf.flags["C_CONTIGUOUS"]
    p = _pix(pixel)
    need = (height - 1) * stride + width * BYTES[p] + offset
    assert buf.size >= need, (buf.size, need)
    return OracleImage(buf.ctypes.data + offset, stride, width, height, p)


def resize(src_buf, width, height, pixel, dst_width, dst_height, filter="cubic", filter_scale=0.70,
           stride=None, offset=0, dst_stride=None):
    """Returns (dst_buf uint8 1-D of dst_height*dst_stride bytes, dst_stride). Padding bytes = 0xCD."""
    p = _pix(pixel)
    stride = stride if stride is not None else row_stride(width, p)
    dst_stride = dst_stride if dst_stride is not None else row_stride(dst_width, p)
    dst = np.full(dst_height * dst_stride, 0xCD, dtype=np.uint8)
    s = _desc(src_buf, width, height, p, stride, offset)
    d = _desc(dst, dst_width, dst_height, p, dst_stride)
    rc = lib().oracle_resize(ctypes.byref(s), ctypes.byref(d), _filt(filter), ctypes.c_float(filter_scale))
    if rc:
        raise RuntimeError("oracle_resize rc=%d" % rc)
    return dst, dst_stride


def color_settings(r=float("nan"), g=float("nan"), b=float("nan")):
    out = (ctypes.c_float * 3)()
    lib().oracle_color_settings(r, g, b, out)
    return float(out[0]), float(out[1]), float(out[2])


def color_convert(src_buf, width, height, pixel, dst_pixel, weights=None, stride=None, offset=0,
                  dst_stride=None):
    p, q = _pix(pixel), _pix(dst_pixel)
    stride = stride if stride is not None else row_stride(width, p)
    dst_stride = dst_stride if dst_stride is not None else row_stride(width, q)
    r, g, b = weights if weights is not None else color_settings()
    dst = np.full(height * dst_stride, 0xCD, dtype=np.uint8)
    s = _desc(src_buf, width, height, p, stride, offset)
    d = _desc(dst, width, height, q, dst_stride)
    rc = lib().oracle_color_convert(ctypes.byref(s), ctypes.byref(d), r, g, b)
    if rc:
        raise RuntimeError("oracle_color_convert rc=%d" % rc)
    return dst, dst_stride


def make_contribs(filter, filter_scale, src_size, dst_size, max_taps=None):
    """Returns (left[int32], right[int32], weights[dst_size, T] float32)."""
    if max_taps is None:
        max_taps = 8
        while True:
            left = np.zeros(dst_size, np.int32)
            right = np.zeros(dst_size, np.int32)
            w = np.zeros((dst_size, max_taps), np.float32)
            rc = lib().oracle_make_contribs(_filt(filter), filter_scale, src_size, dst_size,
                                            left.ctypes.data, right.ctypes.data, w.ctypes.data, max_taps)
            if rc == -10:
                max_taps *= 2
                continue
            break
    else:
        left = np.zeros(dst_size, np.int32)
        right = np.zeros(dst_size, np.int32)
        w = np.zeros((dst_size, max_taps), np.float32)
        rc = lib().oracle_make_contribs(_filt(filter), filter_scale, src_size, dst_size,
                                        left.ctypes.data, right.ctypes.data, w.ctypes.data, max_taps)
    if rc < 0:
        raise RuntimeError("oracle_make_contribs rc=%d" % rc)
    return left, right, w[:, :rc].copy()


def resize_batch_mt(srcs, width, height, pixel, dst_width, dst_height, filter, filter_scale, threads):
    """srcs: list of uint8 1-D arrays (default stride). Returns list of dst arrays."""
    p = _pix(pixel)
    n = len(srcs)
    st, dstst = row_stride(width, p), row_stride(dst_width, p)
    dsts = [np.empty(dst_height * dstst, np.uint8) for _ in range(n)]
    S = (OracleImage * n)(*[_desc(b, width, height, p, st) for b in srcs])
    D = (OracleImage * n)(*[_desc(b, dst_width, dst_height, p, dstst) for b in dsts])
    rc = lib().oracle_resize_batch_mt(n, S, D, _filt(filter), filter_scale, threads)
    if rc:
        raise RuntimeError("oracle_resize_batch_mt rc=%d" % rc)
    return dsts
