"""ctypes binding of libpicha_hip.so (include/picha_hip.h).

The library is the product: there is no Python or CPU fallback.  Importing this module fails
loudly when the shared object is missing, and every compute call fails with PichaError when no
gfx950 device is usable.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PICHA_HIP_LIB: load another build of the same library (A/B measurements of two builds in one session)
SO_PATH = os.environ.get("PICHA_HIP_LIB") or os.path.join(HERE, "libpicha_hip.so")

PIXELS = ["rgb", "rgba", "grey", "greya", "r16", "r16g16", "r16g16b16", "r16g16b16a16"]
FILTERS = ["cubic", "lanczos", "catmulrom", "mitchel", "box", "triangle"]
PIXEL_BYTES = {"rgb": 3, "rgba": 4, "grey": 1, "greya": 2, "r16": 2, "r16g16": 4, "r16g16b16": 6, "r16g16b16a16": 8}
PIXEL_CHANNELS = {"rgb": 3, "rgba": 4, "grey": 1, "greya": 2, "r16": 1, "r16g16": 2, "r16g16b16": 3, "r16g16b16a16": 4}


class PichaError(Exception):
    """Raised with the reference's message strings (e.g. "invalid image", "invalid dimensions")."""

    def __init__(self, message, status=None):
        super().__init__(message)
        self.status = status


class CImage(ctypes.Structure):
    """struct picha_image == NativeImage (reference src/picha.h:202-218)."""

    _fields_ = [
        ("data", ctypes.c_void_p),
        ("stride", ctypes.c_int32),
        ("width", ctypes.c_int32),
        ("height", ctypes.c_int32),
        ("pixel", ctypes.c_int32),
    ]


# every symbol include/picha_hip.h declares: (name, restype, argtypes)
class CCallOpts(ctypes.Structure):
    """picha_call_opts (include/picha_hip.h): per-call options; numerics -1 = process default, 0 exact, 1 fma."""
    _fields_ = [("size", ctypes.c_int32), ("numerics", ctypes.c_int32)]

    def __init__(self, numerics=-1):
        super().__init__(ctypes.sizeof(CCallOpts), {"exact": 0, "fma": 1}.get(numerics, numerics))


_P = ctypes.POINTER(CImage)
_i, _f, _d, _vp, _sz = ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t
SYMBOLS = [
    ("picha_pixel_bytes", _i, [_i]),
    ("picha_pixel_channels", _i, [_i]),
    ("picha_row_stride", _i, [_i, _i]),
    ("picha_pixel_from_name", _i, [ctypes.c_char_p]),
    ("picha_pixel_name", ctypes.c_char_p, [_i]),
    ("picha_filter_from_name", _i, [ctypes.c_char_p]),
    ("picha_filter_name", ctypes.c_char_p, [_i]),
    ("picha_resize_options", _i, [_i, _i, _i, _d, ctypes.POINTER(_i), ctypes.POINTER(_f)]),
    ("picha_color_settings", None, [_d, _d, _d, ctypes.POINTER(_f)]),
    ("picha_hip_make_contribs", _i, [_i, _f, _i, _i, _vp, _vp, _vp, _i, ctypes.POINTER(_i)]),
    ("picha_hip_last_error", ctypes.c_char_p, []),
    ("picha_hip_strerror", ctypes.c_char_p, [_i]),
    ("picha_hip_device_count", _i, []),
    ("picha_hip_device_arch", _i, [ctypes.c_char_p, _sz]),
    ("picha_hip_set_option", _i, [ctypes.c_char_p, ctypes.c_long]),
    ("picha_hip_get_option", ctypes.c_long, [ctypes.c_char_p]),
    ("picha_hip_last_kernel", ctypes.c_char_p, []),
    ("picha_hip_resize", _i, [_P, _P, _i, _f]),
    ("picha_hip_color_convert", _i, [_P, _P, _f, _f, _f]),
    ("picha_hip_resize_batch", _i, [_i, _P, _P, _i, _f]),
    ("picha_hip_resize_ex", _i, [_P, _P, _i, _f, _vp]),
    ("picha_hip_resize_batch_ex", _i, [_i, _P, _P, _i, _f, _vp]),
    ("picha_hip_color_convert_batch", _i, [_i, _P, _P, _f, _f, _f]),
    ("picha_hip_resize_async", _i, [_P, _P, _i, _f, _vp, _vp]),
    ("picha_hip_resize_async_ex", _i, [_P, _P, _i, _f, _vp, _vp, _vp]),
    ("picha_hip_color_convert_async", _i, [_P, _P, _f, _f, _f, _vp, _vp]),
    ("picha_hip_async_drain", _i, []),
    ("picha_hip_async_stats", None, [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    ("picha_hip_async_dispatcher_stats", _i, [_i, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    ("picha_hip_resize_device", _i, [_i, _P, _P, _i, _f, _vp]),
    ("picha_hip_color_convert_device", _i, [_i, _P, _P, _f, _f, _f, _vp]),
    ("picha_hip_resize_device_ex", _i, [_i, _P, _P, _i, _f, _vp, _vp]),
    ("picha_hip_convert_resize_device", _i, [_i, _P, _P, _i, _f, _f, _f, _i, _f, _vp]),
    ("picha_hip_convert_resize", _i, [_P, _P, _i, _f, _f, _f, _i, _f]),
    ("picha_hip_convert_resize_batch", _i, [_i, _P, _P, _i, _f, _f, _f, _i, _f]),
    ("picha_hip_convert_resize_async", _i, [_P, _P, _i, _f, _f, _f, _i, _f, _vp, _vp]),
    ("picha_hip_image_compare_device", _i, [_i, _P, _P, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), _vp]),
    ("picha_hip_image_copy_device", _i, [_i, _P, _P, _vp]),
    ("picha_hip_malloc", _i, [ctypes.POINTER(_vp), _sz]),
    ("picha_hip_free", _i, [_vp]),
    ("picha_hip_malloc_placed", _i, [ctypes.POINTER(_vp), _sz, _i, _P, _P, _i, _f, _i, ctypes.POINTER(_f)]),
    ("picha_hip_memcpy_h2d", _i, [_vp, _vp, _sz, _vp]),
    ("picha_hip_memcpy_d2h", _i, [_vp, _vp, _sz, _vp]),
    ("picha_hip_stream_synchronize", _i, [_vp]),
    ("picha_hip_fill_random", _i, [_vp, _sz, ctypes.c_uint64, _i, _vp]),
    ("picha_fill_random_host", None, [_vp, _sz, ctypes.c_uint64, _i]),
]

DONE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p)

_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.

    PyTorch wheels carry their own libamdhip64 / libhsa-runtime64 and load them by path; a process that has already
    loaded the system runtime (through libpicha_hip.so) and then imports torch ends up with two runtimes and torch
    sees no GPU.  torch is only plumbing here (device-resident batches, bench), but when it is installed we let ITS
    runtime be the one: load it first, without importing torch; libpicha_hip.so then binds to the same copy.
    """
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("PICHA_HIP_SYSTEM_RUNTIME") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libpicha_hip.so (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                "picha_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C picha_amd/csrc`. There is no fallback implementation." % SO_PATH
            )
        _preload_torch_hip_runtime()
        L = ctypes.CDLL(SO_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != 0:
        L = lib()
        msg = L.picha_hip_last_error().decode() or L.picha_hip_strerror(status).decode()
        raise PichaError(msg, status)


def pixel_index(name):
    try:
        return PIXELS.index(name)
    except ValueError:
        return -1
