"""Device-resident batches: images that already live in one GPU's HBM.

torch is used here only as plumbing — it owns the device memory (uint8 tensors) and the HIP
stream; every pixel is produced by libpicha_hip.so through the *_device entry points of the C
ABI (include/picha_hip.h).  This is the path bench.py times and the path a service would use
when decode and encode also run on the GPU.
"""
import ctypes

import torch

from . import _native as N


def row_stride(width, pixel):
    return (N.PIXEL_BYTES[pixel] * width + 3) & ~3


class DeviceBatch:
    """n equal-geometry images in one uint8 tensor [n, pitch] on a GPU (pitch >= height*stride)."""

    def __init__(self, n, width, height, pixel, device="cuda", stride=None, tensor=None):
        if N.pixel_index(pixel) < 0:
            raise N.PichaError("invalid image")
        self.n, self.width, self.height, self.pixel = int(n), int(width), int(height), pixel
        self.stride = int(stride) if stride else row_stride(width, pixel)
        # image pitch padded to 256 B so every image starts aligned for the widest loads
        self.pitch = (self.height * self.stride + 255) // 256 * 256
        if tensor is None:
            tensor = torch.empty((self.n, self.pitch), dtype=torch.uint8, device=device)
        assert tensor.dtype == torch.uint8 and tensor.is_contiguous() and tuple(tensor.shape) == (self.n, self.pitch)
        self.tensor = tensor
        self._desc = None

    @property
    def payload_bytes(self):
        return self.n * self.height * self.width * N.PIXEL_BYTES[self.pixel]

    def descriptors(self):
        """ctypes array of picha_image with DEVICE pointers (cached; tensor storage must not move)."""
        if self._desc is None or self._desc[1] != self.tensor.data_ptr() + getattr(self, "_offset", 0):
            base = self.tensor.data_ptr() + getattr(self, "_offset", 0)
            arr = (N.CImage * self.n)()
            p = N.pixel_index(self.pixel)
            for i in range(self.n):
                arr[i] = N.CImage(base + i * self.pitch, self.stride, self.width, self.height, p)
            self._desc = (arr, base)
        return self._desc[0]

    def fill_random(self, seed):
        """Synthetic pixels (SURVEY.md 8d generator), generated on the device."""
        sample = 2 if self.pixel.startswith("r16") else 1
        stream = torch.cuda.current_stream(self.tensor.device).cuda_stream
        with torch.cuda.device(self.tensor.device):
            N.check(N.lib().picha_hip_fill_random(self.tensor.data_ptr(), self.tensor.numel(), seed, sample, stream))
        return self

    def image_host(self, i):
        """Download image i as a host `Image`."""
        from .image import Image
        off = getattr(self, "_offset", 0)
        span = (self.height - 1) * self.stride + self.width * N.PIXEL_BYTES[self.pixel]
        data = self.tensor[i, off : off + span].cpu().numpy().copy()
        return Image(width=self.width, height=self.height, pixel=self.pixel, stride=self.stride, data=data)

    # ---- Image helpers on the device (reference lib/image.js:57-96) -------------------------------------
    def sub_view(self, x, y, w, h):
        """Image.prototype.subView for every image of the batch: same storage and stride, offset origin."""
        v = DeviceBatch.__new__(DeviceBatch)
        v.n, v.width, v.height, v.pixel, v.stride, v.pitch = self.n, int(w), int(h), self.pixel, self.stride, self.pitch
        v.tensor, v._desc = self.tensor, None
        v._offset = getattr(self, "_offset", 0) + y * self.stride + x * N.PIXEL_BYTES[self.pixel]
        return v

    def _compare(self, other):
        if self.n != other.n:
            raise N.PichaError("batches differ in length")
        s = (ctypes.c_uint64 * self.n)()
        c = (ctypes.c_uint64 * self.n)()
        with torch.cuda.device(self.tensor.device):
            N.check(N.lib().picha_hip_image_compare_device(self.n, self.descriptors(), other.descriptors(), s, c,
                                                           _stream(self.tensor)))
        return list(s), list(c)

    def equal_pixels(self, other):
        """Image.prototype.equalPixels per image -> list of bool (False where geometry or format differ)."""
        _, c = self._compare(other)
        return [v == 0 for v in c]

    def avg_channel_diff(self, other):
        """Image.prototype.avgChannelDiff per image -> list of float (255 where geometry or format differ)."""
        s, _ = self._compare(other)
        rw = self.width * N.PIXEL_BYTES[self.pixel]
        return [255 if v == 0xFFFFFFFFFFFFFFFF else v / (self.height * rw) for v in s]

    def copy_to(self, target):
        """Image.prototype.copy per image: the common top-left rectangle, payload bytes only."""
        if self.n != target.n:
            raise N.PichaError("batches differ in length")
        with torch.cuda.device(self.tensor.device):
            N.check(N.lib().picha_hip_image_copy_device(self.n, self.descriptors(), target.descriptors(), _stream(self.tensor)))
        return target

    def upload(self, i, image):
        assert (image.width, image.height, image.pixel, image.stride) == (self.width, self.height, self.pixel, self.stride)
        n = (self.height - 1) * self.stride + self.width * N.PIXEL_BYTES[self.pixel]
        self.tensor[i, :n].copy_(torch.from_numpy(image.data[:n]))


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def resize_batch(src, dst_width=None, dst_height=None, filter=None, filter_scale=None, out=None, opts=None, numerics=None):
    """picha.resize over a device-resident batch; same option semantics as resizeSync.

    Asynchronous on torch's current stream.  Returns the destination DeviceBatch.
    """
    from .api import _resize_options
    o = dict(opts or {})
    if filter is not None:
        o["filter"] = filter
    if filter_scale is not None:
        o["filterScale"] = filter_scale
    ftag, fscale = _resize_options(o)
    if out is None:
        w = dst_width if dst_width is not None else o.get("width")
        h = dst_height if dst_height is not None else o.get("height")
        if not w or not h or w <= 0 or h <= 0:
            raise N.PichaError("invalid dimensions")
        out = DeviceBatch(src.n, w, h, src.pixel, device=src.tensor.device)
    with torch.cuda.device(src.tensor.device):
        if numerics is None:
            N.check(N.lib().picha_hip_resize_device(src.n, src.descriptors(), out.descriptors(), ftag, fscale, _stream(src.tensor)))
        else:   # per-call numerics flavour ("exact" / "fma")
            import ctypes
            co = N.CCallOpts(numerics)
            N.check(N.lib().picha_hip_resize_device_ex(src.n, src.descriptors(), out.descriptors(), ftag, fscale, ctypes.byref(co), _stream(src.tensor)))
    return out


def color_convert_batch(src, pixel, weights=None, out=None):
    """picha.colorConvert over a device-resident batch (weights = normalised (r, g, b) or None)."""
    from .api import color_settings
    if N.pixel_index(pixel) < 0:
        raise N.PichaError("expected pixel mode")
    r, g, b = weights if weights is not None else color_settings()
    if out is None:
        out = DeviceBatch(src.n, src.width, src.height, pixel, device=src.tensor.device)
    with torch.cuda.device(src.tensor.device):
        N.check(N.lib().picha_hip_color_convert_device(src.n, src.descriptors(), out.descriptors(), r, g, b, _stream(src.tensor)))
    return out


def convert_resize_batch(src, pixel, dst_width, dst_height, weights=None, filter=None, filter_scale=None, order=0, out=None):
    """colorConvert and resize chained on the device (picha_hip_convert_resize_device): order 0 = convert first."""
    from .api import _resize_options, color_settings
    if N.pixel_index(pixel) < 0:
        raise N.PichaError("expected pixel mode")
    o = {}
    if filter is not None:
        o["filter"] = filter
    if filter_scale is not None:
        o["filterScale"] = filter_scale
    ftag, fscale = _resize_options(o)
    r, g, b = weights if weights is not None else color_settings()
    if out is None:
        out = DeviceBatch(src.n, dst_width, dst_height, pixel, device=src.tensor.device)
    with torch.cuda.device(src.tensor.device):
        N.check(N.lib().picha_hip_convert_resize_device(src.n, src.descriptors(), out.descriptors(), int(order), r, g, b,
                                                        ftag, fscale, _stream(src.tensor)))
    return out


class ResizePlan:
    """Pre-marshalled resize of one batch into another: run() is a single C call (bench hot loop)."""

    def __init__(self, src, dst, filter=None, filter_scale=None):
        from .api import _resize_options
        o = {}
        if filter is not None:
            o["filter"] = filter
        if filter_scale is not None:
            o["filterScale"] = filter_scale
        self.ftag, self.fscale = _resize_options(o)
        self.src, self.dst = src, dst
        self._s, self._d = src.descriptors(), dst.descriptors()
        self._fn = N.lib().picha_hip_resize_device

    def run(self, stream=None):
        st = stream if stream is not None else _stream(self.src.tensor)
        rc = self._fn(self.src.n, self._s, self._d, self.ftag, self.fscale, st)
        if rc:
            N.check(rc)


def place_destination(make_dst, make_plan, src, candidates=4, launches=6):
    """Choose where a destination batch lives by measurement (r05).

    The time of a launch depends on where its destination sits in HBM relative to its source — by 3-7 % between buffers allocated one after the other
    (profiles/r04_placement.txt, profiles/r05_placement.txt); nothing in the address predicts it, so a caller that allocates its output arena once can try a few:
    `candidates` batches from make_dst() are all allocated (so each lands somewhere else), make_plan(dst).run() is timed into each, the fastest is kept and the
    others are freed.  Returns (dst, plan, record) — record = {"candidates": [[offset from src in MB, ms per launch], ...], "kept": index}."""
    dsts = [make_dst() for _ in range(max(1, candidates))]
    plans = [make_plan(d) for d in dsts]
    if len(dsts) == 1:
        return dsts[0], plans[0], {"candidates": [[(dsts[0].tensor.data_ptr() - src.tensor.data_ptr()) >> 20, None]], "kept": 0}
    for p in plans:      # tables built, clocks up
        p.run()
    torch.cuda.synchronize(src.tensor.device)
    best = [float("inf")] * len(plans)
    for _ in range(2):   # two interleaved rounds: a slow first round (clock ramp) cannot decide
        for i, p in enumerate(plans):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            p.run()
            e0.record()
            for _ in range(launches):
                p.run()
            e1.record()
            torch.cuda.synchronize(src.tensor.device)
            best[i] = min(best[i], e0.elapsed_time(e1) / launches)
    kept = min(range(len(plans)), key=lambda i: best[i])
    record = {"candidates": [[(d.tensor.data_ptr() - src.tensor.data_ptr()) >> 20, round(ms, 4)] for d, ms in zip(dsts, best)], "kept": kept}
    dst, plan = dsts[kept], plans[kept]
    del dsts, plans
    torch.cuda.empty_cache()
    return dst, plan, record
