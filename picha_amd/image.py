"""`Image` — host-side mirror of the reference's Image class (reference lib/image.js:3-96).

Same fields {width, height, stride, pixel, data}, same constructor defaults and error strings,
same helper names (row, equalPixels, avgChannelDiff, subView, copy, pixelSize).  `data` is a 1-D
numpy uint8 array (the analogue of a Node Buffer; slices share memory like Buffer.slice).

Deliberate deviation: the reference's size table has the typo 'r16b16' for 'r16g16'
(lib/image.js:31), which makes `new Image({pixel: 'r16g16'})` throw; here 'r16g16' is accepted.
"""
import numpy as np

from ._native import PIXEL_BYTES


class Image:
    def __init__(self, opt=None, **kw):
        opt = dict(opt or {}, **kw)
        self.data = opt.get("data")
        self.width = opt.get("width") or 0
        self.height = opt.get("height") or 0
        self.pixel = opt.get("pixel") or "rgba"
        psize = Image.pixelSize(self.pixel)
        self.stride = opt.get("stride") or ((self.width * psize + 3) & ~3)
        if psize == 0:
            raise ValueError("invalid pixel format " + str(self.pixel))
        if self.stride < self.width * psize:
            raise ValueError("stride too short")
        if self.width < 0 or self.height < 0:
            raise ValueError("invalid dimensions")
        if self.data is not None and not isinstance(self.data, np.ndarray):
            self.data = np.frombuffer(self.data, dtype=np.uint8)
        if self.data is not None:
            self.data = self.data.reshape(-1).view(np.uint8)
        if self.stride * self.height != 0 and self.data is None:
            self.data = np.zeros(self.stride * self.height, dtype=np.uint8)  # Buffer.alloc: zero filled
        if self.data is not None and self.data.size < self.stride * (self.height - 1) + self.width * psize:
            raise ValueError("image data too small")

    @staticmethod
    def pixelSize(pixel=None):
        return PIXEL_BYTES.get(pixel, 0)

    def pixel_size(self):
        return PIXEL_BYTES.get(self.pixel, 0)

    def row(self, y):
        return self.data[y * self.stride: y * self.stride + self.width * self.pixel_size()]

    def equalPixels(self, o):
        """Payload bytes only: row padding is never compared (lib/image.js:57-64)."""
        if self.width != o.width or self.height != o.height or self.pixel != o.pixel:
            return False
        for y in range(self.height):
            if not np.array_equal(self.row(y), o.row(y)):
                return False
        return True

    def avgChannelDiff(self, o):
        """Mean absolute BYTE difference over payload bytes (lib/image.js:66-74)."""
        if self.width != o.width or self.height != o.height or self.pixel != o.pixel:
            return 255
        rw = self.width * self.pixel_size()
        s = 0
        for y in range(self.height):
            s += int(np.abs(self.row(y).astype(np.int64) - o.row(y).astype(np.int64)).sum())
        return s / (self.height * rw)

    def subView(self, x, y, w, h):
        p = self.pixel_size()
        off = y * self.stride + x * p
        length = (h - 1) * self.stride + w * p
        return Image(width=w, height=h, pixel=self.pixel, stride=self.stride, data=self.data[off: off + length])

    def copy(self, target):
        if target.pixel != self.pixel:
            raise ValueError("can't copy pixels between different pixel types")
        rw = self.pixel_size() * min(self.width, target.width)
        h = min(self.height, target.height)
        for y in range(h):
            target.data[y * target.stride: y * target.stride + rw] = self.data[y * self.stride: y * self.stride + rw]

    def pixels(self):
        """Payload as an ndarray [height, width, channels] of u8/u16 (a copy; convenience for tests)."""
        from ._native import PIXEL_CHANNELS
        ch = PIXEL_CHANNELS[self.pixel]
        p = self.pixel_size()
        rows = np.stack([self.row(y) for y in range(self.height)]) if self.height else np.zeros((0, 0), np.uint8)
        if p // ch == 2:
            return rows.copy().view(np.uint16).reshape(self.height, self.width, ch)
        return rows.reshape(self.height, self.width, ch).copy()
