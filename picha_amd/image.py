"""`Image` — the raw pixel container of the picha API on the Python side: {width, height, stride, pixel, data}.

Own implementation of the behaviour of the reference's Image class (constructor defaults and the four error
strings: lib/image.js:3-21; helpers: lib/image.js:34-96), written around numpy: `data` is a 1-D uint8 array
(the analogue of a Node Buffer — slices share memory like Buffer.slice) and every helper works on ONE strided
2-D view of the payload bytes instead of per-row loops.

Deliberate deviation: the reference's size table spells 'r16g16' as 'r16b16' (lib/image.js:31), which makes
`new Image({pixel: 'r16g16'})` throw; this table has the right key.
"""
from dataclasses import dataclass

import numpy as np

from ._native import PIXEL_BYTES, PIXEL_CHANNELS


@dataclass(frozen=True)
class _Layout:
    """Geometry of an image request after the API defaults (falsy width / height -> 0, pixel -> 'rgba', rows padded to
    4 bytes) and the checks the reference makes, in its order and with its messages."""
    width: int
    height: int
    pixel: str
    stride: int
    bpp: int

    @staticmethod
    def resolve(req):
        width, height = req.get("width") or 0, req.get("height") or 0
        pixel = req.get("pixel") or "rgba"
        bpp = PIXEL_BYTES.get(pixel, 0)
        stride = req.get("stride") or ((width * bpp + 3) & ~3)
        for bad, message in ((bpp == 0, "invalid pixel format " + str(pixel)),
                             (stride < width * bpp, "stride too short"),
                             (width < 0 or height < 0, "invalid dimensions")):
            if bad:
                raise ValueError(message)
        return _Layout(width, height, pixel, stride, bpp)

    @property
    def payload(self):          # bytes of a row that carry pixels
        return self.width * self.bpp

    @property
    def alloc_bytes(self):      # a freshly allocated image: whole rows
        return self.stride * self.height

    @property
    def min_bytes(self):        # a caller's buffer may stop at the last row's payload
        return self.stride * (self.height - 1) + self.payload


def _as_bytes(buf):
    """Any buffer-like thing as a flat uint8 array over the SAME memory."""
    arr = buf if isinstance(buf, np.ndarray) else np.frombuffer(buf, dtype=np.uint8)
    return arr.reshape(-1).view(np.uint8)


class Image:
    def __init__(self, opt=None, **kw):
        lay = _Layout.resolve(dict(opt or {}, **kw))
        self.width, self.height, self.pixel, self.stride = lay.width, lay.height, lay.pixel, lay.stride
        given = dict(opt or {}, **kw).get("data")
        store = _as_bytes(given) if given is not None else None
        if store is None and lay.alloc_bytes != 0:
            store = np.zeros(lay.alloc_bytes, dtype=np.uint8)      # zero filled, like Buffer.alloc
        if store is not None and store.size < lay.min_bytes:
            raise ValueError("image data too small")
        self.data = store

    # ---- sizes -------------------------------------------------------------------------------------
    @staticmethod
    def pixelSize(pixel=None):
        return PIXEL_BYTES.get(pixel, 0)

    def pixel_size(self):
        return PIXEL_BYTES.get(self.pixel, 0)

    # ---- views -------------------------------------------------------------------------------------
    def _payload(self, width=None, height=None):
        """[rows, payload bytes] view of the pixel bytes (no padding, no copy); optionally only the top-left corner."""
        w = self.width if width is None else width
        h = self.height if height is None else height
        nbytes = w * self.pixel_size()
        if h == 0 or nbytes == 0:
            return np.zeros((h, nbytes), dtype=np.uint8)
        return np.lib.stride_tricks.as_strided(self.data, shape=(h, nbytes), strides=(self.stride, 1))

    def row(self, y):
        start = y * self.stride
        return self.data[start: start + self.width * self.pixel_size()]

    def subView(self, x, y, w, h):
        """A window into the same memory: same stride, data starting at the window's first pixel."""
        bpp = self.pixel_size()
        first = y * self.stride + x * bpp
        return Image(width=w, height=h, pixel=self.pixel, stride=self.stride, data=self.data[first: first + (h - 1) * self.stride + w * bpp])

    # ---- comparisons (payload bytes only: row padding is never looked at) -----------------------------
    def _same_shape(self, other):
        return (self.width, self.height, self.pixel) == (other.width, other.height, other.pixel)

    def equalPixels(self, other):
        return self._same_shape(other) and bool(np.array_equal(self._payload(), other._payload()))

    def avgChannelDiff(self, other):
        """Mean absolute BYTE difference; 255 for images of different shape (the reference's sentinel)."""
        if not self._same_shape(other):
            return 255
        a, b = self._payload().astype(np.int16), other._payload().astype(np.int16)
        return int(np.abs(a - b).sum(dtype=np.int64)) / (self.height * self.width * self.pixel_size())

    # ---- copies ------------------------------------------------------------------------------------
    def copy(self, target):
        """Copies the common top-left rectangle into `target` (payload only)."""
        if target.pixel != self.pixel:
            raise ValueError("can't copy pixels between different pixel types")
        w, h = min(self.width, target.width), min(self.height, target.height)
        if w and h:
            target._payload(w, h)[...] = self._payload(w, h)

    def pixels(self):
        """Payload as an ndarray [height, width, channels] of u8 / u16 (a copy; convenience for tests)."""
        ch = PIXEL_CHANNELS[self.pixel]
        flat = np.ascontiguousarray(self._payload())
        if self.pixel_size() // ch == 2:
            flat = flat.view(np.uint16)
        return flat.reshape(self.height, self.width, ch)
