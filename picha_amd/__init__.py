"""picha_amd — MI355X (gfx950) implementation of picha's pixel hot path.

Drop-in for `picha.resize / resizeSync / colorConvert / colorConvertSync` over
`Image {width, height, stride, pixel, data}`; everything else of picha (codecs) is out of scope.
All pixels are computed by hand-written HIP kernels in libpicha_hip.so (C ABI:
include/picha_hip.h).  There is no CPU fallback.
"""
from ._native import FILTERS, PIXELS, PichaError, lib  # noqa: F401
from .api import (  # noqa: F401
    colorConvert,
    colorConvertSync,
    convertResize,
    convertResizeSync,
    color_settings,
    drain,
    make_contribs,
    resize,
    resizeSync,
)
from .image import Image  # noqa: F401

catalog = {}  # the reference exports its codec catalog here (index.js:8); codecs are out of scope
