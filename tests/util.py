"""Shared helpers for the parity tests: seeded inputs, oracle calls, comparisons."""
import os

import numpy as np

import oracle
from picha_amd import Image
from picha_amd import _native as N

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PIXELS = N.PIXELS
FILTERS = N.FILTERS


def rand_image(rng, width, height, pixel, pad=0, edge_values=False):
    """Random image with `pad` extra bytes of stride; padding bytes are random too."""
    bpp = N.PIXEL_BYTES[pixel]
    stride = ((bpp * width + 3) & ~3) + pad
    data = rng.integers(0, 256, size=height * stride, dtype=np.uint8)
    if edge_values:
        # saturate: alternating 0 / max samples exercise the clamp with negative-lobe filters
        v = rng.choice(np.array([0, 255], dtype=np.uint8), size=height * stride)
        if pixel.startswith("r16"):
            v = np.repeat(rng.choice(np.array([0, 255], dtype=np.uint8), size=(height * stride + 1) // 2), 2)[: height * stride]
        data = v.astype(np.uint8)
    return Image(width=width, height=height, pixel=pixel, stride=stride, data=data)


def payload(img):
    bpp = N.PIXEL_BYTES[img.pixel]
    return np.stack([img.data[y * img.stride: y * img.stride + img.width * bpp] for y in range(img.height)])


def oracle_resize(img, w, h, filter="cubic", filter_scale=0.70):
    buf, st = oracle.resize(np.ascontiguousarray(img.data), img.width, img.height, img.pixel, w, h, filter, filter_scale,
                            stride=img.stride)
    return Image(width=w, height=h, pixel=img.pixel, stride=st, data=buf)


def oracle_convert(img, pixel, weights=None):
    buf, st = oracle.color_convert(np.ascontiguousarray(img.data), img.width, img.height, img.pixel, pixel,
                                   weights=weights, stride=img.stride)
    return Image(width=img.width, height=img.height, pixel=pixel, stride=st, data=buf)


def max_code_diff(a, b):
    """Largest per-sample difference in code values (u8 or u16 samples)."""
    pa, pb = a.pixels().astype(np.int64), b.pixels().astype(np.int64)
    return int(np.abs(pa - pb).max()) if pa.size else 0
