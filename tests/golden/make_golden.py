#!/usr/bin/env python3
"""Regenerate tests/golden/*.npy from the data files the reference's own tests hold.

Run in the build container only (needs /root/reference and Pillow):
    python tests/golden/make_golden.py

The fixtures are DATA (decoded pixels), not reference source:
  test_rgba.npy      <- test/test.png      50x50 rgba   (input of test/color_convert.js:16-19)
  greytest_greya.npy <- test/greytest.png  50x50 greya  (expected output, test/color_convert.js:22-29)
  test2_rgb.npy      <- test/test2.jpg     76x50 rgb, decoded with Pillow/libjpeg-turbo
                        (input of test/resize.js:16-17)
  test2_small_rgb.npy<- test/test2.png     32x24 rgb    (expected output, test/resize.js:20-31)
"""
import os
import sys

import numpy as np
from PIL import Image

REF = os.environ.get("PICHA_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name, mode):
    im = Image.open(os.path.join(REF, "test", name))
    assert im.mode == mode, (name, im.mode)
    return np.ascontiguousarray(np.asarray(im))


def main():
    out = {
        "test_rgba.npy": load("test.png", "RGBA"),
        "greytest_greya.npy": load("greytest.png", "LA"),
        "test2_rgb.npy": load("test2.jpg", "RGB"),
        "test2_small_rgb.npy": load("test2.png", "RGB"),
    }
    for k, v in out.items():
        np.save(os.path.join(HERE, k), v)
        print(k, v.shape, v.dtype)
    return 0


if __name__ == "__main__":
    sys.exit(main())
