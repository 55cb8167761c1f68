"""The CPU oracle against every golden vector the reference's own tests hold for the hot path
(SURVEY.md 8c).  These pin the oracle; everything the GPU tests assert is relative to it."""
import numpy as np

import oracle
from util import GOLDEN


def test_color_convert_rgba_to_greya_matches_reference_fixture():
    """reference test/color_convert.js:22-29: colorConvertSync(test.png, greya) equalPixels greytest.png"""
    rgba = np.load(GOLDEN + "/test_rgba.npy")
    grey = np.load(GOLDEN + "/greytest_greya.npy")
    out, st = oracle.color_convert(rgba.reshape(-1), 50, 50, "rgba", "greya")
    assert st == 100
    assert np.array_equal(out.reshape(50, 50, 2), grey)


def test_grey_color_grey_is_identity():
    """reference test/color_convert.js:30-40"""
    grey = np.load(GOLDEN + "/greytest_greya.npy")
    rgba, _ = oracle.color_convert(np.ascontiguousarray(grey).reshape(-1), 50, 50, "greya", "rgba")
    back, _ = oracle.color_convert(rgba, 50, 50, "rgba", "greya")
    assert np.array_equal(back.reshape(50, 50, 2), grey)


def test_resize_test2_matches_reference_fixture():
    """reference test/resize.js:20-31: resize(test2.jpg, 32x24) vs test2.png; the reference demands
    avgChannelDiff < 2, the restatement reproduces the stored result exactly."""
    src = np.load(GOLDEN + "/test2_rgb.npy")
    small = np.load(GOLDEN + "/test2_small_rgb.npy")
    out, st = oracle.resize(src.reshape(-1), 76, 50, "rgb", 32, 24)  # default options: cubic @ 0.70
    got = out.reshape(24, st)[:, :96].reshape(24, 32, 3)
    assert np.abs(got.astype(int) - small.astype(int)).mean() < 2
    assert np.array_equal(got, small)


def test_default_color_weights_normalise_to_themselves():
    """getSettings always normalises; for the defaults n == 1.0f exactly."""
    r, g, b = oracle.color_settings()
    assert (np.float32(r), np.float32(g), np.float32(b)) == (np.float32(0.299), np.float32(0.587), np.float32(0.114))


def test_tap_tables_of_the_baseline_configs():
    """SURVEY.md a6 [probed]: cfg2 default has 3-5 taps and the stated interior weights."""
    left, right, w = oracle.make_contribs("cubic", 0.70, 3840, 1920)
    n = right - left + 1
    assert n.min() == 3 and n.max() == 5
    interior = w[100]
    expect = np.array([0.0704663, 0.251813, 0.35544, 0.251813, 0.0704663], np.float32)
    assert np.allclose(interior, expect, rtol=2e-6)
    left, right, w = oracle.make_contribs("cubic", 1.0, 3840, 1920)
    n = right - left + 1
    assert n.min() == 4 and n.max() == 7
    left, right, w = oracle.make_contribs("mitchel", 1.0, 7680, 1920)
    assert (right - left + 1).max() == 15
