"""Host-side mirror of the reference interface: Image semantics, option marshalling, error strings,
and the product's tap builder against the oracle.  No GPU needed."""
import math

import numpy as np
import pytest

import oracle
import picha_amd
from picha_amd import Image, PichaError
from picha_amd import _native as N
from picha_amd.shard import shard_range
from util import FILTERS


# ---- Image (reference lib/image.js:3-96) ----------------------------------------------------------
def test_image_defaults_and_zero_fill():
    im = Image(width=5, height=3)
    assert im.pixel == "rgba" and im.stride == 20 and im.data.size == 60 and not im.data.any()
    im = Image(width=5, height=3, pixel="rgb")
    assert im.stride == 16  # (15 + 3) & ~3


@pytest.mark.parametrize("kw,msg", [
    (dict(width=2, height=2, pixel="bgr"), "invalid pixel format bgr"),
    (dict(width=4, height=2, pixel="rgba", stride=8), "stride too short"),
    (dict(width=4, height=2, pixel="rgba", data=np.zeros(8, np.uint8)), "image data too small"),
])
def test_image_constructor_errors(kw, msg):
    with pytest.raises(ValueError) as e:
        Image(**kw)
    assert str(e.value) == msg


def test_image_row_subview_copy_equalpixels():
    """reference test/copy.js:17-21"""
    rng = np.random.default_rng(0)
    im = Image(width=76, height=50, pixel="rgb", data=rng.integers(0, 256, 50 * 228, dtype=np.uint8))
    c = Image(width=30, height=30, pixel="rgb")
    im.copy(c)
    assert im.subView(0, 0, 30, 30).equalPixels(c)
    sub = im.subView(3, 2, 10, 5)
    assert sub.stride == im.stride and np.array_equal(sub.row(0), im.data[2 * 228 + 9: 2 * 228 + 39])
    sub.data[0] ^= 0xFF  # views share memory like Buffer.slice
    assert im.data[2 * 228 + 9] == sub.data[0]
    assert im.avgChannelDiff(im) == 0
    assert im.avgChannelDiff(Image(width=1, height=1, pixel="rgb")) == 255  # shape mismatch sentinel
    with pytest.raises(ValueError):
        im.copy(Image(width=2, height=2, pixel="rgba"))


def test_image_r16g16_is_accepted_deliberate_fix_of_reference_typo():
    assert Image.pixelSize("r16g16") == 4  # reference lib/image.js:31 has 'r16b16'
    Image(width=2, height=2, pixel="r16g16")


# ---- option marshalling ------------------------------------------------------------------------------
def _opts(**o):
    from picha_amd.api import _resize_options
    return _resize_options(o)


def test_resize_option_defaults_and_overrides():
    """getResizeOptions, reference src/resize.cc:173-198"""
    assert _opts() == (0, pytest.approx(np.float32(0.70)))
    assert _opts(filter="lanczos") == (1, 1.0)               # naming a filter resets the scale to 1.0
    assert _opts(filter="cubic") == (0, 1.0)
    assert _opts(filterScale=0.5) == (0, 0.5)
    assert _opts(filter="box", filterScale=2.0) == (4, 2.0)
    for name in FILTERS:
        assert _opts(filter=name)[0] == FILTERS.index(name)


@pytest.mark.parametrize("o,msg", [
    (dict(filter="bicubic"), "invalid filter mode"),
    (dict(filter=3), "invalid filter mode"),
    (dict(filterScale=0), "invalid filter width"),
    (dict(filterScale=-1.0), "invalid filter width"),
    (dict(filterScale=float("nan")), "invalid filter width"),
    (dict(filterScale="wide"), "invalid filter width"),
])
def test_resize_option_errors(o, msg):
    with pytest.raises(PichaError) as e:
        _opts(**o)
    assert str(e.value) == msg


def test_color_settings_match_reference_normalisation():
    """getSettings, reference src/colorconvert.cc:6-22"""
    assert picha_amd.color_settings() == oracle.color_settings()
    assert picha_amd.color_settings(1, 1, 1) == oracle.color_settings(1, 1, 1)
    assert picha_amd.color_settings(0.5, None, 2.0) == oracle.color_settings(0.5, float("nan"), 2.0)
    r, g, b = picha_amd.color_settings(1, 1, 2)
    assert (r, g, b) == (0.25, 0.25, 0.5)


# ---- argument validation: the reference's messages, raised before any device work -----------------
def test_argument_errors_use_reference_messages():
    img = Image(width=4, height=4)
    cases = [
        (lambda: picha_amd.resizeSync(None, {}), "expected: resizeSync(image, opts)"),
        (lambda: picha_amd.resizeSync(img, None), "expected: resizeSync(image, opts)"),
        (lambda: picha_amd.resize(img, {}, None), "expected: resize(image, opts, cb)"),
        (lambda: picha_amd.resizeSync({"width": 4}, {"width": 2, "height": 2}), "invalid image"),
        (lambda: picha_amd.resizeSync(img, {"width": 0, "height": 2}), "invalid dimensions"),
        (lambda: picha_amd.resizeSync(img, {"width": 2}), "invalid dimensions"),
        (lambda: picha_amd.resizeSync(img, {"width": 2 ** 31, "height": 2}), "invalid dimensions"),  # wraps negative
        (lambda: picha_amd.resizeSync(img, {"width": 2, "height": 2, "filter": "nope"}), "invalid filter mode"),
        (lambda: picha_amd.colorConvertSync(img, {}), "expected pixel mode"),
        (lambda: picha_amd.colorConvertSync(img, {"pixel": "cmyk"}), "expected pixel mode"),
        (lambda: picha_amd.colorConvertSync(5, {"pixel": "rgb"}), "expected: colorConvertSync(image, opts)"),
        (lambda: picha_amd.colorConvert(img, {"pixel": "rgb"}, 7), "expected: colorConvert(image, opts, cb)"),
    ]
    for fn, msg in cases:
        with pytest.raises(PichaError) as e:
            fn()
        assert str(e.value) == msg, msg
    short = Image(width=4, height=4)
    short.data = short.data[:10]  # buffer shorter than h*stride - stride + w*bytes
    with pytest.raises(PichaError) as e:
        picha_amd.resizeSync(short, {"width": 2, "height": 2})
    assert str(e.value) == "invalid image"


def test_no_cpu_fallback_without_a_device():
    """The product path must fail loudly when no GPU is usable — never compute on the CPU."""
    if N.lib().picha_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    img = Image(width=8, height=8)
    with pytest.raises(PichaError) as e:
        picha_amd.resizeSync(img, {"width": 4, "height": 4})
    assert e.value.status == -8 and "no HIP device" in str(e.value)
    with pytest.raises(PichaError) as e:
        picha_amd.colorConvertSync(img, {"pixel": "grey"})
    assert e.value.status == -8


# ---- the product's host tap builder vs the oracle (bitwise) ------------------------------------------
@pytest.mark.parametrize("filt", FILTERS)
def test_make_contribs_bitwise_equal_to_oracle(filt):
    geoms = [(3840, 1920), (2160, 1080), (7680, 1920), (4320, 1080), (2160, 512), (3840, 512), (76, 32), (50, 24),
             (640, 320), (37, 16), (29, 11), (20, 50), (14, 35), (33, 33), (150, 20), (130, 300), (9, 4), (1, 1), (2, 7)]
    for (s, d) in geoms:
        for fs in (1.0, 0.7, 1.3, 2.5):
            try:
                el, er, ew = oracle.make_contribs(filt, fs, s, d)
            except RuntimeError as e:
                assert "rc=-7" in str(e)
                with pytest.raises(PichaError) as pe:
                    picha_amd.make_contribs(filt, fs, s, d)
                assert pe.value.status == -7
                continue
            l, r, w = picha_amd.make_contribs(filt, fs, s, d)
            assert np.array_equal(l, el) and np.array_equal(r, er), (filt, s, d, fs)
            assert w.shape == ew.shape and np.array_equal(w.view(np.uint32), ew.view(np.uint32)), (filt, s, d, fs)
            assert np.all(np.isfinite(w))


def test_fill_random_host_is_deterministic_and_typed():
    a = np.zeros(64, np.uint8)
    b = np.zeros(64, np.uint8)
    N.lib().picha_fill_random_host(a.ctypes.data, 64, 7, 1)
    N.lib().picha_fill_random_host(b.ctypes.data, 64, 7, 1)
    assert np.array_equal(a, b) and len(set(a.tolist())) > 20
    N.lib().picha_fill_random_host(b.ctypes.data, 64, 8, 1)
    assert not np.array_equal(a, b)


def test_shard_range_partitions_a_batch():
    for n in (0, 1, 7, 256, 2048, 1000):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
