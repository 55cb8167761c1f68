"""GPU parity for the rows next to the path:

  * colorConvert + resize chained on the device (picha_hip_convert_resize*) must equal the two reference
    operations run back to back (oracle colorConvert -> oracle resize, or the other order), bit for bit;
  * Image.equalPixels / avgChannelDiff / copy / subView on device images (lib/image.js:57-96) must equal the
    reference's byte-wise definitions computed with numpy on the host.
"""
import ctypes

import re

import numpy as np
import os

import pytest

import picha_amd as picha
from picha_amd import _native as N
from util import PIXELS, max_code_diff, oracle_convert, oracle_resize, payload, rand_image

pytestmark = pytest.mark.gpu


def _c(img):
    return N.CImage(img.data.ctypes.data, img.stride, img.width, img.height, N.pixel_index(img.pixel))


def _chain_oracle(img, pixel, w, h, order, weights, filter, fscale):
    if order == 0:
        return oracle_resize(oracle_convert(img, pixel, weights), w, h, filter, fscale)
    return oracle_convert(oracle_resize(img, w, h, filter, fscale), pixel, weights)


CASES = [
    # src pixel, dst pixel, src size, dst size, filter, filterScale
    ("rgb", "grey", (97, 61), (13, 9), "lanczos", 1.0),          # cfg3's pair at a small size
    ("rgba", "greya", (64, 48), (32, 24), "cubic", 0.70),        # streaming kernel in the chain
    ("grey", "rgba", (31, 17), (47, 40), "triangle", 1.0),       # upscale
    ("r16g16b16a16", "rgb", (40, 30), (20, 15), "mitchel", 1.0), # 16 -> 8 bit
    ("greya", "r16g16b16", (23, 19), (11, 10), "box", 1.0),      # the 2 -> 3 channel map quirk, ring aliasing
    ("rgba", "r16g16b16a16", (50, 20), (50, 20), "catmulrom", 1.0),  # same size: still resampled
]


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-%s-%s" % (c[0], c[1], c[4]))
def test_convert_resize_chain_equals_two_reference_calls(case, order):
    sp, dp, (sw, sh), (dw, dh), filt, fscale = case
    rng = np.random.default_rng(hash((sp, dp, order)) & 0xFFFF)
    weights = picha.color_settings(0.2, 0.7, 0.1)
    img = rand_image(rng, sw, sh, sp, pad=4)
    want = _chain_oracle(img, dp, dw, dh, order, weights, filt, fscale)
    got = picha.convertResizeSync(img, {"pixel": dp, "width": dw, "height": dh, "filter": filt, "filterScale": fscale,
                                        "redWeight": 0.2, "greenWeight": 0.7, "blueWeight": 0.1,
                                        "order": "resizeFirst" if order else "convertFirst"})
    assert (got.width, got.height, got.pixel, got.stride) == (dw, dh, dp, want.stride)
    assert np.array_equal(payload(got), payload(want)), "max code diff %d" % max_code_diff(got, want)


FUSED_CASES = [
    # src pixel, src size, dst size, filter, filterScale[, dst pixel]: downscales where the fused (row-staged) kernel serves the chain
    ("rgb", (960, 270), (128, 64), "lanczos", 1.0),     # cfg3's structure: 7.5x / 4.2x, two outputs per thread
    ("rgb", (1200, 120), (150, 30), "mitchel", 1.0),    # 8x, 31 taps: one output per thread
    ("rgba", (770, 200), (110, 50), "cubic", 0.70),     # rgba -> grey, 7x
    ("rgb", (300, 200), (130, 90), "cubic", 1.0),       # mild non-integer downscale
    ("rgb", (97, 61), (13, 9), "lanczos", 1.0),         # tiny
    ("rgba", (2600, 64), (520, 16), "triangle", 1.0),   # several column strips
    # r03: the other fused pairs (kernels_rows.hip variants 6-11)
    ("rgb", (960, 270), (128, 64), "lanczos", 1.0, "greya"),
    ("rgba", (770, 200), (110, 50), "cubic", 0.70, "greya"),     # alpha carried through the luma pair
    ("r16g16b16", (640, 180), (96, 40), "mitchel", 1.0, "r16"),
    ("r16g16b16a16", (900, 126), (120, 30), "lanczos", 1.0, "r16"),   # (120 rows would be an exact 4x: lanczos aliases the ring there)
    ("r16g16b16", (1200, 90), (150, 30), "cubic", 1.0, "grey"),  # 16-bit source quantised to an 8-bit intermediate
    ("r16g16b16a16", (640, 96), (160, 24), "catmulrom", 1.0, "grey"),
]


@pytest.mark.parametrize("case", FUSED_CASES, ids=lambda c: "%s-%s-%dx%d-%s" % (c[0], c[5] if len(c) > 5 else "grey", c[1][0], c[1][1], c[3]))
def test_fused_convert_resize_kernel_equals_two_reference_calls(case):
    """colour convert fused into the resize — the grey intermediate is quantised in registers and
    never exists in HBM — must equal oracle colorConvert followed by oracle resize bit for bit, and equal the two-kernel
    chain (option fuse = 2)."""
    sp, (sw, sh), (dw, dh), filt, fscale = case[:5]
    dpix = case[5] if len(case) > 5 else "grey"
    rng = np.random.default_rng(sw * 7 + dw)
    # default weights, positive custom weights (no clamp needed: the kernel skips it), a negative weight (luma leaves [0, 1]: clamped)
    for weights, wopts in ((None, {}), (picha.color_settings(0.2, 0.7, 0.1), {"redWeight": 0.2, "greenWeight": 0.7, "blueWeight": 0.1}),
                           (picha.color_settings(1.6, -0.9, 0.3), {"redWeight": 1.6, "greenWeight": -0.9, "blueWeight": 0.3})):
        img = rand_image(rng, sw, sh, sp, pad=int(rng.choice([0, 4])), edge_values=bool(weights is not None))
        want = _chain_oracle(img, dpix, dw, dh, 0, weights, filt, fscale)
        opts = dict({"pixel": dpix, "width": dw, "height": dh, "filter": filt, "filterScale": fscale}, **wopts)
        names = {}
        for fuse in (0, 2):
            for chunks in ((0, 1, 3) if fuse == 0 else (0,)):
                for per_lane in ((0, 1) if fuse == 0 else (0,)):
                    N.check(N.lib().picha_hip_set_option(b"fuse", fuse))
                    N.check(N.lib().picha_hip_set_option(b"row_chunks", chunks))
                    N.check(N.lib().picha_hip_set_option(b"per_lane", per_lane))
                    try:
                        got = picha.convertResizeSync(img, opts)
                    finally:
                        for k in (b"fuse", b"row_chunks", b"per_lane"):
                            N.check(N.lib().picha_hip_set_option(k, 0))
                    names[fuse] = N.lib().picha_hip_last_kernel().decode()
                    assert np.array_equal(payload(got), payload(want)), (case, fuse, chunks, per_lane, names[fuse], max_code_diff(got, want))
        assert names[0].startswith("resize_rows<src ch%d" % N.PIXEL_CHANNELS[sp]), names
        assert not names[2].startswith("resize_rows<src ch%d" % N.PIXEL_CHANNELS[sp]), names
        # fused-multiply-add flavour: within one code value of the reference chain
        N.check(N.lib().picha_hip_set_option(b"numerics", 1))
        try:
            got = picha.convertResizeSync(img, opts)
        finally:
            N.check(N.lib().picha_hip_set_option(b"numerics", 0))
        assert max_code_diff(got, want) <= 1


EPILOGUE_CASES = [
    # src pixel, dst pixel, src size, dst size, filter, filterScale: resize -> colorConvert where the accumulator kernel carries
    # the convert as a store epilogue (tap buckets 4 / 5 / 7 / 8 / 10 / 12 / 16)
    ("rgba", "rgb", (640, 360), (320, 180), "cubic", 0.70),          # tiff/png decode -> 2x down -> jpeg's rgb
    ("rgba", "grey", (600, 300), (200, 100), "cubic", 0.70),         # 3x, 8 taps, luma
    ("rgb", "grey", (521, 211), (260, 105), "mitchel", 1.0),         # ~2x non-integer, 7..8 taps
    ("rgb", "rgba", (400, 300), (100, 75), "mitchel", 1.0),          # 4x, 15 taps -> bucket 16, alpha = 1
    ("greya", "rgba", (334, 100), (160, 50), "cubic", 1.0),          # 2 -> 4 channel map (an odd width would send 2-byte pixels to the two-pass kernels)
    ("r16g16b16a16", "rgba", (512, 128), (256, 64), "catmulrom", 1.0),   # 16-bit resize, 8-bit result
    ("r16g16b16a16", "rgb", (480, 96), (160, 32), "cubic", 0.70),
    ("rgba", "r16g16b16a16", (300, 200), (150, 100), "triangle", 1.0),   # 3 taps -> bucket 5, 8 -> 16 bit
    ("grey", "greya", (256, 256), (128, 128), "cubic", 0.70),
    ("r16g16b16a16", "rgba", (512, 128), (256, 64), "lanczos", 1.0),     # ring-aliased vertical table (9 taps on a ring of 8): the row list under the epilogue
]


@pytest.mark.parametrize("case", EPILOGUE_CASES, ids=lambda c: "%s-%s-%dx%d-%s" % (c[0], c[1], c[2][0], c[2][1], c[4]))
def test_resize_then_convert_epilogue_equals_two_reference_calls(case):
    """SURVEY.md 8f row 2, the other order (r03): resize followed by colorConvert as ONE kernel — the finished pixel is packed to
    the resize's format, unpacked, channel-mapped and packed to the destination format at store time (src/resize.cc:130 then
    src/colorconvert.cc:143-150); must equal oracle resize -> oracle colorConvert bit for bit and the two-kernel chain
    (fuse = 2); fma flavour within one code value."""
    sp, dp, (sw, sh), (dw, dh), filt, fscale = case
    rng = np.random.default_rng(sw * 13 + dw)
    for weights, wopts in ((None, {}), (picha.color_settings(0.25, 0.6, 0.15), {"redWeight": 0.25, "greenWeight": 0.6, "blueWeight": 0.15})):
        img = rand_image(rng, sw, sh, sp, pad=int(rng.choice([0, 8])), edge_values=bool(weights is not None))  # (8: keeps 8-byte pixels aligned)
        want = _chain_oracle(img, dp, dw, dh, 1, weights, filt, fscale)
        opts = dict({"pixel": dp, "width": dw, "height": dh, "filter": filt, "filterScale": fscale, "order": "resizeFirst"}, **wopts)
        names = {}
        for fuse in (1, 0, 2):   # 1: the epilogue wherever it exists; 0: the planner's choice (either form); 2: two kernels
            for chunks in ((0, 1, 3) if fuse == 1 else (0,)):
                N.check(N.lib().picha_hip_set_option(b"fuse", fuse))
                N.check(N.lib().picha_hip_set_option(b"row_chunks", chunks))
                try:
                    got = picha.convertResizeSync(img, opts)
                finally:
                    for k in (b"fuse", b"row_chunks"):
                        N.check(N.lib().picha_hip_set_option(k, 0))
                names[fuse] = N.lib().picha_hip_last_kernel().decode()
                assert (got.width, got.height, got.pixel) == (dw, dh, dp)
                assert np.array_equal(payload(got), payload(want)), (case, fuse, chunks, names[fuse], max_code_diff(got, want))
        assert "epilogue -> " + dp in names[1], names
        assert " + " in names[2] and "epilogue" not in names[2], names
        N.check(N.lib().picha_hip_set_option(b"numerics", 1))
        N.check(N.lib().picha_hip_set_option(b"fuse", 1))
        try:
            got = picha.convertResizeSync(img, opts)
        finally:
            N.check(N.lib().picha_hip_set_option(b"numerics", 0))
            N.check(N.lib().picha_hip_set_option(b"fuse", 0))
        # one code value of the RESIZE's sample type: an 8-bit step is 257 steps of a 16-bit destination
        assert max_code_diff(got, want) <= (257 if dp.startswith("r16") and not sp.startswith("r16") else 1)


ROWS_CASES = [
    ("grey", (960, 270), (128, 64), "lanczos", 1.0),          # cfg3b's structure
    ("grey", (1200, 120), (150, 30), "cubic", 1.0),           # 8x, 31 taps: one output per thread
    ("rgba", (770, 200), (110, 50), "cubic", 1.0),
    ("rgba", (3000, 45), (400, 10), "lanczos", 1.0),           # several column strips, one output per thread
    ("greya", (640, 128), (80, 32), "mitchel", 1.0),
    ("r16g16b16a16", (640, 96), (96, 24), "catmulrom", 1.0),
    ("grey", (515, 77), (103, 19), "lanczos", 1.5),           # wider filter scale: 5x with 30 taps
]


@pytest.mark.parametrize("case", ROWS_CASES, ids=lambda c: "%s-%dx%d-%s" % (c[0], c[1][0], c[1][1], c[3]))
def test_row_staged_downscale_kernel_matches_oracle(case):
    """The row-staged kernel (pixels converted once per row into LDS, windows read from LDS) against the oracle, for one and
    two outputs per thread, several row chunkings, padded strides and the fma flavour; option rows = 2 gives the kernels it
    replaces, which must agree byte for byte."""
    pixel, (sw, sh), (dw, dh), filt, fscale = case
    rng = np.random.default_rng(sw + 31 * dw)
    img = rand_image(rng, sw, sh, pixel, pad=int(rng.choice([0, 4, 8])))
    want = oracle_resize(img, dw, dh, filt, fscale)
    opts = {"width": dw, "height": dh, "filter": filt, "filterScale": fscale}
    try:
        for per_lane in (0, 1):
            for chunks in (0, 1, 4):
                N.check(N.lib().picha_hip_set_option(b"rows", 1))
                N.check(N.lib().picha_hip_set_option(b"per_lane", per_lane))
                N.check(N.lib().picha_hip_set_option(b"row_chunks", chunks))
                got = picha.resizeSync(img, opts)
                name = N.lib().picha_hip_last_kernel().decode()
                assert name.startswith("resize_rows<"), name
                assert max_code_diff(got, want) == 0, (case, per_lane, chunks, name)
        N.check(N.lib().picha_hip_set_option(b"rows", 2))
        N.check(N.lib().picha_hip_set_option(b"per_lane", 0))
        N.check(N.lib().picha_hip_set_option(b"row_chunks", 0))
        got = picha.resizeSync(img, opts)
        assert not N.lib().picha_hip_last_kernel().decode().startswith("resize_rows<")
        assert max_code_diff(got, want) == 0
        N.check(N.lib().picha_hip_set_option(b"rows", 1))
        N.check(N.lib().picha_hip_set_option(b"numerics", 1))
        got = picha.resizeSync(img, opts)
        assert N.lib().picha_hip_last_kernel().decode().startswith("resize_rows<") and max_code_diff(got, want) <= 1
    finally:
        for k in (b"rows", b"per_lane", b"row_chunks", b"numerics"):
            N.check(N.lib().picha_hip_set_option(k, 0))


RUN_CASES = [   # pixel, source, destination, filter, filter scale, (period in outputs, period in source pixels, taps)
    ("grey", (960, 270), (128, 64), "lanczos", 1.0, (2, 15, 31)),      # cfg3b's structure: 7.5x in x (period 2 / 15 pixels), 4.2x in y
    ("grey", (960, 540), (128, 72), "cubic", 0.70, (2, 15, 21)),       # 7.5x both ways, picha's default options (21 taps)
    ("grey", (1920, 90), (256, 37), "mitchel", 1.0, (2, 15, 31)),      # two waves per row (four outputs per lane), odd output height
    ("grey", (975, 300), (130, 40), "catmulrom", 0.70, (2, 15, 21)),   # the last lane's run hangs over the row end (130 = 32 * 4 + 2)
    ("greya", (960, 270), (128, 64), "lanczos", 1.0, (2, 15, 31)),     # two-byte pixels: two (default) or four outputs per lane
    ("greya", (1950, 150), (260, 20), "cubic", 0.70, (2, 15, 21)),     # five row pieces of 64 lanes, the last one ragged
    ("grey", (1920, 180), (320, 30), "cubic", 0.70, (1, 6, 17)),       # integer scale, period 1: 6x, 17 taps
    ("grey", (2048, 240), (256, 41), "cubic", 0.70, (1, 8, 23)),       # 8x, 23 taps: four or eight outputs per lane
    ("grey", (960, 150), (640, 100), "cubic", 0.70, (2, 3, 4)),        # 1.5x: period 2 / 3 source pixels, 4 taps (offsets rounded up)
    ("greya", (1230, 90), (820, 60), "cubic", 0.70, (2, 3, 4)),
    ("grey", (1280, 90), (512, 36), "cubic", 0.70, (2, 5, 7)),         # 2.5x
    ("greya", (640, 75), (256, 30), "cubic", 0.70, (2, 5, 7)),
    ("grey", (1536, 90), (512, 30), "cubic", 0.70, (1, 3, 8)),         # 3x
    ("greya", (1024, 64), (256, 16), "cubic", 0.70, (1, 4, 11)),       # 4x
    ("grey", (1280, 100), (256, 20), "cubic", 0.70, (1, 5, 14)),       # 5x
    ("greya", (1280, 100), (256, 20), "cubic", 0.70, (1, 5, 14)),
]
# (default outputs per lane, the other instantiated run lengths)
RUN_PER_LANE = {("grey", 2, 21): (4, (2, 8)), ("grey", 2, 31): (4, (2,)), ("greya", 2, 21): (2, ()), ("greya", 2, 31): (2, (4,)),
                ("grey", 1, 17): (4, ()), ("grey", 1, 23): (4, (8,)), ("grey", 2, 4): (8, (4,)), ("greya", 2, 4): (8, (4,)),
                ("grey", 2, 7): (8, ()), ("greya", 2, 7): (4, ()), ("grey", 1, 8): (8, ()), ("greya", 1, 11): (4, ()),
                ("grey", 1, 14): (4, ()), ("greya", 1, 14): (4, ())}


@pytest.mark.parametrize("case", RUN_CASES, ids=lambda c: "%s-%dx%d-%s" % (c[0], c[1][0], c[1][1], c[3]))
def test_run_kernel_matches_oracle(case):
    """r03: the run kernel (kernels_run.hip: periodic tap tables, a lane resamples a run of 2 / 4 / 8 consecutive outputs with
    wave-uniform weights; the columns whose windows an image edge clips are recomputed by resize_edges_kernel from the stash of
    raw pixels the march leaves) against the oracle: every outputs-per-lane form, row chunkings, the three launch forms, padded
    strides, a device batch, and the fma flavour; run = 2 gives the kernel it replaces, which must agree byte for byte."""
    from picha_amd.device import DeviceBatch, resize_batch
    import torch
    pixel, (sw, sh), (dw, dh), filt, fscale, (period, period_px, taps) = case
    ch = N.PIXEL_CHANNELS[pixel]
    rng = np.random.default_rng(sw + 17 * dh)
    img = rand_image(rng, sw, sh, pixel, pad=int(rng.choice([0, 4, 8])))
    want = oracle_resize(img, dw, dh, filt, fscale)
    opts = {"width": dw, "height": dh, "filter": filt, "filterScale": fscale}
    keys = (b"run", b"per_lane", b"row_chunks", b"work", b"numerics", b"block_waves", b"run_round")
    try:
        N.check(N.lib().picha_hip_set_option(b"run", 1))
        for rnd in ((1, 2) if taps == 31 else (0,)):   # both window-offset roundings where the table has the half-pixel step
            N.check(N.lib().picha_hip_set_option(b"run_round", rnd))
            got = picha.resizeSync(img, opts)
            assert max_code_diff(got, want) == 0, (case, rnd, N.lib().picha_hip_last_kernel())
        N.check(N.lib().picha_hip_set_option(b"run_round", 0))
        default_pl, other_pl = RUN_PER_LANE[(pixel, period, taps)]
        for per_lane in (0,) + (default_pl,) + tuple(other_pl):
            for chunks, work in ((0, 0), (1, 3), (3, 1), (2, 2)):
                N.check(N.lib().picha_hip_set_option(b"per_lane", per_lane))
                N.check(N.lib().picha_hip_set_option(b"row_chunks", chunks))
                N.check(N.lib().picha_hip_set_option(b"work", work))
                got = picha.resizeSync(img, opts)
                name = N.lib().picha_hip_last_kernel().decode()
                pl = per_lane or default_pl
                # 31 taps: 30 with the window offsets rounded up (",ceil"), where that form is instantiated
                assert re.match(r"resize_run<ch%d,u8,Q%d,PX%d,TX(%d|%d),P%d," % (ch, period, period_px, taps, taps - 1, pl), name), name
                assert "+ resize_edges<" in name, name
                assert max_code_diff(got, want) == 0, (case, per_lane, chunks, work, name)
        for k in keys[1:]:
            N.check(N.lib().picha_hip_set_option(k, 0))
        N.check(N.lib().picha_hip_set_option(b"block_waves", 1))
        got = picha.resizeSync(img, opts)
        assert ",bw1" in N.lib().picha_hip_last_kernel().decode() and max_code_diff(got, want) == 0
        N.check(N.lib().picha_hip_set_option(b"block_waves", 0))
        N.check(N.lib().picha_hip_set_option(b"run", 2))
        got = picha.resizeSync(img, opts)
        assert not N.lib().picha_hip_last_kernel().decode().startswith("resize_run<")
        assert max_code_diff(got, want) == 0
        N.check(N.lib().picha_hip_set_option(b"run", 1))
        N.check(N.lib().picha_hip_set_option(b"numerics", 1))
        got = picha.resizeSync(img, opts)
        assert N.lib().picha_hip_last_kernel().decode().startswith("resize_run<") and max_code_diff(got, want) <= 1
        N.check(N.lib().picha_hip_set_option(b"numerics", 0))
        # a device batch
        n = 5
        src = DeviceBatch(n, sw, sh, pixel).fill_random(3 + sw)
        out = resize_batch(src, dw, dh, filter=filt, filter_scale=fscale)
        torch.cuda.synchronize()
        assert N.lib().picha_hip_last_kernel().decode().startswith("resize_run<")
        for i in range(n):
            exp = oracle_resize(src.image_host(i), dw, dh, filt, fscale)
            assert np.array_equal(out.image_host(i).pixels(), exp.pixels()), (case, i)
    finally:
        for k in keys:
            N.check(N.lib().picha_hip_set_option(k, 0))


@pytest.mark.parametrize("pixel", PIXELS)
def test_integer_upscale_kernel_matches_oracle(pixel):
    """r03: resize_up_kernel (kernels_run.hip) — 2x upscales in x with wave-uniform weights and a run of outputs per lane, the last
    four resampled rows in registers, edge columns from the stash by resize_edges_kernel — against the oracle: the filters with
    3- and 4-tap windows, vertical factors 2x / 2.5x / 3.1x, run lengths, row chunkings, padded strides, both numerics; run = 2
    gives the kernels it replaces."""
    rng = np.random.default_rng(5200 + PIXELS.index(pixel))
    served = 0
    for (sw, sh, dw, dh, filt, fscale) in [(96, 40, 192, 80, "cubic", 0.70), (160, 30, 320, 75, "cubic", 0.70), (130, 33, 260, 103, "cubic", 1.0),
                                           (300, 20, 600, 40, "mitchel", 1.0), (64, 64, 128, 200, "triangle", 1.0), (200, 25, 400, 50, "catmulrom", 1.0),
                                           (96, 40, 192, 80, "lanczos", 1.0), (96, 40, 192, 80, "box", 1.0),
                                           (80, 30, 320, 120, "cubic", 0.70), (72, 25, 288, 77, "mitchel", 1.0)]:   # 4x in x
        img = rand_image(rng, sw, sh, pixel, pad=int(rng.choice([0, 8])), edge_values=bool(rng.integers(0, 3) == 0))
        want = oracle_resize(img, dw, dh, filt, fscale)
        opts = {"width": dw, "height": dh, "filter": filt, "filterScale": fscale}
        try:
            for run, per_lane, chunks in ((1, 0, 0), (1, 2, 1), (1, 4, 3), (1, 8, 2), (0, 0, 0), (2, 0, 0)):
                N.check(N.lib().picha_hip_set_option(b"run", run))
                N.check(N.lib().picha_hip_set_option(b"per_lane", per_lane))
                N.check(N.lib().picha_hip_set_option(b"row_chunks", chunks))
                got = picha.resizeSync(img, opts)
                name = N.lib().picha_hip_last_kernel().decode()
                served += name.startswith("resize_up<")
                assert not (run == 2 and name.startswith("resize_up<")), name
                assert max_code_diff(got, want) == 0, (pixel, sw, sh, dw, dh, filt, fscale, run, per_lane, chunks, name)
            N.check(N.lib().picha_hip_set_option(b"run", 1))
            N.check(N.lib().picha_hip_set_option(b"per_lane", 0))
            N.check(N.lib().picha_hip_set_option(b"numerics", 1))
            got = picha.resizeSync(img, opts)
            assert max_code_diff(got, want) <= 1, (pixel, sw, filt, "fma", N.lib().picha_hip_last_kernel())
        finally:
            for k in (b"run", b"per_lane", b"row_chunks", b"numerics"):
                N.check(N.lib().picha_hip_set_option(k, 0))
    # (r16g16b16a16: not instantiated — four outputs per lane measured slower than the accumulator form, two cannot hold the three
    #  clipped columns at the right edge)
    assert served >= (0 if pixel == "r16g16b16a16" else 8), served


def test_run_kernel_is_the_planners_choice_only_where_it_was_measured_faster():
    """Default options: long windows (>= 26 taps) of grey / greya at a periodic scale take the run kernel; 21-tap windows only
    for large grey batches; other formats, non-periodic scales and a forced row-staged kernel never."""
    from picha_amd.device import DeviceBatch, resize_batch
    import torch
    rng = np.random.default_rng(11)
    last = lambda: N.lib().picha_hip_last_kernel().decode()
    for pixel, (sw, sh), (dw, dh), o, expect in [
        ("grey", (960, 270), (128, 64), {"filter": "lanczos"}, True),
        ("greya", (960, 270), (128, 64), {"filter": "lanczos"}, True),
        ("grey", (960, 540), (128, 72), {}, False),                       # 21 taps, one image
        ("rgba", (960, 270), (128, 64), {"filter": "lanczos"}, False),
        ("grey", (1000, 270), (128, 64), {"filter": "lanczos"}, False),    # 7.8125x: period 16
    ]:
        img = rand_image(rng, sw, sh, pixel)
        got = picha.resizeSync(img, dict(o, width=dw, height=dh))
        assert last().startswith("resize_run<") == expect, (pixel, sw, last())
        assert max_code_diff(got, oracle_resize(img, dw, dh, o.get("filter", "cubic"), 1.0 if "filter" in o else 0.70)) == 0
    src = DeviceBatch(192, 960, 540, "grey").fill_random(2)
    out = resize_batch(src, 128, 72)
    torch.cuda.synchronize()
    assert last().startswith("resize_run<ch1,u8,Q2,PX15,TX21,P4,"), last()
    for i in (0, 95, 191):
        assert np.array_equal(out.image_host(i).pixels(), oracle_resize(src.image_host(i), 128, 72).pixels())
    N.check(N.lib().picha_hip_set_option(b"rows", 1))
    try:
        img = rand_image(rng, 960, 270, "grey")
        picha.resizeSync(img, {"width": 128, "height": 64, "filter": "lanczos"})
        assert last().startswith("resize_rows<"), last()
    finally:
        N.check(N.lib().picha_hip_set_option(b"rows", 0))


def test_convert_resize_chain_same_format_is_plain_resize_and_errors_match():
    rng = np.random.default_rng(5)
    img = rand_image(rng, 40, 30, "rgb")
    got = picha.convertResizeSync(img, {"pixel": "rgb", "width": 20, "height": 15})
    assert np.array_equal(payload(got), payload(oracle_resize(img, 20, 15)))
    with pytest.raises(picha.PichaError, match="expected pixel mode"):
        picha.convertResizeSync(img, {"width": 20, "height": 15})
    with pytest.raises(picha.PichaError, match="invalid dimensions"):
        picha.convertResizeSync(img, {"pixel": "grey", "width": 0, "height": 15})
    with pytest.raises(picha.PichaError, match="invalid filter mode"):
        picha.convertResizeSync(img, {"pixel": "grey", "width": 4, "height": 4, "filter": "nearest"})
    L = N.lib()
    src, dst = _c(img), _c(picha.Image(width=4, height=4, pixel="grey"))
    assert L.picha_hip_convert_resize(ctypes.byref(src), ctypes.byref(dst), 7, 0.3, 0.6, 0.1, 0, 1.0) == -10


def test_convert_resize_device_batch_slices_through_the_intermediate_window():
    """A batch larger than the intermediate window is processed in slices; every image must still be right."""
    import torch
    from picha_amd.device import DeviceBatch, convert_resize_batch
    L = N.lib()
    n, sw, sh, dw, dh = 12, 640, 360, 128, 96
    src = DeviceBatch(n, sw, sh, "rgb").fill_random(77)
    old = L.picha_hip_get_option(b"chain_window_mb")
    try:
        N.check(L.picha_hip_set_option(b"chain_window_mb", 1))   # 640*360 grey = 230 KB -> 4 images per slice
        out = convert_resize_batch(src, "grey", dw, dh, filter="lanczos")
        torch.cuda.synchronize()
    finally:
        N.check(L.picha_hip_set_option(b"chain_window_mb", old))
    for i in (0, 3, 4, 7, 11):
        want = _chain_oracle(src.image_host(i), "grey", dw, dh, 0, None, "lanczos", 1.0)
        assert np.array_equal(payload(out.image_host(i)), payload(want)), i


def _host_stats(a, b):
    pa, pb = payload(a).astype(np.int64), payload(b).astype(np.int64)
    return int(np.abs(pa - pb).sum()), int((pa != pb).sum())


@pytest.mark.parametrize("pixel", ["rgba", "rgb", "grey", "r16g16b16"])
def test_device_image_compare_matches_reference_definitions(pixel):
    import torch
    from picha_amd.device import DeviceBatch
    rng = np.random.default_rng(11)
    n, w, h = 5, 211, 37
    a, b = DeviceBatch(n, w, h, pixel), DeviceBatch(n, w, h, pixel)
    imgs_a, imgs_b = [], []
    for i in range(n):
        ia = rand_image(rng, w, h, pixel)
        ib = picha.Image(width=w, height=h, pixel=pixel, stride=ia.stride, data=ia.data.copy())
        if i == 1:                                   # identical payload, different PADDING: still equal
            bpp = N.PIXEL_BYTES[pixel]
            if ib.stride > w * bpp:
                ib.data.reshape(h, ib.stride)[:, w * bpp:] ^= 0xFF
        elif i == 2:                                 # one byte differs
            ib.data[7 * ib.stride + 5] ^= 0x40
        elif i >= 3:                                 # everything differs
            ib = rand_image(rng, w, h, pixel)
        a.upload(i, ia); b.upload(i, ib)
        imgs_a.append(ia); imgs_b.append(ib)
    torch.cuda.synchronize()
    eq = a.equal_pixels(b)
    avg = a.avg_channel_diff(b)
    for i in range(n):
        s, c = _host_stats(imgs_a[i], imgs_b[i])
        assert eq[i] == (c == 0) == imgs_a[i].equalPixels(imgs_b[i])
        assert avg[i] == s / (h * w * N.PIXEL_BYTES[pixel]) == imgs_a[i].avgChannelDiff(imgs_b[i])
    assert eq[:3] == [True, True, False]


def test_device_image_compare_subviews_unaligned_and_mismatched_geometry():
    import torch
    from picha_amd.device import DeviceBatch
    rng = np.random.default_rng(12)
    w, h = 90, 40
    a, b = DeviceBatch(2, w, h, "rgb"), DeviceBatch(2, w, h, "rgb")
    imgs = [rand_image(rng, w, h, "rgb") for _ in range(4)]
    for i in range(2):
        a.upload(i, imgs[i]); b.upload(i, imgs[2 + i])
    torch.cuda.synchronize()
    # rgb subviews at x = 1 vs x = 2: rows at byte offsets 3 and 6 (not congruent mod 4) -> the byte path
    va, vb = a.sub_view(1, 3, 50, 20), b.sub_view(2, 5, 50, 20)
    avg = va.avg_channel_diff(vb)
    for i in range(2):
        ha, hb = imgs[i].subView(1, 3, 50, 20), imgs[2 + i].subView(2, 5, 50, 20)
        assert avg[i] == ha.avgChannelDiff(hb)
    # congruent but unaligned rows -> head / dword / tail path
    va, vb = a.sub_view(1, 0, 77, 9), b.sub_view(5, 2, 77, 9)
    for i, v in enumerate(va.avg_channel_diff(vb)):
        assert v == imgs[i].subView(1, 0, 77, 9).avgChannelDiff(imgs[2 + i].subView(5, 2, 77, 9))
    # different geometry: the reference answers 255 / false without looking at pixels (lib/image.js:58, 66)
    other = a.sub_view(0, 0, 50, 21)
    assert a.sub_view(0, 0, 50, 20).avg_channel_diff(other) == [255, 255]
    assert a.sub_view(0, 0, 50, 20).equal_pixels(other) == [False, False]


def test_device_image_copy_is_the_common_rectangle_payload_only():
    import torch
    from picha_amd.device import DeviceBatch
    rng = np.random.default_rng(13)
    src, dst = DeviceBatch(2, 33, 21, "greya"), DeviceBatch(2, 20, 30, "greya")
    s_imgs = [rand_image(rng, 33, 21, "greya") for _ in range(2)]
    d_imgs = [rand_image(rng, 20, 30, "greya") for _ in range(2)]
    for i in range(2):
        src.upload(i, s_imgs[i]); dst.upload(i, d_imgs[i])
    src.copy_to(dst)
    torch.cuda.synchronize()
    for i in range(2):
        s_imgs[i].copy(d_imgs[i])                      # the host mirror of lib/image.js:89-96
        got = dst.image_host(i)
        assert np.array_equal(got.data[: (30 - 1) * got.stride + 40], d_imgs[i].data[: (30 - 1) * got.stride + 40])
    with pytest.raises(N.PichaError, match="can't copy pixels between different pixel types"):
        src.copy_to(DeviceBatch(2, 20, 30, "rgb"))


def test_convert_resize_async_batches_and_equals_sync():
    import threading
    rng = np.random.default_rng(21)
    imgs = [rand_image(rng, 120, 80, "rgb") for _ in range(12)]
    o = {"pixel": "grey", "width": 40, "height": 30, "filter": "lanczos"}
    results, lock, done = {}, threading.Lock(), threading.Event()

    def make_cb(i):
        def cb(err, img):
            with lock:
                results[i] = (err, img)
                if len(results) == len(imgs):
                    done.set()
        return cb
    for i, im in enumerate(imgs):
        picha.convertResize(im, o, make_cb(i))
    picha.drain()
    assert done.wait(30)
    for i, im in enumerate(imgs):
        err, got = results[i]
        assert err is None
        assert np.array_equal(payload(got), payload(picha.convertResizeSync(im, o)))
    with pytest.raises(picha.PichaError, match="expected pixel mode"):
        picha.convertResize(imgs[0], {"width": 4, "height": 4}, lambda e, i: None)


# ---- r04: neighbour-exchange form of the run kernel (dword-sized pixels at integer scales) and scalar weights for one phase ----
RUN_XCH_CASES = [
    # pixel, (sw, sh), (dw, dh), filter, filterScale, (period_px, taps, window pixels in front of the lane's own)
    ("r16g16b16a16", (1536, 200), (384, 50), "mitchel", 1.0, (4, 15, 5)),    # cfg5's shape: two waves wide
    ("r16g16b16a16", (2048, 96), (512, 24), "mitchel", 1.0, (4, 15, 5)),     # 512 outputs: exactly four waves
    ("r16g16b16a16", (520, 64), (130, 16), "mitchel", 1.0, (4, 15, 5)),      # ragged last wave, one lane beyond the image
    ("r16g16b16a16", (1024, 80), (256, 20), "cubic", 0.70, (4, 11, 3)),
    ("rgba", (2048, 120), (512, 30), "cubic", 0.70, (4, 11, 3)),
    ("rgba", (1000, 64), (250, 16), "mitchel", 1.0, (4, 15, 5)),
    ("rgba", (1536, 90), (512, 30), "cubic", 0.70, (3, 8, 2)),
    ("r16g16b16a16", (512, 1400), (128, 350), "mitchel", 1.0, (4, 15, 5)),    # tall: a phased buffer of 8 rows fills many times over
    ("r16g16b16a16", (512, 228), (128, 57), "mitchel", 1.0, (4, 15, 5)),      # 57 rows in one chunk: rows 55 and 56 complete with the same (last) source row
                                                                              # when the phase buffer holds 7 — the case that lost a row once
]


@pytest.mark.parametrize("case", RUN_XCH_CASES, ids=lambda c: "%s-%dx%d-%s" % (c[0], c[1][0], c[1][1], c[3]))
def test_run_kernel_neighbour_exchange_matches_oracle(case):
    """r04: resize_run<..., xchN>: a lane loads only the pixels it owns, takes the rest of its window from the two neighbouring
    lanes (DPP wave shifts of raw dwords) and the wave's own halo through LDS; wave-uniform weights; the columns an image edge clips
    are recomputed by resize_edges_kernel from the IMAGE.  Against the oracle (bit-exact; fma <= 1): chunkings, launch forms,
    workgroup widths, padded strides, subViews, a device batch; run_xch = 2 gives the kernel it replaces."""
    from picha_amd.device import DeviceBatch, resize_batch
    import torch
    pixel, (sw, sh), (dw, dh), filt, fscale, (px, taps, hl) = case
    rng = np.random.default_rng(sw + 31 * dh)
    img = rand_image(rng, sw, sh, pixel, pad=int(rng.choice([0, 8, 16])))
    want = oracle_resize(img, dw, dh, filt, fscale)
    opts = {"width": dw, "height": dh, "filter": filt, "filterScale": fscale}
    keys = (b"run_xch", b"row_chunks", b"work", b"numerics", b"block_waves", b"run_phase", b"run_edges")
    expect = r"resize_run<ch4,%s,Q1,PX%d,TX%d,P2,NACC\d,(exact|fma),bw\d,xch%d" % ("u16" if pixel.startswith("r16") else "u8", px, taps, hl)
    try:
        N.check(N.lib().picha_hip_set_option(b"run_xch", 1))
        for chunks, work, bw in ((0, 0, 0), (1, 3, 0), (3, 1, 1), (2, 2, 2), (5, 0, 4)):
            N.check(N.lib().picha_hip_set_option(b"run_edges", 2 if chunks == 3 else 0))   # (the edge kernel reading the image, or — r05, auto — the clipped columns inside the march)
            N.check(N.lib().picha_hip_set_option(b"row_chunks", chunks))
            N.check(N.lib().picha_hip_set_option(b"work", work))
            N.check(N.lib().picha_hip_set_option(b"block_waves", bw))
            got = picha.resizeSync(img, opts)
            name = N.lib().picha_hip_last_kernel().decode()
            assert re.match(expect, name) and ("+ resize_edges<" in name) != (",edges" in name), name   # (r05: the clipped columns inside the march, or the edge kernel)
            assert max_code_diff(got, want) == 0, (case, chunks, work, bw, name)
        for k in keys[1:]:
            N.check(N.lib().picha_hip_set_option(k, 0))
        # time-phased stores (finished rows wait in registers for a chip-wide clock phase): phases from 0.6 us (every row flushes) to
        # 10 ms (only full buffers and the end of a chunk flush), one chunk and many
        for phase, chunks in ((6, 0), (9, 1), (12, 3), (20, 1), (20, 2)):
            N.check(N.lib().picha_hip_set_option(b"run_phase", phase))
            N.check(N.lib().picha_hip_set_option(b"row_chunks", chunks))
            got = picha.resizeSync(img, opts)
            name = N.lib().picha_hip_last_kernel().decode()
            assert ",phased%d" % phase in name, name
            assert max_code_diff(got, want) == 0, (case, phase, chunks, name)
        N.check(N.lib().picha_hip_set_option(b"run_phase", 0))
        N.check(N.lib().picha_hip_set_option(b"row_chunks", 0))
        sub = None
        N.check(N.lib().picha_hip_set_option(b"numerics", 1))
        got = picha.resizeSync(img, opts)
        assert re.match(expect, N.lib().picha_hip_last_kernel().decode()) and max_code_diff(got, want) <= 1
        N.check(N.lib().picha_hip_set_option(b"numerics", 0))
        N.check(N.lib().picha_hip_set_option(b"run_xch", 2))
        got = picha.resizeSync(img, opts)
        assert not N.lib().picha_hip_last_kernel().decode().startswith("resize_run<")
        assert max_code_diff(got, want) == 0
        N.check(N.lib().picha_hip_set_option(b"run_xch", 1))
        n = 5
        src = DeviceBatch(n, sw, sh, pixel).fill_random(11 + sw)
        out = resize_batch(src, dw, dh, filter=filt, filter_scale=fscale)
        torch.cuda.synchronize()
        assert re.match(expect, N.lib().picha_hip_last_kernel().decode())
        for i in range(n):
            exp = oracle_resize(src.image_host(i), dw, dh, filt, fscale)
            assert np.array_equal(out.image_host(i).pixels(), exp.pixels()), (case, i)
        del sub
    finally:
        for k in keys:
            N.check(N.lib().picha_hip_set_option(k, 0))


@pytest.mark.parametrize("pixel,dw", [("grey", 512), ("greya", 256)])
def test_run_kernel_scalar_phase_weights_match_oracle(pixel, dw):
    """r04: resize_run<..., ws1>: the second phase of a 7.5x table keeps its weights in SGPRs (31 / 30 registers fewer per lane).
    Same sums in the same order: bit-exact; run_ws = 2 gives the all-VGPR form, byte-equal."""
    sw, sh, dh = dw * 15 // 2, 130, 40
    rng = np.random.default_rng(dw)
    img = rand_image(rng, sw, sh, pixel)
    want = oracle_resize(img, dw, dh, "lanczos", 1.0)
    opts = {"width": dw, "height": dh, "filter": "lanczos"}
    try:
        for rnd in (0, 1, 2):
            N.check(N.lib().picha_hip_set_option(b"run_round", rnd))
            for ws in (0, 2):
                N.check(N.lib().picha_hip_set_option(b"run_ws", ws))
                got = picha.resizeSync(img, opts)
                name = N.lib().picha_hip_last_kernel().decode()
                assert name.startswith("resize_run<") and (",ws" in name) == (ws == 0), name   # (",ws2" in the shared-unpack form)
                assert max_code_diff(got, want) == 0, (pixel, rnd, ws, name)
        for k in (b"run_round", b"run_ws"):
            N.check(N.lib().picha_hip_set_option(k, 0))
        N.check(N.lib().picha_hip_set_option(b"numerics", 1))
        got = picha.resizeSync(img, opts)
        assert ",ws" in N.lib().picha_hip_last_kernel().decode() and max_code_diff(got, want) <= 1
    finally:
        for k in (b"run_round", b"run_ws", b"numerics"):
            N.check(N.lib().picha_hip_set_option(k, 0))


SHU_CASES = [   # pixel, source, destination, filter, filter scale: the shapes the shared-unpack form is instantiated for (grey at 7.5x; rgb at 7.5x / 6x / 8x, 21 / 17 / 23 taps)
    ("grey", (960, 270), (128, 64), "lanczos", 1.0),        # 30 / 31 taps, one wave per row
    ("grey", (3840, 130), (512, 31), "lanczos", 1.0),       # cfg3b's rows: two waves per row, the second wave's last lane ends the row
    ("grey", (975, 300), (130, 40), "catmulrom", 0.70),     # 21 taps; the last lane's run hangs over the row end (130 = 32 * 4 + 2)
    ("grey", (1950, 77), (260, 20), "cubic", 0.70),         # 21 taps, five waves per row, the last one ragged
    ("grey", (3900, 64), (520, 9), "mitchel", 1.0),         # 31 taps, three waves per row (520 = 2 * 256 + 8)
    ("grey", (30, 40), (4, 7), "lanczos", 1.0),             # one lane: every column is a clipped one
    ("rgb", (3840, 150), (512, 20), "cubic", 0.70),         # the jpeg thumbnail's rows: 7.5x, 21 taps, two outputs per lane, four waves per row
    ("rgb", (975, 121), (130, 16), "catmulrom", 0.70),      # ... ragged: 65 lanes
    ("rgb", (1920, 100), (320, 17), "mitchel", 0.70),       # 6x, 17 taps (period 1: every weight scalar)
    ("rgb", (2056, 90), (257, 11), "cubic", 0.70),          # 8x, 23 taps, an odd output width
    ("rgb", (3840, 150), (512, 20), "lanczos", 1.0),        # 30 / 31 taps with two outputs per lane: the window reaches over TWO neighbours (lane + 1 gives all 15 pixels, lane + 2 eight)
    ("rgb", (975, 60), (130, 8), "mitchel", 1.0),           # ... ragged: 65 lanes, the second virtual lane behind the wave in use
    ("greya", (3840, 130), (512, 31), "lanczos", 1.0),      # greya, two neighbours, halo by bytes (46 samples)
    ("greya", (1950, 77), (260, 20), "cubic", 0.70),        # greya 21 taps: one neighbour
    ("rgb", (3840, 264), (480, 33), "lanczos", 1.0),        # 8x lanczos: 33 taps over two neighbours, and a ring-aliased vertical table — the march follows the row list
    ("rgb", (1920, 300), (320, 50), "lanczos", 1.0),        # 6x lanczos: 25 taps, five waves per row
    ("grey", (3840, 264), (480, 33), "lanczos", 1.0),       # grey 8x lanczos: four outputs per lane, one neighbour, row list
    ("greya", (1920, 300), (320, 50), "lanczos", 1.0),      # greya 6x lanczos: two neighbours, halo by bytes, row list
    ("grey", (2056, 90), (257, 11), "cubic", 0.70),         # grey 8x, 23 taps, four outputs per lane, an odd width
    ("greya", (1920, 100), (320, 17), "mitchel", 0.70),     # greya 6x, 17 taps
]


@pytest.mark.parametrize("case", SHU_CASES, ids=lambda c: "%s-%dx%d-%s" % (c[0], c[1][0], c[1][1], c[3]))
def test_run_kernel_shared_unpack_matches_oracle(case):
    """r05: resize_run<..., shu>: a lane converts only the pixels its window starts with and takes the rest of its window, already converted, from
    lane + 1 through LDS (lane 63 from the samples the wave loads one byte per lane); the clipped columns come from the image.  The same float per
    window pixel, the same sums in the same order: bit-exact (fma: <= 1).  Against the oracle over both window roundings, row chunkings, launch forms,
    workgroup widths, padded strides and a sub-view; run_shu = 2 gives the plain form, which must agree byte for byte; run_edges = 1 (the stash)
    rules the form out."""
    pixel, (sw, sh), (dw, dh), filt, fscale = case
    rng = np.random.default_rng(sw + 31 * dh)
    big = rand_image(rng, sw + 8, sh + 3, pixel, pad=int(rng.choice([0, 4, 12])), edge_values=(dw % 3 == 0))
    keys = (b"run", b"run_shu", b"run_edges", b"run_round", b"row_chunks", b"work", b"block_waves", b"numerics")
    opts = {"width": dw, "height": dh, "filter": filt, "filterScale": fscale}
    try:
        N.check(N.lib().picha_hip_set_option(b"run", 1))
        for img in (big.subView(0, 0, sw, sh), big.subView(8, 3, sw, sh), big.subView(5, 1, sw, sh)):   # (odd offsets: the row base is not a dword's)
            want = oracle_resize(img, dw, dh, filt, fscale)
            for rnd in ((0, 1, 2) if fscale == 1.0 and sw % dw else (0,)):   # (both window roundings where the table has the half-pixel step: 31 / 30 taps)
                for chunks, work, bw in ((0, 0, 0), (1, 3, 1), (3, 1, 2), (2, 2, 4)):
                    for k, v in ((b"run_round", rnd), (b"row_chunks", chunks), (b"work", work), (b"block_waves", bw), (b"run_shu", 1), (b"run_edges", 0)):
                        N.check(N.lib().picha_hip_set_option(k, v))
                    got = picha.resizeSync(img, opts)
                    name = N.lib().picha_hip_last_kernel().decode()
                    if dw >= 8:
                        assert name.startswith("resize_run<ch%d,u8,Q" % N.PIXEL_CHANNELS[pixel]) and ",shu" in name and "+ resize_edges<" in name, name
                    assert max_code_diff(got, want) == 0, (case, rnd, chunks, work, bw, name)
                    if name.startswith("resize_run<"):
                        N.check(N.lib().picha_hip_set_option(b"run_shu", 2))
                        plain = picha.resizeSync(img, opts)
                        pname = N.lib().picha_hip_last_kernel().decode()
                        assert ",shu" not in pname and pname.startswith("resize_run<" if pixel != "rgb" and sw % dw else "resize_"), pname   # (rgb, and the row-list shapes: no plain run kernel)
                        assert max_code_diff(plain, got) == 0, (case, rnd, pname)
                        N.check(N.lib().picha_hip_set_option(b"run_shu", 1))
                        N.check(N.lib().picha_hip_set_option(b"run_edges", 1))
                        picha.resizeSync(img, opts)
                        assert ",shu" not in N.lib().picha_hip_last_kernel().decode()
            N.check(N.lib().picha_hip_set_option(b"run_edges", 0))
            N.check(N.lib().picha_hip_set_option(b"numerics", 1))
            got = picha.resizeSync(img, opts)
            assert max_code_diff(got, want) <= 1, (case, N.lib().picha_hip_last_kernel())
            N.check(N.lib().picha_hip_set_option(b"numerics", 0))
    finally:
        for k in keys:
            N.check(N.lib().picha_hip_set_option(k, 0))


@pytest.mark.parametrize("case", [("grey", (3840, 300), (512, 40), "lanczos"), ("greya", (960, 130), (128, 31), "lanczos"), ("rgb", (1920, 135), (256, 18), None),
                                  ("rgba", (1920, 216), (960, 108), "lanczos"), ("r16g16b16a16", (1920, 64), (480, 16), "mitchel")], ids=lambda c: "%s-%dx%d-%s" % (c[0], c[1][0], c[1][1], c[3]))
def test_edge_kernel_block_rows_match_oracle(case):
    """r05, third session: edge_rows caps the output rows per block of resize_edges_kernel (more, shorter waves; the block's LDS is sized for what it holds, not 8 KB):
    every cap gives the oracle's bytes."""
    pixel, (sw, sh), (dw, dh), filt = case
    rng = np.random.default_rng(sw + dh)
    img = rand_image(rng, sw, sh, pixel, pad=4, edge_values=True)
    opts = {"width": dw, "height": dh}
    if filt:
        opts["filter"] = filt
    want = oracle_resize(img, dw, dh, filt or "cubic", 1.0 if filt else 0.70)
    try:
        for er in (0, 1, 5, 16, 33, 64):
            N.check(N.lib().picha_hip_set_option(b"edge_rows", er))
            for chunks in (0, 3):
                N.check(N.lib().picha_hip_set_option(b"row_chunks", chunks))
                got = picha.resizeSync(img, opts)
                assert max_code_diff(got, want) == 0, (case, er, chunks, N.lib().picha_hip_last_kernel())
    finally:
        N.check(N.lib().picha_hip_set_option(b"edge_rows", 0))
        N.check(N.lib().picha_hip_set_option(b"row_chunks", 0))


# (collected only for a -DPICHA_RUN_EFUSE measurement build: PICHA_TEST_EFUSE=1 PICHA_HIP_LIB=<that build>; green in scripts/sessions/r05_call50.sh)
def _edge_blocks_inside_the_grid_match_oracle(geom):
    """r05, third session: run_efuse = 1 — cfg3b's shape (grey 7.5x, 30 taps, shared unpack) carries the edge pass as two more workgroups per (row chunk, image) of
    the march's own plain grid (",edgeblocks" in the name, no resize_edges launch); the march's side lanes store only the columns that are not clipped, so the
    two writers never touch the same byte.  Bit-exact against the oracle over row chunkings, workgroup widths, strides and sub-views (fma: <= 1); forms that cannot carry
    the blocks (item lists, one-wave workgroups) fall back to the edge kernel."""
    (sw, sh), (dw, dh) = geom
    rng = np.random.default_rng(sw * 7 + dh)
    big = rand_image(rng, sw + 8, sh + 3, "grey", pad=int(rng.choice([0, 4, 12])), edge_values=True)
    keys = (b"run", b"run_shu", b"run_efuse", b"row_chunks", b"work", b"block_waves", b"numerics")
    opts = {"width": dw, "height": dh, "filter": "lanczos"}
    try:
        for k, v in ((b"run", 1), (b"run_shu", 1), (b"run_efuse", 1)):
            N.check(N.lib().picha_hip_set_option(k, v))
        picha.resizeSync(big.subView(0, 0, sw, sh), opts)
        if ",edgeblocks" not in N.lib().picha_hip_last_kernel().decode():
            pytest.skip("the in-grid edge blocks are a measurement build (-DPICHA_RUN_EFUSE): slower than the edge kernel in the reference flavour, profiles/r05_ab_edge_blocks.txt")
        for img in (big.subView(0, 0, sw, sh), big.subView(5, 1, sw, sh)):
            want = oracle_resize(img, dw, dh, "lanczos", 1.0)
            for chunks, work, bw in ((0, 0, 0), (1, 0, 2), (3, 0, 4), (7, 0, 2), (2, 1, 2), (2, 0, 1)):
                for k, v in ((b"row_chunks", chunks), (b"work", work), (b"block_waves", bw)):
                    N.check(N.lib().picha_hip_set_option(k, v))
                for numerics, tol in ((0, 0), (1, 1)):
                    N.check(N.lib().picha_hip_set_option(b"numerics", numerics))
                    got = picha.resizeSync(img, opts)
                    name = N.lib().picha_hip_last_kernel().decode()
                    assert name.startswith("resize_run<ch1,u8,Q2,PX15,TX30,P4") and ",shu" in name, name
                    if work == 0 and bw != 1:
                        assert ",edgeblocks" in name and "resize_edges" not in name, name
                    else:
                        assert "+ resize_edges<" in name, name
                    assert max_code_diff(got, want) <= tol, (geom, chunks, work, bw, numerics, name)
                N.check(N.lib().picha_hip_set_option(b"numerics", 0))
    finally:
        for k in keys:
            N.check(N.lib().picha_hip_set_option(k, 0))


if os.environ.get("PICHA_TEST_EFUSE"):
    test_run_kernel_edge_blocks_inside_the_grid_match_oracle = pytest.mark.gpu(pytest.mark.parametrize(
        "geom", [((3840, 300), (512, 40)), ((3840, 2160), (512, 512)), ((1935, 158), (258, 21)), ((7680, 127), (1024, 17))],
        ids=lambda g: "%dx%d-%dx%d" % (g[0] + g[1]))(_edge_blocks_inside_the_grid_match_oracle))


@pytest.mark.parametrize("edges", [1, 2, 3])
@pytest.mark.parametrize("case", [("greya", (960, 130), (128, 31), "lanczos"), ("grey", (960, 130), (128, 31), "lanczos"), ("rgba", (320, 96), (640, 192), None),
                                  ("grey", (1920, 216), (960, 108), "lanczos"), ("r16", (320, 96), (640, 192), "mitchel"), ("rgb", (333, 50), (666, 100), "catmulrom"),
                                  ("greya", (250, 40), (1000, 160), "cubic"), ("r16g16", (131, 33), (262, 66), None)], ids=lambda c: "%s-%dx%d-%s" % (c[0], c[1][0], c[1][1], c[3]))
def test_edge_columns_from_the_stash_or_the_image_match_oracle(case, edges):
    """r05: run_edges — resize_edges_kernel takes the clipped columns' windows from the stash the march writes (1) or from the image (2); 3 = auto, which picks per
    kernel family by measurement and, for the short-window upscale shapes, lets the march produce those columns itself from their own weights (no edge kernel:
    ",edges" in the name).  All against the oracle on the run, ring and upscale kernels, both flavours."""
    pixel, (sw, sh), (dw, dh), filt = case
    rng = np.random.default_rng(sw + dh + edges)
    img = rand_image(rng, sw, sh, pixel, pad=4, edge_values=True)
    opts = {"width": dw, "height": dh}
    if filt:
        opts["filter"] = filt
    want = oracle_resize(img, dw, dh, filt or "cubic", 1.0 if filt else 0.70)
    try:
        for k, v in ((b"run", 1), (b"up", 1), (b"run_shu", 2), (b"run_edges", edges)):
            N.check(N.lib().picha_hip_set_option(k, v))
        for numerics in (0, 1):
            N.check(N.lib().picha_hip_set_option(b"numerics", numerics))
            got = picha.resizeSync(img, opts)
            name = N.lib().picha_hip_last_kernel().decode()
            inside = edges == 3 and name.startswith(("resize_up<", "resize_run<")) and ",edges" in name
            if name.startswith(("resize_up<", "resize_ring<", "resize_run<")):   # (odd sizes: the stash forms may decline and leave the geometry to another kernel)
                assert inside != ("+ resize_edges<" in name), name
            if edges == 3 and dw == 2 * sw:
                assert inside, name   # (the 2x shapes: three or four taps, runs of four or eight outputs)
            assert max_code_diff(got, want) <= numerics, (case, edges, numerics, name)
    finally:
        for k in (b"run", b"up", b"run_shu", b"run_edges", b"numerics"):
            N.check(N.lib().picha_hip_set_option(k, 0))


@pytest.mark.parametrize("case", [
    ("rgba", "grey", 1920, 216, None), ("rgba", "greya", 1920, 216, None), ("rgba", "rgb", 1920, 216, None), ("rgba", "r16", 1536, 120, None),
    ("rgb", "grey", 1920, 216, None), ("r16g16b16a16", "rgba", 1536, 120, None), ("r16g16b16a16", "grey", 1536, 120, None),
], ids=lambda c: "%s-%s-%s" % (c[0], c[1], c[4]))
def test_resize_then_convert_epilogue_on_the_symmetric_table_kernel(case):
    """r04: resize -> colorConvert as ONE kernel for the 2x shapes whose chain used to be resize_sym + a convert
    kernel: the convert is a store epilogue of resize_sym (src/resize.cc:130 then src/colorconvert.cc:143-150).  Against oracle
    resize -> oracle colorConvert (bit-exact; fma within one code value of the resize's sample type, i.e. <= 1 after an 8-bit
    convert, more after 16 -> 8 bit luma sums), byte-equal to the two-kernel chain, default and custom luma weights, chunkings."""
    from picha_amd.device import DeviceBatch, convert_resize_batch
    import torch
    src_pixel, dst_pixel, sw, sh, filt = case
    dw, dh = sw // 2, sh // 2
    fscale = 1.0 if filt else 0.70
    n = 3
    src = DeviceBatch(n, sw, sh, src_pixel).fill_random(31 + sw)
    keys = (b"fuse", b"row_chunks", b"work", b"numerics")
    try:
        custom = picha.color_settings(0.2, 0.5, 0.3)
        for weights in (None, custom):
            want = [oracle_convert(oracle_resize(src.image_host(i), dw, dh, filt or "cubic", fscale), dst_pixel, weights=weights) for i in range(n)]
            if src_pixel == "rgb":   # the symmetric-table kernel serves rgb at 2x in the fma flavour only (reference arithmetic: resize_stream)
                continue
            for chunks, work in ((0, 0), (3, 3), (2, 1)):
                N.check(N.lib().picha_hip_set_option(b"fuse", 1))
                N.check(N.lib().picha_hip_set_option(b"row_chunks", chunks))
                N.check(N.lib().picha_hip_set_option(b"work", work))
                out = convert_resize_batch(src, dst_pixel, dw, dh, filter=filt, order=1, weights=weights)
                torch.cuda.synchronize()
                name = N.lib().picha_hip_last_kernel().decode()
                assert name.startswith("resize_sym<") and ("epilogue -> %s" % dst_pixel) in name, name
                for i in range(n):
                    assert np.array_equal(out.image_host(i).pixels(), want[i].pixels()), (case, weights, chunks, work, i, name)
            N.check(N.lib().picha_hip_set_option(b"row_chunks", 0))
            N.check(N.lib().picha_hip_set_option(b"work", 0))
            N.check(N.lib().picha_hip_set_option(b"fuse", 2))     # the two-kernel chain: byte-equal
            two = convert_resize_batch(src, dst_pixel, dw, dh, filter=filt, order=1, weights=weights)
            torch.cuda.synchronize()
            assert "epilogue" not in N.lib().picha_hip_last_kernel().decode()
            for i in range(n):
                assert np.array_equal(two.image_host(i).pixels(), want[i].pixels())
        N.check(N.lib().picha_hip_set_option(b"fuse", 1))
        N.check(N.lib().picha_hip_set_option(b"numerics", 1))
        out = convert_resize_batch(src, dst_pixel, dw, dh, filter=filt, order=1, weights=custom)
        torch.cuda.synchronize()
        name = N.lib().picha_hip_last_kernel().decode()
        assert "epilogue -> %s" % dst_pixel in name and ",fma" in name, name
        # one code value of the RESIZE's sample type: an 8-bit step is 257 steps of a 16-bit destination
        tol = 257 if dst_pixel.startswith("r16") and not src_pixel.startswith("r16") else 1
        for i in range(n):
            assert max_code_diff(out.image_host(i), want[i]) <= tol, (case, name)
    finally:
        for k in keys:
            N.check(N.lib().picha_hip_set_option(k, 0))


CONVERT_UP_PAIRS = [   # (source, destination): the codec-edge pairs instantiated in kernels_run.hip (PICHA_UPCV_LIST)
    ("rgba", "rgb"), ("grey", "rgb"), ("r16g16b16", "rgb"), ("r16g16b16a16", "rgb"),
    ("rgb", "rgba"), ("greya", "rgba"), ("r16g16b16a16", "rgba"),
    ("rgb", "grey"), ("rgba", "grey"), ("greya", "grey"), ("r16", "grey"),
]


@pytest.mark.parametrize("pair", CONVERT_UP_PAIRS, ids=lambda c: "%s-%s" % c)
def test_convert_then_integer_upscale_as_one_kernel_equals_two_reference_calls(pair):
    """r04: colorConvert -> integer upscale as ONE kernel — resize_up with the convert per window pixel, the intermediate
    image's samples quantised in registers (src/colorconvert.cc:136-152, then src/resize.cc:99-132) — against oracle colorConvert -> oracle
    resize, bit for bit (fma <= 1), and byte-equal to the two-kernel chain: 2x and 4x in x, several vertical factors, padded strides, edge
    values, default and custom luma weights, row chunkings; geometries without an instantiation fall back to the chain."""
    sp, dp = pair
    rng = np.random.default_rng(7100 + CONVERT_UP_PAIRS.index(pair))
    served = 0
    keys = (b"fuse", b"row_chunks", b"numerics")
    geoms = [(96, 40, 192, 80, None, 0.70), (161, 30, 322, 75, None, 0.70), (130, 33, 260, 103, None, 0.70),
             (300, 21, 600, 42, None, 0.70), (80, 30, 320, 120, None, 0.70), (73, 25, 292, 77, None, 0.70),
             (96, 40, 192, 80, "mitchel", 1.0)]    # (4-tap filters at 2x: not instantiated with a convert in front -> the chain)
    try:
        for (sw, sh, dw, dh, filt, fscale) in geoms:
            img = rand_image(rng, sw, sh, sp, pad=int(rng.choice([0, 8])), edge_values=bool(rng.integers(0, 3) == 0))
            for weights in (None, (0.2, 0.5, 0.3)):
                cs = picha.color_settings(*weights) if weights else None
                want = oracle_resize(oracle_convert(img, dp, cs), dw, dh, filt or "cubic", fscale)
                opts = {"pixel": dp, "width": dw, "height": dh, "filterScale": fscale, "order": "convertFirst"}
                if filt: opts["filter"] = filt
                if weights: opts.update(redWeight=weights[0], greenWeight=weights[1], blueWeight=weights[2])
                for fuse, chunks in ((1, 0), (1, 3), (0, 0), (2, 0)):
                    N.check(N.lib().picha_hip_set_option(b"fuse", fuse))
                    N.check(N.lib().picha_hip_set_option(b"row_chunks", chunks))
                    got = picha.convertResizeSync(img, opts)
                    name = N.lib().picha_hip_last_kernel().decode()
                    one = name.startswith("resize_up<src ")
                    served += one and fuse == 1 and chunks == 0
                    assert not (fuse == 2 and one), name
                    assert (got.width, got.height, got.pixel) == (dw, dh, dp)
                    assert np.array_equal(payload(got), payload(want)), (pair, sw, sh, dw, dh, filt, weights, fuse, chunks, name, max_code_diff(got, want))
                N.check(N.lib().picha_hip_set_option(b"fuse", 1))
                N.check(N.lib().picha_hip_set_option(b"row_chunks", 0))
                N.check(N.lib().picha_hip_set_option(b"numerics", 1))
                got = picha.convertResizeSync(img, opts)
                assert max_code_diff(got, want) <= 1, (pair, sw, filt, "fma", N.lib().picha_hip_last_kernel())
                N.check(N.lib().picha_hip_set_option(b"numerics", 0))
    finally:
        for k in keys:
            N.check(N.lib().picha_hip_set_option(k, 0))
    four_x = (sp, dp) in (("rgba", "rgb"), ("grey", "rgb"), ("rgb", "rgba"), ("greya", "rgba"), ("rgb", "grey"), ("rgba", "grey"))
    assert served >= (12 if four_x else 8), served
