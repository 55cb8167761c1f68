"""Randomised parity sweep: geometry, format, filter, filterScale, stride padding, subView offsets and
kernel family drawn at random; every result must equal the oracle bit for bit (fma flavour: <= 1)."""
import numpy as np
import pytest

import picha_amd
from picha_amd import _native as N
from util import FILTERS, PIXELS, max_code_diff, oracle_convert, oracle_resize, payload, rand_image

pytestmark = pytest.mark.gpu


def set_opt(key, v):
    N.check(N.lib().picha_hip_set_option(key.encode(), v))


@pytest.fixture(autouse=True)
def _reset():
    yield
    for k in ("path", "numerics", "per_lane", "row_chunks", "stream_rot", "xcd_remap", "acc", "sym", "block_waves", "cc_mode", "acc_p2", "rows", "xch", "work", "up", "run", "run_round", "run_xch", "run_ws", "run_phase", "run_shu", "run_edges", "fuse"):
        set_opt(k, 0)


def _sizes(rng):
    kind = rng.integers(0, 10)
    if kind == 9:    # exact integer decimation, wide enough for interior waves of the symmetric-table kernel (several outputs per lane)
        f = int(rng.choice([2, 2, 2, 3, 4]))
        dw, dh = int(rng.integers(260, 1100)), int(rng.integers(3, 40))
        return dw * f, dh * f, dw, dh
    if kind == 8:    # very strong horizontal downscale: windows of 37 .. 128+ taps (long-window accumulator kernel, two-pass)
        dw, dh = int(rng.integers(1, 40)), int(rng.integers(1, 30))
        return dw * int(rng.integers(13, 44)) + int(rng.integers(0, 9)), dh * int(rng.integers(1, 9)) + int(rng.integers(0, 4)), dw, dh
    if kind == 6:    # tall upscales: several row chunks, many outputs per source row
        sw, sh = int(rng.integers(1, 50)), int(rng.integers(8, 60))
        return sw, sh, int(rng.integers(1, 120)), sh * int(rng.integers(2, 6)) + int(rng.integers(0, 9))
    if kind == 7:    # wide + tall enough for several workgroups and chunks, arbitrary ratio
        return int(rng.integers(300, 900)), int(rng.integers(40, 160)), int(rng.integers(260, 600)), int(rng.integers(33, 140))
    if kind == 0:    # exact integer decimation (streaming kernels)
        f = int(rng.choice([1, 2, 2, 2, 3, 4]))
        dw, dh = int(rng.integers(1, 300)), int(rng.integers(1, 120))
        return dw * f, dh * f, dw, dh
    if kind == 1:    # tiny
        return tuple(int(v) for v in rng.integers(1, 9, size=4))
    if kind == 2:    # strong downscale
        dw, dh = int(rng.integers(1, 40)), int(rng.integers(1, 30))
        return dw * int(rng.integers(3, 12)) + int(rng.integers(0, 5)), dh * int(rng.integers(3, 9)) + int(rng.integers(0, 4)), dw, dh
    if kind == 3:    # upscale
        sw, sh = int(rng.integers(1, 60)), int(rng.integers(1, 40))
        return sw, sh, sw * int(rng.integers(1, 5)) + int(rng.integers(0, 7)), sh * int(rng.integers(1, 4)) + int(rng.integers(0, 5))
    if kind == 4:    # wide outputs: several workgroups, ragged last lane group
        return int(rng.integers(600, 1500)), int(rng.integers(4, 40)), int(rng.integers(257, 700)), int(rng.integers(2, 30))
    return int(rng.integers(1, 200)), int(rng.integers(1, 150)), int(rng.integers(1, 200)), int(rng.integers(1, 150))


import os
N_SEEDS = int(os.environ.get("PICHA_FUZZ_SEEDS", "6"))   # a longer one-off sweep: PICHA_FUZZ_SEEDS=100
BASE = int(os.environ.get("PICHA_FUZZ_BASE", "1000"))     # ... over other random streams: PICHA_FUZZ_BASE=500000


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_resizes_match_oracle(seed):
    rng = np.random.default_rng(BASE + seed)
    checked = 0
    for _ in range(70):
        sw, sh, dw, dh = _sizes(rng)
        pixel = PIXELS[int(rng.integers(0, 8))]
        img = rand_image(rng, sw + int(rng.integers(0, 4)), sh + int(rng.integers(0, 3)), pixel, pad=int(rng.choice([0, 0, 4, 8, 1, 2, 3])),
                         edge_values=bool(rng.integers(0, 4) == 0))
        if img.width > sw or img.height > sh:  # exercise subViews (offset + parent stride; may be unaligned)
            img = img.subView(img.width - sw, img.height - sh, sw, sh)
        opts = {"width": dw, "height": dh}
        filt, fscale = "cubic", 0.70
        if rng.integers(0, 4) > 0:
            filt = FILTERS[int(rng.integers(0, 6))]
            opts["filter"] = filt
            fscale = 1.0
        if rng.integers(0, 3) == 0:
            fscale = float(rng.choice([0.5, 0.7, 1.0, 1.5, 2.25]))
            opts["filterScale"] = fscale
        try:
            exp = oracle_resize(img, dw, dh, filt, fscale)
        except RuntimeError as e:
            assert "rc=-7" in str(e)
            with pytest.raises(picha_amd.PichaError):
                picha_amd.resizeSync(img, opts)
            continue
        path = int(rng.choice([0, 0, 1, 3, 4]))
        numerics = int(rng.integers(0, 4) == 0)
        set_opt("path", path)
        set_opt("numerics", numerics)
        set_opt("per_lane", int(rng.choice([0, 0, 0, 1, 2, 3, 4])))
        set_opt("sym", int(rng.choice([0, 0, 2])))
        set_opt("block_waves", int(rng.choice([0, 0, 1, 2, 4])))
        set_opt("row_chunks", int(rng.choice([0, 0, 1, 2, 7])))
        set_opt("stream_rot", int(rng.choice([0, 1, 2])))
        set_opt("xcd_remap", int(rng.choice([0, 1, 2])))
        set_opt("acc", int(rng.choice([0, 0, 1, 2])))
        set_opt("acc_p2", int(rng.choice([0, 0, 1, 2])))
        set_opt("rows", int(rng.choice([0, 0, 1, 2])))
        set_opt("xch", int(rng.choice([0, 0, 1, 2])))      # neighbour-exchange forms (r03)
        set_opt("run_xch", int(rng.choice([0, 1, 1, 2])))  # ... of the run kernel (r04), scalar phase weights, time-phased stores
        set_opt("run_ws", int(rng.choice([0, 0, 2])))
        set_opt("run_shu", int(rng.choice([0, 1, 2])))     # shared unpack (r05); where the clipped columns' windows come from
        set_opt("run_edges", int(rng.choice([0, 1, 2])))
        set_opt("run_phase", int(rng.choice([0, -1, 5, 9, 12, 20])))
        set_opt("work", int(rng.choice([0, 0, 1, 2, 3])))  # explicit item list / persistent launch / plain grid (r03)
        set_opt("up", int(rng.choice([0, 1, 2])))          # upscales on the accumulator kernel or the span kernel (r03)
        try:
            got = picha_amd.resizeSync(img, opts)
        except picha_amd.PichaError as e:
            if path not in (3, 4) or "unsupported" not in str(e):
                raise
            set_opt("path", 0)  # forcing a fused kernel is legitimately refused (tap count, alignment): use auto
            path = 0
            got = picha_amd.resizeSync(img, opts)
        d = max_code_diff(got, exp)
        name = N.lib().picha_hip_last_kernel().decode()
        assert d <= (1 if numerics and path != 1 else 0), "seed %d: %dx%d -> %dx%d %s opts %r path %d numerics %d kernel %s stride %d offset %d: diff %d" % (
            seed, sw, sh, dw, dh, pixel, opts, path, numerics, name, img.stride, img.data.ctypes.data % 16, d)
        checked += 1
    assert checked > 40


@pytest.mark.parametrize("seed", range(max(2, N_SEEDS // 3)))
def test_random_half_integer_downscales_match_oracle(seed):
    """r03: 7.5x or 1.5x in x (the run kernel's periods 2 / 15 and 2 / 3 pixels) with random heights, strides, sub-views, filters and launch options;
    grey and greya take the run kernel + edge kernel where an instantiation exists (21 / 31 taps), everything else falls through
    to the other kernels — all of it against the oracle."""
    rng = np.random.default_rng(BASE + 7000 + seed)
    ran = 0
    for _ in range(40):
        dw = 2 * int(rng.integers(2, 200))
        sw = dw * (15 if rng.integers(0, 3) else 3) // 2      # 7.5x, or 1.5x (period 2 / 3 source pixels)
        dh = int(rng.integers(1, 60))
        sh = dh * int(rng.integers(1, 9)) + int(rng.integers(1, 7))
        pixel = str(rng.choice(["grey", "grey", "greya", "greya", "rgb", "r16"]))
        img = rand_image(rng, sw + int(rng.integers(0, 2)) * 4, sh + int(rng.integers(0, 3)), pixel, pad=int(rng.choice([0, 0, 4, 8])),
                         edge_values=bool(rng.integers(0, 4) == 0))
        if img.width > sw or img.height > sh:
            img = img.subView(img.width - sw, img.height - sh, sw, sh)
        opts = {"width": dw, "height": dh}
        filt, fscale = "cubic", 0.70
        if rng.integers(0, 2):
            filt = str(rng.choice(["lanczos", "mitchel", "catmulrom", "cubic", "triangle"]))
            opts["filter"] = filt
            fscale = 1.0
        exp = oracle_resize(img, dw, dh, filt, fscale)
        numerics = int(rng.integers(0, 4) == 0)
        set_opt("numerics", numerics)
        set_opt("run", int(rng.choice([1, 1, 0])))
        set_opt("run_round", int(rng.choice([0, 0, 1, 2])))
        set_opt("run_shu", int(rng.choice([0, 1, 1, 2])))
        set_opt("run_edges", int(rng.choice([0, 0, 1, 2])))
        set_opt("per_lane", int(rng.choice([0, 0, 2, 4, 8])))
        set_opt("row_chunks", int(rng.choice([0, 0, 1, 2, 5])))
        set_opt("work", int(rng.choice([0, 0, 1, 2, 3])))
        set_opt("block_waves", int(rng.choice([0, 0, 1, 2, 4])))
        got = picha_amd.resizeSync(img, opts)
        name = N.lib().picha_hip_last_kernel().decode()
        ran += name.startswith("resize_run<")
        d = max_code_diff(got, exp)
        assert d <= (1 if numerics else 0), "seed %d: %dx%d -> %dx%d %s opts %r numerics %d kernel %s stride %d offset %d: diff %d" % (
            seed, sw, sh, dw, dh, pixel, opts, numerics, name, img.stride, img.data.ctypes.data % 16, d)
    assert ran >= 3, ran   # (coverage, not parity: a seed draws ~40 geometries of which the periodic grey / greya ones take the run kernel)


def test_random_color_converts_match_oracle():
    rng = np.random.default_rng(77 if BASE == 1000 else BASE)
    for _ in range(200 * max(1, N_SEEDS // 6)):
        w, h = int(rng.integers(1, 300)), int(rng.integers(1, 40))
        sp, dp = PIXELS[int(rng.integers(0, 8))], PIXELS[int(rng.integers(0, 8))]
        img = rand_image(rng, w + int(rng.integers(0, 3)), h, sp, pad=int(rng.choice([0, 4])))
        if img.width > w:
            img = img.subView(img.width - w, 0, w, h)
        opts = {"pixel": dp}
        wts = None
        if rng.integers(0, 3) == 0:
            r, g, b = (float(v) for v in rng.uniform(0.05, 2.0, size=3))
            opts.update(redWeight=r, greenWeight=g, blueWeight=b)
            wts = picha_amd.color_settings(r, g, b)
        set_opt("cc_mode", int(rng.choice([0, 0, 1, 2, 3])))
        got = picha_amd.colorConvertSync(img, opts)
        exp = oracle_convert(img, dp, weights=wts)
        assert np.array_equal(payload(got), payload(exp)), (w, h, sp, dp, opts)


N_DEV = int(os.environ.get("PICHA_FUZZ_DEVICE_SEEDS", "8"))


@pytest.mark.parametrize("seed", range(N_DEV))
def test_random_device_batches_match_oracle(seed):
    """The device entry points with images placed at arbitrary byte offsets of one device buffer (base pointers that
    are NOT 4-byte aligned, non-uniform spacing -> pointer tables, mixed geometries in one call -> grouping), including
    the chained convert+resize call.  The host entry points always stage into 256-byte aligned arenas, so only this
    test reaches the kernels' handling of unaligned image bases (re-based descriptors, alignment fall-backs)."""
    import ctypes

    import torch
    rng = np.random.default_rng(BASE * 7 + seed)
    L = N.lib()
    pixel = PIXELS[int(rng.integers(0, 8))]
    bpp = N.PIXEL_BYTES[pixel]
    geoms = []
    for _ in range(int(rng.integers(1, 3))):
        f = int(rng.choice([2, 2, 3, 4, 0, -2, -4]))
        if f < 0:   # integer upscales (resize_up; with a convert in front: its one-kernel form)
            sw, sh = int(rng.integers(4, 120)), int(rng.integers(4, 40))
            geoms.append((sw, sh, sw * -f, int(sh * rng.choice([2, 2.5, 3, 4]))))
        elif f:
            dw, dh = int(rng.integers(4, 200)), int(rng.integers(4, 70))
            geoms.append((dw * f, dh * f, dw, dh))
        else:
            geoms.append((int(rng.integers(8, 260)), int(rng.integers(8, 90)), int(rng.integers(4, 200)), int(rng.integers(4, 120))))
    filt = [None, None, "cubic", "triangle", "lanczos", "mitchel"][int(rng.integers(0, 6))]
    opts = {} if filt is None else {"filter": filt}
    chain_to = PIXELS[int(rng.integers(0, 8))] if rng.integers(0, 3) == 0 else None
    if chain_to and any(g[2] > g[0] for g in geoms) and rng.integers(0, 2):   # a pair the one-kernel convert -> upscale form is instantiated for
        pixel, chain_to = [("rgba", "rgb"), ("grey", "rgb"), ("r16g16b16", "rgb"), ("r16g16b16a16", "rgb"), ("rgb", "rgba"), ("greya", "rgba"), ("r16g16b16a16", "rgba"),
                           ("rgb", "grey"), ("rgba", "grey"), ("greya", "grey"), ("r16", "grey")][int(rng.integers(0, 11))]
        bpp = N.PIXEL_BYTES[pixel]
    order = int(rng.integers(0, 2))
    n = int(rng.integers(2, 7))
    which = [int(rng.integers(0, len(geoms))) for _ in range(n)]
    if rng.integers(0, 2):
        which.sort()          # equal geometries adjacent: grouped launches
    align = int(rng.choice([1, 1, 2, 4, 16]))      # byte alignment of every image base
    if pixel.startswith("r16") or (chain_to or "").startswith("r16"):
        align = max(align, 1)
    srcs, items, off = [], [], int(rng.integers(0, 64))
    for k in which:
        sw, sh, dw, dh = geoms[k]
        img = rand_image(rng, sw, sh, pixel, pad=int(rng.choice([0, 0, 4, 1, 2])))
        off = (off + align - 1) // align * align
        s_off = off
        off += img.data.size + int(rng.integers(0, 9))
        out_pixel = chain_to or pixel
        d_stride = ((N.PIXEL_BYTES[out_pixel] * dw + 3) & ~3) + int(rng.choice([0, 0, 4, 1]))
        off = (off + align - 1) // align * align
        d_off = off
        off += d_stride * dh + int(rng.integers(0, 9))
        items.append((img, s_off, d_off, d_stride, dw, dh, out_pixel))
    buf = torch.zeros(off + 64, dtype=torch.uint8, device="cuda")
    base = buf.data_ptr()
    for img, s_off, *_ in items:
        buf[s_off:s_off + img.data.size].copy_(torch.from_numpy(img.data))
    S = (N.CImage * n)()
    D = (N.CImage * n)()
    for i, (img, s_off, d_off, d_stride, dw, dh, out_pixel) in enumerate(items):
        S[i] = N.CImage(base + s_off, img.stride, img.width, img.height, N.pixel_index(pixel))
        D[i] = N.CImage(base + d_off, d_stride, dw, dh, N.pixel_index(out_pixel))
    from picha_amd.api import _resize_options, color_settings
    ftag, fscale = _resize_options(opts)
    r, g, b = color_settings()
    stream = torch.cuda.current_stream().cuda_stream
    set_opt("work", int(rng.choice([0, 1, 2, 3])))   # item lists / persistent launches index images through the pointer tables
    set_opt("xch", int(rng.choice([0, 1, 2])))
    set_opt("run_xch", int(rng.choice([0, 1, 2])))
    set_opt("run_phase", int(rng.choice([0, 5, 12, 20])))
    set_opt("up", int(rng.choice([0, 1, 2])))
    set_opt("fuse", int(rng.choice([0, 1, 1, 2])))
    if chain_to:
        N.check(L.picha_hip_convert_resize_device(n, S, D, order, r, g, b, ftag, fscale, stream))
    else:
        N.check(L.picha_hip_resize_device(n, S, D, ftag, fscale, stream))
    torch.cuda.synchronize()
    host = buf.cpu().numpy()
    for i, (img, s_off, d_off, d_stride, dw, dh, out_pixel) in enumerate(items):
        got = picha_amd.Image(width=dw, height=dh, pixel=out_pixel, stride=d_stride, data=host[d_off:d_off + d_stride * (dh - 1) + dw * N.PIXEL_BYTES[out_pixel]])
        f, fs = opts.get("filter", "cubic"), (1.0 if opts else 0.70)
        try:
            if chain_to and order == 0:
                exp = oracle_resize(oracle_convert(img, chain_to), dw, dh, f, fs)
            elif chain_to:
                exp = oracle_convert(oracle_resize(img, dw, dh, f, fs), chain_to)
            else:
                exp = oracle_resize(img, dw, dh, f, fs)
        except RuntimeError:
            continue
        assert np.array_equal(payload(got), payload(exp)), "seed %d image %d: %s -> %s %dx%d -> %dx%d opts %r chain %r order %d align %d src+%d dst+%d stride %d/%d kernel %s: diff %d" % (
            seed, i, pixel, out_pixel, img.width, img.height, dw, dh, opts, chain_to, order, align, s_off % 16, d_off % 16, img.stride, d_stride,
            L.picha_hip_last_kernel().decode(), max_code_diff(got, exp))


def test_random_async_requests_from_several_threads_match_oracle():
    """The batching dispatcher under a random mixed load: resize / colorConvert / chained requests of random geometry
    and format submitted concurrently from 4 threads; every callback must deliver the oracle's pixels."""
    import threading
    rng = np.random.default_rng(BASE + 424242)
    reqs = []
    for k in range(240 * max(1, N_SEEDS // 6)):
        pixel = PIXELS[int(rng.integers(0, 8))]
        sw, sh = int(rng.integers(4, 160)), int(rng.integers(4, 100))
        img = rand_image(rng, sw, sh, pixel, pad=int(rng.choice([0, 0, 4, 1])))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            dw, dh = (max(1, sw // 2), max(1, sh // 2)) if rng.integers(0, 2) else (int(rng.integers(1, 120)), int(rng.integers(1, 80)))
            o = {"width": dw, "height": dh}
            if rng.integers(0, 2):
                o["filter"] = FILTERS[int(rng.integers(0, 6))]
            try:
                want = oracle_resize(img, dw, dh, o.get("filter", "cubic"), 1.0 if "filter" in o else 0.70)
            except RuntimeError:
                continue
            reqs.append((picha_amd.resize, img, o, want))
        elif kind == 1:
            to = PIXELS[int(rng.integers(0, 8))]
            reqs.append((picha_amd.colorConvert, img, {"pixel": to}, oracle_convert(img, to)))
        else:
            to = PIXELS[int(rng.integers(0, 8))]
            dw, dh = int(rng.integers(1, 90)), int(rng.integers(1, 60))
            try:
                want = oracle_resize(oracle_convert(img, to), dw, dh, "cubic", 0.70)
            except RuntimeError:
                continue
            reqs.append((picha_amd.convertResize, img, {"pixel": to, "width": dw, "height": dh}, want))
    results, lock = {}, threading.Lock()

    def submit(tid):
        for i in range(tid, len(reqs), 4):
            fn, img, o, _ = reqs[i]
            fn(img, o, (lambda k: (lambda e, r: results.__setitem__(k, (e, r))))(i))
    threads = [threading.Thread(target=submit, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    picha_amd.drain()
    assert len(results) == len(reqs)
    for i, (fn, img, o, want) in enumerate(reqs):
        err, got = results[i]
        assert err is None, (i, err)
        assert np.array_equal(payload(got), payload(want)), (i, fn.__name__, img.pixel, img.width, img.height, o)
