"""BASELINE.json's full sizes on the GPU, checked through properties that do not need a CPU pass over
the whole batch: agreement between independent kernel families on a checksum of the entire output,
per-image independence of batched launches, constants staying constant, colour-convert round trips,
plus an oracle spot check of single images."""
import hashlib

import numpy as np
import pytest
import torch

import picha_amd
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, color_convert_batch, resize_batch
from util import oracle_convert, oracle_resize

pytestmark = pytest.mark.gpu


def set_opt(key, v):
    N.check(N.lib().picha_hip_set_option(key.encode(), v))


@pytest.fixture(autouse=True)
def _reset():
    yield
    for k in ("path", "numerics", "per_lane", "row_chunks", "sym", "block_waves", "xch", "run_phase", "run_xch"):
        set_opt(k, 0)


def digest(batch):
    """sha1 over the payload of every image (row padding excluded: it is never written)."""
    h = hashlib.sha1()
    bpp = N.PIXEL_BYTES[batch.pixel]
    rows = batch.tensor[:, : batch.height * batch.stride].reshape(batch.n, batch.height, batch.stride)[:, :, : batch.width * bpp]
    for i in range(batch.n):
        h.update(rows[i].contiguous().cpu().numpy().tobytes())
    return h.hexdigest()


def test_cfg2_full_batch_kernel_families_agree_and_match_oracle():
    """configs[1]: 256 x 3840x2160 rgba -> 1920x1080, picha default (cubic @ 0.70)."""
    n = 256
    src = DeviceBatch(n, 3840, 2160, "rgba").fill_random(20250226)
    sums = {}
    # (0, 0) = the planner's own choice: the launch bench.py times (all 256 images of it are in the digest); then the
    # streaming and LDS-ring kernels in two lane layouts each, and both forms of the symmetric-table kernel (with and
    # without the neighbour exchange)
    for path, per_lane, xch in ((0, 0, 0), (2, 2, 0), (2, 1, 0), (3, 2, 0), (3, 1, 0), (0, 3, 1), (0, 3, 2)):
        set_opt("path", path)
        set_opt("per_lane", per_lane)
        set_opt("xch", xch)
        out = resize_batch(src, 1920, 1080)
        torch.cuda.synchronize()
        sums["%s (path %d, per_lane %d, xch %d)" % (N.lib().picha_hip_last_kernel().decode(), path, per_lane, xch)] = digest(out)
    assert len({k.split(" (")[0] for k in sums}) >= 6 and len(set(sums.values())) == 1, sums
    set_opt("xch", 0)
    # oracle spot check: first and last image of the batch (bit-exact)
    for i in (0, n - 1):
        exp = oracle_resize(src.image_host(i), 1920, 1080, "cubic", 0.70)
        assert np.array_equal(out.image_host(i).pixels(), exp.pixels())
    # batched launch == the same image resized alone through the host-pointer entry point
    set_opt("path", 0)
    set_opt("per_lane", 0)
    alone = picha_amd.resizeSync(src.image_host(7), {"width": 1920, "height": 1080})
    assert np.array_equal(alone.pixels(), out.image_host(7).pixels())
    # two-pass scratch kernels on one full-size image
    set_opt("path", 1)
    two = picha_amd.resizeSync(src.image_host(7), {"width": 1920, "height": 1080})
    assert np.array_equal(two.pixels(), alone.pixels())


def test_cfg2_constant_images_stay_constant():
    """Normalised taps: a constant image resamples to the same constant, for every value and filter."""
    src = DeviceBatch(8, 3840, 2160, "rgba")
    vals = [0, 1, 77, 128, 200, 254, 255, 31]
    for i, v in enumerate(vals):
        src.tensor[i].fill_(v)
    for filt in (None, "lanczos", "mitchel"):
        out = resize_batch(src, 1920, 1080, filter=filt)
        torch.cuda.synchronize()
        for i, v in enumerate(vals):
            img = out.tensor[i, : 1080 * out.stride]
            assert int(img.min()) == v and int(img.max()) == v, (filt, v)


def test_cfg3_full_size_convert_then_lanczos_pipeline():
    """configs[2]: 3840x2160 rgb -> grey colorConvert, then -> 512x512 lanczos."""
    n = 64
    src = DeviceBatch(n, 3840, 2160, "rgb").fill_random(3)
    grey = color_convert_batch(src, "grey")
    small = resize_batch(grey, 512, 512, filter="lanczos")
    torch.cuda.synchronize()
    for i in (0, n - 1):
        g = oracle_convert(src.image_host(i), "grey")
        assert np.array_equal(grey.image_host(i).pixels(), g.pixels())
        s = oracle_resize(g, 512, 512, "lanczos", 1.0)
        assert np.array_equal(small.image_host(i).pixels(), s.pixels())
    # grey -> rgb -> grey is the identity (L = (v*r + v*g) + v*b with normalised weights)
    back = color_convert_batch(color_convert_batch(grey, "rgb"), "grey")
    torch.cuda.synchronize()
    assert digest(back) == digest(grey)
    # 8-bit -> 16-bit -> 8-bit is the identity too
    wide = color_convert_batch(src, "r16g16b16")
    narrow = color_convert_batch(wide, "rgb")
    torch.cuda.synchronize()
    assert digest(narrow) == digest(src)


def test_cfg5_full_size_16bit_mitchel():
    """configs[4] geometry: 7680x4320 r16g16b16a16 -> 1920x1080 mitchel.  The planner's own choice (path 0) runs at the FULL
    size — 61 440-byte source rows, 265 MB per image: 32-bit row * stride and descriptor arithmetic that a small image
    cannot exercise — and must agree with the span and two-pass kernels on every byte and with the oracle on the first
    and last image."""
    n = 3
    src = DeviceBatch(n, 7680, 4320, "r16g16b16a16").fill_random(5)
    sums = {}
    # planner's choice (r04: the neighbour-exchange run kernel with wave-uniform weights; 3 images are too few for its time-phased
    # stores, so the phased form is forced as well), the r03 neighbour-exchange accumulator kernel it replaced (xch = 1), the
    # one-output accumulator kernel before that (xch = 2), LDS-ring span kernel, two-pass
    for path, xch, phase in ((0, 0, 0), (0, 0, 12), (0, 1, 0), (4, 2, 0), (3, 0, 0), (1, 0, 0)):
        set_opt("path", path)
        set_opt("xch", xch)
        set_opt("run_phase", phase)
        out = resize_batch(src, 1920, 1080, filter="mitchel")
        torch.cuda.synchronize()
        name = N.lib().picha_hip_last_kernel().decode()
        if path == 0 and xch == 0:
            assert name.startswith("resize_run<ch4,u16,Q1,PX4,TX15,P2,NACC4,exact,bw4,xch5") and ("phased12" in name) == (phase == 12), name   # the kernel bench.py times for cfg5
            planner = out
        if path == 0 and xch == 1:
            assert name.startswith("resize_acc<ch4,u16,P2,S4,TX15,NACC4,exact,xch"), name
        if path == 4:
            assert name.startswith("resize_acc<ch4,u16,TX16,NACC4"), name
        sums[(path, xch, phase)] = digest(out)
    set_opt("xch", 0)
    set_opt("run_phase", 0)
    assert len(set(sums.values())) == 1, sums
    for i in (0, n - 1):
        exp = oracle_resize(src.image_host(i), 1920, 1080, "mitchel", 1.0)
        assert np.array_equal(planner.image_host(i).pixels(), exp.pixels()), i


def test_cfg5_per_gpu_batch_of_128_images_kernel_families_agree():
    """configs[4] at its per-GPU batch: 128 x 8K r16g16b16a16 (34 GB resident) -> 1080p mitchel.  Too much for the oracle,
    so the size-independent property: the planner's kernel and the span kernel produce the same digest over all 128
    outputs, and the oracle confirms the last image (whose bytes sit 34 GB into the batch: 64-bit image pitch)."""
    n = 128
    src = DeviceBatch(n, 7680, 4320, "r16g16b16a16").fill_random(6)
    set_opt("path", 0)
    a = resize_batch(src, 1920, 1080, filter="mitchel")
    torch.cuda.synchronize()
    name = N.lib().picha_hip_last_kernel().decode()
    assert name.startswith("resize_run<ch4,u16,Q1,PX4,TX15,P2,NACC4,exact,bw4,xch5,phased12"), name   # time-phased stores at this size
    da = digest(a)
    set_opt("path", 3)
    b = resize_batch(src, 1920, 1080, filter="mitchel", out=a)   # reuse the destination batch
    torch.cuda.synchronize()
    assert N.lib().picha_hip_last_kernel().decode().startswith("resize_span")
    assert digest(b) == da
    exp = oracle_resize(src.image_host(n - 1), 1920, 1080, "mitchel", 1.0)
    assert np.array_equal(b.image_host(n - 1).pixels(), exp.pixels())


# the format pairs the codecs imply (reference index.js:37-72: supportedMap x each codec's `encodes`; README.md:158-159: tiff
# always decodes to rgba; src/pngcodec.cc:61-86: `deep` png decodes to the 16-bit formats; src/jpegcodec.cc:512-514 and
# src/webpcodec.cc:399-401: jpeg encodes rgb / grey, webp encodes rgb / rgba)
CODEC_EDGE_PAIRS = [
    ("rgba", "rgb"), ("greya", "grey"), ("r16", "grey"), ("r16g16", "grey"), ("r16g16b16", "rgb"), ("r16g16b16a16", "rgb"),   # -> jpeg
    ("grey", "rgb"), ("greya", "rgba"), ("r16", "rgb"), ("r16g16", "rgb"), ("r16g16b16a16", "rgba"),                          # -> webp
    ("rgba", "grey"), ("rgb", "grey"), ("r16g16b16", "rgba"),                                                                  # decode -> analysis
]


@pytest.mark.parametrize("pair", CODEC_EDGE_PAIRS, ids=lambda p: "%s-%s" % p)
def test_codec_edge_color_converts_at_4k(pair):
    """Full-size colour converts for every pair a codec boundary implies: the whole batch is checked on the device against the
    4-pixel kernel (cc_mode 1), two images against the oracle."""
    sp, dp = pair
    n = 6
    src = DeviceBatch(n, 3840, 2160, sp).fill_random(31)
    out = color_convert_batch(src, dp)
    torch.cuda.synchronize()
    set_opt("cc_mode", 1)
    try:
        ref = color_convert_batch(src, dp)
        torch.cuda.synchronize()
    finally:
        set_opt("cc_mode", 0)
    assert torch.equal(out.tensor, ref.tensor)
    for i in (0, n - 1):
        exp = oracle_convert(src.image_host(i), dp)
        assert np.array_equal(out.image_host(i).pixels(), exp.pixels())


def test_non_uniform_pitch_batches_use_pointer_tables():
    """Images of one call that are not equally spaced in memory go through device pointer tables."""
    import ctypes
    a = DeviceBatch(5, 640, 480, "rgba").fill_random(11)
    out = DeviceBatch(5, 320, 240, "rgba")
    order = [3, 0, 4, 1, 2]  # permuted: data pointers are no longer base + i * pitch
    S = (N.CImage * 5)(*[a.descriptors()[i] for i in order])
    D = (N.CImage * 5)(*[out.descriptors()[i] for i in order])
    stream = torch.cuda.current_stream().cuda_stream
    N.check(N.lib().picha_hip_resize_device(5, S, D, 0, ctypes.c_float(0.70), stream))
    torch.cuda.synchronize()
    for i in range(5):
        exp = oracle_resize(a.image_host(i), 320, 240, "cubic", 0.70)
        assert np.array_equal(out.image_host(i).pixels(), exp.pixels())


def test_device_entry_point_is_graph_capturable():
    """With the tap tables cached the device entry point neither allocates nor synchronises, so a batch
    resize can be captured into a HIP graph (torch.cuda.CUDAGraph on the capture stream) and replayed."""
    src = DeviceBatch(16, 1280, 720, "rgba").fill_random(21)
    dst = DeviceBatch(16, 640, 360, "rgba")
    from picha_amd.device import ResizePlan
    plan = ResizePlan(src, dst)
    plan.run()  # warm: builds and uploads the tables (allocation + upload happen here, outside the capture)
    torch.cuda.synchronize()
    want = digest(dst)
    dst.tensor.zero_()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            plan.run(stream=s.cuda_stream)
    torch.cuda.synchronize()
    assert digest(dst) != want  # capture only records
    g.replay()
    torch.cuda.synchronize()
    assert digest(dst) == want
    src.fill_random(22)   # new pixels, same graph
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    exp = oracle_resize(src.image_host(3), 640, 360, "cubic", 0.70)
    assert np.array_equal(dst.image_host(3).pixels(), exp.pixels())


@pytest.mark.parametrize("pixel", ["rgb", "grey", "greya"])
def test_full_size_subdword_2x_streaming_equals_span_kernel_and_oracle(pixel):
    """4K -> 1080p in the sub-dword formats: the multi-output streaming kernel (byte-rotated loads, whole-dword
    stores) and the LDS-ring span kernel must agree on every byte of a full batch (compared on the device with
    picha_hip_image_compare_device) and match the oracle on sample images."""
    n = 48
    src = DeviceBatch(n, 3840, 2160, pixel).fill_random(11)
    out = resize_batch(src, 1920, 1080)
    torch.cuda.synchronize()
    name = N.lib().picha_hip_last_kernel().decode()
    assert (name.startswith("resize_sym") or name.startswith("resize_stream")) and (",P4," in name or ",P2," in name), name
    set_opt("path", 3)
    ref = resize_batch(src, 1920, 1080)
    torch.cuda.synchronize()
    assert N.lib().picha_hip_last_kernel().decode().startswith("resize_span")
    set_opt("path", 0)
    assert all(out.equal_pixels(ref))
    assert out.avg_channel_diff(ref) == [0.0] * n
    for i in (0, n - 1):
        exp = oracle_resize(src.image_host(i), 1920, 1080, "cubic", 0.70)
        assert np.array_equal(out.image_host(i).pixels(), exp.pixels())
    set_opt("numerics", 1)
    fma = resize_batch(src, 1920, 1080)
    torch.cuda.synchronize()
    set_opt("numerics", 0)
    d = fma.avg_channel_diff(out)
    assert max(d) < 0.25 and not all(fma.equal_pixels(out)), d   # a few samples differ by one code value, no more


def test_cfg3_full_size_chained_call_equals_the_two_calls():
    from picha_amd.device import convert_resize_batch
    n = 64
    src = DeviceBatch(n, 3840, 2160, "rgb").fill_random(3)
    two = resize_batch(color_convert_batch(src, "grey"), 512, 512, filter="lanczos")
    one = convert_resize_batch(src, "grey", 512, 512, filter="lanczos")
    torch.cuda.synchronize()
    assert all(one.equal_pixels(two))
    other = convert_resize_batch(src, "grey", 512, 512, filter="lanczos", order=1)   # resize in rgb first, then convert
    ref = color_convert_batch(resize_batch(src, 512, 512, filter="lanczos"), "grey")
    torch.cuda.synchronize()
    assert all(other.equal_pixels(ref))
    assert not all(other.equal_pixels(one))   # the two orders are different operations


@pytest.mark.parametrize("n,pixel,dw,dh,filt", [(64, "grey", 384, 216, "cubic"), (256, "grey", 384, 216, "cubic"), (128, "greya", 333, 187, "mitchel")],
                         ids=lambda v: str(v))
def test_long_window_thumbnails_at_the_planners_chunking_match_other_chunkings_and_oracle(n, pixel, dw, dh, filt):
    """r05, third session: 4K -> thumbnails at >= 10x with explicit filters (48-tap windows) run on resize_acc_long, whose chunk count for one- and two-byte
    pixels now fills the queried residency up to one free workgroup slot per CU, or cuts a launch of several rounds back to one round of long chunks
    (256 images: two chunks of 108 rows).  Full batches at the planner's own chunking against forced chunkings (same bytes whatever the chunking) and
    the oracle on the first and last image."""
    src = DeviceBatch(n, 3840, 2160, pixel).fill_random(11)
    sums = {}
    for chunks in (0, 6, 1, 27):
        set_opt("row_chunks", chunks)
        out = resize_batch(src, dw, dh, filter=filt)
        torch.cuda.synchronize()
        name = N.lib().picha_hip_last_kernel().decode()
        assert name.startswith("resize_acc_long<ch%d,u8,TX48" % N.PIXEL_CHANNELS[pixel]), name
        sums[chunks] = digest(out)
        if chunks == 0:
            planner = out
    assert len(set(sums.values())) == 1, sums
    for i in (0, n - 1):
        exp = oracle_resize(src.image_host(i), dw, dh, filt, 1.0)
        assert np.array_equal(planner.image_host(i).pixels(), exp.pixels()), i
