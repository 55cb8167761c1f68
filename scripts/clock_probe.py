"""In-kernel shader clock of the resize kernels under their own load (MI355X guide, DVFS item 6): a diagnostic build
(-DPICHA_CLOCK_PROBE, scripts/build_variant.sh) stamps s_memtime / s_memrealtime at the start and end of every wave;
clock = sum(d memtime) / sum(d memrealtime) x 100 MHz, read after >= 2 s of back-to-back launches on random pixels.

usage: PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_probe.so python scripts/clock_probe.py [label substring ...]
"""
import ctypes
import sys
import time

sys.path.insert(0, '.')
import torch

from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan

CASES = [
    # label, n, pixel, sw, sh, dw, dh, plan options, library options, probe symbol
    ("cfg2 exact", 256, "rgba", 3840, 2160, 1920, 1080, {}, {}, "picha_probe_sym_p1"),
    ("cfg2 fma", 256, "rgba", 3840, 2160, 1920, 1080, {}, {"numerics": 1}, "picha_probe_sym_p1"),
    ("cfg5 exact", 32, "r16g16b16a16", 7680, 4320, 1920, 1080, {"filter": "mitchel"}, {}, "picha_probe_run_p7"),
    ("cfg5 fma", 32, "r16g16b16a16", 7680, 4320, 1920, 1080, {"filter": "mitchel"}, {"numerics": 1}, "picha_probe_run_p7"),
    ("cfg3b exact", 256, "grey", 3840, 2160, 512, 512, {"filter": "lanczos"}, {}, "picha_probe_run_p2"),
    ("cfg3b fma", 256, "grey", 3840, 2160, 512, 512, {"filter": "lanczos"}, {"numerics": 1}, "picha_probe_run_p2"),
    # the secondary cells of VERDICT r03 (bit-exact flavour)
    ("lanczos 8x rgb thumbnails", 64, "rgb", 3840, 2160, 480, 270, {"filter": "lanczos"}, {}, "picha_probe_span_p0"),
    ("lanczos 2x rgba", 64, "rgba", 3840, 2160, 1920, 1080, {"filter": "lanczos"}, {}, "picha_probe_run_p1"),
    ("jpeg thumbnail rgb", 192, "rgb", 3840, 2160, 512, 288, {}, {}, "picha_probe_span_p0"),
    ("cfg1-size batch", 256, "rgba", 640, 480, 320, 240, {}, {}, "picha_probe_stream_p1"),
    ("cfg2 7-tap", 256, "rgba", 3840, 2160, 1920, 1080, {"filter": "cubic"}, {}, "picha_probe_stream_p1"),
]


def main():
    lib = N.lib()
    torch.cuda.set_device(0)
    print("library: %s" % N.SO_PATH)
    only = sys.argv[1:]   # optional: label substrings (e.g. "cfg2" "cfg5")
    for label, n, pixel, sw, sh, dw, dh, popt, lopt, sym in CASES:
        if only and not any(o in label for o in only):
            continue
        try:
            probe = getattr(lib, sym)
        except AttributeError:
            print("%-12s no probe symbol %s in this build" % (label, sym))
            continue
        probe.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
        for k, v in lopt.items():
            N.check(lib.picha_hip_set_option(k.encode(), v))
        src = DeviceBatch(n, sw, sh, pixel).fill_random(1)
        dst = DeviceBatch(n, dw, dh, pixel)
        plan = ResizePlan(src, dst, filter=popt.get("filter"), filter_scale=popt.get("filterScale"))
        plan.run()
        torch.cuda.synchronize()
        name = lib.picha_hip_last_kernel().decode()
        t0 = time.time()
        launches = 0
        while time.time() - t0 < 2.5:   # >= 2 s of back-to-back launches before the stamps that count
            for _ in range(20):
                plan.run()
            torch.cuda.synchronize()
            launches += 20
        SLOTS = 1 << 18
        out = (ctypes.c_ulonglong * (4 * SLOTS))()   # r05: 4 words per wave (d memtime, d memrealtime, start, HW_ID | XCC_ID << 32)
        assert probe(out, 1) == SLOTS
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 100
        e0.record()
        for _ in range(reps):
            plan.run()
        e1.record()
        torch.cuda.synchronize()
        assert probe(out, 0) == SLOTS
        import numpy as np
        a = np.frombuffer(out, dtype=np.uint64).reshape(SLOTS, 4)[:, :2].astype(np.float64)
        a = a[a[:, 1] > 0]
        if len(a) == 0:
            print("%-12s %s: no stamps (the kernel that ran lives in another translation unit than %s)" % (label, name, sym))
            continue
        ghz = a[:, 0] / a[:, 1] * 0.1
        ms = e0.elapsed_time(e1) / reps
        print("%-12s %-86s %.4f ms/launch  in-kernel clock: median %.3f GHz (p10 %.3f, p90 %.3f; by sums %.3f) over the last launch's %d wave slots, "
              "median wave life %.1f us; %d warm-up launches" % (label, name, ms, float(np.median(ghz)), float(np.percentile(ghz, 10)),
              float(np.percentile(ghz, 90)), float(a[:, 0].sum() / a[:, 1].sum() * 0.1), len(a), float(np.median(a[:, 1])) * 0.01, launches))
        for k in lopt:
            N.check(lib.picha_hip_set_option(k.encode(), 0))
        del plan, src, dst
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
