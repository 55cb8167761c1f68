"""Neighbour-exchange forms against the planner's kernels over the dword-sized pixel formats and integer scales (r03).
usage: python scripts/survey_xch.py [n_images]   -> one line per (format, scale, filter): default ms, xch ms, kernels, parity"""
import sys
sys.path.insert(0, '.')
import numpy as np
import torch
import oracle
import picha_amd
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan

def opt(k, v):
    N.check(N.lib().picha_hip_set_option(k.encode(), int(v)))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for pixel in ("rgba", "r16g16", "r16g16b16a16"):
    bpp = N.PIXEL_BYTES[pixel]
    for scale, (dw, dh) in ((2, (1920, 1080)), (3, (1280, 720)), (4, (960, 540))):
        for filt in (None, "mitchel"):
            sw, sh = dw * scale, dh * scale
            nn = max(8, n * 8 // bpp // (scale * scale // 4 if scale > 2 else 1))
            src = DeviceBatch(nn, sw, sh, pixel).fill_random(3)
            dst = DeviceBatch(nn, dw, dh, pixel)
            plan = ResizePlan(src, dst, filter=filt)
            exp, _ = oracle.resize(src.image_host(0).data, sw, sh, pixel, dw, dh, filt or "cubic", 1.0 if filt else 0.70)
            expi = picha_amd.Image(width=dw, height=dh, pixel=pixel, data=exp).pixels().astype(np.int64)
            res = {}
            for rnd in range(3):
                for x in (2, 1):
                    opt("xch", x)
                    plan.run(); torch.cuda.synchronize()
                    if rnd == 0:
                        d = int(np.abs(dst.image_host(0).pixels().astype(np.int64) - expi).max())
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        plan.run()
                    e1.record(); torch.cuda.synchronize()
                    r = res.setdefault(x, {"t": [], "k": N.lib().picha_hip_last_kernel().decode(), "d": d})
                    r["t"].append(e0.elapsed_time(e1) / 10)
            opt("xch", 0)
            a, b = min(res[2]["t"]), min(res[1]["t"])
            alg = nn * (sw * sh + dw * dh) * bpp
            print("%-13s %dx %-8s n=%3d  no-xch %.3f ms (%.3f)  xch %.3f ms (%.3f)  %+5.1f %%  diff %d/%d  %s | %s" % (
                pixel, scale, filt or "default", nn, a, alg / (a * 1e-3) / 8e12, b, alg / (b * 1e-3) / 8e12, (b / a - 1) * 100,
                res[2]["d"], res[1]["d"], res[2]["k"], res[1]["k"]), flush=True)
            del src, dst, plan
