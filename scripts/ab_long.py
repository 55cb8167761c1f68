"""A/B of the long-window accumulator kernel's workgroup size (64 / 128 / 192 / 256 threads against the planner's choice, "auto"),
interleaved on one box over thumbnail geometries; ms per launch, parity of image 0 for every form.

usage: python scripts/ab_long.py [n_images [pixels,comma,separated]]
"""
import sys

sys.path.insert(0, '.')
import numpy as np
import torch

import oracle
import picha_amd
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan

FORMS = [256, 192, 128, 64, 0]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    pixels = sys.argv[2].split(',') if len(sys.argv) > 2 else ['grey', 'greya', 'rgb', 'rgba', 'r16g16b16a16']
    torch.cuda.set_device(0)
    geos = [(3840, 2160, 384, 216), (3840, 2160, 333, 187), (3840, 2160, 320, 180), (3840, 2160, 240, 135), (4032, 3024, 320, 240),
            (6000, 4000, 300, 200), (3840, 2160, 160, 90), (3840, 2160, 256, 144), (5472, 3648, 512, 341)]
    lib = N.lib()
    for pixel in pixels:
        bpp = N.PIXEL_BYTES[pixel]
        for (sw, sh, dw, dh) in geos:
            m = max(4, min(n, int(3.0e9 // (sw * sh * bpp))))
            src = DeviceBatch(m, sw, sh, pixel).fill_random(1)
            dst = DeviceBatch(m, dw, dh, pixel)
            host0 = src.image_host(0).data
            for filt in (None, 'lanczos', 'mitchel'):
                e, _ = oracle.resize(host0, sw, sh, pixel, dw, dh, filt or 'cubic', 1.0 if filt else 0.70)
                exp = picha_amd.Image(width=dw, height=dh, pixel=pixel, data=e).pixels().astype(np.int64)
                plan = ResizePlan(src, dst, filter=filt)
                best = {f: 1e9 for f in FORMS}
                diffs = {}
                name = ''
                for rep in range(3):
                    for f in FORMS:
                        N.check(lib.picha_hip_set_option(b"long_block", f))
                        plan.run()
                        torch.cuda.synchronize()
                        if rep == 0:
                            name = lib.picha_hip_last_kernel().decode()
                            diffs[f] = int(np.abs(dst.image_host(0).pixels().astype(np.int64) - exp).max())
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(5):
                            plan.run()
                        e1.record()
                        torch.cuda.synchronize()
                        best[f] = min(best[f], e0.elapsed_time(e1) / 5)
                N.check(lib.picha_hip_set_option(b"long_block", 0))
                if 'acc_long' not in name:
                    continue
                ref = best[256]
                print('%-13s n=%3d %4dx%-4d -> %4dx%-4d %-8s %-34s' % (pixel, m, sw, sh, dw, dh, filt or 'default', name),
                      '  '.join('%s %.3f (%+.0f%%) d%d' % ('b%d' % f if f else 'auto', best[f], 100 * (best[f] / ref - 1), diffs[f]) for f in FORMS), flush=True)
            del src, dst


main()
