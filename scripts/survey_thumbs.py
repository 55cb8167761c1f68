"""Strong downscales (thumbnails: 10x-20x, windows of 40-128 taps): kernel, ms per launch, fraction of the 8 TB/s roof, parity of image 0.

usage: python scripts/survey_thumbs.py [n_images [pixels,comma,separated]]
"""
import sys

sys.path.insert(0, '.')
import numpy as np
import torch

import oracle
import picha_amd
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    pixels = sys.argv[2].split(',') if len(sys.argv) > 2 else ['grey', 'greya', 'rgb', 'rgba', 'r16g16b16a16']
    torch.cuda.set_device(0)
    geos = [(3840, 2160, 384, 216), (3840, 2160, 333, 187), (3840, 2160, 320, 180), (3840, 2160, 240, 135), (4032, 3024, 320, 240),
            (6000, 4000, 300, 200)]
    for pixel in pixels:
        bpp = N.PIXEL_BYTES[pixel]
        for (sw, sh, dw, dh) in geos:
            m = max(4, min(n, int(3.0e9 // (sw * sh * bpp))))
            src = DeviceBatch(m, sw, sh, pixel).fill_random(1)
            dst = DeviceBatch(m, dw, dh, pixel)
            host0 = src.image_host(0).data
            for filt in (None, 'lanczos', 'mitchel', 'triangle'):
                plan = ResizePlan(src, dst, filter=filt)
                plan.run()
                torch.cuda.synchronize()
                name = N.lib().picha_hip_last_kernel().decode()
                e, _ = oracle.resize(host0, sw, sh, pixel, dw, dh, filt or 'cubic', 1.0 if filt else 0.70)
                exp = picha_amd.Image(width=dw, height=dh, pixel=pixel, data=e).pixels().astype(np.int64)
                diff = int(np.abs(dst.image_host(0).pixels().astype(np.int64) - exp).max())
                best = 1e9
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        plan.run()
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 5)
                frac = m * (sw * sh + dw * dh) * bpp / (best * 1e-3) / 8e12
                print('%-13s n=%3d %4dx%-4d -> %4dx%-4d %-9s %8.3f ms  frac %.3f  diff %d  %s' % (pixel, m, sw, sh, dw, dh, filt or 'default', best, frac, diff, name), flush=True)
            del src, dst


main()
