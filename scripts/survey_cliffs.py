"""Looks for performance cliffs: every filter x a ladder of scale factors x a few formats on one box, default options; prints the
kernel that ran, ms per launch, the fraction of the 8 TB/s roof (read + write bytes) and the parity of image 0, sorted by fraction.

usage: python scripts/survey_cliffs.py [n_images [pixels,comma,separated]]
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    pixels = sys.argv[2].split(',') if len(sys.argv) > 2 else ['grey', 'rgb', 'rgba', 'r16g16b16a16']
    torch.cuda.set_device(0)
    down = [(3840, 2160, w, h) for (w, h) in [(2560, 1440), (1920, 1080), (1536, 864), (1280, 720), (960, 540), (768, 432), (640, 360), (512, 288),
                                             (480, 270), (384, 216), (320, 180), (240, 135), (1000, 563), (333, 187)]]
    up = [(960, 540, w, h) for (w, h) in [(1440, 810), (1920, 1080), (2880, 1620), (3840, 2160), (1000, 563)]]
    filters = [(None, None)] + [(f, None) for f in ('cubic', 'lanczos', 'catmulrom', 'mitchel', 'box', 'triangle')]
    rows = []
    for pixel in pixels:
        bpp = N.PIXEL_BYTES[pixel]
        for (sw, sh, dw, dh) in down + up:
            src = DeviceBatch(n, sw, sh, pixel).fill_random(1)
            dst = DeviceBatch(n, dw, dh, pixel)
            host0 = src.image_host(0).data
            for filt, fs in filters:
                plan = ResizePlan(src, dst, filter=filt, filter_scale=fs)
                try:
                    plan.run()
                    torch.cuda.synchronize()
                except Exception as ex:
                    rows.append((0.0, pixel, sw, sh, dw, dh, filt, 'ERROR %s' % ex, 0.0, -1))
                    continue
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
                frac = n * (sw * sh + dw * dh) * bpp / (best * 1e-3) / 8e12
                rows.append((frac, pixel, sw, sh, dw, dh, filt or 'default', name, best, diff))
            del src, dst
    rows.sort()
    for frac, pixel, sw, sh, dw, dh, filt, name, ms, diff in rows:
        print('%.3f  %-13s %4dx%-4d -> %4dx%-4d %-9s %8.3f ms  diff %d  %s' % (frac, pixel, sw, sh, dw, dh, filt, ms, diff, name))


if __name__ == '__main__':
    main()
