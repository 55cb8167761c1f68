"""Scans formats x filters x a dense ladder of scale factors (small images, default options) for geometries that fall to the
two-pass kernels although their windows are short enough for a one-pass kernel, and checks every result against the oracle.

usage: python scripts/scan_fallbacks.py [pixels,comma,separated]
"""
import collections
import sys

sys.path.insert(0, '.')
import numpy as np

import oracle
import picha_amd
from picha_amd import _native as N


def main():
    pixels = sys.argv[1].split(',') if len(sys.argv) > 1 else list(N.PIXEL_BYTES)
    rng = np.random.default_rng(7)
    filters = [None, 'cubic', 'lanczos', 'catmulrom', 'mitchel', 'box', 'triangle']
    scales = sorted(set([round(1.0 + 0.37 * i, 2) for i in range(64)] + [float(i) for i in range(1, 25)] + [1.5, 2.5, 7.5, 12.5, 12.6]))
    fam = collections.Counter()
    bad = 0
    for pixel in pixels:
        bpp = N.PIXEL_BYTES[pixel]
        for ladder in ('x', 'y', 'xy', 'up'):
            for sc in scales:
                if ladder == 'x':
                    dw, dh = 96, 12; sw, sh = int(round(dw * sc)), 24
                elif ladder == 'y':
                    dw, dh = 80, 9; sw, sh = 160, int(round(dh * sc))
                elif ladder == 'xy':
                    dw, dh = 72, 10; sw, sh = int(round(dw * sc)), int(round(dh * sc))
                else:
                    if sc > 8.0:
                        continue
                    sw, sh = 48, 10; dw, dh = int(round(sw * sc)), int(round(sh * sc))
                stride = (sw * bpp + 3) & ~3
                data = rng.integers(0, 256, size=sh * stride, dtype=np.uint8)
                img = picha_amd.Image(width=sw, height=sh, pixel=pixel, data=data, stride=stride)
                for filt in filters:
                    opts = {'width': dw, 'height': dh}
                    if filt:
                        opts['filter'] = filt
                    try:
                        got = picha_amd.resizeSync(img, opts)
                    except Exception as ex:
                        print('ERROR', pixel, ladder, sc, filt, ex, flush=True)
                        bad += 1
                        continue
                    name = N.lib().picha_hip_last_kernel().decode()
                    e, _ = oracle.resize(data, sw, sh, pixel, dw, dh, filt or 'cubic', 1.0 if filt else 0.70, stride=stride)
                    exp = picha_amd.Image(width=dw, height=dh, pixel=pixel, data=e).pixels().astype(np.int64)
                    diff = int(np.abs(got.pixels().astype(np.int64) - exp).max())
                    fam[name.split('<')[0]] += 1
                    if diff or name == 'twopass':
                        print('%-13s %-2s scale %5.2f  %4dx%-4d -> %4dx%-4d %-9s diff %d  %s' % (pixel, ladder, sc, sw, sh, dw, dh, filt or 'default', diff, name), flush=True)
                        bad += diff != 0
    print('kernel families:', dict(fam))
    print('mismatches / errors:', bad)


main()
