import sys
sys.path.insert(0, '.')
import numpy as np, torch
import picha_amd
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan
torch.cuda.set_device(0)
lib = N.lib()
for pixel in ('grey', 'rgb', 'rgba'):
    bpp = N.PIXEL_BYTES[pixel]
    for (sw, sh, dw, dh) in [(3840, 2160, 320, 180), (3840, 2160, 240, 135), (4032, 3024, 320, 240), (6000, 4000, 300, 200), (3840, 2160, 384, 216)]:
        for n in ((128, 16) if len(sys.argv) < 2 else (int(sys.argv[1]),)):
            m = max(4, min(n, int(3.0e9 // (sw * sh * bpp))))
            src = DeviceBatch(m, sw, sh, pixel).fill_random(1)
            dst = DeviceBatch(m, dw, dh, pixel)
            for filt in (None, 'lanczos'):
                plan = ResizePlan(src, dst, filter=filt)
                res = []
                for ch in (0, 1, 2, 3, 4, 6, 8, 11, 16):
                    N.check(lib.picha_hip_set_option(b"row_chunks", ch))
                    plan.run(); torch.cuda.synchronize()
                    best = 1e9
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(5): plan.run()
                        e1.record(); torch.cuda.synchronize()
                        best = min(best, e0.elapsed_time(e1) / 5)
                    res.append((ch, best))
                N.check(lib.picha_hip_set_option(b"row_chunks", 0))
                name = lib.picha_hip_last_kernel().decode()
                print('%-5s n=%3d %dx%d->%dx%d %-8s' % (pixel, m, sw, sh, dw, dh, filt or 'default'), ' '.join('c%d %.3f' % r for r in res), name, flush=True)
            del src, dst
