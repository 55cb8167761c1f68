import sys
sys.path.insert(0, '.')
import numpy as np, torch
import oracle, picha_amd
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan
torch.cuda.set_device(0)
lib = N.lib()
for pixel, (sw, sh, dw, dh), filt, n in [('rgb', (6000, 4000, 200, 133), 'lanczos', 32), ('rgb', (6000, 4000, 150, 100), None, 32), ('rgba', (8192, 5464, 256, 171), 'lanczos', 12),
                                         ('grey', (6000, 4000, 160, 107), 'mitchel', 64), ('rgb', (6000, 4000, 200, 133), 'lanczos', 1), ('rgb', (6000, 4000, 120, 80), None, 1)]:
    src = DeviceBatch(n, sw, sh, pixel).fill_random(1)
    dst = DeviceBatch(n, dw, dh, pixel)
    host0 = src.image_host(0).data
    e, _ = oracle.resize(host0, sw, sh, pixel, dw, dh, filt or 'cubic', 1.0 if filt else 0.70)
    exp = picha_amd.Image(width=dw, height=dh, pixel=pixel, data=e).pixels().astype(np.int64)
    plan = ResizePlan(src, dst, filter=filt)
    out = []
    for path in (0, 1):
        N.check(lib.picha_hip_set_option(b"path", path))
        try:
            plan.run(); torch.cuda.synchronize()
        except Exception as ex:
            out.append('path=%d error %s' % (path, ex)); continue
        finally:
            pass
        name = lib.picha_hip_last_kernel().decode()
        diff = int(np.abs(dst.image_host(0).pixels().astype(np.int64) - exp).max())
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): plan.run()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 3)
        out.append('%s %.3f ms d%d' % (name[:44], best, diff))
    N.check(lib.picha_hip_set_option(b"path", 0))
    print('%-5s n=%2d %dx%d->%dx%d %-8s' % (pixel, n, sw, sh, dw, dh, filt or 'default'), ' | '.join(out), flush=True)
    del src, dst
