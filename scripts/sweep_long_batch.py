"""Long-window accumulator kernel over batch sizes (thumbnails): planner's chunking ("auto") against fixed chunk counts."""
import sys
sys.path.insert(0, '.')
import torch
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan
torch.cuda.set_device(0)
lib = N.lib()
for pixel, (sw, sh, dw, dh), filt in [('rgba', (6000, 4000, 300, 200), 'lanczos'), ('rgb', (4032, 3024, 320, 240), None), ('grey', (3840, 2160, 320, 180), 'lanczos'),
                                      ('rgb', (3840, 2160, 240, 135), 'lanczos'), ('rgba', (3840, 2160, 384, 216), 'mitchel')]:
    for n in (4, 8, 16, 24, 31, 40, 48, 64, 96):
        if n * sw * sh * N.PIXEL_BYTES[pixel] > 3.2e9:
            continue
        src = DeviceBatch(n, sw, sh, pixel).fill_random(1)
        dst = DeviceBatch(n, dw, dh, pixel)
        plan = ResizePlan(src, dst, filter=filt)
        res = []
        for ch in (0, 4, 6, 8, 11, 16, 22):
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
        b = min(res[1:], key=lambda r: r[1])
        print('%-5s n=%3d %dx%d->%dx%d %-8s' % (pixel, n, sw, sh, dw, dh, filt or 'default'), ' '.join(('auto %.3f' % r[1]) if r[0] == 0 else 'c%d %.3f' % r for r in res),
              '| auto/best fixed %+.0f%%' % (100 * (res[0][1] / b[1] - 1)), flush=True)
        del src, dst
