"""Accumulator kernel (windows up to 36 taps) on small batches: planner's row chunks ("auto") against finer fixed chunk counts."""
import sys
sys.path.insert(0, '.')
import torch
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan
torch.cuda.set_device(0)
lib = N.lib()
CASES_STREAM = [('rgba', (3840, 2160, 1920, 1080), None), ('rgb', (1920, 1080, 960, 540), None), ('rgba', (640, 480, 320, 240), None), ('grey', (3840, 2160, 1280, 720), None),
                ('rgba', (3840, 2160, 960, 540), None), ('rgb', (960, 540, 1920, 1080), None), ('rgba', (1920, 1080, 1280, 720), None), ('grey', (3840, 2160, 512, 288), 'lanczos')]
CASES_SPAN = [('rgb', (960, 540, 1440, 810), None), ('rgba', (1920, 1080, 2560, 1440), None), ('grey', (960, 540, 1000, 563), None), ('rgba', (1280, 720, 1920, 1080), 'lanczos'),
              ('r16g16b16a16', (960, 540, 1440, 810), None)]
CASES_ACC = [('rgb', (3840, 2160, 640, 360), None), ('rgba', (3840, 2160, 500, 281), None), ('rgb', (4032, 3024, 800, 600), None),
                                      ('grey', (3840, 2160, 400, 225), 'lanczos'), ('rgba', (1920, 1080, 300, 169), None), ('rgb', (3840, 2160, 1280, 720), 'mitchel')]
for pixel, (sw, sh, dw, dh), filt in (CASES_STREAM if len(sys.argv) > 1 and sys.argv[1] == 'stream' else CASES_SPAN if len(sys.argv) > 1 and sys.argv[1] == 'span' else CASES_ACC):
    for n in ((1, 2, 4, 8, 16) if len(sys.argv) > 1 else (1, 2, 4, 8, 16, 32, 64)):
        src = DeviceBatch(n, sw, sh, pixel).fill_random(1)
        dst = DeviceBatch(n, dw, dh, pixel)
        plan = ResizePlan(src, dst, filter=filt)
        res = []
        name = ''
        for ch in (0, dh // 16, dh // 8, dh // 4, dh // 2, dh):
            N.check(lib.picha_hip_set_option(b"row_chunks", ch))
            plan.run(); torch.cuda.synchronize()
            if ch == 0:
                name = lib.picha_hip_last_kernel().decode()
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
        print('%-5s n=%3d %dx%d->%dx%d %-8s' % (pixel, n, sw, sh, dw, dh, filt or 'default'), ' '.join(('auto %.4f' % r[1]) if r[0] == 0 else 'c%d %.4f' % r for r in res),
              '| auto/best fixed %+.0f%%' % (100 * (res[0][1] / b[1] - 1)), name[:44], flush=True)
        del src, dst
