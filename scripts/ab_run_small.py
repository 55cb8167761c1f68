"""Run / ring kernels against the accumulator kernel on small batches (images resident), default chunking of each."""
import sys
sys.path.insert(0, '.')
import torch
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan
torch.cuda.set_device(0)
lib = N.lib()
CASES = [('grey', (3840, 2160, 512, 288), 'lanczos'), ('greya', (3840, 2160, 512, 288), 'lanczos'), ('grey', (3840, 2160, 512, 288), None),
         ('grey', (1920, 1080, 1280, 720), None), ('rgba', (3840, 2160, 1920, 1080), 'lanczos'), ('rgb', (960, 540, 1920, 1080), None), ('grey', (1920, 1080, 480, 270), None)]
for pixel, (sw, sh, dw, dh), filt in CASES:
    for n in (1, 2, 4, 8, 16, 32):
        src = DeviceBatch(n, sw, sh, pixel).fill_random(1)
        dst = DeviceBatch(n, dw, dh, pixel)
        plan = ResizePlan(src, dst, filter=filt)
        res = []
        for run in (0, 2):
            N.check(lib.picha_hip_set_option(b"run", run))
            plan.run(); torch.cuda.synchronize()
            name = lib.picha_hip_last_kernel().decode()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): plan.run()
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
            res.append((name.split(',')[0], best))
        N.check(lib.picha_hip_set_option(b"run", 0))
        print('%-5s n=%3d %dx%d->%dx%d %-8s' % (pixel, n, sw, sh, dw, dh, filt or 'default'), 'default %s %.4f | run=2 %s %.4f | default/other %+.0f%%' % (res[0][0], res[0][1], res[1][0], res[1][1], 100 * (res[0][1] / res[1][1] - 1)), flush=True)
        del src, dst
