"""Small launches (1-32 images resident): default plan, ms per launch, for the kernels that serve single-image calls; and the chained
colorConvert -> resize call as one fused kernel against two kernels ("fuse" = 2)."""
import sys
sys.path.insert(0, '.')
import torch
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan, convert_resize_batch
torch.cuda.set_device(0)
lib = N.lib()


def timed(fn):
    fn(); torch.cuda.synchronize()
    name = lib.picha_hip_last_kernel().decode()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    return best, name


CASES = [('grey', (3840, 2160, 512, 288), 'lanczos'), ('rgb', (3840, 2160, 640, 360), None), ('rgba', (3840, 2160, 1920, 1080), None), ('rgba', (3840, 2160, 1920, 1080), 'lanczos'),
         ('rgb', (960, 540, 1920, 1080), None), ('grey', (3840, 2160, 320, 180), 'lanczos'), ('rgba', (640, 480, 320, 240), None)]
for pixel, (sw, sh, dw, dh), filt in CASES:
    out = []
    for n in (1, 2, 4, 8, 16, 32):
        src = DeviceBatch(n, sw, sh, pixel).fill_random(1)
        dst = DeviceBatch(n, dw, dh, pixel)
        plan = ResizePlan(src, dst, filter=filt)
        t, name = timed(plan.run)
        out.append('n=%d %.4f' % (n, t))
        del src, dst
    print('%-5s %dx%d->%dx%d %-8s' % (pixel, sw, sh, dw, dh, filt or 'default'), '  '.join(out), name[:60], flush=True)
for (spix, dpix) in (('rgb', 'grey'), ('rgba', 'grey'), ('rgb', 'greya')):
    for n in (1, 2, 4, 8, 16, 32, 64):
        src = DeviceBatch(n, 3840, 2160, spix).fill_random(1)
        res = []
        for fuse in (0, 2):
            N.check(lib.picha_hip_set_option(b"fuse", fuse))
            t, name = timed(lambda: convert_resize_batch(src, dpix, 512, 512, filter='lanczos'))
            res.append((t, name.split('<')[0]))
        N.check(lib.picha_hip_set_option(b"fuse", 0))
        print('chain %s->%s->512x512 lanczos n=%2d  default %.4f (%s)  two kernels %.4f (last: %s)  default/two %+.0f%%' % (spix, dpix, n, res[0][0], res[0][1], res[1][0], res[1][1], 100 * (res[0][0] / res[1][0] - 1)), flush=True)
        del src
