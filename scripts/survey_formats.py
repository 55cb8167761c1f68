"""Device-resident throughput survey over pixel formats and scale factors (which kernel family serves what, and how well).
usage: python scripts/survey_formats.py [n_images]"""
import sys
sys.path.insert(0, '.')
import torch
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
only = sys.argv[2].split(',') if len(sys.argv) > 2 else None      # e.g. "2x down,3x down"
paths = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else [0]   # library option "path": 0 auto, 3 = span kernel
xcds = [int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else [0]   # library option "xcd_remap"
torch.cuda.set_device(0)
cases = [('2x down', 3840, 2160, 1920, 1080, {}), ('3x down', 3840, 2160, 1280, 720, {}), ('4x down', 3840, 2160, 960, 540, {}),
         ('1.5x down', 1920, 1080, 1280, 720, {}), ('7.5x thumb', 3840, 2160, 512, 288, {}), ('2x up', 960, 540, 1920, 1080, {}),
         ('2x down lanczos', 3840, 2160, 1920, 1080, {'filter': 'lanczos'}), ('2x down triangle', 3840, 2160, 1920, 1080, {'filter': 'triangle'}),
         ('4x down mitchel', 3840, 2160, 960, 540, {'filter': 'mitchel'})]
numerics = int(sys.argv[6]) if len(sys.argv) > 6 else 0
N.check(N.lib().picha_hip_set_option(b'numerics', numerics))
print('%-16s %-14s %-58s %8s %8s %6s' % ('case', 'pixel', 'kernel', 'ms', 'GB/s', 'frac'))
for name, sw, sh, dw, dh, o in cases:
    if only and name not in only:
        continue
    for pixel in (sys.argv[5].split(',') if len(sys.argv) > 5 else ['rgba', 'rgb', 'grey', 'greya', 'r16g16b16a16', 'r16']):
        src = DeviceBatch(n, sw, sh, pixel).fill_random(3)
        dst = DeviceBatch(n, dw, dh, pixel)
        plan = ResizePlan(src, dst, filter=o.get('filter'), filter_scale=o.get('filterScale'))
        for path, xcd in [(a, b) for a in paths for b in xcds]:
            N.check(N.lib().picha_hip_set_option(b'path', path))
            N.check(N.lib().picha_hip_set_option(b'xcd_remap', xcd))
            try:
                for _ in range(3): plan.run()
            except N.PichaError as e:
                print('%-16s %-14s path %d: %s' % (name, pixel, path, e), flush=True)
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): plan.run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            b = n * (sw * sh + dw * dh) * N.PIXEL_BYTES[pixel]
            print('%-16s %-14s %-58s %8.3f %8.1f %6.3f' % (name, pixel, N.lib().picha_hip_last_kernel().decode() + (' xcd%d' % xcd if len(xcds) > 1 else ''), ms, b / ms / 1e6, b / ms / 1e6 / 8000), flush=True)
        N.check(N.lib().picha_hip_set_option(b'path', 0)); N.check(N.lib().picha_hip_set_option(b'xcd_remap', 0))
        del src, dst, plan
