"""Integer-scale downscales: resize_sym_kernel (per outputs-per-lane) against the kernels it replaces, all 8 formats.

usage: python scripts/survey_sym.py [n_images [numerics [cases]]]
Per (case, format): the old planner's choice (sym = 2) and the sym kernel for every outputs-per-lane it is instantiated
for, interleaved; the sym outputs are compared byte for byte with the old kernel's on the device (which the parity tests
pin on the oracle)."""
import sys

sys.path.insert(0, '.')
import torch

from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan


def opt(k, v):
    N.check(N.lib().picha_hip_set_option(k.encode(), int(v)))


n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
numerics = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only = sys.argv[3].split(',') if len(sys.argv) > 3 else None
torch.cuda.set_device(0)
cases = [('2x default', 3840, 2160, 1920, 1080, {}), ('2x cubic7', 3840, 2160, 1920, 1080, {'filter': 'cubic'}),
         ('2x triangle', 3840, 2160, 1920, 1080, {'filter': 'triangle'}), ('3x default', 3840, 2160, 1280, 720, {}),
         ('3x mitchel', 3840, 2160, 1280, 720, {'filter': 'mitchel'}), ('4x default', 3840, 2160, 960, 540, {}),
         ('4x mitchel', 3840, 2160, 960, 540, {'filter': 'mitchel'})]
print('%-12s %-13s %-56s %8s %6s  %s' % ('case', 'pixel', 'kernel', 'ms', 'frac', 'vs old / same bytes'))
for name, sw, sh, dw, dh, o in cases:
    if only and name not in only:
        continue
    for pixel in ['rgba', 'rgb', 'grey', 'greya', 'r16', 'r16g16', 'r16g16b16', 'r16g16b16a16']:
        nn = n if N.PIXEL_BYTES[pixel] <= 4 else max(8, n // 2)
        src = DeviceBatch(nn, sw, sh, pixel).fill_random(3)
        dst = DeviceBatch(nn, dw, dh, pixel)
        plan = ResizePlan(src, dst, filter=o.get('filter'), filter_scale=o.get('filterScale'))
        variants = [{'sym': 2}] + [{'per_lane': p} for p in (1, 2, 3, 4, 8)]
        times, names, same = {}, {}, {}
        ref = None
        for rnd in range(3):
            for vi, v in enumerate(variants):
                for k in ('sym', 'per_lane'):
                    opt(k, 0)
                opt('numerics', numerics)
                for k, x in v.items():
                    opt(k, x)
                try:
                    plan.run()
                    torch.cuda.synchronize()
                except N.PichaError:
                    continue
                kn = N.lib().picha_hip_last_kernel().decode()
                if vi > 0 and not kn.startswith('resize_sym'):
                    continue  # no sym instantiation for this outputs-per-lane
                if rnd == 0:
                    if vi == 0:
                        ref = dst.tensor.clone()
                    else:
                        same[vi] = bool(torch.equal(dst.tensor, ref)) if numerics == 0 else int((dst.tensor.int() - ref.int()).abs().max())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8):
                    plan.run()
                e1.record()
                torch.cuda.synchronize()
                times.setdefault(vi, []).append(e0.elapsed_time(e1) / 8)
                names[vi] = kn
        b = nn * (sw * sh + dw * dh) * N.PIXEL_BYTES[pixel]
        t0 = min(times[0]) if 0 in times else None
        for vi in sorted(times):
            t = min(times[vi])
            print('%-12s %-13s %-56s %8.3f %6.3f  %s' % (name, pixel, names[vi], t, b / t / 1e6 / 8000,
                                                        '' if vi == 0 else '%+.1f%%  %s' % ((t / t0 - 1) * 100, same.get(vi))), flush=True)
        del src, dst, plan, ref
for k in ('sym', 'per_lane', 'numerics'):
    opt(k, 0)
