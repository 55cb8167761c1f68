"""Accumulator kernel: two outputs per lane (acc_p2 = 1) against one (acc_p2 = 2), interleaved, outputs compared byte for byte.
usage: python scripts/survey_acc_p2.py [n_images [numerics]]"""
import sys
sys.path.insert(0, '.')
import torch
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan

def opt(k, v): N.check(N.lib().picha_hip_set_option(k.encode(), int(v)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
numerics = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.cuda.set_device(0)
cases = [('1.5x default', 1920, 1080, 1280, 720, {}), ('2.5x default', 3840, 2160, 1536, 864, {}), ('3x mitchel', 3840, 2160, 1280, 720, {'filter': 'mitchel'}),
         ('4x default', 3840, 2160, 960, 540, {}), ('4x mitchel', 3840, 2160, 960, 540, {'filter': 'mitchel'}), ('5x default', 3840, 2160, 768, 432, {}),
         ('6x default', 3840, 2160, 640, 360, {}), ('7.5x default', 3840, 2160, 512, 288, {}), ('7.5x lanczos', 3840, 2160, 512, 288, {'filter': 'lanczos'})]
print('%-13s %-13s %-50s %8s %6s  %s' % ('case', 'pixel', 'kernel', 'ms', 'frac', 'vs one output per lane / same bytes'))
def run_case(name, sw, sh, dw, dh, o, pixel, nn):
    src = DeviceBatch(nn, sw, sh, pixel).fill_random(3)
    dst = DeviceBatch(nn, dw, dh, pixel)
    plan = ResizePlan(src, dst, filter=o.get('filter'), filter_scale=o.get('filterScale'))
    times, names, same, ref = {}, {}, {}, None
    for rnd in range(3):
        for vi, v in enumerate(({'acc_p2': 2, 'sym': 2, 'rows': 2, 'path': 4}, {'acc_p2': 1, 'sym': 2, 'rows': 2, 'path': 4})):
            for k in ('acc_p2', 'sym', 'rows', 'path'): opt(k, 0)
            opt('numerics', numerics)
            for k, x in v.items(): opt(k, x)
            try:
                plan.run(); torch.cuda.synchronize()
            except N.PichaError as e:
                names[vi] = 'ERR %s' % e; continue
            kn = N.lib().picha_hip_last_kernel().decode()
            if rnd == 0:
                if vi == 0: ref = dst.tensor.clone()
                else: same[vi] = bool(torch.equal(dst.tensor, ref)) if numerics == 0 else int((dst.tensor.int() - ref.int()).abs().max())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): plan.run()
            e1.record(); torch.cuda.synchronize()
            times.setdefault(vi, []).append(e0.elapsed_time(e1) / 8); names[vi] = kn
    b = nn * (sw * sh + dw * dh) * N.PIXEL_BYTES[pixel]
    for vi in sorted(times):
        t = min(times[vi])
        print('%-13s %-13s %-50s %8.3f %6.3f  %s' % (name, pixel, names[vi], t, b / t / 1e6 / 8000,
              '' if vi == 0 or ',P2,' not in names[vi] else '%+.1f%%  %s' % ((t / min(times[0]) - 1) * 100, same.get(vi))), flush=True)
for name, sw, sh, dw, dh, o in cases:
    for pixel in ['rgba', 'rgb', 'grey', 'greya', 'r16', 'r16g16', 'r16g16b16', 'r16g16b16a16']:
        run_case(name, sw, sh, dw, dh, o, pixel, n if N.PIXEL_BYTES[pixel] <= 4 else max(8, n // 2))
run_case('cfg5', 7680, 4320, 1920, 1080, {'filter': 'mitchel'}, 'r16g16b16a16', 16)
for k in ('acc_p2', 'sym', 'rows', 'path', 'numerics'): opt(k, 0)
