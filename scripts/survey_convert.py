"""Colour convert: 4-pixel kernel (cc_mode 1) vs wide kernel (0) vs wide + non-temporal stores (2), interleaved;
outputs compared byte for byte on the device.  usage: python scripts/survey_convert.py [n_images]"""
import sys
sys.path.insert(0, '.')
import torch
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, color_convert_batch

def opt(k, v): N.check(N.lib().picha_hip_set_option(k.encode(), int(v)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.cuda.set_device(0)
pairs = [('grey', 'greya'), ('grey', 'r16'), ('greya', 'r16g16'), ('rgb', 'grey'), ('rgba', 'grey'), ('rgba', 'rgb'), ('rgb', 'rgba'), ('grey', 'rgba'), ('rgba', 'greya'), ('rgba', 'r16g16b16a16'),
         ('r16g16b16a16', 'rgba'), ('rgb', 'r16g16b16'), ('r16g16b16', 'grey'), ('greya', 'rgba'), ('r16', 'grey')]
print('%-14s %-14s %-5s %8s %8s %6s' % ('src', 'dst', 'mode', 'ms', 'GB/s', 'frac'))
for sp, dp in pairs:
    for (w, h) in ((3840, 2160), (1918, 1080)):   # second geometry: rows with padding (no flattening)
        nn = n if max(N.PIXEL_BYTES[sp], N.PIXEL_BYTES[dp]) <= 4 else n // 2
        src = DeviceBatch(nn, w, h, sp).fill_random(3)
        dst = DeviceBatch(nn, w, h, dp)
        ref = None
        res = {}
        for rnd in range(3):
            for mode in (1, 0, 2, 3):
                opt('cc_mode', mode)
                color_convert_batch(src, dp, out=dst); torch.cuda.synchronize()
                if rnd == 0:
                    if mode == 1: ref = dst.tensor.clone()
                    else: assert torch.equal(dst.tensor, ref), (sp, dp, mode)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(8): color_convert_batch(src, dp, out=dst)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(mode, []).append(e0.elapsed_time(e1) / 8)
        b = nn * w * h * (N.PIXEL_BYTES[sp] + N.PIXEL_BYTES[dp])
        for mode in (1, 0, 2, 3):
            t = min(res[mode])
            print('%-14s %-14s %-5d %8.3f %8.1f %6.3f   %dx%d' % (sp, dp, mode, t, b / t / 1e6, b / t / 1e6 / 8000, w, h), flush=True)
        del src, dst, ref
opt('cc_mode', 0)
