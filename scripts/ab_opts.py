"""Interleaved A/B of library options inside ONE process, each variant checked against the oracle.

usage: python scripts/ab_opts.py '<json list of option dicts>' ['<resize opts json>' [n:pixel:sw:sh:dw:dh [rounds [launches]]]]
Every variant starts from all-default options.  Prints per variant: kernel name, per-round ms per launch, min, the
fraction of the 8 TB/s HBM roof for the minimum, and the max code difference of images 0 and n-1 against the oracle.
"""
import json
import sys

sys.path.insert(0, '.')
import numpy as np
import torch

import oracle
import picha_amd
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan

KNOBS = ('numerics', 'per_lane', 'path', 'row_chunks', 'stream_rot', 'dpp', 'xcd_remap', 'acc', 'sym', 'block_waves', 'sym_sync', 'rows', 'xch', 'acc_p2', 'work', 'work_tail', 'work_tail_div', 'up', 'run', 'run_round', 'run_xch', 'run_ws', 'run_phase')
DEFAULTS = {'work_tail': 200, 'work_tail_div': 4}


def opt(k, v):
    N.check(N.lib().picha_hip_set_option(k.encode(), int(v)))


def main():
    variants = json.loads(sys.argv[1])
    plan_opts = json.loads(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] else {}
    shape = sys.argv[3].split(':') if len(sys.argv) > 3 else ['256', 'rgba', '3840', '2160', '1920', '1080']
    rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    launches = int(sys.argv[5]) if len(sys.argv) > 5 else 10
    n, pixel, sw, sh, dw, dh = int(shape[0]), shape[1], int(shape[2]), int(shape[3]), int(shape[4]), int(shape[5])
    torch.cuda.set_device(0)
    src = DeviceBatch(n, sw, sh, pixel).fill_random(1)
    dst = DeviceBatch(n, dw, dh, pixel)
    plan = ResizePlan(src, dst, filter=plan_opts.get('filter'), filter_scale=plan_opts.get('filterScale'))
    filt = plan_opts.get('filter', 'cubic')
    fscale = plan_opts.get('filterScale', 1.0 if 'filter' in plan_opts else 0.70)
    bpp = N.PIXEL_BYTES[pixel]
    alg = n * (sw * sh + dw * dh) * bpp
    exp = {}
    for i in sorted({0, n - 1}):
        e, _ = oracle.resize(src.image_host(i).data, sw, sh, pixel, dw, dh, filt, fscale)
        exp[i] = picha_amd.Image(width=dw, height=dh, pixel=pixel, data=e).pixels().astype(np.int64)
    res, names, diffs = {}, {}, {}
    for rnd in range(rounds):
        for vi, v in enumerate(variants):
            for k in KNOBS:
                opt(k, DEFAULTS.get(k, 0))
            for k, x in v.items():
                opt(k, x)
            try:
                plan.run()
                torch.cuda.synchronize()
            except Exception as ex:  # unsupported combination: report and go on
                names[vi] = 'ERROR %s' % ex
                continue
            if rnd == 0:
                dst.tensor.zero_()
                plan.run()
                torch.cuda.synchronize()
                diffs[vi] = max(int(np.abs(dst.image_host(i).pixels().astype(np.int64) - exp[i]).max()) for i in exp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(launches):
                plan.run()
            e1.record()
            torch.cuda.synchronize()
            names[vi] = N.lib().picha_hip_last_kernel().decode()
            res.setdefault(vi, []).append(e0.elapsed_time(e1) / launches)
    for vi, v in enumerate(variants):
        if vi not in res:
            print('%-50s %s' % (json.dumps(v), names.get(vi)))
            continue
        t = res[vi]
        print('%-64s %-46s %s  min %.3f  frac %.3f  diff %s' % (names[vi], json.dumps(v), ' '.join('%.3f' % x for x in t), min(t),
                                                                 alg / (min(t) * 1e-3) / 8e12, diffs.get(vi)))
    for k in KNOBS:
        opt(k, DEFAULTS.get(k, 0))


if __name__ == '__main__':
    main()
