"""Interleaved A/B of kernel options in one process (box-to-box and warm-up effects cancel)."""
import sys, json, itertools
sys.path.insert(0, '.')
import torch
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan
def opt(k, v): N.check(N.lib().picha_hip_set_option(k.encode(), v))
torch.cuda.set_device(0)
shape = sys.argv[3].split(':') if len(sys.argv) > 3 else ['256', 'rgba', '3840', '2160', '1920', '1080']   # n:pixel:sw:sh:dw:dh
src = DeviceBatch(int(shape[0]), int(shape[2]), int(shape[3]), shape[1]).fill_random(1)
dst = DeviceBatch(int(shape[0]), int(shape[4]), int(shape[5]), shape[1])
plan_opts = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {}
plan = ResizePlan(src, dst, filter=plan_opts.get('filter'), filter_scale=plan_opts.get('filterScale'))
variants = [dict(numerics=n, per_lane=p, path=pa) for n in (0, 1) for (p, pa) in ((2, 2), (1, 2), (4, 2), (2, 3))]
if len(sys.argv) > 1:
    variants = json.loads(sys.argv[1])
res = {}
for rnd in range(4):
    for v in variants:
        for k in ('numerics', 'per_lane', 'path', 'row_chunks', 'stream_rot', 'dpp', 'xcd_remap'): opt(k, 0)  # no state carried between variants
        for k, x in v.items(): opt(k, x)
        plan.run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): plan.run()
        e1.record(); torch.cuda.synchronize()
        name = N.lib().picha_hip_last_kernel().decode() + ' ' + json.dumps({k: x for k, x in v.items() if k not in ('numerics', 'path')})
        res.setdefault(name, []).append(e0.elapsed_time(e1) / 10)
for k, v in res.items():
    print('%-62s %s  min %.3f' % (k, ' '.join('%.3f' % x for x in v), min(v)))
