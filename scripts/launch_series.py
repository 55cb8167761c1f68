import sys, json
sys.path.insert(0, '.')
import torch
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan
torch.cuda.set_device(0)
src = DeviceBatch(256, 3840, 2160, 'rgba').fill_random(1)
dst = DeviceBatch(256, 1920, 1080, 'rgba')
plan = ResizePlan(src, dst)
torch.cuda.synchronize()
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 60
evs = [torch.cuda.Event(enable_timing=True) for _ in range(NL + 1)]
evs[0].record()
for i in range(NL):
    plan.run(); evs[i + 1].record()
torch.cuda.synchronize()
print(' '.join('%.3f' % evs[i].elapsed_time(evs[i + 1]) for i in range(NL)))
