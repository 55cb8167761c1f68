"""Does the placement of the DESTINATION allocation matter for the real kernels?  One source batch, K destination batches allocated one
after the other (all kept alive, so each sits somewhere else in HBM), the same launch timed into each, interleaved over rounds.
usage: python scripts/r04_placement.py [cfg2|cfg5|cfg3b] [K] [spacer_mb]"""
import sys
sys.path.insert(0, '.')
import torch
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan
CASES = {"cfg2": (256, "rgba", 3840, 2160, 1920, 1080, None), "cfg5": (32, "r16g16b16a16", 7680, 4320, 1920, 1080, "mitchel"),
         "cfg3b": (256, "grey", 3840, 2160, 512, 512, "lanczos")}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
spacer_mb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n, pixel, sw, sh, dw, dh, filt = CASES[name]
torch.cuda.set_device(0)
src = DeviceBatch(n, sw, sh, pixel).fill_random(1)
dsts, spacers = [], []
for k in range(K):
    if spacer_mb:
        spacers.append(torch.empty(spacer_mb << 20, dtype=torch.uint8, device="cuda"))
    dsts.append(DeviceBatch(n, dw, dh, pixel))
plans = [ResizePlan(src, d, filter=filt) for d in dsts]
for p in plans:
    p.run()
torch.cuda.synchronize()
res = [[] for _ in plans]
for rnd in range(4):
    for i, p in enumerate(plans):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p.run()
        e0.record()
        for _ in range(10):
            p.run()
        e1.record()
        torch.cuda.synchronize()
        res[i].append(e0.elapsed_time(e1) / 10)
print("%s: %s, spacer %d MB, source at %#x" % (name, N.lib().picha_hip_last_kernel().decode(), spacer_mb, src.tensor.data_ptr()))
for i, r in enumerate(res):
    print("  destination %d at %#x (+%.3f GB from the source): %s  min %.3f ms" % (i, dsts[i].tensor.data_ptr(), (dsts[i].tensor.data_ptr() - src.tensor.data_ptr()) / 2**30,
                                                                                  " ".join("%.3f" % x for x in r), min(r)))
best, worst = min(min(r) for r in res), max(min(r) for r in res)
print("  spread of the minima: %.3f .. %.3f ms (%.1f %%)" % (best, worst, (worst / best - 1) * 100))
