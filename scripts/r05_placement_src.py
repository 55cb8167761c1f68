"""Does the placement of the SOURCE allocation matter too?  K source batches allocated one after the other (same pixels), destinations allocated before / between / after them;
the same launch timed for every (source, destination) pair, interleaved.  usage: python scripts/r05_placement_src.py [cfg2|cfg5] [K]"""
import sys
sys.path.insert(0, '.')
import torch
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan
CASES = {"cfg2": (256, "rgba", 3840, 2160, 1920, 1080, None), "cfg5": (32, "r16g16b16a16", 7680, 4320, 1920, 1080, "mitchel")}
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n, pixel, sw, sh, dw, dh, filt = CASES[name]
torch.cuda.set_device(0)
order = []
d_first = DeviceBatch(n, dw, dh, pixel); order.append(("dst", d_first))          # a destination allocated BEFORE any source
srcs = []
dsts = [d_first]
for k in range(K):
    s = DeviceBatch(n, sw, sh, pixel).fill_random(1); srcs.append(s); order.append(("src", s))
    d = DeviceBatch(n, dw, dh, pixel); dsts.append(d); order.append(("dst", d))  # ... and one right after every source
d_last = DeviceBatch(n, dw, dh, pixel); dsts.append(d_last); order.append(("dst", d_last))
print("%s allocation order: %s" % (name, "  ".join("%s@%#x" % (k, b.tensor.data_ptr()) for k, b in order)))
plans = {(i, j): ResizePlan(s, d, filter=filt) for i, s in enumerate(srcs) for j, d in enumerate(dsts)}
for p in plans.values():
    p.run()
torch.cuda.synchronize()
best = {k: 1e9 for k in plans}
for rnd in range(3):
    for k, p in plans.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p.run()
        e0.record()
        for _ in range(8):
            p.run()
        e1.record()
        torch.cuda.synchronize()
        best[k] = min(best[k], e0.elapsed_time(e1) / 8)
print("kernel: %s" % N.lib().picha_hip_last_kernel().decode())
print("ms per launch (min of 3 rounds), rows = sources in allocation order, columns = destinations in allocation order (dst0 before every source, dst k+1 right after source k, the last after all)")
for i in range(len(srcs)):
    print("  source %d: %s" % (i, "  ".join("%.3f" % best[(i, j)] for j in range(len(dsts)))))
