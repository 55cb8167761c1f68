"""Do the slow destination placements miss the translation caches more (r05)?  Allocates K destination batches, times cfg2's launch into each, then launches
5 x into the slowest, 5 x into the fastest, 5 x slowest, 5 x fastest — the LAST 20 resize_sym dispatches of the process, in that order.  Run under
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY ...; scripts/r05_placement_tlb_parse.py reads the counter csv."""
import sys
sys.path.insert(0, '.')
import torch
import picha_amd
from picha_amd.device import DeviceBatch, ResizePlan
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
src = DeviceBatch(256, 3840, 2160, "rgba", device=dev).fill_random(1)
dsts = [DeviceBatch(256, 1920, 1080, "rgba", device=dev) for _ in range(K)]
plans = [ResizePlan(src, d) for d in dsts]
for p in plans:
    p.run()
torch.cuda.synchronize()
best = [1e9] * K
for _ in range(2):
    for i, p in enumerate(plans):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        p.run(); e0.record()
        for _ in range(4):
            p.run()
        e1.record(); torch.cuda.synchronize()
        best[i] = min(best[i], e0.elapsed_time(e1) / 4)
slow, fast = max(range(K), key=lambda i: best[i]), min(range(K), key=lambda i: best[i])
print("candidates ms:", [round(b, 3) for b in best], "slow", slow, "fast", fast, flush=True)
for which in (slow, fast, slow, fast):
    for _ in range(5):
        plans[which].run()
    torch.cuda.synchronize()
print("done: last 20 resize_sym dispatches = 5 slow, 5 fast, 5 slow, 5 fast", flush=True)
