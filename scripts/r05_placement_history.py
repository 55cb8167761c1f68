"""Placement against allocation HISTORY (r05): successive processes on one box find more and more of their destination candidates on the slow level (profiles/r05_placement_history.txt).
usage: python scripts/r05_placement_history.py [--ballast-gb X] [--ballast-after-src-gb Y] [--cands K] [--src-cands S]
Allocates X GB first (kept until the end), then the cfg2 source batch(es), Y GB more, then K destination batches; times cfg2's launch into every (source, destination) pair."""
import argparse, sys
sys.path.insert(0, '.')
import torch
import picha_amd
from picha_amd.device import DeviceBatch, ResizePlan

ap = argparse.ArgumentParser()
ap.add_argument("--ballast-gb", type=float, default=0)
ap.add_argument("--ballast-after-src-gb", type=float, default=0)
ap.add_argument("--cands", type=int, default=4)
ap.add_argument("--src-cands", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
ballast = [torch.empty(int(a.ballast_gb * (1 << 30)), dtype=torch.uint8, device=dev)] if a.ballast_gb > 0 else []
srcs = [DeviceBatch(256, 3840, 2160, "rgba", device=dev).fill_random(1 + i) for i in range(a.src_cands)]
if a.ballast_after_src_gb > 0:
    ballast.append(torch.empty(int(a.ballast_after_src_gb * (1 << 30)), dtype=torch.uint8, device=dev))
dsts = [DeviceBatch(256, 1920, 1080, "rgba", device=dev) for _ in range(a.cands)]
out = []
for si, s in enumerate(srcs):
    plans = [ResizePlan(s, d) for d in dsts]
    for p in plans:
        p.run()
    torch.cuda.synchronize()
    best = [1e9] * len(plans)
    for _ in range(2):
        for i, p in enumerate(plans):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            p.run(); e0.record()
            for _ in range(6):
                p.run()
            e1.record(); torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / 6)
    out.append([round(b, 3) for b in best])
free, total = torch.cuda.mem_get_info()
print("ballast %g+%g GB src %d dst %d: ms per launch per (src, dst) %s | src at %#x dst0 at %#x free %.0f GB" % (
    a.ballast_gb, a.ballast_after_src_gb, a.src_cands, a.cands, out, srcs[0].tensor.data_ptr(), dsts[0].tensor.data_ptr(), free / (1 << 30)))
