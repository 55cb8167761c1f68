"""Full-size check of the phased stores through a REUSED destination batch (zeroed before every run): every row of images 0 and n-1 against the oracle.
How the lost-last-row bug of the first phased build was found (two rows completing with the last source row while the buffer held 7)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import picha_amd as picha
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan
from util import oracle_resize
def opt(k, v): N.check(N.lib().picha_hip_set_option(k.encode(), v))
n, sw, sh, dw, dh = 32, 7680, 4320, 1920, 1080
src = DeviceBatch(n, sw, sh, "r16g16b16a16").fill_random(1)
dst = DeviceBatch(n, dw, dh, "r16g16b16a16")
plan = ResizePlan(src, dst, filter="mitchel")
idx = (0, n - 1)
exp = {i: oracle_resize(src.image_host(i), dw, dh, "mitchel", 1.0).pixels() for i in idx}
for ph in (-1, 12, 12, 14, 10, 12, 13):
    opt("run_phase", ph)
    for mode in ("warm+zero", "zero", "launches"):
        if mode == "warm+zero":
            plan.run(); torch.cuda.synchronize(); dst.tensor.zero_(); plan.run(); torch.cuda.synchronize()
        elif mode == "zero":
            dst.tensor.zero_(); plan.run(); torch.cuda.synchronize()
        else:
            for _ in range(10): plan.run()
            torch.cuda.synchronize()
        name = N.lib().picha_hip_last_kernel().decode()
        bad = []
        for i in idx:
            got = dst.image_host(i).pixels()
            if not np.array_equal(got, exp[i]):
                m = (got != exp[i]).reshape(dh, dw, -1).any(axis=2)
                d = np.argwhere(m)
                rows = sorted(set(d[:, 0].tolist())); cols = sorted(set(d[:, 1].tolist()))
                z = int((got.reshape(dh, dw, -1)[m] == 0).all(axis=1).sum())
                bad.append((i, len(d), "zero px", z, "rows", rows[:8], "..", rows[-3:], "cols", cols[:4], "..", cols[-3:], "ncols", len(cols), "nrows", len(rows)))
        print("phase", ph, mode, name.split("+")[0], "OK" if not bad else "BAD %s" % bad)
