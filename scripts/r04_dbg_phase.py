import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import picha_amd as picha
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, resize_batch
from util import oracle_resize
def opt(k, v): N.check(N.lib().picha_hip_set_option(k.encode(), v))
opt("run_xch", 1)
for (n, sw, sh, dw, dh) in [(6, 1536, 800, 384, 200), (3, 2048, 1600, 512, 400), (4, 7680, 1080, 1920, 270)]:
    src = DeviceBatch(n, sw, sh, "r16g16b16a16").fill_random(5 + sw)
    exp = [oracle_resize(src.image_host(i), dw, dh, "mitchel", 1.0).pixels() for i in range(n)]
    for ph in (-1, 4, 6, 8, 10, 12, 20):
        for chunks in (0, 3):
            opt("run_phase", ph); opt("row_chunks", chunks)
            for rep in range(3):
                out = resize_batch(src, dw, dh, filter="mitchel")
                torch.cuda.synchronize()
                name = N.lib().picha_hip_last_kernel().decode()
                bad = []
                for i in range(n):
                    got = out.image_host(i).pixels()
                    if not np.array_equal(got, exp[i]):
                        d = np.argwhere((got != exp[i]).reshape(dh, dw, -1).any(axis=2))
                        rows = sorted(set(d[:, 0].tolist())); cols = sorted(set(d[:, 1].tolist()))
                        bad.append((i, len(d), rows[:6], rows[-3:], cols[:4], cols[-2:]))
                print(n, sw, sh, "phase", ph, "chunks", chunks, "rep", rep, name.split("+")[0], "OK" if not bad else "BAD %s" % bad[:2])
