#!/bin/bash
o=gpurun_out/r03/call8; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_chain_and_helpers.py -x -q > $o/pytest_chain.txt 2>&1 || { tail -40 $o/pytest_chain.txt; exit 1; }
tail -2 $o/pytest_chain.txt
# fused vs two-kernel chain for the new pairs, 128 x 4K -> 512x288 (7.5x, lanczos)
python - > $o/ab_fused_pairs.txt 2>&1 <<'PY'
import sys; sys.path.insert(0,'.')
import torch, numpy as np
import picha_amd
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, convert_resize_batch
def opt(k,v): N.check(N.lib().picha_hip_set_option(k.encode(), int(v)))
for sp, dp, n in (("rgb","grey",128),("rgb","greya",128),("rgba","greya",128),("r16g16b16","r16",64),("r16g16b16a16","r16",64),("r16g16b16","grey",64),("r16g16b16a16","grey",64)):
    src = DeviceBatch(n, 3840, 2160, sp).fill_random(3)
    dst = DeviceBatch(n, 512, 512, dp)
    res = {}
    for rnd in range(3):
        for fuse in (0, 2):
            opt("fuse", fuse)
            convert_resize_batch(src, dp, 512, 512, filter="lanczos", out=dst); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): convert_resize_batch(src, dp, 512, 512, filter="lanczos", out=dst)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(fuse, []).append((e0.elapsed_time(e1)/10, N.lib().picha_hip_last_kernel().decode()))
    opt("fuse", 0)
    bpp = N.PIXEL_BYTES[sp]; alg = n*(3840*2160*bpp + 512*512*N.PIXEL_BYTES[dp])
    for fuse in (0, 2):
        t = min(x[0] for x in res[fuse])
        print("%-14s -> %-6s n=%3d fuse=%d  %.3f ms  frac(fused bytes) %.3f  %s" % (sp, dp, n, fuse, t, alg/(t*1e-3)/8e12, res[fuse][0][1]))
    del src, dst
PY
cat $o/ab_fused_pairs.txt
