#!/bin/bash
o=gpurun_out/r03/call9; mkdir -p $o
timeout -k 10 900 python -m pytest tests/test_gpu_chain_and_helpers.py -x -q > $o/pytest_chain.txt 2>&1 || { tail -40 $o/pytest_chain.txt; exit 1; }
tail -2 $o/pytest_chain.txt
python - > $o/ab_epilogue.txt 2>&1 <<'PY'
import sys; sys.path.insert(0,'.')
import torch, numpy as np
import picha_amd
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, convert_resize_batch
def opt(k,v): N.check(N.lib().picha_hip_set_option(k.encode(), int(v)))
print("# resize -> colorConvert, 4K source: one kernel (accumulator kernel + epilogue, fuse 0) vs two kernels through an HBM intermediate (fuse 2); min of 3 rounds x 10 launches; bytes = source read + final image written")
for sp, dp, n, dw, dh, filt in (("rgba","rgb",128,1920,1080,None),("rgba","grey",128,1920,1080,None),("rgba","rgb",128,1280,720,None),("rgb","grey",128,1920,1080,None),("r16g16b16a16","rgba",32,1920,1080,None),("rgba","rgb",128,960,540,None)):
    src = DeviceBatch(n, 3840, 2160, sp).fill_random(3)
    dst = DeviceBatch(n, dw, dh, dp)
    res = {}
    for rnd in range(3):
        for fuse in (0, 2):
            opt("fuse", fuse)
            convert_resize_batch(src, dp, dw, dh, filter=filt, out=dst, order=1); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): convert_resize_batch(src, dp, dw, dh, filter=filt, out=dst, order=1)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(fuse, []).append((e0.elapsed_time(e1)/10, N.lib().picha_hip_last_kernel().decode()))
    opt("fuse", 0)
    alg = n*(3840*2160*N.PIXEL_BYTES[sp] + dw*dh*N.PIXEL_BYTES[dp])
    for fuse in (0, 2):
        t = min(x[0] for x in res[fuse])
        print("%-14s -> %dx%d %-6s n=%3d fuse=%d  %.3f ms  frac %.3f  %s" % (sp, dw, dh, dp, n, fuse, t, alg/(t*1e-3)/8e12, res[fuse][0][1]))
    del src, dst
PY
cat $o/ab_epilogue.txt
