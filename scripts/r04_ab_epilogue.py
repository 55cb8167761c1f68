"""resize -> colorConvert of 4K batches at 2x: one kernel (store epilogue, fuse = 1) against the two-kernel chain (fuse = 2) and the
planner's choice (fuse = 0), interleaved, both numerics flavours.  Byte-equality of the three forms is checked on image 0."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, convert_resize_batch
def opt(k, v): N.check(N.lib().picha_hip_set_option(k.encode(), v))
CASES = [(128, "rgba", "grey", None), (128, "rgba", "rgb", None), (128, "rgba", "greya", None), (128, "rgba", "grey", "cubic"),
         (128, "rgb", "grey", None), (32, "r16g16b16a16", "rgba", None), (32, "r16g16b16a16", "grey", None)]
for n, sp, dp, filt in CASES:
    src = DeviceBatch(n, 3840, 2160, sp).fill_random(3)
    out = DeviceBatch(n, 1920, 1080, dp)
    for num in (0, 1):
        opt("numerics", num)
        res, names, ref = {}, {}, None
        for rnd in range(3):
            for fuse in (2, 1, 0):
                opt("fuse", fuse)
                convert_resize_batch(src, dp, 1920, 1080, filter=filt, order=1, out=out); torch.cuda.synchronize()
                if rnd == 0:
                    img = out.image_host(0).pixels()
                    if ref is None: ref = img
                    same = bool(np.array_equal(img, ref))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): convert_resize_batch(src, dp, 1920, 1080, filter=filt, order=1, out=out)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(fuse, []).append(e0.elapsed_time(e1) / 10)
                names[fuse] = (N.lib().picha_hip_last_kernel().decode(), same if rnd == 0 else names.get(fuse, ("", True))[1])
        for fuse in (2, 1, 0):
            print("%-13s -> 1080p -> %-6s %-6s %-5s fuse %d: %s  min %.3f ms  %s%s" % (sp, dp, filt or "default", "fma" if num else "exact", fuse, " ".join("%.3f" % x for x in res[fuse]),
                  min(res[fuse]), names[fuse][0][:90], "" if names[fuse][1] else "  DIFFERS from the chain"))
    opt("numerics", 0); opt("fuse", 0)
    del src, out
    torch.cuda.empty_cache()
