"""colorConvert -> integer upscale to 1080p: one kernel (resize_up with the convert per window pixel, fuse = 1) against the two-kernel chain (fuse = 2) and
the planner's choice (fuse = 0), interleaved over three rounds, both numerics flavours.  Byte-equality of the three forms is checked on image 0."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from picha_amd import _native as N
from picha_amd.device import DeviceBatch, convert_resize_batch
def opt(k, v): N.check(N.lib().picha_hip_set_option(k.encode(), v))
PAIRS = [("rgba", "rgb"), ("grey", "rgb"), ("r16g16b16", "rgb"), ("r16g16b16a16", "rgb"), ("rgb", "rgba"), ("greya", "rgba"), ("r16g16b16a16", "rgba"),
         ("rgb", "grey"), ("rgba", "grey"), ("greya", "grey"), ("r16", "grey")]
FOUR = {("rgba", "rgb"), ("grey", "rgb"), ("rgb", "rgba"), ("greya", "rgba"), ("rgb", "grey"), ("rgba", "grey")}
CASES = [(256, sp, dp, 960, 540) for sp, dp in PAIRS] + [(512, sp, dp, 480, 270) for sp, dp in PAIRS if (sp, dp) in FOUR]
if len(sys.argv) > 1 and sys.argv[1] == "4x":   # the 4x cases only, behind a spacer allocation of argv[2] MB (another placement of the buffers)
    CASES = [c for c in CASES if c[3] == 480]
    spacer = torch.empty(int(sys.argv[2]) << 20, dtype=torch.uint8, device="cuda") if len(sys.argv) > 2 else None
for n, sp, dp, sw, sh in CASES:
    src = DeviceBatch(n, sw, sh, sp).fill_random(3)
    out = DeviceBatch(n, 1920, 1080, dp)
    for num in (0, 1):
        opt("numerics", num)
        res, names, ref = {}, {}, None
        for rnd in range(3):
            for fuse in (2, 1, 0):
                opt("fuse", fuse)
                convert_resize_batch(src, dp, 1920, 1080, order=0, out=out); torch.cuda.synchronize()
                if rnd == 0:
                    img = out.image_host(0).pixels()
                    if ref is None: ref = img
                    same = bool(np.array_equal(img, ref))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): convert_resize_batch(src, dp, 1920, 1080, order=0, out=out)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(fuse, []).append(e0.elapsed_time(e1) / 10)
                names[fuse] = (N.lib().picha_hip_last_kernel().decode(), same if rnd == 0 else names.get(fuse, ("", True))[1])
        base = min(res[2])
        for fuse in (2, 1, 0):
            print("%3d x %-13s %dx%d -> %-5s 1080p %-5s fuse %d: %s  min %.3f ms (%+5.1f %%)  %s%s" % (n, sp, sw, sh, dp, "fma" if num else "exact", fuse, " ".join("%.3f" % x for x in res[fuse]),
                  min(res[fuse]), 100 * (min(res[fuse]) / base - 1), names[fuse][0][:110], "" if names[fuse][1] else "  DIFFERS from the chain"), flush=True)
    opt("numerics", 0); opt("fuse", 0)
    del src, out
    torch.cuda.empty_cache()
