"""Placement-proof interleaved A/B of library BUILDS and / or OPTIONS (r05).

Where the destination buffer sits in HBM moves cfg2 by up to 5.7 % (profiles/r04_placement.txt) — more than most A/Bs claim.  So every variant compared
here runs on ONE source batch and ONE destination batch, allocated once, in ONE process: each build of the library is dlopen'ed side by side (own tables
and caches, same HIP runtime), each variant = (build, options), and the variants are timed in rotating order over several rounds.  Every variant's
output (all images) is reduced on the device and compared with the first variant's — which is compared with the CPU oracle on images 0 and n-1.

usage: python scripts/ab_shared.py [--case cfg2|cfg5|cfg3b|cfg3|n:pixel:sw:sh:dw:dh[:filter[:scale]]] [--rounds 6] [--reps 10] [--numerics exact|fma]
                                   name=LIB[:key=value,...] ...
       LIB: path of a libpicha_hip*.so, or "." for picha_amd/libpicha_hip.so
prints per variant: kernel name, ms per launch per round, median, min, % against the first variant, parity.
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, '.')
import numpy as np
import torch

import oracle
import picha_amd
from picha_amd import _native as N
from picha_amd.device import DeviceBatch

CASES = {
    "cfg2": (256, "rgba", 3840, 2160, 1920, 1080, None, None),
    "cfg2_7tap": (256, "rgba", 3840, 2160, 1920, 1080, "cubic", None),
    "cfg5": (32, "r16g16b16a16", 7680, 4320, 1920, 1080, "mitchel", None),
    "cfg3b": (256, "grey", 3840, 2160, 512, 512, "lanczos", None),
    "cfg1": (256, "rgba", 640, 480, 320, 240, None, None),
    "lanczos2x": (64, "rgba", 3840, 2160, 1920, 1080, "lanczos", None),
    "lanczos8x": (64, "rgb", 3840, 2160, 480, 270, "lanczos", None),
    "jpegthumb": (192, "rgb", 3840, 2160, 512, 288, None, None),
}
# colorConvert + resize as one call (picha_hip_convert_resize_device, convert first): case -> (resize case fields..., destination pixel)
CHAINED = {
    "cfg3": (256, "rgb", 3840, 2160, 512, 512, "lanczos", None, "grey"),
}


def load(path):
    L = ctypes.CDLL(path)
    for name, res, args in N.SYMBOLS:
        try:
            fn = getattr(L, name)
        except AttributeError:
            continue
        fn.restype = res
        fn.argtypes = args
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="cfg2")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--numerics", default="exact")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    convert_to = None
    if a.case in CHAINED:
        n, pixel, sw, sh, dw, dh, filt, fscale, convert_to = CHAINED[a.case]
    elif a.case in CASES:
        n, pixel, sw, sh, dw, dh, filt, fscale = CASES[a.case]
    else:
        f = a.case.split(":")
        n, pixel, sw, sh, dw, dh = int(f[0]), f[1], int(f[2]), int(f[3]), int(f[4]), int(f[5])
        filt = f[6] if len(f) > 6 and f[6] else None
        fscale = float(f[7]) if len(f) > 7 else None
    from picha_amd.api import _resize_options
    o = {}
    if filt:
        o["filter"] = filt
    if fscale:
        o["filterScale"] = fscale
    ftag, fs = _resize_options(o)

    torch.cuda.set_device(0)
    base = N.lib()   # torch's HIP runtime first, then the shipped library
    libs, variants = {}, []
    for v in a.variants:
        name, rest = v.split("=", 1)
        path, _, optstr = rest.partition(":")
        path = N.SO_PATH if path == "." else os.path.abspath(path)
        if path not in libs:
            libs[path] = base if path == os.path.abspath(N.SO_PATH) else load(path)
        opts = dict(kv.split("=") for kv in optstr.split(",") if kv)
        variants.append((name, libs[path], {k: int(x) for k, x in opts.items()}, os.path.basename(path)))   # UPPER-CASE keys are environment variables of measurement builds
    src = DeviceBatch(n, sw, sh, pixel).fill_random(20250226)
    dst = DeviceBatch(n, dw, dh, convert_to or pixel)
    sd, dd = src.descriptors(), dst.descriptors()
    from picha_amd.api import color_settings
    cr, cg, cb = color_settings()
    stream = torch.cuda.current_stream().cuda_stream
    print("# %s: %d x %dx%d %s -> %dx%d filter %s; ONE source at 0x%x, ONE destination at 0x%x (offset %+d MB); %d rounds x %d launches, %s numerics"
          % (a.case, n, sw, sh, pixel, dw, dh, filt or "default", src.tensor.data_ptr(), dst.tensor.data_ptr(),
             (dst.tensor.data_ptr() - src.tensor.data_ptr()) >> 20, a.rounds, a.reps, a.numerics))
    props = torch.cuda.get_device_properties(0)
    print("# device %s %s" % (torch.cuda.get_device_name(0), getattr(props, "uuid", "")))
    touched = {}

    def apply(L, opts):
        # restore whatever an earlier variant of the same build changed, then set this variant's options
        for k, old in touched.get(id(L), {}).items():
            L.picha_hip_set_option(k.encode(), old)
        touched[id(L)] = {}
        L.picha_hip_set_option(b"numerics", 1 if a.numerics == "fma" else 0)
        for k in [k for k in os.environ if k.startswith("PICHA_SYM_")]:
            del os.environ[k]
        for k, x in opts.items():
            if k.isupper():
                os.environ[k] = str(x)
                continue
            touched[id(L)][k] = L.picha_hip_get_option(k.encode())
            rc = L.picha_hip_set_option(k.encode(), x)
            if rc:
                raise SystemExit("option %s=%d refused by %s" % (k, x, L._name))

    def run(L):
        if convert_to:
            rc = L.picha_hip_convert_resize_device(n, sd, dd, 0, cr, cg, cb, ftag, fs, stream)
        else:
            rc = L.picha_hip_resize_device(n, sd, dd, ftag, fs, stream)
        if rc:
            raise SystemExit("resize failed: %s" % L.picha_hip_last_error().decode())

    def digest():
        t = dst.tensor.view(torch.int32) if dst.tensor.numel() % 4 == 0 else dst.tensor
        t = t.to(torch.int64)
        w = torch.arange(1, t.shape[1] + 1, device=t.device, dtype=torch.int64)
        return int((t * w[None, :]).sum().item() & 0x7FFFFFFFFFFFFFFF), int(t.sum().item())

    res = {v[0]: [] for v in variants}
    kern, par = {}, {}
    ref_digest = None
    for rnd in range(a.rounds):
        order = variants[rnd % len(variants):] + variants[:rnd % len(variants)]
        for name, L, opts, libname in order:
            apply(L, opts)
            dst.tensor.zero_() if rnd == 0 else None
            run(L)
            run(L)
            torch.cuda.synchronize()
            if rnd == 0:
                kern[name] = L.picha_hip_last_kernel().decode()
                d = digest()
                if ref_digest is None:
                    ref_digest = d
                    if not a.no_oracle:
                        diffs = []
                        for i in (0, n - 1):
                            if convert_to:
                                mid, _ = oracle.color_convert(src.image_host(i).data, sw, sh, pixel, convert_to)
                                e, _ = oracle.resize(mid, sw, sh, convert_to, dw, dh, filt or "cubic", fs)
                            else:
                                e, _ = oracle.resize(src.image_host(i).data, sw, sh, pixel, dw, dh, filt or "cubic", fs)
                            exp = picha_amd.Image(width=dw, height=dh, pixel=convert_to or pixel, data=e).pixels().astype(np.int64)
                            diffs.append(int(np.abs(dst.image_host(i).pixels().astype(np.int64) - exp).max()))
                        par[name] = "oracle diff %d (images 0, %d)" % (max(diffs), n - 1)
                    else:
                        par[name] = "reference variant"
                else:
                    par[name] = "== first variant (all %d images)" % n if d == ref_digest else "DIFFERS from the first variant"
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                run(L)
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / a.reps)
    first = float(np.median(res[variants[0][0]]))
    for name, L, opts, libname in variants:
        r = res[name]
        print("%-18s %-28s %-44s median %.4f ms  min %.4f  (%+.1f%%)  rounds: %s | %s | %s"
              % (name, libname, ",".join("%s=%d" % kv for kv in opts.items()) or "-", float(np.median(r)), min(r), 100 * (float(np.median(r)) / first - 1),
                 " ".join("%.3f" % x for x in r), kern[name], par[name]), flush=True)


main()
