"""Per-SIMD occupancy timeline of ONE launch of a shipped kernel (r05; VERDICT r04 item 1b: "find where 29 % of the wave slots go").

A -DPICHA_CLOCK_PROBE build (scripts/build_variant.sh) has every wave record the chip-wide s_memrealtime of its start, its duration, and
HW_REG_HW_ID / HW_REG_XCC_ID.  From one launch's records this script rebuilds, per SIMD (XCC, SE, SH, CU, SIMD):
  * how many waves were resident over time (share of SIMD-time at 0 / 1 / 2 / 3 / 4 ... waves; mean; chip-wide profile over the launch in 50 bins),
  * the turnover gap: from a wave's end to the start of the wave that takes its slot (greedy match of ends to later starts on the same SIMD),
  * the launch's ramp (first start -> every SIMD at its typical level) and tail (first SIMD that runs dry for good -> last end).

usage: PICHA_HIP_LIB=picha_amd/variants/libpicha_hip_probe.so python scripts/r05_timeline.py [case[:opt=value,...] ...]      cases: cfg2 cfg2fma cfg5 cfg3b
"""
import ctypes
import sys
import time

sys.path.insert(0, '.')
import numpy as np
import torch

from picha_amd import _native as N
from picha_amd.device import DeviceBatch, ResizePlan

CASES = {
    "cfg2": (256, "rgba", 3840, 2160, 1920, 1080, {}, {}, "picha_probe_sym_p1"),
    "cfg2fma": (256, "rgba", 3840, 2160, 1920, 1080, {}, {"numerics": 1}, "picha_probe_sym_p1"),
    "cfg5": (32, "r16g16b16a16", 7680, 4320, 1920, 1080, {"filter": "mitchel"}, {}, "picha_probe_run_p7"),
    "cfg3b": (256, "grey", 3840, 2160, 512, 512, {"filter": "lanczos"}, {}, "picha_probe_run_p2"),
    # colorConvert + resize as one call (the fused row-staged kernel): popt carries the destination format
    "cfg3": (256, "rgb", 3840, 2160, 512, 512, {"filter": "lanczos", "convert": "grey"}, {}, "picha_probe_rows_v2"),
    "lanczos2x": (64, "rgba", 3840, 2160, 1920, 1080, {"filter": "lanczos"}, {}, "picha_probe_run_p1"),
    "lanczos8x": (64, "rgb", 3840, 2160, 480, 270, {"filter": "lanczos"}, {}, "picha_probe_span_p0"),
    "jpegthumb": (192, "rgb", 3840, 2160, 512, 288, {}, {}, "picha_probe_span_p0"),
}
SLOTS = 1 << 18


def analyse(rec, ms_launch):
    rec = rec[rec[:, 1] > 0]
    start = rec[:, 2].astype(np.int64)
    end = start + rec[:, 1].astype(np.int64)
    hw = rec[:, 3]
    # HW_ID (gfx9 layout): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]; XCC_ID[3:0] in the upper word
    key = ((hw >> np.uint64(32)) & np.uint64(0xf)) << np.uint64(16) | (hw & np.uint64(0xff30))
    t0, t1 = int(start.min()), int(end.max())
    span = t1 - t0
    simds = np.unique(key)
    ghz = float(np.median(rec[:, 0].astype(np.float64) / rec[:, 1].astype(np.float64))) * 0.1
    print("  waves %d on %d SIMDs; launch span by the wave stamps %.4f ms (events: %.4f ms per launch); in-kernel clock %.3f GHz; median wave life %.1f us (p10 %.1f, p90 %.1f)"
          % (len(rec), len(simds), span * 1e-5, ms_launch, ghz, float(np.median(end - start)) * 0.01, float(np.percentile(end - start, 10)) * 0.01,
             float(np.percentile(end - start, 90)) * 0.01))
    order = np.argsort(key, kind="stable")
    key_s, start_s, end_s = key[order], start[order], end[order]
    bounds = np.flatnonzero(np.diff(key_s)) + 1
    groups = np.split(np.arange(len(key_s)), bounds)
    level_time = np.zeros(16)
    gaps = []
    per_simd_mean = []
    nb = 50
    prof = np.zeros(nb)
    full_at, dry_at = [], []
    for g in groups:
        s, e = np.sort(start_s[g]), np.sort(end_s[g])
        # occupancy steps
        ev = np.concatenate([np.stack([s, np.ones_like(s)], 1), np.stack([e, -np.ones_like(e)], 1)])
        ev = ev[np.lexsort((ev[:, 1], ev[:, 0]))]
        lvl, prev = 0, t0
        peak = 0
        occ_int = 0
        tl = []
        for t, d in ev:
            if t > prev:
                level_time[min(lvl, 15)] += t - prev
                occ_int += lvl * (t - prev)
                tl.append((prev, t, lvl))
            lvl += int(d)
            peak = max(peak, lvl)
            prev = t
        if t1 > prev:
            level_time[0] += t1 - prev
            tl.append((prev, t1, 0))
        per_simd_mean.append(occ_int / span)
        # chip-wide profile in nb bins
        for a, b, l in tl:
            if l == 0:
                continue
            ia, ib = (a - t0) * nb / span, (b - t0) * nb / span
            k = int(ia)
            while k < ib and k < nb:
                lo, hi = max(ia, k), min(ib, k + 1)
                prof[k] += l * (hi - lo)
                k += 1
        # turnover gaps: every end matched with the first later start on this SIMD not yet taken
        j = 0
        for x in e:
            while j < len(s) and s[j] < x:
                j += 1
            if j < len(s):
                gaps.append(s[j] - x)
                j += 1
        # ramp: first time the SIMD reaches its peak level; tail: start of its final stretch below peak / 2
        lv = 0
        for t, d in ev:
            lv += int(d)
            if lv >= peak:
                full_at.append(t - t0)
                break
        dry_at.append(int(e[-1]) - t0)
    per_simd_mean = np.array(per_simd_mean)
    tot = level_time.sum()
    print("  resident waves per SIMD, time-averaged: mean %.2f (p10 %.2f, p90 %.2f over SIMDs); share of SIMD-time at n waves: %s"
          % (per_simd_mean.mean(), np.percentile(per_simd_mean, 10), np.percentile(per_simd_mean, 90),
             "  ".join("n=%d %.1f%%" % (i, 100 * level_time[i] / tot) for i in range(16) if level_time[i] > 0)))
    gaps = np.array(gaps if gaps else [0.0], dtype=np.float64) * 0.01   # a persistent grid has no turnover at all
    print("  turnover gap (a wave's end -> the next wave start on that SIMD): %d matched, median %.2f us, mean %.2f us, p90 %.2f us, p99 %.2f us; sum of gaps = %.1f%% of all wave-slot time (4 slots x SIMDs x span)"
          % (len(gaps), np.median(gaps), gaps.mean(), np.percentile(gaps, 90), np.percentile(gaps, 99), 100 * gaps.sum() * 100 / (4.0 * len(simds) * span)))
    print("  ramp: SIMDs reach their peak level %.1f us (median; p90 %.1f) after the first wave start; last wave of a SIMD ends %.1f us (median; p10 %.1f) before ... span %.1f us"
          % (np.median(full_at) * 0.01, np.percentile(full_at, 90) * 0.01, (span - np.median(dry_at)) * 0.01, (span - np.percentile(dry_at, 10)) * 0.01, span * 0.01))
    print("  chip-wide mean resident waves per SIMD over the launch (50 bins): " + " ".join("%.1f" % (v / len(simds)) for v in prof))


def main():
    lib = N.lib()
    torch.cuda.set_device(0)
    print("library: %s" % N.SO_PATH)
    for label in (sys.argv[1:] or ["cfg2"]):   # "cfg2" or "cfg2:work=4,block_waves=1" (library options for this case)
        case, _, optstr = label.partition(":")
        n, pixel, sw, sh, dw, dh, popt, lopt, sym = CASES[case]
        lopt = dict(lopt)
        lopt.update({k: int(v) for k, v in (kv.split("=") for kv in optstr.split(",") if kv)})
        probe = getattr(lib, sym)
        probe.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
        N.check(lib.picha_hip_set_option(b"numerics", 0))
        for k, v in lopt.items():
            N.check(lib.picha_hip_set_option(k.encode(), v))
        src = DeviceBatch(n, sw, sh, pixel).fill_random(1)
        dst = DeviceBatch(n, dw, dh, popt.get("convert", pixel))
        if popt.get("convert"):
            from picha_amd.device import convert_resize_batch

            class _Chain:
                def run(self):
                    convert_resize_batch(src, popt["convert"], dw, dh, filter=popt.get("filter"), out=dst)
            plan = _Chain()
        else:
            plan = ResizePlan(src, dst, filter=popt.get("filter"), filter_scale=popt.get("filterScale"))
        plan.run()
        torch.cuda.synchronize()
        name = lib.picha_hip_last_kernel().decode()
        t0 = time.time()
        while time.time() - t0 < 2.0:
            for _ in range(20):
                plan.run()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            plan.run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        out = (ctypes.c_ulonglong * (4 * SLOTS))()
        assert probe(out, 1) == SLOTS
        plan.run()   # the one launch that is analysed (back to back with the warm ones above)
        torch.cuda.synchronize()
        assert probe(out, 0) == SLOTS
        rec = np.frombuffer(out, dtype=np.uint64).reshape(SLOTS, 4)
        print("%s  %s" % (label, name))
        analyse(rec, ms)
        for k in lopt:
            N.check(lib.picha_hip_set_option(k.encode(), 0))
        del src, dst


if __name__ == "__main__":
    main()
