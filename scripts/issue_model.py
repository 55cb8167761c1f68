"""VALU issue time of a kernel from its DYNAMIC instruction counts (rocprofv3 PMC passes, scripts/prof.sh summary) at the IN-KERNEL clock
(scripts/r04_clock_probe.py), beside the measured kernel time: the derivation VERDICT r03 asked for instead of "argued issue-bound".

  issue ticks = sum over classes of wave-instructions x ticks per wave-instruction per SIMD (profiles/r01_ubench_valu_issue_rates.txt, 8 waves per SIMD):
      f32 multiplies 1.38 (the SDWA unpack multiplies of 16-bit / grey formats 2.38 — not separable in the counters: given as a range),
      f32 adds 1.37, fused multiply-adds 1.38, conversions 1.69, everything else (moves, DPP, clamps, integer address work) 0.88 .. 2.38
  issue time  = issue ticks / 1024 SIMDs / in-kernel clock

usage: python scripts/issue_model.py <prof summary.txt> <kernel name substring> <in-kernel GHz> [label [n]]
       n: which "launches" line of the trace with that kernel prefix gives the time (the trace truncates names: 0 = the headline flavour, 1 = the other)
"""
import re, sys
path, want, ghz = sys.argv[1], sys.argv[2], float(sys.argv[3])
label = sys.argv[4] if len(sys.argv) > 4 else want
text = open(path).read()
blocks = re.split(r"^== counters \(mean per launch\): ", text, flags=re.M)[1:]
c, name = None, None
for b in blocks:
    head, _, body = b.partition("\n")
    if want in head:
        name = head.strip()
        c = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\S+)\s+([0-9.]+)\s+\(n=", body, flags=re.M)}
        break
if c is None:
    sys.exit("no counters for a kernel matching %r in %s" % (want, path))
ms, nth = None, int(sys.argv[5]) if len(sys.argv) > 5 else 0
for m in re.finditer(r"launches (.*?)n=(\d+)\s+all-launch avg ([0-9.]+) ms\s+median ([0-9.]+) ms\s+last (\d+) launches avg ([0-9.]+) ms", text):
    if want in m.group(1) or want.split("<")[0] in m.group(1):
        ms = float(m.group(6))
        if nth == 0:
            break
        nth -= 1
valu = c.get("SQ_INSTS_VALU", 0.0)
mul, add, fma, cvt = (c.get("SQ_INSTS_VALU_" + k, 0.0) for k in ("MUL_F32", "ADD_F32", "FMA_F32", "CVT"))
other = max(0.0, valu - mul - add - fma - cvt)
lo = mul * 1.38 + add * 1.37 + fma * 1.38 + cvt * 1.69 + other * 0.88
hi = mul * 1.38 + add * 1.37 + fma * 1.38 + cvt * 1.69 + other * 2.38
mid = mul * 1.38 + add * 1.37 + fma * 1.38 + cvt * 1.69 + other * 1.7
t = lambda ticks: ticks / 1024.0 / (ghz * 1e9) * 1e3
print("%s" % label)
print("  kernel            %s" % name[:150])
print("  VALU wave-instructions per launch: %.1f M  (mul %.1f M, add %.1f M, fma %.1f M, cvt %.1f M, other %.1f M)" % (valu / 1e6, mul / 1e6, add / 1e6, fma / 1e6, cvt / 1e6, other / 1e6))
print("  issue ticks per SIMD: %.0f K (other instructions at 1.7 ticks; %.0f K .. %.0f K at 0.88 .. 2.38)" % (mid / 1024e3, lo / 1024e3, hi / 1024e3))
print("  issue time at the in-kernel clock of %.3f GHz: %.3f ms (%.3f .. %.3f)" % (ghz, t(mid), t(lo), t(hi)))
if ms:
    wc, wa = c.get("SQ_WAVE_CYCLES"), c.get("SQ_WAIT_ANY")
    print("  kernel time (last launches of the trace): %.3f ms  ->  VALU issue = %.0f %% of it (%.0f .. %.0f %%)%s" % (
        ms, 100 * t(mid) / ms, 100 * t(lo) / ms, 100 * t(hi) / ms,
        "; waves parked on s_waitcnt %.0f %% of their lifetime, %d waves per launch" % (100 * wa / wc, c.get("SQ_WAVES", 0)) if wc and wa else ""))
fetch, write = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
if fetch and write:
    print("  fabric traffic per launch: read %.3f GB (FETCH_SIZE x 2 on gfx950) + written %.3f GB" % (fetch * 2 * 1024 / 1e9, write * 1024 / 1e9))
