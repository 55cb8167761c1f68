"""VALU utilisation of a kernel from its DYNAMIC instruction counts (rocprofv3 PMC passes, scripts/prof.sh summary) at the IN-KERNEL clock (scripts/clock_probe.py).

  utilisation = VALU wave-instructions per launch x 2.05 cycles / (1024 SIMDs x kernel cycles),  kernel cycles = kernel time x in-kernel clock

2.05 cycles per wave64 instruction per SIMD is the measured issue rate at an even number of co-resident waves (scripts/ubench/valu_issue.hip, profiles/r05_ubench_valu_issue.txt;
the MI355X guide's 2).  It replaces r04's "issue time at 2.45 cycles" (a microbenchmark whose residency was a chip-wide average: VERDICT r04), which put several kernels above 100 % —
a utilisation cannot be.  A stream that contains SDWA, DPP or packed-f32 instructions issues at 4.0 cycles per instruction (same file): for those kernels the figure at 4.0 is given too.
Also printed, from the same counters: waves resident per SIMD (SQ_WAVE_CYCLES), the share of a wave's life parked on s_waitcnt (SQ_WAIT_ANY) and waiting to issue (SQ_WAIT_INST_ANY),
and the fabric traffic.

usage: python scripts/issue_util.py <prof summary.txt> <kernel name substring> <in-kernel GHz> [label [n [sdwa]]]
       n: which "launches" line of the trace with that kernel prefix gives the time (the trace truncates names: 0 = the headline flavour, 1 = the other); sdwa: 1 = the stream has SDWA / DPP
"""
import re, sys
path, want, ghz = sys.argv[1], sys.argv[2], float(sys.argv[3])
label = sys.argv[4] if len(sys.argv) > 4 else want
nth = int(sys.argv[5]) if len(sys.argv) > 5 else 0
sdwa = len(sys.argv) > 6 and sys.argv[6] == "1"
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
ms = None
for m in re.finditer(r"launches (.*?)n=(\d+)\s+all-launch avg ([0-9.]+) ms\s+median ([0-9.]+) ms\s+last (\d+) launches avg ([0-9.]+) ms", text):
    if want in m.group(1) or want.split("<")[0] in m.group(1):
        ms = float(m.group(6))
        if nth == 0:
            break
        nth -= 1
valu = c.get("SQ_INSTS_VALU", 0.0)
mul, add, fma, cvt = (c.get("SQ_INSTS_VALU_" + k, 0.0) for k in ("MUL_F32", "ADD_F32", "FMA_F32", "CVT"))
print("%s" % label)
print("  kernel            %s" % name[:150])
print("  VALU wave-instructions per launch: %.1f M  (mul %.1f M, add %.1f M, fma %.1f M, cvt %.1f M, other %.1f M)" % (valu / 1e6, mul / 1e6, add / 1e6, fma / 1e6, cvt / 1e6, max(0.0, valu - mul - add - fma - cvt) / 1e6))
if ms:
    cyc = ms * 1e-3 * ghz * 1e9
    u = valu * 2.05 / (1024.0 * cyc)
    line = "  kernel time %.4f ms (last launches of the trace) at %.3f GHz in-kernel = %.2f M cycles  ->  VALU utilisation %.0f %%" % (ms, ghz, cyc / 1e6, 100 * u)
    if sdwa:
        line += "  (%.0f %% at the 4.0 cycles of a stream with SDWA / DPP instructions)" % (100 * valu * 4.0 / (1024.0 * cyc))
    print(line)
    wc, wa, wi = c.get("SQ_WAVE_CYCLES"), c.get("SQ_WAIT_ANY"), c.get("SQ_WAIT_INST_ANY")
    if wc and wa:
        print("  waves resident per SIMD %.2f (%d waves per launch); of a wave's life: %.0f %% parked on s_waitcnt, %.0f %% waiting to issue" % (
            wc * 4 / cyc / 1024.0, c.get("SQ_WAVES", 0), 100 * wa / wc, 100 * (wi or 0) / wc))
fetch, write = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
if fetch and write:
    print("  fabric traffic per launch: read %.3f GB (FETCH_SIZE x 2 on gfx950) + written %.3f GB" % (fetch * 2 * 1024 / 1e9, write * 1024 / 1e9))
