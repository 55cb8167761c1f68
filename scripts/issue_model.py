"""VALU issue time of a kernel from its DYNAMIC instruction counts (rocprofv3 PMC passes, scripts/prof.sh summary) at the IN-KERNEL clock
(scripts/r04_clock_probe.py), beside the measured kernel time: the derivation VERDICT r03 asked for instead of "argued issue-bound".

  issue cycles = sum over classes of wave-instructions x cycles per wave-instruction per SIMD, as MEASURED with the number of resident waves checked
  (scripts/ubench/valu_cycles.hip, profiles/r04_ubench_valu_rates.txt — the r01 table it replaces divided by 8 "resident" waves where 4.4 were):
      f32 multiplies / adds / fused multiply-adds / moves / simple integer ops 2.45 (v_mul 2.46, v_add 2.42, v_fma 2.38, v_mov 2.46),
      conversions 4.3 (v_cvt_f32_ubyteN 4.43, v_cvt_pk_u8_f32 4.12, v_cvt_u32_f32 4.45),
      everything else 2.45 .. 4.4 (SDWA / DPP forms, v_med3, SGPR-operand multiplies, v_perm / v_alignbyte: 4.1 - 4.4; not separable in the counters: a range).
      The SDWA unpack multiplies of 16-bit / grey formats are counted as multiplies by the PMC: kernels that use them are UNDER-estimated.
  issue time  = issue cycles / 1024 SIMDs / in-kernel clock

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
FULL, CVT = 2.45, 4.3
# lower figure: EVERY VALU wave-instruction at the full rate — what the mixed streams of the ubench show (a conversion or clamp among multiply-adds costs one issue slot, it
# executes beside the multiply-add pipe); upper figure: the classes added up at their stand-alone rates (conversions 4.3, everything unclassified 4.4)
lo = valu * FULL
hi = (mul + add + fma) * FULL + cvt * CVT + other * 4.4
t = lambda cyc: cyc / 1024.0 / (ghz * 1e9) * 1e3
print("%s" % label)
print("  kernel            %s" % name[:150])
print("  VALU wave-instructions per launch: %.1f M  (mul %.1f M, add %.1f M, fma %.1f M, cvt %.1f M, other %.1f M)" % (valu / 1e6, mul / 1e6, add / 1e6, fma / 1e6, cvt / 1e6, other / 1e6))
print("  issue cycles per SIMD: %.0f K with every instruction at 2.45 cycles (conversions overlapped) .. %.0f K with the classes added up at their stand-alone rates" % (lo / 1024e3, hi / 1024e3))
print("  issue time at the in-kernel clock of %.3f GHz: %.3f .. %.3f ms" % (ghz, t(lo), t(hi)))
if ms:
    wc, wa = c.get("SQ_WAVE_CYCLES"), c.get("SQ_WAIT_ANY")
    print("  kernel time (last launches of the trace): %.3f ms  ->  VALU issue = %.0f .. %.0f %% of it%s" % (
        ms, 100 * t(lo) / ms, 100 * t(hi) / ms,
        "; waves parked on s_waitcnt %.0f %% of their lifetime, %d waves per launch" % (100 * wa / wc, c.get("SQ_WAVES", 0)) if wc and wa else ""))
fetch, write = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
if fetch and write:
    print("  fabric traffic per launch: read %.3f GB (FETCH_SIZE x 2 on gfx950) + written %.3f GB" % (fetch * 2 * 1024 / 1e9, write * 1024 / 1e9))
