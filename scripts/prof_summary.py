#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output of scripts/prof.sh: per-kernel time stats and per-launch counter means."""
import csv, glob, os, sys, collections
root = sys.argv[1]
def find(pattern):
    return sorted(glob.glob(os.path.join(root, pattern), recursive=True))
# kernel stats
for f in find("trace/**/*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, root))
    for row in csv.DictReader(open(f)):
        print("  %-90s calls %5s  avg %10.1f ns  total %12s ns  %5s%%" % (row["Name"][:90], row["Calls"], float(row["AverageNs"]), row["TotalDurationNs"], row["Percentage"]))
# per-launch durations from the trace: the stats file averages EVERY launch, including the ~9 cold ones at the start of a
# process (idle clocks, profiles/r01_launch_series.txt) that bench.py's untimed setup absorbs; the steady state is what
# bench.py's timed region sees
for f in find("trace/**/*kernel_trace.csv"):
    per = collections.OrderedDict()
    for row in csv.DictReader(open(f)):
        per.setdefault(row["Kernel_Name"], []).append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    for n, d in per.items():
        if len(d) >= 12 and any(k in n for k in ("resize", "convert")):
            tail = d[-20:] if len(d) >= 30 else d[len(d) // 2:]
            sd = sorted(d)
            print("  launches %-62s n=%d  all-launch avg %.4f ms  median %.4f ms  last %d launches avg %.4f ms  first 3: %s" % (
                n[:62], len(d), sum(d) / len(d), sd[len(sd) // 2], len(tail), sum(tail) / len(tail), " ".join("%.3f" % x for x in d[:3])))
# per-dispatch trace: VGPR/LDS/grid for our kernels
for f in find("trace/**/*kernel_trace.csv"):
    seen = set()
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if n in seen: continue
        seen.add(n)
        keys = [k for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size_X", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z") if k in row]
        print("  dispatch %-70s %s" % (n[:70], " ".join("%s=%s" % (k, row[k]) for k in keys)))
# counters: mean per dispatch for kernels matching resize/convert
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in find("pmc*/**/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    if not any(s in k for s in ("resize", "convert", "pass_generic")):
        continue
    print("== counters (mean per launch):", k[:100])
    for c in sorted(cs):
        v = cs[c]
        print("  %-40s %18.1f   (n=%d)" % (c, sum(v) / len(v), len(v)))
