#!/usr/bin/env python3
"""profiles/traffic_<workload>.json from a scripts/prof.sh summary: fabric bytes per launch of the kernel bench.py reports.

usage: python scripts/make_traffic.py <summary.txt> <bench_line.json> <round> > profiles/traffic_<workload>.json
The summary's first counter block is the headline kernel (most launches).  gfx950: FETCH_SIZE counts 128-byte read requests as
64 bytes (MI355X_MICROARCH.md, HBM section), so read bytes = FETCH_SIZE * 1024 * 2 = TCC_EA0_RDREQ * 128."""
import json, re, sys
summary, bench, rnd = sys.argv[1], sys.argv[2], int(sys.argv[3])
line = [l for l in open(bench) if l.startswith("{")][-1]
b = json.loads(line)
blocks, cur = [], None
for l in open(summary):
    if l.startswith("== counters"):
        cur = {"name": l.strip(), "c": {}}
        blocks.append(cur)
    elif cur is not None:
        m = re.match(r"\s+(\S+)\s+([0-9.]+)\s+\(n=(\d+)\)", l)
        if m:
            cur["c"][m.group(1)] = (float(m.group(2)), int(m.group(3)))
blk = max(blocks, key=lambda x: x["c"].get("FETCH_SIZE", (0, 0))[1])
c = blk["c"]
read_b = c["FETCH_SIZE"][0] * 1024 * 2
write_b = c["WRITE_SIZE"][0] * 1024
alg = b["roofline"]["algorithmic_bytes_per_launch"]
n = b["config"]["images_per_gpu"]
out = {
    "workload": b["config"]["workload"].split(":")[0],
    "kernel": b["roofline"]["kernel"],
    "round": rnd,
    "source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ_sum, separate passes, mean of %d launches; counter block: %s)" % (summary, c["FETCH_SIZE"][1], blk["name"][:120]),
    "FETCH_SIZE_KB": c["FETCH_SIZE"][0],
    "WRITE_SIZE_KB": c["WRITE_SIZE"][0],
    "TCC_EA0_RDREQ_sum": c.get("TCC_EA0_RDREQ_sum", (None, 0))[0],
    "correction": "gfx950 FETCH_SIZE counts 128-B read requests as 64 B (MI355X_MICROARCH.md, HBM): read bytes = FETCH_SIZE*1024*2 = TCC_EA0_RDREQ*128",
    "read_bytes_per_launch": int(read_b),
    "write_bytes_per_launch": int(write_b),
    "hbm_bytes_per_launch": int(read_b + write_b),
    "algorithmic_bytes_per_launch": alg,
    "ratio_to_algorithmic": round((read_b + write_b) / alg, 3),
    "images_per_launch": n,
    "note": "requests leaving the L2 towards the fabric; Infinity Cache hits are included, so HBM itself moves less",
}
print(json.dumps(out, indent=2))
