"""profiles/valu_<workload>[.fma].json from a scripts/prof.sh summary: dynamic VALU wave-instructions per launch (SQ_INSTS_VALU) of the kernel that ran and the in-kernel
clock measured for it (scripts/clock_probe.py) — what bench.py's roofline_valu is computed from.

usage: python scripts/make_valu.py <summary.txt> <bench line json> <in-kernel GHz exact> [<in-kernel GHz fma>] <round>
"""
import json, re, sys

summary, line = sys.argv[1], json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][0])
ghz = [float(x) for x in sys.argv[3:-1]]
rnd = sys.argv[-1]
workload = line["config"]["workload"].split(":")[0]
n = line["config"]["images_per_gpu"]
blocks, cur = {}, None
for l in open(summary):
    m = re.match(r"== counters \(mean per launch\): (.*)", l)
    if m:
        cur = m.group(1).strip()
        blocks[cur] = {}
    elif cur and re.match(r"\s+\S+\s+[\d.]+\s+\(n=", l):
        k, v = l.split()[:2]
        blocks[cur][k] = float(v)


def write(kernel_name, flavour_ghz, suffix):
    # the counter block of the launch form that ran: template arguments of the mangled name carry the flavour (FMA = true / false)
    fam = kernel_name.split("<")[0].replace("resize_", "resize_")
    want_fma = ",fma" in kernel_name
    for name, c in blocks.items():
        if fam.split("+")[0].strip() not in name.replace("_kernel", ""):
            continue
        if "SQ_INSTS_VALU" not in c:
            continue
        is_fma = c.get("SQ_INSTS_VALU_FMA_F32", 0) > 0 or ", true," in name[:120] and "FMA" in name
        if c.get("SQ_INSTS_VALU_FMA_F32") is not None and (c["SQ_INSTS_VALU_FMA_F32"] > 0) != want_fma:
            continue
        rec = {"workload": workload, "kernel": kernel_name, "valu_insts_per_launch": c["SQ_INSTS_VALU"], "images_per_launch": n, "inkernel_ghz": flavour_ghz,
               "cycles_per_inst": 2.05, "simds": 1024, "round": rnd, "source": summary,
               "classes": {k: c[k] for k in ("SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_CVT") if k in c}}
        path = "profiles/valu_%s%s.json" % (workload, suffix)
        json.dump(rec, open(path, "w"), indent=1)
        print(path, rec["valu_insts_per_launch"], flavour_ghz)
        return
    print("no counter block for", kernel_name)


write(line["config"]["kernel"], ghz[0], "")
if "other_numerics" in line and len(ghz) > 1:
    write(line["other_numerics"]["kernel"], ghz[1], ".fma")
