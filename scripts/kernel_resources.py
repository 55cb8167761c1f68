"""Registers / LDS / scratch of every device kernel in libpicha_hip.so (from the code objects' metadata notes).
usage: python scripts/kernel_resources.py [substring-filter]"""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
so = os.environ.get("PICHA_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "picha_amd", "libpicha_hip.so")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
with tempfile.TemporaryDirectory() as d:
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call(["objcopy", "--dump-section", ".hip_fatbin=" + fat, so, os.path.join(d, "discard.so")])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)] + [len(blob)]
    rows = []
    for i in range(len(starts) - 1):
        part, co = os.path.join(d, "b%d.bin" % i), os.path.join(d, "d%d.co" % i)
        open(part, "wb").write(blob[starts[i]:starts[i + 1]])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        if os.path.getsize(co) == 0:
            continue
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip()
            if k == "agpr_count" and cur.get("name"):
                rows.append(cur); cur = {}
            cur[k] = v
        if cur.get("name"):
            rows.append(cur)
    seen = set()
    for r in rows:
        name = r.get("name", "")
        if name in seen or not name.startswith("_Z"):
            continue
        seen.add(name)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"picha_hip::\(anonymous namespace\)::", "", dem).split("(picha_hip::")[0].replace("void ", "")
        if flt in dem:
            print("%-70s vgpr %4s sgpr %4s lds %6s scratch %5s" % (dem[:70], r.get("vgpr_count"), r.get("sgpr_count"),
                  r.get("group_segment_fixed_size"), r.get("private_segment_fixed_size")))
