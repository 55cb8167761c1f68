"""Instruction mix of a kernel's hot loop from the shipped code object: finds the largest backward-branch region of a kernel, counts
instructions by class (issue cost per class from profiles/r05_ubench_valu_issue.txt: every plain instruction one slot of 2.05 cycles inside a
multiply-add stream at an even number of co-resident waves; a stream that contains SDWA / DPP / packed-f32 forms issues at 4.0 as a whole — shown as a flag, not priced per instruction), lists the waits and nops.  STATIC counts: code of both sides of a branch is counted.

usage: python scripts/isa_loop_stats.py <libpicha_hip.so | object file> <substring of the mangled kernel name> [rows per iteration]
"""
import collections, os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
# issue cycles per wave-instruction per SIMD (r05_ubench_valu_issue.txt)
COST = {"v_mul_f32": 2.05, "v_add_f32": 2.05, "v_fma_f32": 2.05, "v_fmac_f32": 2.05, "v_mul_f32 (sgpr operand)": 2.05, "v_fma_f32 (sgpr operand)": 2.05,
        "sdwa": 2.05, "dpp": 2.05, "v_cvt": 2.05, "v_med3": 2.05, "v_alignbyte": 2.05, "v_mov": 2.05, "v_int": 2.05, "v_other": 2.05}


def classify(op, rest):
    sg = bool(re.search(r"(^|[ ,])s\d+|s\[\d+:\d+\]", rest.split("//")[0].split(",", 1)[1] if "," in rest else ""))
    if "_sdwa" in op: return "sdwa"
    if "_dpp" in op: return "dpp"
    if op in ("v_mul_f32_e32", "v_mul_f32_e64"): return "v_mul_f32 (sgpr operand)" if sg else "v_mul_f32"
    if op.startswith("v_add_f32") or op.startswith("v_sub_f32"): return "v_add_f32"
    if op.startswith("v_fma_f32"): return "v_fma_f32 (sgpr operand)" if sg else "v_fma_f32"
    if op.startswith("v_fmac_f32"): return "v_fma_f32 (sgpr operand)" if sg else "v_fmac_f32"
    if op.startswith("v_cvt"): return "v_cvt"
    if op.startswith("v_med3"): return "v_med3"
    if op.startswith("v_alignbyte"): return "v_alignbyte"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "v_mov"
    if re.match(r"v_(add|sub|and|or|xor|lshl|lshr|ashr|bfe|perm|min|max|mad_u|mul_u|mul_lo|mul_hi|cndmask|cmp|readfirstlane|readlane)", op): return "v_int"
    if op.startswith("v_"): return "v_other"
    if op.startswith("buffer_load") or op.startswith("global_load"): return "vmem load"
    if op.startswith("buffer_store") or op.startswith("global_store"): return "vmem store"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memrealtime") or op.startswith("s_memtime"): return "smem"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


def main():
    path, want = sys.argv[1], sys.argv[2]
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call(["objcopy", "--dump-section", ".hip_fatbin=" + fat, path, os.path.join(d, "x")])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)] + [len(blob)]
        text = None
        for i in range(len(starts) - 1):
            part, co = os.path.join(d, "b%d" % i), os.path.join(d, "c%d.co" % i)
            open(part, "wb").write(blob[starts[i]:starts[i + 1]])
            subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
            if os.path.getsize(co) == 0: continue
            syms = subprocess.run(["nm", co], capture_output=True, text=True).stdout
            hit = [l.split()[-1] for l in syms.splitlines() if want in l and " T " in l]
            if not hit: continue
            text = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + hit[0], co], capture_output=True, text=True).stdout
            print("kernel:", subprocess.run(["c++filt", hit[0]], capture_output=True, text=True).stdout.strip()[:200])
            break
    if text is None:
        sys.exit("kernel not found")
    ins = []   # (address, op, rest)
    for line in text.splitlines():
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m: ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    # backward branches: target = address of next instruction + simm16 * 4
    best = None
    for i, (a, op, rest) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            off = int(rest.split()[0])
            if off >= 32768: off -= 65536
            tgt = a + 4 + 4 * off
            if off < 0 and tgt in addr_index:
                j = addr_index[tgt]
                if best is None or i - j > best[1] - best[0]: best = (j, i)
    j, i = best
    body = ins[j:i + 1]
    print("hot loop: %d instructions (%#x .. %#x), %d source row(s) per iteration" % (len(body), body[0][0], body[-1][0], rows))
    cnt = collections.Counter(classify(op, rest) for _, op, rest in body)
    cycles = 0.0
    for k in sorted(cnt, key=lambda k: -cnt[k]):
        c = COST.get(k)
        t = cnt[k] * c if c else 0.0
        cycles += t
        print("  %-28s %5d per iteration  %7.1f per row%s" % (k, cnt[k], cnt[k] / rows, "   x %.2f cycles = %7.1f per row" % (c, t / rows) if c else ""))
    nvalu = sum(v for k, v in cnt.items() if k in COST)
    half = cnt.get("sdwa", 0) + cnt.get("dpp", 0)
    print("  VALU instructions per row: %.1f; issue cycles per row at 2.05 per instruction: %.1f%s" % (nvalu / rows, cycles / rows,
          "  — the stream contains %d SDWA / DPP instructions: it issues at 4.0 per instruction as a whole (%.1f per row)" % (half, 4.0 * nvalu / rows) if half else ""))
    waits = [(a, rest) for a, op, rest in body if op.startswith("s_waitcnt")]
    print("  s_waitcnt in the loop (%d):" % len(waits), ", ".join(r for _, r in waits))
    nops = [rest for _, op, rest in body if op.startswith("s_nop")]
    print("  s_nop in the loop (%d): %s" % (len(nops), " ".join(nops)))
    # longest run of VALU instructions without a wait, and dependent add chains are not derivable here; print the branch structure
    br = [(a, op, rest.split()[0]) for a, op, rest in body if op.startswith("s_cbranch") or op == "s_branch"]
    print("  branches in the loop (%d)" % len(br))


if __name__ == "__main__":
    main()
