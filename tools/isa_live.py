#!/usr/bin/env python3
"""Which vector registers stay live across a region of a kernel's ISA (linear scan, control flow ignored):
python tools/isa_live.py OBJ KERNEL_SUBSTRING [--begin PATTERN] [--end PATTERN]
Default region: from the first to the last s_set_gpr_idx_on (the register lambda_1 fold)."""
import re, subprocess, sys, tempfile, os
LLVM = "/opt/rocm/lib/llvm/bin"

def disasm(obj):
    tmp = tempfile.mkdtemp()
    fat, co = os.path.join(tmp, "f.fat"), os.path.join(tmp, "f.co")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
    return subprocess.run([LLVM + "/llvm-objdump", "-d", co], capture_output=True, text=True).stdout

def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3): out.append(int(m.group(3)))
        else: out += list(range(int(m.group(1)), int(m.group(2)) + 1))
    return out

def main():
    obj, pat = sys.argv[1], sys.argv[2]
    beg = sys.argv[sys.argv.index("--begin") + 1] if "--begin" in sys.argv else "s_set_gpr_idx_on"
    end = sys.argv[sys.argv.index("--end") + 1] if "--end" in sys.argv else beg
    txt = disasm(obj)
    fn = None
    lines = []
    for ln in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:", ln)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            fn = name if pat in name else None
            if fn and lines: break
            continue
        if fn and ln.strip(): lines.append(ln.split("//")[0].strip())
    idx = [i for i, l in enumerate(lines) if beg in l]
    idx2 = [i for i, l in enumerate(lines) if end in l]
    b, e = idx[0], idx2[-1]
    print("kernel lines", len(lines), "region", b, e)
    def rw(l):
        parts = l.split(None, 1)
        if len(parts) < 2: return [], []
        op, args = parts
        a = [x.strip() for x in args.split(",")]
        store = op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "flat_store", "s_", "v_cmp", "ds_bpermute")) and not op.startswith("v_cmpx")
        if op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "flat_store", "v_cmp_", "s_")):
            return [], sum((regs(x) for x in a), [])
        return regs(a[0]), sum((regs(x) for x in a[1:]), [])
    inside_w, inside_any = set(), set()
    for l in lines[b:e + 1]:
        w, r = rw(l)
        inside_w |= set(w); inside_any |= set(w) | set(r)
    before = set()
    for l in lines[:b]:
        w, r = rw(l); before |= set(w)
    live = set()
    seen = set()
    for l in lines[e + 1:]:
        w, r = rw(l)
        for x in r:
            if x not in seen and x in before and x not in inside_w: live.add(x)
        seen |= set(r) | set(w)
    print("written before, untouched inside, read after before rewritten:", len(live - inside_any))
    print(sorted(live - inside_any))
    print("regs used inside:", len(inside_any))

main()
