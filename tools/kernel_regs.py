#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in an object file or the built library (from the code object's
metadata notes): python tools/kernel_regs.py [azdopt_amd/csrc/build/pool_kernels.o ...] [--grep k_pool]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path, tmp):
    out = os.path.join(tmp, os.path.basename(path) + ".co")
    fat = os.path.join(tmp, os.path.basename(path) + ".fat")
    if subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path], capture_output=True).returncode != 0:
        return None
    r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fat, "--output=" + out], capture_output=True, text=True)
    return out if r.returncode == 0 and os.path.exists(out) and os.path.getsize(out) else None


def kernels(co):
    txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
        yield dict(name=g("name"), vgpr=g("vgpr_count"), agpr=blk.split()[1], sgpr=g("sgpr_count"), spill=g("vgpr_spill_count"),
                   scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"))


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pat = None
    if "--grep" in sys.argv:
        pat = sys.argv[sys.argv.index("--grep") + 1]
        args = [a for a in args if a != pat]
    files = args or sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "azdopt_amd", "csrc", "build", "*.o")))
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            co = code_objects(f, tmp)
            if not co:
                continue
            for k in kernels(co):
                name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip()
                if pat and pat not in name:
                    continue
                print("%-28s vgpr %3s agpr %3s sgpr %3s spill %3s scratch %5s lds %6s  %s" % (
                    os.path.basename(f), k["vgpr"], k["agpr"], k["sgpr"], k["spill"], k["scratch"], k["lds"], name[:110]))


if __name__ == "__main__":
    main()
