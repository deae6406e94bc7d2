#!/usr/bin/env python3
"""Build guard (run by __graft_entry__.build()): what the search kernels assume about their own code generation, checked on
the built objects.
  1. No device function is left out of line in the tree / step / space translation units.  The search core passes references
     to the wave's LDS block and to caller-private structs (RolloutOut); out of line they become generic pointers, every LDS
     access a flat instruction, and the build has twice run wrong that way (DESIGN.md "Two faults, one cause": hipcc 7.2 outlining
     rollout_agent<RamseySpace<5>> in k_async -> memory aperture violation; an out-of-line DenseSpace::add_actions -> wrong trees,
     reproducible with make VARIANT=... and __attribute__((noinline))).  Everything on that path is __forceinline__; this check
     keeps it so.
  2. No kernel's private segment (spills + private arrays) exceeds SCRATCH_BUDGET bytes per lane, none uses a dynamic stack.
    python tools/check_kernels.py [objects ...]      exit status 1 on a violation"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
SCRATCH_BUDGET = 512  # bytes per lane; the largest today is 368 (k_pool_search<DenseSpace<16>>)
TREE_TUS = ("tree_kernels", "async_kernels", "pool_kernels", "ramsey_kernels", "ramsey_async_kernels", "ramsey_pool_kernels", "dense_kernels")


def code_object(path, tmp):
    fat, co = os.path.join(tmp, os.path.basename(path) + ".fat"), os.path.join(tmp, os.path.basename(path) + ".co")
    if subprocess.run([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, path], capture_output=True).returncode != 0:
        return None
    r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fat, "--output=" + co], capture_output=True)
    return co if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) else None


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "azdopt_amd", "csrc", "build", "*.o")))
    bad = []
    n_kernels = 0
    seen = set()
    with tempfile.TemporaryDirectory() as tmp:
        for f in files:
            base = os.path.basename(f)[:-2]
            co = code_object(f, tmp)
            if not co:
                # host-only objects carry no device code; a tree / step / space unit that cannot be read is a FAILED check, not a pass
                if base in TREE_TUS:
                    bad.append("%s: no gfx950 code object could be extracted (llvm-objcopy / clang-offload-bundler missing, or another offload arch): NOT CHECKED" % base)
                continue
            seen.add(base)
            syms = subprocess.run([LLVM + "/llvm-readelf", "-sW", co], capture_output=True, text=True).stdout.splitlines()
            funcs = {l.split()[-1] for l in syms if " FUNC " in l}
            kernels = {l.split()[-1][:-3] for l in syms if l.rstrip().endswith(".kd")}
            if base in TREE_TUS:
                for fn in sorted(funcs - kernels):
                    bad.append("%s: device function left out of line: %s" % (base, demangle(fn)[:160]))
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s*- \.agpr_count:", notes)[1:]:
                name = (re.search(r"\.name:\s*(\S+)", blk) or [None, "?"])[1]
                scratch = int((re.search(r"\.private_segment_fixed_size:\s*(\d+)", blk) or [None, "0"])[1])
                dyn = (re.search(r"\.uses_dynamic_stack:\s*(\S+)", blk) or [None, "false"])[1]
                n_kernels += 1
                if scratch > SCRATCH_BUDGET:
                    bad.append("%s: private segment %d B > %d: %s" % (base, scratch, SCRATCH_BUDGET, demangle(name)[:140]))
                if dyn == "true":
                    bad.append("%s: dynamic stack: %s" % (base, demangle(name)[:140]))
    if not sys.argv[1:]:  # the default run is the build's guard: every unit it exists for must have been looked at
        for tu in TREE_TUS:
            if tu not in seen and not any(tu in b for b in bad):
                bad.append("%s: object missing from the build directory: NOT CHECKED" % tu)
    if n_kernels == 0:
        bad.append("no kernel found in %d object(s): nothing was checked" % len(files))
    for b in bad:
        print("check_kernels:", b)
    print("check_kernels: %d kernels in %d objects, %d violation(s)" % (n_kernels, len(files), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
