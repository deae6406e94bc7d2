"""Copy the summaries that tools/refresh_profiles.sh left under gpurun_out/final2 into profiles/
(newest run files only) and recompute profiles/traffic.json's k_async entry.  Prints the numbers the
docs quote."""
import csv
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "final2")
P = os.path.join(ROOT, "profiles")


def newest(pattern):
    return sorted(glob.glob(os.path.join(O, pattern), recursive=True), key=os.path.getmtime)[-1]


def total(path, counter):
    t, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "k_async" in r["Kernel_Name"]:
            t += float(r["Counter_Value"])
            n += 1
    return t, n


ff, fw = newest("pmc_fetch/**/*counter_collection.csv"), newest("pmc_write/**/*counter_collection.csv")
fetch, n1 = total(ff, "FETCH_SIZE")
write, n2 = total(fw, "WRITE_SIZE")
assert n1 == n2 and n1 > 0
calls = 800 * n1
raw = (fetch + write) * 1024 / calls
corr = (2 * fetch + write) * 1024 / calls
t = json.load(open(os.path.join(P, "traffic.json")))
t["k_async"].update({"FETCH_SIZE_KB_total": fetch, "WRITE_SIZE_KB_total": write, "calls": calls,
                     "hbm_bytes_per_call_raw": raw, "hbm_bytes_per_call": corr})
t["k_async_hbm_bytes_per_call"] = corr
json.dump(t, open(os.path.join(P, "traffic.json"), "w"), indent=1)
shutil.copy(ff, os.path.join(P, "r01_async_pmc_fetch_size.csv"))
shutil.copy(fw, os.path.join(P, "r01_async_pmc_write_size.csv"))
ks = newest("prof/**/*kernel_stats.csv")
shutil.copy(ks, os.path.join(P, "r01_async_kernel_stats.csv"))
shutil.copy(os.path.join(O, "tile_probe.txt"), os.path.join(P, "r01_tile_probe.txt"))
with open(os.path.join(P, "r01_bench_lines.txt"), "w") as f:
    for name in ("default", "barrier", "bf16", "r333", "r333_bf16", "r44"):
        line = open(os.path.join(O, f"bench_{name}.log")).read().strip().splitlines()[-1]
        f.write(f"### bench_{name}\n{line}\n")
        d = json.loads(line)
        print(f"{name:10s} {d['value'] / 1e6:6.2f} M  {d['ms_per_step'] * 1e3:6.1f} us/call  roofline {d['roofline']['achieved']:.1f} GB/s "
              f"frac {d['roofline']['frac']:.4f}  launch {d['roofline']['avg_launch_ms']:.2f} ms  cpu {(d.get('cpu_baseline') or {}).get('value')}")


def clean(p):
    return "".join(l for l in open(os.path.join(O, p)) if "amdgpu.ids" not in l)


txt = open(os.path.join(P, "r01_phase_profile.txt")).read()
out = []
for sec in re.split(r"(?m)^(?=### )", txt):
    h = sec.split("\n", 1)[0]
    if h.startswith("### python tools/phase_profile.py 4096 800 mlp async bf16"):
        sec = h + "\n" + clean("phase_async_bf16.txt") + "\n"
    elif h.startswith("### python tools/phase_profile.py 4096 800 mlp async"):
        sec = h + "\n" + clean("phase_async.txt") + "\n"
    elif h.startswith("### python tools/agent_balance.py async"):
        sec = h + "\n" + clean("balance_async.txt") + "\n"
    elif h.startswith("### python tools/slow_agents.py"):
        sec = h + "\n" + clean("slow_agents.txt") + "\n"
    out.append(sec)
open(os.path.join(P, "r01_phase_profile.txt"), "w").write("".join(out))
for row in csv.DictReader(open(ks)):
    if "k_async" in row["Name"]:
        print("rocprofv3 k_async: launches", row["Calls"], "avg ms", float(row["AverageNs"]) / 1e6)
m = re.search(r'\{"metric.*', open(os.path.join(O, "prof.log")).read())
print("bench HIP-event avg in the same run:", json.loads(m.group(0))["roofline"]["avg_launch_ms"])
print("traffic per call: raw %.1f MB corrected %.1f MB" % (raw / 1e6, corr / 1e6))
