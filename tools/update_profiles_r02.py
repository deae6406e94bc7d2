"""Copy the summaries tools/refresh_profiles_r02.sh (and tools/gpu/calib.sh) left under gpurun_out/final_r02 into profiles/
and add the k_pool entry of profiles/traffic.json.  Prints the numbers the docs quote."""
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "final_r02")
P = os.path.join(ROOT, "profiles")


def newest(pattern):
    return sorted(glob.glob(os.path.join(O, pattern), recursive=True), key=os.path.getmtime)[-1]


def total(path, counter, needle):
    t, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
            t += float(r["Counter_Value"])
            n += 1
    return t, n


def engine_rows(src, dst):  # keep the engine's kernels only (the files also hold torch / runtime kernels)
    rows = list(csv.DictReader(open(src)))
    keep = [r for r in rows if "azd::" in r["Kernel_Name"]]
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)


ff, fw = newest("pmc_fetch/**/*counter_collection.csv"), newest("pmc_write/**/*counter_collection.csv")
fetch, n1 = total(ff, "FETCH_SIZE", "k_pool")
write, n2 = total(fw, "WRITE_SIZE", "k_pool")
assert n1 == n2 and n1 > 0
calls = 800 * n1
raw = (fetch + write) * 1024 / calls
upper = (2 * fetch + write) * 1024 / calls
t = json.load(open(os.path.join(P, "traffic.json")))
t["k_pool"] = {
    "kernel": "k_pool<3> (default step from round 2: 88 evaluator + 168 searcher workgroups; 4096 agents, default bench.py: three launches of 800 calls)",
    "FETCH_SIZE_KB_total": fetch, "WRITE_SIZE_KB_total": write, "calls": calls,
    "hbm_bytes_per_call": raw,
    "hbm_bytes_per_call_if_every_request_were_128B": upper,
    "calibration": "profiles/r02_gather_calib.txt (tools/probes/gather_calib.hip): FETCH_SIZE = 0.500 x the bytes of a coalesced 16-B-per-lane stream "
                   "(the guide's x2 case), but 3.99 x the bytes of random 16-B record reads and 2.00 x those of random 32-B record reads, i.e. 64 B "
                   "per missed record: the counter tallies one 64-B request per gather miss.  k_pool's reads from memory are record gathers "
                   "(the evaluator's wide weight stream hits L2), so FETCH_SIZE is taken as it is; WRITE_SIZE is exact.",
    "commands": [
        "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/final_r02/pmc_fetch -- python3 bench.py --no-cpu-baseline",
        "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/final_r02/pmc_write -- python3 bench.py --no-cpu-baseline",
    ],
}
t["k_pool_hbm_bytes_per_call"] = raw
t["source"] = "r02_pool_pmc_fetch_size.csv + r02_pool_pmc_write_size.csv"
json.dump(t, open(os.path.join(P, "traffic.json"), "w"), indent=1)
engine_rows(ff, os.path.join(P, "r02_pool_pmc_fetch_size.csv"))
engine_rows(fw, os.path.join(P, "r02_pool_pmc_write_size.csv"))
shutil.copy(newest("prof/**/*kernel_stats.csv"), os.path.join(P, "r02_pool_kernel_stats.csv"))
shutil.copy(newest("prof_E/**/*kernel_stats.csv"), os.path.join(P, "r02_configE_kernel_stats.csv"))
shutil.copy(os.path.join(O, "gather_calib.txt"), os.path.join(P, "r02_gather_calib.txt"))
with open(os.path.join(P, "r02_bench_lines.txt"), "w") as f:
    for name in ("default", "driver20", "B_async", "A", "A_async", "C", "C_async", "D", "D_async", "B8192", "B8192_async", "E"):
        line = [x for x in open(os.path.join(O, f"bench_{name}.log")).read().splitlines() if x.startswith('{"metric')][-1]
        f.write(f"### bench_{name}\n{line}\n")
        d = json.loads(line)
        print(f"{name:12s} {d['value'] / 1e6:6.2f} M  {d['ms_per_step'] * 1e3:7.1f} us/call  {d['step_form']:14s} frac {d['roofline']['frac']:.4f}  launch {d['roofline']['avg_launch_ms']:.2f} ms")
txt = lambda p: "".join(x for x in open(os.path.join(O, p)) if "amdgpu.ids" not in x)  # noqa: E731
open(os.path.join(P, "r02_pool_probe.txt"), "w").write(
    "# tools/pool_probe.py [agents] [calls] [dtype] with AZD_POOL_EVAL_WGS = 16 / 48 / 88 (4096 agents f32), 80 (8192 f32), 56 (8192 bf16), then the\n"
    "# asynchronous step, then tools/pool_probe_hash.py (fixed prediction stream: pure search, no evaluator) for pool and async at 4096 / 8192 / 16384 agents\n"
    + txt("pool_probe.txt"))
open(os.path.join(P, "r02_gemm.txt"), "w").write("# tools/time_gemm.py: the evaluator's batched forward (write_predictions_dev), bf16 MFMA GEMM vs the f32 GEMM\n" + txt("gemm.txt"))
for ks, needle in ((newest("prof/**/*kernel_stats.csv"), "k_pool"), (newest("prof_E/**/*kernel_stats.csv"), "k_gemm_bf16"), (newest("prof_E/**/*kernel_stats.csv"), "k_rollout")):
    for row in csv.DictReader(open(ks)):
        if needle in row["Name"]:
            print("rocprofv3", needle, "launches", row["Calls"], "avg ms", float(row["AverageNs"]) / 1e6)
m = [x for x in open(os.path.join(O, "prof.log")).read().splitlines() if x.startswith('{"metric')][-1]
print("bench HIP-event avg in the same run:", json.loads(m)["roofline"]["avg_launch_ms"])
print("k_pool traffic per call: %.1f MB (FETCH %.1f + WRITE %.1f); upper bound with 128-B requests %.1f MB" % (raw / 1e6, fetch * 1024 / calls / 1e6, write * 1024 / calls / 1e6, upper / 1e6))
