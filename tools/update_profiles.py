"""Copy the summaries tools/refresh_profiles.sh left under gpurun_out/final_<round> into profiles/<round>_* and refresh the k_pool
entry of profiles/traffic.json.  Prints the numbers the docs quote.   python tools/update_profiles.py [r03]"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r05"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "final_" + R)
P = os.path.join(ROOT, "profiles")


def newest(pattern):
    hits = sorted(glob.glob(os.path.join(O, pattern), recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None


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


def txt(p):
    return "".join(x for x in open(os.path.join(O, p)) if "amdgpu.ids" not in x)


def bench_line(name):
    path = os.path.join(O, f"bench_{name}.log")
    if not os.path.exists(path):
        return None
    lines = [x for x in open(path).read().splitlines() if x.startswith('{"metric')]
    return lines[-1] if lines else None


ff, fw = newest("pmc_fetch/**/*counter_collection.csv"), newest("pmc_write/**/*counter_collection.csv")
if ff and fw:
    fetch, n1 = total(ff, "FETCH_SIZE", "k_pool")
    write, n2 = total(fw, "WRITE_SIZE", "k_pool")
    assert n1 == n2 and n1 > 0
    calls = 800 * n1
    raw = (fetch + write) * 1024 / calls
    t = json.load(open(os.path.join(P, "traffic.json")))
    t["k_pool"] = {
        "kernel": "k_pool<3> (default step; %s: express lane, measured split; 4096 agents, default bench.py: three launches of 800 calls)" % R,
        "FETCH_SIZE_KB_total": fetch, "WRITE_SIZE_KB_total": write, "calls": calls, "hbm_bytes_per_call": raw,
        "hbm_bytes_per_call_if_every_request_were_128B": (2 * fetch + write) * 1024 / calls,
        "calibration": "profiles/r02_gather_calib.txt (tools/probes/gather_calib.hip): FETCH_SIZE = 0.500 x the bytes of a coalesced 16-B-per-lane stream "
                       "(the guide's x2 case), but 3.99 x the bytes of random 16-B record reads and 2.00 x those of random 32-B record reads, i.e. 64 B "
                       "per missed record: the counter tallies one 64-B request per gather miss.  k_pool's reads from memory are record gathers "
                       "(the evaluator's wide weight stream hits L2), so FETCH_SIZE is taken as it is; WRITE_SIZE is exact.",
        "commands": [
            "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/final_%s/pmc_fetch -- python3 bench.py --no-cpu-baseline" % R,
            "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/final_%s/pmc_write -- python3 bench.py --no-cpu-baseline" % R,
        ],
    }
    t["k_pool_hbm_bytes_per_call"] = raw
    t["source"] = "%s_pool_pmc_fetch_size.csv + %s_pool_pmc_write_size.csv" % (R, R)
    json.dump(t, open(os.path.join(P, "traffic.json"), "w"), indent=1)
    engine_rows(ff, os.path.join(P, R + "_pool_pmc_fetch_size.csv"))
    engine_rows(fw, os.path.join(P, R + "_pool_pmc_write_size.csv"))
    print("k_pool traffic per call: %.1f MB (FETCH %.1f + WRITE %.1f)" % (raw / 1e6, fetch * 1024 / calls / 1e6, write * 1024 / calls / 1e6))
# the driver's window: launches of 20 calls (bench.py --steps 20 --warmup 5 runs the timed launch once after an untimed epoch and warm-up launches of the same shape)
f20, w20 = newest("pmc_fetch20/**/*counter_collection.csv"), newest("pmc_write20/**/*counter_collection.csv")
if f20 and w20:
    def launches_of(path, counter):  # per-dispatch totals of k_pool, in dispatch order
        by = {}
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and "k_pool" in r["Kernel_Name"]:
                by[int(r["Dispatch_Id"])] = by.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        return [by[k] for k in sorted(by)]
    fl, wl_ = launches_of(f20, "FETCH_SIZE"), launches_of(w20, "WRITE_SIZE")
    # the 20-call launches are the ones with the smallest write totals (an 800-call epoch writes 40 x as much)
    small_f = sorted(fl)[: max(1, len(fl) // 2)]
    small_w = sorted(wl_)[: max(1, len(wl_) // 2)]
    f_last, w_last = fl[-1], wl_[-1]  # the timed launch is the last dispatch of the kernel
    t = json.load(open(os.path.join(P, "traffic.json")))
    t["k_pool_driver20"] = {
        "kernel": "k_pool<3>, launches of 20 calls mid-epoch (the driver's bench.py --steps 20 --warmup 5), 4096 agents; %s" % R,
        "calls_per_launch": 20, "FETCH_SIZE_KB_timed_launch": f_last, "WRITE_SIZE_KB_timed_launch": w_last,
        "hbm_bytes_per_call": (f_last + w_last) * 1024 / 20.0,
        "all_k_pool_dispatches_FETCH_KB": fl, "all_k_pool_dispatches_WRITE_KB": wl_,
        "commands": ["rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/final_%s/pmc_fetch20 -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5" % R,
                     "rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/final_%s/pmc_write20 -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5" % R],
    }
    json.dump(t, open(os.path.join(P, "traffic.json"), "w"), indent=1)
    engine_rows(f20, os.path.join(P, R + "_pool_driver20_pmc_fetch_size.csv"))
    engine_rows(w20, os.path.join(P, R + "_pool_driver20_pmc_write_size.csv"))
    print("k_pool traffic per call in the driver's 20-call launch: %.1f MB (FETCH %.1f + WRITE %.1f)" % (
        (f_last + w_last) * 1024 / 20 / 1e6, f_last * 1024 / 20 / 1e6, w_last * 1024 / 20 / 1e6))
# the other configurations' dominant kernels (part tccx): per-call traffic into traffic.json["by_config"]
for cfg, needle in (("A", "k_pool<"), ("C", "k_pool<"), ("D", "k_pool<"), ("E", "k_pool_search<")):
    fx, wx = newest("pmc_fetch_%s/**/*counter_collection.csv" % cfg), newest("pmc_write_%s/**/*counter_collection.csv" % cfg)
    lg = os.path.join(O, "pmc_fetch_%s.log" % cfg)
    if not (fx and wx and os.path.exists(lg)):
        continue
    bl = [x for x in open(lg).read().splitlines() if x.startswith('{"metric')]
    if not bl:
        continue
    d = json.loads(bl[-1])
    fetch, n1 = total(fx, "FETCH_SIZE", needle)
    write, n2 = total(wx, "WRITE_SIZE", needle)
    if n1 == 0 or n1 != n2:
        print("tccx %s: dispatch counts differ (%d / %d): skipped" % (cfg, n1, n2))
        continue
    cpl = d.get("calls_per_launch") or 800.0
    calls = cpl * n1
    t = json.load(open(os.path.join(P, "traffic.json")))
    t.setdefault("by_config", {})[cfg] = {
        "kernel": d["roofline"]["kernel"], "step_form": d["step_form"], "agents_per_gpu": d["config"]["agents_total"], "dtype": d["dtype"],
        "dispatches": n1, "calls_per_launch": cpl, "FETCH_SIZE_KB_total": fetch, "WRITE_SIZE_KB_total": write,
        "hbm_bytes_per_call": (fetch + write) * 1024 / calls,
        "algorithmic_bytes_per_call": d["roofline"]["algorithmic_bytes_per_expansion"] * d["expansions"] / d["steps"],
        "commands": ["rocprofv3 --pmc %s --kernel-trace --output-format csv -d gpurun_out/final_%s/pmc_%s_%s -- python3 bench.py --no-cpu-baseline --config %s" % (c, R, c.split("_")[0].lower(), cfg, cfg)
                     for c in ("FETCH_SIZE", "WRITE_SIZE")], "round": R}
    json.dump(t, open(os.path.join(P, "traffic.json"), "w"), indent=1)
    engine_rows(fx, os.path.join(P, "%s_config%s_pmc_fetch_size.csv" % (R, cfg)))
    engine_rows(wx, os.path.join(P, "%s_config%s_pmc_write_size.csv" % (R, cfg)))
    e = t["by_config"][cfg]
    print("config %s %s traffic per call: %.1f MB (FETCH %.1f + WRITE %.1f) against %.1f MB algorithmic = %.2f x" % (
        cfg, e["kernel"], e["hbm_bytes_per_call"] / 1e6, fetch * 1024 / calls / 1e6, write * 1024 / calls / 1e6, e["algorithmic_bytes_per_call"] / 1e6,
        e["hbm_bytes_per_call"] / e["algorithmic_bytes_per_call"]))
l2 = newest("pmc_l2/**/*counter_collection.csv")
if l2:
    tot = {}
    for r in csv.DictReader(open(l2)):
        if "k_pool" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    with open(os.path.join(P, R + "_pool_pmc_l2.txt"), "w") as f:
        f.write("# rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum over bench.py --steps 800 --warmup 800 (k_pool<3>, 4096 agents; all XCDs summed, all dispatches)\n")
        for k in sorted(tot):
            f.write("%s %.0f\n" % (k, tot[k]))
        if tot.get("TCC_HIT_sum", 0) + tot.get("TCC_MISS_sum", 0) > 0:
            f.write("L2 hit rate %.4f\n" % (tot["TCC_HIT_sum"] / (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"])))
    print(open(os.path.join(P, R + "_pool_pmc_l2.txt")).read())
for src, dst in (("prof/**/*kernel_stats.csv", "_pool_kernel_stats.csv"), ("prof20/**/*kernel_stats.csv", "_pool_driver20_kernel_stats.csv"),
                 ("prof_E/**/*kernel_stats.csv", "_configE_kernel_stats.csv")):
    f = newest(src)
    if f:
        shutil.copy(f, os.path.join(P, R + dst))
        for row in csv.DictReader(open(f)):
            if any(k in row["Name"] for k in ("k_pool", "k_gemm16", "k_rollout", "k_gemm_bf16", "k_ext_")):
                print("rocprofv3 %-28s %-60s launches %5s avg ms %.4f" % (dst, row["Name"][:60], row["Calls"], float(row["AverageNs"]) / 1e6))
names = ("default", "driver20", "B_async", "A", "A_classic", "C", "D", "B8192", "E", "E_epochs", "E_per_call", "E612", "E612_epochs")
with open(os.path.join(P, R + "_bench_lines.txt"), "w") as f:
    for name in names:
        line = bench_line(name)
        if not line:
            continue
        f.write(f"### bench_{name}\n{line}\n")
        d = json.loads(line)
        print(f"{name:12s} {d['value'] / 1e6:6.2f} M  {d['ms_per_step'] * 1e3:7.1f} us/call  {d['step_form']:14s} frac {d['roofline']['frac']:.4f}  "
              f"launch {d['roofline']['avg_launch_ms']:.2f} ms  split {d.get('pool_split')}")
    for extra, title in (("curve.txt", "calls per launch, and calls per request inside a run-ahead window (tools/launch_curve.py; the second block of each config: round 2's library, libazdopt_amd_r02.so)"),
                         ("examples.txt", "examples/c21_tree (the reference's driver loop over the C ABI) at stride 1 (answered from a run-ahead window) and 800: 3 epochs x 800 episodes")):
        if os.path.exists(os.path.join(O, extra)):
            f.write("### %s\n%s" % (title, txt(extra)))
for src, dst, head in (("gemm.txt", "_gemm.txt", "# tools/time_gemm16.py (the LDS-DMA bf16 GEMM in isolation, checked against torch) and tools/time_gemm.py (the evaluator's forward;\n# the last line: the round-2 kernel, AZD_GEMM_OLD=1)\n"),
                       ("pool_split.txt", "_pool_split.txt", "# tools/split_util.py (busy shares of the two sides across fixed splits) and tools/split_feedback.py (the engine's feedback)\n")):
    if os.path.exists(os.path.join(O, src)):
        open(os.path.join(P, R + dst), "w").write(head + txt(src))
if os.path.exists(os.path.join(O, "insts.txt")):
    head = ("# %s_pool_insts.txt -- where the searcher's instructions go (round-4 verdict, item 2).  tools/refresh_profiles.sh insts:\n"
            "#  * phase stamps of the diagnostic build (100 MHz wall clock around each phase of rollout_agent, per agent and call with a new node);\n"
            "#  * wave-instructions by class, per kernel, of the pool step (one launch of 400 calls) and of the launch-per-phase form (100 calls:\n"
            "#    k_rollout = the search proper -- selection levels, table probes, new nodes with lambda_1 / matching / row, cascades --, k_add_actions,\n"
            "#    k_gemm = the evaluator as batched launches), with the expansions and events of the counted launches.\n"
            "# Per expansion (divide by EXPANSIONS): see DESIGN.md section 5 / 6a.  PC sampling is not available on this pool (refused), so the split of\n"
            "# k_rollout's instructions by phase is by the stamps' time shares, not by counting.\n" % R)
    open(os.path.join(P, R + "_pool_insts.txt"), "w").write(head + txt("insts.txt"))
sq = []
for d_ in ("pmc_sq1", "pmc_sq2"):
    if os.path.isdir(os.path.join(O, d_)):
        sq.append(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_sq.py"), os.path.join(O, d_), "k_pool"], capture_output=True, text=True).stdout)
if sq:
    open(os.path.join(P, R + "_pool_pmc_sq.txt"), "w").write("# SQ counters of k_pool<3>, two rocprofv3 --pmc passes over bench.py --steps 800 --warmup 800 (tools/refresh_profiles.sh); tools/pmc_sq.py\n" + "".join(sq))
    print("".join(sq))
# wave-instructions of k_pool per call, by class (the second SQ pass): what bench.py's roofline_issue block is computed from
sq2 = newest("pmc_sq2/**/*counter_collection.csv")
if sq2:
    tot, disp = {}, set()
    for r in csv.DictReader(open(sq2)):
        if "k_pool" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    calls = 800 * len(disp)
    issue = {"kernel": "k_pool<3>, 4096 agents, fp32 3 x 256 (%s): rocprofv3 --pmc SQ_INSTS_* over bench.py --no-cpu-baseline --steps 800 --warmup 800, "
                       "%d launches of 800 calls; ALL waves of the kernel (searchers, evaluators, waiting waves' polls)" % (R, len(disp)),
             "calls": calls,
             "wave_insts_per_call": {k.replace("SQ_INSTS_", "").lower(): v / calls for k, v in sorted(tot.items()) if k.startswith("SQ_INSTS_")}}
    json.dump(issue, open(os.path.join(P, "issue.json"), "w"), indent=1)
    print("k_pool wave-instructions per call:", {k: round(v) for k, v in issue["wave_insts_per_call"].items()})
    # how to divide the sums above: they cover EVERY k_pool dispatch of the profiled command (warm-up epoch + timed epoch), not the timed one
    lg = os.path.join(O, "pmc_sq2.log")
    bl = [x for x in open(lg).read().splitlines() if x.startswith('{"metric')] if os.path.exists(lg) else []
    with open(os.path.join(P, R + "_pool_pmc_sq.txt"), "a") as f:
        f.write("# The sums cover %d dispatches of k_pool = %d calls of 4096 agents (the profiled command's warm-up epoch AND its timed epoch).\n" % (len(disp), calls))
        if bl:
            d = json.loads(bl[-1])
            epc = d["expansions"] / d["steps"]
            f.write("# Expansions per call in the timed epoch of that run: %.0f  ->  wave-instructions per expansion: " % epc)
            f.write(", ".join("%s %.0f" % (k.replace("SQ_INSTS_", ""), v / calls / epc) for k, v in sorted(tot.items()) if k.startswith("SQ_INSTS_") and "MFMA" not in k and k != "SQ_WAVES"))
            f.write(" (MFMA ops are counted inside VALU; profiles/issue.json holds the per-call figures bench.py's roofline_issue block uses)\n")
m = [x for x in open(os.path.join(O, "prof.log")).read().splitlines() if x.startswith('{"metric')] if os.path.exists(os.path.join(O, "prof.log")) else []
if m:
    print("bench HIP-event avg in the profiled run:", json.loads(m[-1])["roofline"]["avg_launch_ms"])
