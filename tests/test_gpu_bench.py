"""bench.py end to end on the GPU at reduced sizes: the contract line (metric, value, roofline, cpu_baseline, step form) of the
default workload inside the driver's short window, and of BASELINE configs[4] on its CU-resident form."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith('{"metric')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_of_the_default_workload_in_the_drivers_window():
    j = bench("--steps", "20", "--warmup", "5", "--agents", "512", "--cpu-seconds", "2")
    assert j["metric"] == "node_expansions_per_s" and j["unit"] == "expansions/s" and j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 5
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert j["value"] > 1e5 and abs(j["ms_per_step"] * 1e-3 * j["value"] - j["expansions"] / 20) < 1e-6 * j["expansions"]
    assert j["step_form"] == "pool" and j["dtype"] == "f32" and "workload" in j["config"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["kernel"].startswith("k_pool<") and r["avg_launch_ms"] > 0
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert j["epoch_boundaries_in_timed_region"] == 0 and j["replicas_identical"] is True
    b = j["best_cost_run"]  # the other half of the metric: the reference driver's loop to lambda_1 + mu < 5.2, beside a frozen-model control
    assert b["goal"] == 5.2 and b["trained"]["epochs"] >= 1 and b["frozen_control"]["epochs"] == b["trained"]["epochs"]
    assert b["trained"]["best_cost"] <= b["trained"]["best_cost_after_first_epoch"]
    w = j["evaluator_weight_stream"]
    assert w["bound"] == "l2->cu" and 0 < w["frac"] < 1 and w["rows_per_batch"] >= 1


def test_bench_line_of_config_e_on_the_pool_searchers():
    j = bench("--config", "E", "--agents", "1024", "--steps", "30", "--warmup", "10", "--no-cpu-baseline")
    assert j["step_form"] == "pool" and j["step_form_reason"] == "" and j["dtype"] == "bf16"
    assert j["pool_split"][0] == 0 and j["pool_split"][1] >= 1  # searcher workgroups only: the evaluator is a stream of GEMM launches
    assert "outside the kernel" in j["evaluator_form"]
    assert j["roofline"]["kernel"] == "k_pool_search<2>" and j["value"] > 1e5
    assert "128" in j["config"]["workload"]  # the slot cap is stated


def test_bench_line_of_config_e_with_the_drivers_slot_range():
    j = bench("--config", "E612", "--agents", "512", "--steps", "20", "--warmup", "10", "--no-cpu-baseline")
    assert j["step_form"] == "pool" and j["roofline"]["kernel"] == "k_pool_search<10>" and j["value"] > 1e4
    assert "612" in j["config"]["workload"]


def test_bench_gpus_2_launches_itself_on_a_one_gpu_box():
    """`python bench.py --gpus 2` with no launcher around it (round-4 verdict, item 3): the two ranks are children of the bench,
    here both on cuda:0 with gloo standing in for RCCL (AZD_BENCH_REHEARSE: a one-GPU box has no second device): one line, world 2,
    the replicas identical after the pooled optimiser step, the population the sum of the shards."""
    env = dict(os.environ, AZD_BENCH_REHEARSE="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "C", "--agents", "384", "--steps", "840", "--warmup", "10",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith('{"metric')]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["world"] == 2 and len(j["per_rank_expansions_per_s"]) == 2 and j["config"]["agents_total"] == 768
    assert j["replicas_identical"] is True and j["epoch_boundaries_in_timed_region"] >= 1 and j["epoch_exchange"]["count"] >= 1
    assert j["dtype"] == "bf16" and j["config"]["baseline_config"] == "C" and "quoted at 8 GPUs" in j["config"]["baseline_config_note"]
    assert j["value"] > 1e5 and j["cpu_baseline"] is None
