#!/usr/bin/env python3
"""bench.py -- node-expansions/s of the c21 self-play hot path on N MI355X (one process per GPU).

A "step" is one NablaOptimizer::par_roll_out_episodes call over the whole agent population
(select / expand / backup + the MLP forward + add_actions + argmin), executed by the CU-resident
persistent kernel k_async (16 agents per workgroup; an agent waiting for its prediction row serves the
workgroup's evaluator on the matrix cores; --barrier-step selects the lock-step form k_persist).  Every
EPOCH_CALLS steps the epoch boundary of the reference driver (04-c21-tree.rs:163-207:
par_update_model, modify_root policy, par_reset_trees) runs INSIDE the timed region.

Workload (BASELINE.json configs[1]): c21 space N = 19 (STATE 304, ACTION 152), 4096 agents per GPU,
fp32 MLP 304-256-256-256-152 (ReLU x3, Sigmoid), n_as_tol = [200, 50, 50] / 25, n_obs_tol = 200,
800 calls per epoch, seeded synthetic roots.  N > 1: agents shard by global id (weak scaling,
4096 per GPU, no data-path collective); the training triple is all-gathered over RCCL once per epoch
so that every rank takes the identical optimiser step.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HIDDEN = (256, 256, 256)
N_OBS_TOL = 200
EPOCH_CALLS = 800
SEED = 0
# The default workload is BASELINE.json configs[1].  The Ramsey workloads (configs[3]: "Ramsey-style
# multicolour edge space", 8192 agents per GPU = 32768 over 4 GPUs) are measured with --workload;
# they keep the drivers' n_as_tol tables (01-r333.rs:128-130, 02-r44.rs:128-130) and use 800-call epochs
# like c21 (the drivers run 6400 / 3200) so that the arenas stay under 20 GB.
WORKLOADS = {
    "c21": dict(kind="c21", n=19, agents=4096, tol=([200, 50, 50], 25), caps={}, name="c21 N=19"),
    "r333": dict(kind="ramsey", n=16, sizes=[3, 3, 3], agents=8192,
                 tol=([200, 200, 200, 100, 100, 100, 50, 50, 50, 25, 25, 25], 10),
                 caps=dict(prediction_capacity=98304), name="Ramsey R(3,3,3) N=16"),
    "r44": dict(kind="ramsey", n=17, sizes=[4, 4], agents=8192, tol=([200, 200, 100, 100, 50, 50, 25, 25], 10),
                caps=dict(prediction_capacity=57344), name="Ramsey R(4,4) N=17"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


class _DevView:
    """zero-copy torch view of an engine-owned device buffer (via __cuda_array_interface__)"""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr="<f4", data=(int(ptr), False), version=2)


def make_space(az, wl):
    if wl["kind"] == "ramsey":
        return az.RamseySpaceNoEdgeRecolor(wl["n"], wl["sizes"])
    return az.ROTModifyParentsOnce(wl["n"])


def algorithmic_bytes(c0, c1, state_dim, action_dim=None, state_bytes=0):
    """Algorithmic bytes the roll-out kernel moves per expansion, SURVEY.md 8(d) with the measured
    D (select calls), deg, A, K per expansion instead of the nominal ones and this build's record
    sizes (node 32 B, arc 16 B, prediction 16 B):  selection reads + new node/arc/key writes +
    state-vector write + cascade read-modify-write.  (add_actions' bytes are not in this kernel.)"""
    d = {k: c1[k] - c0[k] for k in c1}
    exp = max(1, d["EXPANSIONS"])
    sel_bytes = 32 * d["SELECT_CALLS"] + (16 + 32) * d["SUM_DEG"] + 16 * d["SUM_ACTIONS"]
    new_nodes = d["EXPANSIONS"] + d["TERMINALS"]
    new_arcs = new_nodes + d["TRANSPOSITIONS"]
    write_bytes = new_nodes * (32 + 24 + 4) + new_arcs * (16 + 8)
    vec_bytes = 4 * state_dim * d["EXPANSIONS"]
    cascade_bytes = d["CASCADE_NODES"] * (32 + 12 + 16)
    total = sel_bytes + write_bytes + vec_bytes + cascade_bytes
    # add_actions runs inside the persistent step too: prediction row read + new predictions written
    total += 4 * (action_dim if action_dim else state_dim // 2) * d["EXPANSIONS"] + 16 * d["NEW_PREDS"]
    # per-call load + store of the agent's space state (Ramsey: clique counts + neighbourhoods) and its
    # re-read by add_actions; c21's 32-byte parent row is ignored
    total += 3 * state_bytes * d["EXPANSIONS"]
    return total / exp, d


def cpu_baseline(n_threads, steps, wl, krange):
    """CPU restatement of the reference algorithm (oracle, OpenMP over agents like rayon's par_iter;
    CPU fp32 MLP) on a bounded sample of the same workload: the first `steps` calls."""
    from oracle import orc
    B, n, TOL = wl["agents"], wl["n"], wl["tol"]
    if wl["kind"] == "ramsey":
        e = orc.Engine(n, B, threads=n_threads, ramsey=(wl["sizes"], [1.0] * len(wl["sizes"])))
        parents, permitted = orc.gen_ramsey_roots(SEED, 0, 0, B, n, len(wl["sizes"]), *krange)
    else:
        e = orc.Engine(n, B, threads=n_threads)
        parents, permitted = orc.gen_roots(SEED, 0, 0, B, n, *krange)
    dims = (e.S,) + HIDDEN + (e.A,)
    mlp = orc.Mlp(dims, seed=SEED, threads=n_threads)
    e.new_begin(parents, permitted)
    e.new_end(mlp.forward(e.state_vecs()))
    t0 = time.perf_counter()
    for _ in range(steps):
        e.rollout_begin(*TOL)
        e.rollout_end(mlp.forward(e.state_vecs()))
    dt = time.perf_counter() - t0
    exp = e.counters()["EXPANSIONS"]
    return dict(value=exp / dt, unit="expansions/s", cores=n_threads, kind="port",
                sample=f"first {steps} calls of the same workload (B={B}, N={n}, MLP on CPU), {dt:.1f} s; "
                       "CPU restatement of the reference algorithm, not the Rust binary (cannot be built here)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1600)
    ap.add_argument("--warmup", type=int, default=800)
    ap.add_argument("--cpu-steps", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--chunk", type=int, default=800, help="calls per host round trip (<= one epoch)")
    ap.add_argument("--barrier-step", action="store_true", help="lock-step form (k_persist) instead of the default asynchronous step (k_async)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c21")
    ap.add_argument("--agents", type=int, default=0, help="agents per GPU (default: the workload's)")
    ap.add_argument("--mlp-dtype", choices=["f32", "bf16"], default="f32",
                    help="evaluator weight/activation storage for inference (bf16 = BASELINE configs[2]; f32 accumulate either way)")
    args = ap.parse_args()
    wl = dict(WORKLOADS[args.workload])
    if args.agents > 0:
        wl["agents"] = args.agents
    AGENTS_PER_GPU, TOL = wl["agents"], wl["tol"]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import torch
    import torch.distributed as dist
    # rehearsal on a one-GPU box (never used by the driver): AZD_BENCH_REHEARSE=1 puts every rank on
    # cuda:0 and swaps RCCL for gloo with CPU staging, to exercise the N > 1 host logic
    rehearse = os.environ.get("AZD_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    coll_dev = "cpu" if rehearse else f"cuda:{local_rank}"

    import azdopt_amd as az
    from azdopt_amd.parallel import ShardPlan, allgather_training_triple, global_argmin

    plan = ShardPlan(world, rank, AGENTS_PER_GPU)
    space = make_space(az, wl)
    B = plan.local_agents
    B_total = plan.total_agents
    model = az.ActionModel(B_total, space.STATE_DIM, space.ACTION_DIM, hidden=HIDDEN, seed=SEED, device=local_rank,
                           dtype=args.mlp_dtype)
    roots = space.generate_roots(SEED, B, first_agent=plan.first_agent)
    opt = az.NablaOptimizer.par_new(space, roots, model, B, device=local_rank, first_agent=plan.first_agent,
                                    async_step=not args.barrier_step, **wl["caps"])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    calls_done = 0
    epoch = 0
    losses = []

    def epoch_boundary():
        nonlocal epoch
        if world == 1:
            losses.append(opt.par_update_model(N_OBS_TOL))
        else:
            ptrs = opt.observe_dev(N_OBS_TOL)
            views = [torch.as_tensor(_DevView(p, (B, d)), device=f"cuda:{local_rank}")
                     for p, d in zip(ptrs, (space.STATE_DIM, space.ACTION_DIM, space.ACTION_DIM))]
            if rehearse:
                gathered = [g.cuda() for g in allgather_training_triple(dist, torch, [v.cpu() for v in views], world)]
            else:
                # the collective runs on torch-owned copies (20 MB per rank), not on the engine's own allocations
                gathered = allgather_training_triple(dist, torch, [v.clone() for v in views], world)
            torch.cuda.synchronize()
            losses.append(model.update_model_dev(B_total, *[g.data_ptr() for g in gathered], stream=opt.stream()))
        opt.par_reset_trees_policy(SEED, epoch)  # modify_root policy + reset on the device
        epoch += 1

    def run(n_calls):
        nonlocal calls_done
        left = n_calls
        while left > 0:
            k = min(left, args.chunk, EPOCH_CALLS - calls_done % EPOCH_CALLS)
            opt.par_roll_out_episodes(TOL, n_calls=k)
            calls_done += k
            left -= k
            if calls_done % EPOCH_CALLS == 0:
                epoch_boundary()

    run(args.warmup)
    opt.set_timing(True)
    c0 = opt.counters()
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    c1 = opt.counters()
    timing = opt.timing()
    opt.set_timing(False)

    exp_local = c1["EXPANSIONS"] - c0["EXPANSIONS"]
    if world > 1:
        t = torch.tensor([dt, float(exp_local)], dtype=torch.float64, device=coll_dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt_max, exp_total = float(tmax[0]), float(t[1])
    else:
        dt_max, exp_total = dt, float(exp_local)
    am = opt.argmin_data()
    cost = float(sum(am.cost["clique_counts"])) if wl["kind"] == "ramsey" else am.cost["lambda_1"] + len(am.cost["matching"])
    best_eval, best_cost = global_argmin(dist if world > 1 else None, torch, float(am.eval), cost, local_rank, device=coll_dev)

    if rank == 0:
        state_bytes = (space.C * space.E * 4 + 512) if wl["kind"] == "ramsey" else 0
        bytes_per_exp, d = algorithmic_bytes(c0, c1, space.STATE_DIM, space.ACTION_DIM, state_bytes)
        kw = space.KEY_WORDS
        use_async = not args.barrier_step
        dims_txt = "-".join(str(x) for x in (space.STATE_DIM,) + HIDDEN + (space.ACTION_DIM,))
        launches = max(1, timing["rollout_launches"])
        avg_ms = timing["rollout_ms"] / launches
        bytes_per_launch = bytes_per_exp * exp_local / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.workload == "c21":
            try:
                per_call = json.load(open(tpath)).get("k_async_hbm_bytes_per_call" if use_async else "k_persist_hbm_bytes_per_call")
                traffic = per_call * args.steps / launches if per_call else None  # PMC bytes per call x calls per launch
            except Exception:
                traffic = None
        out = {
            "metric": "node_expansions_per_s", "value": exp_total / dt_max, "unit": "expansions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.mlp_dtype, "data": "synthetic",
            "config": {"workload": "%s tree search, %d agents/GPU, %s MLP %s, tol %s/%d, "
                                   "800 calls/epoch incl. update_model+reset_trees"
                                   % (wl["name"], AGENTS_PER_GPU, "fp32" if args.mlp_dtype == "f32" else "bf16-storage", dims_txt,
                                      str(TOL[0]).replace(" ", ""), TOL[1]),
                       "agents_total": B_total, "parallelism": f"agents sharded x{world}"},
            "best_cost_found": best_cost, "best_eval": best_eval,
            "expansions": exp_total, "terminals": d["TERMINALS"], "transpositions": d["TRANSPOSITIONS"],
            "select_calls_per_expansion": d["SELECT_CALLS"] / max(1, d["EXPANSIONS"]),
            "epoch_losses": losses[-3:],
            "calls_per_launch": args.steps / launches,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": ("k_async<%d>" if use_async else "k_persist<%d>") % kw,
                         "algorithmic_bytes_per_expansion": bytes_per_exp, "avg_launch_ms": avg_ms,
                         "mlp_flop_per_launch": 2.0 * sum(a * b for a, b in zip((space.STATE_DIM,) + HIDDEN, HIDDEN + (space.ACTION_DIM,)))
                                                * ((B + 15) // 16 * 16) * args.steps / launches,
                         "note": "latency-bound pointer chasing: the rate target and the 40% roofline target are "
                                 "~3 orders of magnitude apart for this workload (SURVEY.md 8d)"},
        }
        if not args.no_cpu_baseline and world == 1:  # timed beside the GPU run at N = 1 only
            out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 64), args.cpu_steps, wl, space.default_permitted_range())
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
