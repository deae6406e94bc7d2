#!/usr/bin/env python3
"""bench.py -- node-expansions/s of the c21 self-play hot path on N MI355X (one process per GPU).

A "step" is one NablaOptimizer::par_roll_out_episodes call over the whole agent population
(select / expand / backup + the MLP forward + add_actions + argmin).  From 256 agents the engine runs it as the
pool step k_pool (pool_step.inc): searcher workgroups of 16 independent waves pull ready agents from per-XCD
queues, evaluator workgroups on CUs of their own pull batches of posted rows and run the MLP on the matrix
cores; below that as k_async (16 agents per workgroup, a waiting agent serves the workgroup's evaluator).
--step async / barrier force the other CU-resident forms.  Every EPOCH_CALLS steps the epoch boundary of the
reference driver (04-c21-tree.rs:163-207: par_update_model, modify_root policy, par_reset_trees) runs INSIDE
the timed region.

Default workload (BASELINE.json configs[1] = --config B): c21 space N = 19 (STATE 304, ACTION 152), 4096 agents
per GPU, fp32 MLP 304-256-256-256-152 (ReLU x3, Sigmoid), n_as_tol = [200, 50, 50] / 25, n_obs_tol = 200,
800 calls per epoch, seeded synthetic roots.  --config A / C / D select the other BASELINE configs that are
built (A: the reference's own run shape, B = 512 with the 512-1024-512 MLP; C: 8192 agents per GPU, bf16
evaluator storage; D: the Ramsey space, 8192 agents per GPU).  N > 1: agents shard by global id (weak
scaling, no data-path collective); the training triple is all-gathered over RCCL once per epoch so that
every rank takes the identical optimiser step (azdopt_amd.parallel.ShardedOptimizer).
Launching: under torch.distributed.run (RANK / WORLD_SIZE set) a process IS one rank; `python bench.py --gpus N` on its own
starts its N ranks itself, as child processes, before it has made any GPU call (self_launch) and passes rank 0's line through.
`config.baseline_config_note` says which BASELINE.json configuration the run is (C is quoted at 8 GPUs, D at 4).

What the timed window holds is reported, not assumed: `config.workload` is built from what ran, and
`epoch_boundaries_in_timed_region` counts the par_update_model + root policy + par_reset_trees sequences
inside it.  A window shorter than an epoch (the driver's --steps 20) is placed in the MIDDLE of the second
epoch -- after one whole untimed epoch including its optimiser step, on trees of half-epoch size -- instead of
on the freshly reset trees of call 0, and says so.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

N_OBS_TOL = 200
EPOCH_CALLS = 800
SEED = 0
# The default workload is BASELINE.json configs[1].  The Ramsey workloads (configs[3]: "Ramsey-style
# multicolour edge space", 8192 agents per GPU = 32768 over 4 GPUs) are measured with --workload;
# they keep the drivers' n_as_tol tables (01-r333.rs:128-130, 02-r44.rs:128-130) and use 800-call epochs
# like c21 (the drivers run 6400 / 3200) so that the arenas stay under 20 GB.
WORKLOADS = {
    # (prediction arena: an epoch of 800 calls makes at most 801 nodes of at most ACTION / 2 = 76 actions each; the engine's default
    # of 32768 holds the first epochs' trees and overflows -- loudly -- once the trained model sends the searches deeper)
    "c21": dict(kind="c21", n=19, agents=4096, hidden=(256, 256, 256), tol=([200, 50, 50], 25), caps=dict(prediction_capacity=65536), name="c21 N=19"),
    # the reference's own run (04-c21-tree.rs:33-54): B = 512, MLP 304-512-1024-512-152
    "c21ref": dict(kind="c21", n=19, agents=512, hidden=(512, 1024, 512), tol=([200, 50, 50], 25), caps=dict(prediction_capacity=65536), name="c21 N=19"),
    "r333": dict(kind="ramsey", n=16, sizes=[3, 3, 3], agents=8192, hidden=(256, 256, 256),
                 tol=([200, 200, 200, 100, 100, 100, 50, 50, 50, 25, 25, 25], 10),
                 caps=dict(prediction_capacity=98304), name="Ramsey R(3,3,3) N=16"),
    "r44": dict(kind="ramsey", n=17, sizes=[4, 4], agents=8192, hidden=(256, 256, 256),
                tol=([200, 200, 100, 100, 50, 50, 25, 25], 10),
                caps=dict(prediction_capacity=57344), name="Ramsey R(4,4) N=17"),
    # BASELINE configs[4]: the build-defined dense-graph space (oracle/dense_graph.inc), N = 50, G(50, 0.1) roots, 512-wide
    # model, on the pool step with the evaluator's GEMM launches beside the searchers.  max_slots = the most
    # modifiable edge slots a root brings (= predictions a node holds): config E takes 128 (a tree's prediction arena is 2 MB
    # then), E612 the drivers' image of E // 2 = 612 slots (the c21 drivers permit up to half of the action space: 04-c21-tree.rs:85)
    "dense50": dict(kind="dense", n=50, p=0.1, agents=8192, hidden=(512, 512, 512), tol=([200, 50, 50], 25), max_slots=128,
                    caps=dict(prediction_capacity=131072), name="dense graphs N=50"),
    "dense50x612": dict(kind="dense", n=50, p=0.1, agents=8192, hidden=(512, 512, 512), tol=([200, 50, 50], 25), max_slots=612,
                        caps=dict(prediction_capacity=524288), name="dense graphs N=50"),
}
# BASELINE.json configs[0..3] as presets: (workload, agents per GPU, evaluator storage)
CONFIGS = {
    "A": ("c21ref", 512, "f32"),   # configs[0]: the reference's plumbing shape (its BATCH is 512; "64" is a plot title)
    "B": ("c21", 4096, "f32"),     # configs[1]: the metric's configuration on one GPU
    "C": ("c21", 8192, "bf16"),    # configs[2]: 65536 agents over 8 GPUs, bf16 MLP, RCCL all-gather
    "D": ("r44", 8192, "f32"),     # configs[3]: 32768 agents over 4 GPUs, Ramsey space
    "E": ("dense50", 8192, "bf16"),  # configs[4]: N = 50 dense graphs, 512-wide bf16 MLP on MFMA
    "E612": ("dense50x612", 8192, "bf16"),  # the same with roots of up to E // 2 = 612 modifiable slots
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_space(az, wl):
    if wl["kind"] == "dense":
        return az.DenseGraphSpace(wl["n"], wl["p"], max_slots=wl.get("max_slots", 128))
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


def _cpu_run(orc, wl, hidden, B, n_threads, krange, max_calls, budget_s, with_mlp):
    """first calls of the workload on the CPU restatement at population B; returns (expansions/s, calls, seconds).
    Only the restatement's own work is timed (tree phases, and the MLP forward when with_mlp): generating the
    fixed prediction stream of the tree-only run is not."""
    n, TOL = wl["n"], wl["tol"]
    if wl["kind"] == "dense":
        e = orc.Engine(n, B, threads=n_threads, dense=True)
        adj, permitted = orc.gen_dense_roots(SEED, 0, 0, B, n, *krange, p=wl["p"])
        parents = adj.view(np.uint8).reshape(B, -1)
    elif wl["kind"] == "ramsey":
        e = orc.Engine(n, B, threads=n_threads, ramsey=(wl["sizes"], [1.0] * len(wl["sizes"])))
        parents, permitted = orc.gen_ramsey_roots(SEED, 0, 0, B, n, len(wl["sizes"]), *krange)
    else:
        e = orc.Engine(n, B, threads=n_threads)
        parents, permitted = orc.gen_roots(SEED, 0, 0, B, n, *krange)
    mlp = orc.Mlp((e.S,) + tuple(hidden) + (e.A,), seed=SEED, threads=n_threads) if with_mlp else None
    e.new_begin(parents, permitted)
    e.new_end(mlp.forward_fast(e.state_vecs()) if with_mlp else orc.hash_predictions(SEED, 0, B, e.A, 0))
    busy, calls = 0.0, 0
    while calls < max_calls and busy < budget_s:
        t0 = time.perf_counter()
        e.rollout_begin(*TOL)
        t1 = time.perf_counter()
        h = mlp.forward_fast(e.state_vecs()) if with_mlp else orc.hash_predictions(SEED, 0, B, e.A, calls + 1)
        t2 = time.perf_counter()
        e.rollout_end(h)
        t3 = time.perf_counter()
        busy += (t3 - t0) if with_mlp else (t1 - t0) + (t3 - t2)  # the model call is part of the end-to-end time only
        calls += 1
    return e.counters()["EXPANSIONS"] / busy, calls, busy


def cpu_baseline(n_threads, wl, hidden, krange, budget_s=24.0):
    """CPU restatement of the reference algorithm (oracle; OpenMP over agents = rayon's par_iter, barrier,
    model call, barrier: optimizer/mod.rs:159-189) on bounded samples of the same workload, BASELINE.md 3:
    tree-only (fixed prediction stream) and end-to-end (blocked AVX2 fp32 MLP on the host cores) at
    B = 64, 512 and the GPU run's population."""
    from oracle import orc
    pops = sorted({64, 512, wl["agents"]})
    per = budget_s / (2 * len(pops))
    rows = {"tree_only": {}, "end_to_end": {}}
    for B in pops:
        for key, with_mlp in (("tree_only", False), ("end_to_end", True)):
            rate, calls, sec = _cpu_run(orc, wl, hidden, B, n_threads, krange, 800, per, with_mlp)
            rows[key][str(B)] = dict(value=rate, unit="expansions/s", agents=B, calls=calls, seconds=round(sec, 2))
    top = rows["end_to_end"][str(wl["agents"])]
    return dict(value=top["value"], unit="expansions/s", cores=n_threads, kind="port",
                sample="first %d calls of the same workload (B=%d, N=%d, fp32 MLP %s on the CPU, blocked AVX2 loop nest), %.1f s; "
                       "CPU restatement of the reference algorithm, not the Rust binary (cannot be built here)"
                       % (top["calls"], wl["agents"], wl["n"], "-".join(map(str, hidden)), top["seconds"]),
                tree_only=rows["tree_only"], end_to_end=rows["end_to_end"])


def best_cost_run(az, wl, hidden, B, mlp_dtype, max_epochs, goal=5.2):
    """The other half of the metric (BASELINE.json: best-cost-found), outside the timed region: the reference driver's loop
    (graph-state/examples/04-c21-tree.rs:133-208 -- 800 episodes, one optimiser step with Adam lr 1e-4 / L2 1e-6, the root
    policy) on a fresh engine until lambda_1 + mu < 5.2 (:117,125) or max_epochs; then the same loop with the model never
    updated, for as many epochs (control: what the search and the root policy find without learning).  Trajectories over 250
    epochs, a CPU-restatement arm and an lr x 100 arm: profiles/r04_best_cost.txt (tools/best_cost.py)."""
    space = az.ROTModifyParentsOnce(wl["n"])
    kmin, kmax = space.default_permitted_range()
    out = {"goal": goal, "max_epochs": max_epochs, "episodes_per_epoch": 800}
    t0 = time.perf_counter()
    epochs_trained = max_epochs
    for arm in ("trained", "frozen_control"):
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=hidden, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, l2=1e-6, seed=SEED, dtype=mlp_dtype)
        opt = az.NablaOptimizer.par_new(space, space.generate_roots(SEED, B, kmin=kmin, kmax=kmax), model, B, **wl["caps"])
        best, first, ep = None, None, 0
        for ep in range(1, (max_epochs if arm == "trained" else epochs_trained) + 1):
            opt.par_roll_out_episodes(wl["tol"], n_calls=800)
            am = opt.argmin_data()
            best = len(am.cost["matching"]) + am.cost["lambda_1"]
            first = best if first is None else first
            if best < goal:
                break
            if arm == "trained":
                opt.par_update_model(200)
            opt.par_reset_trees_policy(SEED, ep, kmin, kmax)
        if arm == "trained":
            epochs_trained = ep
        out[arm] = {"epochs": ep, "best_cost": best, "best_cost_after_first_epoch": first, "reached_goal": bool(best < goal)}
        del opt, model
    out["seconds"] = time.perf_counter() - t0
    return out


# BASELINE.json configs[] are quoted at fixed GPU counts: which of them a run IS, and which it only shares the per-GPU shape with
BASELINE_GPUS = {"A": (0, None), "B": (1, 1), "C": (2, 8), "D": (3, 4), "E": (4, 8), "E612": (4, 8)}


def baseline_note(cfg, world, agents_total):
    if cfg is None:
        return "not a BASELINE.json configuration (custom workload / agents / storage)"
    idx, gpus = BASELINE_GPUS[cfg]
    if gpus is None:
        return "BASELINE.json configs[0]'s shape (the reference's own run, 512 agents) on %d GPU(s)" % world
    if world == gpus:
        return "BASELINE.json configs[%d] exactly: %d agents over %d GPU(s)" % (idx, agents_total, world)
    return ("the per-GPU shape of BASELINE.json configs[%d] (quoted at %d GPUs) on %d GPU(s): %d agents in all -- the driver's 1/2/4/8 curve "
            "keeps ONE per-GPU workload (weak scaling); `--gpus %d --config %s` is that configuration itself" % (idx, gpus, world, agents_total, gpus, cfg))


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks as child processes of this one --
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set as torch.distributed.run sets them -- BEFORE this process has
    made any GPU call (it never does: a process that has touched the GPU must not be replaced or forked on this pool), pass
    rank 0's stdout through (the one JSON line), and exit with the first non-zero status of any rank (the others are stopped:
    a rank that lost its peers would otherwise sit in a collective until its time-out)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")  # rank 0's stdout (a file, not a pipe: nothing to drain while the ranks run)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print("bench: rank %d exited with status %d; stopping the other ranks" % (r, code), file=sys.stderr)
                    for q in live:
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    out0.seek(0)
    out0 = out0.read()
    sys.stdout.write(out0)
    sys.stdout.flush()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1600)
    ap.add_argument("--warmup", type=int, default=800)
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="CPU time budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--best-cost-epochs", type=int, default=120,
                    help="c21 workloads at N = 1: after the timed region, run the reference driver's loop for at most this many epochs "
                         "(until lambda_1 + mu < 5.2) and a frozen-model control beside it (0: skip)")
    ap.add_argument("--chunk", type=int, default=800, help="calls per host round trip (<= one epoch)")
    ap.add_argument("--barrier-step", action="store_true", help="lock-step form (k_persist) instead of the default asynchronous step (k_async)")
    ap.add_argument("--step", choices=["async", "barrier", "pool"], default=None,
                    help="force a step form (default: the engine's choice -- the pool step k_pool from 256 agents, else k_async)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="BASELINE.json preset: A reference shape (512 agents, 512-1024-512 MLP), B 4096 agents fp32 (default), "
                         "C 8192 agents/GPU bf16 (65536 at --gpus 8), D Ramsey r44 8192 agents/GPU")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None)
    ap.add_argument("--agents", type=int, default=0, help="agents per GPU (default: the workload's)")
    ap.add_argument("--exchange", choices=["torch", "native"], default="torch",
                    help="N > 1: the epoch exchange through torch.distributed (all_gather_into_tensor on the nccl = RCCL backend) or through "
                         "the C ABI's azd_engine_par_update_model_sharded with an ncclComm_t of its own (ncclAllGather on the engine's stream)")
    ap.add_argument("--max-slots", type=int, default=0, help="dense-graph workload: modifiable edge slots a root may bring (default 128; E // 2 = 612 is the drivers' image)")
    ap.add_argument("--prediction-capacity", type=int, default=0, help="override the workload's per-tree prediction arena")
    ap.add_argument("--mlp-dtype", choices=["f32", "bf16"], default=None,
                    help="evaluator weight/activation storage for inference (bf16 = BASELINE configs[2]; f32 accumulate either way)")
    args = ap.parse_args()
    wl_name, agents, dtype = CONFIGS[args.config or "B"]
    if args.workload:
        wl_name = args.workload
        agents = WORKLOADS[wl_name]["agents"]
    wl = dict(WORKLOADS[wl_name])
    wl["agents"] = args.agents if args.agents > 0 else agents
    if args.max_slots > 0:
        wl["max_slots"] = args.max_slots
    if args.prediction_capacity > 0:
        wl["caps"] = dict(wl["caps"], prediction_capacity=args.prediction_capacity)
    mlp_dtype = args.mlp_dtype or dtype
    AGENTS_PER_GPU, TOL, HIDDEN = wl["agents"], wl["tol"], tuple(wl["hidden"])

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)  # (does not return)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run --nproc-per-node %d, or unset WORLD_SIZE and "
                         "let bench.py start its own ranks" % (args.gpus, world, args.gpus))
    if os.environ.get("AZD_BENCH_SPAWN_PROBE") == "1":
        # test hook of the launch path alone (tests/test_parallel_gloo.py, no GPU): rendezvous over gloo with the environment the
        # launcher set, one all-reduce, rank 0 prints what it saw
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t)
        if os.environ.get("AZD_BENCH_SPAWN_PROBE_FAIL") == str(rank):
            raise SystemExit(7)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "spawn_probe", "world": world, "sum": float(t[0]), "gpus": args.gpus, "config": args.config}), flush=True)
        dist.destroy_process_group()
        return
    import torch
    import torch.distributed as dist
    # rehearsal on a one-GPU box (never used by the driver): AZD_BENCH_REHEARSE=1 puts every rank on
    # cuda:0 and swaps RCCL for gloo with CPU staging, to exercise the N > 1 host logic
    rehearse = os.environ.get("AZD_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
        # the ranks share ONE GPU here: the dense-graph space's searcher workgroups of all ranks together must leave CUs to the
        # GEMM launches (a rank with a GPU of its own takes half the chip)
        os.environ.setdefault("AZD_DENSE_POOL_SEARCH_WGS", str(max(8, 96 // max(1, world))))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import azdopt_amd as az
    from azdopt_amd.parallel import ShardedOptimizer

    space = make_space(az, wl)
    sopt = ShardedOptimizer.par_new(
        space, lambda total: az.ActionModel(total, space.STATE_DIM, space.ACTION_DIM, hidden=HIDDEN, seed=SEED,
                                            device=local_rank, dtype=mlp_dtype),
        AGENTS_PER_GPU, dist=dist if world > 1 else None, torch=torch, seed=SEED, device_index=local_rank, rank=rank,
        world_size=world, stage_on_cpu=rehearse, async_step=not (args.barrier_step or args.step == "barrier"),
        pool_step={"pool": True, "async": False, "barrier": False}.get(args.step), **wl["caps"])
    native_comm = None
    if world > 1 and args.exchange == "native" and not rehearse:
        # an RCCL communicator of the bench's own for the native entry point: rank 0 draws the id, torch.distributed carries it
        import ctypes as C

        class _Uid(C.Structure):
            _fields_ = [("internal", C.c_char * 128)]

        rccl = None
        for name in ("librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"):
            try:
                rccl = C.CDLL(name, mode=C.RTLD_GLOBAL)
                break
            except OSError:
                continue
        if rccl is None:
            raise SystemExit("--exchange native: librccl.so not found")
        uid = _Uid()
        if rank == 0 and rccl.ncclGetUniqueId(C.byref(uid)) != 0:
            raise SystemExit("ncclGetUniqueId failed")
        box = [bytes(uid.internal)]
        dist.broadcast_object_list(box, src=0)
        C.memmove(C.byref(uid), box[0], 128)
        native_comm = C.c_void_p()
        rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _Uid, C.c_int]
        if rccl.ncclCommInitRank(C.byref(native_comm), world, uid, rank) != 0:
            raise SystemExit("ncclCommInitRank failed on rank %d" % rank)
        sopt.use_native_exchange(native_comm)
    opt = sopt.shard.opt  # this rank's NablaOptimizer (counters, timing)
    B, B_total = sopt.plan.local_agents, sopt.plan.total_agents
    coll_dev = sopt.coll_device

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    calls_done = 0
    epoch = 0
    losses = []
    boundaries = 0

    def epoch_boundary():
        nonlocal epoch, boundaries
        losses.append(sopt.par_update_model(N_OBS_TOL))  # N > 1: all-gather of the training triple, then the identical step
        sopt.par_reset_trees_policy(SEED, epoch)     # modify_root policy + reset on the device (every space)
        epoch += 1
        boundaries += 1

    def run(n_calls):
        nonlocal calls_done
        left = n_calls
        while left > 0:
            k = min(left, args.chunk, EPOCH_CALLS - calls_done % EPOCH_CALLS)
            sopt.par_roll_out_episodes(TOL, n_calls=k)
            calls_done += k
            left -= k
            if calls_done % EPOCH_CALLS == 0:
                epoch_boundary()

    # A window shorter than an epoch would otherwise sit on the freshly reset trees of call 0 and before the
    # first optimiser step: run one whole epoch (untimed, boundary included) and centre the window in the next
    placed = None
    if args.steps < EPOCH_CALLS and args.warmup + args.steps <= EPOCH_CALLS // 2:
        lead = EPOCH_CALLS + EPOCH_CALLS // 2 - args.steps // 2 - args.warmup
        run(lead)
        placed = (calls_done + args.warmup) % EPOCH_CALLS
    run(args.warmup)
    # per-launch HIP-event timing: free for the CU-resident forms (one launch per <= 800 calls); the launch-per-phase form
    # is replayed from a hipGraph only while it is off, so there it is taken on a short run of its own after the timed one
    form_now = opt.step_form()[0]
    per_call = form_now.startswith("per_call") or (form_now == "none" and wl["kind"] == "dense")
    opt.set_timing(not per_call)
    c0 = opt.counters()
    b0 = boundaries
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    c1 = opt.counters()
    n_bound = boundaries - b0
    form, form_why = opt.step_form()
    timing_note = None
    if per_call:
        k = min(20, EPOCH_CALLS - calls_done % EPOCH_CALLS - 1)
        if k > 0:
            opt.set_timing(True)
            sopt.par_roll_out_episodes(TOL, n_calls=k)
            calls_done += k
            timing_note = "kernel durations from %d extra calls timed launch by launch after the timed region (which replays a hipGraph)" % k
    timing = opt.timing()
    opt.set_timing(False)
    if c1["FAILED"] != 0:
        raise SystemExit("bench: %d agents stopped on a full arena; the rate would count fewer working agents" % c1["FAILED"])

    exp_local = c1["EXPANSIONS"] - c0["EXPANSIONS"]
    per_rank_rates = [exp_local / dt]
    if world > 1:
        t = torch.tensor([dt, float(exp_local)], dtype=torch.float64, device=coll_dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt_max, exp_total = float(tmax[0]), float(t[1])
        rates = torch.empty(world, dtype=torch.float64, device=coll_dev)
        dist.all_gather_into_tensor(rates, torch.tensor([exp_local / dt], dtype=torch.float64, device=coll_dev))
        per_rank_rates = [float(x) for x in rates.cpu()]
    else:
        dt_max, exp_total = dt, float(exp_local)
    # every rank must hold the same parameters after the pooled optimiser steps (64-bit hash, all-gathered); outside the timed region
    replicas_ok = sopt.replicas_identical() if mlp_dtype else True
    if not replicas_ok:
        raise SystemExit("bench: the ranks' model replicas differ after the epoch exchange")
    best_eval, best_cost = sopt.global_argmin()
    if per_call:  # the roofline figures below describe the extra launch-by-launch calls, not the graph replays
        exp_for_roofline = opt.counters()["EXPANSIONS"] - c1["EXPANSIONS"]
    else:
        exp_for_roofline = c1["EXPANSIONS"] - c0["EXPANSIONS"]

    if rank == 0:
        state_bytes = (space.C * space.E * 4 + 512) if wl["kind"] == "ramsey" else (1024 if wl["kind"] == "dense" else 0)
        bytes_per_exp, d = algorithmic_bytes(c0, c1, space.STATE_DIM, space.ACTION_DIM, state_bytes)
        kw = space.KEY_WORDS
        dims_txt = "-".join(str(x) for x in (space.STATE_DIM,) + HIDDEN + (space.ACTION_DIM,))
        launches = max(1, timing["rollout_launches"])
        avg_ms = timing["rollout_ms"] / launches
        bytes_per_launch = bytes_per_exp * exp_for_roofline / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        if wl["kind"] == "dense":  # device keys of the dense space: ranks of the root's modifiable slots, 64 per word
            kw = next(w for w, cap in ((2, 128), (4, 256), (10, 640), (16, 1024)) if space.MAX_SLOTS <= cap)
        kernel = {"async": "k_async<%d>", "barrier": "k_persist<%d>", "per_call": "k_rollout<%d>", "per_call_graph": "k_rollout<%d>",
                  "pool": "k_pool_search<%d>" if wl["kind"] == "dense" else "k_pool<%d>"}.get(form, "?<%d>") % kw
        # HBM traffic is not measured by this process (PMC counters need rocprofv3): it is the per-call figure of the
        # committed profile of the same kernel and workload, scaled to this run's calls per launch, or null
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and wl_name == "c21" and AGENTS_PER_GPU == 4096 and mlp_dtype == "f32" and form in ("async", "pool"):
            try:
                tj = json.load(open(tpath))
                # FETCH_SIZE as counted (one 64-B request per missed record: profiles/r02_gather_calib.txt) + WRITE_SIZE
                cpl = args.steps / launches
                d20 = tj.get("k_pool_driver20")
                if form == "pool" and d20 and abs(cpl - d20["calls_per_launch"]) < 0.5:  # the driver's window: a profile of launches of this very shape
                    traffic = d20["hbm_bytes_per_call"] * cpl
                    traffic_src = "profiles/traffic.json k_pool_driver20 (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of %s in separate passes over the same " \
                                  "command: launches of %.0f calls, the timed one; record gathers calibrated by profiles/r02_gather_calib.txt); not measured in this run" % (kernel, cpl)
                else:
                    per_call = tj["k_pool_hbm_bytes_per_call"] if form == "pool" else tj["k_async"]["hbm_bytes_per_call_raw"]
                    traffic = per_call * cpl
                    traffic_src = "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of %s in separate passes over 800-call launches, " \
                                  "calibrated for record gathers by profiles/r02_gather_calib.txt) x %.0f calls per launch; not measured in this run" \
                                  % (kernel, cpl)
            except Exception:
                traffic = None
        elif os.path.exists(tpath) and getattr(args, "config", None):
            # the other BASELINE configurations: the committed profile of THIS configuration's dominant kernel at its default population
            # (tools/refresh_profiles.sh tccx: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over bench.py --config X), per call x this run's calls
            try:
                e = json.load(open(tpath)).get("by_config", {}).get(args.config)
                if e and e["kernel"] == kernel and e["step_form"] == form and e["agents_per_gpu"] == AGENTS_PER_GPU and e["dtype"] == mlp_dtype:
                    cpl = args.steps / launches
                    traffic = e["hbm_bytes_per_call"] * cpl
                    traffic_src = "profiles/traffic.json by_config[%s] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of %s in separate passes over bench.py --config %s, " \
                                  "round %s) x %.0f calls per launch; not measured in this run" % (args.config, kernel, args.config, e.get("round", "?"), cpl)
            except Exception:
                traffic = None
        # the limiter the kernel actually runs into: a call is a CHAIN of dependent gathers (a node's predictions, then its
        # children's records, per selection step; table probes; cascade records), one round trip each for the wave that runs
        # it.  Bound: resident searching waves / (dependent round trips per call x the idle HBM-miss latency of
        # MI355X_MICROARCH.md, ~900 cycles = 375 ns).  What separates `achieved` from it: queueing behind the other waves'
        # misses on the CU's in-order memory path, and the arithmetic between the loads (lambda_1).
        # Per call with a new node (round 4: a prediction carries its child's summary, PredRec): ONE round trip per selection
        # level -- the node's predictions; the pool step takes the first level's choice from add_actions' registers -- plus the
        # node's own record at the top of every descent from the root (after a terminal or a transposition), two per table
        # probe (slot, then key), two per cascade (the path's records and the node met in one gather, then the arcs of a node
        # with several parents), six for the hand-overs (state in, row / request out, queue words).
        sel = d["SELECT_CALLS"] - (d["EXPANSIONS"] if (form == "pool" and wl["kind"] != "dense") else 0)
        events = d["TERMINALS"] + d["TRANSPOSITIONS"]
        chain = (sel + events + 2 * (d["EXPANSIONS"] + events) + 2 * events) / max(1, d["EXPANSIONS"]) + 6
        waves = opt.pool_split()[1] * 16 if form == "pool" else min(B, 4096)
        lat_peak = waves / (chain * 0.375e-6)
        lat_rate = exp_for_roofline / launches / (avg_ms * 1e-3) if avg_ms > 0 else 0.0
        if n_bound:
            window = "%d calls incl. %d epoch boundar%s (par_update_model + root policy + par_reset_trees; %d calls/epoch)" \
                     % (args.steps, n_bound, "y" if n_bound == 1 else "ies", EPOCH_CALLS)
        elif placed is not None:
            window = "%d calls from call %d of the second %d-call epoch (one untimed epoch incl. its optimiser step before; " \
                     "no epoch boundary inside the window)" % (args.steps, placed, EPOCH_CALLS)
        else:
            window = "%d calls, no epoch boundary inside the window (%d calls/epoch)" % (args.steps, EPOCH_CALLS)
        out = {
            "metric": "node_expansions_per_s", "value": exp_total / dt_max, "unit": "expansions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt_max / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": mlp_dtype, "data": "synthetic",
            "config": {"workload": "%s%s tree search, %d agents/GPU, %s MLP %s, tol %s/%d; timed: %s"
                                   % (wl["name"], (" (roots with %d..%d of the E = %d edge slots modifiable; the engine holds up to %d, E // 2 = %d is the drivers' image)"
                                                   % (space.default_permitted_range() + (space.E, space.MAX_SLOTS, space.E // 2))) if wl["kind"] == "dense" else "",
                                      AGENTS_PER_GPU, "fp32" if mlp_dtype == "f32" else "bf16-storage", dims_txt,
                                      str(TOL[0]).replace(" ", ""), TOL[1], window),
                       "baseline_config": args.config or ("B" if (wl_name, AGENTS_PER_GPU, mlp_dtype) == CONFIGS["B"] else None),
                       "baseline_config_note": baseline_note(args.config or ("B" if (wl_name, AGENTS_PER_GPU, mlp_dtype) == CONFIGS["B"] else None), world, B_total),
                       "agents_total": B_total, "parallelism": f"agents sharded x{world}"},
            "world": world, "per_rank_expansions_per_s": per_rank_rates, "replicas_identical": replicas_ok,
            "epoch_exchange": None if world == 1 else {
                "via": "azd_engine_par_update_model_sharded (ncclAllGather x 3 on the engine's stream + the optimiser step)" if native_comm is not None
                       else ("gloo over host staging (rehearsal)" if rehearse else "torch.distributed all_gather_into_tensor x 3 (backend nccl = RCCL)"),
                "count": getattr(sopt, "exchanges", 0),
                "ms_each": (getattr(sopt, "exchange_ms", 0.0) / max(1, getattr(sopt, "exchanges", 0))),
                "bytes_per_rank_each": 4 * B * (space.STATE_DIM + 2 * space.ACTION_DIM),
                "note": "native: one call = all-gather + optimiser step; torch: the all-gather alone"},
            "step_form": form, "step_form_reason": form_why, "pool_split": list(opt.pool_split()) if form == "pool" else None,
            "evaluator_form": ("outside the kernel: batched bf16 GEMM launches over the rows the searchers have posted, replayed from a hipGraph "
                               "on a second stream while the searchers run") if (form == "pool" and wl["kind"] == "dense") else
                              (("evaluator GROUPS inside the kernel: %d workgroups serve a batch together, their column tiles' weights resident in LDS "
                                "for the whole launch, the batch's activations exchanged through device memory (%d groups, %d-wave slots)" % opt.pool_groups())
                               if (form == "pool" and opt.pool_groups()[0] > 0) else
                               ("evaluator workgroups inside the kernel" if form == "pool" else None)),
            "epoch_boundary_in_timed_region": n_bound > 0, "epoch_boundaries_in_timed_region": n_bound,
            "best_cost_found": best_cost, "best_eval": best_eval,
            "expansions": exp_total, "terminals": d["TERMINALS"], "transpositions": d["TRANSPOSITIONS"],
            "select_calls_per_expansion": d["SELECT_CALLS"] / max(1, d["EXPANSIONS"]),
            "epoch_losses": losses[-3:],
            "calls_per_launch": args.steps / launches,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel,
                         "algorithmic_bytes_per_expansion": bytes_per_exp, "avg_launch_ms": avg_ms, "timing_note": timing_note,
                         "evaluator_ms_per_call": (timing["evaluator_ms"] / launches) if per_call else None,
                         "mlp_flop_per_launch": 2.0 * sum(a * b for a, b in zip((space.STATE_DIM,) + HIDDEN, HIDDEN + (space.ACTION_DIM,)))
                                                * ((B + 15) // 16 * 16) * args.steps / launches,
                         "note": "latency-bound pointer chasing: the rate target and the 40% roofline target are "
                                 "~3 orders of magnitude apart for this workload (SURVEY.md 8d); see roofline_latency"},
            "roofline_latency": {"bound": "hbm-latency", "achieved": lat_rate, "peak": lat_peak, "unit": "expansions/s per GPU",
                                 "frac": lat_rate / lat_peak if lat_peak else None, "searching_waves": waves,
                                 "dependent_round_trips_per_call": chain, "miss_latency_ns": 375,
                                 "note": "peak = searching waves / (dependent gathers per call x idle HBM-miss latency)"},
        }
        # the evaluator's side against the matrix-core peak of the CUs it holds (MI355X_MICROARCH.md: 157.3 TFLOP/s fp32 on the
        # matrix cores, 2.5 PFLOP/s bf16 dense, over 256 CUs): evaluator workgroups of the pool step, or -- dense-graph space -- the
        # CUs the searchers leave to the GEMM launches
        if mlp_dtype and form == "pool":
            flop_per_row = 2.0 * sum(a * b for a, b in zip((space.STATE_DIM,) + HIDDEN, HIDDEN + (space.ACTION_DIM,)))
            split = opt.pool_split()
            cus = (256 - split[1]) if wl["kind"] == "dense" else split[0]
            chip_peak = 157.3 if mlp_dtype == "f32" else 2500.0
            ach = flop_per_row * (exp_total / world) / dt_max / 1e12
            out["evaluator_mfma"] = {"bound": "mfma", "achieved": ach, "peak": chip_peak * cus / 256.0, "unit": "TFLOP/s",
                                     "frac": ach / (chip_peak * cus / 256.0) if cus else None, "cus": cus, "flop_per_row": flop_per_row,
                                     "note": "rows served per second x the model's flop per row, against the dense matrix-core peak of the CUs "
                                             "the evaluator side holds (%s storage)" % ("fp32" if mlp_dtype == "f32" else "bf16")}
        # The epoch's tail (round-4 verdict, item 5): a launch lasts as long as its slowest agent's chain of calls; how far the
        # population's median is from that is what the barrier at the end of a launch costs.  Of the LAST launch of the timed region.
        if form == "pool" and wl["kind"] != "dense":
            try:
                fin = opt.pool_agent_finish_ms()
                fin = fin[fin > 0]
                if fin.size:
                    p50, p90, p99, mx = (float(np.percentile(fin, q)) for q in (50, 90, 99, 100))
                    out["epoch_tail"] = {"calls_in_launch": int(min(args.steps, args.chunk, EPOCH_CALLS)), "agent_finish_ms": {"p50": p50, "p90": p90, "p99": p99, "max": mx},
                                         "launch_over_median": mx / p50 if p50 else None,
                                         "note": "when each agent was through with the last launch's calls (azd_engine_pool_agent_finish); max = the launch"}
            except Exception:
                pass
        # What the searcher SIMDs actually run into since round 4 (profiles/r0x_pool_pmc_sq.txt: a wave issues in 17 % of its cycles and
        # waits for an ISSUE SLOT in another 21 %): instruction issue.  A SIMD takes its turn once every 4 cycles and issues at most one
        # instruction of a class per turn, so a class of I wave-instructions per expansion bounds the chip at SIMDs x clock / (4 I).
        # The counts are not measured by this process (SQ counters need rocprofv3): they are the committed profile of the same kernel
        # and workload (profiles/issue.json, per call), divided by this run's expansions per call.
        ipath = os.path.join(ROOT, "profiles", "issue.json")
        if os.path.exists(ipath) and wl_name == "c21" and AGENTS_PER_GPU == 4096 and mlp_dtype == "f32" and form == "pool":
            try:
                ij = json.load(open(ipath))
                per_exp = {k: v / max(1.0, exp_total / world / args.steps) for k, v in ij["wave_insts_per_call"].items()}
                classes = {"valu": per_exp.get("valu", 0.0), "salu": per_exp.get("salu", 0.0), "lds": per_exp.get("lds", 0.0),
                           "vmem": per_exp.get("vmem_rd", 0.0) + per_exp.get("vmem_wr", 0.0), "smem": per_exp.get("smem", 0.0)}
                clock_ghz = (d["EVAL_LAYER_CLOCKS"] / (d["TICKS_TILE_KLOOP"] * 10.0)) if d.get("TICKS_TILE_KLOOP") else 2.4
                simds = 256 * 4
                worst = max(classes, key=lambda k: classes[k])
                peak_issue = simds * clock_ghz * 1e9 / (4.0 * classes[worst]) if classes[worst] else None
                out["roofline_issue"] = {"bound": "issue", "achieved": lat_rate, "peak": peak_issue, "unit": "expansions/s per GPU",
                                         "frac": lat_rate / peak_issue if peak_issue else None, "binding_class": worst,
                                         "wave_insts_per_expansion": {k: round(v, 1) for k, v in classes.items()},
                                         "wave_insts_per_expansion_total": round(sum(classes.values()), 1),
                                         "simds": simds, "clock_ghz": round(clock_ghz, 3), "cycles_per_issue_turn": 4,
                                         "source": "profiles/issue.json (%s); clock: s_memtime over the evaluator's layers in this run" % ij["kernel"][:60],
                                         "note": "peak = SIMDs x clock / (4 x the binding class's wave-instructions per expansion), all waves of the kernel counted "
                                                 "(searchers, evaluators, polls); classes issue side by side from different waves, so the sum is not the bound"}
            except Exception:
                pass
        if mlp_dtype and form == "pool" and wl["kind"] != "dense" and d.get("EVAL_BATCHES", 0) > 0 and opt.pool_groups()[0] > 0:
            # evaluator groups: no weight stream -- a batch's time is its layer edges (slice out, arrival count, flags, rows' quads in) plus one
            # tile's MFMA chain per layer; what it would be as ONE workgroup's stream is beside it (profiles/r05_bench_lines.txt: AZD_POOL_EVAL_GROUP=0)
            g_, ng_, w_ = opt.pool_groups()
            wbytes = sum(a * b for a, b in zip((space.STATE_DIM,) + HIDDEN, HIDDEN + (space.ACTION_DIM,))) * 4
            chain = sum(((a + 15) // 16) * 4 * 40 for a in (space.STATE_DIM,) + HIDDEN)  # dependent v_mfma_f32_16x16x4_f32 of one tile per layer, 40 clocks each
            us_per_batch = d["TICKS_BATCH"] / 100.0 / d["EVAL_BATCHES"]
            out["evaluator_groups"] = {"members": g_, "groups": ng_, "waves_per_slot": w_, "slots": ng_ * (16 // w_), "us_per_batch": us_per_batch,
                                       "rows_per_batch": d["EVAL_ROWS"] / d["EVAL_BATCHES"], "layer_edges": len(HIDDEN) + 1,
                                       "mfma_chain_us_at_2p4ghz": chain / 2400.0, "weight_bytes_resident": wbytes,
                                       "note": "us_per_batch = batch taken -> rows released, by the slot's leader; the classic form streams the "
                                               "weight_bytes through one CU per batch instead (config A: 86 us per batch)"}
        elif mlp_dtype and form == "pool" and wl["kind"] != "dense" and d.get("EVAL_BATCHES", 0) > 0:
            # what actually bounds an in-kernel evaluator batch: the workgroup's weight stream from L2 -- every
            # batch pulls the whole model through one CU's vector-memory path -- against that path's 64 B per clock
            wbytes = sum(a * b for a, b in zip((space.STATE_DIM,) + HIDDEN, HIDDEN + (space.ACTION_DIM,))) * (4 if mlp_dtype == "f32" else 2)
            us_per_batch = d["TICKS_BATCH"] / 100.0 / d["EVAL_BATCHES"]
            out["evaluator_weight_stream"] = {"bound": "l2->cu", "achieved": wbytes / us_per_batch / 1e3, "peak": 64 * 2.4, "unit": "GB/s per evaluator CU",
                                              "frac": wbytes / us_per_batch / 1e3 / (64 * 2.4), "us_per_batch": us_per_batch,
                                              "rows_per_batch": d["EVAL_ROWS"] / d["EVAL_BATCHES"], "weight_bytes_per_batch": wbytes,
                                              "note": "a batch streams the model's weights once through its CU; measured with a vector-ALU consumer as well "
                                                      "(no MFMA issue at all): the same ~21-25 B per clock under this kernel's load"}
        out["best_cost_run"] = (best_cost_run(az, wl, HIDDEN, AGENTS_PER_GPU, mlp_dtype, args.best_cost_epochs)
                                if (world == 1 and wl["kind"] == "c21" and mlp_dtype and args.best_cost_epochs > 0 and not args.no_cpu_baseline) else None)
        if not args.no_cpu_baseline and world == 1:  # timed beside the GPU run at N = 1 only
            out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 64), wl, HIDDEN, space.default_permitted_range(), args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
