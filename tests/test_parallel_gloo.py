"""world_size-2 gloo tests of the N > 1 path (CPU): shard plan, shard-invariant root / prediction
streams, all-gather of the training triple in global agent order, MINLOC argmin, and the ShardedOptimizer class
end to end (two epochs with the MLP, the pooled optimiser step and the root policy) against ONE optimizer over
the whole population."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from azdopt_amd.parallel import ShardPlan, allgather_training_triple, global_argmin
        from azdopt_amd.space import ROTModifyParentsOnce
        from oracle import orc
        per = 6
        plan = ShardPlan(world, rank, per)
        space = ROTModifyParentsOnce(19)
        # host side of par_new on this shard: roots keyed by GLOBAL agent id
        parents, permitted = space.generate_roots(3, per, first_agent=plan.first_agent)
        full_p, full_m = space.generate_roots(3, plan.total_agents, first_agent=0)
        sl = slice(plan.first_agent, plan.first_agent + per)
        ok_roots = np.array_equal(parents, full_p[sl]) and np.array_equal(permitted, full_m[sl])
        # the oracle stands in for the engine here (no GPU): run the shard, build the local triple
        e = orc.Engine(19, per, threads=1)
        e.new_begin(parents, permitted)
        e.new_end(orc.hash_predictions(3, plan.first_agent, per, e.A, 0))
        for call in range(1, 31):
            e.rollout_begin([200, 50, 50], 25)
            e.rollout_end(orc.hash_predictions(3, plan.first_agent, per, e.A, call))
        obs, w = e.observe(1)
        sv = e.state_vecs()
        local = [torch.from_numpy(x.copy()) for x in (sv, obs, w)]
        gathered = allgather_training_triple(dist, torch, local, world)
        am = e.argmin()
        best = global_argmin(dist, torch, float(am["eval"]), am["lambda1"] + am["matching"], device="cpu")
        q.put((rank, ok_roots, [g.numpy() for g in gathered], float(am["eval"]), best))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharding_gloo(orc):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    # reference: the same population in ONE engine
    per, total = 6, 12
    from azdopt_amd.space import ROTModifyParentsOnce
    space = ROTModifyParentsOnce(19)
    parents, permitted = space.generate_roots(3, total)
    e = orc.Engine(19, total, threads=2)
    e.new_begin(parents, permitted)
    e.new_end(orc.hash_predictions(3, 0, total, e.A, 0))
    for call in range(1, 31):
        e.rollout_begin([200, 50, 50], 25)
        e.rollout_end(orc.hash_predictions(3, 0, total, e.A, call))
    obs, w = e.observe(1)
    sv = e.state_vecs()
    for r in res:  # every rank holds the identical gathered triple, in global agent order
        g_sv, g_obs, g_w = r[2]
        assert np.array_equal(g_sv, sv) and np.array_equal(g_obs.view(np.uint32), obs.view(np.uint32)) and np.array_equal(g_w, w)
    best_eval = min(r[3] for r in res)
    assert all(r[4][0] == best_eval for r in res)
    assert float(e.argmin()["eval"]) == best_eval


def test_shard_plan():
    from azdopt_amd.parallel import ShardPlan
    p = ShardPlan(8, 3, 8192)
    assert (p.first_agent, p.local_agents, p.total_agents) == (3 * 8192, 8192, 65536)
    assert p.owner(3 * 8192 + 5) == 3 and p.local_index(3 * 8192 + 5) == 5
    with pytest.raises(ValueError):
        ShardPlan(2, 2, 4)


# ---------------------------------------------------------------- ShardedOptimizer end to end
TOL = ([200, 50, 50], 25)
DIMS_HIDDEN = (32, 16)


class OracleShard:
    """The local half of a ShardedOptimizer with the CPU oracle standing in for the GPU engine (there is no GPU
    here): the same methods as azdopt_amd.parallel.EngineShard."""
    coll_device = "cpu"

    def __init__(self, orc, n, plan, seed, total_rows):
        from azdopt_amd.space import ROTModifyParentsOnce
        self.orc, self.plan, self.seed = orc, plan, seed
        self.space = ROTModifyParentsOnce(n)
        self.e = orc.Engine(n, plan.local_agents, threads=1)
        self.mlp = orc.Mlp((self.e.S,) + DIMS_HIDDEN + (self.e.A,), seed=seed, threads=1)
        self.kr = self.space.default_permitted_range()
        roots = self.space.generate_roots(seed, plan.local_agents, first_agent=plan.first_agent)
        self.e.new_begin(*roots)
        self.e.new_end(self.mlp.forward(self.e.state_vecs()))
        self.improved = 0

    def roll_out(self, n_as_tol, n_calls):
        imp = 0
        for _ in range(n_calls):
            self.e.rollout_begin(*n_as_tol)
            imp += self.e.rollout_end(self.mlp.forward(self.e.state_vecs()))
        return imp

    def triple(self, n_obs_tol):
        obs, w = self.e.observe(n_obs_tol)
        return [torch.from_numpy(x.copy()) for x in (self.e.state_vecs(), obs, w)]

    def update_local(self, n_obs_tol):
        sv, obs, w = [t.numpy() for t in self.triple(n_obs_tol)]
        return self.mlp.update(sv, obs, w)

    def update_pooled(self, rows, pooled):
        sv, obs, w = [t.numpy() for t in pooled]
        assert sv.shape[0] == rows
        return self.mlp.update(sv, obs, w)

    def reset_policy(self, seed, epoch, kmin=None, kmax=None):
        roots = self.e.modify_roots(seed, epoch, self.plan.first_agent, *self.kr)
        self.e.reset_begin(*roots)
        self.e.reset_end(self.mlp.forward(self.e.state_vecs()))

    def argmin(self):
        am = self.e.argmin()
        return float(am["eval"]), am["lambda1"] + am["matching"], am

    def expansions(self):
        return self.e.counters()["EXPANSIONS"]

    def params_bytes(self):
        return self.mlp.get_params().tobytes()


def _drive(sopt, epochs, calls, lag=0.0):
    """the driver loop of 04-c21-tree.rs:140-208 over a (Sharded)Optimizer; `lag`: this rank dawdles before every epoch
    exchange (ranks reach the collective at different times: it is the collective that lines them up)"""
    import time
    losses = []
    for epoch in range(epochs):
        sopt.par_roll_out_episodes(TOL, n_calls=calls)
        time.sleep(lag)
        losses.append(sopt.par_update_model(1))
        assert sopt.replicas_identical()
        sopt.par_reset_trees_policy(7, epoch)
    sopt.par_roll_out_episodes(TOL, n_calls=5)
    return losses


def _sharded_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from azdopt_amd.parallel import ShardedOptimizer, ShardPlan
        from oracle import orc
        plan = ShardPlan(world, rank, 20 // world)
        shard = OracleShard(orc, 11, plan, 7, plan.total_agents)
        sopt = ShardedOptimizer(shard, plan, dist, torch)
        losses = _drive(sopt, 2, 25, lag=0.15 * ((rank * 3) % world))
        q.put((rank, losses, shard.mlp.get_params(), sopt.global_argmin(), sopt.total_expansions(),
               shard.e.state_vecs(), float(shard.e.argmin()["eval"])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_optimizer_equals_one_optimizer_gloo(orc, world):
    """2 ranks x 10 agents and 4 ranks x 5 agents through ShardedOptimizer (all-gather of the training triple, pooled
    optimiser step, per-rank root policy, MINLOC), the ranks reaching every exchange at different times == one optimizer
    over the 20 agents: losses, parameters (hash-checked across ranks after every step), state vectors, total expansions
    and the best evaluation, bit for bit."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    from azdopt_amd.parallel import ShardedOptimizer, ShardPlan
    plan = ShardPlan(1, 0, 20)
    one = OracleShard(orc, 11, plan, 7, 20)
    sopt = ShardedOptimizer(one, plan, None, torch)
    losses = _drive(sopt, 2, 25)
    params = one.mlp.get_params()
    for r in res:
        assert r[1] == losses                                                      # the pooled loss, every epoch
        assert np.array_equal(r[2].view(np.uint32), params.view(np.uint32))        # replicas in lock-step
        assert r[3][0] == sopt.global_argmin()[0] and r[4] == sopt.total_expansions()
    sv = one.e.state_vecs()
    assert np.array_equal(np.concatenate([r[5] for r in res]), sv)                 # same trees => same current states
    assert min(r[6] for r in res) == float(one.e.argmin()["eval"])
    assert losses[0] > 0 and np.isfinite(losses).all()


def _bench_cmd(*args, **env):
    import subprocess
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)


def test_bench_launches_its_own_ranks():
    """Round-4 verdict, item 3: `python bench.py --gpus N` with no launcher around it starts its N ranks itself (before any GPU call),
    hands them the rendezvous environment torch.distributed.run would, prints rank 0's ONE line and returns the ranks' status.
    The launch path alone, over gloo on CPUs (AZD_BENCH_SPAWN_PROBE: the ranks rendezvous, all-reduce, rank 0 reports)."""
    import json
    r = _bench_cmd("--gpus", "2", "--config", "C", "--steps", "20", "--warmup", "5", AZD_BENCH_SPAWN_PROBE="1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j == {"metric": "spawn_probe", "world": 2, "sum": 3.0, "gpus": 2, "config": "C"}
    # four ranks, one of which fails: the job's status is that rank's, and nobody is left behind in a collective
    r = _bench_cmd("--gpus", "4", AZD_BENCH_SPAWN_PROBE="1", AZD_BENCH_SPAWN_PROBE_FAIL="2")
    assert r.returncode == 7 and "rank 2 exited with status 7" in r.stderr
    assert not [x for x in r.stdout.splitlines() if x.startswith("{")]


def test_bench_under_a_launcher_still_checks_the_world_size():
    r = _bench_cmd("--gpus", "2", AZD_BENCH_SPAWN_PROBE="1", WORLD_SIZE="1", RANK="0")
    # (WORLD_SIZE set: the launcher's job; a mismatch is refused as before)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
