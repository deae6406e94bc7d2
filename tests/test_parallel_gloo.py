"""world_size-2 gloo tests of the N > 1 path (CPU): shard plan, shard-invariant root / prediction
streams, all-gather of the training triple in global agent order, MINLOC argmin."""
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
