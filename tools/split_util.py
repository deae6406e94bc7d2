#!/usr/bin/env python3
"""Evaluator utilisation (time its workgroups spend on batches / their share of the launch) across splits of the pool step:
is the best split the one with a particular utilisation?   python tools/split_util.py CONFIG [ev ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "B"
evs = [int(x) for x in sys.argv[2:]] or [0]
tol = ([200, 50, 50], 25)
os.environ["AZD_POOL_FEEDBACK"] = "0"
for ev in evs:
    if ev:
        os.environ["AZD_POOL_EVAL_WGS"] = str(ev)
    if cfg == "D":
        space = az.RamseySpaceNoEdgeRecolor(17, [4, 4])
        tolx = ([200, 200, 100, 100, 50, 50, 25, 25], 10)
        B, hidden, dtype, caps = 8192, (256, 256, 256), "f32", dict(prediction_capacity=57344)
    else:
        space = az.ROTModifyParentsOnce(19)
        tolx = tol
        B, hidden, dtype = {"A": (512, (512, 1024, 512), "f32"), "B": (4096, (256, 256, 256), "f32"), "C": (8192, (256, 256, 256), "bf16"),
                            "B8": (8192, (256, 256, 256), "f32"), "X": (4096, (384, 384), "f32")}[cfg]
        caps = dict(prediction_capacity=49152) if B > 4096 else {}
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=hidden, seed=0, dtype=dtype)
    opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, **caps)
    opt.par_roll_out_episodes(tolx, n_calls=800)
    rates, utils, rows, us = [], [], [], []
    for ep in range(4):
        opt.par_update_model(200)
        opt.par_reset_trees_policy(0, ep)
        c0 = opt.counters()
        t0 = time.perf_counter()
        opt.par_roll_out_episodes(tolx, n_calls=800)
        dt = time.perf_counter() - t0
        c1 = opt.counters()
        n_ev = opt.pool_split()[0]
        rates.append((c1["EXPANSIONS"] - c0["EXPANSIONS"]) / dt / 1e6)
        utils.append((c1["TICKS_BATCH"] - c0["TICKS_BATCH"]) * 1e-8 / (n_ev * dt))
        us.append(opt.pool_utilisation())
        rows.append((c1["EVAL_ROWS"] - c0["EVAL_ROWS"]) / max(1, c1["EVAL_BATCHES"] - c0["EVAL_BATCHES"]))
    print("config %-2s evaluators %3d searchers %3d: %.2f M exp/s   busy: evaluators %.3f searchers %.3f   rows per batch %.1f"
          % (cfg, opt.pool_split()[0], opt.pool_split()[1], sum(rates) / 4, sum(u[0] for u in us) / 4, sum(u[1] for u in us) / 4, sum(rows) / 4), flush=True)
    del opt, model
