#!/usr/bin/env python3
"""Diagnostic: what the host's calling granularity costs.  Runs one epoch of 800 calls of a BASELINE config (A: 512 agents,
512-1024-512 model, the reference's own shape; B: 4096 agents, 3 x 256) as launches of `chunk` calls each -- chunk 1 is the
reference driver's pattern (04-c21-tree.rs:143: one par_roll_out_episodes per episode, ArgminImprovement looked at every
time) -- and prints expansions/s and the time per call of every launch.
    python tools/launch_curve.py [A|B] [chunk ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "B"
chunks = sys.argv[2:] or ["1", "5", "20", "800", "w1", "w5"]
B, hidden = {"A": (512, (512, 1024, 512)), "B": (4096, (256, 256, 256))}[cfg]
space = az.ROTModifyParentsOnce(19)
tol = ([200, 50, 50], 25)
for spec in chunks:
    window, chunk = spec.startswith("w"), int(spec.lstrip("w"))
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=hidden, seed=0)
    opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B)
    opt.par_roll_out_episodes(tol, n_calls=800)
    opt.par_update_model(200)
    opt.par_reset_trees_policy(0, 0)
    ts = []
    c0 = opt.counters()["EXPANSIONS"]
    t_all = time.perf_counter()
    if window:
        assert opt.run_ahead(tol, 800)
    for i in range(800 // chunk):
        t0 = time.perf_counter()
        opt.par_roll_out_episodes(tol, n_calls=chunk)
        ts.append((time.perf_counter() - t0) / chunk * 1e6)
    t_all = time.perf_counter() - t_all
    exp = opt.counters()["EXPANSIONS"] - c0
    print("config %s calls_per_%s %4d form %s  epoch %8.1f ms  %6.2f M expansions/s   us per call: first %.0f median %.0f last %.0f"
          % (cfg, "request (one launch, run ahead)" if window else "launch", chunk, opt.step_form()[0], t_all * 1e3, exp / t_all / 1e6, ts[0], sorted(ts)[len(ts) // 2], ts[-1]), flush=True)
    del opt, model
