#!/usr/bin/env python3
"""Diagnostic: cost of a call as a function of its position in the epoch and of the launch length.
Runs one epoch of 800 calls as launches of `chunk` calls and prints us per call of every launch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 20
space = az.ROTModifyParentsOnce(19)
model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0)
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B)
tol = ([200, 50, 50], 25)
opt.par_roll_out_episodes(tol, n_calls=800)
opt.par_update_model(200)
opt.par_reset_trees_policy(0, 0)
ts = []
c0 = opt.counters()["EXPANSIONS"]
t_all = time.perf_counter()
for i in range(800 // chunk):
    t0 = time.perf_counter()
    opt.par_roll_out_episodes(tol, n_calls=chunk)
    ts.append((time.perf_counter() - t0) / chunk * 1e6)
t_all = time.perf_counter() - t_all
exp = opt.counters()["EXPANSIONS"] - c0
print("form", opt.step_form()[0], "B", B, "chunk", chunk, "epoch %.1f ms  %.2f M exp/s" % (t_all * 1e3, exp / t_all / 1e6))
print("us per call by launch:", " ".join("%.0f" % x for x in ts))
