#!/usr/bin/env python3
"""Diagnostic: pure search throughput (fixed prediction stream, no evaluator) of a step form.
usage: pool_probe_hash.py [agents] [calls] ; AZD_STEP_FORM=pool|async|barrier, AZD_POOL_SEARCH_WGS apply"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 400
space = az.ROTModifyParentsOnce(19)
model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, 0)
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, pool_step=os.environ.get("AZD_STEP_FORM", "pool") == "pool")
tol = ([200, 50, 50], 25)
opt.par_roll_out_episodes(tol, n_calls=20)
c0 = opt.counters()
t0 = time.perf_counter()
opt.par_roll_out_episodes(tol, n_calls=calls)
dt = time.perf_counter() - t0
c1 = opt.counters()
d = {k: c1[k] - c0[k] for k in c1}
print("hash-stream form", opt.step_form()[0], "split", opt.pool_split(), "B", B, "calls", calls,
      " %.1f us/call  %.2f M exp/s" % (dt / calls * 1e6, d["EXPANSIONS"] / dt / 1e6))
