#!/usr/bin/env python3
"""Diagnostic (make PROFILE=1 build, AZD_LIB=azdopt_amd/libazdopt_amd_prof.so): where a searcher wave's time goes per call of the
dense-graph workload (config E), from the in-kernel phase stamps (100 MHz ticks, summed over agents).
    python tools/probe_dense_phases.py [agents 8192] [max_slots 128]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402

B, calls = (int(sys.argv[1]) if len(sys.argv) > 1 else 8192), 400
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 128
space = az.DenseGraphSpace(50, 0.1, max_slots=slots)
model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(512, 512, 512), seed=0, dtype="bf16")
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, prediction_capacity=131072 if slots <= 128 else 524288)
opt.par_roll_out_episodes(([200, 50, 50], 25), n_calls=50)
c0 = opt.counters()
t0 = time.perf_counter()
opt.par_roll_out_episodes(([200, 50, 50], 25), n_calls=calls)
dt = time.perf_counter() - t0
c1 = opt.counters()
d = {k: c1[k] - c0[k] for k in c1}
n = B * calls
print("form", opt.step_form(), "rate %.2f M/s" % (d["EXPANSIONS"] / dt / 1e6), "pool split", opt.pool_split())
print("per agent-call: total %.1f us  select %.1f  lookup %.1f  newnode %.1f (lambda %.1f matching %.1f)  cascade %.1f"
      % tuple(d[k] / n / 100.0 for k in ("TICKS_TOTAL", "TICKS_SELECT", "TICKS_LOOKUP", "TICKS_NEWNODE", "TICKS_LAMBDA", "TICKS_MATCHING", "TICKS_CASCADE")))
print("expansions/call %.3f  select levels/expansion %.2f  wall per call of a wave: %.1f us (%d waves)"
      % (d["EXPANSIONS"] / n, d["SELECT_CALLS"] / max(1, d["EXPANSIONS"]), dt / calls * opt.pool_split()[1] * 16 / B * 1e6, opt.pool_split()[1] * 16))
