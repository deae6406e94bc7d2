#!/usr/bin/env python3
"""Diagnostic (PROFILE build: AZD_LIB=azdopt_amd/libazdopt_amd_prof.so): where a call of the dense-graph space (config E) goes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
space = az.DenseGraphSpace(50, 0.1)
model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(512, 512, 512), seed=0, dtype="bf16")
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, prediction_capacity=131072)
tol = ([200, 50, 50], 25)
opt.par_roll_out_episodes(tol, n_calls=20)
c0 = opt.counters()
t0 = time.perf_counter()
opt.par_roll_out_episodes(tol, n_calls=calls)
dt = time.perf_counter() - t0
c1 = opt.counters()
d = {k: c1[k] - c0[k] for k in c1}
n = max(1, d["EXPANSIONS"])
us = lambda k: d[k] / 100.0 / n  # noqa: E731
print("dense N=50, %d agents, %d calls: form %s, %.1f us per call, %.2f M exp/s" % (B, calls, opt.step_form()[0], dt / calls * 1e6, n / dt / 1e6))
print("  per expansion: search %.1f us = select %.1f + lookup %.1f + new node %.1f (lambda_1 %.1f, matching %.1f) + cascade %.1f;  longest call %.0f us" % (
    us("TICKS_TOTAL"), us("TICKS_SELECT"), us("TICKS_LOOKUP"), us("TICKS_NEWNODE"), us("TICKS_LAMBDA"), us("TICKS_MATCHING"), us("TICKS_CASCADE"), c1["TICKS_MAX_CALL"] / 100.0))
print("  select calls per expansion %.1f, terminals %d, transpositions %d" % (d["SELECT_CALLS"] / n, d["TERMINALS"], d["TRANSPOSITIONS"]))
