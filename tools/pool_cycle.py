#!/usr/bin/env python3
"""Diagnostic (needs the PROFILE build: AZD_LIB=azdopt_amd/libazdopt_amd_prof.so): where an agent's cycle goes in the
pool step -- searching, waiting for an evaluator, in the evaluator, waiting for a wave."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 800
space = az.ROTModifyParentsOnce(19)
model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0)
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, pool_step=True)
tol = ([200, 50, 50], 25)
c0 = opt.counters()
t0 = time.perf_counter()
opt.par_roll_out_episodes(tol, n_calls=calls)
dt = time.perf_counter() - t0
c1 = opt.counters()
d = {k: c1[k] - c0[k] for k in c1}
rows = max(1, d["EVAL_ROWS"])
us = lambda k: d[k] / 100.0  # noqa: E731  ticks of 10 ns
print("pool step, %d agents, %d calls: %.1f us per call, %.2f M exp/s (diagnostic build)" % (B, calls, dt / calls * 1e6, d["EXPANSIONS"] / dt / 1e6))
print("  per agent and call with a new node (%d of them):" % rows)
print("    search (rollout_agent)        %6.1f us   (select %.1f, lookup %.1f, new node %.1f of which lambda_1 %.1f, cascade %.1f)" % (
    us("TICKS_TOTAL") / rows, us("TICKS_SELECT") / rows, us("TICKS_LOOKUP") / rows, us("TICKS_NEWNODE") / rows, us("TICKS_LAMBDA") / rows, us("TICKS_CASCADE") / rows))
print("    waiting for an evaluator      %6.1f us" % (us("TICKS_WAIT") / rows))
print("    in the evaluator (per batch)  %6.1f us   (%.1f rows per batch)" % (us("TICKS_BATCH") / max(1, d["EVAL_BATCHES"]), rows / max(1, d["EVAL_BATCHES"])))
print("    waiting for a wave            %6.1f us" % (us("TICKS_TILES") / rows))
print("    mean cycle of an agent        %6.1f us   (launch %.1f us per call: the slowest agent's chain)" % (
    (us("TICKS_TOTAL") + us("TICKS_WAIT") + us("TICKS_TILES")) / rows + us("TICKS_BATCH") / max(1, d["EVAL_BATCHES"]), dt / calls * 1e6))
