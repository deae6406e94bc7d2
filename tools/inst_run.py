#!/usr/bin/env python3
"""Workload for the instruction table (profiles/r05_pool_insts.txt): config B's population on one step form, under
rocprofv3 --pmc SQ_INSTS_* (tools/refresh_profiles.sh insts).  Prints the expansions and events of the counted launches.
usage: inst_run.py pool|per_call [agents] [calls]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402

form = sys.argv[1] if len(sys.argv) > 1 else "pool"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 400
space = az.ROTModifyParentsOnce(19)
model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0)
kw = dict(pool_step=True) if form == "pool" else dict(persistent=False, pool_step=False)
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, prediction_capacity=65536, **kw)
tol = ([200, 50, 50], 25)
c0 = opt.counters()
if form == "pool":
    opt.par_roll_out_episodes(tol, n_calls=calls)
else:
    for _ in range(calls):
        opt.par_roll_out_episodes(tol, n_calls=1)
c1 = opt.counters()
d = {k: c1[k] - c0[k] for k in c1}
print("form", opt.step_form(), "agents", B, "calls", calls)
print("EXPANSIONS %d TERMINALS %d TRANSPOSITIONS %d SELECT_CALLS %d SUM_DEG %d SUM_ACTIONS %d CASCADE_NODES %d CURIOSITY_PAIRS %d NEW_PREDS %d" % tuple(
    d[k] for k in ("EXPANSIONS", "TERMINALS", "TRANSPOSITIONS", "SELECT_CALLS", "SUM_DEG", "SUM_ACTIONS", "CASCADE_NODES", "CURIOSITY_PAIRS", "NEW_PREDS")))
