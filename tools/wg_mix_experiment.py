"""Experiment: does the composition of a workgroup (which 16 agents share a CU) matter?  The same
4096 roots are dealt to workgroups (a) as generated, (b) sorted by the number of permitted actions k
and dealt one per quantile, (c) the same by the busy time measured in run (a) (upper bound for any
static predictor), (d) sorted so that like sits with like.  Prints the launch time of 800 calls."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import azdopt_amd as az

B, calls = 4096, 800
space = az.ROTModifyParentsOnce(19)
roots = space.generate_roots(0, B)
tol = ([200, 50, 50], 25)
k = np.array([sum(bin(int(w)).count("1") for w in row) for row in roots[1]])


def run(order, label, prof=False):
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0)
    r = (roots[0][order].copy(), roots[1][order].copy())
    opt = az.NablaOptimizer.par_new(space, r, model, B)
    opt.set_timing(True)
    opt.par_roll_out_episodes(tol, n_calls=calls)
    ms = opt.timing()["rollout_ms"]
    print("%-28s %.2f ms  (%.1f us/call)" % (label, ms, ms * 1e3 / calls), flush=True)
    return opt


def deal(rank_order):
    # rank_order: agents sorted by the predictor; workgroup g gets ranks g, g + G, g + 2G, ...
    G = B // 16
    slots = np.empty(B, dtype=np.int64)
    for r, agent in enumerate(rank_order):
        g, w = r % G, r // G
        slots[g * 16 + w] = agent
    return slots


ident = np.arange(B)
for rep in range(2):
    run(ident, "as generated")
    run(deal(np.argsort(k, kind="stable")), "dealt by k")
    run(np.argsort(k, kind="stable"), "like with like (k)")
if os.environ.get("AZD_LIB", "").endswith("_prof.so"):
    opt = run(ident, "as generated (prof)")
    busy = opt.agent_counters()["TICKS_TOTAL"].astype(np.float64)
    run(deal(np.argsort(busy, kind="stable")), "dealt by measured busy")
    run(np.argsort(busy, kind="stable"), "like with like (busy)")
