"""Diagnostic: is an agent's busy time predictable from the previous epoch?  (PROFILE=1 build.)  Runs the bench's
epoch loop (800 calls, update_model, device root policy + reset) and prints the correlation of per-agent busy
ticks and of per-agent select counts between consecutive epochs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["AZD_LIB"] = os.path.join(ROOT, "azdopt_amd", "libazdopt_amd_prof.so")
sys.path.insert(0, ROOT)
import numpy as np

import azdopt_amd as az

B, calls = 4096, 800
space = az.ROTModifyParentsOnce(19)
model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0)
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B)
tol = ([200, 50, 50], 25)
prev = opt.agent_counters()
busy, sel = [], []
for e in range(4):
    opt.par_roll_out_episodes(tol, n_calls=calls)
    cur = opt.agent_counters()
    busy.append((cur["TICKS_TOTAL"] - prev["TICKS_TOTAL"]).astype(np.float64))
    sel.append((cur["SELECT_CALLS"] - prev["SELECT_CALLS"]).astype(np.float64))
    prev = cur
    opt.par_update_model(200)
    opt.par_reset_trees_policy(0, e)
for e in range(3):
    print("epoch %d -> %d: corr busy %.3f, corr selects %.3f, corr(selects_e, busy_e+1) %.3f; slowest agent busy %.1f us/call (mean %.1f)" % (
        e, e + 1, np.corrcoef(busy[e], busy[e + 1])[0, 1], np.corrcoef(sel[e], sel[e + 1])[0, 1],
        np.corrcoef(sel[e], busy[e + 1])[0, 1], busy[e + 1].max() / 100 / calls, busy[e + 1].mean() / 100 / calls))
