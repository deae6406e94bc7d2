"""Diagnostic: what do the slowest agents spend their time on?  (PROFILE=1 build, asynchronous step)"""
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
roots = space.generate_roots(0, B)
opt = az.NablaOptimizer.par_new(space, roots, model, B)
opt.par_roll_out_episodes(([200, 50, 50], 25), n_calls=calls)
c = {k: v.astype(np.float64) for k, v in opt.agent_counters().items()}
k = np.array([sum(bin(int(w)).count("1") for w in row) for row in roots[1]])
busy = c["TICKS_TOTAL"] / 100 / calls
order = np.argsort(busy)
groups = {"slowest 1%": order[-41:], "slowest 10%": order[-410:], "median 10%": order[B // 2 - 205:B // 2 + 205], "fastest 10%": order[:410]}
print("group          busy  select newnode cascade lookup | selects  events newnodes casc.nodes  deg  actions/sel  k")
for name, idx in groups.items():
    g = lambda key: c[key][idx].sum()  # noqa: E731
    n = len(idx) * calls
    print("%-12s %6.1f %7.1f %7.1f %7.1f %6.1f | %7.2f %7.2f %8.2f %10.2f %4.1f %11.1f %4.0f" % (
        name, g("TICKS_TOTAL") / 100 / n, g("TICKS_SELECT") / 100 / n, g("TICKS_NEWNODE") / 100 / n, g("TICKS_CASCADE") / 100 / n,
        g("TICKS_LOOKUP") / 100 / n, g("SELECT_CALLS") / n, (g("TERMINALS") + g("TRANSPOSITIONS")) / n,
        (g("EXPANSIONS") + g("TERMINALS")) / n, g("CASCADE_NODES") / n, g("SUM_DEG") / max(1, g("SELECT_CALLS")),
        g("SUM_ACTIONS") / max(1, g("SELECT_CALLS")), k[idx].mean()))
