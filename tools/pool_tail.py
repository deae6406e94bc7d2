#!/usr/bin/env python3
"""Diagnostic (needs the PROFILE build: AZD_LIB=azdopt_amd/libazdopt_amd_prof.so): which agents a launch of the pool step
waits for.  Per agent: first touched .. last call done (stamps in counter slots 31 / 30), ticks by phase, work counts."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402
from azdopt_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 800
space = az.ROTModifyParentsOnce(19)
model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0)
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, pool_step=True)
tol = ([200, 50, 50], 25)


def raw():
    out = np.zeros((B, _lib.CTR_COUNT), np.uint64)
    _lib.check(opt._L.azd_engine_agent_counters(opt._h, _lib.ptr(out)), "agent_counters")
    return out.astype(np.int64)


opt.par_roll_out_episodes(tol, n_calls=calls)
opt.par_update_model(200)
opt.par_reset_trees_policy(0, 0)
c0 = raw()
opt.par_roll_out_episodes(tol, n_calls=calls)
c1 = raw()
d = c1 - c0
start, fin = c1[:, 31], c1[:, 30]
t0 = start.min()
life = (fin - t0) / 100.0  # us since the launch's first claim
span = life.max()
print("B %d calls %d: launch %.1f ms; agents finish at (ms): p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (
    B, calls, span / 1e3, *[np.percentile(life, q) / 1e3 for q in (10, 50, 90, 99)], life.max() / 1e3))
names = {0: "new nodes", 1: "terminals", 2: "transpositions", 3: "visited steps", 4: "select calls", 5: "sum deg", 6: "sum actions",
         7: "cascade nodes", 12: "curiosity pairs", 13: "ticks sel preds", 14: "ticks sel kids", 10: "max frontier", 11: "max depth", 21: "ticks max call", 16: "ticks search", 17: "ticks select", 18: "ticks lookup", 19: "ticks new node",
         20: "ticks cascade", 22: "ticks lambda", 23: "ticks matching", 24: "ticks wait eval"}
order = np.argsort(life)
groups = {"all": order, "fastest 10%": order[: B // 10], "middle 10%": order[B * 45 // 100: B * 55 // 100], "slowest 10%": order[-(B // 10):],
          "slowest 1%": order[-(B // 100):]}
print("%-18s" % "per agent" + "".join("%14s" % g for g in groups))
print("%-18s" % "finish ms" + "".join("%14.1f" % (life[ix].mean() / 1e3) for ix in groups.values()))
for k, nm in names.items():
    scale = 0.01 / 1e3 if nm.startswith("ticks") else 1.0  # ticks -> ms
    src = c1 if nm.startswith("max") or nm == "ticks max call" else d  # maxima are not differences
    print("%-18s" % (nm + (" ms" if nm.startswith("ticks") else "")) + "".join("%14.1f" % (src[ix, k].mean() * scale) for ix in groups.values()))
other = life * 1e-3 - (d[:, 16] + d[:, 24]) * 1e-5
print("%-18s" % "rest ms" + "".join("%14.1f" % other[ix].mean() for ix in groups.values()))
print("(rest = finish - search - waiting for an evaluator: batches in flight, waiting for a wave, add_actions)")

print()
print("the 8 slowest agents:")
worst = order[-8:][::-1]
print("%-18s" % "agent" + "".join("%10d" % i for i in worst))
print("%-18s" % "finish ms" + "".join("%10.1f" % (life[i] / 1e3) for i in worst))
print("%-18s" % "first touched ms" + "".join("%10.1f" % ((start[i] - t0) / 1e5) for i in worst))
for k, nm in names.items():
    scale = 0.01 / 1e3 if nm.startswith("ticks") else 1.0
    vals = c1[worst, k] if nm.startswith("max") or nm == "ticks max call" else d[worst, k]
    print("%-18s" % (nm + (" ms" if nm.startswith("ticks") else "")) + "".join("%10.1f" % (v * scale) for v in vals))
