#!/usr/bin/env python3
"""Diagnostic (PROFILE build: AZD_LIB=azdopt_amd/libazdopt_amd_prof.so): which agents a SHORT launch of the pool step waits for --
the driver's window (bench.py --steps 20: one launch of 20 calls in the middle of an epoch).  Per agent: first touched, last call
done, ticks searching / waiting for an evaluator / waiting for a wave, work counts; by finish-time group.
usage: window_tail.py [agents] [calls in the window] [calls before it]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402
from azdopt_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
before = int(sys.argv[3]) if len(sys.argv) > 3 else 390
space = az.ROTModifyParentsOnce(19)
model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0)
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, pool_step=True)
tol = ([200, 50, 50], 25)


def raw():
    out = np.zeros((B, _lib.CTR_COUNT), np.uint64)
    _lib.check(opt._L.azd_engine_agent_counters(opt._h, _lib.ptr(out)), "agent_counters")
    return out.astype(np.int64)


opt.par_roll_out_episodes(tol, n_calls=before)
for _ in range(3):
    opt.par_roll_out_episodes(tol, n_calls=calls)
c0 = raw()
opt.set_timing(True)
opt.par_roll_out_episodes(tol, n_calls=calls)
c1 = raw()
d = c1 - c0
start, fin = c1[:, 31], c1[:, 30]
t0 = start.min()
life = (fin - t0) / 100.0  # us since the launch's first claim
first = (start - t0) / 100.0
print("B %d, a launch of %d calls behind %d: kernel %.3f ms; agents finish at (us): p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f; first touched p50 %.0f p90 %.0f max %.0f" % (
    B, calls, before + 3 * calls, opt.timing()["rollout_ms"], *[np.percentile(life, q) for q in (10, 50, 90, 99)], life.max(), np.percentile(first, 50), np.percentile(first, 90), first.max()))
order = np.argsort(life)
groups = {"all": order, "first 10%": order[: B // 10], "middle 10%": order[B * 45 // 100: B * 55 // 100], "last 10%": order[-(B // 10):], "last 1%": order[-(B // 100):]}
print("%-26s" % "per agent (us)" + "".join("%12s" % g for g in groups))
rows = [("finish", life), ("first touched", first), ("searching (rollout_agent)", d[:, 16] / 100.0), ("waiting for an evaluator", d[:, 24] / 100.0),
        ("waiting for a wave", d[:, 28] / 100.0)]
acc = sum(r[1] for r in rows[1:])
rows.append(("rest (batches, add_actions, hand-over)", life - acc))
for nm, v in rows:
    print("%-26s" % nm[:26] + "".join("%12.0f" % v[ix].mean() for ix in groups.values()))
for k, nm in ((0, "new nodes"), (2, "transpositions"), (1, "terminals"), (4, "select calls"), (7, "cascade nodes")):
    print("%-26s" % nm + "".join("%12.1f" % d[ix, k].mean() for ix in groups.values()))
print("%-26s" % "longest call (us)" + "".join("%12.0f" % (c1[ix, 21].mean() / 100.0) for ix in groups.values()))
