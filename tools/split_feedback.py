#!/usr/bin/env python3
"""The pool step's split between evaluator and searcher workgroups under the engine's measured feedback (engine.hip: pool_fb:
the busy shares of the two sides are balanced from launch to launch): the trajectory of the split and the rate per epoch, beside
fixed splits given on the command line.   python tools/split_feedback.py CONFIG [epochs 14] [fixed ev ...]
CONFIG: A B C D (BASELINE), B8 (8192 agents fp32), X (4096 agents, 384 x 384 model: a shape no sweep was fitted to)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "X"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 14
fixed = [int(x) for x in sys.argv[3:]]


def make():
    if cfg == "D":
        space = az.RamseySpaceNoEdgeRecolor(17, [4, 4])
        tol = ([200, 200, 100, 100, 50, 50, 25, 25], 10)
        B, hidden, dtype, caps = 8192, (256, 256, 256), "f32", dict(prediction_capacity=57344)
    else:
        space = az.ROTModifyParentsOnce(19)
        tol = ([200, 50, 50], 25)
        B, hidden, dtype = {"A": (512, (512, 1024, 512), "f32"), "B": (4096, (256, 256, 256), "f32"), "C": (8192, (256, 256, 256), "bf16"),
                            "B8": (8192, (256, 256, 256), "f32"), "X": (4096, (384, 384), "f32")}[cfg]
        caps = dict(prediction_capacity=49152) if B > 4096 else {}
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=hidden, seed=0, dtype=dtype)
    return az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, **caps), model, tol


def run(n_epochs, label):
    opt, model, tol = make()
    rates, splits = [], []
    for ep in range(n_epochs):
        c0 = opt.counters()["EXPANSIONS"]
        t0 = time.perf_counter()
        opt.par_roll_out_episodes(tol, n_calls=800)
        dt = time.perf_counter() - t0
        rates.append((opt.counters()["EXPANSIONS"] - c0) / dt / 1e6)
        splits.append(opt.pool_split()[0] if opt.step_form()[0] == "pool" else -1)
        opt.par_update_model(200)
        opt.par_reset_trees_policy(0, ep)
    tail = rates[max(1, len(rates) // 2):]
    print("config %-2s %-12s evaluators %s | M exp/s %s | mean of the later half %.2f"
          % (cfg, label, " ".join(map(str, splits)), " ".join("%.1f" % r for r in rates), sum(tail) / len(tail)), flush=True)
    return sum(tail) / len(tail)


os.environ["AZD_POOL_FEEDBACK"] = "0"
res = {}
os.environ.pop("AZD_POOL_EVAL_WGS", None)
res["first guess"] = run(5, "first guess")
for ev in fixed:
    os.environ["AZD_POOL_EVAL_WGS"] = str(ev)
    res["fixed %d" % ev] = run(5, "fixed %d" % ev)
os.environ.pop("AZD_POOL_EVAL_WGS", None)
os.environ["AZD_POOL_FEEDBACK"] = "1"
fb = run(epochs, "feedback")
best = max(res, key=res.get)
print("config %s: feedback %.2f M expansions/s = %.1f %% of the best of the fixed splits tried (%s, %.2f)" % (cfg, fb, 100 * fb / res[best], best, res[best]), flush=True)
