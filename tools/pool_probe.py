#!/usr/bin/env python3
"""Diagnostic: time one launch of the pool step and print the evaluator service counters.
usage: pool_probe.py [agents] [calls] [dtype] ; AZD_POOL_EVAL_WGS / AZD_POOL_SEARCH_WGS / AZD_STEP_FORM apply"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
hidden = tuple(int(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else (256, 256, 256)
space = az.ROTModifyParentsOnce(19)
model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=hidden, seed=0, dtype=dtype)
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, pool_step=os.environ.get("AZD_STEP_FORM", "pool") == "pool")
tol = ([200, 50, 50], 25)
opt.par_roll_out_episodes(tol, n_calls=20)
c0 = opt.counters()
opt.set_timing(True)
t0 = time.perf_counter()
opt.par_roll_out_episodes(tol, n_calls=calls)
dt = time.perf_counter() - t0
c1 = opt.counters()
d = {k: c1[k] - c0[k] for k in c1}
form = opt.step_form()
print("form", form, "split", opt.pool_split(), "B", B, "calls", calls, dtype, hidden)
print("  %.1f us/call  %.2f M exp/s   kernel %.1f ms" % (dt / calls * 1e6, d["EXPANSIONS"] / dt / 1e6, opt.timing()["rollout_ms"]))
if d["EVAL_BATCHES"]:
    print("  batches %d rows %d rows/batch %.2f tiles %d  us/batch %.1f" % (d["EVAL_BATCHES"], d["EVAL_ROWS"], d["EVAL_ROWS"] / d["EVAL_BATCHES"], d["EVAL_TILES"], d["TICKS_BATCH"] / 100.0 / d["EVAL_BATCHES"]))
    print("  per batch: rows-in %.1f us, layers %.1f us, out+release %.1f us" % (d["TICKS_TILE_SETUP"] / 100.0 / d["EVAL_BATCHES"], d["TICKS_TILE_KLOOP"] / 100.0 / d["EVAL_BATCHES"],
          (d["TICKS_BATCH"] - d["TICKS_TILE_SETUP"] - d["TICKS_TILE_KLOOP"]) / 100.0 / d["EVAL_BATCHES"]))
    if d.get("EVAL_LAYER_CLOCKS") and d["TICKS_TILE_KLOOP"]:
        ghz = d["EVAL_LAYER_CLOCKS"] / (d["TICKS_TILE_KLOOP"] * 10.0)
        mfma = sum(((b + 15) // 16) * ((a + 15) // 16) * 4 for a, b in zip((space.STATE_DIM,) + hidden, hidden + (space.ACTION_DIM,))) * 32 / 4
        print("  evaluator CUs' clock in the layers: %.2f GHz; fp32 MFMA issue of a batch: %d clocks per SIMD = %.1f us at that clock" % (ghz, mfma, mfma / ghz / 1e3))
