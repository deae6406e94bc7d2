"""Diagnostic: where does k_rollout spend its time?  Uses the PROFILE=1 build (in-kernel 100 MHz
stamps per phase).  Numbers are SHARES, not product timings (stamps cost cycles)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["AZD_LIB"] = os.path.join(ROOT, "azdopt_amd", "libazdopt_amd_prof.so")
sys.path.insert(0, ROOT)
import time

import numpy as np

import azdopt_amd as az

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
kind = sys.argv[3] if len(sys.argv) > 3 else "mlp"
space = az.ROTModifyParentsOnce(19)
dtype = sys.argv[5] if len(sys.argv) > 5 else "f32"
model = (az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0, dtype=dtype) if kind == "mlp"
         else az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, 0))
async_step = len(sys.argv) > 4 and sys.argv[4] == "async"
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, async_step=async_step)
tol = ([200, 50, 50], 25)
opt.par_roll_out_episodes(tol, n_calls=50)
c0 = opt.counters()
opt.set_timing(True)
t0 = time.perf_counter()
opt.par_roll_out_episodes(tol, n_calls=steps)
dt = time.perf_counter() - t0
c1 = opt.counters()
tm = opt.timing()
d = {k: c1[k] - c0[k] for k in c1}
tot = d["TICKS_TOTAL"]
print(f"B={B} steps={steps} model={kind}: {d['EXPANSIONS'] / dt:.3e} exp/s, rollout kernel avg {tm['rollout_ms'] / tm['rollout_launches'] * 1e3:.1f} us")
print("mean agent busy time per call: %.1f us; max single-agent call: %.1f us" % (tot / (B * steps) / 100, c1["TICKS_MAX_CALL"] / 100))
for k in ("SELECT", "LOOKUP", "NEWNODE", "CASCADE"):
    print(f"  {k:8s} {100 * d['TICKS_' + k] / tot:5.1f}%")
if async_step:
    print("  WAIT     %5.1f%% (prediction wait incl. serving the evaluator); batches %d, rows/batch %.2f, tiles/batch %.1f" % (100 * d["TICKS_WAIT"] / tot, d["EVAL_BATCHES"], d["EVAL_ROWS"] / max(1, d["EVAL_BATCHES"]), d["EVAL_TILES"] / max(1, d["EVAL_BATCHES"])))
    print("  evaluator: %.2f us per tile task (set-up %.2f, k loop %.2f, rest epilogue), %.1f us per batch (open->close)" % (
        d["TICKS_TILES"] / max(1, d["EVAL_TILES"]) / 100, d["TICKS_TILE_SETUP"] / max(1, d["EVAL_TILES"]) / 100,
        d["TICKS_TILE_KLOOP"] / max(1, d["EVAL_TILES"]) / 100, d["TICKS_BATCH"] / max(1, d["EVAL_BATCHES"]) / 100))
if not async_step:
    n = B * steps
    parts = (tot / n / 100, d["TICKS_WAIT"] / n / 100, d["TICKS_TILES"] / n / 100, d["TICKS_BATCH"] / n / 100)
    per_call = tm["rollout_ms"] / steps * 1e3
    print("  barrier step, per call per agent: roll-out %.1f us, waiting at the barrier %.1f us, evaluator phase %.1f us, "
          "add_actions %.1f us; kernel %.1f us per call (unaccounted %.1f: roll-out prologue/epilogue, candidate log)" % (
              parts + (per_call, per_call - sum(parts))))
print("  other    %5.1f%%" % (100 * (tot - sum(d['TICKS_' + k] for k in ("SELECT", "LOOKUP", "NEWNODE", "CASCADE"))) / tot))
print("  of NEWNODE: lambda1 %.1f%%, matching %.1f%% of total" % (100 * d["TICKS_LAMBDA"] / tot, 100 * d["TICKS_MATCHING"] / tot))
ev = d["TERMINALS"] + d["TRANSPOSITIONS"]
print("per call per agent: selects %.2f, events %.2f, cascade nodes %.2f, new nodes %.2f" % (
    d["SELECT_CALLS"] / (B * steps), ev / (B * steps), d["CASCADE_NODES"] / (B * steps), (d["EXPANSIONS"] + d["TERMINALS"]) / (B * steps)))
print("per select: %.2f us; per cascade node: %.2f us; per new node: %.2f us; per lookup: %.2f us" % (
    d["TICKS_SELECT"] / max(1, d["SELECT_CALLS"]) / 100, d["TICKS_CASCADE"] / max(1, d["CASCADE_NODES"]) / 100,
    d["TICKS_NEWNODE"] / max(1, d["EXPANSIONS"] + d["TERMINALS"]) / 100,
    d["TICKS_LOOKUP"] / max(1, d["EXPANSIONS"] + d["TERMINALS"] + d["TRANSPOSITIONS"]) / 100))
print("deg %.2f  actions/node %.1f  depth max %d" % (d["SUM_DEG"] / max(1, d["SELECT_CALLS"]), d["SUM_ACTIONS"] / max(1, d["SELECT_CALLS"]), c1["MAX_DEPTH"]))
