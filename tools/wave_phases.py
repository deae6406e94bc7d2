#!/usr/bin/env python3
"""Diagnostic: where a searcher wave's time goes in the pool step, on a build whose shares are the product build's
(make -C azdopt_amd/csrc VARIANT=phases EXTRA=-DAZD_WAVE_PHASES=1; the stamps are summed per wave and flushed once per launch).
usage: AZD_LIB=azdopt_amd/libazdopt_amd_phases.so python tools/wave_phases.py [agents] [calls] [dtype] [hidden]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 800
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
hidden = tuple(int(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else (256, 256, 256)
space = az.ROTModifyParentsOnce(19)
model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=hidden, seed=0, dtype=dtype)
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, pool_step=True)
tol = ([200, 50, 50], 25)
opt.par_roll_out_episodes(tol, n_calls=20)
c0 = opt.counters()
opt.set_timing(True)
t0 = time.perf_counter()
opt.par_roll_out_episodes(tol, n_calls=calls)
dt = time.perf_counter() - t0
c1 = opt.counters()
d = {k: c1[k] - c0[k] for k in c1}
T = opt.timing()["rollout_ms"] * 1e3  # us
ne, ns = opt.pool_split()
ue, us = opt.pool_utilisation()
waves = ns * 16
n = B * calls
tk = lambda k: d[k] / 100.0 / n  # 100 MHz ticks -> us per call
print("form", opt.step_form(), "split", (ne, ns), "B", B, "calls", calls, dtype, hidden)
print("  %.2f M exp/s, launch %.1f ms, searcher waves %d, busy share %.3f (evaluators %.3f)" % (d["EXPANSIONS"] / dt / 1e6, T / 1e3, waves, us, ue))
per_call = waves * T / n
print("  wave time per call %.1f us = holding an agent %.1f + standing by %.1f" % (per_call, per_call * us, per_call * (1 - us)))
if d["TICKS_TOTAL"] == 0:
    print("  (no phase stamps: not an AZD_WAVE_PHASES build)")
    sys.exit(0)
roll, take = tk("TICKS_TOTAL"), tk("TICKS_WAIT")
add = tk("TICKS_ADD_ACTIONS") if "TICKS_ADD_ACTIONS" in d else 0.0
whole = tk("TICKS_TILES")  # slot 28 of these builds: rollout_agent entry to return
print("  holding an agent: add_actions (fence, PendRec, the arrived row) %.1f + rollout_agent %.1f (agent in %.1f, the call %.1f, write-back out) + hand-over (PendRec, drain, join / queue, counters) %.1f"
      % (add, whole, whole - roll, roll, per_call * us - add - whole))
print("  rollout_agent: select %.1f + lookup %.1f + new node %.1f (lambda_1 %.1f, matching %.1f) + cascade %.1f + rest %.1f"
      % (tk("TICKS_SELECT"), tk("TICKS_LOOKUP"), tk("TICKS_NEWNODE"), tk("TICKS_LAMBDA"), tk("TICKS_MATCHING"), tk("TICKS_CASCADE"),
         roll - tk("TICKS_SELECT") - tk("TICKS_LOOKUP") - tk("TICKS_NEWNODE") - tk("TICKS_CASCADE")))
print("  ticket + wait for its slot, per take: %.1f us (stamped: %.1f per call)" % (take * n / max(1, d["EXPANSIONS"]), take))
print("  per call: select levels %.1f, expansions %.3f, transpositions %.3f, cascade nodes %.1f" % (d["SELECT_CALLS"] / n, d["EXPANSIONS"] / n, d["TRANSPOSITIONS"] / n, d["CASCADE_NODES"] / n))
