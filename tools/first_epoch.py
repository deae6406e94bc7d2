#!/usr/bin/env python3
"""Diagnostic: duration of the first launches of an engine (first touch of the arenas) against the later ones."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
space = az.ROTModifyParentsOnce(19)
model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0)
opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B)
tol = ([200, 50, 50], 25)
for ep in range(4):
    t0 = time.perf_counter()
    opt.par_roll_out_episodes(tol, n_calls=800)
    c = opt.counters()  # synchronises
    t1 = time.perf_counter()
    print("epoch %d: %.1f ms, form %s, split %s" % (ep, (t1 - t0) * 1e3, opt.step_form()[0], opt.pool_split()))
    opt.par_update_model(200)
    opt.par_reset_trees_policy(0, ep)
