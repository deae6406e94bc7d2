#!/usr/bin/env python3
"""Diagnostic: widest cascade level (MAX_FRONTIER) and deepest path the test-sized and bench-sized runs reach."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import azdopt_amd as az  # noqa: E402


def run(name, space, B, hidden, tol, calls, seed=0):
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=hidden, seed=seed)
    opt = az.NablaOptimizer.par_new(space, space.generate_roots(seed, B), model, B)
    opt.par_roll_out_episodes(tol, n_calls=calls)
    c = opt.counters()
    print("%-28s form %-6s max frontier %4d  max depth %3d  cascade nodes %d  failed %d" % (
        name, opt.step_form()[0], c["MAX_FRONTIER"], c["MAX_DEPTH"], c["CASCADE_NODES"], c["FAILED"]))


run("c21 N=19 4096", az.ROTModifyParentsOnce(19), 4096, (256, 256, 256), ([200, 50, 50], 25), 800)
run("r44 N=17 2048", az.RamseySpaceNoEdgeRecolor(17, [4, 4]), 2048, (256, 256, 256), ([200, 200, 100, 100, 50, 50, 25, 25], 10), 800)
run("r333 N=16 2048", az.RamseySpaceNoEdgeRecolor(16, [3, 3, 3]), 2048, (256, 256, 256), ([200, 200, 100, 100, 50, 50, 25, 25], 10), 800)
