"""A/B of the two CU-resident step forms on the bench workload (interleaved rounds, one process)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import azdopt_amd as az
B, calls = 4096, 400
space = az.ROTModifyParentsOnce(19)
tol = ([200, 50, 50], 25)
res = {}
for rnd in range(2):
    for mode in ("barrier", "async"):
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0)
        o = az.NablaOptimizer.par_new(space, space.generate_roots(0, B), model, B, async_step=(mode == "async"))
        o.par_roll_out_episodes(tol, n_calls=50)
        c0 = o.counters()["EXPANSIONS"]
        t0 = time.perf_counter()
        o.par_roll_out_episodes(tol, n_calls=calls)
        dt = time.perf_counter() - t0
        res.setdefault(mode, []).append((o.counters()["EXPANSIONS"] - c0) / dt)
        del o, model
for k, v in res.items():
    print(k, ["%.3e" % x for x in v])
