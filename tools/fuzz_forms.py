#!/usr/bin/env python3
"""Differential fuzz of the step forms on the GPU (tests/test_gpu_fuzz.py runs a 40-case slice; more cases: minutes of GPU time): random spaces, populations,
models, call counts and pool-step knobs; every case runs the same seeded search in two forms -- pool step (in one launch, in
random chunks, or call by call through a run-ahead window) against the asynchronous step for c21 / Ramsey, pool searchers
against the launch-per-phase form for the dense-graph space -- and compares trees, counters, argmin, improvement counts and state vectors.
    python tools/fuzz_forms.py [cases 40] [seed 0]"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import azdopt_amd as az  # noqa: E402
from test_gpu_parity import MAIN_CTRS, assert_tree_equal  # noqa: E402


KNOBS = {"AZD_POOL_EARLY_POST": ["0", "1", "2"], "AZD_POOL_EXPRESS_WGS": ["0", "8", "16"], "AZD_POOL_EVAL_WGS": ["24", "64", "100"],
         "AZD_POOL_READY_LANES": ["0", "1"], "AZD_DENSE_POOL_SEARCH_WGS": ["32", "96", "128"], "AZD_DENSE_POOL_ROUNDS": ["1", "4"],
         "AZD_DENSE_POOL_STREAMS": ["1", "2"]}


def run(cases=40, seed=0):
    saved = {k: os.environ.get(k) for k in list(KNOBS) + ["AZD_DENSE_NO_POOL"]}
    try:
        _run(cases, seed)
    finally:  # whatever happens, the process's environment is what it was
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _run(cases, seed):
    rng = random.Random(seed)

    def same(o1, i1, o2, i2, B, tag):
        assert i1 == i2, (tag, i1, i2)
        c1, c2 = o1.counters(), o2.counters()
        for k in MAIN_CTRS:
            assert c1[k] == c2[k], (tag, k, c1[k], c2[k])
        for t in rng.sample(range(B), min(B, 24)):
            assert_tree_equal(o1.get_tree(t), o2.get_tree(t), f"{tag} agent {t}")
        a1, a2 = o1.argmin_data(), o2.argmin_data()
        assert a1.eval == a2.eval and a1.agent == a2.agent and a1.node == a2.node, tag
        assert np.array_equal(o1.state_vecs(), o2.state_vecs()), tag


    for case in range(cases):
        kind = rng.choice(["c21", "c21", "ramsey", "dense", "dense"])
        seed = rng.randrange(1 << 30)
        env = {k: rng.choice(v) for k, v in KNOBS.items() if rng.random() < 0.4}
        calls = rng.choice([20, 60, 150, 400])
        if kind == "c21":
            n = rng.choice([8, 13, 19, 22])
            B = rng.choice([256, 300, 1024, 3000])
            space = az.ROTModifyParentsOnce(n)
            hidden = rng.choice([(64, 64), (256, 256, 256), (128,)])
            dtype = rng.choice(["f32", "bf16"])
            tol = ([200, 50, 50], 25)
            kw = {}
        elif kind == "ramsey":
            B = rng.choice([256, 512, 2048])
            space = az.RamseySpaceNoEdgeRecolor(rng.choice([16, 17]), [4, 4], [1.0, 1.0]) if rng.random() < 0.5 else az.RamseySpaceNoEdgeRecolor(16, [3, 3, 3], [1.0, 1.0, 1.0])
            hidden = rng.choice([(64, 64), (256, 256)])
            dtype = rng.choice(["f32", "bf16"])
            tol = ([200, 200, 100, 100, 50, 50, 25, 25], 10)
            kw = dict(prediction_capacity=131072)
            calls = min(calls, 150)
        else:
            n = rng.choice([12, 20, 50])
            slots = rng.choice([128, 128, 256, 612])
            B = rng.choice([256, 512, 1500])
            space = az.DenseGraphSpace(n, rng.choice([0.1, 0.3]), max_slots=slots)
            hidden = rng.choice([(64,), (512, 512, 512)])
            dtype = "bf16"
            tol = ([200, 50, 50], 25)
            kw = dict(prediction_capacity=262144)
            calls = min(calls, 150)
        lo, hi = space.default_permitted_range()
        kmin = rng.randint(lo, max(lo, hi // 2))
        roots = space.generate_roots(seed, B, kmin=kmin, kmax=hi)
        mode = rng.choice(["one", "chunks", "window"]) if kind != "dense" else rng.choice(["one", "chunks"])
        tag = f"case {case}: {kind} B={B} calls={calls} hidden={hidden} {dtype} mode={mode} env={env} seed={seed}"
        print(tag, flush=True)
        for k in KNOBS:
            os.environ.pop(k, None)
        os.environ.update(env)

        def mk(**more):
            m = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=hidden, seed=seed, dtype=dtype)
            return az.NablaOptimizer.par_new(space, roots, m, B, **kw, **more)

        o1 = mk(pool_step=True)
        if mode == "one":
            i1 = o1.par_roll_out_episodes(tol, n_calls=calls)
        elif mode == "chunks":
            i1, left = 0, calls
            while left:
                k = min(left, rng.choice([1, 3, 17, 64]))
                i1 += o1.par_roll_out_episodes(tol, n_calls=k)
                left -= k
        else:
            if not o1.run_ahead(tol, calls):  # (the engine is not on the pool step: the calls run as they are asked for)
                print("   window refused")
            i1, left = 0, calls
            while left:
                k = min(left, rng.choice([1, 1, 2, 9]))
                i1 += o1.par_roll_out_episodes(tol, n_calls=k)
                if rng.random() < 0.1:
                    o1.argmin_data()
                left -= k
        if o1.step_form()[0] != "pool":  # (a width the in-kernel evaluator does not take: nothing to compare)
            print("   skipped:", o1.step_form())
            for k in KNOBS:  # (the case's knobs must not outlive it: they once leaked into the rest of the test session)
                os.environ.pop(k, None)
            continue
        for k in KNOBS:
            os.environ.pop(k, None)
        if kind == "dense":
            os.environ["AZD_DENSE_NO_POOL"] = "1"
            o2 = mk()
            i2 = o2.par_roll_out_episodes(tol, n_calls=calls)
            os.environ.pop("AZD_DENSE_NO_POOL")
            assert o2.step_form()[0].startswith("per_call"), o2.step_form()
        else:
            o2 = mk(pool_step=False)
            i2 = o2.par_roll_out_episodes(tol, n_calls=calls)
            assert o2.step_form()[0] == "async", o2.step_form()
        same(o1, i1, o2, i2, B, tag)
        # an epoch boundary on both, and a few more calls
        l1, l2 = o1.par_update_model(3), o2.par_update_model(3)
        assert l1 == l2 or (np.isnan(l1) and np.isnan(l2)), (tag, l1, l2)
        o1.par_reset_trees_policy(seed, 1, kmin, hi)
        o2.par_reset_trees_policy(seed, 1, kmin, hi)
        os.environ.update(env)
        j1 = o1.par_roll_out_episodes(tol, n_calls=15)
        for k in KNOBS:
            os.environ.pop(k, None)
        if kind == "dense":
            os.environ["AZD_DENSE_NO_POOL"] = "1"
        j2 = o2.par_roll_out_episodes(tol, n_calls=15)
        os.environ.pop("AZD_DENSE_NO_POOL", None)
        same(o1, j1, o2, j2, B, tag + " (second epoch)")
        del o1, o2
    print("all", cases, "cases agree")



if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
