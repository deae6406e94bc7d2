"""GPU parity of the Ramsey colour-reassignment space against the CPU oracle:
exported trees, state vectors, live clique counts, observations, argmin and counters must agree
bit-for-bit (integer / index work; the three f32 formulas evaluate, g and h_sa are single IEEE ops)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MAIN_CTRS = ["EXPANSIONS", "TERMINALS", "TRANSPOSITIONS", "VISITED_STEPS", "SELECT_CALLS", "SUM_DEG", "SUM_ACTIONS",
             "CASCADE_NODES", "NEW_PREDS", "ROOT_EXHAUSTED", "MAX_FRONTIER", "MAX_DEPTH", "CURIOSITY_PAIRS"]


@pytest.fixture(scope="module")
def az():
    import azdopt_amd
    if azdopt_amd.device_count() < 1:
        pytest.fail("no gfx950 device: the GPU tests need the HIP path")
    return azdopt_amd


def assert_tree_equal(tg, to, tag=""):
    for f in to.FIELDS:
        a, b = getattr(tg, f), getattr(to, f)
        assert a.shape == b.shape, (tag, f, a.shape, b.shape)
        if a.dtype.kind == "f":
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (tag, f)
        else:
            assert np.array_equal(a.astype(np.int64), b.astype(np.int64)), (tag, f)


def run_ramsey_parity(az, orc, n, sizes, weights, B, kmin, kmax, tol, steps, epochs, seed, n_obs_tol, check_every=1,
                      sample=None, first_agent=0, threads=8, persistent=True):
    space = az.RamseySpaceNoEdgeRecolor(n, sizes, weights)
    C = len(sizes)
    model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, seed, first_agent)
    colors, permitted = space.generate_roots(seed, B, first_agent=first_agent, kmin=kmin, kmax=kmax)
    co, mo = orc.gen_ramsey_roots(seed, 0, first_agent, B, n, C, kmin, kmax)
    assert np.array_equal(colors, co) and np.array_equal(permitted, mo)
    opt = az.NablaOptimizer.par_new(space, (colors, permitted), model, B, first_agent=first_agent, persistent=persistent)
    oe = orc.Engine(n, B, threads=threads, ramsey=(sizes, weights))
    assert (oe.S, oe.A, oe.KW) == (space.STATE_DIM, space.ACTION_DIM, space.KEY_WORDS)
    oe.new_begin(colors, permitted)
    call = 0
    oe.new_end(orc.hash_predictions(seed, first_agent, B, space.ACTION_DIM, call))
    agents = range(B) if sample is None else sample

    def compare(tag):
        assert np.array_equal(opt.state_vecs(), oe.state_vecs()), tag
        for i in agents:
            assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"{tag} agent {i}")
            sg, so = opt.agent_state(i), oe.agent_state(i)
            for k in ("parents", "permitted", "path", "state_pos"):
                assert np.array_equal(sg[k], so[k]), (tag, i, k, sg[k], so[k])
            cg, tg = opt.ramsey_agent_counts(i)
            assert np.array_equal(cg, oe.agent_counts(i)), (tag, i)
        ag, ao = opt.argmin_data(), oe.argmin()
        assert np.array_equal(ag.state["colors"], ao["parents"]), tag
        pw = min(4, space.KEY_WORDS)  # permitted EDGES: E <= 256 bits
        assert np.array_equal(ag.state["permitted"][:pw], ao["permitted"][:pw]) and not ao["permitted"][pw:].any(), tag
        assert ag.eval.tobytes() == ao["eval"].tobytes(), (tag, ag.eval, ao["eval"])
        assert ag.cost["clique_counts"] == oe.argmin_totals()[:C].tolist(), tag
        cg, co_ = opt.counters(), oe.counters()
        for k in MAIN_CTRS:
            assert cg[k] == co_[k], (tag, k, cg[k], co_[k])

    compare("par_new")
    for epoch in range(epochs):
        s = 0
        while s < steps:
            k = min(check_every, steps - s)
            improved_g = opt.par_roll_out_episodes(tol, n_calls=k)
            improved_o = 0
            for _ in range(k):
                oe.rollout_begin(*tol)
                call += 1
                improved_o += oe.rollout_end(orc.hash_predictions(seed, first_agent, B, space.ACTION_DIM, call))
            assert improved_g == improved_o, (epoch, s, improved_g, improved_o)
            s += k
            compare(f"epoch {epoch} step {s}")
        sv, obs, w = opt.observe(n_obs_tol)
        oo, ow = oe.observe(n_obs_tol)
        # h_sa = 1 - c*/c is NaN for a child with c = c* = 0 (a monochromatic-clique-free colouring: the
        # drivers stop there); NaN payloads are not part of the contract, everything else is bit-exact
        nan = np.isnan(oo)
        assert np.array_equal(np.isnan(obs), nan) and np.array_equal(w, ow)
        assert np.array_equal(obs[~nan].view(np.uint32), oo[~nan].view(np.uint32))
        assert np.array_equal(sv, oe.state_vecs())
        ro = oe.modify_roots(seed, epoch, first_agent, kmin, kmax)  # the drivers' modify_root policy (02-r44.rs:196-228)
        rg = opt.modify_roots(seed, epoch, kmin, kmax)               # the same policy on the device
        assert np.array_equal(rg[0], ro[0]) and np.array_equal(rg[1], ro[1]), epoch
        if epoch % 2 == 0:
            opt.par_reset_trees_policy(seed, epoch, kmin, kmax)      # policy + reset without a host round trip
        else:
            opt.par_reset_trees(ro)
        oe.reset_begin(*ro)
        call += 1
        oe.reset_end(orc.hash_predictions(seed, first_agent, B, space.ACTION_DIM, call))
        compare(f"epoch {epoch} reset")
    return opt.counters()


def test_ramsey_triangles_two_colours_every_step(az, orc):
    c = run_ramsey_parity(az, orc, 6, [3, 3], [1.0, 1.0], B=24, kmin=3, kmax=7, tol=([6, 3, 2], 1), steps=60, epochs=2,
                          seed=1, n_obs_tol=2)
    assert c["TERMINALS"] > 0 and c["TRANSPOSITIONS"] > 0 and c["FAILED"] == 0


def test_ramsey_wide_cascades_with_mlp_predictions(az, orc):
    """r44 driven by an MLP's predictions (the same rows every time a state recurs, unlike the hash stream) revisits until the
    DAG under a root is dense: cascades then sweep levels of more than 64 ancestors -- more than one node per lane of the
    level-parallel sweep -- and paths run deeper than the per-agent path stack.  The oracle is fed the GPU's predictions."""
    n, sizes, B, seed, calls = 17, [4, 4], 96, 3, 800
    tol = ([200, 200, 100, 100, 50, 50, 25, 25], 10)
    space = az.RamseySpaceNoEdgeRecolor(n, sizes)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed)
    colors, permitted = space.generate_roots(seed, B)
    opt = az.NablaOptimizer.par_new(space, (colors, permitted), model, B)
    oe = orc.Engine(n, B, threads=8, ramsey=(sizes, [1.0, 1.0]))
    oe.new_begin(colors, permitted)
    oe.new_end(opt.predictions())
    for s in range(calls):
        opt.par_roll_out_episodes(tol)
        oe.rollout_begin(*tol)
        oe.rollout_end(opt.predictions())
    cg, co = opt.counters(), oe.counters()
    for k in MAIN_CTRS:
        assert cg[k] == co[k], (k, cg[k], co[k])
    assert cg["MAX_FRONTIER"] > 64 and cg["MAX_DEPTH"] > 32 and cg["FAILED"] == 0, cg
    assert np.array_equal(opt.state_vecs(), oe.state_vecs())
    for i in range(B):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")


def test_ramsey_cascade_levels_wider_than_the_lds_frontier(az, orc, monkeypatch):
    """The reference's frontier is a BTreeMap (empty_transitions.rs:62-86): unbounded.  Until round 5 a level of more than 256
    distinct ancestors -- what the reference's own r44 run length of 3200 episodes per epoch produces (02-r44.rs:126; 24 of its
    512 agents in the first epoch) -- stopped the agent with FLAG_FRONTIER_CAP.  A level's entries beyond the wave's LDS share now
    go to the agent's slice of a device arena.  Here the LDS share is lowered to 64 entries (test hook), so that the levels of 100+
    ancestors this search reaches run through the arena: every tree and counter against the oracle."""
    monkeypatch.setenv("AZD_DEBUG_FRONTIER_LDS", "64")
    n, sizes, B, seed, calls = 17, [4, 4], 96, 3, 800  # (the population of the test above: its widest level has 147 ancestors)
    tol = ([200, 200, 100, 100, 50, 50, 25, 25], 10)
    space = az.RamseySpaceNoEdgeRecolor(n, sizes)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed)
    colors, permitted = space.generate_roots(seed, B)
    opt = az.NablaOptimizer.par_new(space, (colors, permitted), model, B)
    oe = orc.Engine(n, B, threads=8, ramsey=(sizes, [1.0, 1.0]))
    oe.new_begin(colors, permitted)
    oe.new_end(opt.predictions())
    for s in range(calls):
        opt.par_roll_out_episodes(tol)
        oe.rollout_begin(*tol)
        oe.rollout_end(opt.predictions())
    cg, co = opt.counters(), oe.counters()
    assert cg["FAILED"] == 0, cg
    for k in MAIN_CTRS:
        assert cg[k] == co[k], (k, cg[k], co[k])
    assert cg["MAX_FRONTIER"] > 64, cg["MAX_FRONTIER"]  # beyond the lowered LDS share: the arena held part of a level
    assert np.array_equal(opt.state_vecs(), oe.state_vecs())
    for i in range(B):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")


@pytest.mark.parametrize("persistent", [True, False])
def test_ramsey_k4_and_weights_all_branches(az, orc, persistent):
    c = run_ramsey_parity(az, orc, 9, [4, 3], [1.0, 0.5], B=48, kmin=3, kmax=8, tol=([8, 4, 2], 1), steps=120, epochs=2,
                          seed=2, n_obs_tol=3, check_every=7, persistent=persistent)
    assert c["TERMINALS"] > 0 and c["TRANSPOSITIONS"] > 0 and c["VISITED_STEPS"] > 0 and c["FAILED"] == 0


def test_ramsey_k5_three_colours(az, orc):
    run_ramsey_parity(az, orc, 10, [3, 4, 5], [1.0, 1.0, 2.0], B=32, kmin=4, kmax=10, tol=([10, 5, 3], 2), steps=150, epochs=1,
                      seed=3, n_obs_tol=4, check_every=25)


def test_ramsey_r333_reference_hyperparameters(az, orc):
    """01-r333.rs: N = 16, three colours, triangles; n_as_tol table of the driver (:128-130)"""
    tol = ([200, 200, 200, 100, 100, 100, 50, 50, 50, 25, 25, 25], 10)
    c = run_ramsey_parity(az, orc, 16, [3, 3, 3], [1.0, 1.0, 1.0], B=64, kmin=10, kmax=60, tol=tol, steps=400, epochs=1, seed=0,
                          n_obs_tol=200, check_every=100, sample=range(0, 64, 7))
    assert c["EXPANSIONS"] > 0 and c["FAILED"] == 0


def test_ramsey_r44_reference_hyperparameters(az, orc):
    """02-r44.rs: N = 17, two colours, K4; n_as_tol table of the driver (:128-130)"""
    tol = ([200, 200, 100, 100, 50, 50, 25, 25], 10)
    c = run_ramsey_parity(az, orc, 17, [4, 4], [1.0, 1.0], B=64, kmin=12, kmax=68, tol=tol, steps=400, epochs=1, seed=1,
                          n_obs_tol=200, check_every=100, sample=range(0, 64, 7))
    assert c["EXPANSIONS"] > 0 and c["FAILED"] == 0
    assert c["MAX_DEPTH"] > 32, c  # deeper than the per-agent path stack (PATH_STACK): the cascade's fallback runs too


def test_ramsey_invalid_roots_rejected(az):
    space = az.RamseySpaceNoEdgeRecolor(6, [3, 3])
    model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, 0, 0)
    colors, permitted = space.generate_roots(0, 4, kmin=3, kmax=5)
    bad = colors.copy()
    bad[1, 2] = 2  # colour out of range
    with pytest.raises(az.AzdError):
        az.NablaOptimizer.par_new(space, (bad, permitted), model, 4)
    badp = permitted.copy()
    badp[0, 0] |= np.uint64(1 << 20)  # beyond E = 15
    with pytest.raises(az.AzdError):
        az.NablaOptimizer.par_new(space, (colors, badp), model, 4)


@pytest.mark.parametrize("persistent", [True, False])
def test_ramsey_topology_parity_with_mlp_predictions(az, orc, persistent):
    """r333 end-to-end with the real MLP (STATE 840 is not a multiple of 16: exercises the K tail of the
    in-kernel MFMA evaluator): the oracle is fed the GPU's predictions, trees must match bit-for-bit, and
    the predictions themselves must match the oracle's fp32 MLP on the same state vectors."""
    n, sizes, B, seed = 16, [3, 3, 3], 40, 6  # 40 = 2.5 workgroups of 16 agents
    tol = ([200, 200, 200, 100, 100, 100, 50, 50, 50, 25, 25, 25], 10)
    space = az.RamseySpaceNoEdgeRecolor(n, sizes)
    dims = (space.STATE_DIM, 256, 128, space.ACTION_DIM)
    model = az.ActionModel(B, dims[0], dims[-1], hidden=dims[1:-1], seed=seed)
    om = orc.Mlp(dims, seed=seed, threads=8)
    assert np.array_equal(model.get_params(), om.get_params())
    colors, permitted = space.generate_roots(seed, B)
    opt = az.NablaOptimizer.par_new(space, (colors, permitted), model, B, persistent=persistent)
    oe = orc.Engine(n, B, threads=8, ramsey=(sizes, [1.0] * 3))
    oe.new_begin(colors, permitted)
    oe.new_end(opt.predictions())
    worst = 0.0
    for s in range(60):
        opt.par_roll_out_episodes(tol)
        oe.rollout_begin(*tol)
        sv = oe.state_vecs()
        assert np.array_equal(opt.state_vecs(), sv)
        h = opt.predictions()
        # rows of agents that expanded this call are fresh evaluator outputs of their state vector
        fresh = [i for i in range(B) if oe.agent_state(i)["path"].any()]
        if fresh:
            worst = max(worst, float(np.max(np.abs(h[fresh] - om.forward(sv)[fresh]))))
        oe.rollout_end(h)
    assert worst < 2e-5, worst
    for i in range(B):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")
    loss = opt.par_update_model(5)
    assert np.isfinite(loss)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_ramsey_epoch_is_reproducible_at_full_size(az, dtype):
    """BASELINE config D per GPU (r333, N = 16, 8192 agents = two rounds of workgroups) for an epoch of 400 calls
    in one launch with the MLP in the kernel: batching depends on timing, results must not -- two asynchronous
    runs and (f32) the barrier step give the same counters, argmin, predictions and trees."""
    n, sizes, B, calls = 16, [3, 3, 3], 8192, 400
    tol = ([200, 200, 200, 100, 100, 100, 50, 50, 50, 25, 25, 25], 10)
    space = az.RamseySpaceNoEdgeRecolor(n, sizes)
    roots = space.generate_roots(0, B)
    forms = (True, True, False) if dtype == "f32" else (True, True)  # the barrier step has no bf16 evaluator
    runs = []
    for async_step in forms:
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0, dtype=dtype)
        o = az.NablaOptimizer.par_new(space, roots, model, B, async_step=async_step)
        imp = o.par_roll_out_episodes(tol, n_calls=calls)
        runs.append((o, imp))
    o0, i0 = runs[0]
    c0, a0 = o0.counters(), o0.argmin_data()
    assert c0["EXPANSIONS"] > 0.5 * B * calls and c0["EVAL_ROWS"] == c0["EXPANSIONS"]
    for o, imp in runs[1:]:
        c, am = o.counters(), o.argmin_data()
        assert imp == i0
        for k in ("EXPANSIONS", "TERMINALS", "TRANSPOSITIONS", "VISITED_STEPS", "SELECT_CALLS"):
            assert c[k] == c0[k], k
        assert (am.eval, am.agent, am.node) == (a0.eval, a0.agent, a0.node)
        assert np.array_equal(o.state_vecs().view(np.uint32), o0.state_vecs().view(np.uint32))
        p1, p0 = o.predictions().view(np.uint32), o0.predictions().view(np.uint32)
        assert np.array_equal(p1, p0)
        for i in range(0, B, 511):
            t1, t2 = o.get_tree(i), o0.get_tree(i)
            for f in t1.FIELDS:
                x, y = getattr(t1, f), getattr(t2, f)
                xn, yn = (np.isnan(x), np.isnan(y)) if x.dtype.kind == "f" else (None, None)
                if xn is not None:
                    assert np.array_equal(xn, yn), (i, f)
                    x, y = np.where(xn, 0, x), np.where(yn, 0, y)
                assert x.shape == y.shape and np.array_equal(x, y), (i, f)


def test_ramsey_pool_abort_taken_over_by_k_async_with_a_resume_table_against_the_oracle(az, orc, monkeypatch):
    """Regression for the round-3 fault (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in k_async<RamseySpace<5>>, the build in which
    hipcc had left rollout_agent out of line: DESIGN.md "Two faults, one cause"): r44 (five key words) on the pool step with a
    forced abort, so that k_async<RamseySpace<5>> runs with a NON-NULL resume table -- agents through all calls, agents whose
    row is still due, agents never taken -- and the result is the oracle's."""
    n, sizes, weights, B, seed, calls = 17, [4, 4], [1.0, 1.0], 300, 13, 50
    tol = ([200, 200, 100, 100, 50, 50, 25, 25], 10)
    space = az.RamseySpaceNoEdgeRecolor(n, sizes, weights)
    assert space.KEY_WORDS == 5
    roots = space.generate_roots(seed, B)
    monkeypatch.setenv("AZD_POOL_DEBUG_ABORT_CALL", "5")
    model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, seed).serve_from_pool_evaluators()
    opt = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=True, prediction_capacity=131072)
    imp = opt.par_roll_out_episodes(tol, n_calls=calls)
    form, why = opt.step_form()
    assert form == "async" and why.startswith("pool step aborted"), (form, why)
    oe = orc.Engine(n, B, threads=8, ramsey=(sizes, weights))
    oe.new_begin(*roots)
    oe.new_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, 0))
    io = 0
    for call in range(1, calls + 1):
        oe.rollout_begin(*tol)
        io += oe.rollout_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
    assert imp == io
    cg, co = opt.counters(), oe.counters()
    for k in MAIN_CTRS:
        assert cg[k] == co[k], (k, cg[k], co[k])
    assert np.array_equal(opt.state_vecs(), oe.state_vecs())
    for i in range(0, B, 3):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")


def test_ramsey_product_pool_kernel_with_the_real_model_in_one_launch_against_the_oracle(az, orc):
    """config D's kernel k_pool<RamseySpace<5>, 0> with the real 3 x 256 model, 250 calls in one launch at 256 agents, against the
    oracle fed -- call by call -- with the rows of its own states as the in-kernel evaluator computes them (debug_tile_forward)"""
    n, sizes, weights, B, seed, calls = 17, [4, 4], [1.0, 1.0], 256, 29, 250
    tol = ([200, 200, 100, 100, 50, 50, 25, 25], 10)
    space = az.RamseySpaceNoEdgeRecolor(n, sizes, weights)
    roots = space.generate_roots(seed, B)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed)
    opt = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=True, prediction_capacity=131072)
    oe = orc.Engine(n, B, threads=8, ramsey=(sizes, weights))
    oe.new_begin(*roots)
    oe.new_end(opt.predictions())
    io = 0
    for _ in range(calls):
        oe.rollout_begin(*tol)
        io += oe.rollout_end(opt.debug_tile_forward(oe.state_vecs()))
    ig = opt.par_roll_out_episodes(tol, n_calls=calls)
    assert opt.step_form() == ("pool", "") and ig == io
    cg, co = opt.counters(), oe.counters()
    for k in MAIN_CTRS:
        assert cg[k] == co[k], (k, cg[k], co[k])
    assert np.array_equal(opt.state_vecs(), oe.state_vecs())
    for i in range(0, B, 2):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")
