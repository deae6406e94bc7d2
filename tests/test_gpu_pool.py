"""GPU tests of the pool step (k_pool: agents multiplexed over searcher waves through per-XCD queues, evaluator
workgroups on CUs of their own): same trees, counters, argmin and prediction rows as the other step forms and as
the oracle, whatever wave ran which call and whichever batch carried which row."""
import numpy as np
import pytest

from test_gpu_parity import MAIN_CTRS, TOL_REF, assert_tree_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def az():
    import azdopt_amd
    assert azdopt_amd.device_count() > 0, "no MI355X visible"
    return azdopt_amd


def same_engines(o1, i1, o2, i2, agents):
    assert i1 == i2
    c1, c2 = o1.counters(), o2.counters()
    for k in MAIN_CTRS:
        assert c1[k] == c2[k], k
    for i in agents:
        assert_tree_equal(o1.get_tree(i), o2.get_tree(i), f"agent {i}")
    a1, a2 = o1.argmin_data(), o2.argmin_data()
    assert a1.eval == a2.eval and a1.agent == a2.agent and a1.node == a2.node
    assert np.array_equal(o1.state_vecs(), o2.state_vecs())


def test_xcc_ids_cover_the_chip(az):
    """the pool step keeps a tree on the XCD whose searcher first took it: HW_REG_XCC_ID must tell XCDs apart"""
    from azdopt_amd import _lib
    out = np.full(256, 99, np.uint32)
    _lib.check(az.lib().azd_debug_probe_xcc(0, _lib.ptr(out), 256), "probe_xcc")
    assert out.max() <= 7
    counts = np.bincount(out, minlength=8)
    assert (counts > 0).sum() >= 1 and counts.sum() == 256
    print("blocks per XCC id:", counts.tolist())


def test_pool_step_hash_stream_equals_oracle_and_async(az, orc):
    n, B, seed, calls = 19, 200, 6, 150
    space = az.ROTModifyParentsOnce(n)
    roots = space.generate_roots(seed, B)
    opts = []
    for kw in (dict(pool_step=True), dict(pool_step=False)):
        model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, seed)
        o = az.NablaOptimizer.par_new(space, roots, model, B, **kw)
        imp = o.par_roll_out_episodes(TOL_REF, n_calls=calls)
        opts.append((o, imp))
    assert opts[0][0].step_form()[0] == "pool" and opts[1][0].step_form()[0] == "async"
    same_engines(opts[0][0], opts[0][1], opts[1][0], opts[1][1], range(B))
    oe = orc.Engine(n, B, threads=8)
    oe.new_begin(*roots)
    oe.new_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, 0))
    for call in range(1, calls + 1):
        oe.rollout_begin(*TOL_REF)
        oe.rollout_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
    for i in range(0, B, 9):
        assert_tree_equal(opts[0][0].get_tree(i), oe.export_tree(i), f"agent {i}")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_pool_step_with_mlp_equals_async_and_barrier(az, dtype):
    """the evaluator workgroups run the same MFMA sequence per output element as the other forms' in-kernel
    evaluators: identical prediction rows, hence identical trees after many calls in one launch"""
    n, B, seed, calls = 19, 200, 9, 120
    space = az.ROTModifyParentsOnce(n)
    roots = space.generate_roots(seed, B)
    runs = []
    forms = (dict(pool_step=True), dict(pool_step=False)) + ((dict(async_step=False),) if dtype == "f32" else ())
    for kw in forms:
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed, dtype=dtype)
        o = az.NablaOptimizer.par_new(space, roots, model, B, **kw)
        imp = o.par_roll_out_episodes(TOL_REF, n_calls=calls)
        runs.append((o, imp))
    assert runs[0][0].step_form() == ("pool", "")
    ev, se = runs[0][0].pool_split()
    assert ev >= 1 and se >= 1
    c = runs[0][0].counters()
    assert c["EVAL_ROWS"] == c["EXPANSIONS"] and c["FAILED"] == 0
    for o, imp in runs[1:]:
        same_engines(runs[0][0], runs[0][1], o, imp, range(0, B, 3))
        assert np.array_equal(runs[0][0].predictions().view(np.uint32), o.predictions().view(np.uint32))


def test_pool_step_call_by_call_against_the_oracle(az, orc):
    n, B, seed = 19, 72, 4
    space = az.ROTModifyParentsOnce(n)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed)
    parents, permitted = space.generate_roots(seed, B)
    opt = az.NablaOptimizer.par_new(space, (parents, permitted), model, B, pool_step=True)
    oe = orc.Engine(n, B, threads=8)
    oe.new_begin(parents, permitted)
    oe.new_end(opt.predictions())
    for s in range(60):
        opt.par_roll_out_episodes(TOL_REF)
        oe.rollout_begin(*TOL_REF)
        assert np.array_equal(opt.state_vecs(), oe.state_vecs())
        oe.rollout_end(opt.predictions())
    assert opt.step_form()[0] == "pool"
    for i in range(B):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")
    assert np.isfinite(opt.par_update_model(5))


@pytest.mark.parametrize("B,early_post", [(4096, "0"), (4096, "1"), (4096, "2"), (8192, None), (1024, "2")])
def test_pool_step_full_epoch_equals_async(az, B, early_post, monkeypatch):
    """a whole epoch in one launch: 4096 agents (BASELINE config B) and 8192 (more agents than resident searcher waves:
    every agent migrates between waves and CUs of its XCD many times).  early_post: the request for a prediction row
    leaves before / after the wave computes the new node's cost (0 never, 1 always, 2 by a wave that stood idle; the engine picks by population, all forced here)."""
    if early_post is not None:
        monkeypatch.setenv("AZD_POOL_EARLY_POST", early_post)
    n, calls = 19, 800
    space = az.ROTModifyParentsOnce(n)
    roots = space.generate_roots(0, B)
    runs = []
    for kw in (dict(pool_step=True), dict(pool_step=False)):
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0)
        o = az.NablaOptimizer.par_new(space, roots, model, B, **kw)
        imp = o.par_roll_out_episodes(TOL_REF, n_calls=calls)
        runs.append((o, imp))
    assert runs[0][0].step_form()[0] == "pool"
    same_engines(runs[0][0], runs[0][1], runs[1][0], runs[1][1], range(0, B, 257))
    assert np.array_equal(runs[0][0].predictions().view(np.uint32), runs[1][0].predictions().view(np.uint32))
    # the epoch boundary after a pool launch, and a second epoch on top of it
    for o, _ in runs:
        o.par_update_model(200)
        o.par_reset_trees_policy(0, 0)
    imps = [o.par_roll_out_episodes(TOL_REF, n_calls=100) for o, _ in runs]
    same_engines(runs[0][0], imps[0], runs[1][0], imps[1], range(0, B, 511))


def test_pool_step_reference_shape_and_ramsey(az):
    space = az.ROTModifyParentsOnce(19)
    B = 512
    roots = space.generate_roots(5, B)
    runs = []
    for kw in (dict(pool_step=True), dict(pool_step=False)):
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(512, 1024, 512), seed=5)
        o = az.NablaOptimizer.par_new(space, roots, model, B, **kw)
        runs.append((o, o.par_roll_out_episodes(TOL_REF, n_calls=100)))
    assert runs[0][0].step_form() == ("pool", "")
    same_engines(runs[0][0], runs[0][1], runs[1][0], runs[1][1], range(0, B, 31))
    rs = az.RamseySpaceNoEdgeRecolor(16, [3, 3, 3])
    tol = ([200, 200, 200, 100, 100, 100, 50, 50, 50, 25, 25, 25], 10)
    B = 300
    roots = rs.generate_roots(2, B)
    runs = []
    for kw in (dict(pool_step=True), dict(pool_step=False)):
        model = az.ActionModel(B, rs.STATE_DIM, rs.ACTION_DIM, hidden=(256, 256, 256), seed=2)
        o = az.NablaOptimizer.par_new(rs, roots, model, B, prediction_capacity=98304, **kw)
        runs.append((o, o.par_roll_out_episodes(tol, n_calls=150)))
    assert runs[0][0].step_form() == ("pool", "")
    assert runs[0][1] == runs[1][1]
    c1, c2 = runs[0][0].counters(), runs[1][0].counters()
    for k in MAIN_CTRS:
        assert c1[k] == c2[k], k
    for i in range(0, B, 17):
        assert_tree_equal(runs[0][0].get_tree(i), runs[1][0].get_tree(i), f"ramsey agent {i}")


def test_pool_step_reports_capacity_overflow(az):
    space = az.ROTModifyParentsOnce(19)
    model = az.ActionModel(32, space.STATE_DIM, space.ACTION_DIM, hidden=(64, 64), seed=0)
    opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, 32), model, 32, prediction_capacity=600, pool_step=True)
    with pytest.raises(az.AzdError) as ei:
        opt.par_roll_out_episodes(TOL_REF, n_calls=200)
    assert ei.value.status == 4 and opt.counters()["FAILED"] > 0


@pytest.mark.parametrize("kind", ["layered", "sequence", "ramsey_layered_mlp"])
def test_pool_step_with_layered_states_and_sequence_paths(az, orc, kind):
    """the pool step under the rest of the trait surface: Layered<L, Space> (the state-vector row of several chunks is
    built in the wave's LDS and leaves as write-through stores; chunks the ring does not hold keep their contents),
    sequence-keyed paths (no transposition table), and Layered<2, Ramsey> with the MLP through the evaluator workgroups"""
    if kind == "ramsey_layered_mlp":
        n, sizes, B, seed = 8, [3, 4], 40, 7
        tol = ([6, 3, 2], 1)
        space = az.Layered(az.RamseySpaceNoEdgeRecolor(n, sizes, [1.0, 2.0]), 2)
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(64, 32), seed=seed)
        roots = space.generate_roots(seed, B, kmin=3, kmax=8)
        opt = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=True)
        oe = orc.Engine(n, B, threads=8, ramsey=(sizes, [1.0, 2.0]), layers=2)
        oe.new_begin(*roots)
        oe.new_end(opt.predictions())
        for s in range(50):
            opt.par_roll_out_episodes(tol)
            oe.rollout_begin(*tol)
            assert np.array_equal(opt.state_vecs(), oe.state_vecs()), s
            oe.rollout_end(opt.predictions())
        assert opt.step_form() == ("pool", "")
        for i in range(B):
            assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")
        return
    n, B, seed, kmin, kmax = 8, 40, 3, 2, 10
    tol = ([4, 2, 2], 1)
    layers = 3 if kind == "layered" else 1
    base = az.ROTModifyParentsOnce(n)
    space = az.Layered(base, layers) if layers > 1 else base
    path = az.ActionSet if kind == "layered" else az.ActionSequence
    model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, seed, 0)
    roots = space.generate_roots(seed, B, kmin=kmin, kmax=kmax)
    opt = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=True, path=path)
    oe = orc.Engine(n, B, threads=8, layers=layers, path_kind=0 if kind == "layered" else 1)
    oe.new_begin(*roots)
    call = 0
    oe.new_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
    for s in range(0, 60, 6):
        opt.par_roll_out_episodes(tol, n_calls=6)
        for _ in range(6):
            oe.rollout_begin(*tol)
            call += 1
            oe.rollout_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
        assert np.array_equal(opt.state_vecs(), oe.state_vecs()), s
        for i in range(B):
            assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"step {s} agent {i}")
    assert opt.step_form() == ("pool", "")


# ---------------------------------------------------------------- whole epochs of the pool step against the ORACLE
# The fixed prediction stream served by the pool step's evaluator workgroups (azd_debug_hash_stream_via_evaluators): the rows
# take the MLP's way -- evaluator queues, early post, join counter, ready queues, agents migrating between waves and CUs --
# while their values stay reproducible on the CPU, so whole 800-call launches can be checked against orc.Engine.
_ORACLE_EPOCHS = {}


def _oracle_two_epochs(orc, n, B, seed, calls, n_obs_tol, kmin, kmax):
    key = (n, B, seed, calls)
    if key in _ORACLE_EPOCHS:
        return _ORACLE_EPOCHS[key]
    A = orc.lib().orc_action_dim(n)
    parents, permitted = orc.gen_roots(seed, 0, 0, B, n, kmin, kmax)
    oe = orc.Engine(n, B, threads=16)
    oe.new_begin(parents, permitted)
    call = 0
    oe.new_end(orc.hash_predictions(seed, 0, B, A, call))
    snaps = []
    for epoch in range(2):
        improved = 0
        for _ in range(calls):
            oe.rollout_begin(*TOL_REF)
            call += 1
            improved += oe.rollout_end(orc.hash_predictions(seed, 0, B, A, call))
        snap = dict(improved=improved, counters=oe.counters(), argmin=oe.argmin(), state_vecs=oe.state_vecs(),
                    trees={i: oe.export_tree(i) for i in range(0, B, max(1, B // 24))})
        if epoch == 0:
            snap["obs"], snap["w"] = oe.observe(n_obs_tol)
            snap["root_vecs"] = oe.state_vecs()
            snap["new_roots"] = oe.modify_roots(seed, 0, 0, kmin, kmax)
            oe.reset_begin(*snap["new_roots"])
            call += 1
            oe.reset_end(orc.hash_predictions(seed, 0, B, A, call))
        snaps.append(snap)
    _ORACLE_EPOCHS[key] = (parents, permitted, snaps)
    return _ORACLE_EPOCHS[key]


def _check_against_snapshot(opt, imp, snap, tag):
    assert imp == snap["improved"], tag
    c = opt.counters()
    for k in MAIN_CTRS:
        assert c[k] == snap["counters"][k], (tag, k, c[k], snap["counters"][k])
    assert np.array_equal(opt.state_vecs(), snap["state_vecs"]), tag
    for i, to in snap["trees"].items():
        assert_tree_equal(opt.get_tree(i), to, f"{tag} agent {i}")
    ag, ao = opt.argmin_data(), snap["argmin"]
    assert ag.eval == ao["eval"] and np.array_equal(ag.state["parents"], ao["parents"]) and np.array_equal(ag.state["permitted"], ao["permitted"]), tag
    assert ag.cost["lambda_1"] == ao["lambda1"] and len(ag.cost["matching"]) == ao["matching"], tag


@pytest.mark.parametrize("B,early_post", [(256, "1"), (256, "2"), (3072, "1"), (3072, "2")])
def test_pool_whole_epochs_against_the_oracle(az, orc, B, early_post, monkeypatch):
    """800 calls in ONE k_pool launch, par_update_model + the device root policy, 800 more: trees, counters, improvement counts,
    argmin, observations and the new roots against the oracle.  3072 agents are more than the searching waves the pool
    gives them (every agent changes waves and CUs many times); early_post 1 / 2 = the row's request always / sometimes
    leaves before the wave is through with the agent (PoolArgs::join decides who re-queues it)."""
    monkeypatch.setenv("AZD_POOL_EARLY_POST", early_post)
    n, seed, calls, n_obs_tol = 19, 12, 800, 200
    space = az.ROTModifyParentsOnce(n)
    kmin, kmax = space.default_permitted_range()
    parents, permitted, snaps = _oracle_two_epochs(orc, n, B, seed, calls, n_obs_tol, kmin, kmax)
    roots = space.generate_roots(seed, B)
    assert np.array_equal(roots[0], parents) and np.array_equal(roots[1], permitted)
    model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, seed).serve_from_pool_evaluators()
    opt = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=True)
    imp = opt.par_roll_out_episodes(TOL_REF, n_calls=calls)
    assert opt.step_form() == ("pool", "")
    ev_wgs, se_wgs = opt.pool_split()
    assert ev_wgs >= 1 and se_wgs >= 1
    c = opt.counters()
    assert c["EVAL_ROWS"] == c["EXPANSIONS"] > 0  # every row went through an evaluator workgroup
    _check_against_snapshot(opt, imp, snaps[0], "epoch 0")
    sv, obs, w = opt.observe(n_obs_tol)
    assert np.array_equal(obs.view(np.uint32), snaps[0]["obs"].view(np.uint32)) and np.array_equal(w, snaps[0]["w"])
    assert np.array_equal(sv, snaps[0]["root_vecs"])
    rg = opt.c21_modify_roots(seed, 0, kmin, kmax)
    assert np.array_equal(rg[0], snaps[0]["new_roots"][0]) and np.array_equal(rg[1], snaps[0]["new_roots"][1])
    assert opt.par_update_model(n_obs_tol) == 0.0  # (the fixed stream has nothing to train)
    opt.par_reset_trees_policy(seed, 0)
    imp = opt.par_roll_out_episodes(TOL_REF, n_calls=calls)
    assert opt.step_form() == ("pool", "")
    _check_against_snapshot(opt, imp, snaps[1], "epoch 1")


def test_product_pool_kernel_with_the_real_model_over_whole_epochs_against_the_oracle(az, orc):
    """The PRODUCT kernel k_pool<SP, 0> (no harness bit) with the real 3 x 256 model, 800 calls in one launch at 256 agents, the
    optimiser step and the device root policy, 800 more -- against the oracle.  The oracle cannot compute the model's rows to
    the bit, so it is fed, call by call, with the rows of ITS OWN state vectors as the in-kernel evaluator computes them
    (debug_tile_forward: pool_eval's staging and mlp_tile_task's sums; a row does not depend on the batch it travels in).
    If the launch's trees equal the oracle's at the end, every row and every state along the way did."""
    n, B, seed, calls, n_obs_tol = 19, 256, 31, 800, 200
    space = az.ROTModifyParentsOnce(n)
    kmin, kmax = space.default_permitted_range()
    roots = space.generate_roots(seed, B)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed)
    opt = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=True)
    oe = orc.Engine(n, B, threads=8)
    oe.new_begin(*roots)
    oe.new_end(opt.predictions())  # (the roots' rows come from the model call of par_new: the batched forward)
    for epoch in range(2):
        io = 0
        for _ in range(calls):
            oe.rollout_begin(*TOL_REF)
            io += oe.rollout_end(opt.debug_tile_forward(oe.state_vecs()))
        ig = opt.par_roll_out_episodes(TOL_REF, n_calls=calls)  # ONE launch of the product kernel
        assert opt.step_form() == ("pool", "")
        assert ig == io, epoch
        cg, co = opt.counters(), oe.counters()
        for k in MAIN_CTRS:
            assert cg[k] == co[k], (epoch, k, cg[k], co[k])
        assert cg["EVAL_ROWS"] == cg["EXPANSIONS"] > 100 * B
        assert np.array_equal(opt.state_vecs(), oe.state_vecs()), epoch
        for i in range(B):
            assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"epoch {epoch} agent {i}")
        ag, ao = opt.argmin_data(), oe.argmin()
        assert ag.eval == ao["eval"] and np.array_equal(ag.state["parents"], ao["parents"]), epoch
        if epoch == 0:
            sv, obs, w = opt.observe(n_obs_tol)
            oo, ow = oe.observe(n_obs_tol)
            assert np.array_equal(obs.view(np.uint32), oo.view(np.uint32)) and np.array_equal(w, ow)
            assert np.isfinite(opt.par_update_model(n_obs_tol))  # Adam step on the device: the second epoch runs on new weights
            new_roots = oe.modify_roots(seed, 0, 0, kmin, kmax)
            opt.par_reset_trees_policy(seed, 0)
            oe.reset_begin(*new_roots)
            oe.reset_end(opt.predictions())


def _follow_with_the_oracle(opt, oe, tol, calls):
    """one launch of `calls` calls on the device; the oracle, call by call, fed with the in-kernel evaluator's rows of its own states"""
    io = 0
    for _ in range(calls):
        oe.rollout_begin(*tol)
        io += oe.rollout_end(opt.debug_tile_forward(oe.state_vecs()))
    ig = opt.par_roll_out_episodes(tol, n_calls=calls)
    assert ig == io
    cg, co = opt.counters(), oe.counters()
    for k in MAIN_CTRS:
        assert cg[k] == co[k], (k, cg[k], co[k])
    assert np.array_equal(opt.state_vecs(), oe.state_vecs())


def test_product_pool_kernel_with_bf16_storage_over_a_whole_epoch_against_the_oracle(az, orc):
    """BASELINE configs[2]'s evaluator (bf16 weight / activation storage, f32 accumulate) in the product kernel, 800 calls in one
    launch, against the oracle fed with the same evaluator's rows (debug_tile_forward runs the bf16 tile task)"""
    n, B, seed, calls = 19, 256, 17, 800
    space = az.ROTModifyParentsOnce(n)
    roots = space.generate_roots(seed, B)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed, dtype="bf16")
    opt = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=True)
    oe = orc.Engine(n, B, threads=8)
    oe.new_begin(*roots)
    oe.new_end(opt.predictions())
    _follow_with_the_oracle(opt, oe, TOL_REF, calls)
    assert opt.step_form() == ("pool", "")
    for i in range(0, B, 2):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")


def test_config_c_at_its_real_shape_against_the_oracle(az, orc):
    """BASELINE configs[2] as one case (round-4 verdict, 7a): 8192 agents per GPU, bf16 weight / activation storage, 200 calls in ONE
    launch of the product kernel -- more agents than searcher waves, 32-row evaluator batches, early posts by the agents behind the
    mean -- against the oracle fed with the in-kernel evaluator's rows of its own states: global counters, improvement count, every
    state vector, the argmin, and every 61st tree record by record."""
    n, B, seed, calls = 19, 8192, 41, 200
    space = az.ROTModifyParentsOnce(n)
    roots = space.generate_roots(seed, B)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed, dtype="bf16")
    opt = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=True)
    oe = orc.Engine(n, B, threads=16)
    oe.new_begin(*roots)
    oe.new_end(opt.predictions())
    _follow_with_the_oracle(opt, oe, TOL_REF, calls)
    assert opt.step_form() == ("pool", "")
    c = opt.counters()
    assert c["EXPANSIONS"] > 150 * B and c["EVAL_ROWS"] == c["EXPANSIONS"] and c["EVAL_ROWS"] / c["EVAL_BATCHES"] > 16  # 32-row batches ran
    ag, ao = opt.argmin_data(), oe.argmin()
    assert ag.eval == ao["eval"] and np.array_equal(ag.state["parents"], ao["parents"])
    for i in range(0, B, 61):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")


def test_asynchronous_step_with_the_real_model_in_one_launch_against_the_oracle(az, orc):
    """k_async (agents bound to waves, the evaluator served by the waiting waves of the workgroup: the default below 256 agents)
    with the real model, 300 calls in one launch, against the oracle: its tile task is the pool step's, so the same rows feed it"""
    n, B, seed, calls = 19, 96, 23, 300
    space = az.ROTModifyParentsOnce(n)
    roots = space.generate_roots(seed, B)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed)
    opt = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=False)
    oe = orc.Engine(n, B, threads=8)
    oe.new_begin(*roots)
    oe.new_end(opt.predictions())
    _follow_with_the_oracle(opt, oe, TOL_REF, calls)
    assert opt.step_form()[0] == "async"
    for i in range(B):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")


@pytest.mark.parametrize("model_kind", ["hash", "mlp", "mlp_groups"])
def test_pool_abort_is_taken_over_by_the_async_step(az, orc, model_kind, monkeypatch):
    """PoolCtl::abort (a wait ran into its bound) no longer fails the call: the asynchronous step takes the launch over
    where every agent stands -- some through all calls, some waiting for a row that no evaluator will serve, some never
    taken -- and the results are those of an undisturbed run.  The abort is raised by the test hook, mid-launch."""
    n, B, seed, calls = 19, 300, 21, 60
    space = az.ROTModifyParentsOnce(n)
    roots = space.generate_roots(seed, B)

    def make_model():
        if model_kind == "hash":
            return az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, seed).serve_from_pool_evaluators()
        return az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed)

    ref = az.NablaOptimizer.par_new(space, roots, make_model(), B, pool_step=False)
    imp_ref = ref.par_roll_out_episodes(TOL_REF, n_calls=calls)
    if model_kind == "mlp_groups":  # the evaluator groups' waits leave on the flag too (the hook sits in the first slot's leader there)
        monkeypatch.setenv("AZD_POOL_EVAL_GROUP", "8")
    monkeypatch.setenv("AZD_POOL_DEBUG_ABORT_CALL", "7")
    opt = az.NablaOptimizer.par_new(space, roots, make_model(), B, pool_step=True)
    imp = opt.par_roll_out_episodes(TOL_REF, n_calls=calls)
    form, why = opt.step_form()
    assert form == "async" and why.startswith("pool step aborted"), (form, why)
    same_engines(opt, imp, ref, imp_ref, range(B))
    if model_kind == "hash":
        oe = orc.Engine(n, B, threads=8)
        oe.new_begin(*roots)
        oe.new_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, 0))
        for call in range(1, calls + 1):
            oe.rollout_begin(*TOL_REF)
            oe.rollout_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
        for i in range(0, B, 7):
            assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")
    # the engine is usable afterwards and stays with the asynchronous step
    monkeypatch.delenv("AZD_POOL_DEBUG_ABORT_CALL")
    imp2, imp2_ref = opt.par_roll_out_episodes(TOL_REF, n_calls=30), ref.par_roll_out_episodes(TOL_REF, n_calls=30)
    form, why = opt.step_form()
    assert form == "async" and "aborted" in why
    same_engines(opt, imp2, ref, imp2_ref, range(0, B, 5))


def test_pool_grid_is_clamped_to_what_the_device_holds(az, monkeypatch):
    """searcher and evaluator workgroups wait for each other, so all of them must be resident: the grid follows the occupancy
    query (here: a pretended capacity), whatever the overrides ask for; without room for one of each, the asynchronous step"""
    n, B, seed, calls = 19, 512, 3, 40
    space = az.ROTModifyParentsOnce(n)
    roots = space.generate_roots(seed, B)
    runs = {}
    for cap, ev_wgs in ((None, None), ("40", None), ("24", "200"), ("1", None)):
        if cap is None:
            monkeypatch.delenv("AZD_POOL_MAX_RESIDENT", raising=False)
        else:
            monkeypatch.setenv("AZD_POOL_MAX_RESIDENT", cap)
        if ev_wgs is None:
            monkeypatch.delenv("AZD_POOL_EVAL_WGS", raising=False)
        else:
            monkeypatch.setenv("AZD_POOL_EVAL_WGS", ev_wgs)
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed)
        o = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=True)
        runs[(cap, ev_wgs)] = (o, o.par_roll_out_episodes(TOL_REF, n_calls=calls))
    full = runs[(None, None)]
    assert full[0].step_form() == ("pool", "")
    for key, want in ((("40", None), 40), (("24", "200"), 24)):
        o, imp = runs[key]
        assert o.step_form() == ("pool", "")
        ev, se = o.pool_split()
        assert ev >= 1 and se >= 1 and ev + se <= want, (key, ev, se)
        same_engines(o, imp, full[0], full[1], range(0, B, 37))
    o, imp = runs[("1", None)]
    form, why = o.step_form()
    assert form == "async" and "resident" in why, (form, why)
    same_engines(o, imp, full[0], full[1], range(0, B, 37))


def _window_models(az, kind, B, seed):
    if kind == "ramsey":
        space = az.RamseySpaceNoEdgeRecolor(17, [4, 4], [1.0, 1.0])
        mk = lambda: az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256), seed=seed)
        tol = ([200, 200, 100, 100, 50, 50, 25, 25], 10)
    else:
        space = az.ROTModifyParentsOnce(19)
        if kind == "hash":
            mk = lambda: az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, seed).serve_from_pool_evaluators()
        else:
            mk = lambda: az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed)
        tol = TOL_REF
    return space, mk, tol


def _argmin_tuple(a):
    return (a.eval, a.agent, a.node) + tuple(np.asarray(v).tobytes() for _, v in sorted(a.state.items()))


@pytest.mark.parametrize("kind,B", [("hash", 300), ("mlp", 1024), ("ramsey", 256)])
def test_run_ahead_window_hands_out_the_calls_of_separate_launches(az, orc, kind, B):
    """azd_engine_run_ahead: n calls in one launch, asked for one at a time -- each call's ArgminImprovement, the argmin record
    after every improvement, and the trees at the end are those of n launches of one call (hash stream: and of the oracle,
    call by call)."""
    seed, calls = 9, 120
    space, mk, tol = _window_models(az, kind, B, seed)
    roots = space.generate_roots(seed, B)
    ref = az.NablaOptimizer.par_new(space, roots, mk(), B, pool_step=True)
    want, want_rec = [], []
    for _ in range(calls):
        want.append(ref.par_roll_out_episodes(tol, n_calls=1))
        if want[-1]:
            want_rec.append(_argmin_tuple(ref.argmin_data()))
    assert ref.step_form() == ("pool", "") and sum(want) > 0
    opt = az.NablaOptimizer.par_new(space, roots, mk(), B, pool_step=True)
    assert opt.run_ahead(tol, calls)
    got, got_rec = [], []
    for _ in range(calls):
        got.append(opt.par_roll_out_episodes(tol, n_calls=1))
        if got[-1]:
            got_rec.append(_argmin_tuple(opt.argmin_data()))
    assert got == want
    assert got_rec == want_rec
    same_engines(opt, sum(got), ref, sum(want), range(0, B, 7))
    if kind == "hash":
        oe = orc.Engine(19, B, threads=8)
        oe.new_begin(*roots)
        oe.new_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, 0))
        per_call = []
        for call in range(1, calls + 1):
            oe.rollout_begin(*tol)
            per_call.append(oe.rollout_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call)))
        assert got == per_call
    # the window is closed: the next call is a launch of its own, in both engines
    assert opt.par_roll_out_episodes(tol, n_calls=5) == ref.par_roll_out_episodes(tol, n_calls=5)
    same_engines(opt, 0, ref, 0, range(0, B, 31))


def test_run_ahead_window_in_chunks_closed_early_and_refused(az):
    """calls asked for in chunks; a window closed by another engine call (the calls not asked for have run); a request the
    window cannot answer (another tolerance table); an engine whose step cannot publish calls (the hint is refused)"""
    seed, B, calls = 4, 512, 50
    space, mk, tol = _window_models(az, "mlp", B, seed)
    roots = space.generate_roots(seed, B)
    ref = az.NablaOptimizer.par_new(space, roots, mk(), B, pool_step=True)
    want = [ref.par_roll_out_episodes(tol, n_calls=1) for _ in range(calls)]
    opt = az.NablaOptimizer.par_new(space, roots, mk(), B, pool_step=True)
    assert opt.run_ahead(tol, calls)
    got = [opt.par_roll_out_episodes(tol, n_calls=7) for _ in range(7)] + [opt.par_roll_out_episodes(tol, n_calls=1)]
    assert got == [sum(want[i:i + 7]) for i in range(0, 49, 7)] + [want[49]]
    same_engines(opt, 0, ref, 0, range(0, B, 13))
    # closed early: 20 of 50 calls asked for, then the counters are read
    assert opt.run_ahead(tol, calls)
    want2 = [ref.par_roll_out_episodes(tol, n_calls=1) for _ in range(calls)]
    got2 = [opt.par_roll_out_episodes(tol, n_calls=1) for _ in range(20)]
    assert got2 == want2[:20]
    assert opt.counters()["EXPANSIONS"] == ref.counters()["EXPANSIONS"]  # all 50 have run
    same_engines(opt, 0, ref, 0, range(0, B, 13))
    # another tolerance table: the window is closed (its 30 calls have run) and the call runs as a launch of its own
    assert opt.run_ahead(tol, 30)
    other = ([100, 20], 5)
    ref.par_roll_out_episodes(tol, n_calls=30)
    assert opt.par_roll_out_episodes(other, n_calls=3) == ref.par_roll_out_episodes(other, n_calls=3)
    same_engines(opt, 0, ref, 0, range(0, B, 13))
    # the asynchronous step does not publish calls: refused, and the calls run when asked for
    a1 = az.NablaOptimizer.par_new(space, roots, mk(), B, pool_step=False)
    assert not a1.run_ahead(tol, calls)
    assert [a1.par_roll_out_episodes(tol, n_calls=1) for _ in range(calls)] == want
    assert not opt.run_ahead(tol, 5000)  # more than one launch holds


def test_run_ahead_window_survives_an_aborted_launch(az, monkeypatch):
    """the pool launch behind a window aborts (test hook): the asynchronous step completes its calls; the improvements of the
    calls it completed are reported together, none is lost, and the state is that of an undisturbed run"""
    seed, B, calls = 21, 300, 60
    space, mk, tol = _window_models(az, "mlp", B, seed)
    roots = space.generate_roots(seed, B)
    ref = az.NablaOptimizer.par_new(space, roots, mk(), B, pool_step=False)
    imp_ref = ref.par_roll_out_episodes(tol, n_calls=calls)
    monkeypatch.setenv("AZD_POOL_DEBUG_ABORT_CALL", "7")
    opt = az.NablaOptimizer.par_new(space, roots, mk(), B, pool_step=True)
    assert opt.run_ahead(tol, calls)
    got = [opt.par_roll_out_episodes(tol, n_calls=1) for _ in range(calls)]
    form, why = opt.step_form()
    assert form == "async" and why.startswith("pool step aborted"), (form, why)
    same_engines(opt, sum(got), ref, imp_ref, range(B))


def test_agent_counters_say_when_they_are_not_per_agent(az):
    """Round-4 advisor finding: under the pool step of the product build a searcher wave adds its counts to ONE block when it
    leaves the launch, so azd_engine_agent_counters is no per-agent table there.  The engine says so
    (azd_engine_agent_counters_per_agent), the host mirror refuses to hand the blocks out as per-agent values, and the sums / maxima
    of the blocks are pinned equal to the asynchronous form's, whose blocks ARE per agent."""
    n, B, seed, calls = 19, 320, 4, 60
    space = az.ROTModifyParentsOnce(n)
    roots = space.generate_roots(seed, B)
    pair = []
    for pool in (True, False):
        o = az.NablaOptimizer.par_new(space, roots, az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed), B, pool_step=pool)
        assert o._L.azd_engine_agent_counters_per_agent(o._h) == 1
        o.par_roll_out_episodes(TOL_REF, n_calls=calls)
        pair.append(o)
    pool, asy = pair
    assert pool.step_form()[0] == "pool" and asy.step_form()[0] == "async"
    per_agent = asy.agent_counters()
    assert asy._L.azd_engine_agent_counters_per_agent(asy._h) == 1
    assert per_agent["EXPANSIONS"].max() <= calls and per_agent["EXPANSIONS"].sum() == asy.counters()["EXPANSIONS"]
    if "prof" not in az._lib.LIB_PATH:  # (the diagnostic build keeps per-agent blocks under the pool step too)
        assert pool._L.azd_engine_agent_counters_per_agent(pool._h) == 0
        with pytest.raises(RuntimeError, match="searcher waves"):
            pool.agent_counters()
    blocks = pool.agent_counters(allow_wave_blocks=True)
    cp, ca = pool.counters(), asy.counters()
    for k in MAIN_CTRS:
        assert cp[k] == ca[k], k
        if k not in ("MAX_FRONTIER", "MAX_DEPTH"):
            assert int(blocks[k].sum()) == cp[k], k
    # a fresh par_new clears the counters and with them the attribution flag
    from azdopt_amd import _lib
    parents, permitted = pool._roots(*roots)
    _lib.check(pool._L.azd_engine_par_new(pool._h, _lib.ptr(parents), _lib.ptr(permitted)), "par_new")
    assert pool._L.azd_engine_agent_counters_per_agent(pool._h) == 1


@pytest.mark.parametrize("g", [8, 16])
def test_evaluator_groups_equal_the_classic_form_and_the_async_step(az, monkeypatch, g):
    """Evaluator groups (pool_eval_group: g workgroups per batch, weight slices resident in LDS, activations exchanged through device
    memory) compute every prediction row with mlp_tile_task's MFMA sequence: identical rows, hence identical trees, counters and
    argmin as the classic evaluator workgroups and as the asynchronous step -- forced here on the 3 x 256 model (AZD_POOL_EVAL_GROUP)."""
    n, B, seed, calls = 19, 320, 12, 150
    space = az.ROTModifyParentsOnce(n)
    roots = space.generate_roots(seed, B)
    runs = []
    for form in ("group", "classic", "async"):
        if form == "group":
            monkeypatch.setenv("AZD_POOL_EVAL_GROUP", str(g))
        else:
            monkeypatch.setenv("AZD_POOL_EVAL_GROUP", "0")
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed)
        o = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=form != "async")
        imp = o.par_roll_out_episodes(TOL_REF, n_calls=calls)
        runs.append((o, imp))
    assert runs[0][0].step_form() == ("pool", "") and runs[0][0].pool_groups()[0] == g and runs[0][0].pool_groups()[1] >= 1
    assert runs[1][0].step_form() == ("pool", "") and runs[1][0].pool_groups() == (0, 0, 0)
    c = runs[0][0].counters()
    assert c["EVAL_ROWS"] == c["EXPANSIONS"] > 50 * B and c["FAILED"] == 0
    for o, imp in runs[1:]:
        same_engines(runs[0][0], runs[0][1], o, imp, range(0, B, 3))
        assert np.array_equal(runs[0][0].predictions().view(np.uint32), o.predictions().view(np.uint32))


def test_evaluator_groups_at_the_reference_shape_against_the_oracle(az, orc):
    """BASELINE configs[0] (the reference's own run: 512 agents, 304-512-1024-512-152 fp32, 04-c21-tree.rs:46-54): the engine forms
    evaluator groups by itself there (5.1 MB of weights per classic batch).  One launch of 300 calls of the product kernel against
    the oracle fed with the classic tile task's rows of its own states (debug_tile_forward), then an optimiser step and 100 more."""
    n, B, seed, calls = 19, 512, 9, 300
    space = az.ROTModifyParentsOnce(n)
    roots = space.generate_roots(seed, B)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(512, 1024, 512), seed=seed)
    opt = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=True)
    oe = orc.Engine(n, B, threads=8)
    oe.new_begin(*roots)
    oe.new_end(opt.predictions())
    _follow_with_the_oracle(opt, oe, TOL_REF, calls)
    members, groups, waves = opt.pool_groups()
    assert opt.step_form() == ("pool", "") and members >= 8 and groups >= 1 and waves in (1, 2, 4), (members, groups, waves)
    for i in range(0, B, 7):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")
    oe.observe(200)  # (optimizer/mod.rs:262-267: the training step leaves the ROOTS' vectors in state_vecs, on both sides)
    assert np.isfinite(opt.par_update_model(200))  # new weights: the next launch loads its LDS slices afresh
    _follow_with_the_oracle(opt, oe, TOL_REF, 100)
    for i in range(0, B, 7):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i} after the optimiser step")
