"""GPU parity tests of the dense-graph space (BASELINE configs[4]; space_dense.inc against oracle/dense_graph.inc): trees
(keys as action-id sets), state vectors, costs (lambda_1 as f64 bits, matching numbers), counters, observations and
argmin, bit for bit; N = 50 with the 512-wide bf16 model on the launch-per-phase form with the batched MFMA GEMM."""
import numpy as np
import pytest

from test_gpu_parity import MAIN_CTRS, assert_tree_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def az():
    import azdopt_amd
    assert azdopt_amd.device_count() > 0, "no MI355X visible"
    return azdopt_amd


def compare(opt, oe, agents, tag):
    assert np.array_equal(opt.state_vecs(), oe.state_vecs()), tag
    for i in agents:
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"{tag} agent {i}")
        sg, so = opt.agent_state(i), oe.agent_state(i)
        for k in so:
            assert np.array_equal(sg[k], so[k]), (tag, i, k, sg[k], so[k])  # lambda_1 compared as an f64: bit-identical procedure
    cg, co = opt.counters(), oe.counters()
    for k in MAIN_CTRS:
        assert cg[k] == co[k], (tag, k, cg[k], co[k])
    ag, ao = opt.argmin_data(), oe.argmin()
    assert ag.eval == ao["eval"] and ag.cost["lambda_1"] == ao["lambda1"] and len(ag.cost["matching"]) == ao["matching"], tag
    assert np.array_equal(ag.state["adj"].view(np.uint8), ao["parents"]) and np.array_equal(ag.state["permitted"], ao["permitted"]), tag


def run_dense_parity(az, orc, n, B, p, kmin, kmax, tol, steps, epochs, seed, check_every, max_slots=128, policy=False, pool=False, **caps):
    """policy: the epoch boundary is the drivers' modify_root on the DEVICE (par_reset_trees_policy: node choice in key
    order, path replay, fresh connected roots, re-drawn slots) against the oracle's, instead of fresh roots from the host.
    pool: the space's pool step (k_pool_search: agents multiplexed over searcher waves, running ahead of each other) with the
    fixed prediction stream served like a model's rows: collected, computed and handed back by launches on a second stream"""
    space = az.DenseGraphSpace(n, p, max_slots=max_slots)
    model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, seed)
    if pool:
        model = model.serve_from_pool_evaluators()
        caps = dict(caps, pool_step=True)
    roots = space.generate_roots(seed, B, kmin=kmin, kmax=kmax)
    opt = az.NablaOptimizer.par_new(space, roots, model, B, **caps)
    oe = orc.Engine(n, B, threads=8, dense=True, dense_p=p)
    oe.new_begin(*roots)
    call = 0
    oe.new_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
    compare(opt, oe, range(B), "par_new")
    for epoch in range(epochs):
        s = 0
        while s < steps:
            k = min(check_every, steps - s)
            ig = opt.par_roll_out_episodes(tol, n_calls=k)
            io = 0
            for _ in range(k):
                oe.rollout_begin(*tol)
                call += 1
                io += oe.rollout_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
            assert ig == io
            s += k
            compare(opt, oe, range(B), f"epoch {epoch} step {s}")
        if pool:
            assert opt.step_form() == ("pool", "")
            c = opt.counters()
            assert c["EVAL_ROWS"] == c["EXPANSIONS"] > 0  # every row went through the evaluator's launches
        else:
            assert opt.step_form()[0] == "per_call" and "dense-graph space" in opt.step_form()[1]
        sv, obs, w = opt.observe(2)
        oo, ow = oe.observe(2)
        assert np.array_equal(obs.view(np.uint32), oo.view(np.uint32)) and np.array_equal(w, ow) and np.array_equal(sv, oe.state_vecs())
        if policy:
            roots = oe.modify_roots(seed, epoch, 0, kmin, kmax)
            got = opt.modify_roots(seed, epoch, kmin, kmax)  # the policy alone: the new roots in the host's format
            assert np.array_equal(got[0], roots[0]) and np.array_equal(got[1], roots[1]), epoch
            opt.par_reset_trees_policy(seed, epoch, kmin, kmax)  # policy + reset without a host round trip
        else:
            roots = space.generate_roots(seed, B, epoch=epoch + 1, kmin=kmin, kmax=kmax)  # `modify_root`: fresh seeded roots from the host
            opt.par_reset_trees(roots)
        oe.reset_begin(*roots)
        call += 1
        oe.reset_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
        compare(opt, oe, range(B), f"epoch {epoch} reset")
    return opt.counters()


def test_dense_parity_small_every_step(az, orc):
    c = run_dense_parity(az, orc, 8, 12, 0.4, 2, 10, ([6, 3], 2), steps=40, epochs=2, seed=5, check_every=1)
    assert c["TRANSPOSITIONS"] > 0 and c["TERMINALS"] > 0


def test_dense_parity_n20(az, orc):
    c = run_dense_parity(az, orc, 20, 24, 0.2, 5, 60, ([50, 20, 10], 5), steps=120, epochs=2, seed=2, check_every=30)
    assert c["EXPANSIONS"] > 1000 and c["TRANSPOSITIONS"] > 0


def test_dense_parity_n50_reference_tolerances(az, orc):
    """BASELINE configs[4]'s N = 50 (E = 1225, ACTION 2450, STATE 3676) with up to 128 modifiable slots per root"""
    c = run_dense_parity(az, orc, 50, 32, 0.1, 5, 128, ([200, 50, 50], 25), steps=100, epochs=1, seed=1, check_every=50)
    assert c["EXPANSIONS"] > 2000 and c["MAX_DEPTH"] >= 3


@pytest.mark.parametrize("n,B,p,kmin,kmax,tol,steps,every,slots", [
    (8, 12, 0.4, 2, 10, ([6, 3], 2), 40, 1, 128),
    (20, 300, 0.2, 5, 60, ([50, 20, 10], 5), 120, 40, 128),
    (50, 96, 0.1, 5, 128, ([200, 50, 50], 25), 100, 50, 128),
    (50, 40, 0.1, 150, 250, ([200, 50, 50], 25), 60, 30, 256),
    (50, 24, 0.1, 400, 612, ([200, 50, 50], 25), 40, 20, 612),   # the drivers' image: up to E // 2 slots per root
    (50, 12, 0.08, 700, 1000, ([200, 50, 50], 25), 24, 12, 1024),
])
def test_dense_pool_step_against_the_oracle(az, orc, n, B, p, kmin, kmax, tol, steps, every, slots):
    """the space's CU-resident form: searcher workgroups only, the evaluator outside the kernel.  Trees, state vectors, costs,
    counters, argmin, observations and the device root policy against the oracle, as for the launch-per-phase form"""
    c = run_dense_parity(az, orc, n, B, p, kmin, kmax, tol, steps=steps, epochs=2, seed=7, check_every=every, max_slots=slots, policy=True, pool=True)
    assert c["EXPANSIONS"] > 100


def test_dense_pool_step_with_more_agents_than_waves_against_the_oracle(az, orc, monkeypatch):
    """four searcher workgroups (64 waves) for 300 agents: every agent queues for a wave and changes waves and CUs many times --
    the product regime of config E (8192 agents on 2048 waves) -- against the oracle, with the device root policy and a second epoch"""
    monkeypatch.setenv("AZD_DENSE_POOL_SEARCH_WGS", "4")
    c = run_dense_parity(az, orc, 20, 300, 0.2, 5, 60, ([50, 20, 10], 5), steps=120, epochs=2, seed=11, check_every=60, policy=True, pool=True)
    assert c["EXPANSIONS"] > 20000


def test_dense_pool_step_at_config_e_population_against_the_oracle(az, orc):
    """N = 50, 4096 agents (twice the searching waves), 100 calls per launch, device root policy, a second epoch: counters, state
    vectors, argmin, the new roots and a sample of the trees against the oracle"""
    n, B, p, kmin, kmax, seed, calls = 50, 4096, 0.1, 5, 128, 3, 100
    tol = ([200, 50, 50], 25)
    space = az.DenseGraphSpace(n, p, max_slots=128)
    model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, seed).serve_from_pool_evaluators()
    roots = space.generate_roots(seed, B, kmin=kmin, kmax=kmax)
    opt = az.NablaOptimizer.par_new(space, roots, model, B, pool_step=True)
    oe = orc.Engine(n, B, threads=16, dense=True, dense_p=p)
    oe.new_begin(*roots)
    call = 0
    oe.new_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
    for epoch in range(2):
        ig = opt.par_roll_out_episodes(tol, n_calls=calls)
        assert opt.step_form() == ("pool", "")
        io = 0
        for _ in range(calls):
            oe.rollout_begin(*tol)
            call += 1
            io += oe.rollout_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
        assert ig == io
        assert opt.pool_split()[1] * 16 < B  # fewer searching waves than agents
        compare(opt, oe, range(0, B, 97), f"epoch {epoch}")
        new_roots = oe.modify_roots(seed, epoch, 0, kmin, kmax)
        got = opt.modify_roots(seed, epoch, kmin, kmax)
        assert np.array_equal(got[0], new_roots[0]) and np.array_equal(got[1], new_roots[1]), epoch
        opt.par_reset_trees_policy(seed, epoch, kmin, kmax)
        oe.reset_begin(*new_roots)
        call += 1
        oe.reset_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))


@pytest.mark.parametrize("B,max_slots,kmax", [(640, 128, 128), (256, 612, 612)])
def test_dense_pool_step_with_the_bf16_model_equals_the_launch_per_phase_form(az, monkeypatch, B, max_slots, kmax):
    """config E's model on the pool step: the rows the searchers post are gathered into the batched bf16 GEMMs (k_gemm16 with row
    lists, whatever batch a row lands in) and scattered back -- same trees, counters, argmin, prediction rows and training step
    as one launch per phase"""
    n, seed, calls = 50, 5, 70
    tol = ([200, 50, 50], 25)
    space = az.DenseGraphSpace(n, 0.1, max_slots=max_slots)
    roots = space.generate_roots(seed, B, kmin=5, kmax=kmax)
    runs = []
    for pool in (True, False):
        if not pool:
            monkeypatch.setenv("AZD_DENSE_NO_POOL", "1")
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(512, 512, 512), seed=seed, dtype="bf16")
        o = az.NablaOptimizer.par_new(space, roots, model, B, prediction_capacity=131072)
        imp = o.par_roll_out_episodes(tol, n_calls=calls)
        assert o.step_form()[0] == ("pool" if pool else "per_call_graph"), o.step_form()
        loss = o.par_update_model(3)
        o.par_reset_trees_policy(seed, 0, 5, kmax)
        imp2 = o.par_roll_out_episodes(tol, n_calls=30)
        runs.append((o, imp, imp2, loss))
    monkeypatch.delenv("AZD_DENSE_NO_POOL")
    (o0, i0, j0, l0), (o1, i1, j1, l1) = runs
    assert (i0, j0) == (i1, j1) and l0 == l1
    c0, c1 = o0.counters(), o1.counters()
    for k in MAIN_CTRS:
        assert c0[k] == c1[k], k
    assert c0["EVAL_ROWS"] == c0["EXPANSIONS"]
    for i in range(0, B, 5):
        assert_tree_equal(o0.get_tree(i), o1.get_tree(i), f"agent {i}")
    a0, a1 = o0.argmin_data(), o1.argmin_data()
    assert a0.eval == a1.eval and a0.agent == a1.agent and a0.node == a1.node
    assert np.array_equal(o0.state_vecs(), o1.state_vecs())
    assert np.array_equal(o0.predictions().view(np.uint32), o1.predictions().view(np.uint32))


@pytest.mark.parametrize("n,abort_at,calls", [(20, 3, 60), (12, 40, 1100)])
def test_dense_pool_abort_is_completed_by_the_launch_per_phase_kernels(az, monkeypatch, n, abort_at, calls):
    """the abort flag of a dense pool launch (test hook: raised behind the k-th batch the evaluator hands back) no longer fails the
    call: the launch-per-phase kernels complete the launch agent by agent from where each one stands -- some through all their calls,
    some waiting for a row that will not come, some never taken -- with the results of an undisturbed run; 1100 calls = a second
    chunk of the log, which the launch-per-phase form runs too"""
    B, seed = 300, 11
    tol = ([50, 20, 10], 5)
    space = az.DenseGraphSpace(n, 0.2)
    # (the long case: roots with few slots -- small trees that are exhausted long before the calls are, so that the cascade's frontier stays narrow)
    roots = space.generate_roots(seed, B, kmin=3, kmax=7) if calls > 800 else space.generate_roots(seed, B)
    caps = {}

    def mk():
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(128, 128), seed=seed, dtype="bf16")
        return az.NablaOptimizer.par_new(space, roots, model, B, pool_step=True, **caps)

    monkeypatch.setenv("AZD_DENSE_NO_POOL", "1")
    ref = mk()
    imp_ref = ref.par_roll_out_episodes(tol, n_calls=calls)
    monkeypatch.delenv("AZD_DENSE_NO_POOL")
    monkeypatch.setenv("AZD_POOL_DEBUG_ABORT_CALL", str(abort_at))
    opt = mk()
    imp = opt.par_roll_out_episodes(tol, n_calls=calls)
    form, why = opt.step_form()
    assert form.startswith("per_call") and "aborted" in why, (form, why)
    monkeypatch.delenv("AZD_POOL_DEBUG_ABORT_CALL")
    assert imp == imp_ref
    c0, c1 = opt.counters(), ref.counters()
    for k in MAIN_CTRS:
        assert c0[k] == c1[k], k
    for i in range(B):
        assert_tree_equal(opt.get_tree(i), ref.get_tree(i), f"agent {i}")
    a0, a1 = opt.argmin_data(), ref.argmin_data()
    assert a0.eval == a1.eval and a0.agent == a1.agent and a0.node == a1.node
    assert np.array_equal(opt.state_vecs(), ref.state_vecs())
    # usable afterwards, on the launch-per-phase form
    assert opt.par_roll_out_episodes(tol, n_calls=10) == ref.par_roll_out_episodes(tol, n_calls=10)
    assert opt.step_form()[0].startswith("per_call")
    for i in range(0, B, 11):
        assert_tree_equal(opt.get_tree(i), ref.get_tree(i), f"agent {i} afterwards")


def test_dense_roots_are_validated(az):
    space = az.DenseGraphSpace(12, 0.3)
    model = az.TrivialModel(space.STATE_DIM, space.ACTION_DIM)
    adj, slots = space.generate_roots(0, 4)
    bad = adj.copy().view(np.uint64).reshape(4, 12)
    bad[1, 3] ^= np.uint64(1 << 7)  # not symmetric any more
    with pytest.raises(az.AzdError):
        az.NablaOptimizer.par_new(space, (bad.view(np.uint8).reshape(4, -1), slots), model, 4)
    lone = np.zeros((4, 12), np.uint64)  # not connected
    with pytest.raises(az.AzdError):
        az.NablaOptimizer.par_new(space, (lone.view(np.uint8).reshape(4, -1), slots), model, 4)
    opt = az.NablaOptimizer.par_new(space, (adj, slots), model, 4)
    with pytest.raises(az.AzdError):  # more slots than the engine's keys hold (max_slots = 128 -> 64 * 2 ranks)
        opt.par_reset_trees_policy(0, 0, 5, 200)
    many = az.DenseGraphSpace(30, 0.2)  # E = 435: a root with 300 slots does not fit the default key width
    with pytest.raises(az.AzdError):
        az.NablaOptimizer.par_new(many, many.generate_roots(0, 2, kmin=300, kmax=300), az.TrivialModel(many.STATE_DIM, many.ACTION_DIM), 2)


@pytest.mark.parametrize("n,B,p,kmin,kmax,max_slots", [(8, 12, 0.4, 2, 10, 128), (8, 10, 0.5, 6, 6, 128), (20, 16, 0.2, 5, 150, 190)])
def test_dense_device_root_policy(az, orc, n, B, p, kmin, kmax, max_slots):
    """the drivers' modify_root (04-c21-tree.rs:172-206) for this space on the device: same new roots as the oracle's policy --
    chosen node (BTreeMap order of the action-id sets = order of the rank sets), replayed path, fresh connected G(n, p) when
    the root is at its slot limit (kmin == kmax forces that branch), re-drawn slot masks -- and the same trees afterwards"""
    c = run_dense_parity(az, orc, n, B, p, kmin, kmax, ([6, 3], 2), steps=30, epochs=4, seed=9, check_every=15, max_slots=max_slots, policy=True)
    assert c["EXPANSIONS"] > 0


@pytest.mark.parametrize("max_slots,kmin,kmax", [(612, 300, 612), (1024, 700, 1000), (256, 129, 256)])
def test_dense_n50_beyond_128_slots(az, orc, max_slots, kmin, kmax):
    """roots with up to E / 2 = 612 modifiable slots (the drivers' image: up to half of the action space), and beyond: nodes
    hold up to 64 KW predictions (KW = 4 / 10 / 16 key words), selection runs chunk by chunk (select_big), the keys are wider"""
    c = run_dense_parity(az, orc, 50, 6, 0.1, kmin, kmax, ([30, 10, 5], 3), steps=60, epochs=2, seed=4, check_every=30, max_slots=max_slots,
                         policy=True, prediction_capacity=131072)
    assert c["EXPANSIONS"] > 300 and c["SUM_ACTIONS"] > 100 * c["SELECT_CALLS"]


def test_dense_n50_with_the_512_wide_bf16_model(az, orc):
    """config E's model: 3676 -> 512 -> 512 -> 512 -> 2450 with bf16 storage on the batched MFMA GEMM
    (v_mfma_f32_32x32x16_bf16), one launch per phase; the oracle is fed the GPU's prediction rows and must grow the
    same trees; then one optimiser step on the f32 master weights"""
    n, B, seed = 50, 48, 3
    tol = ([200, 50, 50], 25)
    space = az.DenseGraphSpace(n, 0.1)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(512, 512, 512), seed=seed, dtype="bf16")
    roots = space.generate_roots(seed, B)
    opt = az.NablaOptimizer.par_new(space, roots, model, B)
    oe = orc.Engine(n, B, threads=8, dense=True)
    oe.new_begin(*roots)
    oe.new_end(opt.predictions())
    for s in range(40):
        opt.par_roll_out_episodes(tol)
        oe.rollout_begin(*tol)
        assert np.array_equal(opt.state_vecs(), oe.state_vecs())
        oe.rollout_end(opt.predictions())
    assert opt.step_form()[0] == "per_call"
    for i in range(B):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")
    # the engine handed the evaluator the state vectors as bf16 rows written by the search kernels themselves (write_vec16);
    # the evaluator's own entry point converts the f32 rows: same prediction rows bit for bit
    again = np.zeros((B, space.ACTION_DIM), np.float32)
    model.write_predictions(opt.state_vecs(), again)
    assert np.array_equal(again.view(np.uint32), opt.predictions().view(np.uint32))
    loss = opt.par_update_model(2)
    assert np.isfinite(loss) and loss >= 0


def test_hipgraph_replay_of_the_per_call_form_equals_launch_by_launch(az, monkeypatch):
    """with the MLP evaluator the launches of a call (roll-out, GEMMs, add_actions, argmin) are captured once in a
    hipGraph and replayed per call: same trees, counters, argmin and improvement count as launching them one by one.
    Third run: the population cut into three sub-populations, each with a stream and a graph of its own, running their calls
    independently (the per-call argmin replayed from the candidate log): the form large populations take."""
    n, B, seed = 20, 40, 6
    tol = ([50, 20, 10], 5)
    space = az.DenseGraphSpace(n, 0.2)
    roots = space.generate_roots(seed, B)
    runs = []
    for graph in (True, False, "subs"):
        if graph == "subs":
            monkeypatch.setenv("AZD_PER_CALL_STREAMS", "3")
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(128, 128), seed=seed, dtype="bf16")
        o = az.NablaOptimizer.par_new(space, roots, model, B)
        if graph:
            imp = o.par_roll_out_episodes(tol, n_calls=60)
            assert o.step_form()[0] == "per_call_graph"
            imp += o.par_roll_out_episodes(tol, n_calls=30)  # the cached graph again
        else:
            imp = sum(o.par_roll_out_episodes(tol, n_calls=1) for _ in range(90))
            assert o.step_form()[0] == "per_call"
        runs.append((o, imp))
    monkeypatch.delenv("AZD_PER_CALL_STREAMS")
    (o2, i2) = runs.pop()
    assert i2 == runs[0][1]
    c0, c2 = runs[0][0].counters(), o2.counters()
    for k in MAIN_CTRS:
        assert c0[k] == c2[k], k
    for i in range(B):
        assert_tree_equal(runs[0][0].get_tree(i), o2.get_tree(i), f"sub-populations, agent {i}")
    a0, a2 = runs[0][0].argmin_data(), o2.argmin_data()
    assert a0.eval == a2.eval and a0.agent == a2.agent and a0.node == a2.node
    assert np.array_equal(runs[0][0].state_vecs(), o2.state_vecs())
    (o0, i0), (o1, i1) = runs
    assert i0 == i1
    c0, c1 = o0.counters(), o1.counters()
    for k in MAIN_CTRS:
        assert c0[k] == c1[k], k
    for i in range(B):
        assert_tree_equal(o0.get_tree(i), o1.get_tree(i), f"agent {i}")
    assert o0.argmin_data().eval == o1.argmin_data().eval
    # c21 on the launch-per-phase form takes the same path
    sp = az.ROTModifyParentsOnce(19)
    m = az.ActionModel(64, sp.STATE_DIM, sp.ACTION_DIM, hidden=(256, 256, 256), seed=1)
    o = az.NablaOptimizer.par_new(sp, sp.generate_roots(1, 64), m, 64, persistent=False)
    o.par_roll_out_episodes(([200, 50, 50], 25), n_calls=10)
    assert o.step_form()[0] == "per_call_graph"


@pytest.mark.parametrize("env", [{"AZD_DENSE_POOL_WAVES": "8", "AZD_DENSE_POOL_SEARCH_WGS": "256"}, {"AZD_DENSE_POOL_WAVES": "2"}, {"AZD_DENSE_POOL_SEARCH_WGS": "250"}])
def test_dense_pool_knobs_that_leave_no_room_for_the_gemm_launches_are_refused(az, monkeypatch, env):
    """Round-4 verdict, 7c: AZD_DENSE_POOL_WAVES = 8 / 6 used to be accepted, let the searcher workgroups cover the chip, and cost a
    4-s wait bound, an abort and the engine's demotion to the launch-per-phase form.  A setting that would leave the evaluator's
    GEMM launches less than a quarter of the CUs (or a wave count outside 4..16) is now refused before anything is launched."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n, B, seed = 50, 640, 5
    space = az.DenseGraphSpace(n, 0.1, max_slots=128)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(512, 512, 512), seed=seed, dtype="bf16")
    o = az.NablaOptimizer.par_new(space, space.generate_roots(seed, B), model, B, prediction_capacity=131072)
    with pytest.raises(az.AzdError) as ei:
        o.par_roll_out_episodes(([200, 50, 50], 25), n_calls=5)
    assert ei.value.status == 1 and "AZD_DENSE_POOL" in str(ei.value)  # AZD_ERR_INVALID_ARGUMENT, naming the knob
    for k in env:
        monkeypatch.delenv(k)
    o.par_roll_out_episodes(([200, 50, 50], 25), n_calls=5)  # (the engine is not demoted by the refusal)
    assert o.step_form()[0] == "pool"
