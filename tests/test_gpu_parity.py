"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Bit-exact for everything integer / index / f32-selection related (tree topology, counters,
visit counts n_t, state vectors, observations, argmin, root policy); stated tolerances for the
MLP.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_REF = ([200, 50, 50], 25)  # 04-c21-tree.rs:136-138


@pytest.fixture(scope="module")
def az():
    import azdopt_amd
    assert azdopt_amd.device_count() > 0, "no MI355X visible"
    return azdopt_amd


def assert_tree_equal(tg, to, tag=""):
    for f in to.FIELDS:
        a, b = getattr(tg, f), getattr(to, f)
        assert a.shape == b.shape, (tag, f, a.shape, b.shape)
        if a.dtype.kind == "f":
            same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
        else:
            same = np.array_equal(a.astype(np.int64), b.astype(np.int64))
        if not same:
            idx = np.argwhere(a != b)
            raise AssertionError(f"{tag} field {f}: first mismatch at {idx[:3].tolist()} gpu={a[tuple(idx[0])]} oracle={b[tuple(idx[0])]}")


MAIN_CTRS = ("EXPANSIONS", "TERMINALS", "TRANSPOSITIONS", "VISITED_STEPS", "SELECT_CALLS", "SUM_DEG", "SUM_ACTIONS",
             "CASCADE_NODES", "NEW_PREDS", "ROOT_EXHAUSTED", "MAX_DEPTH", "CURIOSITY_PAIRS", "FAILED")


def test_f32_primitives_bit_exact(az):
    """sqrt(|x - y|) and x - (x - y) on the GPU equal numpy f32 bit for bit: uniform values, subnormal
    differences, 1-ulp neighbours, exact zero, and every exponent (the selection rule's curiosity sum
    is built from these two operations)."""
    rng = np.random.default_rng(0)
    n = 1 << 18
    x = rng.random(2 * n, dtype=np.float32)
    x[: n // 8] *= np.float32(1e-38)  # subnormal differences
    x[n // 8: n // 4] = np.nextafter(x[n // 8: n // 4], np.float32(2))  # 1-ulp neighbours
    x[2 * 100] = x[2 * 100 + 1]  # exact zero
    # second half: |x - y| sweeps all finite exponents with random mantissas (y = 0)
    bits = rng.integers(0, 0x7F800000, n, dtype=np.uint32)
    x[n::2] = bits[: n // 2].view(np.float32)
    x[n + 1::2] = 0
    out = np.zeros(4 * n, np.float32)
    L = az.lib()
    from azdopt_amd import _lib
    _lib.check(L.azd_debug_probe_math(0, _lib.ptr(x), _lib.ptr(out), n), "probe")
    a, b = x[0::2], x[1::2]
    want_sqrt = np.sqrt(np.abs(a - b))
    want_sub = a - (a - b)
    assert np.array_equal(out[0::4].view(np.uint32), want_sqrt.view(np.uint32))  # the kernels' azd_sqrt
    assert np.array_equal(out[1::4].view(np.uint32), want_sqrt.view(np.uint32))  # sqrtf in a kernel of its own
    assert np.array_equal(out[2::4].view(np.uint32), want_sqrt.view(np.uint32))  # (float)sqrt((double)x)
    assert np.array_equal(out[3::4].view(np.uint32), want_sub.view(np.uint32))


def test_hash_stream_matches_oracle(az, orc):
    m = az.HashStreamModel(304, 152, seed=5, first_agent=10)
    buf = np.zeros((7, 152), np.float32)
    for call in range(3):
        m.write_predictions(np.zeros((7, 304), np.float32), buf)
        assert np.array_equal(buf, orc.hash_predictions(5, 10, 7, 152, call))


def run_parity(az, orc, n, B, kmin, kmax, tol, steps, epochs, seed, n_obs_tol, check_every=1, sample=None,
               first_agent=0, threads=8):
    space = az.ROTModifyParentsOnce(n)
    model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, seed, first_agent)
    parents, permitted = space.generate_roots(seed, B, first_agent=first_agent, kmin=kmin, kmax=kmax)
    po, mo = orc.gen_roots(seed, 0, first_agent, B, n, kmin, kmax)
    assert np.array_equal(parents, po) and np.array_equal(permitted, mo)
    opt = az.NablaOptimizer.par_new(space, (parents, permitted), model, B, first_agent=first_agent)
    oe = orc.Engine(n, B, threads=threads)
    oe.new_begin(parents, permitted)
    call = 0
    oe.new_end(orc.hash_predictions(seed, first_agent, B, space.ACTION_DIM, call))
    agents = range(B) if sample is None else sample

    def compare(tag):
        assert np.array_equal(opt.state_vecs(), oe.state_vecs()), tag
        for i in agents:
            assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"{tag} agent {i}")
            sg, so = opt.agent_state(i), oe.agent_state(i)
            for k in so:
                assert np.array_equal(sg[k], so[k]), (tag, i, k, sg[k], so[k])
        ag, ao = opt.argmin_data(), oe.argmin()
        assert np.array_equal(ag.state["parents"], ao["parents"]) and np.array_equal(ag.state["permitted"], ao["permitted"])
        assert ag.eval == ao["eval"] and ag.cost["lambda_1"] == ao["lambda1"] and len(ag.cost["matching"]) == ao["matching"]
        cg, co = opt.counters(), oe.counters()
        for k in MAIN_CTRS:
            assert cg[k] == co[k], (tag, k, cg[k], co[k])

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
            assert improved_g == improved_o
            s += k
            compare(f"epoch {epoch} step {s}")
        sv, obs, w = opt.observe(n_obs_tol)
        oo, ow = oe.observe(n_obs_tol)
        assert np.array_equal(obs.view(np.uint32), oo.view(np.uint32)) and np.array_equal(w, ow)
        assert np.array_equal(sv, oe.state_vecs())
        ro = oe.modify_roots(seed, epoch, first_agent, kmin, kmax)
        for device in (True, False):  # device kernel and host C++ restatement of the root policy
            rg = opt.c21_modify_roots(seed, epoch, kmin, kmax, device=device)
            assert np.array_equal(rg[0], ro[0]) and np.array_equal(rg[1], ro[1]), (epoch, device)
        if epoch % 2 == 0:
            opt.par_reset_trees_c21(seed, epoch, kmin, kmax)  # policy + reset without a host round trip
        else:
            opt.par_reset_trees(rg)
        oe.reset_begin(*ro)
        call += 1
        oe.reset_end(orc.hash_predictions(seed, first_agent, B, space.ACTION_DIM, call))
        compare(f"epoch {epoch} reset")
    return opt.counters()


def test_parity_n5_every_step(az, orc):
    c = run_parity(az, orc, 5, 8, 1, 5, ([3, 2], 1), steps=12, epochs=3, seed=11, n_obs_tol=1)
    assert c["TERMINALS"] > 0 and c["ROOT_EXHAUSTED"] > 0


def test_parity_n8_all_branches(az, orc):
    c = run_parity(az, orc, 8, 16, 2, 10, ([4, 2, 2], 1), steps=60, epochs=2, seed=3, n_obs_tol=2)
    for k in ("EXPANSIONS", "TERMINALS", "TRANSPOSITIONS", "VISITED_STEPS", "CASCADE_NODES"):
        assert c[k] > 0


def test_parity_n13_two_key_words(az, orc):
    run_parity(az, orc, 13, 8, 3, 32, ([6, 3], 2), steps=40, epochs=1, seed=2, n_obs_tol=3, check_every=10)


def test_parity_n22_four_key_words(az, orc):
    run_parity(az, orc, 22, 4, 5, 104, ([6, 3], 2), steps=30, epochs=1, seed=4, n_obs_tol=3, check_every=10)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_parity_n19_reference_hyperparameters(az, orc, seed):
    # the reference's own c21 configuration (04-c21-tree.rs:33-54,133-138) at B = 64
    run_parity(az, orc, 19, 64, 5, 76, TOL_REF, steps=200, epochs=1, seed=seed, n_obs_tol=200, check_every=50)


def test_parity_full_size_b4096(az, orc):
    """BASELINE config B (4096 agents, N = 19): global counters, state vectors and argmin equal the
    oracle's after every block of steps; full tree equality on a sample of agents; then visit-count
    (n_t) L-inf = 0 over all trees' roots' children via the observation buffers."""
    sample = list(range(0, 4096, 97))
    run_parity(az, orc, 19, 4096, 5, 76, TOL_REF, steps=120, epochs=1, seed=0, n_obs_tol=20, check_every=40,
               sample=sample, threads=16)


def test_sharded_agents_are_shard_invariant(az, orc):
    """An engine that owns agents [first, first+B) of a larger population reproduces exactly those
    agents' trees (the multi-GPU partition needs no data-path collective)."""
    run_parity(az, orc, 19, 32, 5, 76, TOL_REF, steps=60, epochs=1, seed=7, n_obs_tol=10, check_every=30,
               first_agent=4096 + 32)


def test_external_model_split_phase(az, orc):
    """Engine without evaluator driven through *_begin/_end with host predictions (any NablaModel)."""
    n, B, seed = 19, 16, 9
    space = az.ROTModifyParentsOnce(n)
    parents, permitted = space.generate_roots(seed, B)
    opt = az.NablaOptimizer(space, None, B)
    oe = orc.Engine(n, B, threads=4)
    rng = np.random.default_rng(1)
    opt.par_new_begin(parents, permitted)
    oe.new_begin(parents, permitted)
    h = rng.random((B, space.ACTION_DIM), dtype=np.float32)
    opt.par_new_end(h)
    oe.new_end(h)
    for s in range(50):
        opt.roll_out_begin(TOL_REF)
        oe.rollout_begin(*TOL_REF)
        assert np.array_equal(opt.state_vecs(), oe.state_vecs())
        h = rng.random((B, space.ACTION_DIM), dtype=np.float32)
        assert opt.roll_out_end(h) == oe.rollout_end(h)
    for i in range(B):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")
    with pytest.raises(az.AzdError):
        opt.par_roll_out_episodes(TOL_REF)  # no evaluator: must fail loudly, not fall back


def test_capacity_overflow_is_reported(az):
    space = az.ROTModifyParentsOnce(19)
    model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, 0)
    opt = az.NablaOptimizer.par_new(space, space.generate_roots(0, 8), model, 8, node_capacity=16, arc_capacity=64,
                                    prediction_capacity=4096)
    with pytest.raises(az.AzdError) as ei:
        opt.par_roll_out_episodes(TOL_REF, n_calls=200)
    assert ei.value.status == 4  # AZD_ERR_CAPACITY
    assert opt.counters()["FAILED"] > 0


def test_invalid_roots_rejected(az):
    space = az.ROTModifyParentsOnce(19)
    model = az.TrivialModel(space.STATE_DIM, space.ACTION_DIM)
    parents, permitted = space.generate_roots(0, 4)
    parents = parents.copy()
    parents[0, 5] = 7  # parent >= child
    with pytest.raises(az.AzdError):
        az.NablaOptimizer.par_new(space, (parents, permitted), model, 4)


def test_trivial_model_semantics(az, orc):
    """TrivialModel leaves h_theta untouched: zeros after par_new (model/mod.rs:13, optimizer/mod.rs:71)."""
    n, B = 19, 8
    space = az.ROTModifyParentsOnce(n)
    model = az.TrivialModel(space.STATE_DIM, space.ACTION_DIM)
    parents, permitted = space.generate_roots(3, B)
    opt = az.NablaOptimizer.par_new(space, (parents, permitted), model, B)
    oe = orc.Engine(n, B)
    z = np.zeros((B, space.ACTION_DIM), np.float32)
    oe.new_begin(parents, permitted)
    oe.new_end(z)
    for _ in range(30):
        opt.par_roll_out_episodes(TOL_REF)
        oe.rollout_begin(*TOL_REF)
        oe.rollout_end(z)
    for i in range(B):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")
    assert opt.par_update_model(200) == 0.0


# ------------------------------------------------------------------ evaluator (MLP)
MLP_ATOL = 2e-5  # fp32 MFMA vs CPU fp32: summation order only


def test_mlp_forward_matches_cpu(az, orc):
    import torch
    for dims, B in (((304, 256, 256, 256, 152), 4096), ((304, 512, 1024, 512, 152), 512), ((10, 24, 5), 37)):
        m = az.ActionModel(B, dims[0], dims[-1], hidden=dims[1:-1], seed=3)
        om = orc.Mlp(dims, seed=3, threads=8)
        assert np.array_equal(m.get_params(), om.get_params())  # same seeded init on both sides
        rng = np.random.default_rng(0)
        x = (rng.random((B, dims[0])) < 0.3).astype(np.float32)
        y = np.zeros((B, dims[-1]), np.float32)
        m.write_predictions(x, y)
        yo = om.forward(x)
        assert np.max(np.abs(y - yo)) < MLP_ATOL
        # independent torch fp32 reference of the same op
        p = torch.from_numpy(m.get_params())
        t = torch.from_numpy(x)
        off = 0
        for l in range(len(dims) - 1):
            W = p[off: off + dims[l] * dims[l + 1]].reshape(dims[l + 1], dims[l]); off += W.numel()
            b = p[off: off + dims[l + 1]]; off += b.numel()
            t = t @ W.T + b
            t = torch.sigmoid(t) if l == len(dims) - 2 else torch.relu(t)
        assert np.max(np.abs(y - t.numpy())) < MLP_ATOL


@pytest.mark.parametrize("B", [512, 1300])  # 1300 rows: the gradient reductions run as three ragged parts of the batch
def test_mlp_update_matches_cpu_and_torch(az, orc, B):
    import torch
    dims = (304, 256, 256, 256, 152)
    cfg = dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, l2=1e-6)
    m = az.ActionModel(B, dims[0], dims[-1], hidden=dims[1:-1], seed=5, **cfg)
    om = orc.Mlp(dims, lr=cfg["lr"], l2=cfg["l2"], seed=5, threads=8)
    layers = []
    p0 = torch.from_numpy(m.get_params().copy())
    off = 0
    for l in range(len(dims) - 1):
        lin = torch.nn.Linear(dims[l], dims[l + 1])
        n = dims[l] * dims[l + 1]
        lin.weight.data = p0[off: off + n].reshape(dims[l + 1], dims[l]).clone(); off += n
        lin.bias.data = p0[off: off + dims[l + 1]].clone(); off += dims[l + 1]
        layers += [lin, torch.nn.Sigmoid() if l == len(dims) - 2 else torch.nn.ReLU()]
    net = torch.nn.Sequential(*layers)
    optim = torch.optim.Adam(net.parameters(), lr=cfg["lr"], betas=cfg["betas"], eps=cfg["eps"], weight_decay=cfg["l2"])
    rng = np.random.default_rng(0)
    for it in range(3):
        x = (rng.random((B, dims[0])) < 0.3).astype(np.float32)
        obs = rng.random((B, dims[-1]), dtype=np.float32)
        w = (rng.random((B, dims[-1])) < 0.05).astype(np.float32)
        lg = m.update_model(x, obs, w)
        lo = om.update(x, obs, w)
        tw = torch.from_numpy(w) / torch.from_numpy(w).sum()
        loss = (tw * (net(torch.from_numpy(x)) - torch.from_numpy(obs)) ** 2).sum()
        optim.zero_grad(); loss.backward(); optim.step()
        assert abs(lg - lo) < 1e-5 * max(1.0, abs(lo)) and abs(lg - loss.item()) < 1e-5 * max(1.0, abs(lo))
        pt = torch.cat([q.data.reshape(-1) for lin in net if isinstance(lin, torch.nn.Linear) for q in (lin.weight, lin.bias)]).numpy()
        pg = m.get_params()
        # Adam's first steps move every parameter by ~lr regardless of gradient scale: compare on that scale
        assert np.max(np.abs(pg - om.get_params())) < 0.05 * cfg["lr"], it
        assert np.max(np.abs(pg - pt)) < 0.05 * cfg["lr"], it


@pytest.mark.parametrize("persistent,async_step", [(True, False), (True, True), (False, False)])
def test_topology_parity_with_mlp_predictions(az, orc, persistent, async_step):
    """End-to-end with the real MLP (in-kernel MFMA evaluator of the persistent step, or the
    launch-per-phase form): feed the GPU's own predictions to the oracle; trees must match
    bit-for-bit (MLP parity is tolerance-checked separately)."""
    n, B, seed = 19, 72, 4  # 72 = 4.5 workgroups of 16 agents: exercises the ragged last group
    space = az.ROTModifyParentsOnce(n)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=seed)
    parents, permitted = space.generate_roots(seed, B)
    opt = az.NablaOptimizer.par_new(space, (parents, permitted), model, B, persistent=persistent, async_step=async_step)
    oe = orc.Engine(n, B, threads=8)
    oe.new_begin(parents, permitted)
    oe.new_end(opt.predictions())
    for s in range(80):
        opt.par_roll_out_episodes(TOL_REF)
        oe.rollout_begin(*TOL_REF)
        assert np.array_equal(opt.state_vecs(), oe.state_vecs())
        oe.rollout_end(opt.predictions())
    for i in range(B):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")
    loss = opt.par_update_model(5)
    assert np.isfinite(loss)


def test_persistent_step_equals_launch_per_phase(az, orc):
    """The CU-resident persistent step and the launch-per-phase form are the same computation:
    identical trees / counters / argmin with the fixed prediction stream (many calls per launch),
    and in-kernel MLP rows within the MLP tolerance of the batched GEMM path."""
    n, B, seed = 19, 200, 6
    space = az.ROTModifyParentsOnce(n)
    roots = space.generate_roots(seed, B)
    def same_engines(o1, i1, o2, i2):
        assert i1 == i2
        c1, c2 = o1.counters(), o2.counters()
        for k in MAIN_CTRS:
            assert c1[k] == c2[k], k
        for i in range(B):
            t1, t2 = o1.get_tree(i), o2.get_tree(i)
            for f in t1.FIELDS:
                a, b = getattr(t1, f), getattr(t2, f)
                assert a.shape == b.shape and np.array_equal(a.view(np.uint32) if a.dtype.kind == "f" else a, b.view(np.uint32) if b.dtype.kind == "f" else b), (i, f)
        a1, a2 = o1.argmin_data(), o2.argmin_data()
        assert a1.eval == a2.eval and a1.agent == a2.agent and a1.node == a2.node

    opts = []
    for persistent, async_step in ((True, False), (True, True), (False, False)):
        model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, seed)
        o = az.NablaOptimizer.par_new(space, roots, model, B, persistent=persistent, async_step=async_step)
        imp = o.par_roll_out_episodes(TOL_REF, n_calls=150)
        opts.append((o, imp, model))
    same_engines(opts[0][0], opts[0][1], opts[1][0], opts[1][1])
    same_engines(opts[0][0], opts[0][1], opts[2][0], opts[2][1])
    # with the MLP, the asynchronous evaluator service and the barrier form run the same MFMA
    # sequence per output element: identical trees after many calls in one launch
    mopts = []
    for async_step in (True, False):
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=9)
        o = az.NablaOptimizer.par_new(space, roots, model, B, async_step=async_step)
        imp = o.par_roll_out_episodes(TOL_REF, n_calls=120)
        mopts.append((o, imp, model))
    same_engines(mopts[0][0], mopts[0][1], mopts[1][0], mopts[1][1])
    assert mopts[0][0].counters()["EVAL_ROWS"] == mopts[0][0].counters()["EXPANSIONS"]
    # MLP rows: same states, both evaluator forms
    preds = []
    for persistent in (True, False):
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=2)
        o = az.NablaOptimizer.par_new(space, roots, model, B, persistent=persistent)
        o.par_roll_out_episodes(TOL_REF, n_calls=1)
        preds.append((o.state_vecs(), o.predictions()))
    assert np.array_equal(preds[0][0], preds[1][0])
    assert np.max(np.abs(preds[0][1] - preds[1][1])) < MLP_ATOL


def test_async_epoch_is_reproducible_at_full_size(az):
    """BASELINE config B for a whole epoch (4096 agents, 800 calls in one launch, the MLP in the kernel): which rows
    share a batch and which wave computes which tile depends on timing, the results must not -- two runs
    give the same counters, argmin, predictions and trees, and so does the barrier step."""
    n, B, calls = 19, 4096, 800
    space = az.ROTModifyParentsOnce(n)
    roots = space.generate_roots(0, B)
    runs = []
    for kw in (dict(pool_step=False), dict(pool_step=False), dict(async_step=False), dict()):  # async twice, barrier, the engine's default (pool)
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(256, 256, 256), seed=0)
        o = az.NablaOptimizer.par_new(space, roots, model, B, **kw)
        imp = o.par_roll_out_episodes(TOL_REF, n_calls=calls)
        runs.append((o, imp))
    o0, i0 = runs[0]
    assert [o.step_form()[0] for o, _ in runs] == ["async", "async", "barrier", "pool"]
    c0, a0 = o0.counters(), o0.argmin_data()
    assert c0["EXPANSIONS"] > 0.8 * B * calls and c0["EVAL_ROWS"] == c0["EXPANSIONS"]
    for o, imp in runs[1:]:
        c, am = o.counters(), o.argmin_data()
        assert imp == i0
        for k in MAIN_CTRS:
            assert c[k] == c0[k], k
        assert (am.eval, am.agent, am.node) == (a0.eval, a0.agent, a0.node)
        assert np.array_equal(o.state_vecs(), o0.state_vecs())
        assert np.array_equal(o.predictions().view(np.uint32), o0.predictions().view(np.uint32))
        for i in range(0, B, 257):
            t1, t2 = o.get_tree(i), o0.get_tree(i)
            for f in t1.FIELDS:
                x, y = getattr(t1, f), getattr(t2, f)
                assert x.shape == y.shape and np.array_equal(x.view(np.uint32) if x.dtype.kind == "f" else x,
                                                             y.view(np.uint32) if y.dtype.kind == "f" else y), (i, f)


def test_mlp_update_is_deterministic(az):
    """The optimiser step is a pure function of (parameters, batch): two evaluators fed the same rows end
    with bit-identical parameters (the batch-split gradient reductions add their parts in a fixed order), which
    is what keeps the replicas of a multi-GPU run in lock-step without a parameter broadcast."""
    dims, B = (304, 256, 256, 256, 152), 5000
    rng = np.random.default_rng(1)
    x = (rng.random((B, dims[0])) < 0.3).astype(np.float32)
    obs = rng.random((B, dims[-1]), dtype=np.float32)
    w = (rng.random((B, dims[-1])) < 0.05).astype(np.float32)
    params = []
    for rep in range(2):
        m = az.ActionModel(B, dims[0], dims[-1], hidden=dims[1:-1], seed=11)
        for it in range(3):
            m.update_model(x, obs, w)
        params.append(m.get_params())
    assert np.array_equal(params[0].view(np.uint32), params[1].view(np.uint32))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_async_evaluator_ragged_widths(az, orc, dtype):
    """The asynchronous step's evaluator reads a fragment-major copy of the weights, padded with zeros to
    whole 16x16 fragments: widths that are not multiples of 16 (input 88 = 5.5 k-steps, hidden 48 and 32,
    output 44 = 2.75 column tiles) give the same rows as the batched GEMM path, also after an optimiser step
    and after set_params (the copy follows every change of the parameters)."""
    n, B, seed = 11, 100, 3
    space = az.ROTModifyParentsOnce(n)
    assert (space.STATE_DIM, space.ACTION_DIM) == (88, 44)
    roots = space.generate_roots(seed, B)
    tol = MLP_ATOL if dtype == "f32" else 1e-3
    def rows(persistent, params=None, update=False):
        model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(48, 32), seed=7, dtype=dtype)
        if params is not None:
            model.set_params(params)
        o = az.NablaOptimizer.par_new(space, roots, model, B, persistent=persistent)
        if update:
            o.par_roll_out_episodes(TOL_REF, n_calls=3)
            o.par_update_model(5)
            o.par_reset_trees(roots)
        o.par_roll_out_episodes(TOL_REF, n_calls=1)
        if persistent:
            assert o.counters()["EVAL_ROWS"] > 0  # the in-kernel evaluator ran, not a fallback
        return o.state_vecs(), o.predictions(), model.get_params()
    for update in (False, True):
        s1, p1, w1 = rows(True, update=update)
        s2, p2, w2 = rows(False, update=update)
        assert np.array_equal(s1, s2)
        assert np.max(np.abs(w1 - w2)) < 1e-6
        assert np.max(np.abs(p1 - p2)) < tol, update
    rng = np.random.default_rng(0)
    w = (rng.standard_normal(w1.shape) * 0.2).astype(np.float32)
    s1, p1, _ = rows(True, params=w)
    s2, p2, _ = rows(False, params=w)
    assert np.array_equal(s1, s2) and np.max(np.abs(p1 - p2)) < tol


def test_reference_cost_vectors_on_the_device(az, orc):
    """The reference's own cost vectors through the device functions (node mode and full mode):
    star K_{1,4}: lambda_1 = 2, mu = 1; path P5: lambda_1 = 2 cos(pi/6), mu = 2 (ordered_edge.rs:198-234);
    the 20-vertex tree of connected_bitset_graph/mod.rs:394-422: mu = 9.  lambda_1 equals the oracle's
    bit for bit."""
    import ctypes as C
    import json
    import os
    from azdopt_amd import _lib
    from test_oracle_golden import rooted_relabelling
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_unit_vectors.json")))
    g = gold["tree20_matching_number"]
    cases = [([0, 0, 0, 0, 0], 2.0, 1), ([0, 0, 1, 2, 0], 2 * np.cos(np.pi / 6), 2),
             (rooted_relabelling(g["n_vertices"], g["edges"], last=11), None, g["matching_number"])]
    for parents, lam_want, mu_want in cases:
        n = len(parents)
        p = np.array([parents], np.uint8)
        for full in (0, 1):
            lam, mu, ms = np.zeros(1, np.float64), np.zeros(1, np.int32), C.c_float()
            _lib.check(az.lib().azd_debug_probe_cost(0, _lib.ptr(p), n, 1, 1, full, _lib.ptr(lam), _lib.ptr(mu), C.byref(ms)), "probe_cost")
            assert mu[0] == mu_want
            if lam_want is not None:
                assert abs(lam[0] - lam_want) < 1e-6
            fn = orc.lib().orc_lambda1_sturm if full else orc.lib().orc_lambda1_node
            assert lam[0] == fn(_lib.ptr(p), n)


def test_near_degenerate_trees_on_the_device(az, orc):
    """Round-4 advisor finding (high): the double brooms, where lambda_2 sits beside lambda_1 and the secant window of the
    lambda_1 bracket never caught the root.  Every split of the two-hub family (tests/test_oracle_golden.py:double_brooms,
    N = 6 .. AZD_C21_MAX_N) through the device's cost function in both modes: bit for bit the oracle's f64, and within 1e-12 of
    LAPACK in full mode."""
    import ctypes as C
    from azdopt_amd import _lib
    from test_oracle_golden import double_brooms, _lapack_lambda1
    by_n = {}
    for parents in double_brooms():
        by_n.setdefault(len(parents), []).append(parents)
    checked = 0
    for n, trees in sorted(by_n.items()):
        p = np.array(trees, np.uint8)
        for full in (0, 1):
            lam, mu, ms = np.zeros(len(trees), np.float64), np.zeros(len(trees), np.int32), C.c_float()
            _lib.check(az.lib().azd_debug_probe_cost(0, _lib.ptr(p), n, len(trees), 1, full, _lib.ptr(lam), _lib.ptr(mu), C.byref(ms)), "probe_cost")
            fn = orc.lib().orc_lambda1_sturm if full else orc.lib().orc_lambda1_node
            for i, parents in enumerate(trees):
                row = np.ascontiguousarray(p[i])
                assert lam[i] == fn(_lib.ptr(row), n), (n, parents, full)
                assert mu[i] == orc.lib().orc_maximum_matching(_lib.ptr(row), n, None)
                if full:
                    assert abs(lam[i] - _lapack_lambda1(parents)) < 1e-12 * n
                checked += 1
    assert checked > 3000
