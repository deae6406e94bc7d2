"""GPU parity for the other path encodings: ActionSequence / OrderedActionSet keys
(no transpositions, BTreeMap order = lexicographic sequence order in the root policy) and
ActionMultiset (coincides with ActionSet on ActionsNeverRepeat spaces) against the CPU oracle."""
import numpy as np
import pytest

from test_gpu_ramsey import MAIN_CTRS, assert_tree_equal, az  # noqa: F401

pytestmark = pytest.mark.gpu


def run(az, orc, space, ramsey, path, kmin, kmax, B, tol, steps, epochs, seed, n_obs_tol, check_every):
    kind = path.PATH_KIND
    model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, seed, 0)
    roots = space.generate_roots(seed, B, kmin=kmin, kmax=kmax)
    opt = az.NablaOptimizer.par_new(space, roots, model, B, path=path)
    oe = orc.Engine(space.n, B, threads=8, ramsey=ramsey, path_kind=kind)
    oe.new_begin(*roots)
    call = 0
    oe.new_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))

    def compare(tag):
        assert np.array_equal(opt.state_vecs(), oe.state_vecs()), tag
        for i in range(B):
            assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"{tag} agent {i}")
        cg, co = opt.counters(), oe.counters()
        for k in MAIN_CTRS:
            assert cg[k] == co[k], (tag, k, cg[k], co[k])
        assert opt.argmin_data().eval.tobytes() == oe.argmin()["eval"].tobytes(), tag

    compare("par_new")
    for epoch in range(epochs):
        for s in range(0, steps, check_every):
            k = min(check_every, steps - s)
            ig = opt.par_roll_out_episodes(tol, n_calls=k)
            io = 0
            for _ in range(k):
                oe.rollout_begin(*tol)
                call += 1
                io += oe.rollout_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
            assert ig == io
            compare(f"epoch {epoch} step {s + k}")
        sv, obs, w = opt.observe(n_obs_tol)
        oo, ow = oe.observe(n_obs_tol)
        nan = np.isnan(oo)
        assert np.array_equal(np.isnan(obs), nan) and np.array_equal(w, ow)
        assert np.array_equal(obs[~nan].view(np.uint32), oo[~nan].view(np.uint32))
        ro = oe.modify_roots(seed, epoch, 0, kmin, kmax)
        rg = opt.modify_roots(seed, epoch, kmin, kmax)  # device policy, key order of this encoding
        assert np.array_equal(rg[0], ro[0]) and np.array_equal(rg[1], ro[1]), epoch
        opt.par_reset_trees_policy(seed, epoch, kmin, kmax)
        oe.reset_begin(*ro)
        call += 1
        oe.reset_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
        compare(f"epoch {epoch} reset")
    return opt.counters()


@pytest.mark.parametrize("path_name", ["ActionSequence", "OrderedActionSet"])
def test_c21_sequence_keys(az, orc, path_name):
    space = az.ROTModifyParentsOnce(8)
    c = run(az, orc, space, None, getattr(az, path_name), 2, 10, B=40, tol=([4, 2, 2], 1), steps=90, epochs=3, seed=3,
            n_obs_tol=2, check_every=9)
    assert c["TRANSPOSITIONS"] == 0 and c["TERMINALS"] > 0 and c["VISITED_STEPS"] > 0 and c["FAILED"] == 0


def test_c21_reference_size_sequence_keys(az, orc):
    space = az.ROTModifyParentsOnce(19)
    c = run(az, orc, space, None, az.ActionSequence, 5, 76, B=48, tol=([200, 50, 50], 25), steps=300, epochs=2, seed=1,
            n_obs_tol=200, check_every=100)
    assert c["TRANSPOSITIONS"] == 0 and c["FAILED"] == 0


def test_ramsey_sequence_keys(az, orc):
    sizes, weights = [3, 4], [1.0, 2.0]
    space = az.RamseySpaceNoEdgeRecolor(8, sizes, weights)
    c = run(az, orc, space, (sizes, weights), az.ActionSequence, 3, 8, B=32, tol=([6, 3, 2], 1), steps=80, epochs=3, seed=5,
            n_obs_tol=2, check_every=8)
    assert c["TRANSPOSITIONS"] == 0 and c["TERMINALS"] > 0 and c["FAILED"] == 0


def test_multiset_coincides_with_set(az, orc):
    """ActionMultiset on a never-repeat space: same trees as ActionSet, transpositions included"""
    space = az.ROTModifyParentsOnce(8)
    c = run(az, orc, space, None, az.ActionMultiset, 2, 10, B=24, tol=([4, 2, 2], 1), steps=60, epochs=2, seed=3,
            n_obs_tol=2, check_every=10)
    assert c["TRANSPOSITIONS"] > 0 and c["FAILED"] == 0


def test_unlicensed_path_is_refused(az):
    class NoAxioms:
        SPACE_ID, n = 1, 8

    with pytest.raises(TypeError):
        az.NablaOptimizer(NoAxioms(), None, 4, path=az.ActionSet)
    with pytest.raises(TypeError):
        az.NablaOptimizer(NoAxioms(), None, 4, path=az.OrderedActionSet)


@pytest.mark.parametrize("layers,persistent,async_step", [(2, True, True), (3, True, False), (3, False, False)])
def test_c21_layered_history(az, orc, layers, persistent, async_step):
    """Layered<L, Space> (space/layered.rs): state vectors carry the last L states of the path, everything
    else is the plain search; all three step forms against the oracle"""
    n, B, seed, kmin, kmax = 8, 40, 3, 2, 10
    tol = ([4, 2, 2], 1)
    space = az.Layered(az.ROTModifyParentsOnce(n), layers)
    model = az.HashStreamModel(space.STATE_DIM, space.ACTION_DIM, seed, 0)
    roots = space.generate_roots(seed, B, kmin=kmin, kmax=kmax)
    opt = az.NablaOptimizer.par_new(space, roots, model, B, persistent=persistent, async_step=async_step)
    oe = orc.Engine(n, B, threads=8, layers=layers)
    assert oe.S == space.STATE_DIM
    oe.new_begin(*roots)
    call = 0
    oe.new_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
    assert np.array_equal(opt.state_vecs(), oe.state_vecs())
    for epoch in range(2):
        for s in range(0, 60, 6):
            opt.par_roll_out_episodes(tol, n_calls=6)
            for _ in range(6):
                oe.rollout_begin(*tol)
                call += 1
                oe.rollout_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
            assert np.array_equal(opt.state_vecs(), oe.state_vecs()), (epoch, s)
            for i in range(B):
                assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"epoch {epoch} step {s} agent {i}")
        sv, obs, w = opt.observe(2)
        oo, ow = oe.observe(2)
        assert np.array_equal(sv, oe.state_vecs()) and np.array_equal(obs, oo) and np.array_equal(w, ow)
        ro = oe.modify_roots(seed, epoch, 0, kmin, kmax)
        opt.par_reset_trees_policy(seed, epoch, kmin, kmax)
        oe.reset_begin(*ro)
        call += 1
        oe.reset_end(orc.hash_predictions(seed, 0, B, space.ACTION_DIM, call))
        assert np.array_equal(opt.state_vecs(), oe.state_vecs())


def test_ramsey_layered_history_with_mlp(az, orc):
    """Layered<2, Ramsey> with the MLP evaluator (input = two stacked state vectors)"""
    n, sizes, B, seed = 8, [3, 4], 24, 7
    tol = ([6, 3, 2], 1)
    space = az.Layered(az.RamseySpaceNoEdgeRecolor(n, sizes, [1.0, 2.0]), 2)
    model = az.ActionModel(B, space.STATE_DIM, space.ACTION_DIM, hidden=(64, 32), seed=seed)
    roots = space.generate_roots(seed, B, kmin=3, kmax=8)
    opt = az.NablaOptimizer.par_new(space, roots, model, B)
    oe = orc.Engine(n, B, threads=8, ramsey=(sizes, [1.0, 2.0]), layers=2)
    oe.new_begin(*roots)
    oe.new_end(opt.predictions())
    for s in range(50):
        opt.par_roll_out_episodes(tol)
        oe.rollout_begin(*tol)
        assert np.array_equal(opt.state_vecs(), oe.state_vecs()), s
        oe.rollout_end(opt.predictions())
    for i in range(B):
        assert_tree_equal(opt.get_tree(i), oe.export_tree(i), f"agent {i}")
    assert np.isfinite(opt.par_update_model(2))
