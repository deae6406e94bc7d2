"""The two independent CPU restatements (oracle/azd_oracle.cpp and oracle/py_oracle.py)
must agree bit-for-bit on exported trees, state vectors, observations, argmin and
the seeded root policy.  This is what pins the tree / optimizer semantics, for
which the reference holds no fixture."""
import numpy as np
import pytest

from oracle import py_oracle as po


def unpack_roots(parents, permitted, A):
    out = []
    for i in range(parents.shape[0]):
        mask = sum(int(permitted[i, w]) << (64 * w) for w in range(permitted.shape[1]))
        out.append(([int(x) for x in parents[i]], {a for a in range(A) if mask >> a & 1}))
    return out


def pack_roots(roots, n, kw):
    parents = np.zeros((len(roots), n), np.uint8)
    permitted = np.zeros((len(roots), kw), np.uint64)
    for i, (p, m) in enumerate(roots):
        parents[i] = p
        for a in m:
            permitted[i, a >> 6] |= np.uint64(1 << (a & 63))
    return parents, permitted


def assert_same_trees(ce, pe, B):
    for i in range(B):
        tc = ce.export_tree(i)
        tp = pe.export_tree(i, ce.KW)
        for f in tc.FIELDS:
            a, b = getattr(tc, f), tp[f]
            assert a.shape == b.shape, (i, f, a.shape, b.shape)
            if a.dtype.kind == "f":
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (i, f)
            else:
                assert np.array_equal(a.astype(np.int64), b.astype(np.int64)), (i, f)


def run_pair(orc, n, B, kmin, kmax, tol, tol_default, steps, epochs, seed, n_obs_tol, seq=False, layers=1):
    ce = orc.Engine(n, B, threads=2, path_kind=1 if seq else 0, layers=layers)
    pe = po.PyEngine(n, B, seq=seq, layers=layers)
    A = ce.A
    parents, permitted = orc.gen_roots(seed, 0, 0, B, n, kmin, kmax)
    ce.new_begin(parents, permitted)
    pe.new_begin(unpack_roots(parents, permitted, A))
    assert np.array_equal(ce.state_vecs(), pe.vecs)
    call = 0
    h = orc.hash_predictions(seed, 0, B, A, call)
    ce.new_end(h)
    pe.new_end(h)
    assert_same_trees(ce, pe, B)
    stats = dict(improved=0)
    for epoch in range(epochs):
        for _ in range(steps):
            ce.rollout_begin(tol, tol_default)
            pe.rollout_begin(tol, tol_default)
            assert np.array_equal(ce.state_vecs(), pe.vecs)
            call += 1
            h = orc.hash_predictions(seed, 0, B, A, call)
            ic, ip = ce.rollout_end(h), pe.rollout_end(h)
            assert ic == ip
            stats["improved"] += ic
            am = ce.argmin()
            assert [int(x) for x in am["parents"]] == pe.argmin["parents"]
            assert am["eval"] == pe.argmin["eval"] and am["lambda1"] == pe.argmin["lambda1"]
            assert am["matching"] == pe.argmin["matching"]
        assert_same_trees(ce, pe, B)
        oc, wc = ce.observe(n_obs_tol)
        op, wp = pe.observe(n_obs_tol)
        assert np.array_equal(oc.view(np.uint32), op.view(np.uint32)) and np.array_equal(wc, wp)
        assert np.array_equal(ce.state_vecs(), pe.vecs)
        rp, rm = ce.modify_roots(seed, epoch, 0, kmin, kmax)
        new_roots = pe.modify_roots(seed, epoch, 0, kmin, kmax)
        pp, pm = pack_roots(new_roots, n, ce.KW)
        assert np.array_equal(rp, pp) and np.array_equal(rm, pm)
        ce.reset_begin(rp, rm)
        pe.reset_begin(new_roots)
        call += 1
        h = orc.hash_predictions(seed, 0, B, A, call)
        ce.reset_end(h)
        pe.reset_end(h)
        assert_same_trees(ce, pe, B)
    stats.update(ce.counters())
    return stats


def test_cross_n5(orc):
    s = run_pair(orc, 5, 6, 1, 5, [3, 2], 1, steps=12, epochs=3, seed=11, n_obs_tol=1)
    assert s["FAILED"] == 0 and s["TERMINALS"] > 0 and s["ROOT_EXHAUSTED"] > 0


def test_cross_n8_exercises_all_branches(orc):
    s = run_pair(orc, 8, 8, 2, 10, [4, 2, 2], 1, steps=60, epochs=2, seed=3, n_obs_tol=2)
    assert s["FAILED"] == 0
    for k in ("EXPANSIONS", "TERMINALS", "TRANSPOSITIONS", "VISITED_STEPS", "CASCADE_NODES"):
        assert s[k] > 0, (k, s)


def test_cross_sequence_paths_never_transpose(orc):
    """P = ActionSequence / OrderedActionSet: keys are the actions in the order taken, so a path can
    only meet itself -- the search graph is a tree, and the BTreeMap order the root policy sees is
    the lexicographic order of sequences"""
    s = run_pair(orc, 8, 8, 2, 10, [4, 2, 2], 1, steps=60, epochs=2, seed=3, n_obs_tol=2, seq=True)
    assert s["FAILED"] == 0 and s["TRANSPOSITIONS"] == 0
    for k in ("EXPANSIONS", "TERMINALS", "VISITED_STEPS", "CASCADE_NODES"):
        assert s[k] > 0, (k, s)
    s = run_pair(orc, 5, 6, 1, 5, [3, 2], 1, steps=12, epochs=3, seed=11, n_obs_tol=1, seq=True)
    assert s["FAILED"] == 0 and s["TRANSPOSITIONS"] == 0 and s["ROOT_EXHAUSTED"] > 0


@pytest.mark.parametrize("layers", [2, 3])
def test_cross_layered_history(orc, layers):
    """Layered<L, Space>: the state vector carries the last L states of the path (chunks of states the ring
    does not hold keep their earlier contents); the search itself is unchanged"""
    plain = run_pair(orc, 8, 8, 2, 10, [4, 2, 2], 1, steps=60, epochs=2, seed=3, n_obs_tol=2)
    s = run_pair(orc, 8, 8, 2, 10, [4, 2, 2], 1, steps=60, epochs=2, seed=3, n_obs_tol=2, layers=layers)
    for k in ("EXPANSIONS", "TERMINALS", "TRANSPOSITIONS", "VISITED_STEPS", "CASCADE_NODES"):
        assert s[k] == plain[k] > 0, k


@pytest.mark.parametrize("seed", [0, 1])
def test_cross_n19_reference_shape(orc, seed):
    # reference hyper-parameters scaled down: tol table [200,50,50]/25 -> [8,3,3]/2 so revisits occur quickly
    s = run_pair(orc, 19, 4, 5, 76, [8, 3, 3], 2, steps=40, epochs=1, seed=seed, n_obs_tol=4)
    assert s["FAILED"] == 0 and s["EXPANSIONS"] > 0 and s["VISITED_STEPS"] > 0


# ---------------------------------------------------------------- Ramsey space
def _unpack_ramsey(colors, permitted, E):
    out = []
    for i in range(colors.shape[0]):
        mask = sum(int(permitted[i, w]) << (64 * w) for w in range(permitted.shape[1]))
        out.append(([int(c) for c in colors[i]], {e for e in range(E) if mask >> e & 1}))
    return out


def _pack_ramsey(roots, E, kw):
    colors = np.zeros((len(roots), E), np.uint8)
    permitted = np.zeros((len(roots), kw), np.uint64)
    for i, (c, m) in enumerate(roots):
        colors[i] = c
        for e in m:
            permitted[i, e >> 6] |= np.uint64(1 << (e & 63))
    return colors, permitted


@pytest.mark.parametrize("seq", [False, True])
@pytest.mark.parametrize("n,sizes,weights,kmin,kmax,seed", [(6, [3, 3], [1.0, 1.0], 3, 7, 0), (7, [3, 4], [1.0, 2.0], 3, 8, 1),
                                                              (6, [3, 3, 3], [1.0, 0.5, 2.0], 2, 6, 2), (8, [4, 5], [1.0, 1.0], 4, 9, 3)])
def test_ramsey_restatements_agree_bit_for_bit(orc, n, sizes, weights, kmin, kmax, seed, seq):
    """C++ oracle (incremental counts) vs the Python restatement (recount from the definition) under the
    same prediction stream: trees, state vectors, observations, argmin and the root policy must agree."""
    from oracle import py_ramsey as pr
    B, tol, tol_default, steps, epochs, n_obs_tol = 5, [6, 3, 2], 1, 45, 2, 2
    layers = 2 if (seq and len(sizes) == 2) else 1  # also exercise Layered<2, _> on two of the cases
    ce = orc.Engine(n, B, threads=2, ramsey=(sizes, weights), path_kind=1 if seq else 0, layers=layers)
    pe = pr.PyRamseyEngine(n, sizes, weights, B, seq=seq, layers=layers)
    E = n * (n - 1) // 2
    colors, permitted = orc.gen_ramsey_roots(seed, 0, 0, B, n, len(sizes), kmin, kmax)
    ce.new_begin(colors, permitted)
    pe.new_begin(_unpack_ramsey(colors, permitted, E))
    assert np.array_equal(ce.state_vecs(), pe.vecs)
    call = 0
    h = orc.hash_predictions(seed, 0, B, ce.A, call)
    ce.new_end(h)
    pe.new_end(h)
    assert_same_trees(ce, pe, B)
    for epoch in range(epochs):
        for _ in range(steps):
            ce.rollout_begin(tol, tol_default)
            pe.rollout_begin(tol, tol_default)
            assert np.array_equal(ce.state_vecs(), pe.vecs)
            call += 1
            h = orc.hash_predictions(seed, 0, B, ce.A, call)
            assert ce.rollout_end(h) == pe.rollout_end(h)
            am = ce.argmin()
            assert am["eval"].tobytes() == np.float32(pe.argmin["eval"]).tobytes()
            assert am["parents"].tolist() == pe.argmin["state"].colors
            assert ce.argmin_totals()[:len(sizes)].tolist() == pe.argmin["state"].totals
        assert_same_trees(ce, pe, B)
        for i in range(B):
            assert ce.agent_counts(i).tolist() == pe.states[i].counts
        oc, wc = ce.observe(n_obs_tol)
        op, wp = pe.observe(n_obs_tol)
        nan = np.isnan(oc)
        assert np.array_equal(nan, np.isnan(op)) and np.array_equal(wc, wp)
        assert np.array_equal(oc[~nan].view(np.uint32), op[~nan].view(np.uint32))
        assert np.array_equal(ce.state_vecs(), pe.vecs)
        rc = ce.modify_roots(seed, epoch, 0, kmin, kmax)
        rp = pe.modify_roots(seed, epoch, 0, kmin, kmax)
        pc = _pack_ramsey(rp, E, ce.KW)
        assert np.array_equal(rc[0], pc[0]) and np.array_equal(rc[1], pc[1])
        ce.reset_begin(*rc)
        pe.reset_begin(rp)
        call += 1
        h = orc.hash_predictions(seed, 0, B, ce.A, call)
        ce.reset_end(h)
        pe.reset_end(h)
        assert_same_trees(ce, pe, B)
