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


def run_pair(orc, n, B, kmin, kmax, tol, tol_default, steps, epochs, seed, n_obs_tol):
    ce = orc.Engine(n, B, threads=2)
    pe = po.PyEngine(n, B)
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


@pytest.mark.parametrize("seed", [0, 1])
def test_cross_n19_reference_shape(orc, seed):
    # reference hyper-parameters scaled down: tol table [200,50,50]/25 -> [8,3,3]/2 so revisits occur quickly
    s = run_pair(orc, 19, 4, 5, 76, [8, 3, 3], 2, steps=40, epochs=1, seed=seed, n_obs_tol=4)
    assert s["FAILED"] == 0 and s["EXPANSIONS"] > 0 and s["VISITED_STEPS"] > 0
