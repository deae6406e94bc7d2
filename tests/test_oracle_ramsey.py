"""Pins the oracle's Ramsey space layer (oracle/azd_oracle.cpp: count_cliques_inside, RamseyCounts::new,
reassign_color, action_data, write_vec, evaluate, g / h) against
  * the reference's own unit-test vector and randomized properties (ramsey_counts/space.rs:205-313),
  * a brute-force clique counter written here from the definition (itertools), and
  * hand-evaluated formulas for the reward/observation maps."""
import itertools
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_unit_vectors.json")))
F = np.float32


def colex(v, u):
    v, u = max(v, u), min(v, u)
    return v * (v + 1) // 2 - (v - u)


def edges(n):
    return [(v, u) for v in range(n) for u in range(v)]


def brute_counts(n, sizes, colors):
    """counts[c][e(v,u)] = #cliques of size sizes[c]-2 of colour c inside the common c-neighbourhood of
    v and u (for every pair, adjacent or not); totals[c] = #monochromatic sizes[c]-cliques."""
    E = n * (n - 1) // 2
    adj = np.zeros((len(sizes), n, n), bool)
    for e, (v, u) in enumerate(edges(n)):
        adj[colors[e], v, u] = adj[colors[e], u, v] = True
    counts = np.zeros((len(sizes), E), np.int32)
    totals = np.zeros(len(sizes), np.int32)

    def is_clique(c, vs):
        return all(adj[c, a, b] for a, b in itertools.combinations(vs, 2))

    for c, k in enumerate(sizes):
        for e, (v, u) in enumerate(edges(n)):
            common = [w for w in range(n) if adj[c, v, w] and adj[c, u, w]]
            counts[c, e] = sum(1 for vs in itertools.combinations(common, k - 2) if is_clique(c, vs))
        totals[c] = sum(1 for vs in itertools.combinations(range(n), k) if is_clique(c, vs))
    return counts, totals


def test_reference_k5_vector(orc):
    g = GOLD["ramsey_k5_red_blue_k4_counts"]
    n = g["n_vertices"]
    colors = np.zeros(n * (n - 1) // 2, np.uint8)
    for a, b in g["blue_edges"]:
        colors[colex(a, b)] = 1
    for a, b in g["red_edges"]:
        assert colors[colex(a, b)] == 0
    counts, totals = orc.ramsey_counts_new(n, g["sizes"], colors)
    assert counts.tolist() == g["counts"]
    assert totals.tolist() == [0, 0]


@pytest.mark.parametrize("n,sizes", [(6, [3, 3]), (7, [3, 3, 3]), (8, [4, 4]), (8, [4, 5]), (9, [3, 4, 5]), (8, [2, 3])])
def test_counts_new_matches_brute_force(orc, n, sizes):
    rng = np.random.default_rng(n * 131 + len(sizes))
    for _ in range(4):
        colors = rng.integers(0, len(sizes), n * (n - 1) // 2).astype(np.uint8)
        counts, totals = orc.ramsey_counts_new(n, sizes, colors)
        bc, bt = brute_counts(n, sizes, colors)
        assert np.array_equal(counts, bc)
        assert np.array_equal(totals, bt)


@pytest.mark.parametrize("n,sizes", [(30, [3, 3, 3]), (5, [4, 4]), (17, [4, 4]), (16, [3, 3, 3]), (12, [4, 5]), (11, [5, 5]),
                                     (10, [3, 4, 5, 3])])
def test_incremental_counts_equal_recount_after_every_action(orc, n, sizes):
    """the reference's two randomized tests (first two cases are theirs), extended to K4/K5 targets"""
    rng = np.random.default_rng(7 * n + sum(sizes))
    E, C = n * (n - 1) // 2, len(sizes)
    colors = rng.integers(0, C, E).astype(np.uint8)
    cur = colors.copy()
    order = rng.permutation(E)  # every edge permitted: play until terminal
    actions = []
    for step, e in enumerate(order):
        nc = int((cur[e] + 1 + rng.integers(0, C - 1)) % C)
        actions.append(int(e + nc * E))
        cur[e] = nc
        if n > 17 and step % 29 and step != E - 1:
            continue  # keep the big case to a few minutes: recount every 29th step and at the end
        got_colors, counts, totals = orc.ramsey_act_sequence(n, sizes, colors, actions)
        assert np.array_equal(got_colors, cur)
        rc, rt = orc.ramsey_counts_new(n, sizes, cur)
        assert np.array_equal(counts, rc), step
        assert np.array_equal(totals, rt), step


def _engine(orc, n, sizes, weights, B, seed, kmin, kmax):
    e = orc.Engine(n, B, threads=1, ramsey=(sizes, weights))
    colors, permitted = orc.gen_ramsey_roots(seed, 0, 0, B, n, len(sizes), kmin, kmax)
    e.new_begin(colors, permitted)
    return e, colors, permitted


def test_state_vector_actions_and_g(orc):
    n, sizes, weights = 7, [3, 4], [1.0, 2.5]
    E, C = 21, 2
    e, colors, permitted = _engine(orc, n, sizes, weights, 5, 3, 4, 9)
    assert (e.S, e.A, e.KW, e.RB) == (E * (2 * C + 1), E * C, 1, E)
    sv = e.state_vecs()
    h = orc.hash_predictions(3, 0, 5, e.A, 0)
    e.new_end(h)
    for i in range(5):
        counts, totals = brute_counts(n, sizes, colors[i])
        perm = [p for p in range(E) if int(permitted[i, 0]) >> p & 1]
        assert 4 <= len(perm) <= 9
        # write_vec (ramsey_counts/space.rs:122-153): counts, edge bools per colour, permitted edges
        want = np.zeros(e.S, F)
        want[:C * E] = counts.reshape(-1)
        for c in range(C):
            want[C * E + c * E:C * E + (c + 1) * E] = colors[i] == c
        want[2 * C * E + np.array(perm)] = 1
        assert np.array_equal(sv[i], want)
        # evaluate (:159-165), action_data order (:88-120), g (:167-172)
        c_s = F(0)
        for c in range(C):
            c_s = F(c_s + F(totals[c]) * F(weights[c]))
        t = e.export_tree(i)
        assert t.c[0] == c_s
        ids, gs = [], []
        for p in perm:
            old = int(colors[i, p])
            for nc in range(C):
                if nc == old:
                    continue
                a = p + nc * E
                r = F(F(counts[old, p]) * F(weights[old]) - F(counts[nc, p]) * F(weights[nc]))
                hh = h[i, a]
                ids.append(a)
                gs.append(F(F(c_s * hh) + F(r * F(F(1) - hh))))
        assert t.p_aid.tolist() == ids
        assert np.array_equal(t.p_g, np.array(gs, F))


def test_observations_use_one_minus_ratio(orc):
    n, sizes, weights = 8, [3, 3], [1.0, 1.0]
    B = 6
    e, colors, permitted = _engine(orc, n, sizes, weights, B, 11, 3, 6)
    e.new_end(orc.hash_predictions(11, 0, B, e.A, 0))
    for call in range(1, 120):
        e.rollout_begin([4, 2], 1)
        e.rollout_end(orc.hash_predictions(11, 0, B, e.A, call))
    obs, w = e.observe(2)
    seen = 0
    for i in range(B):
        t = e.export_tree(i)
        act = t.act_begin + t.exhausted < t.act_end
        for k in range(len(t.e_src)):
            if t.e_src[k] != 0:
                continue
            ch = t.e_dst[k]
            a = t.p_aid[t.e_pp[k]]
            if (not act[ch]) or t.n_t[ch] >= 2:
                with np.errstate(divide="ignore", invalid="ignore"):
                    want = F(F(1) - F(t.c_star[ch] / t.c[ch]))  # h_sa (:174-177)
                assert w[i, a] == 1
                assert obs[i, a].tobytes() == want.tobytes()
                seen += 1
            else:
                assert w[i, a] == 0
    assert seen > 10
    # live counts of every agent agree with a recount of its current colouring
    for i in range(B):
        st = e.agent_state(i)
        rc, rt = brute_counts(n, sizes, st["parents"])
        assert np.array_equal(e.agent_counts(i), rc)
        if st["path"].any():  # standing on the node created by the last call: costs[i] is that node's cost
            assert np.array_equal(e.agent_totals(i)[:2], rt)
