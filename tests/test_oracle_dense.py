"""The oracle's dense-graph space (oracle/dense_graph.inc; BASELINE configs[4]) against the reference's own test vectors
for general graphs, against brute force, and its space axioms.  CPU only."""
import itertools
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_graph_vectors.json")))


def cut_edges(orc, adj):
    n = len(adj)
    return sorted((u, v) for v in range(n) for u in range(v) if (int(adj[v]) >> u) & 1 and orc.dense_is_cut_edge(adj, v, u))


@pytest.mark.parametrize("case", GOLD["matching"], ids=lambda c: c["name"])
def test_matching_numbers_of_the_reference_tests(orc, case):
    adj = orc.adjacency(case["n"], case["edges"])
    assert orc.dense_matching_reference(adj) == case["matching_number"]  # the literal branch and bound
    assert orc.dense_matching_exact(adj) == case["matching_number"]      # what the search uses (Edmonds)
    assert orc.dense_matching_tutte(adj) == case["matching_number"]      # the Tutte-rank cross-check (rounds 1-2 used it)


@pytest.mark.parametrize("case", GOLD["cut_edges"], ids=lambda c: c["name"])
def test_cut_edges_of_the_reference_tests_under_every_relabeling(orc, case):
    n = case["n"]
    perms = itertools.permutations(range(n)) if n <= 4 else [tuple(range(n)), (4, 3, 2, 1, 0), (1, 0, 3, 2, 4)]
    for sigma in perms:  # block.rs:233-264 runs all of S_4
        adj = orc.adjacency(n, [(sigma[u], sigma[v]) for u, v in case["edges"]])
        want = sorted(tuple(sorted((sigma[u], sigma[v]))) for u, v in case["cut_edges"])
        assert cut_edges(orc, adj) == want, sigma


def random_connected(rng, n, p):
    """a random spanning tree plus G(n, p): connected whatever p (sparse near-trees at small p, dense graphs at large p)"""
    order = rng.permutation(n)
    es = {tuple(sorted((int(order[i]), int(order[rng.integers(0, i)])))) for i in range(1, n)}
    es |= {(u, v) for v in range(n) for u in range(v) if rng.random() < p}
    edges = sorted(es)
    a = np.zeros((n, n))
    for u, v in edges:
        a[u, v] = a[v, u] = 1
    return edges, a


def test_exact_matching_on_larger_graphs_against_networkx_free_checks(orc):
    """Edmonds against the Tutte rank (independent algebraic route) on graphs too large for the branch and bound, including
    blossom-rich ones (odd cycles glued together, dense graphs) and graphs with many unmatched vertices (stars, spiders)"""
    rng = np.random.default_rng(7)
    for trial in range(300):
        n = int(rng.integers(15, 65))
        edges, _ = random_connected(rng, n, rng.random() * 0.25 + 0.03)
        adj = orc.adjacency(n, edges)
        assert orc.dense_matching_exact(adj) == orc.dense_matching_tutte(adj), (trial, n)
    for n in (5, 21, 50, 64):
        star = orc.adjacency(n, [(0, i) for i in range(1, n)])
        assert orc.dense_matching_exact(star) == 1
        cyc = orc.adjacency(n, [(i, (i + 1) % n) for i in range(n)])
        assert orc.dense_matching_exact(cyc) == n // 2
        comp = orc.adjacency(n, [(u, v) for v in range(n) for u in range(v)])
        assert orc.dense_matching_exact(comp) == n // 2
    # triangles joined at a hub: 1 + (number of triangles) ... each triangle gives one edge off the hub, the hub adds none
    tri = []
    for k in range(10):
        a, b = 1 + 2 * k, 2 + 2 * k
        tri += [(0, a), (0, b), (a, b)]
    assert orc.dense_matching_exact(orc.adjacency(21, tri)) == 10


def test_tutte_rank_equals_branch_and_bound_and_cut_edges_equal_the_definition(orc):
    rng = np.random.default_rng(0)
    import scipy.sparse.csgraph as cg
    for trial in range(200):
        n = int(rng.integers(4, 15))
        edges, a = random_connected(rng, n, rng.random() * 0.5 + 0.1)
        adj = orc.adjacency(n, edges)
        assert orc.dense_matching_tutte(adj) == orc.dense_matching_reference(adj), trial
        assert orc.dense_matching_exact(adj) == orc.dense_matching_reference(adj), trial
        want = []
        for u, v in edges:  # a cut edge is one whose removal disconnects the graph
            b = a.copy()
            b[u, v] = b[v, u] = 0
            if cg.connected_components(b)[0] > 1:
                want.append((u, v))
        assert cut_edges(orc, adj) == sorted(want), trial


def test_lambda1_against_lapack(orc):
    """power iteration until the Collatz-Wielandt bracket is 1e-4 wide, value = the Rayleigh quotient of the last pair (error
    ~ bracket^2 / gap): against LAPACK on random connected graphs, on sparse near-trees at N = 50 and 64 (paths,
    caterpillars: the two largest eigenvalues of A + I within 0.4 %), on a bipartite star (A alone would oscillate)"""
    rng = np.random.default_rng(1)
    worst = 0.0
    for trial in range(200):
        n = int(rng.integers(4, 65))
        edges, a = random_connected(rng, n, rng.random() * 0.4 + 0.03)
        worst = max(worst, abs(orc.dense_lambda1(orc.adjacency(n, edges)) - np.linalg.eigvalsh(a)[-1]))
    assert worst < 5e-8, worst
    for n in (50, 64):
        path = [(i, i + 1) for i in range(n - 1)]
        cat = [(i, i + 1) for i in range(n // 2 - 1)] + [(i, n // 2 + i) for i in range(n - n // 2)]  # a caterpillar
        for edges in (path, cat):
            a = np.zeros((n, n))
            for u, v in edges:
                a[u, v] = a[v, u] = 1
            err = abs(orc.dense_lambda1(orc.adjacency(n, edges)) - np.linalg.eigvalsh(a)[-1])
            assert err < 2e-6, (n, err)  # (the bracket alone, at the old 600-step cap, left 2.3e-5 / 1.4e-4 on the paths)
    n = 50
    star = [(0, i) for i in range(1, n)]
    assert abs(orc.dense_lambda1(orc.adjacency(n, star)) - np.sqrt(n - 1)) < 1e-7


def test_space_dimensions_roots_and_search(orc):
    n, B = 10, 5
    e = orc.Engine(n, B, threads=2, dense=True)
    E = n * (n - 1) // 2
    assert (e.S, e.A, e.KW, e.RB) == (3 * E + 1, 2 * E, (2 * E + 63) // 64, 8 * n)  # 05-ah.rs:39-40; action.rs:10-27
    adj, slots = orc.gen_dense_roots(4, 0, 0, B, n, 3, 12, 0.35)
    e.new_begin(adj.view(np.uint8).reshape(B, -1), slots)
    sv = e.state_vecs()
    for i in range(B):
        k = sum(bin(int(w)).count("1") for w in slots[i])
        assert 3 <= k <= 12 and sv[i, 3 * E] == np.float32(k) / np.float32(E)
        bools = [(int(adj[i, v]) >> u) & 1 for v in range(n) for u in range(v)]
        assert sv[i, :E].tolist() == bools                      # edge_bools, mod.rs:127-137
        assert sv[i, E:3 * E].sum() == k                        # every modifiable slot is either addable or deletable
    e.new_end(orc.hash_predictions(4, 0, B, e.A, 0))
    for call in range(1, 150):
        e.rollout_begin([20, 5, 5], 3)
        e.rollout_end(orc.hash_predictions(4, 0, B, e.A, call))
    c = e.counters()
    assert c["EXPANSIONS"] > 100 and c["TRANSPOSITIONS"] > 0 and c["FAILED"] == 0
    # ActionsNeverRepeat + ActionOrderIndependent: the state of a node is its root plus its SET of actions, in any order
    t = e.export_tree(0)
    root = adj[0].copy()
    for node in range(1, min(len(t.c), 40)):
        acts = [a for a in range(e.A) if (int(t.keys[node, a >> 6]) >> (a & 63)) & 1]
        assert len({a % E for a in acts}) == len(acts)          # a slot at most once
        g = root.copy()
        for a in acts[::-1]:
            slot = a % E
            v = 1
            while v * (v + 1) // 2 <= slot:
                v += 1
            u = slot - v * (v - 1) // 2
            assert ((int(g[v]) >> u) & 1) == (1 if a >= E else 0)  # Delete removes a present edge, Add adds an absent one
            g[v] ^= np.uint64(1 << u)
            g[u] ^= np.uint64(1 << v)
        lam = orc.dense_lambda1(g)
        mu = orc.dense_matching_exact(g)
        assert mu == orc.dense_matching_reference(g)
        # evaluate = squish(mu + lambda_1): slope 1 / (ceil(sqrt(N - 1)) + (N + 1) / 2 - 2) (04-c21-tree.rs:58-74)
        want = np.float32(1.0 / (3 + 5 - 2)) * ((np.float32(mu) + np.float32(lam)) - np.float32(2))
        assert t.c[node] == want


def test_dense_root_policy_of_the_oracle_keeps_roots_legal(orc):
    """the drivers' modify_root over the dense-graph space (orc_c21_modify_roots): every new root is a connected graph with
    kmin..kmax of its E slots open; a root at its slot limit whose search found nothing better is replaced by a fresh G(n, p)"""
    n, B, kmin, kmax, seed = 10, 24, 3, 12, 5
    E = n * (n - 1) // 2
    e = orc.Engine(n, B, threads=4, dense=True, dense_p=0.35)
    adj, slots = orc.gen_dense_roots(seed, 0, 0, B, n, kmin, kmax, p=0.35)
    e.new_begin(adj.view(np.uint8).reshape(B, -1), slots)
    e.new_end(orc.hash_predictions(seed, 0, B, 2 * E, 0))
    for call in range(1, 25):
        e.rollout_begin([5, 3], 2)
        e.rollout_end(orc.hash_predictions(seed, 0, B, 2 * E, call))
    parents, permitted = e.modify_roots(seed, 0, 0, kmin, kmax)
    new_adj = parents.view(np.uint64).reshape(B, n)
    changed = 0
    for i in range(B):
        k = sum(bin(int(w)).count("1") for w in permitted[i])
        assert kmin <= k <= kmax and not any(int(w) for w in permitted[i][(E + 63) // 64:]), (i, k)
        a = new_adj[i]
        seen, frontier = 1, 1
        while frontier:
            nxt = 0
            for v in range(n):
                if (frontier >> v) & 1:
                    nxt |= int(a[v])
            frontier = nxt & ~seen
            seen |= nxt
        assert seen == (1 << n) - 1, i  # connected
        for v in range(n):
            for u in range(n):
                assert ((int(a[v]) >> u) & 1) == ((int(a[u]) >> v) & 1)
        changed += int(not np.array_equal(a, adj[i]))
    assert changed > 0  # some roots moved to a node of their tree (or were redrawn)
    # the same call again gives the same roots (seeded), another epoch different slot draws
    p2, m2 = e.modify_roots(seed, 0, 0, kmin, kmax)
    assert np.array_equal(p2, parents) and np.array_equal(m2, permitted)
    p3, m3 = e.modify_roots(seed, 1, 0, kmin, kmax)
    assert not np.array_equal(m3, permitted)
