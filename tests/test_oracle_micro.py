"""Hand-worked micro-case for the tree / optimizer rules, which the reference holds no fixture for.

The scenario below was stepped BY HAND from the cited reference lines (not from either
restatement); both restatements must reproduce every number.  c21 space, N = 5: vertices 0..4,
actions (parent, child) -> id: (0,2)=0 (1,2)=1 (0,3)=2 (1,3)=3 (2,3)=4 (ordered_edge.rs:35-42).
Root = star K_{1,4} (parents all 0), permitted {1,3,4}; n_as_tol = 1 at every depth.

call 1  root has no children -> max_curiosity picks the FIRST MIN of c - g (next_action.rs:62-70):
        h = (a1 .5, a3 .2, a4 .9) -> a3.  New node 1 = {3}; act drops ids 2,3,4 -> permitted {1};
        not terminal -> expansion, the agent stays on node 1 (tree/mod.rs:180-216).
call 2  continues FROM node 1 (optimizer keeps state_pos): a1 -> new node 2 = {1,3}, nothing left ->
        terminal.  cascade_new_terminal (empty_transitions.rs:50-87): node 1 exhausted 1, c* not
        improved (equal cost) -> n_t 1, now inactive -> root gets n = 1: exhausted 1, n_t 1.
        Back at the root: its only child is inactive -> no revisit; curiosity over ALL children:
        sqrt|c1 - .5| < sqrt|c1 - .9| -> a4.  New node 3 = {4} -> expansion.
call 3  from node 3: a1 -> new node 4 = {1,4} = the path P5 (cheaper), terminal.  Cascade: node 3
        exhausted 1, c* IMPROVES to c4 so n_t stays 0 (:66-70), inactive -> root exhausted 2, n_t 2.
        Root: one candidate left, a1 -> new node 5 = {1}, permitted {3,4} -> expansion.
call 4  from node 5, h = (a3 .7, a4 .1) -> a4: key {1,4} EXISTS -> transposition arc 5->4
        (tree/mod.rs:172-179), cascade_old_node (:89-127): node 5 exhausted 1 (node 4 is inactive),
        c* improves to c4, n_t = max(0, n_t(node 4) = 0); node 5 still active -> root gets n = 0:
        exhausted stays 2, n_t 3.  Root: revisit_choice finds node 5 active with n_t 0 < 1 ->
        Visited.  Node 5: child node 4 inactive -> curiosity -> a3: key {1,3} EXISTS (node 2) ->
        second transposition: node 5 exhausted 2, c* (c4) not improved by c1 -> n_t 1, inactive ->
        root exhausted 3, n_t 4, inactive.  Root: next_action = None with an empty path
        (tree/mod.rs:220-229) -> the call ends without an expansion."""
import numpy as np
import pytest

from oracle import py_oracle as po

F = np.float32
N = 5


def cost(parents):
    a = np.zeros((N, N))
    for v in range(1, N):
        a[v, parents[v]] = a[parents[v], v] = 1
    lam = np.linalg.eigvalsh(a)[-1]
    mu = 2 if any(parents[v] != 0 for v in (2, 3)) else 1  # star: 1; every other tree here: 2
    slope = F(1.0) / F(3.0)  # C_UPPER = ceil(sqrt(4)) + 3 = 5, C_LOWER = 2 (04-c21-tree.rs:58-74)
    return slope * ((F(mu) + F(lam)) - F(2))


def h_rows():
    h = np.zeros((5, 1, 5), F)
    h[0, 0, [1, 3, 4]] = [0.5, 0.2, 0.9]  # par_new
    h[1, 0, 1] = 0.3                     # node 1
    h[2, 0, 1] = 0.3                     # node 3
    h[3, 0, [3, 4]] = [0.7, 0.1]         # node 5
    return h


def expected():
    c_r = cost([0, 0, 0, 0, 0])
    c_1 = cost([0, 0, 0, 1, 0])
    c_4 = cost([0, 0, 1, 2, 0])
    assert cost([0, 0, 1, 1, 0]) == c_1 and cost([0, 0, 0, 2, 0]) == c_1 and cost([0, 0, 1, 0, 0]) == c_1
    assert abs(float(c_4) - 2 * np.cos(np.pi / 6) / 3) < 1e-6 and c_4 < c_1 and c_r < c_4
    h = h_rows()
    g = lambda c, x: F(c - F(x))  # noqa: E731  g = c_s - h (04-c21-tree.rs:103)
    return dict(
        c=[c_r, c_1, c_1, c_1, c_4, c_1], c_star=[c_r, c_1, c_1, c_4, c_4, c_4],
        n_t=[4, 1, 0, 0, 0, 1], exhausted=[3, 1, 0, 1, 0, 2],
        act_begin=[0, 3, 0, 4, 0, 5], act_end=[3, 4, 0, 5, 0, 7],
        keys=[0, 1 << 3, (1 << 1) | (1 << 3), 1 << 4, (1 << 1) | (1 << 4), 1 << 1],
        e_src=[0, 1, 0, 3, 0, 5, 5], e_dst=[1, 2, 3, 4, 5, 4, 2], e_pp=[1, 3, 2, 4, 0, 6, 5],
        p_aid=[1, 3, 4, 1, 1, 3, 4],
        p_g=[g(c_r, h[0, 0, 1]), g(c_r, h[0, 0, 3]), g(c_r, h[0, 0, 4]), g(c_1, h[1, 0, 1]), g(c_1, h[2, 0, 1]),
             g(c_1, h[3, 0, 3]), g(c_1, h[3, 0, 4])],
        p_edge=[4, 0, 2, 1, 3, 6, 5])


def check(t, want):
    for k, v in want.items():
        got = np.asarray(t[k] if isinstance(t, dict) else getattr(t, k))
        if k in ("c", "c_star", "p_g"):
            assert got.astype(F).tobytes() == np.array(v, F).tobytes(), (k, got, v)
        else:
            assert got.reshape(len(v)).astype(np.int64).tolist() == list(v), (k, got, v)


def test_hand_worked_scenario_cpp_oracle(orc):
    parents = np.zeros((1, N), np.uint8)
    permitted = np.array([[(1 << 1) | (1 << 3) | (1 << 4)]], np.uint64)
    h = h_rows()
    e = orc.Engine(N, 1)
    e.new_begin(parents, permitted)
    e.new_end(h[0])
    for call in range(1, 5):
        e.rollout_begin([1], 1)
        e.rollout_end(h[min(call, 4)])
    check(e.export_tree(0), expected())
    c = e.counters()
    assert (c["EXPANSIONS"], c["TERMINALS"], c["TRANSPOSITIONS"], c["VISITED_STEPS"], c["ROOT_EXHAUSTED"]) == (3, 2, 2, 1, 1)
    # a further call on the exhausted root changes nothing
    e.rollout_begin([1], 1)
    e.rollout_end(h[4])
    check(e.export_tree(0), expected())
    # write_observations (tree/mod.rs:242-264): every root child is inactive -> h_sa = c_child*
    obs, w = e.observe(200)
    want = expected()
    assert w[0].tolist() == [0, 1, 0, 1, 1]
    assert obs[0, 1] == want["c_star"][5] and obs[0, 3] == want["c_star"][1] and obs[0, 4] == want["c_star"][3]


def test_hand_worked_scenario_python_oracle():
    h = h_rows()
    e = po.PyEngine(N, 1)
    e.new_begin([([0] * N, {1, 3, 4})])
    e.new_end(h[0])
    for call in range(1, 5):
        e.rollout_begin([1], 1)
        e.rollout_end(h[min(call, 4)])
    check(e.export_tree(0, 1), expected())


@pytest.mark.gpu
def test_hand_worked_scenario_device():
    import azdopt_amd as az
    space = az.ROTModifyParentsOnce(N)
    parents = np.zeros((1, N), np.uint8)
    permitted = np.array([[(1 << 1) | (1 << 3) | (1 << 4)]], np.uint64)
    h = h_rows()
    opt = az.NablaOptimizer(space, None, 1)
    opt.par_new_begin(parents, permitted)
    opt.par_new_end(h[0])
    for call in range(1, 5):
        opt.roll_out_begin(([1], 1))
        opt.roll_out_end(h[min(call, 4)])
    check(opt.get_tree(0), expected())
    c = opt.counters()
    assert (c["EXPANSIONS"], c["TERMINALS"], c["TRANSPOSITIONS"], c["VISITED_STEPS"], c["ROOT_EXHAUSTED"]) == (3, 2, 2, 1, 1)


# ---------------------------------------------------------------- forced two-parent cascades, stepped by hand
def _diamond(c_r, c_p1, c_p2, c_m, extra_p2=False):
    """R --a0--> P1 --a2--> M, R --a1--> P2 --a2'--> M: a hand-built diamond (what ActionSet keys give for
    {a0, a2} = {a1, a2'}), P1 and P2 with M as their only action (P2 with one more unexpanded action if extra_p2)."""
    t = po.PyTree()
    h = np.zeros(8, F)
    r = t.add_node(frozenset(), F(c_r))
    t.add_actions(r, [0, 1], h)
    p1 = t.add_node(frozenset({0}), F(c_p1))
    t.add_edge(r, p1, 0)
    t.add_actions(p1, [2], h)
    p2 = t.add_node(frozenset({1}), F(c_p2))
    t.add_edge(r, p2, 1)
    t.add_actions(p2, [3, 4] if extra_p2 else [3], h)
    m = t.add_node(frozenset({0, 2}), F(c_m))
    t.add_edge(p1, m, t.node[p1]["a0"])
    t.add_edge(p2, m, t.node[p2]["a0"])  # the transposition arc: M's second in-neighbour
    t.add_actions(m, [5], h)
    return t, (r, p1, p2, m)


def test_two_parent_cascade_new_terminal_by_hand():
    """empty_transitions.rs:50-87 through a node with TWO in-neighbours.  M's only action leads to a new terminal T
    with c_T = 0.25.  By hand: sweep {M: (0.25, 1)}: M exhausted 1, c* 0.5 > 0.25 -> 0.25 (n_t stays 0), now inactive ->
    (0.25, 1) to both in-neighbours (:75-85).  Level {P1, P2}, smallest index first (:62 pop_first): P1 exhausted 1,
    c* 0.4 -> 0.25, inactive -> R gets (0.25, 1); P2 exhausted 1, c* 0.2 is NOT improved -> n_t 1 (:66-70), inactive ->
    R's entry is UPDATED: c = min(0.25, 0.25), n = 1 + 1 (:80-84).  R: exhausted 2, c* 0.3 -> 0.25, inactive; no
    in-neighbours, the sweep ends."""
    t, (r, p1, p2, m) = _diamond(0.3, 0.4, 0.2, 0.5)
    tt = t.add_node(frozenset({0, 2, 5}), F(0.25))
    e = t.add_edge(m, tt, t.node[m]["a0"])
    t.cascade(e, old=False)
    got = [(nd["x"], nd["n"], nd["cs"]) for nd in t.node]
    assert got == [(2, 0, F(0.25)), (1, 0, F(0.25)), (1, 1, F(0.2)), (1, 0, F(0.25)), (0, 0, F(0.25))]
    assert [t.active(i) for i in (r, p1, p2, m)] == [False] * 4


def test_two_parent_cascade_old_node_by_hand():
    """empty_transitions.rs:89-127.  M's only action hits an EXISTING inactive node X (c* 0.35, n_t 7): initial info
    (0.35, 1) (:92-99).  M: exhausted 1, c* 0.5 -> 0.35, then n_t = max(0, 7) = 7 (:110), inactive -> (0.35, 1) to P1, P2.
    P1: exhausted 1, c* 0.4 -> 0.35, n_t = max(0, 7) = 7, inactive -> R (0.35, 1).  P2 (one more unexpanded action):
    exhausted 1 of 2, c* 0.2 not improved -> n_t 1, then max(1, 7) = 7; still ACTIVE -> its info carries n = 0, R's
    entry becomes (0.35, 1 + 0).  R: exhausted 1 of 2, c* 0.3 not improved -> n_t 1 -> max(1, 7) = 7; active."""
    t, (r, p1, p2, m) = _diamond(0.3, 0.4, 0.2, 0.5, extra_p2=True)
    x = t.add_node(frozenset({9}), F(0.35))  # an inactive node somewhere else in the tree (no actions = inactive)
    t.node[x]["n"] = 7
    e = t.add_edge(m, x, t.node[m]["a0"])
    t.cascade(e, old=True)
    got = [(nd["x"], nd["n"], nd["cs"]) for nd in t.node[:4]]
    assert got == [(1, 7, F(0.3)), (1, 7, F(0.35)), (1, 7, F(0.2)), (1, 7, F(0.35))]
    assert [t.active(i) for i in (r, p1, p2, m)] == [True, False, True, False]


def test_a_sweep_never_meets_an_already_inactive_ancestor():
    """empty_transitions.rs:69-73 sends `1` for every visited node that is inactive AFTER its update, whether or not it
    was inactive before.  Re-counting would need a sweep to visit a node that was already inactive -- which cannot
    happen: a node turns inactive only when all its actions are counted, every counted child being inactive itself, so
    an inactive node has no active descendant; a sweep starts at the node the agent stands on (active: it has just
    offered an unvisited action) and visits that node's ancestors only.  Checked here over random searches (the merge of
    two infos for one ancestor, by contrast, happens all the time)."""
    hits = dict(visits=0, inactive_before=0, merges=0)
    orig = po.PyTree.cascade

    def watched(self, e, old):
        u0, t, _ = self.edge[e]
        level = {u0}
        seen_before = {u: self.active(u) for u in range(len(self.node))}
        while level:  # the ancestors the sweep will visit, level by level
            nxt = set()
            for u in level:
                hits["visits"] += 1
                hits["inactive_before"] += 0 if seen_before[u] else 1
                parents = [self.edge[ie][0] for ie in self.inn[u]]
                hits["merges"] += len(parents) - len(set(parents) - nxt) if parents else 0
                nxt.update(parents)
            level = nxt
        return orig(self, e, old)

    po.PyTree.cascade = watched
    try:
        rng = np.random.default_rng(0)
        for n in (5, 6, 7):
            _, A = po.dims(n)
            for _ in range(12):
                e = po.PyEngine(n, 1)
                parents = [0] * n
                for v in range(2, n - 1):
                    parents[v] = int(rng.integers(0, v))
                perm = set(int(x) for x in rng.choice(A, int(rng.integers(2, A + 1)), replace=False))
                e.new_begin([(parents, perm)])
                e.new_end(rng.random((1, A), dtype=np.float32))
                for _ in range(120):
                    e.rollout_begin([3, 2], 1)
                    e.rollout_end(rng.random((1, A), dtype=np.float32))
    finally:
        po.PyTree.cascade = orig
    assert hits["visits"] > 2000 and hits["merges"] > 100 and hits["inactive_before"] == 0, hits


def test_the_c21_spaces_actions_commute_which_the_device_builds_states_on():
    """The device no longer applies a descent's actions one by one: a new node's state is root + {the path's action set}, applied at
    once (C21Space::act_set).  That is the reference's state only because the space's actions commute and never repeat
    (ActionOrderIndependent + ActionsNeverRepeat, rooted_tree/space.rs:124-125; space/axioms.rs:16-19).  Checked here on the Python
    restatement of `act` (space.rs:56-73): every legal sequence from seeded roots, taken step by step, ends on the state that ANY
    order of its action set gives, and no action of a path is legal again further down it."""
    import random

    from oracle import py_oracle as po

    rng = random.Random(5)
    for n in (8, 13, 19):
        S, A = po.dims(n)
        for agent in range(40):
            parents, permitted = po.fresh_root(1, 0, agent, n, rng.randint(5, A // 2))
            parents, permitted = list(parents), set(permitted)
            p0, m0 = list(parents), set(permitted)
            taken = []
            while True:
                legal = po.legal_actions(n, parents, permitted)
                if not legal or len(taken) >= 16:
                    break
                a = rng.choice(legal)
                assert a not in taken
                po.act(parents, permitted, a)
                taken.append(a)
                assert a not in po.legal_actions(n, parents, permitted)
            for _ in range(4):
                order = taken[:]
                rng.shuffle(order)
                p1, m1 = list(p0), set(m0)
                for a in order:
                    po.act(p1, m1, a)
                assert p1 == parents and m1 == permitted, (n, agent, taken, order)
