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
