"""Second, independent CPU restatement of the hot path (pure Python + numpy f32).

TEST INFRASTRUCTURE ONLY.  Written from the behavioural spec in SURVEY.md
Appendix A / section 8 (which cites the reference file:line for every rule), NOT from
oracle/azd_oracle.cpp: different containers (dict keyed by frozenset, per-node
Python lists of edges iterated in reverse), so an agreement of exported trees
between the two restatements pins the tree/optimizer semantics that the
reference itself has no fixture for.  Small cases only (pure-Python loops).

Reference files followed (relative to /root/reference):
  az-discrete-opt/src/nabla/tree/{mod,next_action,empty_transitions,graph_operations,state_weight}.rs
  az-discrete-opt/src/nabla/optimizer/mod.rs
  graph-state/src/rooted_tree/{mod,modify_parent_once,ordered_edge,space}.rs, simple_graph/edge.rs
  graph-state/examples/04-c21-tree.rs
"""
import math

import numpy as np

F = np.float32
M64 = (1 << 64) - 1


# ---------------------------------------------------------------- generators
def splitmix(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def key4(a, b, c, d):
    return splitmix(splitmix(splitmix(splitmix(a & M64) ^ (b & M64)) ^ (c & M64)) ^ (d & M64))


def below(r, n):
    return ((r >> 32) * n) >> 32


D_ROOT, D_PRED, D_RESET = 0x726F6F74, 0x70726564, 0x72657365


def dims(n):
    return (n - 1) * (n - 2) - 2, (n - 1) * (n - 2) // 2 - 1  # STATE_DIM, ACTION_DIM (space.rs:46-48)


def shuffle_prefix(seed, domain, agent, A, k):
    perm = list(range(A))
    for j in range(k):
        r = j + below(key4(seed, domain, agent, 64 + j), A - j)
        perm[j], perm[r] = perm[r], perm[j]
    return set(perm[:k])


def fresh_root(seed, domain, agent, n, k):
    _, A = dims(n)
    parents = [0] * n
    for v in range(2, n - 1):
        parents[v] = below(key4(seed, domain, agent, v), v)
    return parents, shuffle_prefix(seed, domain, agent, A, k)


def gen_root(seed, epoch, agent, n, kmin, kmax):
    domain = D_ROOT ^ ((epoch << 32) & M64)
    k = kmin + below(key4(seed, domain, agent, 0), kmax - kmin + 1)
    return fresh_root(seed, domain, agent, n, k)


def hash_prediction_row(seed, agent, call, A):
    return np.array([F(key4(seed ^ D_PRED, agent, call, a) >> 40) * F(1.0 / 16777216.0) for a in range(A)], F)


# ---------------------------------------------------------------- c21 space
def edge_index(parent, child):  # ordered_edge.rs:35-38 via edge.rs:48-53
    return child * (child - 1) // 2 + parent - 1


def edge_from_index(idx):  # ordered_edge.rs:40-42 via edge.rs:55-65
    pos, v = idx + 1, 1
    while pos >= v * (v + 1) // 2:
        v += 1
    return v - (v * (v + 1) // 2 - pos), v  # (parent, child)


def act(parents, permitted, a):  # space.rs:56-73
    p, ch = edge_from_index(a)
    parents[ch] = p
    for u in range(ch):
        permitted.discard(edge_index(u, ch))


def legal_actions(n, parents, permitted):  # space.rs:75-89
    cur = {edge_index(parents[c], c) for c in range(2, n - 1)}
    return [a for a in sorted(permitted) if a not in cur]


def write_vec(n, parents, permitted):  # space.rs:91-101
    S, A = dims(n)
    v = np.zeros(S, F)
    for c in range(2, n - 1):
        v[edge_index(parents[c], c)] = 1
    for a in permitted:
        v[A + a] = 1
    return v


def posdef(parents, n, x):
    # division-free subtree characteristic polynomials (DESIGN.md "lambda_1"): phi_v > 0 for all v
    P = [1.0] * n
    Q = [0.0] * n
    ok = True
    for v in range(n - 1, 0, -1):
        phi = x * P[v] - Q[v]
        if not phi > 0.0:
            ok = False
        p = parents[v]
        Q[p] = Q[p] * phi + P[p] * P[v]
        P[p] = P[p] * phi
    phi0 = x * P[0] - Q[0]
    if not phi0 > 0.0:
        ok = False
    return ok, phi0


def lambda1(parents, n, node_mode=False):
    """cost contract for lambda_1: 33-section of the bracket, <= 15 rounds; node_mode stops as
    soon as both bracket ends round to the same f32 (what the evaluation uses)"""
    # initial bracket: among the trees on n vertices the path has the smallest lambda_1, 2 cos(pi / (n + 1)), the star the
    # largest, sqrt(n - 1); both rounded to f32 and widened by 2^-20 (so that every restatement starts from the same doubles)
    lo = float(F(2.0 * math.cos(math.pi / (n + 1)))) - 2.0 ** -20
    hi = float(F(math.sqrt(float(n - 1)))) + 2.0 ** -20
    # round 4: with phi_0 (the tree's characteristic polynomial) known at both ends and of the signs of a simple crossing, the
    # round's trial points go into a window around the secant's estimate of the root (half-width 8 (span / 2)^2, at least
    # span / 1024, clipped to the bracket); the ends are only ever replaced by trial points, so the bracket stays a bracket.
    # round 5: the third window of a solve that misses the root (all 32 points on one side of it) switches the window off for the
    # rest of the solve; every round that is no miss shrinks the bracket 33-fold at least, so 15 rounds give what 12 plain rounds
    # give (33^12 > 2^60)
    flo = fhi = 0.0
    have_lo = have_hi = False
    misses = 0
    for _ in range(15):
        if node_mode and F(lo) == F(hi):
            break
        wlo, whi = lo, hi
        in_window = False
        if misses < 3 and have_lo and have_hi and flo < 0.0 and fhi > 0.0:
            span = hi - lo
            den = flo - fhi
            tt = flo / den
            st = span * tt
            c = lo + st
            half = span * 0.5
            hh = half * half
            d = 8.0 * hh
            dmin = span * 0.0009765625
            if d < dmin:
                d = dmin
            if d < half:
                wa, wb = c - d, c + d
                if wa > lo:
                    wlo = wa
                if wb < hi:
                    whi = wb
                in_window = True
        ws = whi - wlo
        w = ws / 33.0
        xs = [wlo + w * float(j + 1) for j in range(32)]
        ph = [0.0] * 32
        first = 32
        for j in range(32):
            ok, ph[j] = posdef(parents, n, xs[j])
            if ok:
                first = j
                break
        if first > 0:
            lo, flo, have_lo = xs[first - 1], ph[first - 1], True
        if first < 32:
            hi, fhi, have_hi = xs[first], ph[first], True
        if in_window and (first == 0 or first == 32):
            misses += 1
    return hi


def matching_size(parents, n):  # ordered_edge.rs:94-124
    avail = [True] * n
    m = 0
    while True:
        leaf = list(avail)
        for i in range(1, n):
            if avail[i]:
                leaf[parents[i]] = False
        for i in range(1, n):
            if leaf[i]:
                avail[i] = False
                if avail[parents[i]]:
                    avail[parents[i]] = False
                    m += 1
        if sum(avail) < 2:
            return m


def evaluate(n, lam, mu):  # 04-c21-tree.rs:58-74,98-102
    r = int(np.floor(np.sqrt(n - 1)))
    while (r + 1) * (r + 1) <= n - 1:
        r += 1
    while r * r > n - 1:
        r -= 1
    sq = r if r * r == n - 1 else r + 1
    upper = sq + (n + 1) // 2
    slope = F(1.0) / F(upper - 2)
    return slope * ((F(mu) + F(lam)) - F(2))


def cost_eval(n, parents, full=False):
    lam = lambda1(parents, n, node_mode=not full)
    mu = matching_size(parents, n)
    return lam, mu, evaluate(n, lam, mu)


# ---------------------------------------------------------------- search tree
class PyTree:
    def __init__(self):
        self.pos = {}      # frozenset(action ids) -> node index
        self.node = []     # dicts: c, cs, n, x, a0, a1
        self.out = []      # per node: edge ids in creation order (iterate reversed = newest first)
        self.inn = []
        self.edge = []     # (src, dst, pp)
        self.pred = []     # [a_id, g, edge or None]

    def add_node(self, key, c):
        self.node.append(dict(c=c, cs=c, n=0, x=0, a0=0, a1=0))
        self.out.append([])
        self.inn.append([])
        self.pos[key] = len(self.node) - 1
        return len(self.node) - 1

    def add_edge(self, u, v, pp):
        self.edge.append((u, v, pp))
        e = len(self.edge) - 1
        self.out[u].append(e)
        self.inn[v].append(e)
        self.pred[pp][2] = e
        return e

    def active(self, i):
        nd = self.node[i]
        return nd["a0"] + nd["x"] < nd["a1"]

    def add_actions(self, i, acts, h_row):
        nd = self.node[i]
        nd["a0"] = len(self.pred)
        for a in acts:
            self.pred.append([a, nd["c"] - F(h_row[a]), None])  # g = c_s - h (04-c21-tree.rs:103)
        nd["a1"] = len(self.pred)

    def select(self, i, tol):
        if not self.active(i):
            return None
        nd = self.node[i]
        r = None
        for e in reversed(self.out[i]):
            k = self.edge[e][1]
            if not self.active(k):
                continue
            cand = (self.node[k]["n"], self.node[k]["cs"], e)
            if r is None or (cand[0], cand[1]) < (r[0], r[1]):
                r = cand
        if r is not None and r[0] < tol:
            return ("V", r[2])
        kids = [self.node[self.edge[e][1]]["cs"] for e in reversed(self.out[i])]
        best = None
        for pp in range(nd["a0"], nd["a1"]):
            if self.pred[pp][2] is not None:
                continue
            v = nd["c"] - self.pred[pp][1]
            if not kids:
                if best is None or v < best[0]:
                    best = (v, pp)
            else:
                s = F(0)
                for k in kids:
                    s = s + np.sqrt(np.abs(k - v))
                if best is None or not (s < best[0]):
                    best = (s, pp)
        if best is not None:
            return ("U", best[1])
        if r is not None:
            return ("V", r[2])
        return None

    def cascade(self, e, old):
        u0, t, _ = self.edge[e]
        tn = self.node[t]
        n_t_target = tn["n"]
        cur = {u0: [tn["cs"], (0 if self.active(t) else 1) if old else 1]}
        while cur:
            nxt = {}
            for u in sorted(cur):
                c, x = cur[u]
                nd = self.node[u]
                nd["x"] += x
                if nd["cs"] > c:
                    nd["cs"] = c
                else:
                    nd["n"] += 1
                if old:
                    nd["n"] = max(nd["n"], n_t_target)
                up_x = 0 if self.active(u) else 1
                for ie in reversed(self.inn[u]):
                    p = self.edge[ie][0]
                    if p in nxt:
                        nxt[p][0] = min(nxt[p][0], c)
                        nxt[p][1] += up_x
                    else:
                        nxt[p] = [c, up_x]
            cur = nxt


class PyEngine:
    """NablaOptimizer<ROTModifyParentsOnce<N>, M, ActionSet> with an injectable model."""

    seq = False  # path encoding: False = ActionSet / ActionMultiset (path/set.rs, multiset.rs),
    #              True = ActionSequence / OrderedActionSet (path/sequence.rs, ord_set.rs: a Vec in push order)

    def __init__(self, n, batch, seq=False, layers=1):
        self.n, self.B, self.seq = n, batch, seq
        self.S, self.A = dims(n)
        self.set_layers(layers)

    # ---- Layered<L, Space> (space/layered.rs, nabla/space/mod.rs:41-111): the state is a ring of the last
    # L states (state/layers.rs); self.states[i] is back(), self.older[i] the states before it (oldest
    # first, at most L - 1).  The state-vector rows persist across calls; only the chunks of states the
    # ring holds are rewritten.
    def set_layers(self, layers):
        self.L = layers
        self.S_inner = self.S
        self.S = self.S_inner * layers
        self.vecs = np.zeros((self.B, self.S), F)
        self.older = [[] for _ in range(self.B)]

    def clone_state(self, st):
        return (list(st[0]), set(st[1]))

    def inner_vec(self, st):
        return write_vec(self.n, *st)

    def push_layer(self, i):
        """Layered::act pushes a clone of back() before acting on it (nabla/space/mod.rs:65-71)"""
        if self.L > 1:
            self.older[i].append(self.clone_state(self.states[i]))
            if len(self.older[i]) > self.L - 1:
                self.older[i].pop(0)

    def write_row(self, i, back, older):
        k = self.S_inner
        for j, st in enumerate(list(older) + [back]):
            self.vecs[i, j * k:(j + 1) * k] = self.inner_vec(st)

    def key(self, path):
        """the transposition key P of a path (tree/mod.rs:29 BTreeMap<P, NodeIndex>)"""
        return tuple(path) if self.seq else frozenset(path)

    def actions_taken(self, key):
        """ActionPath::actions_taken, which is also what the derived Ord compares lexicographically"""
        return list(key) if self.seq else sorted(key)

    def new_begin(self, roots):  # roots: list of (parents list, permitted set)
        self.roots = [(list(p), set(m)) for p, m in roots]
        self.states = [(list(p), set(m)) for p, m in roots]
        self.costs = [cost_eval(self.n, p) for p, _ in self.roots]
        self.paths = [[] for _ in roots]
        self.posn = [0] * self.B
        self.inspected = [0] * self.B
        self.older = [[] for _ in range(self.B)]  # Layers::new: a ring of one
        for i in range(self.B):
            self.write_row(i, self.states[i], [])

    def _root_tree(self, i, h_row):
        t = PyTree()
        t.add_node(self.key([]), self.costs[i][2])
        t.add_actions(0, legal_actions(self.n, *self.roots[i]), h_row)
        return t

    def new_end(self, h):
        self.trees = [self._root_tree(i, h[i]) for i in range(self.B)]
        best = min(range(self.B), key=lambda i: (self.costs[i][2], i))
        lam, mu, _ = cost_eval(self.n, self.states[best][0], full=True)
        self.argmin = dict(parents=list(self.states[best][0]), permitted=set(self.states[best][1]),
                           lambda1=lam, matching=mu, eval=self.costs[best][2])

    def _step(self, i, tol, tol_default):
        t, n = self.trees[i], self.n
        parents, permitted = self.states[i]
        path = self.paths[i]
        while True:
            tl = tol[len(path)] if len(path) < len(tol) else tol_default
            ch = t.select(self.posn[i], tl)
            if ch is None:
                assert not path
                return
            if ch[0] == "V":
                _, dst, pp = t.edge[ch[1]]
                a = t.pred[pp][0]
                path.append(a)
                self.push_layer(i)
                act(parents, permitted, a)
                self.posn[i] = dst
                continue
            pp = ch[1]
            a = t.pred[pp][0]
            path.append(a)
            key = self.key(path)
            hit = t.pos.get(key)
            if hit is not None:
                e = t.add_edge(self.posn[i], hit, pp)
                t.cascade(e, True)
            else:
                self.push_layer(i)
                act(parents, permitted, a)
                self.costs[i] = cost_eval(n, parents)
                v = t.add_node(key, self.costs[i][2])
                e = t.add_edge(self.posn[i], v, pp)
                if legal_actions(n, parents, permitted):
                    self.posn[i] = v
                    return
                t.cascade(e, False)
            parents[:] = self.roots[i][0]
            permitted.clear()
            permitted.update(self.roots[i][1])
            self.older[i].clear()  # state.clone_from(root): the root's ring holds one state
            path.clear()
            self.posn[i] = 0

    def rollout_begin(self, tol, tol_default):
        for i in range(self.B):
            self._step(i, tol, tol_default)
            if self.paths[i]:
                self.write_row(i, self.states[i], self.older[i])

    def rollout_end(self, h):
        for i in range(self.B):
            if self.paths[i]:
                self.trees[i].add_actions(self.posn[i], legal_actions(self.n, *self.states[i]), h[i])
        return self._update_argmin()

    def _update_argmin(self):  # optimizer/mod.rs:194-246; cross-tree ties -> lowest tree index
        best = None
        for i, t in enumerate(self.trees):
            for j in range(self.inspected[i], len(t.node)):
                c = t.node[j]["c"]
                if c < self.argmin["eval"] and (best is None or c < best[0]):
                    best = (c, i, j)
            self.inspected[i] = len(t.node)
        if best is None:
            return 0
        _, i, j = best
        parents, permitted = list(self.roots[i][0]), set(self.roots[i][1])
        key = next(k for k, v in self.trees[i].pos.items() if v == j)
        for a in self.actions_taken(key):
            act(parents, permitted, a)
        lam, mu, ev = cost_eval(self.n, parents, full=True)
        self.argmin = dict(parents=parents, permitted=permitted, lambda1=lam, matching=mu, eval=ev)
        return 1

    def observe(self, n_obs_tol):  # optimizer/mod.rs:262-278 + tree/mod.rs:242-264
        obs = np.zeros((self.B, self.A), F)
        w = np.zeros((self.B, self.A), F)
        for i, t in enumerate(self.trees):
            self.write_row(i, self.roots[i], [])
            for e in reversed(t.out[0]):
                _, k, pp = t.edge[e]
                if (not t.active(k)) or t.node[k]["n"] >= n_obs_tol:
                    obs[i, t.pred[pp][0]] = t.node[k]["cs"]  # h_sa = c_child* (04-c21-tree.rs:104)
                    w[i, t.pred[pp][0]] = 1
        return obs, w

    def modify_roots(self, seed, epoch, first_agent, kmin, kmax):  # 04-c21-tree.rs:172-206, seeded
        domain = D_RESET ^ ((epoch << 32) & M64)
        out = []
        for i, t in enumerate(self.trees):
            agent = first_agent + i
            r0, r1 = key4(seed, domain, agent, 0), key4(seed, domain, agent, 1)
            parents, permitted = list(self.roots[i][0]), set(self.roots[i][1])
            order = sorted(t.pos.items(), key=lambda kv: self.actions_taken(kv[0]))  # BTreeMap order: lexicographic
            c_root, c_root_star = t.node[0]["c"], t.node[0]["cs"]
            if c_root == c_root_star:
                kcur = len(permitted)
                if kcur == kmax:
                    out.append(fresh_root(seed, domain, agent, self.n, kmin + below(r1, kmax - kmin + 1)))
                    continue
                keep = [k for k, v in order if t.node[v]["c"] == c_root]
                k_new = kcur + below(r1, kmax - kcur + 1)
            else:
                thr = (c_root + F(3.0) * c_root_star) / F(4.0)
                keep = [k for k, v in order if t.node[v]["c"] <= thr]
                k_new = kmin + below(r1, kmax - kmin + 1)
            for a in self.actions_taken(keep[below(r0, len(keep))]):
                act(parents, permitted, a)
            out.append((parents, shuffle_prefix(seed, domain, agent, self.A, k_new)))
        return out

    def reset_begin(self, roots):
        self.roots = [(list(p), set(m)) for p, m in roots]
        self.states = [(list(p), set(m)) for p, m in roots]
        self.costs = [cost_eval(self.n, p) for p, _ in self.roots]
        self.paths = [[] for _ in roots]
        self.posn = [0] * self.B
        self.older = [[] for _ in range(self.B)]  # Layers::new: a ring of one
        for i in range(self.B):
            self.write_row(i, self.states[i], [])

    def reset_end(self, h):
        self.trees = [self._root_tree(i, h[i]) for i in range(self.B)]
        self.inspected = [0] * self.B

    def export_tree(self, i, kw):
        t = self.trees[i]
        keys = np.zeros((len(t.node), kw), np.uint64)
        for k, v in t.pos.items():
            for a in k:
                keys[v, a >> 6] |= np.uint64(1 << (a & 63))
        return dict(
            c=np.array([nd["c"] for nd in t.node], F), c_star=np.array([nd["cs"] for nd in t.node], F),
            n_t=np.array([nd["n"] for nd in t.node], np.uint32), exhausted=np.array([nd["x"] for nd in t.node], np.uint32),
            act_begin=np.array([nd["a0"] for nd in t.node], np.uint32), act_end=np.array([nd["a1"] for nd in t.node], np.uint32),
            keys=keys, e_src=np.array([e[0] for e in t.edge], np.uint32), e_dst=np.array([e[1] for e in t.edge], np.uint32),
            e_pp=np.array([e[2] for e in t.edge], np.uint32), p_aid=np.array([p[0] for p in t.pred], np.uint32),
            p_g=np.array([p[1] for p in t.pred], F), p_edge=np.array([-1 if p[2] is None else p[2] for p in t.pred], np.int32))
