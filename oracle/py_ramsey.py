"""Second, independent CPU restatement of the Ramsey space under the search tree (pure Python).

TEST INFRASTRUCTURE ONLY.  It shares oracle/py_oracle.py's tree (selection, cascade -- written from
SURVEY.md Appendix A) and restates the space (graph-state/src/ramsey_counts/space.rs:40-176) in the
most literal way available: the per-edge counts are RECOUNTED FROM THE DEFINITION after every action
(RamseyCounts::new, ramsey_counts/mod.rs:20-68, by enumerating vertex subsets), never maintained
incrementally -- which is exactly the property the reference's own tests assert
(space.rs:205-283).  oracle/azd_oracle.cpp and the device kernels maintain them incrementally, so
agreement of exported trees pins both the incremental update and the space/tree glue.
Small cases only (N <= 9)."""
import itertools

import numpy as np

from . import py_oracle as po

F = np.float32


def edges(n):
    return [(v, u) for v in range(n) for u in range(v)]  # colex order, simple_graph/edge.rs:67-69


class RamseyState:
    def __init__(self, n, sizes, colors, permitted):
        self.n, self.sizes = n, sizes
        self.colors = list(colors)      # colour per colex edge position
        self.permitted = set(permitted)  # permitted_edges (no_recolor.rs:12)
        self.recount()

    def clone(self):
        s = RamseyState.__new__(RamseyState)
        s.n, s.sizes, s.colors, s.permitted = self.n, self.sizes, list(self.colors), set(self.permitted)
        s.counts, s.totals = [list(r) for r in self.counts], list(self.totals)
        return s

    def recount(self):
        n, C = self.n, len(self.sizes)
        ed = edges(n)
        adj = [[[False] * n for _ in range(n)] for _ in range(C)]
        for e, (v, u) in enumerate(ed):
            adj[self.colors[e]][v][u] = adj[self.colors[e]][u][v] = True

        def clique(c, vs):
            return all(adj[c][a][b] for a, b in itertools.combinations(vs, 2))

        self.counts, self.totals = [], []
        for c, k in enumerate(self.sizes):
            row = []
            for v, u in ed:
                common = [w for w in range(n) if adj[c][v][w] and adj[c][u][w]]
                row.append(sum(1 for vs in itertools.combinations(common, k - 2) if clique(c, vs)))
            self.counts.append(row)
            self.totals.append(sum(1 for vs in itertools.combinations(range(n), k) if clique(c, vs)))

    def act(self, a):  # space.rs:48-54, :71-86
        E = len(self.colors)
        e, nc = a % E, a // E
        self.colors[e] = nc
        self.permitted.discard(e)
        self.recount()


class RamseyTree(po.PyTree):
    def add_actions_r(self, i, acts, h_row, c_weights):
        """graph_operations.rs:32-56 with g = c_s h + r (1 - h) (space.rs:167-172)"""
        nd = self.node[i]
        nd["a0"] = len(self.pred)
        for a, (oc, ncol, old_count, new_count) in acts:
            r = F(F(old_count) * c_weights[oc] - F(new_count) * c_weights[ncol])
            h = F(h_row[a])
            self.pred.append([a, F(F(nd["c"] * h) + F(r * F(F(1) - h))), None])
        nd["a1"] = len(self.pred)


class PyRamseyEngine(po.PyEngine):
    """NablaOptimizer<RamseySpaceNoEdgeRecolor<B32, N, E, C>, M, ActionSet> with an injectable model."""

    def __init__(self, n, sizes, weights, batch, seq=False, layers=1):
        self.n, self.B, self.sizes, self.seq = n, batch, list(sizes), seq
        self.w = [F(x) for x in weights]
        self.C, self.E = len(sizes), n * (n - 1) // 2
        self.S, self.A = self.E * (2 * self.C + 1), self.E * self.C
        self.set_layers(layers)

    def clone_state(self, st):
        return st.clone()

    def inner_vec(self, st):
        return self.write_vec(st)

    # ---- space
    def evaluate(self, st):  # space.rs:159-165
        s = F(0)
        for c in range(self.C):
            s = F(s + F(F(st.totals[c]) * self.w[c]))
        return s

    def action_data(self, st):  # space.rs:88-120: edges ascending, then new colours ascending
        out = []
        for e in range(self.E):
            if e not in st.permitted:
                continue
            oc = st.colors[e]
            for nc in range(self.C):
                if nc != oc:
                    out.append((e + nc * self.E, (oc, nc, st.counts[oc][e], st.counts[nc][e])))
        return out

    def write_vec(self, st):  # space.rs:122-153
        v = np.zeros(self.S_inner, F)
        C, E = self.C, self.E
        for c in range(C):
            for e in range(E):
                v[c * E + e] = st.counts[c][e]
                v[C * E + c * E + e] = 1 if st.colors[e] == c else 0
        for e in st.permitted:
            v[2 * C * E + e] = 1
        return v

    # ---- optimizer
    def new_begin(self, roots):  # roots: list of (colors, permitted edge set)
        self.roots = [RamseyState(self.n, self.sizes, c, m) for c, m in roots]
        self.states = [r.clone() for r in self.roots]
        self.costs = [self.evaluate(r) for r in self.roots]
        self.paths = [[] for _ in roots]
        self.posn = [0] * self.B
        self.inspected = [0] * self.B
        self.older = [[] for _ in range(self.B)]
        for i in range(self.B):
            self.write_row(i, self.states[i], [])

    def _root_tree(self, i, h_row):
        t = RamseyTree()
        t.add_node(self.key([]), self.costs[i])
        t.add_actions_r(0, self.action_data(self.roots[i]), h_row, self.w)
        return t

    def new_end(self, h):
        self.trees = [self._root_tree(i, h[i]) for i in range(self.B)]
        best = min(range(self.B), key=lambda i: (self.costs[i], i))
        self.argmin = dict(state=self.states[best].clone(), eval=self.costs[best])

    def _step(self, i, tol, tol_default):
        t = self.trees[i]
        st = self.states[i]
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
                st.act(a)
                self.posn[i] = dst
                continue
            pp = ch[1]
            a = t.pred[pp][0]
            path.append(a)
            key = self.key(path)
            hit = t.pos.get(key)
            if hit is not None:
                t.cascade(t.add_edge(self.posn[i], hit, pp), True)
            else:
                self.push_layer(i)
                st.act(a)
                self.costs[i] = self.evaluate(st)
                v = t.add_node(key, self.costs[i])
                e = t.add_edge(self.posn[i], v, pp)
                if st.permitted:  # not terminal
                    self.posn[i] = v
                    return
                t.cascade(e, False)
            st = self.states[i] = self.roots[i].clone()
            self.older[i].clear()
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
                self.trees[i].add_actions_r(self.posn[i], self.action_data(self.states[i]), h[i], self.w)
        best = None
        for i, t in enumerate(self.trees):  # optimizer/mod.rs:194-246; cross-tree ties -> lowest tree index
            for j in range(self.inspected[i], len(t.node)):
                c = t.node[j]["c"]
                if c < self.argmin["eval"] and (best is None or c < best[0]):
                    best = (c, i, j)
            self.inspected[i] = len(t.node)
        if best is None:
            return 0
        _, i, j = best
        st = self.roots[i].clone()
        for a in self.actions_taken(next(k for k, v in self.trees[i].pos.items() if v == j)):
            st.act(a)
        self.argmin = dict(state=st, eval=self.evaluate(st))
        return 1

    def observe(self, n_obs_tol):  # tree/mod.rs:242-264 with h_sa = 1 - c*/c (space.rs:174-177)
        obs = np.zeros((self.B, self.A), F)
        w = np.zeros((self.B, self.A), F)
        for i, t in enumerate(self.trees):
            self.write_row(i, self.roots[i], [])
            for e in reversed(t.out[0]):
                _, k, pp = t.edge[e]
                if (not t.active(k)) or t.node[k]["n"] >= n_obs_tol:
                    with np.errstate(divide="ignore", invalid="ignore"):
                        obs[i, t.pred[pp][0]] = F(F(1) - F(t.node[k]["cs"] / t.node[k]["c"]))
                    w[i, t.pred[pp][0]] = 1
        return obs, w

    def modify_roots(self, seed, epoch, first_agent, kmin, kmax):  # 02-r44.rs:196-228, seeded
        domain = po.D_RESET ^ ((epoch << 32) & po.M64)
        out = []
        for i, t in enumerate(self.trees):
            agent = first_agent + i
            r0, r1 = po.key4(seed, domain, agent, 0), po.key4(seed, domain, agent, 1)
            st = self.roots[i].clone()
            order = sorted(t.pos.items(), key=lambda kv: self.actions_taken(kv[0]))  # BTreeMap order
            c_root, c_root_star = t.node[0]["c"], t.node[0]["cs"]
            if c_root == c_root_star:
                kcur = len(st.permitted)
                if kcur == kmax:
                    k = kmin + po.below(r1, kmax - kmin + 1)
                    colors = [po.below(po.key4(seed, domain, agent, 1024 + e), self.C) for e in range(self.E)]
                    out.append((colors, po.shuffle_prefix(seed, domain, agent, self.E, k)))
                    continue
                keep = [k for k, v in order if t.node[v]["c"] == c_root]
                k_new = kcur + po.below(r1, kmax - kcur + 1)
            else:
                thr = (c_root + F(3.0) * c_root_star) / F(4.0)
                keep = [k for k, v in order if t.node[v]["c"] <= thr]
                k_new = kmin + po.below(r1, kmax - kmin + 1)
            for a in self.actions_taken(keep[po.below(r0, len(keep))]):
                st.act(a)
            out.append((list(st.colors), po.shuffle_prefix(seed, domain, agent, self.E, k_new)))
        return out

    def reset_begin(self, roots):
        self.roots = [RamseyState(self.n, self.sizes, c, m) for c, m in roots]
        self.states = [r.clone() for r in self.roots]
        self.costs = [self.evaluate(r) for r in self.roots]
        self.paths = [[] for _ in roots]
        self.posn = [0] * self.B
        self.older = [[] for _ in range(self.B)]
        for i in range(self.B):
            self.write_row(i, self.states[i], [])
