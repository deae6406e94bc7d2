"""ctypes binding of the CPU oracle (oracle/libazd_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (azdopt_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

CTR = dict(EXPANSIONS=0, TERMINALS=1, TRANSPOSITIONS=2, VISITED_STEPS=3, SELECT_CALLS=4, SUM_DEG=5,
           SUM_ACTIONS=6, CASCADE_NODES=7, NEW_PREDS=8, ROOT_EXHAUSTED=9, MAX_FRONTIER=10, MAX_DEPTH=11,
           CURIOSITY_PAIRS=12, FAILED=15)
CTR_COUNT = 16


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "libazd_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    u8p, u32p, u64p, i32p, f32p, f64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_uint64, C.c_int32,
                                                                 C.c_float, C.c_double))
    vp = C.c_void_p

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_state_dim", C.c_int, C.c_int)
    sig("orc_action_dim", C.c_int, C.c_int)
    sig("orc_key_words", C.c_int, C.c_int)
    sig("orc_edge_colex_position", C.c_int, C.c_int, C.c_int)
    sig("orc_edge_from_colex_position", None, C.c_int, i32p, i32p)
    sig("orc_action_index", C.c_int, C.c_int, C.c_int)
    sig("orc_action_from_index", None, C.c_int, i32p, i32p)
    sig("orc_all_possible_parent_modifications", C.c_int, vp, C.c_int, vp)
    sig("orc_lambda1_jacobi", C.c_double, vp, C.c_int)
    sig("orc_lambda1_sturm", C.c_double, vp, C.c_int)
    sig("orc_lambda1_node", C.c_double, vp, C.c_int)
    sig("orc_lambda1_plain", C.c_double, vp, C.c_int, C.c_int)
    sig("orc_lambda1_rounds", C.c_int, vp, C.c_int, C.c_int, C.c_int)
    sig("orc_maximum_matching", C.c_int, vp, C.c_int, vp)
    sig("orc_c21_eval", C.c_float, C.c_int, C.c_double, C.c_int)
    sig("orc_key4", C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64)
    sig("orc_gen_roots", None, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp)
    sig("orc_hash_predictions", None, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_uint64, vp)
    sig("orc_create", vp, C.c_int, C.c_int, C.c_int)
    sig("orc_destroy", None, vp)
    sig("orc_set_path_kind", None, vp, C.c_int)
    sig("orc_set_layers", None, vp, C.c_int)
    sig("orc_new_begin", None, vp, vp, vp)
    sig("orc_new_end", None, vp, vp)
    sig("orc_rollout_begin", None, vp, vp, C.c_int, C.c_uint32)
    sig("orc_rollout_end", C.c_int, vp, vp)
    sig("orc_observe", None, vp, C.c_uint32, vp, vp)
    sig("orc_reset_begin", None, vp, vp, vp)
    sig("orc_reset_end", None, vp, vp)
    sig("orc_c21_modify_roots", None, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, vp, vp)
    sig("orc_state_vecs", f32p, vp)
    sig("orc_counters", None, vp, vp)
    sig("orc_argmin", None, vp, vp, vp, f64p, i32p, f32p)
    sig("orc_tree_sizes", None, vp, C.c_int, i32p, i32p, i32p)
    sig("orc_export_tree", None, vp, C.c_int, *([vp] * 13))
    sig("orc_agent_state", None, vp, C.c_int, vp, vp, vp, u32p, f64p, i32p)
    sig("orc_create_ramsey", vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int)
    sig("orc_gen_ramsey_roots", None, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
        vp, vp)
    sig("orc_argmin_totals", None, vp, vp)
    sig("orc_agent_totals", None, vp, C.c_int, vp)
    sig("orc_agent_counts", None, vp, C.c_int, vp)
    for name in ("state_dim", "action_dim", "key_words", "root_bytes"):
        sig("orc_engine_" + name, C.c_int, vp)
    sig("orc_ramsey_counts_new", None, C.c_int, C.c_int, vp, vp, vp, vp)
    sig("orc_ramsey_act_sequence", None, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp)
    sig("orc_create_dense", vp, C.c_int, C.c_int, C.c_int)
    sig("orc_set_dense_p", None, vp, C.c_uint32)
    sig("orc_gen_dense_roots", None, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, vp, vp)
    sig("orc_dense_matching_tutte", C.c_int, vp, C.c_int)
    sig("orc_dense_matching_exact", C.c_int, vp, C.c_int)
    sig("orc_dense_matching_reference", C.c_int, vp, C.c_int)
    sig("orc_dense_is_cut_edge", C.c_int, vp, C.c_int, C.c_int)
    sig("orc_dense_lambda1", C.c_double, vp, C.c_int)
    sig("orc_mlp_create", vp, C.c_int, vp, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
        C.c_uint64, C.c_int)
    sig("orc_mlp_destroy", None, vp)
    sig("orc_mlp_num_params", C.c_int64, vp)
    sig("orc_mlp_get_params", None, vp, vp)
    sig("orc_mlp_set_params", None, vp, vp)
    sig("orc_mlp_forward", None, vp, C.c_int, vp, vp)
    sig("orc_mlp_forward_fast", C.c_int, vp, C.c_int, vp, vp)
    sig("orc_mlp_update", C.c_float, vp, C.c_int, vp, vp, vp)
    _LIB = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def gen_roots(seed, epoch, first_agent, count, n, kmin, kmax):
    L = lib()
    kw = L.orc_key_words(n)
    parents = np.zeros((count, n), np.uint8)
    permitted = np.zeros((count, kw), np.uint64)
    L.orc_gen_roots(seed, epoch, first_agent, count, n, kmin, kmax, _p(parents), _p(permitted))
    return parents, permitted


def gen_ramsey_roots(seed, epoch, first_agent, count, n, n_colors, kmin, kmax):
    e = n * (n - 1) // 2
    kw = (e * n_colors + 63) // 64
    colors = np.zeros((count, e), np.uint8)
    permitted = np.zeros((count, kw), np.uint64)
    lib().orc_gen_ramsey_roots(seed, epoch, first_agent, count, n, n_colors, kmin, kmax, _p(colors), _p(permitted))
    return colors, permitted


def gen_dense_roots(seed, epoch, first_agent, count, n, kmin, kmax, p=0.2):
    """seeded roots of the dense-graph space: (adj u64 [count, n], modifiable slots u64 [count, KW])"""
    e = n * (n - 1) // 2
    kw = (2 * e + 63) // 64
    adj = np.zeros((count, n), np.uint64)
    slots = np.zeros((count, kw), np.uint64)
    lib().orc_gen_dense_roots(seed, epoch, first_agent, count, n, kmin, kmax, int(round(p * (1 << 24))), _p(adj), _p(slots))
    return adj, slots


def adjacency(n, edges):
    """neighbourhood bitsets (u64 per vertex) of an edge list"""
    adj = np.zeros(n, np.uint64)
    for u, v in edges:
        adj[u] |= np.uint64(1) << np.uint64(v)
        adj[v] |= np.uint64(1) << np.uint64(u)
    return adj


def dense_matching_tutte(adj):
    adj = np.ascontiguousarray(adj, np.uint64)
    return lib().orc_dense_matching_tutte(_p(adj), len(adj))


def dense_matching_exact(adj):
    adj = np.ascontiguousarray(adj, np.uint64)
    return lib().orc_dense_matching_exact(_p(adj), len(adj))


def dense_matching_reference(adj):
    adj = np.ascontiguousarray(adj, np.uint64)
    return lib().orc_dense_matching_reference(_p(adj), len(adj))


def dense_is_cut_edge(adj, v, u):
    adj = np.ascontiguousarray(adj, np.uint64)
    return bool(lib().orc_dense_is_cut_edge(_p(adj), v, u))


def dense_lambda1(adj):
    adj = np.ascontiguousarray(adj, np.uint64)
    return lib().orc_dense_lambda1(_p(adj), len(adj))


def ramsey_counts_new(n, sizes, colors):
    """RamseyCounts::new: (counts [C][E], totals [C]) of the colouring given per colex edge position"""
    sizes = np.ascontiguousarray(sizes, np.int32)
    colors = np.ascontiguousarray(colors, np.uint8)
    c, e = len(sizes), n * (n - 1) // 2
    counts = np.zeros((c, e), np.int32)
    totals = np.zeros(c, np.int32)
    lib().orc_ramsey_counts_new(n, c, _p(sizes), _p(colors), _p(counts), _p(totals))
    return counts, totals


def ramsey_act_sequence(n, sizes, colors, actions):
    """apply action ids in order with the incremental reassign_color; returns (colors, counts, totals)"""
    sizes = np.ascontiguousarray(sizes, np.int32)
    colors = np.array(colors, np.uint8)
    actions = np.ascontiguousarray(actions, np.int32)
    c, e = len(sizes), n * (n - 1) // 2
    counts = np.zeros((c, e), np.int32)
    totals = np.zeros(c, np.int32)
    lib().orc_ramsey_act_sequence(n, c, _p(sizes), _p(colors), _p(actions), len(actions), _p(counts), _p(totals))
    return colors, counts, totals


def hash_predictions(seed, first_agent, count, action_dim, call):
    out = np.zeros((count, action_dim), np.float32)
    lib().orc_hash_predictions(seed, first_agent, count, action_dim, call, _p(out))
    return out


class Tree:
    """Exported tree arrays (same field names as azdopt_amd's export)."""
    FIELDS = ("c", "c_star", "n_t", "exhausted", "act_begin", "act_end", "keys", "e_src", "e_dst", "e_pp",
              "p_aid", "p_g", "p_edge")

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def equal(self, other):
        for f in self.FIELDS:
            a, b = getattr(self, f), getattr(other, f)
            if a.shape != b.shape:
                return False, f"{f}: shape {a.shape} != {b.shape}"
            if a.dtype.kind == "f":
                same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
            else:
                same = np.array_equal(a.astype(np.int64), b.astype(np.int64))
            if not same:
                idx = np.argwhere(a != b)
                return False, f"{f}: first mismatch at {idx[0] if len(idx) else '?'}"
        return True, ""


class Engine:
    """NablaOptimizer-shaped driver of the oracle with an injectable model."""

    def __init__(self, n, batch, threads=1, ramsey=None, path_kind=0, layers=1, dense=False, dense_p=0.2):
        """ramsey = (sizes, weights) selects RamseySpaceNoEdgeRecolor<B32, n, E, C>; default the c21 space.
        path_kind: 0 ActionSet / ActionMultiset, 1 ActionSequence / OrderedActionSet"""
        self.L = lib()
        self.n, self.B = n, batch
        if dense:
            self.h = self.L.orc_create_dense(n, batch, threads)
            self.L.orc_set_dense_p(self.h, int(round(dense_p * (1 << 24))))  # fresh roots of the root policy
        elif ramsey is None:
            self.h = self.L.orc_create(n, batch, threads)
        else:
            sizes = np.ascontiguousarray(ramsey[0], np.int32)
            weights = np.ascontiguousarray(ramsey[1], np.float32)
            self.C = len(sizes)
            self.h = self.L.orc_create_ramsey(n, len(sizes), _p(sizes), _p(weights), batch, threads)
        assert self.h
        self.S, self.A, self.KW = (self.L.orc_engine_state_dim(self.h), self.L.orc_engine_action_dim(self.h),
                                   self.L.orc_engine_key_words(self.h))
        self.RB = self.L.orc_engine_root_bytes(self.h)
        self.L.orc_set_path_kind(self.h, path_kind)
        if layers > 1:  # Layered<L, Space>
            self.L.orc_set_layers(self.h, layers)
            self.S = self.L.orc_engine_state_dim(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def state_vecs(self):
        ptr = self.L.orc_state_vecs(self.h)
        return np.ctypeslib.as_array(ptr, shape=(self.B, self.S)).copy()

    def new_begin(self, parents, permitted):
        parents = np.ascontiguousarray(parents, np.uint8)
        permitted = np.ascontiguousarray(permitted, np.uint64)
        self.L.orc_new_begin(self.h, _p(parents), _p(permitted))

    def new_end(self, h):
        h = np.ascontiguousarray(h, np.float32)
        self.L.orc_new_end(self.h, _p(h))

    def rollout_begin(self, tol, tol_default):
        t = np.ascontiguousarray(tol, np.uint32)
        self.L.orc_rollout_begin(self.h, _p(t), len(t), tol_default)

    def rollout_end(self, h):
        h = np.ascontiguousarray(h, np.float32)
        return self.L.orc_rollout_end(self.h, _p(h))

    def observe(self, n_obs_tol):
        obs = np.zeros((self.B, self.A), np.float32)
        w = np.zeros((self.B, self.A), np.float32)
        self.L.orc_observe(self.h, n_obs_tol, _p(obs), _p(w))
        return obs, w

    def reset_begin(self, parents, permitted):
        parents = np.ascontiguousarray(parents, np.uint8)
        permitted = np.ascontiguousarray(permitted, np.uint64)
        self.L.orc_reset_begin(self.h, _p(parents), _p(permitted))

    def reset_end(self, h):
        h = np.ascontiguousarray(h, np.float32)
        self.L.orc_reset_end(self.h, _p(h))

    def modify_roots(self, seed, epoch, first_agent, kmin, kmax):
        parents = np.zeros((self.B, self.RB), np.uint8)
        permitted = np.zeros((self.B, self.KW), np.uint64)
        self.L.orc_c21_modify_roots(self.h, seed, epoch, first_agent, kmin, kmax, _p(parents), _p(permitted))
        return parents, permitted

    def counters(self):
        out = np.zeros(CTR_COUNT, np.uint64)
        self.L.orc_counters(self.h, _p(out))
        return {k: int(out[v]) for k, v in CTR.items()}

    def argmin_totals(self):
        t = np.zeros(4, np.int32)
        self.L.orc_argmin_totals(self.h, _p(t))
        return t

    def agent_totals(self, agent):
        t = np.zeros(4, np.int32)
        self.L.orc_agent_totals(self.h, agent, _p(t))
        return t

    def agent_counts(self, agent):
        c = np.zeros((self.C, self.RB), np.int32)
        self.L.orc_agent_counts(self.h, agent, _p(c))
        return c

    def argmin(self):
        parents = np.zeros(self.RB, np.uint8)
        permitted = np.zeros(self.KW, np.uint64)
        lam, mu, ev = C.c_double(), C.c_int32(), C.c_float()
        self.L.orc_argmin(self.h, _p(parents), _p(permitted), C.byref(lam), C.byref(mu), C.byref(ev))
        return dict(parents=parents, permitted=permitted, lambda1=lam.value, matching=mu.value, eval=np.float32(ev.value))

    def tree_sizes(self, agent):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self.L.orc_tree_sizes(self.h, agent, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def export_tree(self, agent):
        nn, ne, npred = self.tree_sizes(agent)
        t = Tree(c=np.zeros(nn, np.float32), c_star=np.zeros(nn, np.float32), n_t=np.zeros(nn, np.uint32),
                 exhausted=np.zeros(nn, np.uint32), act_begin=np.zeros(nn, np.uint32),
                 act_end=np.zeros(nn, np.uint32), keys=np.zeros((nn, self.KW), np.uint64),
                 e_src=np.zeros(ne, np.uint32), e_dst=np.zeros(ne, np.uint32), e_pp=np.zeros(ne, np.uint32),
                 p_aid=np.zeros(npred, np.uint32), p_g=np.zeros(npred, np.float32),
                 p_edge=np.zeros(npred, np.int32))
        self.L.orc_export_tree(self.h, agent, *[_p(getattr(t, f)) for f in Tree.FIELDS])
        return t

    def agent_state(self, agent):
        parents = np.zeros(self.RB, np.uint8)
        permitted = np.zeros(self.KW, np.uint64)
        path = np.zeros(self.KW, np.uint64)
        pos, lam, mu = C.c_uint32(), C.c_double(), C.c_int32()
        self.L.orc_agent_state(self.h, agent, _p(parents), _p(permitted), _p(path), C.byref(pos), C.byref(lam),
                               C.byref(mu))
        return dict(parents=parents, permitted=permitted, path=path, state_pos=pos.value, lambda1=lam.value,
                    matching=mu.value)


class Mlp:
    def __init__(self, dims, final_act=2, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, l2=1e-6, seed=0, threads=1):
        self.L = lib()
        self.dims = list(dims)
        d = np.asarray(dims, np.int32)
        self.h = self.L.orc_mlp_create(len(dims) - 1, _p(d), final_act, lr, beta1, beta2, eps, l2, seed, threads)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_mlp_destroy(self.h)
            self.h = None

    def num_params(self):
        return self.L.orc_mlp_num_params(self.h)

    def get_params(self):
        out = np.zeros(self.num_params(), np.float32)
        self.L.orc_mlp_get_params(self.h, _p(out))
        return out

    def set_params(self, p):
        p = np.ascontiguousarray(p, np.float32)
        assert p.size == self.num_params()
        self.L.orc_mlp_set_params(self.h, _p(p))

    def forward(self, states):
        states = np.ascontiguousarray(states, np.float32)
        b = states.shape[0]
        out = np.zeros((b, self.dims[-1]), np.float32)
        self.L.orc_mlp_forward(self.h, b, _p(states), _p(out))
        return out

    def forward_fast(self, states):
        """the same rows bit for bit from the blocked, vectorised loop nest (CPU-baseline timing)"""
        states = np.ascontiguousarray(states, np.float32)
        b = states.shape[0]
        out = np.zeros((b, self.dims[-1]), np.float32)
        self.vectorised = bool(self.L.orc_mlp_forward_fast(self.h, b, _p(states), _p(out)))
        return out

    def update(self, states, obs, w):
        states = np.ascontiguousarray(states, np.float32)
        obs = np.ascontiguousarray(obs, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        return self.L.orc_mlp_update(self.h, states.shape[0], _p(states), _p(obs), _p(w))
