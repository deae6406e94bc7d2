"""NablaOptimizer: the batch driver (az-discrete-opt/src/nabla/optimizer/mod.rs), device resident."""
import ctypes as C

import numpy as np

from . import _lib
from .space import ActionSet


class ArgminData:
    """az-discrete-opt/src/log.rs:1-11 for the c21 space"""

    def __init__(self, rec, n, kw):
        self.state = dict(parents=np.array(rec.parents[:n], np.uint8), permitted=np.array(rec.permitted[:kw], np.uint64))
        m = rec.matching_size
        self.cost = dict(lambda_1=rec.lambda_1, matching=[(rec.matching[2 * i], rec.matching[2 * i + 1]) for i in range(m)])
        self.eval = np.float32(rec.eval)
        self.agent, self.node = rec.agent, rec.node


class RamseyArgminData:
    """az-discrete-opt/src/log.rs:1-11 for the Ramsey space: state = colouring + permitted edges,
    cost = TotalCounts"""

    def __init__(self, rec, space):
        self.state = dict(colors=np.array(rec.colors[:space.E], np.uint8), permitted=np.array(rec.permitted[:], np.uint64))
        self.cost = dict(clique_counts=[int(x) for x in rec.totals[:space.C]])
        self.eval = np.float32(rec.eval)
        self.agent, self.node = rec.agent, rec.node


class DenseArgminData:
    """az-discrete-opt/src/log.rs:1-11 for the dense-graph space: state = neighbourhoods + open slots,
    cost = Conjecture2Dot1Cost {lambda_1, matching number}"""

    def __init__(self, rec, space):
        slots = np.zeros(space.KEY_WORDS, np.uint64)  # the open slots as a bitmap over the E edge positions, in the host's key width
        ow = (space.E + 63) // 64
        slots[:ow] = rec.permitted[:ow]
        self.state = dict(adj=np.array(rec.adj[:space.n], np.uint64), permitted=slots)
        self.cost = dict(lambda_1=rec.lambda_1, matching=[None] * rec.matching_size)  # the matching itself is not constructed
        self.eval = np.float32(rec.eval)
        self.agent, self.node = rec.agent, rec.node


class TreeView:
    """Raw arrays of one SearchTree (tree/mod.rs:28-32): nodes, arcs, predictions, keys."""
    FIELDS = ("c", "c_star", "n_t", "exhausted", "act_begin", "act_end", "keys", "e_src", "e_dst", "e_pp",
              "p_aid", "p_g", "p_edge")

    def __init__(self, **kw):
        self.__dict__.update(kw)


MAX_NODE_CAPACITY, MAX_ARC_CAPACITY, MAX_PREDICTION_CAPACITY = 65536, 65535, 1 << 20  # include/azdopt_amd.h: AZD_MAX_*_CAPACITY


def tree_capacities(episodes, max_actions_per_node):
    """Arena capacities for an epoch of `episodes` calls per tree (a call adds one node, a handful of arcs -- the new arc plus the
    transpositions on the way -- and one node's predictions), as keyword arguments of `NablaOptimizer.par_new`.  The packed
    records limit a tree to 65536 nodes, 65535 arcs and 2^20 predictions (the reference's u32 indices do not): an epoch beyond them
    is refused HERE, naming the argument, rather than by azd_engine_create."""
    node = max(4096, 2 * episodes + 64)  # one expansion per call, plus the terminal nodes met on the way
    arc = max(8192, 8 * episodes + 64)   # the new arc plus the transpositions of the re-descents (r44's dense DAGs: 5-6 per call)
    pred = max(32768, (episodes + 1) * max_actions_per_node + 128)
    for name, want, limit in (("node_capacity", node, MAX_NODE_CAPACITY), ("prediction_capacity", pred, MAX_PREDICTION_CAPACITY)):
        if want > limit:
            raise ValueError(f"{episodes} episodes per epoch need {name} = {want}, beyond the record format's {limit}: "
                             f"use at most {(limit - 64) // 2 if name == 'node_capacity' else (limit - 128) // max_actions_per_node - 1} episodes per epoch")
    # arcs: 8 per call is a generous estimate, not a need; an epoch that does run out stops its agent with AZD_ERR_CAPACITY
    return dict(node_capacity=node, arc_capacity=min(arc, MAX_ARC_CAPACITY), prediction_capacity=pred)


class NablaOptimizer:
    """NablaOptimizer<Space, M, P>; P = ActionSet (default), ActionMultiset, ActionSequence, OrderedActionSet.

    par_new / par_roll_out_episodes / par_update_model / par_reset_trees / argmin_data / get_trees
    keep the reference's names and meaning (optimizer/mod.rs:30-36,39,121,249,284,361).  The
    `init_states` and `modify_root` closures stay on the host and hand over packed roots."""

    def __init__(self, space, model, batch, device=0, first_agent=0, node_capacity=0, arc_capacity=0,
                 prediction_capacity=0, path=ActionSet, persistent=True, async_step=True, pool_step=None):
        """step form: pool_step=None lets the engine choose (pool step from 256 agents, else the asynchronous one),
        True / False force the pool / asynchronous step, async_step=False the lock-step one, persistent=False one
        launch per phase"""
        if not hasattr(path, "PATH_KIND") or not path.licensed_for(space):
            raise TypeError("path encoding %r is not licensed for this space (space/axioms.rs:12-19)" % (path,))
        self.space, self.model, self.batch, self.first_agent = space, model, batch, first_agent
        self._L = _lib.lib()
        cfg = _lib.EngineConfig(space.SPACE_ID, space.n, batch, device, node_capacity, arc_capacity,
                                prediction_capacity, first_agent,
                                (0 if persistent else _lib.ENGINE_NO_PERSISTENT_STEP) | (0 if async_step else _lib.ENGINE_BARRIER_STEP)
                                | (_lib.ENGINE_POOL_STEP if pool_step else 0)
                                | (_lib.ENGINE_ASYNC_STEP if (pool_step is False and async_step) else 0))
        cfg.path_kind = path.PATH_KIND
        cfg.layers = getattr(space, "layers", 1)
        if space.SPACE_ID == _lib.SPACE_DENSE:
            cfg.max_slots = space.MAX_SLOTS
            cfg.dense_p = space.p
        if space.SPACE_ID == _lib.SPACE_RAMSEY:
            cfg.n_colors = space.C
            for i in range(space.C):
                cfg.clique_sizes[i] = space.sizes[i]
                cfg.color_weights[i] = space.weights[i]
        self._h = C.c_void_p()
        ev = model._h if model is not None else None
        _lib.check(self._L.azd_engine_create(C.byref(self._h), C.byref(cfg), ev), "azd_engine_create")

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.azd_engine_destroy(self._h)
            self._h = C.c_void_p()

    @classmethod
    def par_new(cls, space, init_states, model, batch, **kw):
        """optimizer/mod.rs:39-118.  `init_states` is either packed roots (parents, permitted)
        or a callable(count, first_agent) returning them."""
        opt = cls(space, model, batch, **kw)
        roots = init_states(batch, opt.first_agent) if callable(init_states) else init_states
        parents, permitted = opt._roots(*roots)
        _lib.check(opt._L.azd_engine_par_new(opt._h, _lib.ptr(parents), _lib.ptr(permitted)), "par_new")
        return opt

    def _roots(self, parents, permitted):
        parents = np.ascontiguousarray(parents, np.uint8).reshape(self.batch, self.space.ROOT_BYTES)
        permitted = np.ascontiguousarray(permitted, np.uint64).reshape(self.batch, self.space.KEY_WORDS)
        return parents, permitted

    @staticmethod
    def _tol(n_as_tol):
        table, default = n_as_tol
        return np.ascontiguousarray(table, np.uint32), int(default)

    def par_roll_out_episodes(self, n_as_tol, n_calls=1):
        """optimizer/mod.rs:121-191.  n_as_tol = (table, default) (04-c21-tree.rs:136-138).
        Returns the number of calls that improved the argmin (0 = ArgminImprovement::Unchanged)."""
        t, d = self._tol(n_as_tol)
        imp = C.c_int32()
        _lib.check(self._L.azd_engine_par_roll_out_episodes(self._h, _lib.ptr(t), len(t), d, n_calls, C.byref(imp)),
                   "par_roll_out_episodes")
        return imp.value

    def run_ahead(self, n_as_tol, n_calls):
        """Start the next n_calls calls of par_roll_out_episodes(n_as_tol, ...) now, in one launch; the calls that ask for them --
        one at a time as in 04-c21-tree.rs:142-150, or in chunks -- launch nothing and are answered as the kernel completes
        them (azd_engine_run_ahead).  Returns False when this engine's step form cannot do that: the calls then run when asked for."""
        t, d = self._tol(n_as_tol)
        ok = C.c_int32()
        _lib.check(self._L.azd_engine_run_ahead(self._h, _lib.ptr(t), len(t), d, n_calls, C.byref(ok)), "run_ahead")
        return bool(ok.value)

    def par_update_model(self, n_obs_tol):
        """optimizer/mod.rs:249-281"""
        loss = C.c_float()
        _lib.check(self._L.azd_engine_par_update_model(self._h, n_obs_tol, C.byref(loss)), "par_update_model")
        return loss.value

    def par_update_model_sharded(self, n_obs_tol, nccl_comm):
        """par_update_model over the pooled rows of all ranks of an ncclComm_t (raw handle, e.g. from a C host or
        ctypes): all-gather of the training triple on the engine's stream + the identical optimiser step"""
        loss = C.c_float()
        _lib.check(self._L.azd_engine_par_update_model_sharded(self._h, n_obs_tol, nccl_comm, C.byref(loss)),
                   "par_update_model_sharded")
        return loss.value

    def par_reset_trees(self, modify_root):
        """optimizer/mod.rs:284-360.  `modify_root` is packed new roots or a callable(optimizer)
        returning them (e.g. lambda o: o.c21_modify_roots(seed, epoch))."""
        roots = modify_root(self) if callable(modify_root) else modify_root
        parents, permitted = self._roots(*roots)
        _lib.check(self._L.azd_engine_par_reset_trees(self._h, _lib.ptr(parents), _lib.ptr(permitted)), "par_reset_trees")

    def c21_modify_roots(self, seed, epoch, kmin=None, kmax=None, device=True):
        """The driver's modify_root policy (04-c21-tree.rs:172-206), seeded; evaluated by the device
        kernel (default) or by the host C++ restatement on exported node data."""
        lo, hi = self.space.default_permitted_range()
        kmin = lo if kmin is None else kmin
        kmax = hi if kmax is None else kmax
        parents = np.zeros((self.batch, self.space.ROOT_BYTES), np.uint8)
        permitted = np.zeros((self.batch, self.space.KEY_WORDS), np.uint64)
        fn = self._L.azd_engine_modify_roots_dev if device else self._L.azd_c21_modify_roots
        _lib.check(fn(self._h, seed, epoch, kmin, kmax, _lib.ptr(parents), _lib.ptr(permitted)), "azd_c21_modify_roots")
        return parents, permitted

    def par_reset_trees_policy(self, seed, epoch, kmin=None, kmax=None):
        """par_reset_trees with the drivers' modify_root policy (04-c21-tree.rs:172-206, 02-r44.rs:196-228)
        evaluated on the device: no host round trip at the epoch boundary."""
        lo, hi = self.space.default_permitted_range()
        kmin = lo if kmin is None else kmin
        kmax = hi if kmax is None else kmax
        _lib.check(self._L.azd_engine_par_reset_trees_policy(self._h, seed, epoch, kmin, kmax), "par_reset_trees_policy")

    par_reset_trees_c21 = par_reset_trees_policy
    modify_roots = c21_modify_roots

    def argmin_data(self):
        """optimizer/mod.rs:361"""
        if self.space.SPACE_ID == _lib.SPACE_RAMSEY:
            rec = _lib.RamseyArgmin()
            _lib.check(self._L.azd_engine_ramsey_argmin_data(self._h, C.byref(rec)), "ramsey_argmin_data")
            return RamseyArgminData(rec, self.space)
        if self.space.SPACE_ID == _lib.SPACE_DENSE:
            rec = _lib.DenseArgmin()
            _lib.check(self._L.azd_engine_dense_argmin_data(self._h, C.byref(rec)), "dense_argmin_data")
            return DenseArgminData(rec, self.space)
        rec = _lib.Argmin()
        _lib.check(self._L.azd_engine_argmin_data(self._h, C.byref(rec)), "argmin_data")
        return ArgminData(rec, self.space.n, self.space.KEY_WORDS)

    def get_model_mut(self):
        return self.model

    # ---- split-phase forms for an external NablaModel (cut at the model call)
    def par_new_begin(self, parents, permitted):
        parents, permitted = self._roots(parents, permitted)
        _lib.check(self._L.azd_engine_par_new_begin(self._h, _lib.ptr(parents), _lib.ptr(permitted)), "par_new_begin")

    def par_new_end(self, h):
        h = np.ascontiguousarray(h, np.float32)
        _lib.check(self._L.azd_engine_par_new_end(self._h, _lib.ptr(h)), "par_new_end")

    def roll_out_begin(self, n_as_tol):
        t, d = self._tol(n_as_tol)
        _lib.check(self._L.azd_engine_roll_out_begin(self._h, _lib.ptr(t), len(t), d), "roll_out_begin")

    def roll_out_end(self, h):
        h = np.ascontiguousarray(h, np.float32)
        imp = C.c_int32()
        _lib.check(self._L.azd_engine_roll_out_end(self._h, _lib.ptr(h), C.byref(imp)), "roll_out_end")
        return imp.value

    def reset_begin(self, parents, permitted):
        parents, permitted = self._roots(parents, permitted)
        _lib.check(self._L.azd_engine_reset_begin(self._h, _lib.ptr(parents), _lib.ptr(permitted)), "reset_begin")

    def reset_end(self, h):
        h = np.ascontiguousarray(h, np.float32)
        _lib.check(self._L.azd_engine_reset_end(self._h, _lib.ptr(h)), "reset_end")

    def observe(self, n_obs_tol):
        """par_update_model without the model call: (state_vecs, observations, action_weights)"""
        sv = np.zeros((self.batch, self.space.STATE_DIM), np.float32)
        obs = np.zeros((self.batch, self.space.ACTION_DIM), np.float32)
        w = np.zeros((self.batch, self.space.ACTION_DIM), np.float32)
        _lib.check(self._L.azd_engine_observe(self._h, n_obs_tol, _lib.ptr(sv), _lib.ptr(obs), _lib.ptr(w)), "observe")
        return sv, obs, w

    def observe_dev(self, n_obs_tol):
        """device pointers of the training triple (for an all-gather across GPUs)"""
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(self._L.azd_engine_observe_dev(self._h, n_obs_tol, C.byref(a), C.byref(b), C.byref(c)), "observe_dev")
        return a.value, b.value, c.value

    def state_vecs(self):
        sv = np.zeros((self.batch, self.space.STATE_DIM), np.float32)
        _lib.check(self._L.azd_engine_read_state_vecs(self._h, _lib.ptr(sv)), "read_state_vecs")
        return sv

    def predictions(self):
        h = np.zeros((self.batch, self.space.ACTION_DIM), np.float32)
        _lib.check(self._L.azd_engine_read_predictions(self._h, _lib.ptr(h)), "read_predictions")
        return h

    def debug_tile_forward(self, states):
        """Test entry: prediction rows exactly as the CU-resident step forms' in-kernel evaluator computes them, for the given
        state vectors [rows, STATE_DIM] (a checker's: tests/test_gpu_pool.py follows a launch of the product kernel with them)."""
        st = np.ascontiguousarray(states, np.float32).reshape(-1, self.space.STATE_DIM)
        out = np.zeros((st.shape[0], self.space.ACTION_DIM), np.float32)
        _lib.check(self._L.azd_engine_debug_tile_forward(self._h, _lib.ptr(st), _lib.ptr(out), st.shape[0]), "debug_tile_forward")
        return out

    # ---- introspection
    def tree_sizes(self, agent):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self._L.azd_engine_tree_sizes(self._h, agent, C.byref(a), C.byref(b), C.byref(c)), "tree_sizes")
        return a.value, b.value, c.value

    def get_tree(self, agent):
        nn, ne, npred = self.tree_sizes(agent)
        kw = self.space.KEY_WORDS
        t = TreeView(c=np.zeros(nn, np.float32), c_star=np.zeros(nn, np.float32), n_t=np.zeros(nn, np.uint32),
                     exhausted=np.zeros(nn, np.uint32), act_begin=np.zeros(nn, np.uint32),
                     act_end=np.zeros(nn, np.uint32), keys=np.zeros((nn, kw), np.uint64),
                     e_src=np.zeros(ne, np.uint32), e_dst=np.zeros(ne, np.uint32), e_pp=np.zeros(ne, np.uint32),
                     p_aid=np.zeros(npred, np.uint32), p_g=np.zeros(npred, np.float32), p_edge=np.zeros(npred, np.int32))
        _lib.check(self._L.azd_engine_export_tree(self._h, agent, *[_lib.ptr(getattr(t, f)) for f in TreeView.FIELDS]),
                   "export_tree")
        return t

    def get_trees(self):
        """optimizer/mod.rs:34-36"""
        return [self.get_tree(i) for i in range(self.batch)]

    def agent_state(self, agent):
        parents = np.zeros(self.space.ROOT_BYTES, np.uint8)
        permitted = np.zeros(self.space.KEY_WORDS, np.uint64)
        path = np.zeros(self.space.KEY_WORDS, np.uint64)
        pos, lam, mu = C.c_uint32(), C.c_double(), C.c_int32()
        _lib.check(self._L.azd_engine_agent_state(self._h, agent, _lib.ptr(parents), _lib.ptr(permitted), _lib.ptr(path),
                                                  C.byref(pos), C.byref(lam), C.byref(mu)), "agent_state")
        return dict(parents=parents, permitted=permitted, path=path, state_pos=pos.value, lambda1=lam.value, matching=mu.value)

    def ramsey_agent_counts(self, agent):
        """live (counts [C, E], totals [C]) of an agent's RamseyCounts state"""
        counts = np.zeros((self.space.C, self.space.E), np.int32)
        totals = np.zeros(4, np.int32)
        _lib.check(self._L.azd_engine_ramsey_agent_counts(self._h, agent, _lib.ptr(counts), _lib.ptr(totals)), "ramsey_agent_counts")
        return counts, totals[:self.space.C]

    def counters(self):
        out = np.zeros(_lib.CTR_COUNT, np.uint64)
        _lib.check(self._L.azd_engine_counters(self._h, _lib.ptr(out)), "counters")
        return {k: int(out[v]) for k, v in _lib.CTR.items()}

    def agent_counters(self, allow_wave_blocks=False):
        """per-agent counters, unreduced: dict name -> uint64 array [batch].  After a pool launch of the product build the blocks
        hold what searcher WAVES counted (only sums / maxima mean anything): refused then unless `allow_wave_blocks` -- per-agent
        attribution under the pool step is the diagnostic build's (make PROFILE=1), or run with pool_step=False."""
        if not allow_wave_blocks and self._L.azd_engine_agent_counters_per_agent(self._h) == 0:
            raise RuntimeError("agent_counters: the pool step of the product build attributes counters to searcher waves, not agents "
                               "(use counters() for the sums, the PROFILE build or pool_step=False for per-agent values)")
        out = np.zeros((self.batch, _lib.CTR_COUNT), np.uint64)
        _lib.check(self._L.azd_engine_agent_counters(self._h, _lib.ptr(out)), "agent_counters")
        return {k: out[:, v].copy() for k, v in _lib.CTR.items()}

    def pool_groups(self):
        """evaluator groups of the last pool launch: (members per group, groups, waves per slot); (0, 0, 0) = the classic form"""
        g, n, w = C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._L.azd_engine_pool_groups(self._h, C.byref(g), C.byref(n), C.byref(w)), "pool_groups")
        return g.value, n.value, w.value

    def pool_agent_finish_ms(self):
        """pool step: when each agent was through with the last launch's calls, ms from the launch's start (numpy [batch])"""
        out = np.zeros(self.batch, np.uint64)
        _lib.check(self._L.azd_engine_pool_agent_finish(self._h, _lib.ptr(out)), "pool_agent_finish")
        return out.astype(np.float64) * 1e-5

    def set_timing(self, enabled=True):
        _lib.check(self._L.azd_engine_set_timing(self._h, int(enabled)), "set_timing")

    def timing(self):
        a, b, n = C.c_double(), C.c_double(), C.c_uint64()
        _lib.check(self._L.azd_engine_timing(self._h, C.byref(a), C.byref(b), C.byref(n)), "timing")
        return dict(rollout_ms=a.value, evaluator_ms=b.value, rollout_launches=n.value)

    def stream(self):
        return self._L.azd_engine_stream(self._h)

    STEP_FORMS = {0: "none", 1: "async", 2: "barrier", 3: "per_call", 4: "pool", 5: "per_call_graph"}

    def pool_split(self):
        """(evaluator, searcher) workgroups of the last pool-step launch"""
        a, b = C.c_int32(), C.c_int32()
        _lib.check(self._L.azd_engine_pool_split(self._h, C.byref(a), C.byref(b)), "pool_split")
        return a.value, b.value

    def pool_utilisation(self):
        """(evaluator, searcher) busy shares of the last pool-step launch"""
        a, b = C.c_double(), C.c_double()
        _lib.check(self._L.azd_engine_pool_utilisation(self._h, C.byref(a), C.byref(b)), "pool_utilisation")
        return a.value, b.value

    def step_form(self):
        """(form, reason) of the last par_roll_out_episodes: "async" / "barrier" (CU-resident) or "per_call"
        (one launch per phase), and why a faster form was not taken"""
        form, why = C.c_int32(), C.c_char_p()
        _lib.check(self._L.azd_engine_step_form(self._h, C.byref(form), C.byref(why)), "step_form")
        return self.STEP_FORMS.get(form.value, "?"), (why.value or b"").decode()
