"""Multi-GPU form of NablaOptimizer: one process per GPU, agents sharded by global id.

During episodes there is NO data-path collective: trees never reference each other
(optimizer/mod.rs:159-189 zips disjoint per-agent slices) and every agent's random streams are keyed
by its GLOBAL id, so a shard reproduces exactly the trees it would have in a single-GPU run.
Once per epoch (`par_update_model`, optimizer/mod.rs:249-281) the training triple (state_vecs,
observations, action_weights) is all-gathered (RCCL over xGMI via torch.distributed backend "nccl";
gloo on CPU in the tests) because the loss normaliser sum(w) is global over the batch
(model/dfdx.rs:106,110); every rank then takes the identical, deterministic optimiser step on the pooled
rows, so the model replicas stay bit-identical without a parameter broadcast.  The best-cost report is a
MINLOC over ranks of 16 bytes.

What differs from ONE optimizer over the whole population: `par_roll_out_episodes` returns the number of
calls that improved THIS RANK's argmin (each rank tracks the best of its own shard, as the per-call scan of
optimizer/mod.rs:194-246 would over its trees); the global best is `global_argmin()`.  Trees, state vectors,
training rows, loss and parameters are those of the single optimizer bit for bit
(tests/test_parallel_gloo.py, tests/test_gpu_parity.py::test_sharded_agents_are_shard_invariant).

A host that is not Python does the same with three calls of the C ABI: azd_engine_observe_dev ->
ncclAllGather x 3 -> azd_evaluator_update_model_dev, or the one call azd_engine_par_update_model_sharded
("Multi-GPU from C")."""


class ShardPlan:
    """Contiguous agent ranges: rank r owns global agents [r * per_rank, (r + 1) * per_rank)."""

    def __init__(self, world_size, rank, agents_per_rank):
        if not (0 <= rank < world_size) or agents_per_rank <= 0:
            raise ValueError("bad shard plan")
        self.world_size, self.rank = world_size, rank
        self.local_agents = agents_per_rank
        self.total_agents = agents_per_rank * world_size
        self.first_agent = rank * agents_per_rank

    def owner(self, global_agent):
        return global_agent // self.local_agents

    def local_index(self, global_agent):
        return global_agent - self.first_agent


def allgather_training_triple(dist, torch, local_tensors, world_size):
    """all-gather [B, d] tensors into [world*B, d] in rank order (= global agent order)."""
    out = []
    for t in local_tensors:
        g = torch.empty((world_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(g, t.contiguous())
        out.append(g)
    return out


def global_argmin(dist, torch, local_eval, local_cost, local_rank=0, device=None):
    """MINLOC over ranks of (eval, rank): returns (best_eval, best_cost).  Ties -> lowest rank,
    matching the lowest-agent-index rule inside a rank."""
    if dist is None:
        return float(local_eval), float(local_cost)
    if device is None:
        device = f"cuda:{local_rank}" if dist.get_backend() == "nccl" else "cpu"
    world = dist.get_world_size()
    mine = torch.tensor([float(local_eval), float(local_cost)], dtype=torch.float64, device=device)
    flat = torch.empty(world * 2, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(flat, mine)
    allv = flat.cpu().reshape(world, 2)
    best = min(range(world), key=lambda r: (float(allv[r, 0]), r))
    return float(allv[best, 0]), float(allv[best, 1])


class _DevView:
    """zero-copy torch view of an engine-owned device buffer (via __cuda_array_interface__)"""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr="<f4", data=(int(ptr), False), version=2)


class EngineShard:
    """The local half of a ShardedOptimizer on a GPU: a NablaOptimizer over this rank's agents and an
    ActionModel replica sized for the POOLED batch.  `stage_on_cpu` routes the collective through host
    tensors (gloo rehearsal of the N > 1 host logic on a one-GPU box; RCCL never needs it)."""

    def __init__(self, opt, model, torch, device_index=0, stage_on_cpu=False):
        self.opt, self.model, self.torch = opt, model, torch
        self.device = f"cuda:{device_index}"
        self.stage_on_cpu = stage_on_cpu
        self.coll_device = "cpu" if stage_on_cpu else self.device

    def roll_out(self, n_as_tol, n_calls):
        return self.opt.par_roll_out_episodes(n_as_tol, n_calls=n_calls)

    def triple(self, n_obs_tol):
        """the training triple of this shard as tensors the collective may read (copies: the all-gather runs on
        torch-owned memory, not on the engine's own allocations)"""
        ptrs = self.opt.observe_dev(n_obs_tol)
        sp = self.opt.space
        B = self.opt.batch
        views = [self.torch.as_tensor(_DevView(p, (B, d)), device=self.device)
                 for p, d in zip(ptrs, (sp.STATE_DIM, sp.ACTION_DIM, sp.ACTION_DIM))]
        return [v.cpu() if self.stage_on_cpu else v.clone() for v in views]

    def update_local(self, n_obs_tol):
        return self.opt.par_update_model(n_obs_tol)

    def update_pooled(self, rows, pooled):
        pooled = [g.to(self.device) for g in pooled] if self.stage_on_cpu else pooled
        self.torch.cuda.synchronize()
        return self.model.update_model_dev(rows, *[g.data_ptr() for g in pooled], stream=self.opt.stream())

    def reset_policy(self, seed, epoch, kmin=None, kmax=None):
        self.opt.par_reset_trees_policy(seed, epoch, kmin, kmax)

    def reset(self, roots):
        self.opt.par_reset_trees(roots)

    def argmin(self):
        am = self.opt.argmin_data()
        cost = float(sum(am.cost["clique_counts"])) if "clique_counts" in am.cost else am.cost["lambda_1"] + len(am.cost["matching"])
        return float(am.eval), cost, am

    def expansions(self):
        return self.opt.counters()["EXPANSIONS"]

    def params_bytes(self):
        return self.model.get_params().tobytes()


class ShardedOptimizer:
    """NablaOptimizer<Space, M, P> (optimizer/mod.rs) over a population sharded across the ranks of a
    torch.distributed process group: same method names and meaning; `par_update_model` pools the training rows
    of all ranks before the optimiser step.  `shard` is the local half (EngineShard on a GPU; the tests drive
    the same class with the CPU oracle behind the same five methods).  dist = None or world size 1: a plain
    single-GPU optimizer, no collective."""

    def __init__(self, shard, plan, dist=None, torch=None, coll_device=None):
        self.shard, self.plan = shard, plan
        self.dist = dist if (dist is not None and plan.world_size > 1) else None
        self.torch = torch
        self.coll_device = coll_device if coll_device is not None else getattr(shard, "coll_device", "cpu")
        if self.dist is not None and self.dist.get_world_size() != plan.world_size:
            raise ValueError("shard plan and process group disagree on the world size")

    @classmethod
    def par_new(cls, space, model_factory, agents_per_rank, dist=None, torch=None, seed=0, device_index=0, rank=None,
                world_size=None, stage_on_cpu=False, **engine_kw):
        """optimizer/mod.rs:39-118 on every rank: `model_factory(total_agents)` builds this rank's replica (same
        seed on every rank = same parameters), roots come from the seeded `init_states` keyed by global agent id."""
        from .optimizer import NablaOptimizer
        if dist is not None and dist.is_initialized():
            rank = dist.get_rank() if rank is None else rank
            world_size = dist.get_world_size() if world_size is None else world_size
        rank, world_size = rank or 0, world_size or 1
        plan = ShardPlan(world_size, rank, agents_per_rank)
        model = model_factory(plan.total_agents)
        roots = space.generate_roots(seed, plan.local_agents, first_agent=plan.first_agent)
        opt = NablaOptimizer.par_new(space, roots, model, plan.local_agents, device=device_index,
                                     first_agent=plan.first_agent, **engine_kw)
        return cls(EngineShard(opt, model, torch, device_index, stage_on_cpu), plan, dist, torch)

    # ---- the reference's method names
    def par_roll_out_episodes(self, n_as_tol, n_calls=1):
        """optimizer/mod.rs:121-191 on this rank's agents; no collective.  Returns the calls that improved this
        rank's argmin."""
        return self.shard.roll_out(n_as_tol, n_calls)

    def par_update_model(self, n_obs_tol):
        """optimizer/mod.rs:249-281 over the pooled rows of all ranks, identical on every rank.  `native_comm` set
        (use_native_exchange): the one C-ABI call azd_engine_par_update_model_sharded does observe -> ncclAllGather x 3 ->
        optimiser step on the engine's own stream instead of torch.distributed."""
        if self.dist is None:
            return self.shard.update_local(n_obs_tol)
        import time
        if getattr(self, "native_comm", None) is not None:
            t0 = time.perf_counter()
            loss = self.shard.opt.par_update_model_sharded(n_obs_tol, self.native_comm)
            self.exchange_ms = getattr(self, "exchange_ms", 0.0) + 1e3 * (time.perf_counter() - t0)  # (all-gather + step: one call)
            self.exchanges = getattr(self, "exchanges", 0) + 1
            return loss
        local = self.shard.triple(n_obs_tol)
        sync = getattr(self.torch.cuda, "synchronize", None) if str(self.coll_device).startswith("cuda") else None
        if sync:
            sync()
        t0 = time.perf_counter()
        pooled = allgather_training_triple(self.dist, self.torch, local, self.plan.world_size)
        if sync:
            sync()
        self.exchange_ms = getattr(self, "exchange_ms", 0.0) + 1e3 * (time.perf_counter() - t0)  # the all-gather alone
        self.exchanges = getattr(self, "exchanges", 0) + 1
        return self.shard.update_pooled(self.plan.total_agents, pooled)

    def use_native_exchange(self, nccl_comm):
        """epoch exchange through azd_engine_par_update_model_sharded with this ncclComm_t (raw handle) from now on"""
        self.native_comm = nccl_comm

    def replicas_identical(self):
        """every rank took the identical optimiser step: all-gather of a 64-bit hash of this rank's parameters"""
        import hashlib
        h = int.from_bytes(hashlib.blake2b(self.shard.params_bytes(), digest_size=8).digest(), "little") >> 1
        if self.dist is None:
            return True
        mine = self.torch.tensor([h], dtype=self.torch.int64, device=self.coll_device)
        allv = self.torch.empty(self.plan.world_size, dtype=self.torch.int64, device=self.coll_device)
        self.dist.all_gather_into_tensor(allv, mine)
        return bool((allv == allv[0]).all().item())

    def par_reset_trees_policy(self, seed, epoch, kmin=None, kmax=None):
        """optimizer/mod.rs:284-360 with the drivers' modify_root policy on the device; per rank, no collective"""
        self.shard.reset_policy(seed, epoch, kmin, kmax)

    def par_reset_trees(self, roots):
        self.shard.reset(roots)

    def argmin_data(self):
        """this rank's ArgminData (optimizer/mod.rs:361)"""
        return self.shard.argmin()[2]

    def global_argmin(self):
        """(best eval, its cost) over all ranks: 16-byte MINLOC, ties to the lowest rank"""
        ev, cost, _ = self.shard.argmin()
        return global_argmin(self.dist, self.torch, ev, cost, device=self.coll_device)

    def total_expansions(self):
        n = float(self.shard.expansions())
        if self.dist is None:
            return n
        t = self.torch.tensor([n], dtype=self.torch.float64, device=self.coll_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t[0])
