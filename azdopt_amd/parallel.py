"""Multi-GPU plumbing: one process per GPU, agents sharded by global id.

During episodes there is NO data-path collective: trees never reference each other
(optimizer/mod.rs:159-189 zips disjoint per-agent slices) and every agent's random streams are keyed
by its GLOBAL id, so a shard reproduces exactly the trees it would have in a single-GPU run.
Once per epoch the training triple (state_vecs, observations, action_weights) is all-gathered
(RCCL over xGMI via torch.distributed backend "nccl"; gloo on CPU in the tests) because the loss
normaliser sum(w) is global over the batch (model/dfdx.rs:106,110); every rank then takes the
identical optimiser step, so the replicas stay in lock-step without a parameter broadcast.
The best-cost report is a MINLOC over ranks of 16 bytes."""


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
