"""Sharding of the hot path over the GPUs of one node (one process per GPU, torch.distributed over RCCL).

* Table interpolation (interp1/interp2): the query axis splits into contiguous per-rank shards, the table is
  replicated; no collective is needed unless the caller wants the whole result vector on every rank
  (all_gather_results).
* EventDrivenMap::ComputeF: realisations split into contiguous per-rank shards (the realisation is already
  the unit of work, EventDrivenMap.cu:196); the only exchange is an all-reduce(sum) of the partial block
  [S fp64 sums | accepted count | S restricted positions of realisation 0] (MI_EDM_PARTIAL_LEN(S) = 2S+1
  scalars) -- not an all-gather.
"""
import numpy as np


def shard_bounds(n, rank, world):
    """Contiguous, balanced split of n units: shard r = [lo, hi); the first n % world shards get one extra."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_results(local, world_sizes=None):
    """Reassemble per-rank result shards (torch tensors on this rank's device) on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    if world_sizes is None:
        out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local)
        return out
    # ragged shards: pad every shard to the largest one (collectives want equal sizes), trim afterwards
    kmax = int(max(world_sizes))
    padded = torch.zeros(kmax, dtype=local.dtype, device=local.device)
    padded[:local.numel()] = local
    out = torch.empty(world * kmax, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * kmax:r * kmax + int(k)] for r, k in enumerate(world_sizes)])


class ShardedResidual:
    """ComputeF over realisations sharded across ranks.

    local_partial(Z, lo, hi) -> float64[2S+1]: this rank's partial block -- un-normalised per-spike sums over its
    accepted realisations [lo, hi), its accepted count, and (rank with lo == 0, reference averaging) the restricted
    position of realisation 0, which the reference leaves out of the sum unless the count is 1.  On the GPU that is
    EventDrivenMap(n_real=hi-lo, real_offset=lo).ComputeF(Z, want_partial=True)[1]; tests inject the oracle.
    finish(Z, sums_and_count) -> f: mi_edm_residual_from_sums (host arithmetic of EventDrivenMap.cu:237-239).
    """

    def __init__(self, n_real_total, local_partial, finish, rank=None, world=None, device="cpu"):
        import torch.distributed as dist
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        self.lo, self.hi = shard_bounds(n_real_total, self.rank, self.world)
        self._local, self._finish, self._device = local_partial, finish, device

    def ComputeF(self, Z):
        import torch
        import torch.distributed as dist
        part = np.asarray(self._local(Z, self.lo, self.hi), dtype=np.float64)
        t = torch.from_numpy(part.copy()).to(self._device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)           # 2S+1 scalars: latency-bound, tens of us on xGMI
        return self._finish(Z, t.cpu().numpy())
