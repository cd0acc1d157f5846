"""Multi-GPU use of the engine: one process per GPU, the batch dimension sharded across ranks.

Every (n, c) plane is transformed independently, so there is NO collective on
the data path and no halo exchange: each rank owns its slice of x, yl and every yh[j].  The only
collective is a one-off broadcast of the filter banks from rank 0 at start-up (tens of floats over
RCCL/xGMI: latency only), so that every rank provably filters with identical taps.
"""
import torch
import torch.distributed as dist


def broadcast_filter_banks(module, src=0, group=None):
    """Broadcast every filter buffer / frozen parameter of ``module`` from rank ``src``."""
    if not (dist.is_available() and dist.is_initialized()):
        return module
    with torch.no_grad():
        for t in list(module.buffers()) + list(module.parameters()):
            dist.broadcast(t, src=src, group=group)
    return module


def shard_bounds(n, world, rank):
    """[lo, hi) of the batch slice owned by ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(x, group=None):
    """This rank's slice of a batch that every rank holds in full (dim 0)."""
    if not (dist.is_available() and dist.is_initialized()):
        return x
    lo, hi = shard_bounds(x.shape[0], dist.get_world_size(group), dist.get_rank(group))
    return x[lo:hi]


def gather_batch(y, n_total, group=None):
    """all_gather of per-rank outputs back to a full batch (outside any timed region; only for
    callers that need the result on every rank)."""
    if not (dist.is_available() and dist.is_initialized()):
        return y
    world = dist.get_world_size(group)
    sizes = [hi - lo for lo, hi in (shard_bounds(n_total, world, r) for r in range(world))]
    cap = max(sizes)                       # all_gather wants equal shapes: pad the short shards
    mine = y.new_zeros((cap,) + tuple(y.shape[1:]))
    mine[:y.shape[0]] = y
    outs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(outs, mine, group=group)
    return torch.cat([o[:n] for o, n in zip(outs, sizes)], dim=0)
