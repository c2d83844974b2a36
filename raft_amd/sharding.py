"""Batch sharding of frame pairs across GPUs.

The correlation path has no exchange step: every frame pair's volume, pyramid and lookups are
independent, so multi-GPU operation is one process per GPU working on its own contiguous slice
of the pairs.  The only communication is bookkeeping (barrier / max of the elapsed time), done
through ``torch.distributed`` (backend ``nccl`` == RCCL on ROCm; ``gloo`` in the CPU tests).
"""
import torch


def shard_range(total, rank, world):
    """Contiguous slice [lo, hi) of ``total`` frame pairs owned by ``rank``; slices differ in
    size by at most one and cover 0..total exactly."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_over_ranks(value, device="cpu"):
    """MAX of a python float over all ranks (identity when not initialised)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
