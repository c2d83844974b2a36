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


def init_ranks(backend, device=None):
    """``init_process_group`` the way bench.py's main() does it for N > 1 ranks (RANK / WORLD_SIZE / MASTER_* from
    the environment, as torch.distributed.run sets them): GPU ranks bind the group to their device (``device_id=``:
    RCCL communicators are created eagerly on that GPU), CPU ranks (gloo: the dry mode and the CPU tests) do not."""
    import torch.distributed as dist
    if device is not None and torch.device(device).type == "cuda":
        dist.init_process_group(backend, device_id=torch.device(device))
    else:
        dist.init_process_group(backend)
    return dist


class GradBucket:
    """The one collective of the path's training configuration (BASELINE configs[3]): the all-reduce of
    a DDP-sized gradient bucket (RAFT-basic: 5,257,536 fp32 = 21 MB), standing in for the model around the correlation
    block (the reference replicates with nn.DataParallel, train.py:342).  The bucket is a DUMMY (zeros, not
    gradients): the benchmark times the correlation path, the bucket only puts the collective's traffic on the links.

    ``start()`` issues the all-reduce asynchronously (``async_op=True``: on RCCL it runs on the communicator's own
    stream, next to the backward kernels -- what DDP's bucket overlap amounts to), ``finish()`` waits for it (GPU: the
    current stream waits, the host does not).  Without an initialised process group both are no-ops."""

    def __init__(self, numel, device, dtype=torch.float32):
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.buf = torch.zeros(numel, dtype=dtype, device=device) if self.dist is not None else None
        self.work = None
        self.launched = 0

    def start(self):
        if self.dist is not None:
            self.work = self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.SUM, async_op=True)
            self.launched += 1

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
