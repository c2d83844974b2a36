"""CPU, world_size 2, gloo: the N>1 path of bench.py (batch sharding of frame pairs, barrier,
max-over-ranks timing).  No data-path collective exists on this path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raft_amd.sharding import shard_range, max_over_ranks, sum_over_ranks


def test_shard_range_partitions_exactly():
    for total in [0, 1, 7, 8, 9, 64, 1000]:
        for world in [1, 2, 3, 4, 8]:
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(13, rank, world)
    dist.barrier()
    elapsed = 1.0 + rank            # rank 1 is the slow one
    mx = max_over_ranks(elapsed)
    done = sum_over_ranks(hi - lo)
    q.put((rank, lo, hi, mx, done))
    dist.destroy_process_group()


def test_two_rank_gloo_timing_and_sharding():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, mx0, d0), (r1, lo1, hi1, mx1, d1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 7, 7, 13)
    assert mx0 == mx1 == 2.0          # MAX over ranks, as bench.py reports
    assert d0 == d1 == 13.0           # whole-job units = sum over ranks
