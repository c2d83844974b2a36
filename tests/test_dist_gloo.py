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


def _bucket_worker(rank, world, port, q):
    """What bench.py's main() and ThingsTrain.step do for N > 1: init_ranks + GradBucket.start / finish."""
    from raft_amd.sharding import GradBucket, init_ranks
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    d = init_ranks("gloo", torch.device("cpu"))
    b = GradBucket(5_257_536, torch.device("cpu"))          # bench.py's GRAD_BUCKET
    b.buf.fill_(float(rank + 1))
    b.start()
    x = torch.ones(4).sum().item()                          # "backward" work between issue and wait
    b.finish()
    first = (float(b.buf[0]), float(b.buf[-1]))
    b.start(); b.finish()                                   # a second step reuses the bucket
    q.put((rank, first, float(b.buf[7]), b.launched, x))
    d.destroy_process_group()


def test_gradient_bucket_allreduce_two_ranks_gloo():
    """The bench's only collective (config 3) through the same function bench.py calls, 2 CPU ranks over gloo,
    async issue + wait: SUM over ranks, bucket reusable from step to step; no-op without a process group."""
    from raft_amd.sharding import GradBucket
    solo = GradBucket(16, torch.device("cpu"))
    solo.start(); solo.finish()
    assert solo.buf is None and solo.launched == 0
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, first, second, launched, _ in res:
        assert first == (3.0, 3.0) and second == 6.0 and launched == 2


def test_bench_things_train_dry_two_ranks_runs_the_allreduce():
    """`bench.py --config things_train --gpus 2 --device cpu`: the dry mode makes the same GradBucket calls per step
    as ThingsTrain.step, over gloo: one all-reduce per warm-up and timed step."""
    out, line = _run_bench(["--gpus", "2", "--config", "things_train"])
    assert out.returncode == 0, out.stderr[-2000:]
    assert line["ranks_seen"] == 2 and line["dry_run"] is True
    assert line["allreduces_launched"] == line["steps"] + line["warmup_steps_run"] == 4


def test_train_step_harness_two_ranks_gloo():
    """tools/train_step.py (DDP train-step harness) wired correctly for N > 1: two
    CPU ranks over gloo with the stock-operator correlation block (the product kernels need a
    GPU; on the GPU box the same script runs with --corr ours over RCCL)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "tools", "train_step.py"),
           "--device", "cpu", "--corr", "stock", "--steps", "2", "--warmup", "1", "--batch", "1", "--iters", "2",
           "--size", "128", "160"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["final_loss"] == line["final_loss"]


def _run_bench(extra, env_extra=None):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(OMP_NUM_THREADS="2", **(env_extra or {}))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--device", "cpu", "--steps", "3",
                          "--warmup", "1"] + extra, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_2_without_torchrun_launches_two_ranks():
    """`python bench.py --gpus 2` with no torchrun must start two ranks itself (dry mode: gloo, no
    kernels) and report what actually ran: ranks_seen == n_gpus == 2, never a figure scaled by the flag."""
    out, line = _run_bench(["--gpus", "2"])
    assert out.returncode == 0, out.stderr[-2000:]
    assert line["ranks_seen"] == 2 and line["n_gpus"] == 2 and line["dry_run"] is True and line["value"] is None
    assert len([l for l in out.stdout.splitlines() if l.startswith("{")]) == 1     # rank 0 only


def test_bench_refuses_a_rank_count_that_differs_from_gpus():
    out, line = _run_bench(["--gpus", "2"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and line is None
    assert "--gpus 2 but 1 rank(s)" in out.stderr
    out, line = _run_bench([])
    assert out.returncode == 0 and line["ranks_seen"] == 1 and line["n_gpus"] == 1
