#!/usr/bin/env python3
"""DDP train-step harness (BASELINE.json configs[3]): RAFT-basic
(harness/raft_model.py) on synthetic FlyingThings-sized crops (400x720, batch 6 per GPU, 12 refinement
iterations), sequence loss, AdamW + StepLR + gradient clipping as core/train_origin.py:84-95,
207-216, one process per GPU with DistributedDataParallel -- the gradient all-reduce
(5.26 M fp32 = 21 MB per step, bucketed and overlapped with backward by DDP) is the only collective
and runs on RCCL over xGMI (backend "nccl").  The reference uses single-process nn.DataParallel
(train.py:342), which is not replicated.

    python tools/train_step.py [--steps K] [--warmup W] [--corr ours|ours_fused|stock|alt] [--batch 6] [--iters 12]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P tools/train_step.py ...
Prints one JSON line from rank 0: image pairs/s over all ranks, ms per step, and the share of the
step spent in the correlation block's forward (HIP events) -- backward runs inside autograd."""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from harness.raft_model import RAFT, sequence_loss, synthetic_pair   # noqa: E402
from harness.stock_corr import StockCorrBlock                        # noqa: E402


def corr_factory(kind):
    if kind == "stock":
        return lambda a, b: StockCorrBlock(a, b, 4, 4)
    from raft_amd import AlternateCorrBlock, CorrBlock
    if kind == "alt":
        return lambda a, b: AlternateCorrBlock(a, b, num_levels=4, radius=4)
    return lambda a, b: CorrBlock(a, b, num_levels=4, radius=4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--corr", default="ours", choices=["ours", "ours_fused", "stock", "alt"])
    ap.add_argument("--batch", type=int, default=6)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--size", type=int, nargs=2, default=[400, 720])
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = args.device == "cuda"
    dev = torch.device("cuda", local) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl" if on_gpu else "gloo", **({"device_id": dev} if on_gpu else {}))

    torch.manual_seed(0)                                   # identical initial weights on every rank
    # ours_fused: lookup fused with the motion encoder's convc1 + ReLU (differentiable, raft_amd/corr.py:_LookupConv)
    model = RAFT(corr_factory("ours" if args.corr == "ours_fused" else args.corr),
                 fuse_convc1=args.corr == "ours_fused").to(dev).train()
    for m in model.modules():                              # freeze_bn (core/raft.py:60-63), as every stage after chairs
        if isinstance(m, nn.BatchNorm2d):
            m.eval()
    net = nn.parallel.DistributedDataParallel(model, device_ids=[local] if on_gpu else None) if world > 1 else model
    opt = torch.optim.AdamW(net.parameters(), lr=2e-4, weight_decay=5e-5, eps=1e-8)
    sched = torch.optim.lr_scheduler.StepLR(opt, round(100000 * 0.8))
    i1, i2, gt = synthetic_pair(args.batch, *args.size, seed=100 + rank, device=dev)   # each rank its own shard
    valid = torch.ones_like(gt[:, 0])

    def sync():
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    def step():
        opt.zero_grad(set_to_none=True)
        loss, metrics = sequence_loss(net(i1, i2, iters=args.iters), gt, valid)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        sched.step()
        return float(loss.detach())

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el)
    if rank == 0:
        print(json.dumps({"metric": "training image pairs/sec (RAFT-basic, synthetic FlyingThings crops)",
                          "value": world * args.steps * args.batch / el, "n_gpus": world, "ms_per_step": el / args.steps * 1e3,
                          "steps": args.steps, "warmup": args.warmup, "corr": args.corr, "batch_per_gpu": args.batch,
                          "iters": args.iters, "crop": args.size, "final_loss": loss, "scaling": "weak",
                          "collective": "DDP gradient all-reduce, 5257536 fp32" if world > 1 else None}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
