#!/usr/bin/env python3
"""Tiny driver for rocprofv3 passes (kernel trace or PMC): a few steps of every kernel of the path at one of the
bench shapes.  usage: prof_step.py [steps] [sintel|kitti_alt|things_train|1080p]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = sys.argv[2] if len(sys.argv) > 2 else "sintel"
dev = torch.device("cuda:0")
f1, f2, coords, gouts = bench.make_inputs(torch, cfg, dev, 1234)
L, R = bench.LEVELS, bench.RADIUS
from raft_amd import AlternateCorrBlock, CorrBlock  # noqa: E402
big = cfg == "1080p"
with torch.no_grad():
    for _ in range(steps):                                   # the bench step: build + lookups
        cb = CorrBlock(f1, f2, num_levels=L, radius=R, **({"precision": "bf16", "volume_dtype": "bf16"} if big else {}))
        for c in coords:
            cb(c)
        del cb
    for _ in range(steps):                                   # the path's other forward kernels, same shape
        for prec, vol in (("bf16", "fp32"), ("bf16x3", "fp32"), ("fp16x2", "fp32"), ("fp16x2", "fp16")):
            cb = CorrBlock(f1, f2, num_levels=L, radius=R, precision=prec, volume_dtype=vol)
            cb(coords[1])
            del cb
        if not big:
            cbf = CorrBlock(f1, f2, num_levels=L, radius=R)
            wt = torch.randn(256, L * (2 * R + 1) ** 2, device=dev) / 18
            cbf.lookup_conv1x1(coords[1], wt, torch.zeros(256, device=dev))
            del cbf
        alt = AlternateCorrBlock(f1, f2, num_levels=L, radius=R)
        alt(coords[0]); alt(coords[2])
if not big:
    g1, g2 = f1.detach().clone().requires_grad_(True), f2.detach().clone().requires_grad_(True)
    K = len(coords) if cfg in ("kitti_alt", "things_train") else 3      # those two: the bench step itself
    gs = gouts[:K] if gouts is not None else [torch.randn(f1.shape[0], L * (2 * R + 1) ** 2, *f1.shape[-2:], device=dev)] * K
    for _ in range(steps):                                   # backward kernels
        g1.grad = g2.grad = None
        cb = CorrBlock(g1, g2, num_levels=L, radius=R)
        torch.autograd.backward([cb(c) for c in coords[:K]], gs)
        g1.grad = g2.grad = None
        ab = AlternateCorrBlock(g1, g2, num_levels=L, radius=R, tiles="window" if cfg == "kitti_alt" else "pixel")
        torch.autograd.backward([ab(c) for c in coords[:K]], gs)
        del cb, ab
torch.cuda.synchronize()
print("done")
