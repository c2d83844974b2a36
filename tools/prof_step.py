#!/usr/bin/env python3
"""Tiny driver for rocprofv3 PMC passes: a few steps of the bench workload (build + 32 lookups)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
f1, f2, coords = bench.make_inputs(dev, 1234)
from raft_amd import AlternateCorrBlock, CorrBlock  # noqa: E402
with torch.no_grad():
    for _ in range(steps):
        bench.one_pair(f1, f2, coords)
    for _ in range(steps):                                   # the path's other kernels, same shape
        for prec in ("bf16", "bf16x3", "fp16x2"):
            CorrBlock(f1, f2, num_levels=bench.LEVELS, radius=bench.RADIUS, precision=prec)
        cbf = CorrBlock(f1, f2, num_levels=bench.LEVELS, radius=bench.RADIUS)
        wt = torch.randn(256, bench.LEVELS * (2 * bench.RADIUS + 1) ** 2, device=dev) / 18
        cbf.lookup_conv1x1(coords[1], wt, torch.zeros(256, device=dev))
        alt = AlternateCorrBlock(f1, f2, num_levels=bench.LEVELS, radius=bench.RADIUS)
        alt(coords[0]); alt(coords[2])
g1, g2 = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
for _ in range(steps):                                       # backward kernels (3 lookups per pair)
    g1.grad = g2.grad = None
    cb = CorrBlock(g1, g2, num_levels=bench.LEVELS, radius=bench.RADIUS)
    sum(cb(c).square().sum() for c in coords[:3]).backward()
    g1.grad = g2.grad = None
    AlternateCorrBlock(g1, g2, num_levels=bench.LEVELS, radius=bench.RADIUS)(coords[2]).square().sum().backward()
torch.cuda.synchronize()
print("done")
