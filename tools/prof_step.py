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
with torch.no_grad():
    for _ in range(steps):
        bench.one_pair(f1, f2, coords)
torch.cuda.synchronize()
print("done")
