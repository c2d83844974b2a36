#!/usr/bin/env python3
"""Experiment: lookup kernel time under different coordinate patterns (cache behaviour)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from raft_amd import CorrBlock

dev = torch.device("cuda:0")
f1, f2, coords = bench.make_inputs(dev, 1234)
cb = CorrBlock(f1, f2, num_levels=4, radius=4)

def timeit(name, clist, reps=10):
    for c in clist: cb(c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for c in clist: cb(c)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1)*1e3/(reps*len(clist)):8.2f} us/lookup")

with torch.no_grad():
    timeit("drift sigma=3 per iter (bench)", coords)
    timeit("same coords repeated (iter 5)", [coords[5]] * 32)
    timeit("identity grid repeated", [coords[0]] * 32)
    far = [c + 1000.0 for c in coords]
    timeit("all taps out of range (no reads)", far)
    c3 = [c.permute(0, 2, 3, 1).reshape(1, -1, 2).contiguous() for c in coords]
    timeit("fork layout, drift", c3)
