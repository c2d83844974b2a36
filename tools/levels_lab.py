#!/usr/bin/env python3
"""Experiment harness (not product code): build time against the number of pyramid levels, per volume type.
usage: levels_lab.py [lib.so]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import _lib
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    _lib.LIB_PATH = os.path.abspath(sys.argv.pop(1))
from raft_amd import CorrBlock
import bench
dev = torch.device("cuda:0")
for cfg, reps in (("sintel", 100), ("1080p", 8)):
    f1, f2, coords, _ = bench.make_inputs(torch, cfg, dev, 1)
    with torch.no_grad():
        for prec, vol in (("bf16", "fp32"), ("bf16", "bf16")):
            res = {}
            for L in (1, 2, 3, 4):
                res[L] = round(bench.ev_time_us(torch, lambda: CorrBlock(f1, f2, num_levels=L, radius=4, precision=prec, volume_dtype=vol),
                                                reps=reps, warm=max(3, reps // 4)), 1)
                torch.cuda.empty_cache()
            print(os.path.basename(_lib.LIB_PATH), cfg, f"{prec}/{vol} build us by num_levels:", res, flush=True)
