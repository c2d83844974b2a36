#!/usr/bin/env python3
"""Experiment harness (not product code): times the builds into 16-bit volumes.  usage: build16_lab.py [lib.so]"""
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
for cfg, reps in (("sintel", 100), ("things_train", 30), ("1080p", 8)):
    f1, f2, coords, _ = bench.make_inputs(torch, cfg, dev, 1)
    res = {}
    with torch.no_grad():
        for prec, vol in (("bf16", "fp32"), ("bf16", "bf16"), ("fp16x2", "fp32"), ("fp16x2", "fp16")):
            res[f"{prec}/{vol}"] = round(bench.ev_time_us(torch, lambda: CorrBlock(f1, f2, num_levels=4, radius=4, precision=prec, volume_dtype=vol),
                                                          reps=reps, warm=max(3, reps // 4)), 1)
            torch.cuda.empty_cache()
    print(os.path.basename(_lib.LIB_PATH), cfg, "build us (precision/volume):", res, flush=True)
