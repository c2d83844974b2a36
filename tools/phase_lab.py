#!/usr/bin/env python3
"""Experiment harness (not product code): persistent 16-bit build with the two workgroups of a CU started in opposite
phases (RC_PHASE_DELAY = units of s_sleep(127) for workgroups in odd wave slots).  usage: phase_lab.py tools/libraftcorr_lab.so"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from raft_amd import CorrBlock
import bench
dev = torch.device("cuda:0")
f1, f2, coords, _ = bench.make_inputs(torch, "1080p", dev, 1)
def t():
    return round(bench.ev_time_us(torch, lambda: CorrBlock(f1, f2, num_levels=4, radius=4, precision="bf16", volume_dtype="bf16"), reps=8, warm=3), 1)
with torch.no_grad():
    os.environ["RC_PERSIST"] = "0"; os.environ["RC_RUN_LEN"] = "32"
    ref = CorrBlock(f1, f2, num_levels=4, radius=4, precision="bf16", volume_dtype="bf16")(coords[3]).clone()
    print("per-patch kernel:", t(), flush=True)
    os.environ["RC_PERSIST"] = "1"
    for rl in ("32", "8"):
        os.environ["RC_RUN_LEN"] = rl
        for d in ("0", "1", "2", "3", "4", "6"):
            os.environ["RC_PHASE_DELAY"] = d
            out = CorrBlock(f1, f2, num_levels=4, radius=4, precision="bf16", volume_dtype="bf16")(coords[3])
            print("persistent run_len", rl, "delay", d, ":", t(), "same bits" if torch.equal(out, ref) else "DIFFERENT", flush=True)
