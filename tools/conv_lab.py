#!/usr/bin/env python3
"""Experiment harness (not product code): times the fused lookup + convc1 call.  usage: conv_lab.py [lib.so]"""
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
for cfg in ("sintel", "things_train"):
    f1, f2, coords, _ = bench.make_inputs(torch, cfg, dev, 1)
    with torch.no_grad():
        cb = CorrBlock(f1, f2, num_levels=4, radius=4)
        wt = torch.randn(256, 324, device=dev) / 18
        bias = torch.zeros(256, device=dev)
        t_f = bench.ev_time_us(torch, lambda: cb.lookup_conv1x1(coords[1], wt, bias), reps=50, warm=5)
        t_l = bench.ev_time_us(torch, lambda: cb(coords[1]), reps=50, warm=5)
        conv = torch.nn.Conv2d(324, 256, 1).to(dev)
        x = cb(coords[1])
        t_c = bench.ev_time_us(torch, lambda: torch.relu(conv(x)), reps=50, warm=5)
    print(os.path.basename(_lib.LIB_PATH), cfg, f"fused {t_f:.1f} us | lookup {t_l:.1f} + conv2d+relu {t_c:.1f} us", flush=True)
