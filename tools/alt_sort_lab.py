#!/usr/bin/env python3
"""Experiment harness (not product code): on-the-fly forward with pixel tiles against window-sorted tiles.
usage: alt_sort_lab.py [lib.so] [H W B]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import _lib
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    _lib.LIB_PATH = os.path.abspath(sys.argv.pop(1))
from raft_amd import AlternateCorrBlock
import bench
H, W, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (47, 156, 1)
D, L, R = 256, 4, 4
dev = torch.device("cuda:0")
lib = _lib.load()
g = torch.Generator(device="cpu").manual_seed(3)
f1 = torch.randn(B, D, H, W, generator=g).to(dev)
f2 = torch.randn(B, D, H, W, generator=g).to(dev)
yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
base = torch.stack([xx, yy], 0)[None].float().repeat(B, 1, 1, 1)
st = torch.cuda.current_stream(dev).cuda_stream
with torch.no_grad():
    blk = AlternateCorrBlock(f1, f2, num_levels=L, radius=R)
    wb = lib.raftcorr_alt_lookup_fwd_sorted_workspace_bytes(B, H, W, L)
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    o1 = torch.empty(B, L * 81, H, W, device=dev)
    o2 = torch.empty_like(o1)
    for tag, sig in (("identity", 0.0), ("smooth+7.3", -1.0), ("sigma1", 1.0), ("sigma3", 3.0), ("sigma4.2", 4.2), ("sigma6", 6.0)):
        c = (base + 7.3 if sig < 0 else base + sig * torch.randn(B, 2, H, W, generator=g)).to(dev).contiguous()
        plain = lambda: lib.raftcorr_alt_lookup_fwd(blk._p1.data_ptr(), blk._p2ptrs, blk._scales.data_ptr(), B, H, W, D, L, R,
                                                    c.data_ptr(), o1.data_ptr(), st)
        srt = lambda: lib.raftcorr_alt_lookup_fwd_sorted(blk._p1.data_ptr(), blk._p2ptrs, blk._scales.data_ptr(), B, H, W, D, L, R,
                                                         c.data_ptr(), o2.data_ptr(), work.data_ptr(), wb, st)
        o1.fill_(float("nan")); o2.fill_(float("nan"))
        assert plain() == 0 and srt() == 0
        torch.cuda.synchronize()
        same = f"{(o1 - o2).abs().max().item() / o1.abs().max().item():.1e}"
        t1 = bench.ev_time_us(torch, plain, reps=30, warm=5)
        t2 = bench.ev_time_us(torch, srt, reps=30, warm=5)
        nt = work[:4 * B * L].view(torch.int32).cpu().tolist()
        print(f"{H}x{W} B={B} {tag:10s} maxdiff/max={same} pixel tiles {t1:7.1f} us  sorted {t2:7.1f} us  tiles/level {nt}", flush=True)
