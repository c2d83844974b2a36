#!/usr/bin/env python3
"""Experiment harness (not product code): times the on-the-fly kernels of a library variant.
usage: alt_lab.py [path/to/libraftcorr_variant.so] [H W B]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import _lib
if len(sys.argv) > 1 and sys.argv[1].endswith(".so"):
    _lib.LIB_PATH = os.path.abspath(sys.argv.pop(1))
from raft_amd import AlternateCorrBlock
import bench
H, W, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (47, 156, 1)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
f1 = torch.randn(B, 256, H, W, generator=g).to(dev); f2 = torch.randn(B, 256, H, W, generator=g).to(dev)
coords = [c.to(dev) for c in bench.flow_coords(torch, B, H, W, 6, g)]
lib = _lib.load()
with torch.no_grad():
    blk = AlternateCorrBlock(f1, f2, 4, 4)
    o = torch.empty(B, 324, H, W, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    res = {}
    for tag, c in (("identity", coords[0]), ("sigma3", coords[1]), ("sigma4.2", coords[2]), ("sigma6", coords[4])):
        res[tag] = round(bench.ev_time_us(torch, lambda: lib.raftcorr_alt_lookup_fwd(blk._p1.data_ptr(), blk._p2ptrs, blk._scales.data_ptr(), B, H, W, 256, 4, 4, c.data_ptr(), o.data_ptr(), st), reps=50, warm=5), 1)
print(os.path.basename(_lib.LIB_PATH), f"{H}x{W} B={B} alt fwd us:", res)
if "STAMPS" in _lib.LIB_PATH:
    import numpy as np
    for tag, c in (("identity", coords[0]), ("sigma4.2", coords[2])):
        o.zero_()
        lib.raftcorr_alt_lookup_fwd(blk._p1.data_ptr(), blk._p2ptrs, blk._scales.data_ptr(), B, H, W, 256, 4, 4, c.data_ptr(), o.data_ptr(), st)
        torch.cuda.synchronize()
        raw = o.view(-1).cpu().numpy().view(np.uint64)
        nwg = 8 * ((((W + 7) // 8) * ((H + 7) // 8) + 7) // 8) * 4 * B
        stp = raw[:nwg * 16].reshape(nwg, 16).astype(np.int64)
        live = stp[:, 1] > 0
        stp = stp[live]
        t0 = stp[:, 0].min()
        print(tag, "WGs", len(stp), "span us", (stp[:, :14].max() - t0) / 100.0)
        for l in range(4):
            m = stp[stp[:, 14] == l]
            if len(m) == 0: continue
            rel = (m[:, :14] - t0) / 100.0
            npass = np.minimum((m[:, 15] + 191) // 192, 3)
            # stamps: 0 start, 1 setup, then per pass: K start, K end, S written; last: end
            print(f" level {l}: U avg {m[:,15].mean():.0f} passes {npass.mean():.1f} | start {rel[:,0].mean():.2f} setup {rel[:,1].mean():.2f} | pass0: Kstart {rel[:,2].mean():.2f} Kend {rel[:,3].mean():.2f} Swritten {rel[:,4].mean():.2f} | last stamp avg {np.array([rel[i, 2 + 3 * npass[i]] for i in range(len(m))]).mean():.2f} max {np.array([rel[i, 2 + 3 * npass[i]] for i in range(len(m))]).max():.2f}")
