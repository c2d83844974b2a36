#!/usr/bin/env python3
"""Experiment harness (not product code): times the on-the-fly backward of a library variant.
usage: alt_bwd_lab.py [path/to/libraftcorr_variant.so] [H W B]"""
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
D, L, R = 256, 4, 4
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
f1 = torch.randn(B, D, H, W, generator=g).to(dev).requires_grad_(True); f2 = torch.randn(B, D, H, W, generator=g).to(dev)
coords = [c.to(dev) for c in bench.flow_coords(torch, B, H, W, 6, g)]
gout = torch.randn(B, L * 81, H, W, generator=g).to(dev)
lib = _lib.load()
blk = AlternateCorrBlock(f1, f2, L, R)
with torch.no_grad():
    d1 = torch.empty(B, H, W, D, device=dev)
    d2 = [torch.empty(B, H >> l, W >> l, D, device=dev) for l in range(L)]
    dlv = _lib.ptr_array(d2)
    wb = lib.raftcorr_alt_lookup_bwd_workspace_bytes(B, H, W, D, L, R)
    work = torch.empty(wb, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    res = {}
    for tag, c in (() if os.environ.get("WINDOW_ONLY") else (("identity", coords[0]), ("sigma3", coords[1]), ("sigma4.2", coords[2]), ("sigma6", coords[4]))):
        res[tag] = round(bench.ev_time_us(torch, lambda: lib.raftcorr_alt_lookup_bwd(
            blk._t1.data_ptr(), blk._t2ptrs, blk._scales.data_ptr(), B, H, W, D, L, R, c.data_ptr(), gout.data_ptr(),
            d1.data_ptr(), dlv, int(os.environ.get('ACC', '0')), work.data_ptr(), wb, st), reps=30, warm=5), 1)
    swb = lib.raftcorr_alt_lookup_bwd_sorted_workspace_bytes(B, H, W, D, L, R)
    swork = torch.empty(swb, dtype=torch.uint8, device=dev)
    res_w = {}
    for tag, c in (("identity", coords[0]), ("sigma3", coords[1]), ("sigma4.2", coords[2]), ("sigma6", coords[4])):
        res_w[tag] = round(bench.ev_time_us(torch, lambda: lib.raftcorr_alt_lookup_bwd_sorted(
            blk._p1.data_ptr(), blk._t2ptrs, blk._scales.data_ptr(), B, H, W, D, L, R, c.data_ptr(), gout.data_ptr(),
            d1.data_ptr(), dlv, int(os.environ.get('ACC', '0')), None, swork.data_ptr(), swb, st), reps=30, warm=5), 1)
    # with the groups of the forward pass: the backward call skips its own count / scan / fill
    gb = lib.raftcorr_alt_lookup_fwd_sorted_workspace_bytes(B, H, W, L)
    groups = torch.empty(gb, dtype=torch.uint8, device=dev)
    o = torch.empty(B, L * (2 * R + 1) ** 2, H, W, device=dev)
    res_g, res_f = {}, {}
    for tag, c in (("identity", coords[0]), ("sigma3", coords[1]), ("sigma4.2", coords[2]), ("sigma6", coords[4])):
        fwd = lambda: lib.raftcorr_alt_lookup_fwd_sorted(blk._p1.data_ptr(), blk._p2ptrs, blk._scales.data_ptr(), B, H, W, D, L, R,
                                                         c.data_ptr(), o.data_ptr(), groups.data_ptr(), gb, st)
        res_f[tag] = round(bench.ev_time_us(torch, fwd, reps=30, warm=5), 1)
        res_g[tag] = round(bench.ev_time_us(torch, lambda: lib.raftcorr_alt_lookup_bwd_sorted(
            blk._p1.data_ptr(), blk._t2ptrs, blk._scales.data_ptr(), B, H, W, D, L, R, c.data_ptr(), gout.data_ptr(),
            d1.data_ptr(), dlv, int(os.environ.get('ACC', '0')), groups.data_ptr(), swork.data_ptr(), swb, st), reps=30, warm=5), 1)
    print(os.path.basename(_lib.LIB_PATH), f"{H}x{W} B={B} alt fwd, window groups, us:", res_f, flush=True)
    print(os.path.basename(_lib.LIB_PATH), f"{H}x{W} B={B} alt bwd, window groups brought by the caller, us:", res_g, flush=True)
print(os.path.basename(_lib.LIB_PATH), f"{H}x{W} B={B} alt bwd us:", res, flush=True)
print(os.path.basename(_lib.LIB_PATH), f"{H}x{W} B={B} alt bwd, window groups, us:", res_w, flush=True)
