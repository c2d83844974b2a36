#!/usr/bin/env python3
"""Timings of the other BASELINE.json configurations' pieces (not the headline bench):
alt (on-the-fly) fwd/bwd at KITTI 47x156, all-pairs fwd+bwd at FlyingThings 50x90 B=6."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import CorrBlock, AlternateCorrBlock

dev = torch.device("cuda:0")

def ev_time(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

def coords_for(B, H, W, sigma, g):
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    base = torch.stack([xx, yy], 0)[None].repeat(B, 1, 1, 1)
    return (base + sigma * torch.randn(B, 2, H, W, generator=g)).to(dev)

def smooth_coords_for(B, H, W, amp, g):
    """piecewise-smooth flow, the kind RAFT iterates on: a coarse random field upsampled + 0.25 px noise"""
    coarse = amp * torch.randn(B, 2, 2, 3, generator=g)
    flow = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=True)
    return coords_for(B, H, W, 0.25, g) + flow.to(dev)

g = torch.Generator().manual_seed(0)
res = {}
# config 3: KITTI alt path
B, D, H, W = 1, 256, 47, 156
f1 = torch.randn(B, D, H, W, generator=g).to(dev); f2 = torch.randn(B, D, H, W, generator=g).to(dev)
co = coords_for(B, H, W, 4.0, g)
with torch.no_grad():
    blk = AlternateCorrBlock(f1, f2, 4, 4)
    res["kitti_alt_fwd_us"] = ev_time(lambda: blk(co))
    cb = CorrBlock(f1, f2, 4, 4)
    res["kitti_allpairs_lookup_us"] = ev_time(lambda: cb(co))
    res["kitti_allpairs_build_us"] = ev_time(lambda: CorrBlock(f1, f2, 4, 4))
f1g = f1.clone().requires_grad_(True); f2g = f2.clone().requires_grad_(True)
gout = torch.randn(B, 324, H, W, generator=g).to(dev)
def alt_fb():
    f1g.grad = None; f2g.grad = None
    AlternateCorrBlock(f1g, f2g, 4, 4)(co).backward(gout)
res["kitti_alt_fwd_bwd_us"] = ev_time(alt_fb, reps=5, warm=2)
co_s = smooth_coords_for(B, H, W, 8.0, g)
with torch.no_grad():
    res["kitti_alt_fwd_smooth_us"] = ev_time(lambda: blk(co_s))
    co_i = coords_for(B, H, W, 0.0, g)
    res["kitti_alt_fwd_identity_us"] = ev_time(lambda: blk(co_i))
def alt_fb_s():
    f1g.grad = None; f2g.grad = None
    AlternateCorrBlock(f1g, f2g, 4, 4)(co_s).backward(gout)
res["kitti_alt_fwd_bwd_smooth_us"] = ev_time(alt_fb_s, reps=5, warm=2)
# bf16-operand build at the headline shape (Sintel 55x128) and at config 5's 1080p shape
for tag, (Hs, Ws) in {"sintel": (55, 128), "1080p": (136, 240)}.items():
    a = torch.randn(1, 256, Hs, Ws, generator=g).to(dev); bb = torch.randn(1, 256, Hs, Ws, generator=g).to(dev)
    with torch.no_grad():
        res[f"{tag}_build_fp32_us"] = ev_time(lambda: CorrBlock(a, bb, 4, 4), reps=10)
        res[f"{tag}_build_bf16_us"] = ev_time(lambda: CorrBlock(a, bb, 4, 4, precision="bf16"), reps=10)
        res[f"{tag}_build_bf16x3_us"] = ev_time(lambda: CorrBlock(a, bb, 4, 4, precision="bf16x3"), reps=10)
        res[f"{tag}_build_fp16x2_us"] = ev_time(lambda: CorrBlock(a, bb, 4, 4, precision="fp16x2"), reps=10)
        cbx = CorrBlock(a, bb, 4, 4)
        cox = coords_for(1, Hs, Ws, 3.0, g)
        res[f"{tag}_lookup_us"] = ev_time(lambda: cbx(cox))
        blkx = AlternateCorrBlock(a, bb, 4, 4)
        res[f"{tag}_alt_fwd_us"] = ev_time(lambda: blkx(cox), reps=10)
        del cbx
# achievable HBM rates on this box (quote fractions of nominal and of achievable)
big = torch.empty(1 << 30, dtype=torch.float32, device=dev)          # 4 GiB
big2 = torch.empty(1 << 30, dtype=torch.float32, device=dev)
res["hbm_fill_GBs"] = big.numel() * 4 / ev_time(lambda: big.zero_(), reps=5) / 1e3
res["hbm_copy_GBs"] = 2 * big.numel() * 4 / ev_time(lambda: big2.copy_(big), reps=5) / 1e3
res["hbm_read_GBs"] = big.numel() * 4 / ev_time(lambda: big.sum(), reps=5) / 1e3
del big, big2
# lookup fused with convc1 (324 -> 256, bias, ReLU) vs lookup + stock conv2d, Sintel shape
a = torch.randn(1, 256, 55, 128, generator=g).to(dev); bb = torch.randn(1, 256, 55, 128, generator=g).to(dev)
wt = (torch.randn(256, 324, 1, 1, generator=g) / 18).to(dev); bs = torch.randn(256, generator=g).to(dev)
cox = coords_for(1, 55, 128, 3.0, g)
with torch.no_grad():
    cbx = CorrBlock(a, bb, 4, 4)
    res["sintel_lookup_plus_conv2d_us"] = ev_time(lambda: torch.relu(torch.nn.functional.conv2d(cbx(cox), wt, bs)))
    res["sintel_lookup_conv1x1_fused_us"] = ev_time(lambda: cbx.lookup_conv1x1(cox, wt, bs))
    del cbx
# config 4: FlyingThings crops B=6, 12 lookups fwd + bwd
B, D, H, W = 6, 256, 50, 90
f1 = torch.randn(B, D, H, W, generator=g).to(dev).requires_grad_(True)
f2 = torch.randn(B, D, H, W, generator=g).to(dev).requires_grad_(True)
cos = [coords_for(B, H, W, 2.0 * (k + 1) ** 0.5, g) for k in range(12)]
gouts = [torch.randn(B, 324, H, W, generator=g).to(dev) for _ in range(12)]
def things_fwd():
    with torch.no_grad():
        cb = CorrBlock(f1, f2, 4, 4)
        for c in cos: cb(c)
def things_fb():
    f1.grad = None; f2.grad = None
    cb = CorrBlock(f1, f2, 4, 4)
    loss = 0
    for c, go in zip(cos, gouts): loss = loss + (cb(c) * go).sum()
    loss.backward()
res["things_b6_fwd_us"] = ev_time(things_fwd, reps=10)
res["things_b6_fwd_bwd_us"] = ev_time(things_fb, reps=5, warm=2)
print(json.dumps(res))
