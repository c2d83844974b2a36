#!/usr/bin/env python3
"""Randomised parity run (test infrastructure, not product code): random shapes, radii, level counts, coordinate
conventions and flow fields through every block of the library, forward and backward, against the oracle's restatement
of the reference on the CPU.  Prints one line per failure and a summary; exit status 1 if anything exceeded its bar.
    python tools/fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raft_amd import AlternateCorrBlock, CorrBlock   # noqa: E402
from oracle import corr_oracle as O                  # noqa: E402


def rel(a, b):
    s = float(np.abs(b).max())
    return float(np.abs(a - b).max()) / (s if s > 0 else 1.0)


budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
t_end = time.time() + budget
n, worst, fails = 0, {}, []
while time.time() < t_end:
    B = int(rs.randint(1, 4)); D = int(rs.choice([3, 8, 17, 32, 64, 96, 128, 200, 256]))
    H, W = int(rs.randint(8, 49)), int(rs.randint(8, 81))
    L = int(rs.randint(1, 5))
    while min(H, W) >> (L - 1) < 2:         # a level of size 1 makes the reference divide by zero (2 x / (W - 1) - 1): NaN there
        L -= 1
    r = int(rs.choice([0, 1, 2, 3, 4, 4, 4, 5, 7]))
    kind = rs.choice(["smooth", "noise", "wild"])
    block = rs.choice(["all_pairs", "all_pairs_accumulate", "all_pairs_fp16x2", "fork", "on_the_fly", "on_the_fly_window"])
    K = int(rs.randint(1, 5))
    f1n, f2n = rs.randn(B, D, H, W).astype(np.float32), rs.randn(B, D, H, W).astype(np.float32)
    if rs.rand() < 0.2:
        f1n *= 10.0 ** rs.uniform(-3, 3)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    base = np.stack([xx, yy], 0)[None].repeat(B, 0).astype(np.float32)
    cons = []
    for _ in range(K):
        if kind == "smooth":
            c = base + rs.uniform(-6, 6, (B, 2, 1, 1)).astype(np.float32) + 0.3 * rs.randn(B, 2, H, W).astype(np.float32)
        elif kind == "noise":
            c = base + rs.uniform(1, 8) * rs.randn(B, 2, H, W).astype(np.float32)
        else:
            c = (rs.rand(B, 2, H, W) * np.array([W + 30, H + 30]).reshape(1, 2, 1, 1) - 15).astype(np.float32)
        cons.append(np.ascontiguousarray(c, dtype=np.float32))
    C = L * (2 * r + 1) ** 2
    fork = block == "fork"
    if fork:
        cons = [np.ascontiguousarray(c.reshape(B, 2, H * W).transpose(0, 2, 1)) for c in cons]
        gouts = [rs.randn(B, H * W, C).astype(np.float32) for _ in cons]
    else:
        gouts = [rs.randn(B, C, H, W).astype(np.float32) for _ in cons]

    def run(make, device):
        f1 = torch.from_numpy(f1n).to(device).requires_grad_(True)
        f2 = torch.from_numpy(f2n).to(device).requires_grad_(True)
        cb = make(f1, f2)
        outs = [cb(torch.from_numpy(c).to(device)) for c in cons]
        torch.autograd.backward(outs, [torch.from_numpy(g).to(device) for g in gouts])
        return [o.detach().cpu().numpy() for o in outs], f1.grad.cpu().numpy(), f2.grad.cpu().numpy()

    make = {"all_pairs": lambda a, b: CorrBlock(a, b, num_levels=L, radius=r),
            "fork": lambda a, b: CorrBlock(a, b, num_levels=L, radius=r),
            "all_pairs_accumulate": lambda a, b: CorrBlock(a, b, num_levels=L, radius=r, backward="accumulate"),
            "all_pairs_fp16x2": lambda a, b: CorrBlock(a, b, num_levels=L, radius=r, precision="fp16x2"),
            "on_the_fly": lambda a, b: AlternateCorrBlock(a, b, num_levels=L, radius=r),
            "on_the_fly_window": lambda a, b: AlternateCorrBlock(a, b, num_levels=L, radius=r, tiles="window")}[block]
    tag = f"{block} B{B} D{D} {H}x{W} L{L} r{r} {kind} K{K}"
    try:
        outs, g1, g2 = run(make, dev)
        ro, o1, o2 = run(lambda a, b: O.CorrBlockOracle(a, b, L, r), "cpu")
    except Exception as e:                                  # an error is a finding too
        fails.append((tag, "exception", repr(e)[:200]))
        print("FAIL", tag, repr(e)[:200], flush=True)
        n += 1
        continue
    errs = {"out": max(rel(a, b) for a, b in zip(outs, ro)), "dfmap1": rel(g1, o1), "dfmap2": rel(g2, o2)}
    bar = 2e-5 if block.startswith("on_the_fly") else 1e-5
    for k, v in errs.items():
        worst[(block, k)] = max(worst.get((block, k), 0.0), v)
        if not (v <= bar):
            fails.append((tag, k, v))
            print("FAIL", tag, k, v, flush=True)
    n += 1
    if n % 50 == 0:
        print(f"{n} cases, {len(fails)} failures", flush=True)
print(f"{n} cases, {len(fails)} failures")
for k in sorted(worst):
    print(f"  worst {k[0]:22s} {k[1]:7s} {worst[k]:.2e}")
sys.exit(1 if fails else 0)
