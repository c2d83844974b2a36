#!/usr/bin/env python3
"""Where the gradient tolerances of tests/ come from: the error of the REFERENCE's own fp32
backward pass (stock PyTorch CPU ops in the reference's order = oracle.CorrBlockOracle) against the same
computation in float64, next to this library's error against float64.  Metric: max|a - b| / max|b| (tests/conftest.py).
    python tools/grad_tolerance.py  ->  one JSON line (profiles/r02_grad_tolerance.json)"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import corr_oracle as O                     # checker
from raft_amd import AlternateCorrBlock, CorrBlock


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


dev = torch.device("cuda:0")
res = {}
for name, (B, D, H, W, nl) in {"D64_24x40_3lookups": (1, 64, 24, 40, 3), "D256_47x78_12lookups": (1, 256, 47, 78, 12)}.items():
    L, r = 4, 4
    rs = np.random.RandomState(3)
    f1n, f2n = rs.randn(B, D, H, W).astype(np.float32), rs.randn(B, D, H, W).astype(np.float32)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    cons = [np.stack([xx, yy], 0)[None].astype(np.float32) + (1 + k) * rs.randn(B, 2, H, W).astype(np.float32) for k in range(nl)]
    gn = rs.randn(B, L * 81, H, W).astype(np.float32)

    def run(make, device, dtype, need_co):
        f1 = torch.from_numpy(f1n).to(device=device, dtype=dtype).requires_grad_(True)
        f2 = torch.from_numpy(f2n).to(device=device, dtype=dtype).requires_grad_(True)
        cs = [torch.from_numpy(c).to(device=device, dtype=dtype).requires_grad_(need_co) for c in cons]
        blk = make(f1, f2)
        torch.autograd.backward([blk(c) for c in cs], [torch.from_numpy(gn).to(device=device, dtype=dtype)] * nl)
        return (f1.grad.double().cpu().numpy(), f2.grad.double().cpu().numpy(),
                cs[0].grad.double().cpu().numpy() if need_co else None)

    g64 = run(lambda a, b: O.CorrBlockOracle(a, b, L, r), "cpu", torch.float64, True)
    g32 = run(lambda a, b: O.CorrBlockOracle(a, b, L, r), "cpu", torch.float32, True)
    ours = run(lambda a, b: CorrBlock(a, b, num_levels=L, radius=r), dev, torch.float32, True)
    alt = run(lambda a, b: AlternateCorrBlock(a, b, num_levels=L, radius=r), dev, torch.float32, False)
    res[name] = {
        "reference_fp32_vs_fp64": {"dfmap1": rel(g32[0], g64[0]), "dfmap2": rel(g32[1], g64[1]), "dcoords": rel(g32[2], g64[2])},
        "CorrBlock_vs_fp64": {"dfmap1": rel(ours[0], g64[0]), "dfmap2": rel(ours[1], g64[1]), "dcoords": rel(ours[2], g64[2])},
        "AlternateCorrBlock_vs_fp64": {"dfmap1": rel(alt[0], g64[0]), "dfmap2": rel(alt[1], g64[1])},
        "CorrBlock_vs_reference_fp32": {"dfmap1": rel(ours[0], g32[0]), "dfmap2": rel(ours[1], g32[1]), "dcoords": rel(ours[2], g32[2])},
    }
print(json.dumps(res))
