#!/usr/bin/env python3
"""End-to-end RAFT-basic inference (harness/raft_model.py, seed-0 random weights, synthetic frame
pair at Sintel 436x1024) with each correlation block: frame pairs/s for the whole model, the time
the correlation path takes inside it, and the flow difference to the stock-operator block.
    python tools/raft_e2e.py [iters]          -> one JSON line (profiles/r01_raft_e2e.json)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from harness.raft_model import RAFT, InputPadder, synthetic_pair   # noqa: E402
from harness.stock_corr import StockCorrBlock                       # noqa: E402
from raft_amd import AlternateCorrBlock, CorrBlock                  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
i1, i2, _ = synthetic_pair(1, 436, 1024, seed=0, device=dev)
pad = InputPadder(i1.shape)
a, b = pad.pad(i1, i2)

factories = {
    "stock_ops_gpu": lambda x, y: StockCorrBlock(x, y, 4, 4),
    "all_pairs_fp32": lambda x, y: CorrBlock(x, y, num_levels=4, radius=4),
    "all_pairs_bf16": lambda x, y: CorrBlock(x, y, num_levels=4, radius=4, precision="bf16"),
    "all_pairs_fp16x2": lambda x, y: CorrBlock(x, y, num_levels=4, radius=4, precision="fp16x2"),
    "all_pairs_fp16x2_fused_convc1": lambda x, y: CorrBlock(x, y, num_levels=4, radius=4, precision="fp16x2"),
    "on_the_fly": lambda x, y: AlternateCorrBlock(x, y, num_levels=4, radius=4),
}


class Timed:
    """Wraps a correlation block: accumulates the GPU time of its constructor and calls."""

    def __init__(self, factory):
        self.factory, self.spans = factory, []

    def _span(self, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.spans.append((e0, e1))
        return out

    def __call__(self, x, y):
        blk = self._span(lambda: self.factory(x, y))
        timed = self

        class _Blk:
            def __call__(self, c):
                return timed._span(lambda: blk(c))

            def lookup_conv1x1(self, *a, **k):
                return timed._span(lambda: blk.lookup_conv1x1(*a, **k))
        return _Blk()

    def total_ms(self):
        torch.cuda.synchronize()
        t = sum(e0.elapsed_time(e1) for e0, e1 in self.spans)
        self.spans = []
        return t


res, flows = {"iters": iters, "image": "436x1024 (padded 440x1024)", "weights": "random, seed 0"}, {}
for name, fac in factories.items():
    torch.manual_seed(0)
    timed = Timed(fac)
    model = RAFT(timed, fuse_convc1=name.endswith("fused_convc1")).to(dev).eval()
    with torch.no_grad():
        for _ in range(3):
            model(a, b, iters=iters, test_mode=True)
        timed.total_ms()
        torch.cuda.synchronize()
        n, t0 = 10, time.perf_counter()
        for _ in range(n):
            _, up = model(a, b, iters=iters, test_mode=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    flows[name] = pad.unpad(up).cpu().numpy()
    res[name] = {"pairs_per_s": n / el, "ms_per_pair": el / n * 1e3, "corr_ms_per_pair": timed.total_ms() / n}

# The same model with the whole forward pass captured once into a HIP graph and replayed: no host
# launch cost, so the correlation path's share of the GPU time is what is left to see.  (The
# entry points of libraftcorr.so launch on the caller's stream, allocate nothing and never
# synchronise, so they are capturable like the stock operators.)
for name in ("stock_ops_gpu", "all_pairs_fp32", "all_pairs_fp16x2_fused_convc1"):
    torch.manual_seed(0)
    model = RAFT(factories[name], fuse_convc1=name.endswith("fused_convc1")).to(dev).eval()
    with torch.no_grad():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                model(a, b, iters=iters, test_mode=True)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            _, up = model(a, b, iters=iters, test_mode=True)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        n, t0 = 20, time.perf_counter()
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    flows[name + "_graph"] = pad.unpad(up).cpu().numpy()
    res[name + "_graph"] = {"pairs_per_s": n / el, "ms_per_pair": el / n * 1e3,
                            "note": "whole forward pass replayed as one HIP graph"}
    del g
ref = flows["stock_ops_gpu"]
for name in flows:
    res[name]["epe_vs_stock_ops_px"] = float(np.sqrt(((flows[name] - ref) ** 2).sum(1)).mean())
print(json.dumps(res))
