#!/usr/bin/env python3
"""Diagnostic (not product code): the train-step gradient comparison of tests/test_gpu_harness.py, ours twice."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from harness.raft_model import RAFT, sequence_loss, synthetic_pair
from oracle import corr_oracle as O
from raft_amd import CorrBlock
dev = torch.device("cuda:0")

def grads(factory):
    torch.manual_seed(0)
    model = RAFT(factory).to(dev).train()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    i1, i2, gt = synthetic_pair(2, 136, 176, seed=3, device=dev)
    preds = model(i1, i2, iters=4)
    loss, _ = sequence_loss(preds, gt, torch.ones_like(gt[:, 0]))
    loss.backward()
    return float(loss.detach()), {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}

l_ref, g_ref = grads(lambda a, b: O.CorrBlockOracle(a, b, 4, 4))
which = sys.argv[1] if len(sys.argv) > 1 else "ours"
if which == "ours":
    runs = [grads(lambda a, b: CorrBlock(a, b, num_levels=4, radius=4)) for _ in range(3)]
else:                                                        # the reference against itself, more often
    runs = [grads(lambda a, b: O.CorrBlockOracle(a, b, 4, 4)) for _ in range(4)]
gmax = max(np.abs(g).max() for g in g_ref.values())
for i, (l, g) in enumerate(runs):
    worst = max((float(np.abs(g_ref[k] - g[k]).max() / (np.abs(g_ref[k]).max() + 1e-2 * gmax)), k) for k in g_ref)
    same = all(np.array_equal(g[k], runs[0][1][k]) for k in g)
    print(f"run {i}: loss diff {abs(l - l_ref):.2e}, worst {worst[0]:.2e} at {worst[1]}, equal to run 0: {same}", flush=True)
