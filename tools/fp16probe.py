import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from raft_amd import CorrBlock
dev = torch.device("cuda", 0)
f1, f2, coords = bench.make_inputs(dev, 1234)
with torch.no_grad():
    for _ in range(20): bench.one_pair(f1, f2, coords)
    for prec in ("fp16x2", "bf16x3", "bf16", "fp32", "fp16x2"):
        for reps in (20, 200):
            t = bench.ev_time_us(lambda: CorrBlock(f1, f2, num_levels=4, radius=4, precision=prec), reps=reps)
            print(prec, reps, round(t, 1))
