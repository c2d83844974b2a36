import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from raft_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import bench
dev = torch.device('cuda:0')
for cfg in ("things_train", "kitti_alt", "sintel"):
    wl = bench.ThingsTrain(torch, cfg if cfg == "things_train" else "things_train", dev, 0, 1, None) if cfg == "things_train" else None
    if wl is None: continue
    for _ in range(5): wl.step(None)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ts = []
    for _ in range(20):
        wl.step(ev); torch.cuda.synchronize(); ts.append((ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])))
    import numpy as np
    print(cfg, "RC_KSPLIT_TARGET", os.environ.get("RC_KSPLIT_TARGET"), "fwd/bwd ms", np.median([t[0] for t in ts]), np.median([t[1] for t in ts]), flush=True)
