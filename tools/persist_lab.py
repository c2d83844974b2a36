#!/usr/bin/env python3
"""Experiment harness (not product code): the persistent 16-bit build against the per-patch kernel (RC_PERSIST=0/1 in a
lab build of the library): bit-equality of the pyramids and build times.  usage: persist_lab.py tools/libraftcorr_lab.so"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from raft_amd import CorrBlock
import bench
dev = torch.device("cuda:0")
cfgs = sys.argv[2:] or ["1080p", "sintel", "things_train"]
for cfg in cfgs:
    f1, f2, coords, _ = bench.make_inputs(torch, cfg, dev, 1)
    for vol in ("bf16", "fp32"):
        res, pyr = {}, {}
        for mode in ("0", "1"):
            os.environ["RC_PERSIST"] = mode
            for rl in (("32",) if mode == "0" else ("32", "16", "8", "4")):
                os.environ["RC_RUN_LEN"] = rl
                with torch.no_grad():
                    cb = CorrBlock(f1, f2, num_levels=4, radius=4, precision="bf16", volume_dtype=vol)
                    key = mode + "/" + rl
                    pyr[key] = [t.clone() for t in cb._pyr.slabs(cb._pyr.flat)] if cfg != "1080p" or vol == "bf16" else None
                    out = cb(coords[3]).clone()
                    del cb
                    res[key] = round(bench.ev_time_us(torch, lambda: CorrBlock(f1, f2, num_levels=4, radius=4, precision="bf16", volume_dtype=vol),
                                                      reps=8, warm=3), 1)
                    if mode == "0":
                        ref_out, ref_pyr = out, pyr[key]
                    else:
                        same = torch.equal(out, ref_out) and (pyr[key] is None or all(torch.equal(x, y) for x, y in list(zip(pyr[key], ref_pyr))[:3]))
                        res[key] = (res[key], "same bits" if same else "DIFFERENT")
                    pyr[key] = None
                torch.cuda.empty_cache()
        print(cfg, "bf16 ->", vol, "volume: build us by RC_PERSIST/RC_RUN_LEN:", res, flush=True)
