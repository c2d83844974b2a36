#!/usr/bin/env python3
"""Experiment harness (not product code): loops one build for a few seconds while sampling rocm-smi's power and clock
read-outs, to see whether the build runs power-limited.  usage: power_probe.py [cfg] [volume_dtype] [seconds]"""
import os, subprocess, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raft_amd import CorrBlock
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "1080p"
vol = sys.argv[2] if len(sys.argv) > 2 else "bf16"
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
dev = torch.device("cuda:0")
f1, f2, coords, _ = bench.make_inputs(torch, cfg, dev, 1)
samples, stop = [], False
def sample():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showuse", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append((time.time(), o.strip()))
        except Exception as e:                      # noqa: BLE001
            samples.append((time.time(), "ERR " + repr(e)))
        time.sleep(0.3)
print("idle:", subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--csv"], capture_output=True, text=True).stdout, flush=True)
th = threading.Thread(target=sample); th.start()
t0 = time.time(); n = 0
with torch.no_grad():
    while time.time() - t0 < secs:
        for _ in range(50):
            cb = CorrBlock(f1, f2, num_levels=4, radius=4, precision="bf16", volume_dtype=vol)
            n += 1
        torch.cuda.synchronize()
el = time.time() - t0
stop = True; th.join()
print(f"{cfg} bf16->{vol}: {n} builds in {el:.2f}s = {el / n * 1e6:.0f} us/build (incl. host)", flush=True)
for t, o in samples:
    print(f"t={t - t0:.1f}s", o.replace("\n", " | "), flush=True)
