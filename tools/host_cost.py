"""Host time to enqueue one bench step (build + 32 lookups) vs the GPU time it takes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raft_amd import CorrBlock

dev = torch.device("cuda", 0)
f1, f2, coords = bench.make_inputs(dev, 1234)
with torch.no_grad():
    for _ in range(20):
        bench.one_pair(f1, f2, coords)
    torch.cuda.synchronize()
    host = []
    for _ in range(50):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cb = CorrBlock(f1, f2, num_levels=bench.LEVELS, radius=bench.RADIUS)
        t1 = time.perf_counter()
        for c in coords:
            cb(c)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        host.append((t1 - t0, (t2 - t1) / len(coords), t3 - t0))
    host.sort(key=lambda x: x[2])
    m = host[len(host) // 2]
    print(f"host: ctor {m[0]*1e6:.1f} us, per lookup {m[1]*1e6:.2f} us, step wall {m[2]*1e6:.1f} us")

# the same step captured once into a HIP graph and replayed
with torch.no_grad():
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            bench.one_pair(f1, f2, coords)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = bench.one_pair(f1, f2, coords)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    print(f"graph replay: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per step")
    t0 = time.perf_counter()
    for _ in range(200):
        bench.one_pair(f1, f2, coords)
    torch.cuda.synchronize()
    print(f"stream launches: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per step")
    for nev in (3, 2, 1, 0):
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(200)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(200):
            if nev > 0: ev[s][0].record()
            cb = CorrBlock(f1, f2, num_levels=bench.LEVELS, radius=bench.RADIUS)
            if nev > 1: ev[s][1].record()
            for c in coords:
                out = cb(c)
            if nev > 2: ev[s][2].record()
        torch.cuda.synchronize()
        print(f"stream launches, {nev} event records per step: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per step")
