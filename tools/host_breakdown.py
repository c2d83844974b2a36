"""Where the host time of one CorrBlock lookup goes (Python + ctypes + hipLaunchKernel)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raft_amd import CorrBlock, _lib, corr as C

dev = torch.device("cuda", 0)
f1, f2, coords = bench.make_inputs(dev, 1234)
lib = _lib.load()

def t(fn, n=20000):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    el = time.perf_counter() - t0
    torch.cuda.synchronize()
    return el / n * 1e6

with torch.no_grad():
    cb = CorrBlock(f1, f2, num_levels=4, radius=4)
    c = coords[3]
    p = cb._pyr
    shape = p.out_shape(4, _lib.MODE_CANONICAL)
    out = torch.empty(shape, dtype=torch.float32, device=dev)
    print(f"torch.empty            {t(lambda: torch.empty(shape, dtype=torch.float32, device=dev)):.2f} us")
    print(f"_stream                {t(lambda: C._stream(c)):.2f} us")
    print(f"_guard                 {t(lambda: C._guard(c)):.2f} us")
    print(f"_lib.load              {t(lambda: _lib.load()):.2f} us")
    print(f"shape checks           {t(lambda: (c.is_cuda, c.dim(), tuple(c.shape) == (p.B, 2, p.H, p.W), c.dtype is torch.float32, c.is_contiguous(), torch.is_grad_enabled())):.2f} us")
    st = C._stream(c)
    print(f"ctypes call, early return (radius -1) {t(lambda: lib.raftcorr_lookup_fwd(p.ptrs, p.B, p.H, p.W, p.L, -1, c.data_ptr(), 1, out.data_ptr(), st)):.2f} us")
    print(f"ctypes call + launch   {t(lambda: lib.raftcorr_lookup_fwd(p.ptrs, p.B, p.H, p.W, p.L, 4, c.data_ptr(), 1, out.data_ptr(), st), 3000):.2f} us (GPU-bound if > kernel time 7.4)")
    print(f"data_ptr x2            {t(lambda: (c.data_ptr(), out.data_ptr())):.2f} us")
    print(f"cb(c)                  {t(lambda: cb(c), 3000):.2f} us")
