"""Context for the build's roofline fraction: the vendor library's fp32 GEMM on the same shape
(7040 x 7040 x 256, what torch.matmul in core/corr.py:59 runs), sustained and in short bursts."""
import torch, time
dev = torch.device("cuda", 0)
N, D = 7040, 256
torch.backends.cuda.matmul.allow_tf32 = False
a = torch.randn(1, N, D, device=dev)          # fmap1.view(B, D, N).transpose(1, 2)
b = torch.randn(1, D, N, device=dev)
out = torch.empty(1, N, N, device=dev)
for reps in (30, 3000):
    for _ in range(5):
        torch.matmul(a, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.matmul(a, b, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"rocBLAS/hipBLASLt sgemm {N}x{N}x{D}, {reps} back-to-back: {us:.1f} us = {2*N*N*D/us*1e-6:.1f} TF (writes level 0 only, row-major, unscaled)")
