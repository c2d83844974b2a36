"""Seeded inputs of the full-size fixtures K1 / T1 / P1 (numpy's frozen legacy generator), shared by
the generator (make_golden.py, which imports the reference and runs only in the build container) and
by the GPU tests, which regenerate the inputs instead of storing them.  No reference code here."""
import numpy as np

D, L, R = 256, 4, 4


def _grid(B, H, W):
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    return np.stack([xx, yy], 0)[None].repeat(B, 0).astype(np.float32)


def k1_inputs():
    """BASELINE.json configs[2] shape: KITTI 375x1242 -> 47x156, B = 1; canonical coords = grid + N(0, 4^2)."""
    g = np.random.RandomState(4701)
    B, H, W = 1, 47, 156
    f1 = g.randn(B, D, H, W).astype(np.float32)
    f2 = g.randn(B, D, H, W).astype(np.float32)
    co = (_grid(B, H, W) + 4.0 * g.randn(B, 2, H, W)).astype(np.float32)
    gout = g.randn(B, L * (2 * R + 1) ** 2, H, W).astype(np.float32)
    return f1, f2, co, gout


def t1_inputs(iters=12):
    """BASELINE.json configs[3] shape: FlyingThings crops 400x720 -> 50x90, B = 6, 12 lookups whose
    coordinates drift like a flow estimate; loss = sum_k w_k <out_k, gout>."""
    g = np.random.RandomState(5090)
    B, H, W = 6, 50, 90
    f1 = g.randn(B, D, H, W).astype(np.float32)
    f2 = g.randn(B, D, H, W).astype(np.float32)
    base, drift, coords = _grid(B, H, W), np.zeros((B, 2, H, W), np.float32), []
    for _ in range(iters):
        drift = drift + (1.5 * g.randn(B, 2, H, W)).astype(np.float32)
        coords.append((base + drift).astype(np.float32))
    gout = g.randn(B, L * (2 * R + 1) ** 2, H, W).astype(np.float32)
    weights = [0.5 + 0.1 * k for k in range(iters)]
    return f1, f2, coords, gout, weights


def p1_inputs():
    """BASELINE.json configs[4] shape: 1088x1920 -> 136x240, B = 1; fractional canonical coords."""
    g = np.random.RandomState(1080)
    B, H, W = 1, 136, 240
    f1 = g.randn(B, D, H, W).astype(np.float32)
    f2 = g.randn(B, D, H, W).astype(np.float32)
    co = (_grid(B, H, W) + 6.0 * g.randn(B, 2, H, W)).astype(np.float32)
    queries = np.sort(g.choice(H * W, 128, replace=False))
    queries[:4] = [0, W - 1, H * W - W, H * W - 1]            # the four corners
    return f1, f2, co, np.sort(queries)
