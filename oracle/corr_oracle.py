"""CPU oracle for RAFT's correlation hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, what the reference computes on the path that
``raft_amd`` replaces.  It is the checker for the HIP kernels and the
``cpu_baseline`` of ``bench.py``; the product (``raft_amd/``) never imports it.

Two independent restatements live here:

* ``*_torch`` functions run the *same stock PyTorch CPU operators in the same
  order* as the reference (``matmul`` -> ``/sqrt(D)`` -> ``avg_pool2d`` ->
  normalise -> ``grid_sample`` -> ``cat``).  On the same machine they are
  bit-identical to an import of the reference; they are what the golden vectors
  under ``tests/golden/`` (generated from the real reference by
  ``tests/golden/make_golden.py``) pin.
* ``*_direct`` functions are a from-first-principles numpy/float64 version
  (explicit dot products, explicit 2x2 means, explicit 4-tap bilinear with
  zero padding in *pixel* coordinates).  They share no operator with the
  reference and are used for known-answer checks and as the high-precision
  truth when a tolerance has to be judged.

Parity pinning: the reference has no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned by outputs of the reference
itself, imported on CPU in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).  The on-the-fly
("alternate") CUDA kernels cannot be built here (needs nvcc + CUDA device);
for them the oracle is pinned through the mathematical identity
"alt == canonical all-pairs lookup" against the same golden vectors, plus
``oracle/alt_corr_oracle.c`` which restates the CUDA kernel's arithmetic.

Reference citations (paths under /root/reference):
  core/corr.py:12-61     CorrBlock (build, pyramid, lookup)
  core/corr.py:64-92     AlternateCorrBlock
  core/utils/utils.py:57-71  bilinear_sampler
  alt_cuda_corr/correlation_kernel.cu:19-119   corr_forward_kernel
  alt_cuda_corr/correlation_kernel.cu:123-256  corr_backward_kernel
"""
from __future__ import annotations

import math
import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# torch restatement (same operators as the reference)
# --------------------------------------------------------------------------

def all_pairs_corr_torch(fmap1: torch.Tensor, fmap2: torch.Tensor) -> torch.Tensor:
    """corr.py:53-61 -- (B,D,H,W)x2 -> (B,H,W,1,H,W), matmul then / sqrt(D)."""
    batch, dim, ht, wd = fmap1.shape
    f1 = fmap1.view(batch, dim, ht * wd)
    f2 = fmap2.view(batch, dim, ht * wd)
    corr = torch.matmul(f1.transpose(1, 2), f2)
    corr = corr.view(batch, ht, wd, 1, ht, wd)
    return corr / torch.sqrt(torch.tensor(dim).float())


def build_pyramid_torch(fmap1, fmap2, num_levels):
    """corr.py:13-27 -- list of (B*H*W, 1, H_l, W_l); avg_pool2d(2, 2) chain."""
    corr = all_pairs_corr_torch(fmap1, fmap2)
    batch, h1, w1, dim, h2, w2 = corr.shape
    corr = corr.reshape(batch * h1 * w1, dim, h2, w2)
    pyramid = [corr]
    for _ in range(num_levels - 1):
        corr = F.avg_pool2d(corr, 2, stride=2)
        pyramid.append(corr)
    return pyramid


def bilinear_sampler_torch(img, coords):
    """utils/utils.py:57-65 -- pixel coords -> [-1,1] -> grid_sample(align_corners=True)."""
    H, W = img.shape[-2:]
    xgrid, ygrid = coords.split([1, 1], dim=-1)
    xgrid = 2 * xgrid / (W - 1) - 1
    ygrid = 2 * ygrid / (H - 1) - 1
    grid = torch.cat([xgrid, ygrid], dim=-1)
    return F.grid_sample(img, grid, align_corners=True)


def _delta(r, device):
    """corr.py:38-40 -- delta[a, b] = (a - r, b - r); it is ADDED to (x, y), so the slow
    window index a is the x offset and the fast index b is the y offset."""
    d = torch.linspace(-r, r, 2 * r + 1, device=device)
    return torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1)


def lookup_fork_torch(pyramid, coords, radius):
    """corr.py:29-51 -- coords (B, N, 2) in level-0 pixel units, (x, y); the SAME coords at
    every level (no /2**i); output (B, N, L*(2r+1)^2) float32."""
    r = radius
    batch, hw, _ = coords.shape
    flat = coords.flatten(0, 1)
    out = []
    for corr in pyramid:
        delta = _delta(r, coords.device).view(1, 2 * r + 1, 2 * r + 1, 2)
        coords_lvl = flat[:, None, None] + delta
        s = bilinear_sampler_torch(corr, coords_lvl)
        out.append(s.view(batch, hw, -1))
    return torch.cat(out, dim=-1).contiguous().float()


def lookup_canonical_torch(pyramid, coords, radius):
    """Canonical RAFT lookup that raft.py:125 / update.py:82-91 / AlternateCorrBlock
    (corr.py:75-92) expect: coords (B, 2, H, W) pixel units -> permute(0,2,3,1), per level
    coords / 2**i, output (B, L*(2r+1)^2, H, W).  Assembled from the reference's own
    primitives (pyramid + bilinear_sampler), as SURVEY.md section 8(c) C1 prescribes."""
    r = radius
    c = coords.permute(0, 2, 3, 1)
    B, H, W, _ = c.shape
    out = []
    for i, corr in enumerate(pyramid):
        delta = _delta(r, coords.device).view(1, 2 * r + 1, 2 * r + 1, 2)
        centroid = c.reshape(B * H * W, 1, 1, 2) / 2 ** i
        s = bilinear_sampler_torch(corr, centroid + delta)
        out.append(s.view(B, H, W, -1))
    out = torch.cat(out, dim=-1)
    return out.permute(0, 3, 1, 2).contiguous().float()


class CorrBlockOracle:
    """Drop-in restatement of the reference classes' call surface, for timing and parity."""

    def __init__(self, fmap1, fmap2, num_levels=2, radius=4):
        self.num_levels = num_levels
        self.radius = radius
        self.corr_pyramid = build_pyramid_torch(fmap1, fmap2, num_levels)

    def __call__(self, coords):
        if coords.dim() == 3:
            return lookup_fork_torch(self.corr_pyramid, coords, self.radius)
        return lookup_canonical_torch(self.corr_pyramid, coords, self.radius)


def alt_feature_pyramid_torch(fmap2, num_levels):
    """corr.py:69-73 -- avg_pool2d(2,2) chain of the target features (levels 0..L-1 used)."""
    pyr = [fmap2]
    for _ in range(num_levels - 1):
        fmap2 = F.avg_pool2d(fmap2, 2, stride=2)
        pyr.append(fmap2)
    return pyr


# --------------------------------------------------------------------------
# first-principles numpy restatement (float64 by default)
# --------------------------------------------------------------------------

def build_pyramid_direct(fmap1, fmap2, num_levels, dtype=np.float64):
    """vol[b,q,t] = <f1[b,:,q], f2[b,:,t]> / sqrt(D); level l = 2x2 mean of level l-1 over the
    TARGET dims with floor truncation (odd trailing row/col dropped).  Returns a list of
    arrays shaped (B, N, H_l, W_l)."""
    f1 = np.asarray(fmap1, dtype=dtype)
    f2 = np.asarray(fmap2, dtype=dtype)
    B, D, H, W = f1.shape
    vol = np.einsum("bdq,bdt->bqt", f1.reshape(B, D, H * W), f2.reshape(B, D, H * W))
    vol = (vol / math.sqrt(D)).reshape(B, H * W, H, W)
    pyr = [vol]
    for _ in range(num_levels - 1):
        v = pyr[-1]
        h2, w2 = v.shape[2] // 2, v.shape[3] // 2
        v = v[:, :, : 2 * h2, : 2 * w2]
        v = 0.25 * (v[:, :, 0::2, 0::2] + v[:, :, 0::2, 1::2] + v[:, :, 1::2, 0::2] + v[:, :, 1::2, 1::2])
        pyr.append(v)
    return pyr


def _sample_zero_pad(slab, x, y):
    """4-tap bilinear of slab (..., H, W) at pixel coords x, y (same leading shape + window
    dims), taps outside [0,W-1]x[0,H-1] contribute zero.  slab: (M, H, W); x, y: (M, K)."""
    M, H, W = slab.shape
    x0 = np.floor(x)
    y0 = np.floor(y)
    fx = x - x0
    fy = y - y0
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    out = np.zeros(x.shape, dtype=slab.dtype)
    rows = np.arange(M)[:, None]
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xi = x0 + dx
            yi = y0 + dy
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            v = slab[rows, np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)]
            out += np.where(ok, v, 0.0) * wy * wx
    return out


def lookup_direct(pyr, coords_xy, radius, per_level_scale):
    """pyr: list of (B, N, H_l, W_l); coords_xy: (B, N, 2) pixel (x, y).  Returns
    (B, N, L*rd*rd) with channel = l*rd*rd + a*rd + b, sample = (x + a - r, y + b - r)
    (a = slow = x offset; corr.py:38-44).  per_level_scale divides coords by 2**l
    (corr.py:86 / upstream RAFT); the fork's CorrBlock does not (corr.py:42)."""
    r = radius
    rd = 2 * r + 1
    B, N, _ = coords_xy.shape
    a = np.arange(rd) - r
    offx = np.repeat(a, rd)[None, :]      # slow index -> x
    offy = np.tile(a, rd)[None, :]        # fast index -> y
    outs = []
    for l, v in enumerate(pyr):
        s = (2.0 ** l) if per_level_scale else 1.0
        c = np.asarray(coords_xy, dtype=v.dtype).reshape(B * N, 2) / s
        x = c[:, 0:1] + offx
        y = c[:, 1:2] + offy
        o = _sample_zero_pad(v.reshape(B * N, v.shape[2], v.shape[3]), x, y)
        outs.append(o.reshape(B, N, rd * rd))
    return np.concatenate(outs, axis=-1)


def alt_forward_level_direct(fmap1_nhwc, fmap2_nhwc, coords, radius, dtype=np.float64):
    """Restates corr_forward_kernel (correlation_kernel.cu:59-116) for one level.
    fmap1 (B,H1,W1,C), fmap2 (B,H2,W2,C), coords (B,Nc,H1,W1,2) pixel (x,y) of THIS level.
    For every integer tap (floor(y)-r+iy, floor(x)-r+ix), iy,ix in [0, rd]: s = <f1, f2[tap]>
    (0 outside fmap2), splat s*{dy*dx, dy*(1-dx), (1-dy)*dx, (1-dy)*(1-dx)} into cells
    (iy-1,ix-1), (iy-1,ix), (iy,ix-1), (iy,ix); output channel = iy' + rd*ix'.
    Returns (B, Nc, rd*rd, H1, W1), NOT scaled by 1/sqrt(C) (corr.py:92 does that)."""
    f1 = np.asarray(fmap1_nhwc, dtype=dtype)
    f2 = np.asarray(fmap2_nhwc, dtype=dtype)
    co = np.asarray(coords, dtype=dtype)
    B, H1, W1, C = f1.shape
    _, H2, W2, _ = f2.shape
    Nc = co.shape[1]
    r = radius
    rd = 2 * r + 1
    out = np.zeros((B, Nc, rd * rd, H1, W1), dtype=dtype)
    for b in range(B):
        for n in range(Nc):
            x = co[b, n, :, :, 0]
            y = co[b, n, :, :, 1]
            x0 = np.floor(x)
            y0 = np.floor(y)
            dx = x - x0
            dy = y - y0
            x0 = x0.astype(np.int64)
            y0 = y0.astype(np.int64)
            for iy in range(rd + 1):
                for ix in range(rd + 1):
                    h2 = y0 - r + iy
                    w2 = x0 - r + ix
                    ok = (h2 >= 0) & (h2 < H2) & (w2 >= 0) & (w2 < W2)
                    g = f2[b, np.clip(h2, 0, H2 - 1), np.clip(w2, 0, W2 - 1)]   # (H1,W1,C)
                    s = np.where(ok, np.einsum("hwc,hwc->hw", f1[b], g), 0.0)
                    if iy > 0 and ix > 0:
                        out[b, n, (iy - 1) + rd * (ix - 1)] += s * dy * dx
                    if iy > 0 and ix < rd:
                        out[b, n, (iy - 1) + rd * ix] += s * dy * (1 - dx)
                    if iy < rd and ix > 0:
                        out[b, n, iy + rd * (ix - 1)] += s * (1 - dy) * dx
                    if iy < rd and ix < rd:
                        out[b, n, iy + rd * ix] += s * (1 - dy) * (1 - dx)
    return out


def alt_backward_level_direct(fmap1_nhwc, fmap2_nhwc, coords, corr_grad, radius, dtype=np.float64):
    """Restates corr_backward_kernel (correlation_kernel.cu:177-254) for one level: per tap
    g = sum(<=4 cells) corr_grad*weight; fmap1_grad += g*f2[tap]; fmap2_grad[tap] += g*f1;
    coords_grad is returned as zeros (the reference allocates it and never writes it,
    correlation_kernel.cu:307)."""
    f1 = np.asarray(fmap1_nhwc, dtype=dtype)
    f2 = np.asarray(fmap2_nhwc, dtype=dtype)
    co = np.asarray(coords, dtype=dtype)
    cg = np.asarray(corr_grad, dtype=dtype)
    B, H1, W1, C = f1.shape
    _, H2, W2, _ = f2.shape
    Nc = co.shape[1]
    r = radius
    rd = 2 * r + 1
    g1 = np.zeros_like(f1)
    g2 = np.zeros_like(f2)
    hh, ww = np.meshgrid(np.arange(H1), np.arange(W1), indexing="ij")
    for b in range(B):
        for n in range(Nc):
            x = co[b, n, :, :, 0]
            y = co[b, n, :, :, 1]
            x0 = np.floor(x)
            y0 = np.floor(y)
            dx = x - x0
            dy = y - y0
            x0 = x0.astype(np.int64)
            y0 = y0.astype(np.int64)
            for iy in range(rd + 1):
                for ix in range(rd + 1):
                    g = np.zeros((H1, W1), dtype=dtype)
                    if iy > 0 and ix > 0:
                        g += cg[b, n, (iy - 1) + rd * (ix - 1)] * dy * dx
                    if iy > 0 and ix < rd:
                        g += cg[b, n, (iy - 1) + rd * ix] * dy * (1 - dx)
                    if iy < rd and ix > 0:
                        g += cg[b, n, iy + rd * (ix - 1)] * (1 - dy) * dx
                    if iy < rd and ix < rd:
                        g += cg[b, n, iy + rd * ix] * (1 - dy) * (1 - dx)
                    h2 = y0 - r + iy
                    w2 = x0 - r + ix
                    ok = (h2 >= 0) & (h2 < H2) & (w2 >= 0) & (w2 < W2)
                    g = np.where(ok, g, 0.0)
                    h2c = np.clip(h2, 0, H2 - 1)
                    w2c = np.clip(w2, 0, W2 - 1)
                    g1[b] += g[:, :, None] * f2[b, h2c, w2c]
                    np.add.at(g2[b], (h2c[ok], w2c[ok]), (g[:, :, None] * f1[b])[ok])
    return g1, g2, np.zeros_like(co)


def alt_lookup_direct(fmap1, fmap2, coords, num_levels, radius, dtype=np.float64):
    """corr.py:64-92 end to end: fmap (B,D,H,W), coords (B,2,H,W) -> (B, L*rd*rd, H, W)."""
    f1 = torch.as_tensor(np.asarray(fmap1, dtype=dtype))
    f2 = torch.as_tensor(np.asarray(fmap2, dtype=dtype))
    co = np.asarray(coords, dtype=dtype).transpose(0, 2, 3, 1)       # corr.py:76
    B, H, W, _ = co.shape
    D = f1.shape[1]
    pyr = alt_feature_pyramid_torch(f2, num_levels)                   # corr.py:69-73
    f1n = f1.permute(0, 2, 3, 1).contiguous().numpy()
    outs = []
    for i in range(num_levels):
        f2n = pyr[i].permute(0, 2, 3, 1).contiguous().numpy()
        ci = (co / 2 ** i).reshape(B, 1, H, W, 2)                     # corr.py:86
        outs.append(alt_forward_level_direct(f1n, f2n, ci, radius, dtype)[:, 0])
    corr = np.stack(outs, axis=1).reshape(B, -1, H, W)               # corr.py:90-91
    return corr / math.sqrt(D)                                       # corr.py:92
