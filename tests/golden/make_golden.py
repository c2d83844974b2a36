#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (damien911224/RAFT) on CPU.

Runs only in the build container, where the reference is mounted read-only at
/root/reference; it imports ``core/corr.py`` (CorrBlock) and ``core/utils/utils.py``
(bilinear_sampler) from there and records their outputs on seeded inputs.  The resulting
``tests/golden/*.npz`` files are data (inputs + expected outputs) and are committed; the
reference itself never travels to the GPU box.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
    python tests/golden/make_golden.py full       # only the full-size fixtures K1 / T1 / P1

Fixture ids follow SURVEY.md section 8(c):
  F1  fork-mode forward, L=4, r=4                      (CorrBlock as shipped, corr.py:29-51)
  F2  F1 + autograd gradients w.r.t. fmap1/fmap2/coords
  F3  fork-mode defaults as used by ours.py:370-377 (L=2, r=4, pixel-centre coords, odd map)
  F4  edge cases: integer / negative / out-of-range coords, r=3, D=128
  C1  canonical-mode forward ((B,2,H,W) coords, coords/2**l per level, (B,324,H,W) out),
      assembled from the reference's own corr pyramid + bilinear_sampler
  C2  C1 + autograd gradients
  S1  Sintel-shaped (D=256, 55x128, L=4, r=4) checksums + sampled rows, both modes
  K1  KITTI shape (47x156, D=256, L=4, r=4), canonical mode, fractional coords: sampled output
      rows + gradients (sampled entries + checksums) for a fixed upstream gradient
  T1  FlyingThings shape (B=6, 50x90), 12 lookups, weighted sum -> one backward pass
  P1  1080p shape (136x240): 128 sampled queries -- their volume rows built with torch.matmul,
      pooled with avg_pool2d, sampled with the reference's bilinear_sampler (queries are independent)
Every fixture keeps all level sizes >= 2: the reference divides by (H_l - 1) / (W_l - 1)
(utils/utils.py:61-62) and returns NaN for a 1-pixel level.
Inputs come from numpy's frozen legacy generator (RandomState), so tests can regenerate the
large S1 inputs instead of storing them.
"""
import os
import sys
import warnings

import numpy as np
import torch

REF = "/root/reference/core"
HERE = os.path.dirname(os.path.abspath(__file__))

warnings.filterwarnings("ignore")
sys.path.insert(0, REF)
from corr import CorrBlock                      # noqa: E402  (the reference)
from utils.utils import bilinear_sampler        # noqa: E402  (the reference)


def rs(seed):
    return np.random.RandomState(seed)


def t(a, grad=False):
    x = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return x.requires_grad_(grad)


def canonical_lookup_ref(cb, coords, radius):
    """(B,2,H,W) coords -> (B, L*rd^2, H, W) with the reference's primitives: per level
    bilinear_sampler(cb.corr_pyramid[l], coords/2**l + delta), delta as corr.py:38-40."""
    r = radius
    c = coords.permute(0, 2, 3, 1)
    B, H, W, _ = c.shape
    out = []
    for i in range(cb.num_levels):
        corr = cb.corr_pyramid[i]
        dx = torch.linspace(-r, r, 2 * r + 1)
        dy = torch.linspace(-r, r, 2 * r + 1)
        delta = torch.stack(torch.meshgrid(dy, dx), axis=-1)
        centroid = c.reshape(B * H * W, 1, 1, 2) / 2 ** i
        s = bilinear_sampler(corr, centroid + delta.view(1, 2 * r + 1, 2 * r + 1, 2))
        out.append(s.view(B, H, W, -1))
    out = torch.cat(out, dim=-1)
    return out.permute(0, 3, 1, 2).contiguous().float()


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in kw.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def fork_case(seed, B, D, H, W, L, r, coords=None, grad=False):
    g = rs(seed)
    f1 = t(g.randn(B, D, H, W), grad)
    f2 = t(g.randn(B, D, H, W), grad)
    if coords is None:
        coords = g.rand(B, H * W, 2) * np.array([W - 1, H - 1])
    co = t(coords, grad)
    cb = CorrBlock(f1, f2, num_levels=L, radius=r)
    out = cb(co)
    res = dict(fmap1=f1, fmap2=f2, coords=co, out=out, num_levels=L, radius=r)
    if grad:
        gout = t(g.randn(*out.shape))
        out.backward(gout)
        res.update(gout=gout, dfmap1=f1.grad, dfmap2=f2.grad, dcoords=co.grad)
    return res


def canon_case(seed, B, D, H, W, L, r, grad=False, sigma=2.5):
    g = rs(seed)
    f1 = t(g.randn(B, D, H, W), grad)
    f2 = t(g.randn(B, D, H, W), grad)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    base = np.stack([xx, yy], 0)[None].repeat(B, 0).astype(np.float32)
    co = t(base + sigma * g.randn(B, 2, H, W), grad)
    cb = CorrBlock(f1, f2, num_levels=L, radius=r)
    out = canonical_lookup_ref(cb, co, r)
    res = dict(fmap1=f1, fmap2=f2, coords=co, out=out, num_levels=L, radius=r)
    if grad:
        gout = t(g.randn(*out.shape))
        out.backward(gout)
        res.update(gout=gout, dfmap1=f1.grad, dfmap2=f2.grad, dcoords=co.grad)
    return res


def sampled(a, idx):
    return a.reshape(-1)[idx]


def full_size_cases():
    """K1 / T1 / P1: the BASELINE.json shapes that the small fixtures do not reach.  Inputs are
    regenerated by the tests (tests/golden/inputs.py); only sampled outputs are stored."""
    import torch.nn.functional as F
    sys.path.insert(0, HERE)
    import inputs as I
    torch.set_num_threads(8)
    L, r = I.L, I.R
    pick = rs(77)
    # ---- K1
    f1n, f2n, con, goutn = I.k1_inputs()
    f1, f2, co = t(f1n, True), t(f2n, True), t(con, True)
    cb = CorrBlock(f1, f2, num_levels=L, radius=r)
    out = canonical_lookup_ref(cb, co, r)
    out.backward(t(goutn))
    B, _, H, W = f1n.shape
    rows = np.sort(pick.choice(H * W, 64, replace=False))
    gidx = np.sort(pick.choice(f1n.size, 4096, replace=False))
    save("K1_kitti", rows=rows, out_rows=out[0].reshape(-1, H * W)[:, rows].T, out_sum=out.double().sum().item(),
         out_abs=out.double().abs().sum().item(), gidx=gidx,
         dfmap1_s=sampled(f1.grad, gidx), dfmap2_s=sampled(f2.grad, gidx),
         dfmap1_abs=f1.grad.double().abs().sum().item(), dfmap2_abs=f2.grad.double().abs().sum().item(),
         dfmap1_max=f1.grad.abs().max().item(), dfmap2_max=f2.grad.abs().max().item(),
         dcoords=co.grad, num_levels=L, radius=r, shape=np.array(f1n.shape))
    del cb, out, f1, f2, co
    # ---- T1
    f1n, f2n, coords, goutn, wts = I.t1_inputs()
    f1, f2 = t(f1n, True), t(f2n, True)
    gout = t(goutn)
    cb = CorrBlock(f1, f2, num_levels=L, radius=r)
    B, _, H, W = f1n.shape
    rows = np.sort(pick.choice(B * H * W, 64, replace=False))      # (b, q) pairs, flattened
    rec, loss = {}, 0
    for k, (c, wk) in enumerate(zip(coords, wts)):
        o = canonical_lookup_ref(cb, t(c), r)
        loss = loss + wk * (o * gout).sum()
        if k in (0, 5, 11):
            rec[f"out_rows_{k}"] = o.detach().permute(0, 2, 3, 1).reshape(B * H * W, -1)[rows].clone()
        rec[f"out_sum_{k}"] = o.detach().double().sum().item()
        del o
    loss.backward()
    gidx = np.sort(pick.choice(f1n.size, 4096, replace=False))
    save("T1_things", rows=rows, gidx=gidx, dfmap1_s=sampled(f1.grad, gidx), dfmap2_s=sampled(f2.grad, gidx),
         dfmap1_abs=f1.grad.double().abs().sum().item(), dfmap2_abs=f2.grad.double().abs().sum().item(),
         dfmap1_max=f1.grad.abs().max().item(), dfmap2_max=f2.grad.abs().max().item(),
         num_levels=L, radius=r, shape=np.array(f1n.shape), **rec)
    del cb, loss, f1, f2
    # ---- P1
    f1n, f2n, con, qs = I.p1_inputs()
    B, D, H, W = f1n.shape
    f1q = t(f1n).view(D, H * W)[:, qs]                                        # (D, Q)
    vol = torch.matmul(f1q.t(), t(f2n).view(D, H * W)).view(len(qs), 1, H, W)  # corr.py:59, these queries only
    vol = vol / torch.sqrt(torch.tensor(D).float())                            # corr.py:61
    cq = t(con)[0].view(2, H * W)[:, qs].t().reshape(len(qs), 1, 1, 2)         # (Q,1,1,2) (x, y)
    outs = []
    d = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(d, d), axis=-1).view(1, 2 * r + 1, 2 * r + 1, 2)   # corr.py:38-40
    for i in range(L):
        outs.append(bilinear_sampler(vol, cq / 2 ** i + delta).view(len(qs), -1))
        vol = F.avg_pool2d(vol, 2, stride=2)                                   # corr.py:26
    save("P1_1080p", queries=qs, out_rows=torch.cat(outs, -1), num_levels=L, radius=r, shape=np.array(f1n.shape))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "full":      # K1 / T1 / P1 only (minutes of CPU time)
        return full_size_cases()
    torch.set_num_threads(1)     # fixed summation order for the record
    # F1 / F2
    save("F1_fork_fwd", **fork_case(0, 2, 16, 16, 24, 4, 4))
    save("F2_fork_bwd", **fork_case(1, 2, 16, 10, 14, 3, 4, grad=True))
    # F3: ours.py:258-273 pixel-centre reference points, defaults L=2, r=4, odd map, D=24
    H, W = 13, 17
    yy, xx = np.meshgrid(np.arange(H) + 0.5, np.arange(W) + 0.5, indexing="ij")
    centres = np.stack([xx.reshape(-1), yy.reshape(-1)], -1)[None]
    save("F3_fork_defaults", **fork_case(2, 1, 24, H, W, 2, 4, coords=centres))
    # F4: edge cases (integer, negative, beyond the far edge, exactly on the last pixel), r=3, D=128
    H, W = 10, 12
    g = rs(3)
    co = g.rand(1, H * W, 2) * np.array([W - 1, H - 1])
    co[0, 0:20] = np.floor(co[0, 0:20])                    # exactly integer
    co[0, 20:40] = -g.rand(20, 2) * 6                      # negative
    co[0, 40:60] += np.array([W, H])                        # beyond the far edge
    co[0, 60] = [W - 1, H - 1]
    co[0, 61] = [0, 0]
    co[0, 62] = [-4.0, -4.0]
    co[0, 63] = [W - 1 + 4.0, H - 1 + 4.0]
    co[0, 64] = [-100.0, 50.0]                             # window entirely outside
    save("F4_fork_edges", **fork_case(4, 1, 128, H, W, 3, 3, coords=co))
    # C1 / C2
    save("C1_canon_fwd_a", **canon_case(5, 1, 32, 16, 22, 4, 4))
    save("C1_canon_fwd_b", **canon_case(6, 2, 64, 17, 20, 4, 4))       # odd H -> floor pooling
    save("C2_canon_bwd", **canon_case(7, 2, 32, 11, 16, 3, 4, grad=True))
    # S1: Sintel shape.  Inputs are regenerated by the tests from RandomState(1234).
    g = rs(1234)
    B, D, H, W, L, r = 1, 256, 55, 128, 4, 4
    f1 = t(g.randn(B, D, H, W))
    f2 = t(g.randn(B, D, H, W))
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    base = np.stack([xx, yy], 0)[None].astype(np.float32)
    drift = 3.0 * g.randn(B, 2, H, W)
    co4 = t(base + drift)
    co3 = co4.permute(0, 2, 3, 1).reshape(B, H * W, 2).contiguous()
    torch.set_num_threads(8)
    cb = CorrBlock(f1, f2, num_levels=L, radius=r)
    out_fork = cb(co3)
    out_canon = canonical_lookup_ref(cb, co4, r)
    rows = np.sort(rs(99).choice(H * W, 64, replace=False))
    lvl_sum = np.array([p.double().sum().item() for p in cb.corr_pyramid])
    lvl_abs = np.array([p.double().abs().sum().item() for p in cb.corr_pyramid])
    # a few complete volume rows (query slabs) per level, for the build kernel
    vrows = rows[:4]
    slabs = {f"slab_l{l}": cb.corr_pyramid[l][vrows, 0] for l in range(L)}
    save("S1_sintel", rows=rows, vrows=vrows, level_sum=lvl_sum, level_abs=lvl_abs,
         out_fork_rows=out_fork[0, rows], out_canon_rows=out_canon[0].reshape(-1, H * W)[:, rows].T,
         out_fork_sum=out_fork.double().sum().item(), out_canon_sum=out_canon.double().sum().item(),
         num_levels=L, radius=r, shape=np.array([B, D, H, W]), **slabs)
    full_size_cases()


if __name__ == "__main__":
    main()
