"""Canonical RAFT correlation block on stock PyTorch operators (matmul, avg_pool2d, grid_sample),
device-agnostic.  This is the BASELINE the harness times on the same GPU ("what running the
reference's operator sequence on this device costs", core/corr.py:12-61 with the per-level
coords / 2**l of core/corr.py:82) -- not the parity oracle (that lives in oracle/ and is only
used by tests) and not a fallback of the product."""
import torch
import torch.nn.functional as F


class StockCorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels, self.radius = num_levels, radius
        B, D, H, W = fmap1.shape
        corr = torch.matmul(fmap1.view(B, D, H * W).transpose(1, 2), fmap2.view(B, D, H * W))
        corr = corr.view(B * H * W, 1, H, W) / torch.sqrt(torch.tensor(float(D)))
        self.pyramid = [corr]
        for _ in range(num_levels - 1):
            corr = F.avg_pool2d(corr, 2, stride=2)
            self.pyramid.append(corr)

    def __call__(self, coords):
        r = self.radius
        B, _, H, W = coords.shape
        coords = coords.permute(0, 2, 3, 1)
        d = torch.linspace(-r, r, 2 * r + 1, device=coords.device)
        delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1)       # added to (x, y) in this order
        out = []
        for l, vol in enumerate(self.pyramid):
            c = coords.reshape(B * H * W, 1, 1, 2) / 2 ** l + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
            hl, wl = vol.shape[-2:]
            grid = torch.stack([2 * c[..., 0] / (wl - 1) - 1, 2 * c[..., 1] / (hl - 1) - 1], dim=-1)
            out.append(F.grid_sample(vol, grid, align_corners=True).view(B, H, W, -1))
        return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()
