"""RAFT-basic and RAFT-small inference/training models on stock PyTorch modules, with the correlation block injected.

Why it exists: the correlation path is only meaningful inside RAFT's
recurrent loop -- every lookup's coordinates depend on the previous lookup's output -- so the
end-to-end claim "same flow as with the reference's CorrBlock" needs the loop around it.  The
architecture and the parameter names follow the reference so that its checkpoints
(`module.`-prefixed state dicts, demo.py:43-46) load unchanged:
  feature / context encoder  core/extractor_origin.py:6-55,116-189  (fnet: instance norm, cnet: batch norm)
  motion encoder + SepConvGRU + flow head + mask head  core/update.py:6-14,33-60,79-97,114-136
  loop, pixel-unit coords, convex upsampling  core/raft.py:65-85,87-145
  InputPadder  core/utils/utils.py:7-24
  RAFT-small (RAFT(..., small=True)): core/raft.py:31-35,50-53 (hidden 96, context 64, radius 3, 128-channel features),
  bottleneck encoder core/extractor_origin.py:58-113,192-263, motion encoder + ConvGRU core/update.py:16-31,62-77,99-112,
  bilinear x8 upsampling instead of the convex mask (core/raft.py:135-136)
The model never imports a correlation implementation: `corr_factory(fmap1, fmap2)` must return a
callable `coords (B,2,H/8,W/8) -> (B, 4*81, H/8, W/8)` (raft_amd.CorrBlock, raft_amd.AlternateCorrBlock,
or the oracle's restatement of the reference).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

CORR_LEVELS, CORR_RADIUS = 4, 4          # core/raft.py:40-41
HIDDEN, CONTEXT = 128, 128               # core/raft.py:38-39


def _norm(kind, ch):
    if kind == "instance":
        return nn.InstanceNorm2d(ch)
    if kind == "batch":
        return nn.BatchNorm2d(ch)
    if kind == "none":
        return nn.Sequential()
    raise ValueError(kind)


class ResidualBlock(nn.Module):
    def __init__(self, cin, cout, norm, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.norm1, self.norm2 = _norm(norm, cout), _norm(norm, cout)
        self.downsample = None
        if stride != 1:
            self.norm3 = _norm(norm, cout)
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride), self.norm3)

    def forward(self, x):
        y = F.relu(self.norm1(self.conv1(x)))
        y = F.relu(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return F.relu(x + y)


class BasicEncoder(nn.Module):
    """1/8-resolution encoder: 7x7 s2 stem, three 2-block stages (64, 96 s2, 128 s2), 1x1 head."""

    def __init__(self, output_dim, norm):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3)
        self.norm1 = _norm(norm, 64)
        widths, strides, cin = (64, 96, 128), (1, 2, 2), 64
        for i, (w, s) in enumerate(zip(widths, strides), start=1):
            setattr(self, f"layer{i}", nn.Sequential(ResidualBlock(cin, w, norm, s), ResidualBlock(w, w, norm, 1)))
            cin = w
        self.conv2 = nn.Conv2d(128, output_dim, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        pair = isinstance(x, (tuple, list))
        if pair:
            n = x[0].shape[0]
            x = torch.cat(x, 0)
        x = F.relu(self.norm1(self.conv1(x)))
        x = self.conv2(self.layer3(self.layer2(self.layer1(x))))
        return torch.split(x, [n, n], 0) if pair else x


class Bottleneck(nn.Module):
    """1x1 (to a quarter of the width) -> 3x3 (carries the stride) -> 1x1, residual."""

    def __init__(self, cin, cout, norm, stride):
        super().__init__()
        mid = cout // 4
        self.conv1 = nn.Conv2d(cin, mid, 1)
        self.conv2 = nn.Conv2d(mid, mid, 3, padding=1, stride=stride)
        self.conv3 = nn.Conv2d(mid, cout, 1)
        self.norm1, self.norm2, self.norm3 = _norm(norm, mid), _norm(norm, mid), _norm(norm, cout)
        self.downsample = None
        if stride != 1:
            self.norm4 = _norm(norm, cout)
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride), self.norm4)

    def forward(self, x):
        y = F.relu(self.norm1(self.conv1(x)))
        y = F.relu(self.norm2(self.conv2(y)))
        y = F.relu(self.norm3(self.conv3(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return F.relu(x + y)


class SmallEncoder(nn.Module):
    """1/8-resolution encoder of RAFT-small: 7x7 s2 stem (32), three 2-bottleneck stages (32, 64 s2, 96 s2), 1x1 head."""

    def __init__(self, output_dim, norm):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 32, 7, stride=2, padding=3)
        self.norm1 = _norm(norm, 32)
        cin = 32
        for i, (w, s) in enumerate(zip((32, 64, 96), (1, 2, 2)), start=1):
            setattr(self, f"layer{i}", nn.Sequential(Bottleneck(cin, w, norm, s), Bottleneck(w, w, norm, 1)))
            cin = w
        self.conv2 = nn.Conv2d(96, output_dim, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        pair = isinstance(x, (tuple, list))
        if pair:
            n = x[0].shape[0]
            x = torch.cat(x, 0)
        x = F.relu(self.norm1(self.conv1(x)))
        x = self.conv2(self.layer3(self.layer2(self.layer1(x))))
        return torch.split(x, [n, n], 0) if pair else x


class SmallMotionEncoder(nn.Module):
    def __init__(self, radius):
        super().__init__()
        self.convc1 = nn.Conv2d(CORR_LEVELS * (2 * radius + 1) ** 2, 96, 1)
        self.convf1 = nn.Conv2d(2, 64, 7, padding=3)
        self.convf2 = nn.Conv2d(64, 32, 3, padding=1)
        self.conv = nn.Conv2d(128, 80, 3, padding=1)

    def forward(self, flow, corr):
        # `corr`: the (B, 196, H, W) lookup result, or relu(convc1(lookup)) when the block fused it (96 channels)
        c1 = corr if corr.shape[1] == self.convc1.out_channels else F.relu(self.convc1(corr))
        flo = F.relu(self.convf2(F.relu(self.convf1(flow))))
        return torch.cat([F.relu(self.conv(torch.cat([c1, flo], 1))), flow], 1)


class ConvGRU(nn.Module):
    def __init__(self, hidden, inp):
        super().__init__()
        for gate in "zrq":
            setattr(self, "conv" + gate, nn.Conv2d(hidden + inp, hidden, 3, padding=1))

    def forward(self, h, x):
        hx = torch.cat([h, x], 1)
        z, r = torch.sigmoid(self.convz(hx)), torch.sigmoid(self.convr(hx))
        q = torch.tanh(self.convq(torch.cat([r * h, x], 1)))
        return (1 - z) * h + z * q


class SmallUpdateBlock(nn.Module):
    def __init__(self, radius, hidden=96):
        super().__init__()
        self.encoder = SmallMotionEncoder(radius)
        self.gru = ConvGRU(hidden, 82 + 64)
        self.flow_head = FlowHead(hidden, 128)

    def forward(self, net, inp, corr, flow):
        net = self.gru(net, torch.cat([inp, self.encoder(flow, corr)], 1))
        return net, None, self.flow_head(net)


class BasicMotionEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        planes = CORR_LEVELS * (2 * CORR_RADIUS + 1) ** 2
        self.convc1 = nn.Conv2d(planes, 256, 1)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)

    def forward(self, flow, corr):
        # `corr` is the (B, 324, H, W) lookup result, or -- when the correlation block fuses convc1
        # into its lookup (RAFT(..., fuse_convc1=True)) -- already relu(convc1(lookup))
        c1 = corr if corr.shape[1] == self.convc1.out_channels else F.relu(self.convc1(corr))
        cor = F.relu(self.convc2(c1))
        flo = F.relu(self.convf2(F.relu(self.convf1(flow))))
        return torch.cat([F.relu(self.conv(torch.cat([cor, flo], 1))), flow], 1)


class SepConvGRU(nn.Module):
    def __init__(self, hidden=128, inp=128 + 128):
        super().__init__()
        for tag, k, pad in (("1", (1, 5), (0, 2)), ("2", (5, 1), (2, 0))):
            for gate in "zrq":
                setattr(self, f"conv{gate}{tag}", nn.Conv2d(hidden + inp, hidden, k, padding=pad))

    def _half(self, h, x, tag):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(getattr(self, "convz" + tag)(hx))
        r = torch.sigmoid(getattr(self, "convr" + tag)(hx))
        q = torch.tanh(getattr(self, "convq" + tag)(torch.cat([r * h, x], 1)))
        return (1 - z) * h + z * q

    def forward(self, h, x):
        return self._half(self._half(h, x, "1"), x, "2")       # horizontal, then vertical


class FlowHead(nn.Module):
    def __init__(self, cin=128, hidden=256):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, hidden, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden, 2, 3, padding=1)

    def forward(self, x):
        return self.conv2(F.relu(self.conv1(x)))


class BasicUpdateBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = BasicMotionEncoder()
        self.gru = SepConvGRU(HIDDEN, 128 + HIDDEN)
        self.flow_head = FlowHead(HIDDEN, 256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, 64 * 9, 1))

    def forward(self, net, inp, corr, flow):
        net = self.gru(net, torch.cat([inp, self.encoder(flow, corr)], 1))
        return net, 0.25 * self.mask(net), self.flow_head(net)


def coords_grid(B, H, W, device):
    """(B, 2, H, W) pixel coordinates, channel 0 = x (core/utils/utils.py:74-77)."""
    ys, xs = torch.meshgrid(torch.arange(H, device=device), torch.arange(W, device=device), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].repeat(B, 1, 1, 1)


def convex_upsample(flow, mask):
    """(B,2,H,W) at 1/8 resolution -> (B,2,8H,8W): per-pixel softmax-weighted 3x3 combination."""
    B, _, H, W = flow.shape
    mask = torch.softmax(mask.view(B, 1, 9, 8, 8, H, W), dim=2)
    nb = F.unfold(8 * flow, [3, 3], padding=1).view(B, 2, 9, 1, 1, H, W)
    up = torch.sum(mask * nb, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(B, 2, 8 * H, 8 * W)


class InputPadder:
    """Pads images so that H and W are divisible by 8 (sintel: symmetric; kitti: top only)."""

    def __init__(self, dims, mode="sintel"):
        self.ht, self.wd = dims[-2:]
        ph = (((self.ht // 8) + 1) * 8 - self.ht) % 8
        pw = (((self.wd // 8) + 1) * 8 - self.wd) % 8
        self._pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2] if mode == "sintel" else \
                    [pw // 2, pw - pw // 2, 0, ph]

    def pad(self, *images):
        return [F.pad(x, self._pad, mode="replicate") for x in images]

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        c = [self._pad[2], ht - self._pad[3], self._pad[0], wd - self._pad[1]]
        return x[..., c[0]:c[1], c[2]:c[3]]


def upflow8(flow):
    """(B,2,H,W) -> (B,2,8H,8W), bilinear, align_corners=True, values x8 (core/utils/utils.py:80-82)."""
    return 8 * F.interpolate(flow, size=(8 * flow.shape[2], 8 * flow.shape[3]), mode="bilinear", align_corners=True)


class RAFT(nn.Module):
    def __init__(self, corr_factory, fuse_convc1=False, small=False):
        """fuse_convc1: the correlation block must offer lookup_conv1x1 (differentiable)
        (raft_amd.CorrBlock does): the motion encoder's first 1x1 conv + ReLU then runs inside
        the lookup kernel and the (B, 324, H, W) tensor is never materialised.
        small: RAFT-small -- the factory must build a radius-3 block on 128-channel features."""
        super().__init__()
        self.corr_factory = corr_factory
        self.fuse_convc1 = fuse_convc1
        self.small = small
        if small:
            self.hidden, self.context = 96, 64
            self.fnet = SmallEncoder(128, "instance")
            self.cnet = SmallEncoder(self.hidden + self.context, "none")
            self.update_block = SmallUpdateBlock(radius=3, hidden=self.hidden)
        else:
            self.hidden, self.context = HIDDEN, CONTEXT
            self.fnet = BasicEncoder(256, "instance")
            self.cnet = BasicEncoder(HIDDEN + CONTEXT, "batch")
            self.update_block = BasicUpdateBlock()

    def load_reference_checkpoint(self, state_dict):
        """Accepts the reference's DataParallel checkpoints (keys prefixed with `module.`)."""
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        return self.load_state_dict(sd, strict=True)

    def forward(self, image1, image2, iters=12, flow_init=None, test_mode=False):
        image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
        image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
        fmap1, fmap2 = self.fnet([image1, image2])
        corr_fn = self.corr_factory(fmap1.float(), fmap2.float())
        net, inp = torch.split(self.cnet(image1), [self.hidden, self.context], 1)
        net, inp = torch.tanh(net), torch.relu(inp)
        B, _, H, W = image1.shape
        coords0 = coords_grid(B, H // 8, W // 8, image1.device)
        coords1 = coords0.clone() if flow_init is None else coords0 + flow_init
        preds = []
        for _ in range(iters):
            coords1 = coords1.detach()
            if self.fuse_convc1:
                enc = self.update_block.encoder.convc1
                corr = corr_fn.lookup_conv1x1(coords1, enc.weight, enc.bias, relu=True)
            else:
                corr = corr_fn(coords1)
            net, mask, delta = self.update_block(net, inp, corr, coords1 - coords0)
            coords1 = coords1 + delta
            preds.append(upflow8(coords1 - coords0) if mask is None else convex_upsample(coords1 - coords0, mask))
        return (coords1 - coords0, preds[-1]) if test_mode else preds


def sequence_loss(preds, flow_gt, valid, gamma=0.8, max_flow=400.0):
    """core/train_origin.py:49-79.  NB the reference computes the gamma weights and then
    overwrites them with 1.0 (:62): every prediction is weighted equally, which is kept here."""
    mag = torch.sum(flow_gt ** 2, dim=1).sqrt()
    valid = (valid >= 0.5) & (mag < max_flow)
    loss = 0.0
    for p in preds:
        loss = loss + (valid[:, None] * (p - flow_gt).abs()).mean()
    epe = torch.sum((preds[-1] - flow_gt) ** 2, dim=1).sqrt().view(-1)[valid.view(-1)]
    return loss, {"epe": epe.mean().item(), "1px": (epe < 1).float().mean().item()}


def synthetic_pair(B, H, W, seed=0, device="cpu", max_disp=12.0):
    """A textured frame, a smooth ground-truth flow and the second frame obtained by sampling the
    first one backwards along it (so the pair has real correspondences for the volume to find)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.zeros(B, 3, H, W)
    for s in (4, 8, 16, 32, 64):                       # multi-scale random texture
        img = img + F.interpolate(torch.rand(B, 3, max(H // s, 2), max(W // s, 2), generator=g), size=(H, W),
                                  mode="bilinear", align_corners=False)
    img1 = 255.0 * (img - img.amin()) / (img.amax() - img.amin())
    flow = F.interpolate(max_disp * (torch.rand(B, 2, 3, 4, generator=g) * 2 - 1), size=(H, W), mode="bicubic",
                         align_corners=True)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    gx = (xs[None] - flow[:, 0]) / (W - 1) * 2 - 1
    gy = (ys[None] - flow[:, 1]) / (H - 1) * 2 - 1
    img2 = F.grid_sample(img1, torch.stack([gx, gy], -1), mode="bilinear", padding_mode="border", align_corners=True)
    return img1.to(device), img2.to(device), flow.to(device)
