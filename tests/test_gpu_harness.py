"""End-to-end check of the drop-in claim (north star "matching EPE"): a
RAFT-basic model (harness/raft_model.py, stock PyTorch modules, seed-0 random weights -- the
trained checkpoints are not available offline) is run with the oracle's restatement of the
reference CorrBlock and with each of this library's correlation blocks on the same frame pair;
the recurrent loop feeds every lookup's output back into the next lookup's coordinates, so
any systematic difference would grow with the iteration count."""
import numpy as np
import pytest
import torch

from harness.raft_model import RAFT, InputPadder, sequence_loss, synthetic_pair

pytestmark = pytest.mark.gpu


def _flows(dev, factory, iters, size=(200, 320), seed=0, fuse=False, small=False):
    torch.manual_seed(0)
    model = RAFT(factory, fuse_convc1=fuse, small=small).to(dev).eval()
    i1, i2, _ = synthetic_pair(1, *size, seed=seed, device=dev)
    pad = InputPadder(i1.shape)
    a, b = pad.pad(i1, i2)
    with torch.no_grad():
        preds = model(a, b, iters=iters)
    return [pad.unpad(p).cpu().numpy() for p in preds]


def _epe(a, b):
    return float(np.sqrt(((a - b) ** 2).sum(axis=1)).mean())


@pytest.fixture(scope="module")
def reference_flows(dev):
    from oracle import corr_oracle as O
    return _flows(dev, lambda a, b: O.CorrBlockOracle(a, b, 4, 4), 32)


@pytest.mark.parametrize("which", ["all_pairs", "on_the_fly"])
def test_flow_matches_reference_corr_over_32_iterations(which, dev, reference_flows):
    """EPE between the flows produced with the reference's correlation and with ours, full
    resolution, after 12 and after 32 iterations: <= 0.01 px (the north star's bar)."""
    from raft_amd import AlternateCorrBlock, CorrBlock
    fac = (lambda a, b: CorrBlock(a, b, num_levels=4, radius=4)) if which == "all_pairs" else \
          (lambda a, b: AlternateCorrBlock(a, b, num_levels=4, radius=4))
    ours = _flows(dev, fac, 32)
    assert np.isfinite(ours[-1]).all() and np.abs(reference_flows[-1]).max() > 1.0    # the loop moved
    for it in (11, 31):
        assert _epe(ours[it], reference_flows[it]) <= 0.01, f"iteration {it + 1}"


@pytest.mark.parametrize("which", ["all_pairs", "all_pairs_fused_convc1", "on_the_fly"])
def test_raft_small_flow_matches_reference_corr(which, dev):
    """RAFT-small (core/raft.py:31-35: radius 3, 128-channel features, 96-channel convc1, ConvGRU, bilinear upsampling)
    around the same correlation blocks: EPE vs the run with the reference's CorrBlock <= 0.01 px after 12 and 32
    iterations.  Exercises the radius-3 instantiations of the lookup kernels and the 196 -> 96 fused convolution."""
    from oracle import corr_oracle as O
    from raft_amd import AlternateCorrBlock, CorrBlock
    ref = _flows(dev, lambda a, b: O.CorrBlockOracle(a, b, 4, 3), 32, small=True)
    fac = (lambda a, b: AlternateCorrBlock(a, b, num_levels=4, radius=3)) if which == "on_the_fly" else \
          (lambda a, b: CorrBlock(a, b, num_levels=4, radius=3))
    ours = _flows(dev, fac, 32, small=True, fuse=which.endswith("convc1"))
    assert ours[-1].shape == ref[-1].shape and np.isfinite(ours[-1]).all() and np.abs(ref[-1]).max() > 1.0
    for it in (11, 31):
        assert _epe(ours[it], ref[it]) <= 0.01, f"iteration {it + 1}"


def test_flow_with_fused_convc1_and_split_precision_build(dev, reference_flows):
    """The fast inference configuration: fp16x2 build + lookup fused with convc1 -- still within the
    north star's 0.01 px of the reference CorrBlock after 12 and 32 iterations."""
    from raft_amd import CorrBlock
    ours = _flows(dev, lambda a, b: CorrBlock(a, b, num_levels=4, radius=4, precision="fp16x2"), 32, fuse=True)
    for it in (11, 31):
        assert _epe(ours[it], reference_flows[it]) <= 0.01, f"iteration {it + 1}"


def test_flow_on_the_reference_demo_frames(dev):
    """BASELINE.json configs[0]'s input: the frame pair the reference's demo.py runs on
    (tests/golden/demo_frames.npz, 436 x 1024 -> the Sintel map 55 x 128), seed-0 random weights.
    Natural images give smooth, highly self-similar feature maps -- a different regime from the
    synthetic pair above.  12 iterations, all four blocks against the oracle's CorrBlock."""
    import os
    from oracle import corr_oracle as O
    from raft_amd import AlternateCorrBlock, CorrBlock
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "demo_frames.npz"))
    i1 = torch.from_numpy(z["frame1"]).permute(2, 0, 1)[None].float().to(dev)
    i2 = torch.from_numpy(z["frame2"]).permute(2, 0, 1)[None].float().to(dev)
    pad = InputPadder(i1.shape)
    a, b = pad.pad(i1, i2)
    assert a.shape[-2:] == (440, 1024)

    def run(factory, fuse=False):
        torch.manual_seed(0)
        model = RAFT(factory, fuse_convc1=fuse).to(dev).eval()
        with torch.no_grad():
            return pad.unpad(model(a, b, iters=12)[-1]).cpu().numpy()

    ref = run(lambda x, y: O.CorrBlockOracle(x, y, 4, 4))
    assert np.isfinite(ref).all() and np.abs(ref).max() > 0.1
    assert _epe(run(lambda x, y: CorrBlock(x, y, num_levels=4, radius=4)), ref) <= 0.01
    assert _epe(run(lambda x, y: AlternateCorrBlock(x, y, num_levels=4, radius=4)), ref) <= 0.01
    assert _epe(run(lambda x, y: CorrBlock(x, y, num_levels=4, radius=4, precision="fp16x2"), fuse=True), ref) <= 0.01
    assert _epe(run(lambda x, y: CorrBlock(x, y, num_levels=4, radius=4, precision="bf16x3")), ref) <= 0.01


def test_flow_with_bf16_build_stays_close(dev, reference_flows):
    """precision="bf16": the volume differs by the bf16 input rounding; the flow after 12
    iterations stays within 0.05 px of the fp32 reference on this pair (reported, loose bound)."""
    from raft_amd import CorrBlock
    ours = _flows(dev, lambda a, b: CorrBlock(a, b, num_levels=4, radius=4, precision="bf16"), 12)
    assert _epe(ours[11], reference_flows[11]) <= 0.05


class _Recording:
    """Correlation factory that keeps what crosses the block's boundary in a training step: the two feature maps,
    every lookup's coordinates and -- through hooks -- the gradient that arrives at every lookup's output and the
    gradients the block hands back to the feature maps."""

    def __init__(self, make):
        self.make, self.f, self.df, self.coords, self.douts = make, [], [None, None], [], {}

    def __call__(self, fmap1, fmap2):
        self.f = [fmap1, fmap2]
        for i, t in enumerate(self.f):
            t.register_hook(lambda g, i=i: self.df.__setitem__(i, g.detach().clone()))
        blk = self.make(fmap1, fmap2)

        def lookup(coords):
            k = len(self.coords)
            self.coords.append(coords.detach().clone())
            out = blk(coords)
            out.register_hook(lambda g, k=k: self.douts.__setitem__(k, g.detach().clone()))
            return out
        return lookup


def test_training_step_gradients_match_reference_corr(dev):
    """One training step (sequence loss over 4 iterations, core/train_origin.py:49-79; freeze_bn as core/raft.py:60-63).
    The quantity under test is what the correlation block hands back: d(loss)/d(fmap1), d(loss)/d(fmap2).  The model
    around the block is stock PyTorch and its MIOpen convolutions are not reproducible from run to run (the reference
    against itself moves parameters by up to 9e-4 of their scale, tools/train_grad_diag.py), so the block is
    ISOLATED: the step runs once with this library's block, the tensors at the block's boundary are recorded (feature
    maps, the 4 lookups' coordinates, the 4 upstream gradients), and the reference's CorrBlock (oracle restatement,
    autograd through matmul / avg_pool2d / grid_sample) is replayed on exactly those tensors.  Bar: 1e-5 of each batch
    item's gradient scale, both maps, both blocks -- no noise margin."""
    from conftest import rel_err_by_item
    from oracle import corr_oracle as O
    from raft_amd import AlternateCorrBlock, CorrBlock

    def step(make):
        torch.manual_seed(0)
        rec = _Recording(make)
        model = RAFT(rec).to(dev).train()
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
        i1, i2, gt = synthetic_pair(2, 136, 176, seed=3, device=dev)
        preds = model(i1, i2, iters=4)
        loss, _ = sequence_loss(preds, gt, torch.ones_like(gt[:, 0]))
        loss.backward()
        grads = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}
        return rec, float(loss.detach()), grads

    for name, make in (("all_pairs", lambda a, b: CorrBlock(a, b, num_levels=4, radius=4)),
                       ("on_the_fly", lambda a, b: AlternateCorrBlock(a, b, num_levels=4, radius=4))):
        rec, loss, grads = step(make)
        assert len(rec.coords) == 4 and sorted(rec.douts) == [0, 1, 2, 3] and rec.df[0] is not None and rec.df[1] is not None
        assert any(k.startswith("fnet.") and np.abs(g).max() > 0 for k, g in grads.items())      # the gradient reached the encoder
        f1 = rec.f[0].detach().float().cpu().requires_grad_(True)
        f2 = rec.f[1].detach().float().cpu().requires_grad_(True)
        ref = O.CorrBlockOracle(f1, f2, 4, 4)
        torch.autograd.backward([ref(c.cpu()) for c in rec.coords], [rec.douts[k].cpu() for k in range(4)])
        e1 = rel_err_by_item(rec.df[0].cpu().numpy(), f1.grad.numpy())
        e2 = rel_err_by_item(rec.df[1].cpu().numpy(), f2.grad.numpy())
        print(f"train step, {name}: d fmap1 {max(e1):.2e}, d fmap2 {max(e2):.2e} of the item's scale (loss {loss:.4f})")
        from conftest import log_errors
        log_errors("train_step_isolated_" + name, dfmap1_by_item=e1, dfmap2_by_item=e2)
        # the on-the-fly block samples in pixel coordinates (its reference: the CUDA kernel), the replayed reference is the
        # all-pairs block with its coordinate round trip: 3e-5 between the reference's own two paths (test_gpu_parity K1)
        bar = 1e-5 if name == "all_pairs" else 3e-5
        assert max(e1) <= bar and max(e2) <= bar, (name, e1, e2)


def test_training_step_parameter_gradients_sanity(dev):
    """The same step judged at the PARAMETERS (every encoder / update-block weight), ours vs the oracle's CorrBlock
    inside the same stock model: loss equal to 1e-5, every parameter gradient within the stock model's own
    run-to-run differences (measured here: three reference runs) plus 1e-4 of its scale and 5e-5 of the largest
    gradient.  The correlation gradients themselves agree to 1e-6-grade (isolated test above, which carries the tight
    bar); the encoder's backward pass (instance norm: differences of large sums) amplifies such rounding-level
    differences -- fnet.conv1.weight, whose gradient is 50x smaller than the largest one, moves by 1.5e-5 of the
    largest gradient even when the reference is bit-stable against itself (measured, round 3).  A wiring check: a
    wrong correlation gradient shows up at 1e-1 and more."""
    from oracle import corr_oracle as O
    from raft_amd import CorrBlock

    def grads(factory):
        torch.manual_seed(0)
        model = RAFT(factory).to(dev).train()
        for m in model.modules():                               # freeze_bn, core/raft.py:60-63
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
        i1, i2, gt = synthetic_pair(2, 136, 176, seed=3, device=dev)
        preds = model(i1, i2, iters=4)
        loss, _ = sequence_loss(preds, gt, torch.ones_like(gt[:, 0]))
        loss.backward()
        return float(loss.detach()), {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters()
                             if p.grad is not None}

    l_ref, g_ref = grads(lambda a, b: O.CorrBlockOracle(a, b, 4, 4))
    noise = {k: 0.0 for k in g_ref}
    for _ in range(2):
        _, g_again = grads(lambda a, b: O.CorrBlockOracle(a, b, 4, 4))
        noise = {k: max(noise[k], float(np.abs(g_ref[k] - g_again[k]).max())) for k in g_ref}
    l_our, g_our = grads(lambda a, b: CorrBlock(a, b, num_levels=4, radius=4))
    assert abs(l_ref - l_our) <= 1e-5 * abs(l_ref)
    assert g_ref.keys() == g_our.keys() and any(k.startswith("fnet.") for k in g_ref)
    gmax = max(np.abs(g).max() for g in g_ref.values())
    floor = max(noise.values())                                 # the stock model against itself, worst parameter
    worst = 0.0
    for k in g_ref:       # parameters in front of a norm layer have a zero gradient (1e-8 noise): atol
        scale = np.abs(g_ref[k]).max()
        worst = max(worst, float(np.abs(g_ref[k] - g_our[k]).max() / (scale + 1e-2 * gmax)))
        assert np.abs(g_ref[k] - g_our[k]).max() <= 1e-4 * scale + 5e-5 * gmax + 4.0 * max(noise[k], floor), (k, noise[k], floor)
    print(f"train-step parameter gradients: worst difference {worst:.1e} of a parameter's scale; "
          f"reference against itself: {floor / gmax:.1e} of the largest gradient")
