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


def _flows(dev, factory, iters, size=(200, 320), seed=0, fuse=False):
    torch.manual_seed(0)
    model = RAFT(factory, fuse_convc1=fuse).to(dev).eval()
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


def test_training_step_gradients_match_reference_corr(dev):
    """One training step's gradients (sequence loss over 4 iterations, core/train_origin.py:49-79)
    w.r.t. every encoder / update-block parameter: ours vs the oracle's CorrBlock under autograd."""
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
    # The model around the correlation block is stock PyTorch, and its convolution kernels are not reproducible from
    # run to run on this stack: the REFERENCE against itself (tools/train_grad_diag.py ref, 4 runs x 3 processes)
    # differs by 1e-6 .. 9e-4 of a parameter's scale (update_block.flow_head.conv1.bias, encoder.convf1.bias,
    # encoder.convc1.weight), in a few discrete steps -- kernel selection, not rounding.  The bar is therefore the
    # noise floor measured here (two more reference runs) plus a margin of 2e-3 of the parameter's scale; a wrong
    # correlation gradient shows up at 1e-1 and more.
    noise = {k: 0.0 for k in g_ref}
    for _ in range(2):
        _, g_again = grads(lambda a, b: O.CorrBlockOracle(a, b, 4, 4))
        noise = {k: max(noise[k], float(np.abs(g_ref[k] - g_again[k]).max())) for k in g_ref}
    l_our, g_our = grads(lambda a, b: CorrBlock(a, b, num_levels=4, radius=4))
    assert abs(l_ref - l_our) <= 1e-5 * abs(l_ref)
    assert g_ref.keys() == g_our.keys() and any(k.startswith("fnet.") for k in g_ref)
    gmax = max(np.abs(g).max() for g in g_ref.values())
    worst = 0.0
    for k in g_ref:       # parameters in front of a norm layer have a zero gradient (1e-8 noise): atol
        scale = np.abs(g_ref[k]).max()
        worst = max(worst, float(np.abs(g_ref[k] - g_our[k]).max() / (scale + 1e-2 * gmax)))
        assert np.abs(g_ref[k] - g_our[k]).max() <= 2e-3 * scale + 1e-5 * gmax + 4.0 * noise[k], (k, noise[k])
    print(f"train-step gradients: worst per-parameter difference {worst:.1e} of its scale; "
          f"reference against itself: {max(noise.values()) / gmax:.1e} of the largest gradient")
