"""CPU: pin the oracle (oracle/) against golden vectors recorded from the real reference.

Bars: the torch restatement executes the reference's operator sequence, so it must match the
recorded outputs to float32 round-off of a different BLAS threading (<= 2e-6 relative); the
first-principles float64 restatement and the C restatement of the CUDA kernel use different
arithmetic orders and coordinate handling and must match within 1e-5 relative (the north-star
fp32 bar)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import corr_oracle as O
from oracle import alt_corr_c

FORK = ["F1_fork_fwd", "F2_fork_bwd", "F3_fork_defaults", "F4_fork_edges"]
CANON = ["C1_canon_fwd_a", "C1_canon_fwd_b", "C2_canon_bwd"]


def _t(a, grad=False):
    return torch.from_numpy(np.array(a)).requires_grad_(grad)


@pytest.mark.parametrize("name", FORK + CANON)
def test_torch_restatement_matches_reference(name):
    g = load_golden(name)
    L, r = int(g["num_levels"]), int(g["radius"])
    grad = "gout" in g
    f1, f2, co = _t(g["fmap1"], grad), _t(g["fmap2"], grad), _t(g["coords"], grad)
    out = O.CorrBlockOracle(f1, f2, L, r)(co)
    assert out.shape == g["out"].shape and out.dtype == torch.float32
    assert rel_err(out.detach().numpy(), g["out"]) <= 2e-6
    if grad:
        out.backward(_t(g["gout"]))
        assert rel_err(f1.grad.numpy(), g["dfmap1"]) <= 2e-6
        assert rel_err(f2.grad.numpy(), g["dfmap2"]) <= 2e-6
        assert rel_err(co.grad.numpy(), g["dcoords"]) <= 2e-5


@pytest.mark.parametrize("name", FORK + CANON)
def test_direct_restatement_matches_reference(name):
    g = load_golden(name)
    L, r = int(g["num_levels"]), int(g["radius"])
    pyr = O.build_pyramid_direct(g["fmap1"], g["fmap2"], L)
    co = g["coords"]
    if co.ndim == 3:
        out = O.lookup_direct(pyr, co, r, per_level_scale=False)
    else:
        B, _, H, W = co.shape
        xy = co.transpose(0, 2, 3, 1).reshape(B, H * W, 2)
        out = O.lookup_direct(pyr, xy, r, per_level_scale=True)
        out = out.reshape(B, H, W, -1).transpose(0, 3, 1, 2)
    assert rel_err(out, g["out"]) <= 1e-5


def test_known_answers_integer_coords_and_zero_padding():
    """SURVEY 8(c) known-answer checks: integer coords pick volume entries exactly; taps outside
    the map contribute 0; vol[q,t] = <f1[:,q], f2[:,t]>/sqrt(D)."""
    rs = np.random.RandomState(11)
    B, D, H, W, r = 1, 8, 6, 7, 2
    f1, f2 = rs.randn(B, D, H, W), rs.randn(B, D, H, W)
    pyr = O.build_pyramid_direct(f1, f2, 1)
    vol = pyr[0]
    q, t = 9, 20
    assert np.isclose(vol[0, q].reshape(-1)[t], f1.reshape(D, -1)[:, q] @ f2.reshape(D, -1)[:, t] / np.sqrt(D))
    xy = np.zeros((1, H * W, 2))
    xy[0, :, 0], xy[0, :, 1] = 3.0, 2.0
    out = O.lookup_direct(pyr, xy, r, False).reshape(H * W, 5, 5)
    for a in range(5):
        for b in range(5):
            x, y = 3 + a - r, 2 + b - r
            exp = vol[0, :, y, x] if (0 <= x < W and 0 <= y < H) else 0.0
            assert np.allclose(out[:, a, b], exp)
    tor = O.lookup_fork_torch([torch.from_numpy(vol.reshape(H * W, 1, H, W)).float()],
                              torch.from_numpy(xy).float(), r).numpy().reshape(H * W, 5, 5)
    assert rel_err(tor, out) <= 1e-6


@pytest.mark.parametrize("name", CANON)
def test_alt_c_restatement_matches_canonical_reference(name):
    """A1: on-the-fly correlation == canonical all-pairs lookup (pooling the volume over the
    target dims == correlating with avg-pooled fmap2).  Checks the C restatement of
    correlation_kernel.cu through the AlternateCorrBlock composition (corr.py:75-92)."""
    g = load_golden(name)
    L, r = int(g["num_levels"]), int(g["radius"])
    f1, f2, co = _t(g["fmap1"]), _t(g["fmap2"]), g["coords"]
    B, D, H, W = f1.shape
    pyr = O.alt_feature_pyramid_torch(f2, L)
    f1n = f1.permute(0, 2, 3, 1).contiguous().numpy()
    cperm = co.transpose(0, 2, 3, 1)
    outs, outs64 = [], []
    for i in range(L):
        f2n = pyr[i].permute(0, 2, 3, 1).contiguous().numpy()
        ci = np.ascontiguousarray((cperm / 2 ** i).reshape(B, 1, H, W, 2))
        outs.append(alt_corr_c.forward(f1n, f2n, ci, r)[0][:, 0])
        outs64.append(O.alt_forward_level_direct(f1n, f2n, ci, r)[:, 0])
    corr = np.stack(outs, 1).reshape(B, -1, H, W) / np.sqrt(np.float32(D))
    corr64 = np.stack(outs64, 1).reshape(B, -1, H, W) / np.sqrt(D)
    assert rel_err(corr, g["out"]) <= 1e-5
    assert rel_err(corr64, g["out"]) <= 1e-5


def test_alt_c_backward_matches_direct_and_autograd():
    g = load_golden("C2_canon_bwd")
    r = int(g["radius"])
    f1, f2 = _t(g["fmap1"]), _t(g["fmap2"])
    B, D, H, W = f1.shape
    f1n = f1.permute(0, 2, 3, 1).contiguous().numpy()
    f2n = f2.permute(0, 2, 3, 1).contiguous().numpy()
    co = np.ascontiguousarray(g["coords"].transpose(0, 2, 3, 1).reshape(B, 1, H, W, 2))
    rs = np.random.RandomState(5)
    cg = rs.randn(B, 1, (2 * r + 1) ** 2, H, W).astype(np.float32)
    g1, g2, gc = alt_corr_c.backward(f1n, f2n, co, cg, r)
    d1, d2, dc = O.alt_backward_level_direct(f1n, f2n, co, cg, r)
    assert rel_err(g1, d1) <= 1e-5 and rel_err(g2, d2) <= 1e-5
    assert not gc.any() and not dc.any()       # reference never writes coords_grad
    # level-0 alt backward == autograd of the level-0 all-pairs lookup (unscaled by sqrt(D))
    f1t, f2t = _t(g["fmap1"], True), _t(g["fmap2"], True)
    pyr = O.build_pyramid_torch(f1t, f2t, 1)
    out = O.lookup_canonical_torch(pyr, _t(g["coords"]), r) * np.sqrt(np.float32(D))
    out.backward(torch.from_numpy(cg[:, 0]))
    assert rel_err(g1, f1t.grad.permute(0, 2, 3, 1).numpy()) <= 1e-5
    assert rel_err(g2, f2t.grad.permute(0, 2, 3, 1).numpy()) <= 1e-5


def test_sintel_shape_checksums():
    """S1: the oracle reproduces the reference at the benchmark shape (sizes the small fixtures
    do not reach: 55 -> 27 -> 13 -> 6 floor pooling, N = 7040)."""
    g = load_golden("S1_sintel")
    B, D, H, W = (int(v) for v in g["shape"])
    L, r = int(g["num_levels"]), int(g["radius"])
    rs = np.random.RandomState(1234)
    f1 = torch.from_numpy(rs.randn(B, D, H, W).astype(np.float32))
    f2 = torch.from_numpy(rs.randn(B, D, H, W).astype(np.float32))
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    base = np.stack([xx, yy], 0)[None].astype(np.float32)
    co4 = torch.from_numpy((base + 3.0 * rs.randn(B, 2, H, W)).astype(np.float32))
    co3 = co4.permute(0, 2, 3, 1).reshape(B, H * W, 2).contiguous()
    cb = O.CorrBlockOracle(f1, f2, L, r)
    for l in range(L):
        p = cb.corr_pyramid[l]
        assert abs(p.double().abs().sum().item() - g["level_abs"][l]) <= 1e-6 * g["level_abs"][l]
        assert rel_err(p[g["vrows"], 0].numpy(), g[f"slab_l{l}"]) <= 2e-6
    rows = g["rows"]
    assert rel_err(cb(co3)[0, rows].numpy(), g["out_fork_rows"]) <= 2e-6
    oc = cb(co4)[0].reshape(-1, H * W)[:, rows].T.numpy()
    assert rel_err(oc, g["out_canon_rows"]) <= 2e-6


# ---- full-size fixtures K1 / T1 / P1 (BASELINE.json configs[2], [3], [4] shapes) -----------------

def _inputs():
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import inputs
    return inputs


def test_k1_kitti_oracle_forward_backward():
    """K1: the oracle (torch restatement) reproduces the reference at 47x156 with fractional
    coordinates, forward and gradients; the pixel-coordinate float64 restatement stays inside the
    1e-5 bar there (the reference's normalise -> denormalise round trip grows with W)."""
    I = _inputs()
    g = load_golden("K1_kitti")
    f1n, f2n, con, goutn = I.k1_inputs()
    f1, f2, co = _t(f1n, True), _t(f2n, True), _t(con, True)
    out = O.CorrBlockOracle(f1, f2, I.L, I.R)(co)
    H, W = f1n.shape[-2:]
    rows = g["rows"]
    assert rel_err(out.detach()[0].reshape(-1, H * W)[:, rows].T.numpy(), g["out_rows"]) <= 2e-6
    out.backward(_t(goutn))
    for name, t in (("dfmap1", f1.grad), ("dfmap2", f2.grad)):
        err = np.abs(t.numpy().reshape(-1)[g["gidx"]] - g[name + "_s"]).max() / float(g[name + "_max"])
        assert err <= 2e-6, (name, err)
        assert abs(t.double().abs().sum().item() - float(g[name + "_abs"])) <= 1e-6 * float(g[name + "_abs"])
    assert rel_err(co.grad.numpy(), g["dcoords"]) <= 2e-5
    # pixel-coordinate float64 restatement on the sampled queries only (cheap)
    pyr = O.build_pyramid_direct(f1n, f2n, I.L)
    xy = con.transpose(0, 2, 3, 1).reshape(1, H * W, 2)[:, rows]
    sub = [p[:, rows] for p in pyr]
    direct = O.lookup_direct(sub, xy, I.R, per_level_scale=True)[0]
    assert rel_err(direct, g["out_rows"]) <= 1e-5


def test_p1_1080p_direct_restatement():
    """P1: 128 sampled queries of the 136x240 map against the float64 pixel-coordinate restatement:
    the reference's coordinate round trip at W = 240 stays inside the 1e-5 bar."""
    I = _inputs()
    g = load_golden("P1_1080p")
    f1n, f2n, con, qs = I.p1_inputs()
    assert np.array_equal(qs, g["queries"])
    B, D, H, W = f1n.shape
    f1q = f1n.reshape(D, H * W)[:, qs].astype(np.float64)
    vol = (f1q.T @ f2n.reshape(D, H * W).astype(np.float64) / np.sqrt(D)).reshape(1, len(qs), H, W)
    pyr = [vol]
    for _ in range(I.L - 1):
        v = pyr[-1]
        h2, w2 = v.shape[2] // 2, v.shape[3] // 2
        v = v[:, :, :2 * h2, :2 * w2]
        pyr.append(0.25 * (v[:, :, 0::2, 0::2] + v[:, :, 0::2, 1::2] + v[:, :, 1::2, 0::2] + v[:, :, 1::2, 1::2]))
    xy = con[0].reshape(2, H * W)[:, qs].T[None]
    direct = O.lookup_direct(pyr, xy, I.R, per_level_scale=True)[0]
    err = rel_err(direct, g["out_rows"])
    assert err <= 1e-5, err
