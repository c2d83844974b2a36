"""raftcorr_build_pyramid_noalloc (the allocation-free entry point, include/raftcorr.h) through
the C ABI against raftcorr_build_pyramid_prec, the existing entry point it is a variant of: same
kernels, so the pyramids must be bit-identical -- on two calls in a row, the second with a smaller
input (the built-in scratch keeps the first call's contents), and for a shape beyond the built-in
scratch the call must be refused with a clear error.  Collected last (file name) so that a failure
here cannot hide other tests under `pytest -x`."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build_both(lib, _lib, dev, B, D, H, W, L, prec, seed):
    g = torch.Generator().manual_seed(seed)
    f1 = torch.randn(B, D, H, W, generator=g).to(dev)
    f2 = torch.randn(B, D, H, W, generator=g).to(dev)
    layout = _lib.level_layout(H, W, L)
    N = H * W
    stream = torch.cuda.current_stream(dev).cuda_stream

    def levels(fill):
        return [torch.full((B * N * slab,), fill, dtype=torch.float32, device=dev) for (_, _, slab) in layout]

    ref, got = levels(-1.0), levels(-2.0)
    wbytes = lib.raftcorr_build_workspace_bytes(B, D, H, W)
    work = torch.empty(wbytes, dtype=torch.uint8, device=dev)
    rc = lib.raftcorr_build_pyramid_prec(f1.data_ptr(), f2.data_ptr(), B, D, H, W, L, _lib.ptr_array(ref),
                                         work.data_ptr(), wbytes, _lib.PRECISIONS[prec], stream)
    assert rc == 0, lib.raftcorr_last_error()
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated(dev)
    rc = lib.raftcorr_build_pyramid_noalloc(f1.data_ptr(), f2.data_ptr(), B, D, H, W, L, _lib.ptr_array(got),
                                            _lib.PRECISIONS[prec], stream)
    assert rc == 0, lib.raftcorr_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated(dev) == before
    # logical cells only: padding cells hold don't-care values in both
    for l, (h, w, slab) in enumerate(layout):
        pos = torch.from_numpy(_lib.level_positions(H, W, l)).to(dev)
        a = ref[l].view(B * N, slab).index_select(1, pos)
        b = got[l].view(B * N, slab).index_select(1, pos)
        assert torch.equal(a, b), (prec, (B, D, H, W), l)
    return got


def test_noalloc_build_matches_workspace_build_twice_in_a_row(dev):
    from raft_amd import _lib
    lib = _lib.load()
    # first call: Sintel shape (the largest the scratch is sized for together with KITTI), then smaller
    # inputs; every arithmetic option (fp16x2 also uses the range pass' partial maxima in the scratch)
    for prec in ("fp32", "fp16x2", "bf16x3", "bf16"):
        _build_both(lib, _lib, dev, 1, 256, 55, 128, 4, prec, seed=11)
        _build_both(lib, _lib, dev, 2, 40, 13, 19, 3, prec, seed=12)
    _build_both(lib, _lib, dev, 1, 256, 47, 156, 4, "fp32", seed=13)          # KITTI
    # the lookup consumes the pyramid it wrote, as after the workspace build
    from oracle import corr_oracle as O
    B, D, H, W, L, r = 1, 32, 16, 24, 4, 4
    g = torch.Generator().manual_seed(5)
    f1, f2 = torch.randn(B, D, H, W, generator=g), torch.randn(B, D, H, W, generator=g)
    co = torch.rand(B, H * W, 2, generator=g) * torch.tensor([W - 1.0, H - 1.0])
    layout = _lib.level_layout(H, W, L)
    lv = [torch.empty(B * H * W * slab, dtype=torch.float32, device=dev) for (_, _, slab) in layout]
    stream = torch.cuda.current_stream(dev).cuda_stream
    d1, d2, dco = f1.to(dev), f2.to(dev), co.to(dev)
    assert lib.raftcorr_build_pyramid_noalloc(d1.data_ptr(), d2.data_ptr(), B, D, H, W, L, _lib.ptr_array(lv),
                                              0, stream) == 0, lib.raftcorr_last_error()
    out = torch.empty(B, H * W, L * 81, dtype=torch.float32, device=dev)
    assert lib.raftcorr_lookup_fwd(_lib.ptr_array(lv), B, H, W, L, r, dco.data_ptr(), _lib.MODE_FORK,
                                   out.data_ptr(), stream) == 0, lib.raftcorr_last_error()
    ref = O.CorrBlockOracle(f1, f2, L, r)(co).numpy()
    assert np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max() <= 1e-5


def test_noalloc_build_rejects_what_does_not_fit(dev):
    from raft_amd import _lib
    lib = _lib.load()
    B, D, H, W, L = 1, 256, 136, 240, 4                    # 1080p: 108 MB of operand panels
    assert lib.raftcorr_build_workspace_bytes(B, D, H, W) > 32 << 20
    ptrs = (ctypes.c_void_p * L)(256, 256, 256, 256)       # never dereferenced: the size check comes first
    rc = lib.raftcorr_build_pyramid_noalloc(256, 256, B, D, H, W, L, ptrs, 0, None)
    assert rc == -2
    msg = lib.raftcorr_last_error()
    assert b"scratch" in msg and b"33554432" in msg, msg
