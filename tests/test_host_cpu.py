"""CPU: the C-ABI library loads and exports every symbol include/raftcorr.h declares, the
host-side shape logic matches the reference, and argument validation works (no compute)."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from raft_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "raftcorr.h")).read()
    declared = set(re.findall(r"\b(raftcorr_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().raftcorr_version() == 1


def test_level_dims_match_avg_pool_floor():
    from raft_amd import _lib
    lib = _lib.load()
    for (H, W, L) in [(55, 128, 4), (47, 156, 4), (50, 90, 4), (136, 240, 4), (13, 17, 2), (9, 9, 3)]:
        hl = (ctypes.c_int32 * L)()
        wl = (ctypes.c_int32 * L)()
        assert lib.raftcorr_level_dims(H, W, L, hl, wl) == 0
        x = torch.zeros(1, 1, H, W)
        exp = []
        for _ in range(L):
            exp.append(tuple(x.shape[-2:]))
            x = torch.nn.functional.avg_pool2d(x, 2, stride=2)
        assert [(hl[i], wl[i]) for i in range(L)] == exp == _lib.level_dims(H, W, L)


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    from raft_amd import _lib
    lib = _lib.load()
    ptrs = (ctypes.c_void_p * 4)()
    assert lib.raftcorr_build_pyramid(None, None, 1, 256, 55, 128, 4, ptrs, None) == -1
    assert b"null" in lib.raftcorr_last_error()
    assert lib.raftcorr_build_pyramid(8, 8, 0, 256, 55, 128, 4, ptrs, None) == -1
    assert lib.raftcorr_build_pyramid(8, 8, 1, 256, 55, 128, 9, ptrs, None) == -1
    assert lib.raftcorr_lookup_fwd(ptrs, 1, 55, 128, 4, -1, 8, 0, 8, None) == -1
    assert lib.raftcorr_lookup_fwd(ptrs, 1, 55, 128, 4, 4, 8, 7, 8, None) == -1
    assert lib.raftcorr_alt_forward(8, 8, 8, 8, 1, 1, 4, 4, 4, 4, 0, 4, None) == -1
    assert lib.raftcorr_alt_forward(8, 8, 8, 8, 1, 1, 4, 4, 4, 4, 4096, 4, None) == -2


def test_host_classes_validate_before_any_launch():
    from raft_amd import CorrBlock, AlternateCorrBlock, alt_cuda_corr
    f = torch.zeros(1, 8, 6, 6)
    with pytest.raises(RuntimeError, match="fmap1 must be a CUDA tensor"):
        CorrBlock(f, f)
    with pytest.raises(RuntimeError, match="fmap1 must be a CUDA tensor"):
        AlternateCorrBlock(f, f)
    with pytest.raises(RuntimeError, match="fmap1 must be a CUDA tensor"):
        alt_cuda_corr.forward(f, f, torch.zeros(1, 1, 6, 6, 2), 4)


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under raft_amd/ may reference it."""
    pkg = os.path.join(ROOT, "raft_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, fn)).read()
                assert "oracle" not in src.replace("no CPU or PyTorch fallback", ""), os.path.join(dp, fn)


def test_missing_extension_fails_loudly(monkeypatch):
    from raft_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_load_error", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libraftcorr.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
