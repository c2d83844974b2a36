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


def test_level_layout_matches_avg_pool_floor_and_is_a_bijection():
    import numpy as np
    from raft_amd import _lib
    for (H, W, L) in [(55, 128, 4), (47, 156, 4), (50, 90, 4), (136, 240, 4), (13, 17, 2), (9, 9, 3),
                      (40, 72, 6)]:
        lay = _lib.level_layout(H, W, L)
        x = torch.zeros(1, 1, H, W)
        exp = []
        for _ in range(L):
            exp.append(tuple(x.shape[-2:]))
            if min(x.shape[-2:]) >= 2:
                x = torch.nn.functional.avg_pool2d(x, 2, stride=2)
            else:
                x = x[..., : x.shape[-2] // 2, : x.shape[-1] // 2]
        assert [(h, w) for h, w, _ in lay] == exp == _lib.level_dims(H, W, L)
        for l, (h, w, slab) in enumerate(lay):
            pos = _lib.level_positions(H, W, l)
            assert pos.shape == (h * w,)
            assert len(np.unique(pos)) == h * w and (h * w == 0 or (pos.min() >= 0 and pos.max() < slab))
        if L >= 4:      # pooled cells sit where the build kernel's epilogue and the fold put them
            p = [_lib.level_positions(H, W, l).reshape(H >> l, W >> l) for l in range(4)]
            PW = (W + 15) // 16
            for y in range(0, (H >> 3) << 3, 3):
                for xx in range(0, (W >> 3) << 3, 5):
                    c = int(p[0][y, xx]); patch, n, j = c >> 7, (c >> 5) & 3, c & 31
                    dy1, dx1, dx2 = (j >> 2) & 1, (j >> 3) & 1, (j >> 4) & 1
                    ph, pwi = divmod(patch, PW)
                    assert patch * 32 + n * 8 + (dx1 | dy1 << 1 | dx2 << 2) == p[1][y // 2, xx // 2]
                    assert patch * 8 + (dx2 | (n >> 1) << 1 | (n & 1) << 2) == p[2][y // 4, xx // 4]
                    assert ((ph >> 1) * PW + pwi) * 4 + (n & 1) + (ph & 1) * 2 == p[3][y // 8, xx // 8]
            for l in range(4):      # every even-aligned 2x2 block = 4 consecutive, 16-byte aligned cells
                h, w = H >> l, W >> l
                for y in range(0, h - 1, 2):
                    for xx in range(0, w - 1, 2):
                        b = int(p[l][y, xx])
                        assert b % 4 == 0
                        assert [int(p[l][y, xx + 1]), int(p[l][y + 1, xx]), int(p[l][y + 1, xx + 1])] == [b + 1, b + 2, b + 3]


def test_c_abi_rejects_bad_arguments_without_touching_the_gpu():
    from raft_amd import _lib
    lib = _lib.load()
    ptrs = (ctypes.c_void_p * 4)()
    assert lib.raftcorr_build_pyramid(None, None, 1, 256, 55, 128, 4, ptrs, None, 0, None) == -1
    assert b"null" in lib.raftcorr_last_error()
    assert lib.raftcorr_build_pyramid(8, 8, 0, 256, 55, 128, 4, ptrs, None, 0, None) == -1
    assert lib.raftcorr_build_pyramid(8, 8, 1, 256, 55, 128, 9, ptrs, None, 0, None) == -1
    assert lib.raftcorr_build_workspace_bytes(1, 256, 55, 128) == 6 * 256 * 128 * (55 + 56) + 1024
    assert lib.raftcorr_lookup_fwd(ptrs, 1, 55, 128, 4, -1, 8, 0, 8, None) == -1
    some = (ctypes.c_void_p * 4)(8, 8, 8, 8)
    assert lib.raftcorr_lookup_fwd(some, 1, 6, 8, 4, 4, 8, 0, 8, None) == -1        # level 3 of a 6x8 map is empty
    assert b"empty" in lib.raftcorr_last_error()
    assert lib.raftcorr_lookup_fwd(ptrs, 1, 55, 128, 4, 4, 8, 7, 8, None) == -1
    assert lib.raftcorr_alt_forward(8, 8, 8, 8, 1, 1, 4, 4, 4, 4, 0, 4, None) == -1
    assert lib.raftcorr_alt_forward(8, 8, 8, 8, 1, 1, 4, 4, 4, 4, 16, 8, None) == -2   # radius > 7
    # the entry points of round 2: sizes are plain host arithmetic, bad arguments come back as errors
    assert lib.raftcorr_alt_prepare_workspace_bytes(1, 256, 47, 156, 4) >= 4 * 256 * (23 * 78 + 11 * 39 + 5 * 19)
    assert lib.raftcorr_alt_prepare_workspace_bytes(1, 256, 47, 156, 9) == 0
    assert lib.raftcorr_alt_prepare(None, None, 1, 256, 47, 156, 4, None, None, None, None, None, None, 0, None) == -1
    assert lib.raftcorr_alt_features_bwd(8, some, 1, 8, 6, 8, 4, 8, 8, None) == -1      # level 3 of a 6x8 map is empty
    assert lib.raftcorr_alt_lookup_fwd_sorted_workspace_bytes(1, 47, 156, 4) > 4 * 64 * ((47 * 156 + 63) // 64)     # one list of groups for all levels
    assert lib.raftcorr_alt_lookup_fwd_sorted(8, some, 8, 1, 47, 156, 256, 4, 4, 8, 8, None, 0, None) == -1
    assert b"null" in lib.raftcorr_last_error()
    assert lib.raftcorr_alt_lookup_bwd(8, some, 8, 1, 47, 156, 256, 4, 4, 8, 8, 8, some, 0, None, 0, None) == -1
    assert b"workspace" in lib.raftcorr_last_error()


def test_host_classes_validate_before_any_launch():
    from raft_amd import CorrBlock, AlternateCorrBlock, alt_cuda_corr
    f = torch.zeros(1, 8, 6, 6)
    with pytest.raises(RuntimeError, match="fmap1 must be a CUDA tensor"):
        CorrBlock(f, f)
    with pytest.raises(RuntimeError, match="fmap1 must be a CUDA tensor"):
        AlternateCorrBlock(f, f)
    with pytest.raises(RuntimeError, match="tiles must be"):
        AlternateCorrBlock(f, f, tiles="sorted")
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


def test_fast_call_paths_reach_the_same_entry_point():
    """The per-iteration lookup call has two host paths (CPython glue raft_amd/_pyfast.so and an
    argtypes-free ctypes handle); both must hand their arguments to raftcorr_lookup_fwd unchanged.
    Checked without a GPU through the argument validation (radius < 0 is rejected before any launch)."""
    import ctypes
    from raft_amd import _lib, _pyfast
    lib = _lib.load()
    ptrs = (ctypes.c_void_p * 4)(16, 32, 48, 64)
    rc = _pyfast.lookup_fwd(_lib.fn_addr("raftcorr_lookup_fwd"), ctypes.addressof(ptrs), 1, 55, 128, 4, -3,
                            4096, _lib.MODE_CANONICAL, 8192, None)
    assert rc == -1 and b"radius = -3" in lib.raftcorr_last_error()
    c = ctypes.c_int
    rc = _lib.fast("raftcorr_lookup_fwd")(ptrs, c(1), c(55), c(128), c(4), c(-5), ctypes.c_void_p(4096),
                                          _lib.C_MODE[_lib.MODE_CANONICAL], ctypes.c_void_p(8192), ctypes.c_void_p(0))
    assert rc == -1 and b"radius = -5" in lib.raftcorr_last_error()
    # a wrong level count reaches the same check through both (argument order intact)
    rc = _pyfast.lookup_fwd(_lib.fn_addr("raftcorr_lookup_fwd"), ctypes.addressof(ptrs), 1, 55, 128, 9, 4,
                            4096, _lib.MODE_CANONICAL, 8192, 0)
    assert rc == -1 and b"num_levels = 9" in lib.raftcorr_last_error()
