"""ctypes wrapper around oracle/liboracle_altcorr.so.  TEST INFRASTRUCTURE ONLY.

Exposes the reference's pybind surface (alt_cuda_corr/correlation.cpp:51-54) on the CPU:
``forward(fmap1, fmap2, coords, radius) -> [corr]`` and
``backward(fmap1, fmap2, coords, corr_grad, radius) -> [fmap1_grad, fmap2_grad, coords_grad]``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_altcorr.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        fp = ctypes.POINTER(ctypes.c_float)
        _lib.alt_corr_forward_ref.argtypes = [fp] * 4 + [ctypes.c_int] * 8
        _lib.alt_corr_forward_ref.restype = None
        _lib.alt_corr_backward_ref.argtypes = [fp] * 7 + [ctypes.c_int] * 8
        _lib.alt_corr_backward_ref.restype = None
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def forward(fmap1, fmap2, coords, radius):
    lib = _load()
    f1, f2, co = _c(fmap1), _c(fmap2), _c(coords)
    B, H1, W1, C = f1.shape
    _, H2, W2, _ = f2.shape
    N = co.shape[1]
    rd = 2 * radius + 1
    corr = np.empty((B, N, rd * rd, H1, W1), dtype=np.float32)
    lib.alt_corr_forward_ref(_p(f1), _p(f2), _p(co), _p(corr), B, N, H1, W1, H2, W2, C, radius)
    return [corr]


def backward(fmap1, fmap2, coords, corr_grad, radius):
    lib = _load()
    f1, f2, co, cg = _c(fmap1), _c(fmap2), _c(coords), _c(corr_grad)
    B, H1, W1, C = f1.shape
    _, H2, W2, _ = f2.shape
    N = co.shape[1]
    g1 = np.empty_like(f1)
    g2 = np.empty_like(f2)
    gc = np.empty_like(co)
    lib.alt_corr_backward_ref(_p(f1), _p(f2), _p(co), _p(cg), _p(g1), _p(g2), _p(gc),
                              B, N, H1, W1, H2, W2, C, radius)
    return [g1, g2, gc]
