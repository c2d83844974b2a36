"""Drop-in for the reference's compiled ``alt_cuda_corr`` module
(alt_cuda_corr/correlation.cpp:51-54): ``forward`` / ``backward`` with the same arguments,
layouts, return lists and error behaviour, running on libraftcorr.so.

    import raft_amd.alt_cuda_corr as alt_cuda_corr     # instead of the CUDA extension
    corr, = alt_cuda_corr.forward(fmap1_nhwc, fmap2_nhwc, coords, radius)     # corr.py:87
"""
import torch

from . import _lib


def _check_input(**tensors):
    dev = None
    for name, t in tensors.items():
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")       # CHECK_CUDA, correlation.cpp:19
        if not t.is_contiguous():
            raise RuntimeError(f"{name} must be contiguous")          # CHECK_CONTIGUOUS, :20
        if t.dtype != torch.float32:
            # the reference's packed_accessor32<float> throws for any other dtype (kernel.cu:279)
            raise RuntimeError(f"{name} must be float32")
        if dev is None:
            dev = t.device
        elif t.device != dev:       # raw pointers below: another GPU's tensor would fault, not raise
            raise RuntimeError(f"{name} is on {t.device}, expected {dev}")


def _dims(fmap1, fmap2, coords):
    if fmap1.dim() != 4 or fmap2.dim() != 4 or coords.dim() != 5:
        raise RuntimeError("expected fmap1 (B,H1,W1,C), fmap2 (B,H2,W2,C), coords (B,N,H1,W1,2)")
    B, H1, W1, C = fmap1.shape
    B2, H2, W2, C2 = fmap2.shape
    Bc, N, Hc, Wc, two = coords.shape
    if B2 != B or C2 != C or Bc != B or (Hc, Wc) != (H1, W1) or two != 2:
        raise RuntimeError("alt_cuda_corr: inconsistent shapes")
    return B, N, H1, W1, H2, W2, C


def forward(fmap1, fmap2, coords, radius):
    """-> [corr (B, N, (2r+1)^2, H1, W1)], unscaled dot products (correlation.cpp:23-33)."""
    _check_input(fmap1=fmap1, fmap2=fmap2, coords=coords)
    B, N, H1, W1, H2, W2, C = _dims(fmap1, fmap2, coords)
    lib = _lib.load()
    rd = 2 * radius + 1
    corr = torch.empty(B, N, rd * rd, H1, W1, dtype=torch.float32, device=fmap1.device)
    with torch.cuda.device(fmap1.device):
        _lib.check(lib.raftcorr_alt_forward(fmap1.data_ptr(), fmap2.data_ptr(), coords.data_ptr(),
                                            corr.data_ptr(), B, N, H1, W1, H2, W2, C, radius,
                                            torch.cuda.current_stream(fmap1.device).cuda_stream),
                   "alt_forward")
    return [corr]


def backward(fmap1, fmap2, coords, corr_grad, radius):
    """-> [fmap1_grad, fmap2_grad, coords_grad]; coords_grad is zeros, as in the reference
    (correlation.cpp:36-48, correlation_kernel.cu:307)."""
    _check_input(fmap1=fmap1, fmap2=fmap2, coords=coords, corr_grad=corr_grad)
    B, N, H1, W1, H2, W2, C = _dims(fmap1, fmap2, coords)
    rd = 2 * radius + 1
    if tuple(corr_grad.shape) != (B, N, rd * rd, H1, W1):
        raise RuntimeError(f"corr_grad must be {(B, N, rd * rd, H1, W1)}; got {tuple(corr_grad.shape)}")
    lib = _lib.load()
    g1 = torch.empty_like(fmap1)
    g2 = torch.empty_like(fmap2)
    gc = torch.empty_like(coords)
    with torch.cuda.device(fmap1.device):
        _lib.check(lib.raftcorr_alt_backward(fmap1.data_ptr(), fmap2.data_ptr(), coords.data_ptr(),
                                             corr_grad.data_ptr(), g1.data_ptr(), g2.data_ptr(),
                                             gc.data_ptr(), B, N, H1, W1, H2, W2, C, radius,
                                             torch.cuda.current_stream(fmap1.device).cuda_stream),
                   "alt_backward")
    return [g1, g2, gc]
