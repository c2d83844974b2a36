"""raft_amd -- MI355X-native correlation hot path of RAFT (drop-in for core/corr.py)."""
from .corr import CorrBlock, AlternateCorrBlock   # noqa: F401
from . import alt_cuda_corr                                          # noqa: F401

__all__ = ["CorrBlock", "AlternateCorrBlock", "alt_cuda_corr"]
