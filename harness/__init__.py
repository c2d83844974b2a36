"""Harness around the correlation path: a stock-PyTorch RAFT-basic model whose
only non-stock part is the correlation block it is given.  Not part of the product package."""
