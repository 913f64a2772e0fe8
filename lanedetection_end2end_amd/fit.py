"""Differentiable weighted-least-squares lane fit: module surface of the reference's
``Weighted_least_squares`` (BEV/Networks/LSQ_layer.py:90-167, BP/Networks/LSQ_layer.py:72-154)
on top of the fused HIP kernels (``lf_wls_fwd`` / ``lf_wls_bwd``).
"""
import torch
import torch.nn as nn

from . import ops


def fit_lanes(logits, grid, zero_rows=0, order=2, reg_ls=0.0, y_offset=1.0, activation="square",
              use_cholesky=False, return_masked=True, check_singular=True):
    """Fused activation -> row mask -> normal equations -> solve.

    logits (N,K,H,W) fp32 on the GPU, grid (H*W,2) or (N,H*W,2) fp32.
    Returns (beta (N,K,order+1) fp64, masked (N,K,H,W) fp32 or None, status (N*K) int32).
    Raises RuntimeError for a singular system when ``check_singular`` (one D2H sync, like the
    reference's torch.inverse); with check_singular=False inspect ``status`` yourself.
    """
    return ops.WLSFit.apply(logits, grid, int(zero_rows), int(order), float(reg_ls), float(y_offset),
                            ops.ACT_KINDS[activation], 1 if use_cholesky else 0, bool(return_masked),
                            bool(check_singular))


class WeightedLeastSquares(nn.Module):
    """Same constructor / forward contract as the reference's ``Weighted_least_squares``:
    ``forward(W, grid) -> (beta0, beta1, beta2, beta3)`` with W the (already activated and
    masked) weight maps viewable as (N, nclasses, P) and grid (N, P, 2).  beta_k is
    (N, order+1, 1); beta2/beta3 are None unless nclasses > 3 (LSQ_layer.py:104,152).
    """
    y_offset = 1.0          # BEV: y = 1 - grid_y (LSQ_layer.py:109)
    max_order = 2           # BEV implements orders 0..2 (:110-118)
    out_dtype = torch.float32
    cholesky_drops_reg = False   # BEV regularises before either factorisation (LSQ_layer.py:120-126)

    def __init__(self, size, nclasses, order, no_cuda=False, reg_ls=0, use_cholesky=False):
        super().__init__()
        if order < 0 or order > self.max_order:
            raise NotImplementedError(
                'Requested order {} for polynomial fit is not implemented'.format(order))
        self.size = tuple(size)
        self.nclasses = nclasses
        self.order = order
        self.reg_ls_value = float(reg_ls)
        self.use_cholesky = bool(use_cholesky)
        self.check_singular = True

    def forward(self, W, grid):
        P = grid.size(1)
        Wm = W.reshape(-1, self.nclasses, 1, P)
        reg = 0.0 if (self.use_cholesky and self.cholesky_drops_reg) else self.reg_ls_value
        beta, _, _ = fit_lanes(Wm, grid[: Wm.size(0)], 0, self.order, reg, self.y_offset, "none",
                               self.use_cholesky, return_masked=False, check_singular=self.check_singular)
        return split_lanes(beta, self.nclasses, self.out_dtype)


def split_lanes(beta, nclasses, dtype):
    """(N,K,D) fp64 -> the reference's 4-tuple of (N,D,1) tensors (None for absent lanes)."""
    # (one conversion and ONE unbind: its backward is a single stack -- per-lane slices cost a zero fill, a copy and an add per
    # lane and step in autograd, ~4 us each on a stream that has no gaps to hide them in)
    outs = [b.unsqueeze(2) for b in torch.unbind(beta.to(dtype), 1)][:4]
    if nclasses <= 3:
        outs = outs[:2]
    while len(outs) < 4:
        outs.append(None)
    return tuple(outs)
