"""End-to-end CPU oracle steps at arbitrary size (TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``).

One training step of each BASELINE.json workload composed from the pinned pieces of this package:
``erfnet_oracle.erfnet_forward`` (torch CPU, fp32 or fp64, autograd) -> ``fit_oracle`` (numpy fp64: WLS fit with
its analytic backward, area / back-projection / cross-entropy loss) -> ``dec.backward``.  Every piece is checked
against outputs of the real reference by ``tests/test_oracle_golden.py``; the composition itself is checked against
the reference's end-to-end goldens at 4x256x512 (``tests/test_oracle_golden.py::test_e2e_oracle_matches_reference_goldens``).

Used by ``tests/test_baseline_configs_gpu.py`` (the BASELINE configs at their own sizes) and by ``bench.py``'s
``parity`` / ``cpu_baseline`` legs.  Reference call sites: BEV/main.py:213-223,264-265 (model -> Area_Loss per lane ->
backward), BP/main.py:256-263 (early_return + CrossEntropy), :286-305 (backprojection_loss averaged over lanes).

dtype is the BACKBONE dtype ("cpu32" / "cpu64" in the parity triples) and the dtype the projective grid is rounded to
(the reference model cast with .double() builds its grid in fp64, its fp32 self in fp32; the HIP path uses the fp32 grid).
The fit and the losses always run in fp64 on the logits that backbone produced, which makes |cpu32 - cpu64| the
backbone's fp32 noise plus the grid rounding -- a floor at or below the real reference's (whose fp32 fit adds ~2e-5 of
its own).
"""
from collections import OrderedDict

import numpy as np
import torch

from . import erfnet_oracle, fit_oracle


def _trainable(P, dtype):
    Pd = erfnet_oracle.cast_params(P, dtype)
    for k, v in Pd.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    return Pd


def _grad_norms(Pd):
    out = OrderedDict()
    for k, v in Pd.items():
        if v.is_floating_point() and "running" not in k:
            out[k] = None if v.grad is None else float(v.grad.double().norm())
    return out


def _finish(dec, dlogits, Pd, extra):
    dec.backward(torch.from_numpy(dlogits).to(dec.dtype))
    out = dict(logits=dec.detach().double().numpy(), dlogits=dlogits, grad_norms=_grad_norms(Pd))
    out.update(extra)
    return out


def bev_step(x, P, gt, dtype=torch.float64, resize=256, mask_percentage=0.3, order=2, weight_funct="none",
             training=True, keep_masks=None):
    """BEV tree: backbone -> square activation + row mask + WLS (normalised coordinates) -> Area_Loss on lanes 0, 1.
    x (N,3,R,2R) torch fp32; gt (N,4,3) numpy.  Returns dict(beta (N,2,3), loss, logits, dlogits, grad_norms)."""
    Pd = _trainable(P, dtype)
    _, dec = erfnet_oracle.erfnet_forward(x.to(dtype), Pd, training=training, keep_masks=keep_masks)
    M, _ = fit_oracle.bev_homography()
    gdt = np.float64 if dtype == torch.float64 else np.float32
    grid = fit_oracle.projective_grid(resize, 2 * resize, M.astype(np.float32), True, gdt)      # M is fp32 in both (BEV :30)
    zr = fit_oracle.zero_rows_of(resize, mask_percentage)
    c = fit_oracle.wls_forward(dec.detach().numpy(), grid, zr, order, 0.0, 1.0, "square")
    gb = np.zeros_like(c["beta"])
    loss = 0.0
    for k in range(2):
        l, g = fit_oracle.area_loss(c["beta"][:, k], gt[:, k], order, weight_funct)
        loss += l
        gb[:, k] = g
    return _finish(dec, fit_oracle.wls_backward(c, gb), Pd, dict(beta=c["beta"], loss=float(loss)))


def bp_step(x, P, lanes, valid, dtype=torch.float64, resize=256, nclasses=4, mask_percentage=0.2, order=2,
            training=True):
    """BP tree: backbone -> WLS in pixel coordinates (y = 255 - grid_y) -> backprojection_loss averaged over the lanes.
    The grid is sanitised on masked rows (pole of the homography at 320x640, SURVEY.md section 7: the reference
    returns NaN there; the HIP kernels never read masked rows).  Returns beta (N,K,d+1), x_cal (N,K,56), loss, ..."""
    Pd = _trainable(P, dtype)
    _, dec = erfnet_oracle.erfnet_forward(x.to(dtype), Pd, training=training)
    M, _ = fit_oracle.bp_homography(resize)
    # the BP model computes its grid ONCE, in fp32, at construction (BP/Networks/LSQ_layer.py:219,231): an fp64 run of the
    # reference uses that same fp32 grid widened
    grid = fit_oracle.projective_grid(resize, 2 * resize, M.astype(np.float32), False, np.float32).astype(np.float64)
    grid[~np.isfinite(grid)] = 0.0
    zr = fit_oracle.zero_rows_of(resize, mask_percentage)
    c = fit_oracle.wls_forward(dec.detach().numpy(), grid, zr, order, 0.0, 255.0, "square")
    setup = fit_oracle.backproj_setup(order, resize)
    gb = np.zeros_like(c["beta"])
    loss, xc = 0.0, []
    for k in range(nclasses):
        l, x_cal, g = fit_oracle.backproj_loss(c["beta"][:, k], lanes[:, k], valid[:, k], setup)
        loss += l / nclasses
        gb[:, k] = g / nclasses
        xc.append(x_cal)
    return _finish(dec, fit_oracle.wls_backward(c, gb), Pd, dict(beta=c["beta"], x_cal=np.stack(xc, 1), loss=float(loss)))


def seg_step(x, P, target, dtype=torch.float64, nclasses=2, weight_seg=30.0, training=True):
    """Segmentation branch (end_to_end=False, early_return): backbone with nclasses+1 outputs -> class-weighted CE,
    weights [1] + [weight_seg] * nclasses (BP/Loss_crit.py:64-65; --weight_seg defaults to 30, Networks/utils.py:75)."""
    Pd = _trainable(P, dtype)
    _, dec = erfnet_oracle.erfnet_forward(x.to(dtype), Pd, training=training)
    wts = np.array([1.0] + [weight_seg] * nclasses)
    loss, grad = fit_oracle.cross_entropy_2d(dec.detach().numpy(), target, wts)
    return _finish(dec, grad, Pd, dict(loss=float(loss)))


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def triple(hip, ref32, ref64):
    """(|hip - cpu64|, |hip - cpu32|, |cpu32 - cpu64|), each relative to max |cpu64|."""
    return relerr(hip, ref64), relerr(hip, ref32), relerr(ref32, ref64)
