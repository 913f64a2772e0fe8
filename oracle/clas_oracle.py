"""CPU restatement of the ``--clas`` heads and the test-time lane decoding.

TEST INFRASTRUCTURE ONLY: imported by ``tests/`` (and ``oracle/gen_golden_clas.py``) as the checker, never by
the product.  Pinned by ``tests/golden/clas.npz``, generated from the real reference classes
(``Classification`` BP/Networks/LSQ_layer.py:150-207, ``Projections`` BP/test.py:128-186) by
``python -m oracle.gen_golden_clas``.

``classification_forward`` is functional torch (fp32 or fp64, CPU); its backward is torch autograd of the
same expression, like erfnet_oracle.  ``decode_lanes`` is numpy fp64.
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import fit_oracle

BN_EPS, BN_MOM = 1e-5, 0.1          # nn.BatchNorm2d defaults (LSQ_layer.py:159,165,169,173)


def clas_param_spec(class_type, rows=32, cols=64, channels_in=128, resize=256, tree="bp"):
    """state_dict keys/shapes of ``Classification`` in registration order (BP/Networks/LSQ_layer.py:152-190; ``tree="bev"``:
    BEV/Networks/LSQ_layer.py:172-207, whose line head is four ``Linear(128, 3)``)."""
    spec = OrderedDict()
    for name, ci, co, k in (("conv1", channels_in, 128, 1), ("conv2", 128, 128, 3), ("conv3", 128, 64, 3),
                            ("conv4", 64, 64, 3)):
        spec[name + ".weight"] = (co, ci, k, k)
        spec[name + ".bias"] = (co,)
        spec[name + "_bn.weight"] = (co,)
        spec[name + "_bn.bias"] = (co,)
        spec[name + "_bn.running_mean"] = (co,)
        spec[name + "_bn.running_var"] = (co,)
        spec[name + "_bn.num_batches_tracked"] = ()
    if class_type == "line":
        spec["fully_connected1.weight"] = (128, 64 * rows * cols // 4)
        spec["fully_connected1.bias"] = (128,)
        if tree == "bev":
            for i in range(1, 5):
                spec["fully_connected_line%d.weight" % i] = (3, 128)
                spec["fully_connected_line%d.bias" % i] = (3,)
        else:
            spec["fully_connected_line1.weight"] = (4, 128)
            spec["fully_connected_line1.bias"] = (4,)
    else:
        spec["fully_connected_horizon.weight"] = (resize, 64 * rows)
        spec["fully_connected_horizon.bias"] = (resize,)
    return spec


def make_clas_params(class_type, seed=0, **kw):
    """Seeded fp32 parameters (He-scaled weights, non-trivial BN affine / running stats)."""
    rng = np.random.default_rng(seed)
    P = OrderedDict()
    for k, shp in clas_param_spec(class_type, **kw).items():
        if k.endswith("num_batches_tracked"):
            P[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_var"):
            P[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        elif k.endswith("running_mean"):
            P[k] = torch.from_numpy((0.1 * rng.standard_normal(shp)).astype(np.float32))
        elif "_bn.weight" in k:
            P[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        elif k.endswith(".bias"):
            P[k] = torch.from_numpy((0.05 * rng.standard_normal(shp)).astype(np.float32))
        else:
            fan_in = int(np.prod(shp[1:]))
            P[k] = torch.from_numpy((rng.standard_normal(shp) * np.sqrt(2.0 / fan_in)).astype(np.float32))
    return P


def _block(x, P, name, training, stats_out, pre_out=None, flips=None):
    """relu(bn(conv(x))) -- LSQ_layer.py:193-196.  ``pre_out``: dict filled with the ReLU's pre-activation (detached);
    ``flips``: {name: flat indices} whose ReLU derivative mask is inverted -- an fp32 implementation decides the sign of a
    pre-activation that lies inside fp32 rounding of zero differently from fp64, and a test may ask for the fp64 gradient under
    either decision of such an element (tests/test_clas_gpu.py)."""
    w = P[name + ".weight"]
    z = F.conv2d(x, w, P[name + ".bias"], stride=1, padding=(w.shape[2] - 1) // 2)
    if training:
        mean = z.mean((0, 2, 3))
        var = z.var((0, 2, 3), unbiased=False)
        if stats_out is not None:
            n = z.numel() // z.shape[1]
            stats_out[name + "_bn.running_mean"] = ((1 - BN_MOM) * P[name + "_bn.running_mean"].to(z.dtype)
                                                    + BN_MOM * mean).detach()
            stats_out[name + "_bn.running_var"] = ((1 - BN_MOM) * P[name + "_bn.running_var"].to(z.dtype)
                                                   + BN_MOM * var * n / (n - 1)).detach()
    else:
        mean, var = P[name + "_bn.running_mean"].to(z.dtype), P[name + "_bn.running_var"].to(z.dtype)
    zh = (z - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
    a = zh * P[name + "_bn.weight"][None, :, None, None] + P[name + "_bn.bias"][None, :, None, None]
    if pre_out is not None:
        pre_out[name] = a.detach()
    if flips and name in flips:
        mask = (a > 0).reshape(-1).clone()
        idx = torch.as_tensor(list(flips[name]), dtype=torch.long)
        mask[idx] = ~mask[idx]
        return a * mask.reshape(a.shape).to(a.dtype)
    return F.relu(a)


def classification_trunk(x, P, training=True, stats_out=None, pre_out=None, flips=None):
    for name in ("conv1", "conv2", "conv3", "conv4"):
        x = _block(x, P, name, training, stats_out, pre_out, flips)
    return x


def classification_forward(x, P, class_type, training=True, stats_out=None, pre_out=None, flips=None):
    """x (N,128,rows,cols) -> line logits (N,4) or horizon logits (N,resize) -- BP/Networks/LSQ_layer.py:192-207.
    With the BEV tree's parameters (``fully_connected_line4`` present) the line logits are the four 3-way heads stacked
    to (N,3,4): ``cat((x1,x2,x3,x4), 2)`` of (N,3,1,1) views -- BEV/Networks/LSQ_layer.py:218-226."""
    y = classification_trunk(x, P, training, stats_out, pre_out, flips)
    if class_type == "line":
        f = F.max_pool2d(y, 2, 2).reshape(y.shape[0], -1)
        f = F.relu(F.linear(f, P["fully_connected1.weight"], P["fully_connected1.bias"]))
        if "fully_connected_line4.weight" in P:
            return torch.stack([F.linear(f, P["fully_connected_line%d.weight" % i], P["fully_connected_line%d.bias" % i])
                                for i in range(1, 5)], 2)
        return F.linear(f, P["fully_connected_line1.weight"], P["fully_connected_line1.bias"])
    f = y.mean(3).reshape(y.shape[0], -1)                     # AvgPool2d((1, cols)) with cols == width
    return F.linear(f, P["fully_connected_horizon.weight"], P["fully_connected_horizon.bias"])


def cast_params(P, dtype):
    return OrderedDict((k, v.detach().clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in P.items())


# ---- Projections / test-time decoding --------------------------------------------------------------------
def projection_constants(resize=256):
    """y_d, y_prime, y_eval of Projections.__init__ (BP/test.py:133-146), fp64."""
    M, M_inv = fit_oracle.bp_homography(resize, False)
    y_d = (np.arange(160, 720, 10) - 80).astype(np.float64) / 2.5
    y_prime = (M[1, 1] * y_d + M[1, 2]) / (M[2, 1] * y_d + M[2, 2])
    return M_inv.astype(np.float64), y_prime, 255 - y_prime


def compute_coordinates(beta, resize=256):
    """beta (N, order+1) fp64, highest power first -> x (N, 56) in the 1280-wide frame (BP/test.py:172-186)."""
    M_inv, y_prime, y_eval = projection_constants(resize)
    order = beta.shape[1] - 1
    Y = np.stack([y_eval ** (order - k) for k in range(order + 1)], 1)          # (56, order+1)
    xp = beta @ Y.T
    pts = np.stack([xp, np.broadcast_to(y_prime, xp.shape), np.ones_like(xp)], 1)   # (N, 3, 56)
    t = np.einsum("ij,njs->nis", M_inv, pts)
    return t[:, 0] / t[:, 2] * 2.5


def decode_lanes(betas, line_pred=None, horizon_pred=None, resize=256):
    """The post-processing of test_model (BP/test.py:66-91): betas (N, L, order+1); line_pred (N,4) in the
    dataset's order; horizon_pred (N,) int.  Returns (float lanes, np.int_ rounded lanes)."""
    N, L, _ = betas.shape
    lanes = np.stack([compute_coordinates(betas[:, l], resize) for l in range(L)], 1)
    if line_pred is not None:
        lp = line_pred[:, [1, 2, 0, 3]][:, :L]
        lanes[np.broadcast_to((1 - lp[:, :, None]).astype(bool), lanes.shape)] = -2
    if horizon_pred is not None:
        bounds = np.trunc((horizon_pred.astype(np.int64) - 160) / 10).astype(np.int64)
        for k, b in enumerate(bounds):
            lanes[k, :, :int(b)] = -2
    lanes[lanes > 1279] = -2
    lanes[lanes < 0] = -2
    return lanes, np.int_(np.round(lanes))
