"""The reference's step on the vendor stack: PyTorch-ROCm eager ops (MIOpen convolutions / batch norm, ATen elementwise,
rocBLAS bmm + inverse) -- what a user of the reference gets on an MI355X by calling ``.cuda()`` (BEV/main.py:77-83).

TEST / BASELINE INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): ``bench.py`` times it AFTER its timed region as
``miopen_baseline`` (a second, non-graded baseline beside ``cpu_baseline``); nothing under the package imports it.
/root/reference does not exist on the GPU box, so this is a functional restatement with the very ``torch.nn.functional``
calls the reference's modules make:
  backbone   BEV/Networks/ERFNet.py:11-157    F.conv2d / F.conv_transpose2d / F.max_pool2d / F.batch_norm(eps=1e-3) /
                                              F.dropout2d / F.relu
  fit        BEV/Networks/LSQ_layer.py:66-167,310-326  square activation, index_fill of the top rows, grid = [x,y,1] M^T,
                                              Y = [y^2, y, 1] with y = 1 - grid_y, Z = (W Y)^T (W Y), beta = Z^-1 (W Y)^T (W x) via
                                              torch.bmm and torch.inverse, one lane at a time
  loss       BEV/Loss_crit.py:98-134          closed-form integral of the squared difference, weight 'none', order 2
``tests/test_vendor_baseline_cpu.py`` pins it on CPU against ``erfnet_oracle`` / ``fit_oracle`` (which are pinned against
the real reference's goldens).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import erfnet_oracle, fit_oracle

BN_EPS, BN_MOM = erfnet_oracle.BN_EPS, erfnet_oracle.BN_MOMENTUM


def _bn(x, P, p, training):
    return F.batch_norm(x, P[p + ".running_mean"], P[p + ".running_var"], P[p + ".weight"], P[p + ".bias"], training, BN_MOM, BN_EPS)


def backbone(x, P, training=True, dropout=True):
    """(encoder_output, decoder_output) with nn-module semantics: running statistics are updated in place in train mode."""
    enc, y = None, x
    for p, kind, _, _, pdrop, d in erfnet_oracle.layer_table():
        if p == "decoder.layers.0":
            enc = y
        if kind == "down":
            y = torch.cat([F.conv2d(y, P[p + ".conv.weight"], P[p + ".conv.bias"], stride=2, padding=1), F.max_pool2d(y, 2, stride=2)], 1)
            y = F.relu(_bn(y, P, p + ".bn", training))
        elif kind == "nb1d":
            t = F.relu(F.conv2d(y, P[p + ".conv3x1_1.weight"], P[p + ".conv3x1_1.bias"], padding=(1, 0)))
            t = F.conv2d(t, P[p + ".conv1x3_1.weight"], P[p + ".conv1x3_1.bias"], padding=(0, 1))
            t = F.relu(_bn(t, P, p + ".bn1", training))
            t = F.relu(F.conv2d(t, P[p + ".conv3x1_2.weight"], P[p + ".conv3x1_2.bias"], padding=(d, 0), dilation=(d, 1)))
            t = F.conv2d(t, P[p + ".conv1x3_2.weight"], P[p + ".conv1x3_2.bias"], padding=(0, d), dilation=(1, d))
            t = _bn(t, P, p + ".bn2", training)
            if pdrop > 0 and dropout:
                t = F.dropout2d(t, pdrop, training)
            y = F.relu(t + y)
        else:
            y = F.conv_transpose2d(y, P[p + ".conv.weight"], P[p + ".conv.bias"], stride=2, padding=1, output_padding=1)
            y = F.relu(_bn(y, P, p + ".bn", training))
    return enc, F.conv_transpose2d(y, P["decoder.output_conv.weight"], P["decoder.output_conv.bias"], stride=2)


def bev_grid(resize, device, dtype=torch.float32):
    """(1, H*W, 2) projective grid in normalised coordinates, the fp32 ops of ProjectiveGridGenerator (LSQ_layer.py:66-87)."""
    M, _ = fit_oracle.bev_homography()
    g = fit_oracle.projective_grid(resize, 2 * resize, M.astype(np.float32), True, np.float32)
    return torch.from_numpy(g).to(device=device, dtype=dtype).unsqueeze(0)


def wls_fit(dec, grid, zero_rows):
    """beta per lane, (N, 3, 1) each: the reference's statement sequence for order 2, no regulariser (LSQ_layer.py:103-150)."""
    N, K, H, W = dec.shape
    act = dec ** 2                                             # activation 'square'
    rows = torch.arange(zero_rows, device=dec.device)
    masked = act.index_fill(2, rows, 0)
    g = grid.expand(N, -1, -1)
    x_map = g[:, :, 0:1]
    y_map = 1 - g[:, :, 1:2]
    Y = torch.cat((y_map ** 2, y_map, torch.ones_like(y_map)), 2)
    Wm = masked.reshape(N, K, -1)
    betas = []
    for k in range(K):
        Wk = Wm[:, k, :].unsqueeze(2)
        Yk = Wk * Y
        Z = torch.bmm(Yk.transpose(1, 2), Yk)
        X = torch.bmm(Yk.transpose(1, 2), Wk * x_map)
        betas.append(torch.bmm(torch.inverse(Z), X))
    return betas


def area_loss(beta, gt, t=0.7):
    """Area_Loss(order 2, weight 'none') for one lane: mean over the lanes whose ground truth is not all zero (Loss_crit.py:98-134)."""
    d = beta.squeeze(-1) - gt
    a, b, c = d[:, 0], d[:, 1], d[:, 2]
    v = a * a * t ** 5 / 5 + 2 * a * b * t ** 4 / 4 + (b * b + 2 * a * c) * t ** 3 / 3 + 2 * b * c * t ** 2 / 2 + c * c * t
    keep = (gt != 0).all(1)
    return v[keep].mean() if bool(keep.any()) else v.sum() * 0


def bev_step(x, P, gt, grid, zero_rows, dropout=True):
    """forward + loss + backward of the BEV workload; gradients land in P[k].grad.  Returns (loss, beta0, beta1)."""
    _, dec = backbone(x, P, True, dropout)
    b = wls_fit(dec, grid, zero_rows)
    loss = area_loss(b[0], gt[:, 0]) + area_loss(b[1], gt[:, 1])
    for v in P.values():
        if v.is_floating_point() and v.requires_grad:
            v.grad = None
    loss.backward()
    return loss.detach(), b[0].detach(), b[1].detach()


def trainable_params(seed, device, out_channels=2):
    P = erfnet_oracle.make_params(seed=seed, out_channels=out_channels)
    out = {}
    for k, v in P.items():
        v = v.to(device)
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
        out[k] = v
    return out
