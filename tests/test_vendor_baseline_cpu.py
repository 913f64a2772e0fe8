"""CPU: the vendor-stack baseline step (oracle/vendor_baseline.py: the reference's torch.nn.functional calls, timed by
bench.py on the GPU as ``miopen_baseline``) computes what the pinned oracle computes."""
import numpy as np
import torch

from conftest import relerr
from oracle import e2e_oracle, erfnet_oracle, fit_oracle, inputs, vendor_baseline


def test_vendor_baseline_step_matches_the_oracle():
    N, R = 2, 64
    x = torch.from_numpy(inputs.images(N, R, 2 * R, seed=31))
    gt = inputs.bev_gt_params(N, seed=32)
    P32 = erfnet_oracle.make_params(seed=9, out_channels=2)
    ref = e2e_oracle.bev_step(x, P32, gt, torch.float64, R)
    P = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in P32.items()}
    for k, v in P.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    grid = vendor_baseline.bev_grid(R, "cpu", torch.float64)
    zr = fit_oracle.zero_rows_of(R, 0.3)
    loss, b0, b1 = vendor_baseline.bev_step(x.double(), P, torch.from_numpy(gt).double(), grid, zr, dropout=False)
    assert abs(float(loss) - ref["loss"]) < 1e-6 * abs(ref["loss"])      # (the reference's fp32 grid in both)
    beta = torch.stack([b0, b1], 1)[..., 0].numpy()
    assert relerr(beta, ref["beta"]) < 1e-6
    gn = ref["grad_norms"]
    for k in ("encoder.initial_block.conv.weight", "encoder.layers.9.conv3x1_2.weight", "decoder.layers.1.bn1.weight",
              "decoder.output_conv.weight"):
        assert abs(float(P[k].grad.norm()) - gn[k]) < 1e-5 * gn[k], k
    assert P["encoder.output_conv.weight"].grad is None
    # train mode updated the running statistics in place (nn.BatchNorm2d semantics)
    assert float(P["encoder.layers.3.bn1.running_mean"].abs().max()) > 0
