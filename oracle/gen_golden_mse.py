"""Generate tests/golden/mse.npz by running the REAL reference ``MSE_Loss`` (``--loss_policy mse``) on CPU, authoring
container only:  python -m oracle.gen_golden_mse

``MSE_Loss(options).forward(params, gt_params)`` = ``nn.MSELoss()(params.squeeze(-1), gt_params)``
(BEV/Loss_crit.py:137-150; BP/Loss_crit.py:147-160 is the same class).  fp32 as shipped and fp64, loss and gradient, for
(N,3,1) BEV-style and (N,4,1) BP order-3 parameters.  Inputs are reproducible from ``mse_inputs`` (only outputs are stored).
"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch

from . import ref_shims

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def mse_inputs(D):
    rng = np.random.default_rng(40 + D)
    N = 7
    return rng.uniform(-0.5, 0.5, (N, D, 1)), rng.uniform(-0.5, 0.5, (N, D))


def main():
    assert ref_shims.available(), "needs /root/reference"
    out = {}
    for tree in ("bev", "bp"):
        ref = ref_shims.load(tree)
        crit = ref.Loss_crit.MSE_Loss(Namespace(no_cuda=True))
        for D in (3, 4):
            p, g = mse_inputs(D)
            for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
                pt = torch.from_numpy(p).to(dtype).requires_grad_(True)
                L = crit(pt, torch.from_numpy(g).to(dtype))
                L.backward()
                out["%s_d%d_%s_loss" % (tree, D, tag)] = L.detach().numpy()
                out["%s_d%d_%s_grad" % (tree, D, tag)] = pt.grad.numpy()
    path = os.path.join(OUT, "mse.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024), len(out), "arrays")


if __name__ == "__main__":
    sys.exit(main())
