"""CPU: the --clas heads / lane-decoding oracle vs golden vectors produced by the REAL reference classes
(oracle/gen_golden_clas.py).  Pins oracle/clas_oracle.py."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, relerr
from oracle import clas_oracle
from oracle.gen_golden_clas import clas_inputs, decode_inputs


@pytest.fixture(scope="module")
def golden_clas():
    return np.load(os.path.join(GOLDEN, "clas.npz"), allow_pickle=False)


def _grad_sample(g):
    g = g.numpy()
    return g if g.size <= 20000 else g.reshape(-1)[::97]


@pytest.fixture(scope="module")
def golden_clas_bev():
    return np.load(os.path.join(GOLDEN, "clas_bev.npz"), allow_pickle=False)


@pytest.mark.parametrize("tree,class_type", [("bp", "line"), ("bp", "horizon"), ("bev", "line")])
@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-11), ("f32", torch.float32, 2e-4)])
def test_classification_head(golden_clas, golden_clas_bev, tree, class_type, tag, dtype, tol):
    """bev/line: the BEV tree's four 3-way heads -> (N,3,4) (BEV/Networks/LSQ_layer.py:198-205,218-226), golden from the
    real BEV class (oracle/gen_golden_clas.py -> clas_bev.npz)."""
    if tree == "bev":
        golden_clas = golden_clas_bev
    x, g = clas_inputs(class_type, tree)
    P = clas_oracle.cast_params(clas_oracle.make_clas_params(class_type, seed=7, tree=tree), dtype)
    for k, v in P.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    xt = torch.from_numpy(x).to(dtype).requires_grad_(True)
    stats = {}
    y = clas_oracle.classification_forward(xt, P, class_type, True, stats)
    (y * torch.from_numpy(g).to(dtype)).sum().backward()
    pre = "%s_%s_" % (class_type, tag)
    assert y.shape == golden_clas[pre + "train_out"].shape and (tree != "bev" or y.shape[1:] == (3, 4))
    assert relerr(y.detach().numpy(), golden_clas[pre + "train_out"]) < tol
    assert relerr(xt.grad.numpy()[:, ::8, ::4, ::4], golden_clas[pre + "gx_sample"]) < tol
    for k in ("conv1_bn.running_mean", "conv4_bn.running_var"):
        assert relerr(stats[k].numpy(), golden_clas[pre + k]) < tol
    keys = [str(k) for k in golden_clas[class_type + "_grad_keys"]]
    assert keys == [k for k, v in P.items() if v.requires_grad]
    norms = dict(zip(keys, golden_clas[pre + "grad_norms"]))
    for k in keys:
        if k in ("conv1.bias", "conv2.bias", "conv3.bias", "conv4.bias"):
            # a bias in front of a train-mode BatchNorm has an identically zero gradient: only rounding noise
            assert np.abs(_grad_sample(P[k].grad)).max() < max(tol, 1e-6) * norms[k[:-4] + "weight"], k
            continue
        assert relerr(_grad_sample(P[k].grad), golden_clas[pre + "grad_" + k]) < tol, k
    # eval mode uses the running statistics the train step left behind
    Pe = clas_oracle.cast_params(P, dtype)
    for k, v in stats.items():
        Pe[k] = v
    with torch.no_grad():
        ye = clas_oracle.classification_forward(torch.from_numpy(x).to(dtype), Pe, class_type, False)
    assert relerr(ye.numpy(), golden_clas[pre + "eval_out"]) < tol


@pytest.mark.parametrize("order", [1, 2, 3])
def test_decode_lanes(golden_clas, order):
    beta, line, horizon = decode_inputs(order)
    x = np.stack([clas_oracle.compute_coordinates(beta[:, l]) for l in range(beta.shape[1])], 1)
    assert relerr(x, golden_clas["decode_x_o%d" % order]) < 1e-12
    lanes, ints = clas_oracle.decode_lanes(beta, line, horizon)
    ref = golden_clas["decode_lanes_o%d" % order]
    assert (lanes == -2).sum() > 0 and ((lanes == -2) == (ref == -2)).all()
    assert relerr(lanes, ref) < 1e-12
    assert (ints == golden_clas["decode_int_o%d" % order]).all()
