"""Generate tests/golden/clas.npz by running the REAL reference classes on CPU (authoring container only):
    python -m oracle.gen_golden_clas

* ``Classification('line' | 'horizon')`` (BP/Networks/LSQ_layer.py:150-207): train-mode forward + backward and
  eval-mode forward, fp32 as shipped and fp64 (``.double()``), on seeded inputs / parameters.
* ``Projections.compute_coordinates`` (BP/test.py:128-186) for orders 1..3, and the gating statements of
  ``test_model`` (BP/test.py:72-91), which live inline in that function and are therefore replayed here
  with the same torch statements on the real ``Projections`` output.
Inputs are reproducible from ``clas_inputs`` below, so only the reference's outputs are stored.
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

from . import clas_oracle, ref_shims

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
HEAD_N = 2


def clas_inputs(class_type, tree="bp"):
    """Encoder-like input (post-ReLU, N x 128 x 32 x 64) and the upstream gradient of the logits.
    Seed 83: the smallest |pre-ReLU value| over the four blocks is 1.4e-6 (fp64), clear of fp32 rounding --
    seeds with a 3e-8 near-tie make fp32 and fp64 runs take different ReLU branches at one element.
    ``tree="bev"``: the BEV line head's logits are (N, 3, 4)."""
    rng = np.random.default_rng(83)
    x = np.maximum(rng.standard_normal((HEAD_N, 128, 32, 64)), 0).astype(np.float32)
    nout = 4 if class_type == "line" else 256
    g = rng.standard_normal((HEAD_N, nout)).astype(np.float32)
    if tree == "bev" and class_type == "line":
        g = np.random.default_rng(84).standard_normal((HEAD_N, 3, 4)).astype(np.float32)
    return x, g


def decode_inputs(order, N=6, L=4):
    """Plausible BP-space lane polynomials (x' in [0, 512) over y_eval in [0, 255]) + head outputs."""
    rng = np.random.default_rng(90 + order)
    beta = np.zeros((N, L, order + 1))
    beta[..., -1] = rng.uniform(60, 450, (N, L))
    if order >= 1:
        beta[..., -2] = rng.uniform(-0.6, 0.6, (N, L))
    if order >= 2:
        beta[..., -3] = rng.uniform(-2e-3, 2e-3, (N, L))
    if order >= 3:
        beta[..., -4] = rng.uniform(-4e-6, 4e-6, (N, L))
    line = (rng.uniform(0, 1, (N, 4)) > 0.3).astype(np.float32)
    horizon = (rng.integers(15, 30, N) * 10).astype(np.int32)
    return beta, line, horizon


def gen_heads(ref, out, tree="bp"):
    """``tree="bev"``: the BEV tree's class (BEV/Networks/LSQ_layer.py:170-228) -- only its line head differs from BP's
    (four ``Linear(128, 3)`` -> (N,3,4)), so only that one is stored (-> tests/golden/clas_bev.npz)."""
    for class_type in (("line",) if tree == "bev" else ("line", "horizon")):
        x, g = clas_inputs(class_type, tree)
        for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            P = clas_oracle.make_clas_params(class_type, seed=7, tree=tree)
            m = ref.LSQ_layer.Classification(class_type, size=(32, 64), channels_in=128, resize=256)
            assert list(m.state_dict().keys()) == list(P.keys())
            m.load_state_dict(P)
            m = m.to(dtype).train()
            xt = torch.from_numpy(x).to(dtype).requires_grad_(True)
            y = m(xt)
            (y * torch.from_numpy(g).to(dtype)).sum().backward()
            pre = "%s_%s_" % (class_type, tag)
            out[pre + "train_out"] = y.detach().numpy()
            out[pre + "gx_sample"] = xt.grad.numpy()[:, ::8, ::4, ::4].copy()
            out[pre + "gx_norm"] = np.array(float(xt.grad.double().norm()))
            sd = m.state_dict()
            for k in ("conv1_bn.running_mean", "conv4_bn.running_var"):
                out[pre + k] = sd[k].numpy().copy()
            names, norms = [], []
            for k, p in m.named_parameters():
                names.append(k)
                norms.append(float(p.grad.double().norm()))
                gnp = p.grad.numpy()
                out[pre + "grad_" + k] = (gnp if gnp.size <= 20000 else gnp.reshape(-1)[::97]).copy()
            out[pre + "grad_norms"] = np.array(norms)
            if tag == "f32":
                out[class_type + "_grad_keys"] = np.array(names)
            m.eval()
            with torch.no_grad():
                out[pre + "eval_out"] = m(torch.from_numpy(x).to(dtype)).numpy()


def _load_test_module():
    """BP/test.py with its unavailable imports stubbed (ujson -> json; cv2 stub from ref_shims)."""
    sys.modules.setdefault("ujson", json)
    return importlib.import_module("test")


def gen_decode(out):
    ref_shims.load("bp")                       # puts the BP tree on sys.path, installs the cv2 stub
    test_mod = _load_test_module()
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self          # Projections.__init__ moves its constants to the GPU
    try:
        for order in (1, 2, 3):
            beta, line, horizon = decode_inputs(order)
            N, L, _ = beta.shape
            opts = types.SimpleNamespace(resize=256, order=order, batch_size=N)
            params = test_mod.Projections(opts)
            xs = [params.compute_coordinates(torch.from_numpy(beta[:, l, :, None])) for l in range(L)]
            lanes_pred = torch.stack(xs, dim=1)
            out["decode_x_o%d" % order] = lanes_pred.numpy().copy()
            # replay of BP/test.py:77-91 on the reference's own coordinates
            line_pred = torch.from_numpy(line)[:, [1, 2, 0, 3]]
            lanes_pred[(1 - line_pred[:, :, None]).bool().expand_as(lanes_pred)] = -2
            bounds = torch.div(torch.from_numpy(horizon) - 160, 10, rounding_mode="trunc")
            for k, bound in enumerate(bounds):
                lanes_pred[k, :, :bound.item()] = -2
            lanes_pred[lanes_pred > 1279] = -2
            lanes_pred[lanes_pred < 0] = -2
            out["decode_lanes_o%d" % order] = lanes_pred.numpy().copy()
            out["decode_int_o%d" % order] = np.int_(np.round(lanes_pred.numpy()))
    finally:
        torch.Tensor.cuda = orig_cuda


def main():
    assert ref_shims.available(), "needs /root/reference"
    torch.set_num_threads(os.cpu_count())
    out = {}
    gen_heads(ref_shims.load("bp"), out)
    gen_decode(out)
    path = os.path.join(OUT, "clas.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024), len(out), "arrays")
    out = {}
    gen_heads(ref_shims.load("bev"), out, tree="bev")
    path = os.path.join(OUT, "clas_bev.npz")
    np.savez_compressed(path, **out)
    print(path, "%.1f KB" % (os.path.getsize(path) / 1024), len(out), "arrays")


if __name__ == "__main__":
    sys.exit(main())
