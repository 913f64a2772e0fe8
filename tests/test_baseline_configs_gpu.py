"""The BASELINE.json configurations AT THEIR OWN SIZES against the CPU oracle (fp32 and fp64 backbone legs of
oracle/e2e_oracle.py, whose composition is pinned to the real reference's end-to-end runs by
tests/test_oracle_golden.py::test_e2e_oracle_matches_reference_goldens):

  C2  BEV, 2 lanes, 32 x 3 x 256 x 512, train mode (batch statistics over all 32 images), fp32 matrix cores -- and the same
      step in precision mode fp32x9 with the SHIPPED kernel-selection rule (no test-size override);
  C3  Backprojection tree, 4 lanes, 320 x 640 (batch 4; masked-row pole of the homography sanitised in the oracle only);
  C5  segmentation branch, Cout = 3, 512 x 1024 (batch 2), class-weighted cross entropy.

Reference call sites: BEV/main.py:213-223,264-265; BP/main.py:256-263,286-305.

Criterion (lane coefficients / back-projected x, loss, logits, d loss / d logits): |hip - cpu64| <= 2 * |cpu32 - cpu64| -- the HIP path may be no further from the fp64 truth than twice the distance of
the reference arithmetic's own fp32 run -- with a small absolute floor where the fp32 leg happens to land on the fp64 one.
The norm is the RMS for the tensors (the statistic that is stable when two independent roundoff-noise fields are compared:
both legs are draws of the same noise process through a chaotic train-mode network) and the maximum for the scalars and the
lane coefficients; the maximum over the 1e7..3e7 elements of a tensor is printed too and held to 4x (one extreme sample of
one draw against one extreme sample of another).  All three numbers are printed.  Dropout is off (p = 0): the draw of torch's generator cannot be shared with the oracle at
this size; the masked path is covered in tests/test_backbone_gpu.py.
Parameter gradients (round 3): every tensor against the fp64 oracle evaluated STRAIGHT-THROUGH at the engine's own forward state
and driven by the engine's own d loss / d logits -- backward arithmetic only, held to 5e-5 of each tensor's maximum at these sizes.
"""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import e2e_oracle, erfnet_oracle, inputs

pytestmark = pytest.mark.gpu


def _args(N, R, K, tree, end_to_end=True):
    return Namespace(batch_size=N, nclasses=K, resize=R, end_to_end=end_to_end, mod="erfnet", layers=18, channels_in=3,
                     pretrained=False, pool=True, activation_layer="square", no_cuda=False, order=2, reg_ls=0.0,
                     use_cholesky=False, mask_percentage=0.3 if tree == "bev" else 0.2, clas=False, no_mapping=False,
                     loss_policy="area" if tree == "bev" else "backproject", weight_seg=30, weight_funct="none")


def _prepare(model, P, precision):
    model.net.load_state_dict(P)
    model = model.cuda()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0
    model.net.precision = precision
    return model.train()


def _check(name, hip, r32, r64, abs_floor, rel=True, rms=False):
    hip, r32, r64 = (np.asarray(v, dtype=np.float64) for v in (hip, r32, r64))
    scale = max(np.abs(r64).max(), 1e-300) if rel else 1.0
    e64, e32, floor = np.abs(hip - r64).max() / scale, np.abs(hip - r32).max() / scale, np.abs(r32 - r64).max() / scale
    print("%-28s max: |hip-cpu64| %.3e   |hip-cpu32| %.3e   |cpu32-cpu64| %.3e" % (name, e64, e32, floor))
    if rms:
        q = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2))) / scale
        r64e, r32e, rfl = q(hip, r64), q(hip, r32), q(r32, r64)
        print("%-28s rms: |hip-cpu64| %.3e   |hip-cpu32| %.3e   |cpu32-cpu64| %.3e" % ("", r64e, r32e, rfl))
        assert r64e <= max(2.0 * rfl, abs_floor), (name, "rms", r64e, rfl)
        assert e64 <= max(2.5 * floor, abs_floor), (name, "max", e64, floor)       # max norms: extreme-value statistic (measured <= 1.9)
    else:
        assert e64 <= max(2.0 * floor, abs_floor), (name, e64, floor)
    return e64, floor


def _engine_state(model, logits, N, H, W):
    """Every saved forward tensor of the engine (keyed like the oracle's taps), fetched BEFORE backward releases the workspace."""
    from test_backbone_gpu import fetch_all
    return fetch_all(model.net, model.net._plan(N, H, W), logits.grad_fn.ws, N, H, W)


def _check_grads_straight_through(model, P, x, state, dlogits, tol=3e-5):
    """Every parameter gradient at the configuration's OWN size, sharply: |hip - cpu64| <= tol = 3e-5 of the tensor's maximum.
    The oracle is evaluated straight-through at the engine's forward state -- saved tensors, pool arg-maxes AND the ReLU
    derivative masks the engine's forward pass decided (erfnet_oracle._relu: the saved post-ReLU tensors, and
    sign(fma(t2, scale, shift)) with the engine's folded bn1 vectors for the one ReLU per block whose output is never stored) --
    and driven by the engine's own d loss / d logits, so the difference is backward arithmetic only.  Round 2 held the gradient
    NORMS to max(2 x floor, 2e-2) through the chaotic end-to-end chain, which would pass a wrong scale factor on a small tensor.
    Round 3, first version: same state but the oracle's OWN masks -- errors of 1e-4 (decoder) to 1e-2 (encoder) of the tensor's
    maximum, on the fp32 CPU leg as well, which looked like the accuracy limit of fp32 sums over 2.6e5 terms and was NOT: they
    were single ReLU ties (a pre-activation inside fp32 rounding of zero) decided differently by the two evaluations, each
    carrying its element's whole gradient into every parameter sum below it.  With the masks taken from the state, measured at
    full size: C2 (32 x 256 x 512) worst 6.1e-6 on the fp32 cores and 1.5e-5 in mode fp32x9, C3 4.0e-6, C5 4.9e-6; the CPU's own
    fp32 evaluation of the same sums sits at 0.5-1.2e-6 (it accumulates BatchNorm sums in fp64), printed beside each tensor."""
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    legs = {}
    for dt in (torch.float64, torch.float32):
        Pd = erfnet_oracle.cast_params(P, dt)
        for k, v in Pd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
        _, dec = erfnet_oracle.erfnet_forward(x.to(dt), Pd, training=True, override=state)
        dec.backward(dlogits.to(dt).cpu())
        legs[dt] = Pd
    P64, P32 = legs[torch.float64], legs[torch.float32]
    gmax = max(float(v.grad.abs().max()) for v in P64.values() if v.grad is not None)
    rows = []
    for k, p in model.net.named_parameters():
        g64 = P64[k].grad
        if g64 is None:
            assert p.grad is None, k
            continue
        scale = float(g64.abs().max())
        if scale < 1e-6 * gmax:          # conv biases in front of a train-mode BatchNorm: analytically zero
            assert float(p.grad.abs().max()) < 1e-4 * gmax, k
            continue
        e = float((p.grad.cpu().double() - g64).abs().max()) / scale
        fl = float((P32[k].grad.double() - g64).abs().max()) / scale
        rows.append((e / max(fl, 1e-30), e, fl, k))
    rows.sort(reverse=True)
    ratios = np.array([r[0] for r in rows])
    print("parameter gradients vs the straight-through oracle (|hip-cpu64|, |cpu32-cpu64|, relative to each tensor's max):")
    for r, e, fl, k in rows[:6]:
        print("    ratio %5.2f   hip %.2e   cpu32 %.2e   %s" % (r, e, fl, k))
    print("    worst hip error %.2e (tol %.0e); |hip-cpu64| / |cpu32-cpu64| median %.2f max %.2f over %d tensors"
          % (max(r[1] for r in rows), tol, float(np.median(ratios)), float(ratios.max()), len(ratios)))
    for r, e, fl, k in rows:
        assert e <= tol, (k, e, fl)


_ORACLE_CACHE = {}


def _oracle(key, fn):
    if key not in _ORACLE_CACHE:
        torch.set_num_threads(min(os.cpu_count() or 1, 64))
        _ORACLE_CACHE[key] = (fn(torch.float32), fn(torch.float64))
    return _ORACLE_CACHE[key]


@pytest.mark.parametrize("precision", ["fp32", "fp32x9"])
def test_c2_bev_32x256x512(precision):
    from lanedetection_end2end_amd import _lib
    from lanedetection_end2end_amd.bev.Loss_crit import Area_Loss
    from lanedetection_end2end_amd.bev.Networks.LSQ_layer import Net
    N, R = 32, 256
    P = erfnet_oracle.make_params(seed=4, out_channels=2)
    x = torch.from_numpy(inputs.images(N, R, 2 * R, seed=161))
    gt = inputs.bev_gt_params(N, seed=162)
    o32, o64 = _oracle("c2", lambda dt: e2e_oracle.bev_step(x, P, gt, dt, R))
    lib = _lib.load()
    lib.lf_debug_set_split_any_size(0)            # the shipped selection rule of the split kernels
    try:
        model = _prepare(Net(_args(N, R, 2, "bev")), P, precision)
        crit = Area_Loss(2, "none")
        gtc = torch.from_numpy(gt).cuda()
        b0, b1, _, _, _, _, output, _, _ = model(x.cuda(), True)
        output.retain_grad()
        state = _engine_state(model, output, N, R, 2 * R)
        loss = crit(b0, gtc[:, 0]) + crit(b1, gtc[:, 1])
        loss.backward()
    finally:
        lib.lf_debug_set_split_any_size(1)
    beta = torch.stack([b0, b1], 1)[..., 0].detach().cpu().numpy()
    print("C2 BEV 32x3x256x512, precision %s" % precision)
    _check("lane coefficients (rel)", beta, o32["beta"], o64["beta"], 1e-5)
    _check("loss (rel)", [float(loss)], [o32["loss"]], [o64["loss"]], 1e-5)
    _check("logits (rel)", output.detach().cpu().numpy(), o32["logits"], o64["logits"], 2e-5, rms=True)
    _check("d loss / d logits (rel)", output.grad.cpu().numpy(), o32["dlogits"], o64["dlogits"], 1e-4, rms=True)
    _check_grads_straight_through(model, P, x, state, output.grad)


def test_bev_distance_ratio_over_seeds():
    """the ratio |hip - cpu64| / |cpu32 - cpu64| is a random variable (two fp32 evaluation orders of a
    chaotic train-mode network against one fp64 run), so ONE seed says little.  Six seeds at 8 x 3 x 256 x 512 (the headline's
    geometry and statistics path, a quarter of its batch so that the twelve CPU legs stay within a couple of minutes): the
    distribution of the ratio is printed for the lane coefficients (max norm over 6 numbers per image pair: the noisiest), the
    logits and d loss / d logits (RMS).  Measured in round 3: logits 1.21-1.30 (median 1.24), d loss / d logits 1.20-1.35 (1.22),
    coefficients 1.17-2.05 (1.34): the HIP path is SYSTEMATICALLY ~1.25x further from fp64 than oneDNN's fp32 -- a K = 384 fma
    chain in one accumulator against blocked partial sums; two accumulator sets per output tile would halve the chain but cost 64
    registers (two waves per SIMD instead of three).  Held to: RMS ratios median <= 1.3, max <= 1.5; coefficients median <= 1.5,
    max <= 2.5."""
    from lanedetection_end2end_amd.bev.Loss_crit import Area_Loss
    from lanedetection_end2end_amd.bev.Networks.LSQ_layer import Net
    N, R = 8, 256
    ratios = {"beta": [], "logits": [], "dlogits": []}
    model = None
    for seed in range(6):
        P = erfnet_oracle.make_params(seed=40 + seed, out_channels=2)
        x = torch.from_numpy(inputs.images(N, R, 2 * R, seed=500 + seed))
        gt = inputs.bev_gt_params(N, seed=600 + seed)
        torch.set_num_threads(min(os.cpu_count() or 1, 64))
        o32, o64 = (e2e_oracle.bev_step(x, P, gt, dt, R) for dt in (torch.float32, torch.float64))
        if model is None:
            model = _prepare(Net(_args(N, R, 2, "bev")), P, "fp32")
        else:
            model.net.load_state_dict(P)
        crit = Area_Loss(2, "none")
        gtc = torch.from_numpy(gt).cuda()
        model.zero_grad(set_to_none=True)
        b0, b1, _, _, _, _, output, _, _ = model(x.cuda(), True)
        output.retain_grad()
        (crit(b0, gtc[:, 0]) + crit(b1, gtc[:, 1])).backward()
        beta = torch.stack([b0, b1], 1)[..., 0].detach().cpu().numpy()
        rms = lambda a, b: float(np.sqrt(np.mean((np.asarray(a, dtype=np.float64) - b) ** 2)))
        ratios["beta"].append(np.abs(beta - o64["beta"]).max() / max(np.abs(o32["beta"] - o64["beta"]).max(), 1e-30))
        ratios["logits"].append(rms(output.detach().cpu().numpy(), o64["logits"]) / rms(o32["logits"], o64["logits"]))
        ratios["dlogits"].append(rms(output.grad.cpu().numpy(), o64["dlogits"]) / rms(o32["dlogits"], o64["dlogits"]))
    for k, v in ratios.items():
        v = np.array(v)
        print("|hip - cpu64| / |cpu32 - cpu64|  %-8s  per seed %s   median %.2f  max %.2f" % (k, np.round(v, 2), np.median(v), v.max()))
    for k, v in ratios.items():
        lim = (1.5, 2.5) if k == "beta" else (1.3, 1.5)
        assert np.median(v) <= lim[0] and max(v) <= lim[1], (k, v)


def test_c3_bp_4x320x640():
    from lanedetection_end2end_amd.bp.Loss_crit import backprojection_loss
    from lanedetection_end2end_amd.bp.Networks.LSQ_layer import Net
    N, R, K = 4, 320, 4
    P = erfnet_oracle.make_params(seed=5, out_channels=K)
    x = torch.from_numpy(inputs.images(N, R, 2 * R, seed=171))
    lanes, valid = inputs.bp_targets(N, K, 256, seed=172)
    o32, o64 = _oracle("c3", lambda dt: e2e_oracle.bp_step(x, P, lanes, valid, dt, R, K))
    args = _args(N, R, K, "bp")
    model = _prepare(Net(args), P, "fp32")
    crit = backprojection_loss(args)
    out = model(x.cuda(), torch.zeros(N, K), True)
    betas, output = out[:4], out[5]
    output.retain_grad()
    state = _engine_state(model, output, N, R, 2 * R)
    loss, xcals = 0, []
    for k in range(K):
        l, xc = crit(betas[k], torch.from_numpy(lanes[:, k]).cuda(), torch.from_numpy(valid[:, k]).cuda())
        loss = loss + l
        xcals.append(xc.detach().cpu().numpy())
    loss = loss / K
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(b).all() for b in betas)      # the reference itself is NaN here
    print("C3 BP 4x3x320x640, 4 lanes, fp32")
    # cond(Z) ~ 1e8 in pixel coordinates: the meaningful quantities are the back-projected x (pixels) and the loss
    _check("x_cal (pixels, abs)", np.stack(xcals, 1), o32["x_cal"], o64["x_cal"], 1e-3, rel=False)
    _check("loss (rel)", [float(loss)], [o32["loss"]], [o64["loss"]], 1e-5)
    _check("logits (rel)", output.detach().cpu().numpy(), o32["logits"], o64["logits"], 2e-5, rms=True)
    _check("d loss / d logits (rel)", output.grad.cpu().numpy(), o32["dlogits"], o64["dlogits"], 1e-4, rms=True)
    _check_grads_straight_through(model, P, x, state, output.grad)


def test_c5_seg_2x512x1024():
    from lanedetection_end2end_amd.bp.Loss_crit import define_loss_crit
    from lanedetection_end2end_amd.bp.Networks.LSQ_layer import Net
    N, R, K = 2, 512, 2
    P = erfnet_oracle.make_params(seed=6, out_channels=K + 1)
    x = torch.from_numpy(inputs.images(N, R, 2 * R, seed=181))
    target = inputs.seg_targets(N, R, 2 * R, K + 1, seed=182)
    o32, o64 = _oracle("c5", lambda dt: e2e_oracle.seg_step(x, P, target, dt, K, 30.0))
    args = _args(N, R, K, "bp", end_to_end=False)
    model = _prepare(Net(args), P, "fp32")
    _, crit = define_loss_crit(args)
    logits = model(x.cuda(), torch.zeros(N, K), False, early_return=True)
    logits.retain_grad()
    state = _engine_state(model, logits, N, R, 2 * R)
    loss = crit(logits, torch.from_numpy(target).cuda())
    loss.backward()
    print("C5 segmentation 2x3x512x1024, Cout 3, fp32")
    _check("loss (rel)", [float(loss)], [o32["loss"]], [o64["loss"]], 1e-5)
    _check("logits (rel)", logits.detach().cpu().numpy(), o32["logits"], o64["logits"], 2e-5, rms=True)
    _check("d loss / d logits (rel)", logits.grad.cpu().numpy(), o32["dlogits"], o64["dlogits"], 1e-4, rms=True)
    _check_grads_straight_through(model, P, x, state, logits.grad)


def _train_curve(precision, steps, N, R, K, P, x, lanes, valid):
    from lanedetection_end2end_amd.bp.Loss_crit import backprojection_loss
    from lanedetection_end2end_amd.bp.Networks.LSQ_layer import Net
    from lanedetection_end2end_amd.optim import FusedAdam
    args = _args(N, R, K, "bp")
    model = _prepare(Net(args), P, precision)
    crit = backprojection_loss(args)
    params = [p for p in model.parameters()]
    opt = FusedAdam(params, lr=1e-4)                     # the reference default (BP/Networks/utils.py: lr 1e-4, Adam)
    lt = [torch.from_numpy(lanes[:, k]).cuda() for k in range(K)]
    vt = [torch.from_numpy(valid[:, k]).cuda() for k in range(K)]
    xg = x.cuda()
    curve = []
    for _ in range(steps):
        out = model(xg, torch.zeros(N, K), True)
        loss = sum(crit(out[k], lt[k], vt[k])[0] for k in range(K)) / K
        for p in params:
            p.grad = None
        loss.backward()
        opt.step()
        curve.append(float(loss))
    return np.asarray(curve)


def test_c3_bf16_training_tracks_fp32():
    """BASELINE config 3 names bf16.  The reference has no reduced-precision mode, so there is no reference result to match;
    what CAN be tested end to end is that training in the bf16 modes behaves like training in fp32: the same 40 Adam steps
    (lr 1e-4, the reference default) on the same 4 x 320 x 640 batch from the same initial weights, backprojection loss on
    4 lanes, train-mode BatchNorm, Dropout off.  Criterion: every loss finite, the loss falls in every mode, and the mean of
    the last ten losses of each bf16 mode is within 5 % of the fp32 run's and the first loss within 2 % (measured on MI355X: 0.02 % / 0.17 % and 0.65 % / 0.61 %) (the per-step losses themselves decorrelate: a
    train-mode network amplifies ANY 2^-9 perturbation, see test_bf16_matrix_core_mode_network)."""
    N, R, K, steps = 4, 320, 4, 40
    P = erfnet_oracle.make_params(seed=5, out_channels=K)
    x = torch.from_numpy(inputs.images(N, R, 2 * R, seed=171))
    lanes, valid = inputs.bp_targets(N, K, 256, seed=172)
    curves = {m: _train_curve(m, steps, N, R, K, P, x, lanes, valid) for m in ("fp32", "bf16_mfma", "bf16")}
    ref = curves["fp32"]
    for m, c in curves.items():
        print("%-10s loss: first %.4f  steps 10/20/30 %.4f %.4f %.4f  mean of last ten %.4f" % (m, c[0], c[10], c[20], c[30], c[-10:].mean()))
        assert np.isfinite(c).all(), m
        assert c[-10:].mean() < 0.9 * c[:3].mean(), (m, "loss did not fall")
    for m in ("bf16_mfma", "bf16"):
        assert abs(curves[m][0] - ref[0]) <= 0.02 * abs(ref[0]), (m, curves[m][0], ref[0])
        assert abs(curves[m][-10:].mean() - ref[-10:].mean()) <= 0.05 * ref[-10:].mean(), (m, curves[m][-10:].mean(), ref[-10:].mean())
