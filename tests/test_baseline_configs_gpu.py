"""The BASELINE.json configurations against the CPU oracle (fp32 and fp64 backbone legs of oracle/e2e_oracle.py, whose
composition is pinned to the real reference's end-to-end runs by
tests/test_oracle_golden.py::test_e2e_oracle_matches_reference_goldens).

DIRECT oracle comparisons (a CPU fp64 run of the whole step beside the HIP one):
  C2  BEV, 2 lanes, 32 x 3 x 256 x 512 -- ITS OWN SIZE -- train mode (batch statistics over all 32 images), fp32 matrix cores,
      and the same step in precision mode fp32x9 with the SHIPPED kernel-selection rule (no test-size override);
  C3  Backprojection tree, 4 lanes, 320 x 640 at batch 4 (masked-row pole of the homography sanitised in the oracle only);
  C5  segmentation branch, Cout = 3, 512 x 1024 at batch 2, class-weighted cross entropy.
TRANSITIVE checks at C3's and C5's OWN batch sizes (64, fp32 and bf16; 16) -- a CPU fp64 step at those sizes does not fit the
suite's time cap (Missing #2):
  eval mode   image i of the full-batch HIP run == the same image in a batch-4 (batch-2) HIP run, logits and lane coefficients
              (the small-batch run is oracle-checked right here, eval mode, both CPU legs); parameter gradients of a linear
              functional of the logits == the sum over the chunks' gradients;
  train mode  every BatchNorm mean / variance the engine folded at the full batch == an fp64 reduction of the engine's own
              pre-BN tensor, and the running statistics it wrote == the momentum update from those.
Together with the per-kernel parity tests (tests/test_backbone_gpu.py: every addressing path at ragged sizes) this leaves no
batch-size-dependent arithmetic unchecked: everything else in the network is per-image.

Reference call sites: BEV/main.py:213-223,264-265; BP/main.py:256-263,286-305.

Criterion (lane coefficients / back-projected x, loss, logits, d loss / d logits): |hip - cpu64| <= 1.5 * |cpu32 - cpu64| (round 4;
2x before the convolutions' two-accumulator summation) -- the HIP path may be no further from the fp64 truth than 1.5x the
distance of the reference arithmetic's own fp32 run -- with a small absolute floor where the fp32 leg happens to land on the
fp64 one.
The norm is the RMS for the tensors (the statistic that is stable when two independent roundoff-noise fields are compared:
both legs are draws of the same noise process through a chaotic train-mode network) and the maximum for the scalars and the
lane coefficients; the maximum over the 1e7..3e7 elements of a tensor is printed too and held to 2x (one extreme sample of
one draw against one extreme sample of another).  All three numbers are printed.  Dropout is off (p = 0): the draw of torch's generator cannot be shared with the oracle at
this size; the masked path is covered in tests/test_backbone_gpu.py.
Parameter gradients (round 3): every tensor against the fp64 oracle evaluated STRAIGHT-THROUGH at the engine's own forward state
and driven by the engine's own d loss / d logits -- backward arithmetic only, held to 3e-5 of each tensor's maximum at these sizes.
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


# scalars and lane coefficients (a maximum over <= 200 numbers of one noise draw against another): measured over six seeds in round 4
# 0.87 - 1.71, so a single draw is held to 2x; the RMS statistics of the 1e7-element tensors to 1.5x (measured 0.83 - 0.91)
MAXNORM_BAND = 2.0


def _check(name, hip, r32, r64, abs_floor, rel=True, rms=False):
    hip, r32, r64 = (np.asarray(v, dtype=np.float64) for v in (hip, r32, r64))
    scale = max(np.abs(r64).max(), 1e-300) if rel else 1.0
    e64, e32, floor = np.abs(hip - r64).max() / scale, np.abs(hip - r32).max() / scale, np.abs(r32 - r64).max() / scale
    print("%-28s max: |hip-cpu64| %.3e   |hip-cpu32| %.3e   |cpu32-cpu64| %.3e" % (name, e64, e32, floor))
    if rms:
        q = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2))) / scale
        r64e, r32e, rfl = q(hip, r64), q(hip, r32), q(r32, r64)
        print("%-28s rms: |hip-cpu64| %.3e   |hip-cpu32| %.3e   |cpu32-cpu64| %.3e" % ("", r64e, r32e, rfl))
        assert r64e <= max(1.5 * rfl, abs_floor), (name, "rms", r64e, rfl)
        assert e64 <= max(2.0 * floor, abs_floor), (name, "max", e64, floor)       # max norms: extreme-value statistic
    else:
        assert e64 <= max(MAXNORM_BAND * floor, abs_floor), (name, e64, floor)
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
    logits and d loss / d logits (RMS).
    Round 3 measured logits 1.21-1.30 (median 1.24), d loss / d logits 1.20-1.35 (1.22), coefficients 1.17-2.05 (1.34): the HIP
    path was SYSTEMATICALLY ~1.25x further from fp64 than oneDNN's fp32.  Round 4 found the cause on the CPU
    (tools/accum_study.py: the fp32 oracle with the kernel's summation order emulated reproduces 1.31x; it is the ONE fma chain
    per output element over K = 192 / 384 products, chiefly in the 64-channel layers -- not the folded BatchNorm, not the fp32
    statistics partials) and removed it: the tap-GEMM sums segments of 32 products into a second accumulator set (emulated
    0.90x).  Measured on MI355X with the same six seeds and CPU legs, round-3 library vs round-4 library on one box:
    logits 1.24 -> 0.85 (0.83-0.88), d loss / d logits 1.22 -> 0.86 (0.79-0.91), coefficients 1.34 -> 1.23 (0.87-1.71).
    Held to: RMS ratios median <= 1.0, max <= 1.1; coefficients median <= 1.3, max <= 2.0."""
    from lanedetection_end2end_amd import _lib
    from lanedetection_end2end_amd.bev.Loss_crit import Area_Loss
    from lanedetection_end2end_amd.bev.Networks.LSQ_layer import Net
    N, R = 8, 256
    # (round 6: precision mode fp32x9 -- fp32 tensors and accumulation, exact products from 3-way bf16 splits, ONE chain per output
    # element in the split kernel -- through the same seeds and CPU legs: the accuracy statement behind bench.py's fp32_split_x9 figure)
    modes = ("fp32", "fp32x9")
    ratios = {m: {"beta": [], "beta_rms": [], "logits": [], "dlogits": []} for m in modes}
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
        rms = lambda a, b: float(np.sqrt(np.mean((np.asarray(a, dtype=np.float64) - b) ** 2)))
        outs = {}
        for mode in modes:
            model.net.precision = mode
            # (at batch 8 the split kernel's shipped size rule -- >= 192 workgroups of 512 pixels -- would hand every launch back to the
            # fp32 cores: lift it for this mode, so that what is measured IS the split arithmetic)
            _lib.load().lf_debug_set_split_any_size(1 if mode == "fp32x9" else 0)
            model.zero_grad(set_to_none=True)
            b0, b1, _, _, _, _, output, _, _ = model(x.cuda(), True)
            output.retain_grad()
            (crit(b0, gtc[:, 0]) + crit(b1, gtc[:, 1])).backward()
            _lib.load().lf_debug_set_split_any_size(0)
            outs[mode] = output.detach().clone()
            beta = torch.stack([b0, b1], 1)[..., 0].detach().cpu().numpy()
            r = ratios[mode]
            r["beta"].append(np.abs(beta - o64["beta"]).max() / max(np.abs(o32["beta"] - o64["beta"]).max(), 1e-30))
            r["beta_rms"].append(rms(beta, o64["beta"]) / max(rms(o32["beta"], o64["beta"]), 1e-30))     # over all 48 coefficients
            r["logits"].append(rms(output.detach().cpu().numpy(), o64["logits"]) / rms(o32["logits"], o64["logits"]))
            r["dlogits"].append(rms(output.grad.cpu().numpy(), o64["dlogits"]) / rms(o32["dlogits"], o64["dlogits"]))
        model.net.precision = "fp32"
        assert not torch.equal(outs["fp32"], outs["fp32x9"])          # the split kernels really ran
    for mode in modes:
        for k, v in ratios[mode].items():
            v = np.array(v)
            print("[%-6s] |hip - cpu64| / |cpu32 - cpu64|  %-8s  per seed %s   median %.2f  max %.2f" % (mode, k, np.round(v, 2), np.median(v), v.max()))
    for k, v in ratios["fp32"].items():
        lim = (1.3, 2.0) if k.startswith("beta") else (1.0, 1.1)
        assert np.median(v) <= lim[0] and max(v) <= lim[1], (k, v)
    # fp32x9 keeps ONE fp32 chain over the whole contraction in the 64- / 128-channel launches it takes (round 3's fp32 kernel did the same
    # and sat at 1.21-1.35): held to the reference arithmetic's own distance from fp64 times 1.5 (RMS) / the maximum-norm band
    for k, v in ratios["fp32x9"].items():
        lim = (1.6, 2.5) if k.startswith("beta") else (1.4, 1.5)
        assert np.median(v) <= lim[0] and max(v) <= lim[1], ("fp32x9", k, v)


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
    what CAN be tested end to end is that training in the bf16 mode behaves like training in fp32: the same 40 Adam steps
    (lr 1e-4, the reference default) on the same 4 x 320 x 640 batch from the same initial weights, backprojection loss on
    4 lanes, train-mode BatchNorm, Dropout off.  Criterion: every loss finite, the loss falls in every mode, and the mean of
    the last ten losses of each bf16 mode is within 5 % of the fp32 run's and the first loss within 2 % (measured on MI355X: 0.02 % / 0.17 % and 0.65 % / 0.61 %) (the per-step losses themselves decorrelate: a
    train-mode network amplifies ANY 2^-9 perturbation, see test_bf16_matrix_core_mode_network)."""
    N, R, K, steps = 4, 320, 4, 40
    P = erfnet_oracle.make_params(seed=5, out_channels=K)
    x = torch.from_numpy(inputs.images(N, R, 2 * R, seed=171))
    lanes, valid = inputs.bp_targets(N, K, 256, seed=172)
    curves = {m: _train_curve(m, steps, N, R, K, P, x, lanes, valid) for m in ("fp32", "bf16")}
    ref = curves["fp32"]
    for m, c in curves.items():
        print("%-10s loss: first %.4f  steps 10/20/30 %.4f %.4f %.4f  mean of last ten %.4f" % (m, c[0], c[10], c[20], c[30], c[-10:].mean()))
        assert np.isfinite(c).all(), m
        assert c[-10:].mean() < 0.9 * c[:3].mean(), (m, "loss did not fall")
    for m in ("bf16",):
        assert abs(curves[m][0] - ref[0]) <= 0.02 * abs(ref[0]), (m, curves[m][0], ref[0])
        assert abs(curves[m][-10:].mean() - ref[-10:].mean()) <= 0.05 * ref[-10:].mean(), (m, curves[m][-10:].mean(), ref[-10:].mean())


# ---------------------------------------------------------------------------------------------------------------------------
# C3 and C5 AT THEIR OWN BATCH SIZES (64 in fp32 and bf16; 16): transitive checks (module docstring)
# ---------------------------------------------------------------------------------------------------------------------------

def _bn_sites():
    """(layer index, slot of the pre-BN tensor, bn index within the layer, state_dict prefix of the BatchNorm, channels)."""
    out = []
    for li, (prefix, kind, cin, cout, _, _) in enumerate(erfnet_oracle.layer_table()):
        if kind == "nb1d":
            out += [(li, 1, 0, prefix + ".bn1", cout), (li, 3, 1, prefix + ".bn2", cout)]
        else:
            out.append((li, 0, 0, prefix + ".bn", cout))
    return out


def _layer_hw(H, W):
    hw, h, w = [], H, W
    for prefix, kind, *_ in erfnet_oracle.layer_table():
        if kind == "down":
            h, w = h // 2, w // 2
        elif kind == "up":
            h, w = h * 2, w * 2
        hw.append((h, w))
    return hw


def _check_bn_statistics_at_full_batch(net, x, precision):
    """Train-mode forward at the configuration's own batch; every BatchNorm's folded statistics and running-statistics update
    against an fp64 reduction (on the GPU, torch) of the engine's own pre-BN tensor as stored (bf16 tensors in mode "bf16")."""
    from lanedetection_end2end_amd import _lib
    lib = _lib.load()
    N, _, H, W = x.shape
    net.precision = precision
    net.train()
    fresh = {k: v.clone() for k, v in net.state_dict().items() if "running" in k}
    _, dec = net(x, True)[:2]
    plan, ws = net._plan(N, H, W), dec.grad_fn.ws
    hw = _layer_hw(H, W)
    sd = net.state_dict()
    worst = dict(mean=0.0, rstd=0.0, rmean=0.0, rvar=0.0)
    for li, slot, bn, prefix, C in _bn_sites():
        h, w = hw[li]
        off = lib.lf_erfnet_activation_offset(plan.handle, li, slot)
        n = N * h * w * C
        if precision == "bf16":
            t = ws.view(torch.bfloat16)[2 * off: 2 * off + n]
        else:
            t = ws.view(torch.float32)[off: off + n]
        t = t.view(-1, C).double()
        mean, var = t.mean(0), t.var(0, unbiased=False)
        cnt = t.shape[0]
        del t
        vec = [ws.view(torch.float32)[o: o + C].double() for o in
               (lib.lf_erfnet_bn_vector_offset(plan.handle, li, bn, 2), lib.lf_erfnet_bn_vector_offset(plan.handle, li, bn, 3))]
        rstd_e, mean_e = vec[0], -vec[1] / vec[0]
        rstd = torch.rsqrt(var + erfnet_oracle.BN_EPS)
        std = torch.sqrt(var + erfnet_oracle.BN_EPS)
        e_rstd = float(((rstd_e - rstd).abs() / rstd).max())
        e_mean = float(((mean_e - mean).abs() / (mean.abs() + std)).max())
        rm = 0.9 * fresh[prefix + ".running_mean"].double() + 0.1 * mean
        rv = 0.9 * fresh[prefix + ".running_var"].double() + 0.1 * var * cnt / (cnt - 1)
        e_rm = float(((sd[prefix + ".running_mean"].double() - rm).abs() / (rm.abs() + 0.1 * std)).max())
        e_rv = float(((sd[prefix + ".running_var"].double() - rv).abs() / rv).max())
        for k, v in zip(("mean", "rstd", "rmean", "rvar"), (e_mean, e_rstd, e_rm, e_rv)):
            worst[k] = max(worst[k], v)
        # fp32 partial sums of y and y^2 per 256-pixel tile, combined in fp64: var = E[y^2] - mean^2 keeps ~1e-7 * (1 + mean^2 / var)
        # (mode "bf16": the stem and the pooled channels take their statistics from the fp32 values BEFORE the bf16 store, the
        # tap-GEMM epilogues from the values as stored: the stored tensor's own statistics differ by its rounding noise, 2^-9)
        tol = 1e-5 if precision == "fp32" else 3e-4
        assert e_rstd < tol and e_mean < tol and e_rm < tol and e_rv < 2 * tol, (prefix, e_mean, e_rstd, e_rm, e_rv)
    print("BatchNorm statistics at batch %d (%s), %d BatchNorms, worst relative error: batch mean %.1e, rstd %.1e, running mean "
          "%.1e, running var %.1e" % (N, precision, len(_bn_sites()), worst["mean"], worst["rstd"], worst["rmean"], worst["rvar"]))
    assert torch.isfinite(dec).all()


def _eval_chunks_equal_full_batch(model, call, x, chunk, precision, P, oracle_check):
    """Eval mode (running statistics): the full-batch run == the per-chunk runs, image by image (logits, lane coefficients), and
    the parameter gradients of sum(logits * gy) at the full batch == the sum of the chunks' gradients.  The first chunk's logits
    are checked against the CPU oracle in the same mode (fp32 cores: both legs; bf16: bf16-level agreement with fp64)."""
    N = x.shape[0]
    net = model.net
    net.precision = precision
    model.eval()
    gy = torch.from_numpy(np.random.default_rng(77).standard_normal(tuple(call(x[:chunk])[1].shape[1:])).astype(np.float32)).cuda()
    for p in model.parameters():
        p.grad = None
    betas_f, logits_f = call(x)
    (logits_f * gy).sum().backward()
    g_full = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
    lmax = float(logits_f.detach().abs().max())
    g_sum = {k: torch.zeros_like(v, dtype=torch.float64) for k, v in g_full.items()}
    worst_l = worst_b = 0.0
    first = None
    for c in range(0, N, chunk):
        for p in model.parameters():
            p.grad = None
        betas_c, logits_c = call(x[c: c + chunk])
        if first is None:
            first = logits_c.detach()
        (logits_c * gy).sum().backward()
        for k, p in net.named_parameters():
            if p.grad is not None:
                g_sum[k] += p.grad.double()
        worst_l = max(worst_l, float((logits_c.detach() - logits_f.detach()[c: c + chunk]).abs().max()) / lmax)
        for bf, bc in zip(betas_f, betas_c):
            if bf is not None:
                worst_b = max(worst_b, float((bc.detach() - bf.detach()[c: c + chunk]).abs().max() / bf.detach().abs().max()))
    worst_g = 0.0
    for k, v in g_full.items():
        scale = float(g_sum[k].abs().max())
        if scale == 0.0:
            continue
        worst_g = max(worst_g, float((v.double() - g_sum[k]).abs().max()) / scale)
    print("batch %d vs %d chunks of %d (%s, eval): logits %.1e of max, lane coefficients %.1e, parameter gradients vs the chunk sum %.1e"
          % (N, N // chunk, chunk, precision, worst_l, worst_b, worst_g))
    assert worst_l <= 1e-6 and worst_b <= 1e-6
    assert worst_g <= 2e-5     # summation order of the split-K weight gradient differs with the batch (measured 1.0e-6, fp32 and bf16: eval mode is per image, so the rounded tensors of a chunk ARE the full batch's)
    if oracle_check:
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        torch.set_num_threads(min(os.cpu_count() or 1, 64))
        legs = {}
        for dt in (torch.float32, torch.float64):
            Pd = erfnet_oracle.cast_params(sd, dt)
            with torch.no_grad():
                legs[dt] = erfnet_oracle.erfnet_forward(x[:chunk].cpu().to(dt), Pd, training=False)[1].double().numpy()
        got = first.cpu().double().numpy()
        rms = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)))
        e64, fl = rms(got, legs[torch.float64]), rms(legs[torch.float32], legs[torch.float64])
        scale = float(np.abs(legs[torch.float64]).max())
        print("first chunk vs the CPU oracle (eval): rms |hip-cpu64| %.2e  |cpu32-cpu64| %.2e (of max |logits|)" % (e64 / scale, fl / scale))
        if precision == "fp32":
            assert e64 <= max(1.5 * fl, 2e-6 * scale), (e64, fl)
        else:
            assert e64 <= 1e-2 * scale, (e64, scale)            # bf16 products, fp32 accumulation, 70 layers (measured 4.4e-3)


def _warm_running_stats(model, call, x_small):
    """Non-trivial running statistics for the eval-mode runs: one train-mode forward on a small batch."""
    model.train()
    with torch.no_grad():
        call(x_small)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_c3_bp_batch64_transitive(precision):
    """BASELINE config 3 at its own size and dtype: 64 x 3 x 320 x 640, 4 lanes, fp32 and bf16."""
    from lanedetection_end2end_amd.bp.Networks.LSQ_layer import Net
    N, R, K = 64, 320, 4
    P = erfnet_oracle.make_params(seed=5, out_channels=K)
    x = torch.from_numpy(inputs.images(N, R, 2 * R, seed=171)).cuda()
    model = _prepare(Net(_args(N, R, K, "bp")), P, "fp32")
    gl = torch.zeros(4, K)

    def call(xb):
        out = model(xb, gl, True)
        return out[:4], out[5]
    _check_bn_statistics_at_full_batch(model.net, x, precision)
    model.net.load_state_dict(P)                                   # fresh running statistics again
    model.net.precision = "fp32"
    _warm_running_stats(model, call, x[:4])
    _eval_chunks_equal_full_batch(model, call, x, 4, precision, P, oracle_check=True)
    model.net.precision = "fp32"


def test_c5_seg_batch16_transitive():
    """BASELINE config 5's per-GPU shard at its own size: 16 x 3 x 512 x 1024, Cout = 3 (segmentation branch, early_return)."""
    from lanedetection_end2end_amd.bp.Networks.LSQ_layer import Net
    N, R, K = 16, 512, 2
    P = erfnet_oracle.make_params(seed=6, out_channels=K + 1)
    x = torch.from_numpy(inputs.images(N, R, 2 * R, seed=181)).cuda()
    model = _prepare(Net(_args(N, R, K, "bp", end_to_end=False)), P, "fp32")
    gl = torch.zeros(2, K)

    def call(xb):
        return (None,), model(xb, gl, False, early_return=True)
    _check_bn_statistics_at_full_batch(model.net, x, "fp32")
    model.net.load_state_dict(P)
    _warm_running_stats(model, call, x[:2])
    _eval_chunks_equal_full_batch(model, call, x, 2, "fp32", P, oracle_check=True)
