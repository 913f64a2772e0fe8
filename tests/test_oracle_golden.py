"""CPU: the oracle restatement vs golden vectors produced by the REAL reference
(oracle/gen_golden.py).  This is what pins the oracle (prompt section 3 / SURVEY 8c)."""
import numpy as np
import pytest
import torch

from conftest import relerr
from oracle import erfnet_oracle, fit_oracle, inputs


def test_homographies(golden_fit):
    M, Mi = fit_oracle.bev_homography()
    assert np.abs(M.astype(np.float32) - golden_fit["bev_M_f32"]).max() == 0
    assert np.abs(Mi.astype(np.float32) - golden_fit["bev_Minv_f32"]).max() == 0
    # closed-form known answer, independent of our own stub
    known = np.array([[-0.5000001788, -2.5000002384, 0.7500001192],
                      [0, -5.5000004768, 1.5000002384], [0, -5.0000004768, 1.0]])
    assert np.abs(M.astype(np.float32) - known).max() < 1e-9   # the survey printed the fp32-cast matrix
    for r in (256, 320):
        M, Mi = fit_oracle.bp_homography(r)
        assert np.abs(M - golden_fit["bp_M_%d" % r]).max() < 1e-12
        assert np.abs(Mi - golden_fit["bp_Minv_%d" % r]).max() < 1e-12
    known_inv = np.array([[0.81647198, -0.91871967, 47.03844774], [0, -0.09758359, 46.79964554],
                          [0, -0.00358453, 1]])
    assert np.abs(fit_oracle.bp_homography(256)[1] - known_inv).max() < 1e-7


def test_grids(golden_fit):
    M, _ = fit_oracle.bev_homography()
    g = fit_oracle.projective_grid(64, 128, M, True, np.float32)
    assert relerr(g, golden_fit["bev_grid_64x128_f32"]) < 3e-7
    M, _ = fit_oracle.bp_homography(256)
    g = fit_oracle.projective_grid(256, 512, M, False, np.float32).reshape(256, 512, 2)[::8, ::8]
    ref = golden_fit["bp_grid_256x512_f32_sample"]
    ok = np.isfinite(ref) & (np.abs(ref) < 1e4)      # away from the pole row of the homography
    assert np.abs(g[ok] - ref[ok]).max() / np.abs(ref[ok]).max() < 1e-6


@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("reg", [0.0, 1e-3])
def test_wls_bev(golden_fit, order, reg):
    N, K, H, W = 2, 2, 64, 128
    o = inputs.lane_like_logits(N, K, H, W, seed=11)
    grid = golden_fit["bev_grid_64x128_f32"]          # the reference's own fp32 grid
    zr = fit_oracle.zero_rows_of(H, 0.3)
    gb = np.random.default_rng(5).standard_normal((2, N, order + 1, 1))[..., 0].transpose(1, 0, 2)
    key = "bev_wls_o%d_r%g_" % (order, reg)
    # fp64 golden used an fp64 grid: rebuild it the same way
    M, _ = fit_oracle.bev_homography()
    grid64 = fit_oracle.projective_grid(H, W, M.astype(np.float32), True, np.float64)
    c = fit_oracle.wls_forward(o, grid64, zr, order, reg, 1.0, "square")
    g = fit_oracle.wls_backward(c, gb)
    assert relerr(c["beta"], golden_fit[key + "f64_beta"]) < 1e-9
    assert relerr(g, golden_fit[key + "f64_grad"]) < 1e-8
    # fp32 reference = fp64 oracle up to the reference's own rounding noise (SURVEY 8c noise floor)
    c32 = fit_oracle.wls_forward(o, grid, zr, order, reg, 1.0, "square")
    assert relerr(c32["beta"], golden_fit[key + "f32_beta"]) < 5e-4


@pytest.mark.parametrize("order", [2, 3])
def test_wls_bp(golden_fit, order):
    N, K, H, W = 1, 4, 256, 512
    o = inputs.lane_like_logits(N, K, H, W, seed=12)
    M, _ = fit_oracle.bp_homography(H)
    grid64 = fit_oracle.projective_grid(H, W, M.astype(np.float32), False, np.float64)
    zr = fit_oracle.zero_rows_of(H, 0.3)
    gb = np.random.default_rng(6).standard_normal((4, N, order + 1, 1))[..., 0].transpose(1, 0, 2)
    c = fit_oracle.wls_forward(o, grid64, zr, order, 0.0, 255.0, "square", skip_masked=False)
    g = fit_oracle.wls_backward(c, gb)
    key = "bp_wls_o%d_c0_f64" % order
    # cond(Z) ~ 1e8 (order 2) / 1e12 (order 3): compare the fitted curve, not raw coefficients
    ys = np.linspace(5, 175, 9)     # inside the data's range of y = 255 - grid_y (rows >= 77)
    Yv = np.stack([ys ** (order - j) for j in range(order + 1)], 1)
    fit_a, fit_b = c["beta"] @ Yv.T, golden_fit[key + "_beta"] @ Yv.T
    tol = 1e-7 if order == 2 else 1e-4
    assert np.abs(fit_a - fit_b).max() < tol * max(1.0, np.abs(fit_b).max())
    assert relerr(g[:, :, ::8, ::8], golden_fit[key + "_grad_sample"]) < 100 * tol
    # GELS path (gels.py) solves the same normal equations
    keyc = "bp_wls_o%d_c1_f64" % order
    if keyc + "_beta" in golden_fit:
        fit_c = golden_fit[keyc + "_beta"] @ Yv.T
        assert np.abs(fit_a - fit_c).max() < 1e-4 * max(1.0, np.abs(fit_c).max())


def test_gels_matches_wls():
    rng = np.random.default_rng(3)
    A = rng.standard_normal((2, 50, 3))
    b = rng.standard_normal((2, 50, 1))
    x, AtA = fit_oracle.gels_forward(A, b)
    ref = np.stack([np.linalg.lstsq(A[i], b[i], rcond=None)[0] for i in range(2)])
    assert relerr(x, ref) < 1e-10
    gA, gb = fit_oracle.gels_backward(A, b, x, AtA, np.ones_like(x))
    At = torch.tensor(A, requires_grad=True)
    bt = torch.tensor(b, requires_grad=True)
    torch.linalg.solve(At.transpose(1, 2) @ At, At.transpose(1, 2) @ bt).sum().backward()
    assert relerr(gA, At.grad.numpy()) < 1e-9 and relerr(gb, bt.grad.numpy()) < 1e-9


@pytest.mark.parametrize("order,wf", [(2, "none"), (2, "linear"), (2, "quadratic"), (1, "none")])
def test_area_loss(golden_fit, order, wf):
    beta, gt = golden_fit["area_beta"][:, : order + 1], golden_fit["area_gt"][:, : order + 1]
    L, g = fit_oracle.area_loss(beta, gt, order, wf)
    assert abs(L - golden_fit["area_o%d_%s_f64_loss" % (order, wf)]) < 1e-14
    assert relerr(g, golden_fit["area_o%d_%s_f64_grad" % (order, wf)][..., 0]) < 1e-12
    assert abs(L - golden_fit["area_o%d_%s_f32_loss" % (order, wf)]) < 1e-6


def test_area_and_trapezoid_known_answers(golden_fit):
    b = np.array([[0.1, -0.2, 0.5], [0, 0.1, 0.4]])
    g = np.array([[0.05, -0.1, 0.45], [0.01, 0.2, 0.5]])
    vals = [fit_oracle.area_loss(b, g, 2, wf)[0] for wf in ("none", "linear", "quadratic")]
    assert np.allclose(vals, golden_fit["area_survey"], rtol=0, atol=1e-15)
    assert np.allclose(vals, [6.947098e-3, 4.142810e-3, 2.725337e-3], rtol=2e-6)
    g2 = g.copy(); g2[1] = 0
    assert abs(fit_oracle.area_loss(b, g2, 2, "none")[0] - 4.987852e-4) < 1e-9
    tz = fit_oracle.trapezoidal(b, g)
    assert np.allclose(tz, golden_fit["trapezoid_survey"], atol=1e-14)
    assert np.allclose(tz, [0.0162169654, 0.0956433862], atol=1e-9)
    assert fit_oracle.area_loss(b, np.zeros_like(g), 2, "none")[0] == 0.0       # no lane kept


@pytest.mark.parametrize("order", [2, 3])
def test_backproj_loss(golden_fit, order):
    s = fit_oracle.backproj_setup(order)
    assert relerr(s["y_prime"], golden_fit["bp_yprime_o%d" % order]) < 1e-12
    assert relerr(s["Y"], golden_fit["bp_Y_o%d" % order]) < 1e-12
    lanes, valid = inputs.bp_targets(6, 1, 256, seed=31)
    L, xc, g = fit_oracle.backproj_loss(golden_fit["bp_loss_o%d_beta" % order], lanes[:, 0], valid[:, 0], s)
    assert abs(L - golden_fit["bp_loss_o%d_loss" % order]) < 1e-9 * abs(L)
    assert relerr(xc, golden_fit["bp_loss_o%d_xcal" % order]) < 1e-12
    assert relerr(g, golden_fit["bp_loss_o%d_grad" % order][..., 0]) < 1e-10


def test_backproj_known_answer():
    """beta=(0,0,256), x_gt=250, valid[8:]=1."""
    s = fit_oracle.backproj_setup(2)
    assert np.allclose(s["y_d"][:4], [32, 36, 40, 44])
    assert np.allclose(s["y_prime"][:4], [-864.3932, -343.2867, -148.4714, -46.5554], atol=1e-3)
    valid = np.zeros((1, 56)); valid[:, 8:] = 1
    L, xc, _ = fit_oracle.backproj_loss(np.array([[0, 0, 256.0]]), np.full((1, 56), 250.0), valid, s)
    assert abs(L - 22.1856955666) < 1e-6
    assert np.allclose(xc[0, 8:12], [255.839215, 255.788970, 255.738725, 255.688480], atol=1e-5)
    L0, _, g0 = fit_oracle.backproj_loss(np.array([[0, 0, 256.0]]), np.full((1, 56), 250.0), valid * 0, s)
    assert L0 == 0.0 and not g0.any()


def test_cross_entropy(golden_fit):
    tgt = inputs.seg_targets(2, 8, 16, 3, seed=41)
    L, g = fit_oracle.cross_entropy_2d(golden_fit["ce_logits"], tgt, [1.0, 30.0, 30.0])
    assert abs(L - golden_fit["ce_loss"]) < 2e-6 * abs(L)
    assert relerr(g, golden_fit["ce_grad"]) < 2e-6


def test_param_spec_counts():
    spec = erfnet_oracle.param_spec(3, 2)
    n_param = sum(int(np.prod(s)) for k, s in spec.items()
                  if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_param == 2063344 and len(spec) == 345
    assert sum(int(np.prod(s)) for k, s in erfnet_oracle.param_spec(3, 4).items()
               if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))) == 2063732


@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-10), ("f32", torch.float32, 2e-4)])
def test_backbone_oracle(golden_backbone, tag, dtype, tol):
    N, H, W, Cout = 2, 64, 128, 2
    x = torch.from_numpy(inputs.images(N, H, W, seed=51)).to(dtype)
    gy = torch.from_numpy(np.random.default_rng(52).standard_normal((N, Cout, H, W))).to(dtype)
    P = erfnet_oracle.cast_params(erfnet_oracle.make_params(seed=3, out_channels=Cout), dtype)
    keys = [k for k, v in P.items() if v.is_floating_point() and "running" not in k]
    for k in keys:
        P[k].requires_grad_(True)
    stats = {}
    enc, dec = erfnet_oracle.erfnet_forward(x, P, training=True, stats_out=stats)
    (dec * gy).sum().backward()
    assert relerr(enc.detach(), golden_backbone["bb_train_enc_" + tag]) < tol
    assert relerr(dec.detach(), golden_backbone["bb_train_dec_" + tag]) < tol
    for k in ("encoder.initial_block.bn.running_mean", "encoder.layers.9.bn2.running_var",
              "decoder.layers.3.bn.running_var"):
        assert relerr(stats[k], golden_backbone["bb_train_%s_%s" % (k, tag)]) < max(tol, 1e-6)
    # fp32-vs-fp32 gradients of the early layers differ by ~2 % between two CPU evaluation orders
    # (chaotic amplification through 39 train-mode BNs): that is the reference's own noise floor.
    gtol = 50 * tol if tag == "f64" else 5e-2
    gkeys = list(golden_backbone["bb_grad_keys"])
    norms = golden_backbone["bb_grad_norms_" + tag]
    for k, n in zip(gkeys, norms):
        if n < 0:
            assert P[k].grad is None, k          # encoder.output_conv never gets a grad
        elif n > 1e-6 * norms.max():
            assert abs(float(P[k].grad.double().norm()) - n) < gtol * max(n, 1e-3 * norms.max()), k
    for name in golden_backbone.files:
        if name.startswith("bb_grad_") and name.endswith(tag) and "norms" not in name:
            k = name[len("bb_grad_"):-len(tag) - 1]
            ref = golden_backbone[name]
            if np.abs(ref).max() > 1e-6 * norms.max():
                assert relerr(P[k].grad, ref) < gtol, k
    P.update(stats)      # the golden eval pass ran after one train-mode step updated the running stats
    with torch.no_grad():
        _, dec = erfnet_oracle.erfnet_forward(x, P, training=False)
    assert relerr(dec, golden_backbone["bb_eval_dec_" + tag]) < max(tol, 1e-6 if tag == 'f32' else 0)


def test_e2e_oracle_matches_reference_goldens(golden_e2e):
    """oracle/e2e_oracle.py (the composition the full-size GPU tests and bench.py's parity leg use) against the real
    reference's end-to-end runs at 4x3x256x512 (BEV) and 2x3x256x512 (BP), fp64 legs: lane coefficients, loss,
    d loss / d logits, every parameter-gradient norm."""
    from oracle import e2e_oracle, erfnet_oracle, inputs
    import torch
    N, R = 4, 256
    x = torch.from_numpy(inputs.images(N, R, 2 * R, seed=61))
    gt = inputs.bev_gt_params(N, seed=62)
    o = e2e_oracle.bev_step(x, erfnet_oracle.make_params(seed=4, out_channels=2), gt, torch.float64, R)
    assert relerr(o["beta"], golden_e2e["e2e_bev_beta_f64"]) < 1e-9
    assert abs(o["loss"] - float(golden_e2e["e2e_bev_loss_f64"])) < 1e-10 * abs(o["loss"])
    assert relerr(o["logits"][:, :, ::16, ::16], golden_e2e["e2e_bev_logits_sample_f64"]) < 1e-10
    assert relerr(o["dlogits"][:, :, ::16, ::16], golden_e2e["e2e_bev_dlogits_sample_f64"]) < 1e-7
    keys, n64 = list(golden_e2e["e2e_bev_grad_keys"]), golden_e2e["e2e_bev_grad_norms_f64"]
    for k, ref in zip(keys, n64):
        k = k[len("net."):] if k.startswith("net.") else k
        got = o["grad_norms"][k]
        if ref < 0:
            assert got is None, k
        else:
            assert abs(got - ref) <= 1e-7 * max(ref, n64.max() * 1e-6), (k, got, ref)
    # BP tree
    N, K = 2, 4
    x = torch.from_numpy(inputs.images(N, R, 2 * R, seed=71))
    lanes, valid = inputs.bp_targets(N, K, R, seed=72)
    o = e2e_oracle.bp_step(x, erfnet_oracle.make_params(seed=5, out_channels=K), lanes, valid, torch.float64, R, K)
    assert np.abs(o["x_cal"] - golden_e2e["e2e_bp_xcal_f64"]).max() < 1e-6         # pixels
    assert abs(o["loss"] - float(golden_e2e["e2e_bp_loss_f64"])) < 1e-8 * abs(o["loss"])
    assert relerr(o["dlogits"][:, :, ::16, ::16], golden_e2e["e2e_bp_dlogits_sample_f64"]) < 1e-6


@pytest.mark.parametrize("tree", ["bev", "bp"])
@pytest.mark.parametrize("D", [3, 4])
def test_mse_loss_vs_reference_golden(tree, D):
    """fit_oracle.mse_loss vs the real reference's MSE_Loss (--loss_policy mse; oracle/gen_golden_mse.py)."""
    import os
    from conftest import GOLDEN
    from oracle.gen_golden_mse import mse_inputs
    G = np.load(os.path.join(GOLDEN, "mse.npz"))
    p, g = mse_inputs(D)
    L, grad = fit_oracle.mse_loss(p, g)
    assert abs(L - float(G["%s_d%d_f64_loss" % (tree, D)])) < 1e-14
    assert np.abs(grad - G["%s_d%d_f64_grad" % (tree, D)]).max() < 1e-15
    assert abs(L - float(G["%s_d%d_f32_loss" % (tree, D)])) < 1e-6 * L


def test_straight_through_masks_of_the_other_state():
    """erfnet_oracle's straight-through mode takes the ReLU DERIVATIVE masks from the state it is evaluated at (_relu): with its
    own taps (and its own folded bn1 vectors) as that state the gradients are those of the plain run; with one post-ReLU
    element of the state flipped to zero the gradients change -- the mask really comes from the state, not from sign(z)."""
    N, H, W = 1, 32, 64
    P = erfnet_oracle.make_params(seed=3, out_channels=2)
    x = torch.from_numpy(inputs.images(N, H, W, seed=5)).double()
    gy = torch.from_numpy(np.random.default_rng(6).standard_normal((N, 2, H, W)))

    def grads(override):
        Pd = erfnet_oracle.cast_params(P, torch.float64)
        keys = [k for k, v in Pd.items() if v.is_floating_point() and "running" not in k]
        for k in keys:
            Pd[k].requires_grad_(True)
        taps = {}
        _, dec = erfnet_oracle.erfnet_forward(x, Pd, training=True, taps=taps, override=override)
        (dec * gy).sum().backward()
        return {k: Pd[k].grad for k in keys if Pd[k].grad is not None}, taps, Pd

    g0, taps, Pd = grads(None)
    state = {k: v.detach() for k, v in taps.items()}
    for prefix, kind, _, _, _, _ in erfnet_oracle.layer_table():
        if kind == "nb1d":
            t2 = state[prefix + "#1"]
            mean, var = t2.mean(dim=(0, 2, 3)), t2.var(dim=(0, 2, 3), unbiased=False)
            sc = Pd[prefix + ".bn1.weight"].detach() * torch.rsqrt(var + erfnet_oracle.BN_EPS)
            state[prefix + "#bn1"] = (sc, Pd[prefix + ".bn1.bias"].detach() - mean * sc)
    g1, _, _ = grads(state)
    assert max(relerr(g1[k], g0[k]) for k in g0) < 1e-12
    flipped = dict(state)
    t3 = state["decoder.layers.1#2"].clone()
    idx = torch.nonzero(t3 > 0)[0]
    t3[tuple(idx)] = 0.0                                   # one element of one saved post-ReLU tensor
    flipped["decoder.layers.1#2"] = t3
    g2, _, _ = grads(flipped)
    assert relerr(g2["decoder.layers.1.conv3x1_2.bias"], g0["decoder.layers.1.conv3x1_2.bias"]) > 1e-9
