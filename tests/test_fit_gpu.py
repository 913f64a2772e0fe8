"""GPU parity: fused WLS kernels and the loss kernels (through the C ABI) vs the CPU oracle
and vs golden vectors produced by the real reference.

Tolerances (north_star: 1e-5 relative): the HIP path accumulates in fp64, so it is compared
with the fp64 evaluation of the reference formula at 1e-6 or tighter; the reference's own
fp32 result is compared at its measured noise floor (printed).
"""
import numpy as np
import pytest
import torch

from conftest import relerr
from oracle import fit_oracle, inputs

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.as_tensor(np.asarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


@pytest.fixture(scope="module")
def lf():
    import lanedetection_end2end_amd as pkg
    from lanedetection_end2end_amd import fit, geometry, losses, ops
    return type("ns", (), dict(pkg=pkg, fit=fit, geometry=geometry, losses=losses, ops=ops))


@pytest.mark.parametrize("order", [0, 1, 2])
@pytest.mark.parametrize("reg", [0.0, 1e-3])
def test_wls_bev_vs_golden(lf, golden_fit, order, reg):
    N, K, H, W = 2, 2, 64, 128
    o = inputs.lane_like_logits(N, K, H, W, seed=11)
    grid = golden_fit["bev_grid_64x128_f32"]
    zr = fit_oracle.zero_rows_of(H, 0.3)
    ot = dev(o).requires_grad_(True)
    beta, masked, status = lf.fit.fit_lanes(ot, dev(grid), zr, order, reg, 1.0, "square")
    gb = np.random.default_rng(5).standard_normal((2, N, order + 1, 1))[..., 0].transpose(1, 0, 2)
    (beta * dev(gb)).sum().backward()
    key = "bev_wls_o%d_r%g_" % (order, reg)
    e64 = relerr(beta.detach().cpu(), golden_fit[key + "f64_beta"])
    e32 = relerr(beta.detach().cpu(), golden_fit[key + "f32_beta"])
    floor = relerr(golden_fit[key + "f32_beta"], golden_fit[key + "f64_beta"])
    # sharp: the fp64 formula on the SAME fp32 grid values the kernel reads (the golden fp64 run rebuilt its
    # grid in fp64, a 6e-8 relative change of every coordinate that cond(Z) ~ 1e3 turns into ~1e-6 on beta)
    c = fit_oracle.wls_forward(o, grid.astype(np.float64), zr, order, reg, 1.0, "square")
    e_same = relerr(beta.detach().cpu(), c["beta"])
    print("order %d reg %g: |hip-oracle(same grid)| %.2e |hip-ref64| %.2e  |hip-ref32| %.2e  |ref32-ref64| %.2e"
          % (order, reg, e_same, e64, e32, floor))
    assert e_same < 2e-7
    assert e64 < 1e-5
    assert e32 < 2 * floor + 1e-6
    assert relerr(ot.grad.cpu(), fit_oracle.wls_backward(c, gb)) < 2e-6
    assert relerr(ot.grad.cpu(), golden_fit[key + "f64_grad"]) < 1e-4
    assert int(status.abs().sum()) == 0
    ref_masked = o.astype(np.float32) ** 2
    ref_masked[:, :, :zr] = 0
    assert np.array_equal(masked.cpu().numpy(), ref_masked)          # bit-exact: fp32 square + mask


@pytest.mark.parametrize("order", [2, 3])
@pytest.mark.parametrize("chol", [False, True])
def test_wls_bp_vs_golden(lf, golden_fit, order, chol):
    N, K, H, W = 1, 4, 256, 512
    o = inputs.lane_like_logits(N, K, H, W, seed=12)
    M, _ = lf.geometry.get_homography(256)
    grid = lf.geometry.projective_grid(H, W, M, False)
    zr = fit_oracle.zero_rows_of(H, 0.3)
    ot = dev(o).requires_grad_(True)
    beta, _, _ = lf.fit.fit_lanes(ot, grid.cuda(), zr, order, 0.0, 255.0, "square", use_cholesky=chol)
    gb = np.random.default_rng(6).standard_normal((4, N, order + 1, 1))[..., 0].transpose(1, 0, 2)
    (beta * dev(gb)).sum().backward()
    key = "bp_wls_o%d_c0_f64" % order
    ys = np.linspace(5, 175, 9)
    Yv = np.stack([ys ** (order - j) for j in range(order + 1)], 1)
    fa, fb = beta.detach().cpu().numpy() @ Yv.T, golden_fit[key + "_beta"] @ Yv.T
    tol = 1e-6 if order == 2 else 1e-3          # cond(Z) ~ 1e8 / 5e12: see SURVEY 8c noise floor
    assert np.abs(fa - fb).max() < tol * np.abs(fb).max()
    assert relerr(ot.grad.cpu().numpy()[:, :, ::8, ::8], golden_fit[key + "_grad_sample"]) < 100 * tol


@pytest.mark.parametrize("act", ["square", "abs", "relu", "sigmoid", "softplus", "none"])
def test_wls_activations_vs_oracle(lf, act):
    N, K, H, W = 3, 2, 32, 64
    o = inputs.lane_like_logits(N, K, H, W, seed=3) + 0.2
    M, _ = fit_oracle.bev_homography()
    grid = fit_oracle.projective_grid(H, W, M.astype(np.float32), True, np.float32)
    zr = 10
    ot = dev(o).requires_grad_(True)
    beta, _, _ = lf.fit.fit_lanes(ot, dev(grid), zr, 2, 0.0, 1.0, act)
    gb = np.random.default_rng(1).standard_normal((N, K, 3))
    (beta * dev(gb)).sum().backward()
    c = fit_oracle.wls_forward(o, grid.astype(np.float64), zr, 2, 0.0, 1.0, act)
    g = fit_oracle.wls_backward(c, gb)
    assert relerr(beta.detach().cpu(), c["beta"]) < 1e-5
    assert relerr(ot.grad.cpu(), g) < 1e-4


def test_wls_full_size_and_known_answers(lf):
    """C2-size maps.  Known answers of SURVEY 8c(ii),(iii); linearity-in-weight property:
    scaling all logits of a lane by a constant does not change its beta."""
    N, K, H, W = 32, 2, 256, 512
    M, _ = lf.geometry.bev_homography()
    grid = lf.geometry.projective_grid(H, W, M, True).cuda()
    zr = 77
    o = torch.ones(N, K, H, W, device="cuda")
    col = torch.arange(W, device="cuda", dtype=torch.float32) / 511
    # weight map W = col/511 (the WLS layer squares it again): logit = sqrt(W) under 'square'
    o[1, 0] = (col ** 0.5)[None, :]
    o[1, 1] = ((1 - col) ** 0.5)[None, :]
    beta, _, _ = lf.fit.fit_lanes(o, grid, zr, 2, 0.0, 1.0, "square", return_masked=False)
    b = beta.cpu().numpy()
    assert np.allclose(b[0, 0], [1.3322623e-07, -1.2208050e-03, 0.49987791730], atol=2e-6)
    assert np.allclose(b[1, 0], [2.6307e-07, 0.31158333448, 0.53115833822], atol=5e-6)
    assert np.allclose(b[1, 1], [1.0515e-07, -0.31402500728, 0.46859750179], atol=5e-6)
    o2 = torch.from_numpy(inputs.lane_like_logits(N, K, H, W, seed=8)).cuda()
    b1, _, _ = lf.fit.fit_lanes(o2, grid, zr, 2, 0.0, 1.0, "square", return_masked=False)
    b2, _, _ = lf.fit.fit_lanes(o2 * 1.7, grid, zr, 2, 0.0, 1.0, "square", return_masked=False)
    assert relerr(b2.cpu(), b1.cpu()) < 1e-6      # fp32 logits * 1.7 round differently; weights scale by 1.7^4
    # determinism: identical bits on a second run
    b3, _, _ = lf.fit.fit_lanes(o2, grid, zr, 2, 0.0, 1.0, "square", return_masked=False)
    assert torch.equal(b1, b3)


def test_wls_singular_raises(lf):
    N, K, H, W = 2, 2, 32, 64
    M, _ = lf.geometry.bev_homography()
    grid = lf.geometry.projective_grid(H, W, M, True).cuda()
    o = torch.zeros(N, K, H, W, device="cuda")          # all-zero weights: Z = 0
    with pytest.raises(RuntimeError):
        lf.fit.fit_lanes(o, grid, 8, 2, 0.0, 1.0, "square")
    beta, _, status = lf.fit.fit_lanes(o, grid, 8, 2, 0.0, 1.0, "square", check_singular=False)
    assert status.cpu().tolist() == [1] * (N * K)
    with pytest.raises(RuntimeError):
        lf.fit.fit_lanes(o, grid, 8, 2, 0.0, 1.0, "square", use_cholesky=True)
    # reg_ls makes it solvable, beta = 0
    beta, _, status = lf.fit.fit_lanes(o, grid, 8, 2, 1e-3, 1.0, "square")
    assert float(beta.abs().max()) == 0.0


def test_wls_module_surface(lf, golden_fit):
    """Reference-compatible Weighted_least_squares.forward(W, grid) -> 4-tuple."""
    from lanedetection_end2end_amd.bev.Networks.LSQ_layer import Weighted_least_squares
    N, K, H, W = 2, 2, 64, 128
    o = inputs.lane_like_logits(N, K, H, W, seed=11)
    zr = fit_oracle.zero_rows_of(H, 0.3)
    masked = o ** 2
    masked[:, :, :zr] = 0
    grid = dev(golden_fit["bev_grid_64x128_f32"]).unsqueeze(0).expand(N, -1, -1)
    ls = Weighted_least_squares(torch.Size([N, K, H, W]), K, 2, False, 0, False)
    b0, b1, b2, b3 = ls(dev(masked), grid)
    assert b2 is None and b3 is None and b0.shape == (N, 3, 1) and b0.dtype == torch.float32
    ref = golden_fit["bev_wls_o2_r0_f64_beta"]
    assert relerr(torch.stack([b0, b1], 1)[..., 0].cpu(), ref) < 1e-5


@pytest.mark.parametrize("order,wf", [(2, "none"), (2, "linear"), (2, "quadratic"), (1, "none")])
@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("f64", torch.float64)])
def test_area_loss(lf, golden_fit, order, wf, tag, dtype):
    beta = golden_fit["area_beta"][:, : order + 1]
    gt = golden_fit["area_gt"][:, : order + 1]
    b = dev(beta, dtype).requires_grad_(True)
    L = lf.losses.Area_Loss(order, wf)(b, dev(gt, dtype))
    (3.0 * L).backward()
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    assert abs(float(L) - golden_fit["area_o%d_%s_f64_loss" % (order, wf)]) < tol
    assert relerr(b.grad.cpu() / 3.0, golden_fit["area_o%d_%s_f64_grad" % (order, wf)]) < max(tol, 1e-10) * 10
    assert L.dtype == dtype and b.grad.shape == b.shape


def test_area_loss_no_lane(lf):
    b = torch.rand(4, 3, 1, device="cuda", requires_grad=True)
    L = lf.losses.Area_Loss(2, "none")(b, torch.zeros(4, 3, device="cuda"))
    L.backward()
    assert float(L) == 0.0 and float(b.grad.abs().max()) == 0.0


@pytest.mark.parametrize("order", [2, 3])
def test_backproj_loss(lf, golden_fit, order):
    from argparse import Namespace
    crit = lf.losses.backprojection_loss(Namespace(resize=256, no_mapping=False, order=order, batch_size=6,
                                                   no_cuda=False))
    assert relerr(crit.y_prime.cpu(), golden_fit["bp_yprime_o%d" % order]) < 1e-12
    lanes, valid = inputs.bp_targets(6, 1, 256, seed=31)
    b = dev(golden_fit["bp_loss_o%d_beta" % order]).requires_grad_(True)
    L, xc = crit(b, dev(lanes[:, 0]), dev(valid[:, 0]))
    L.backward()
    assert abs(float(L) - golden_fit["bp_loss_o%d_loss" % order]) < 1e-10 * abs(float(L))
    assert relerr(xc.cpu(), golden_fit["bp_loss_o%d_xcal" % order]) < 1e-12
    assert relerr(b.grad.cpu(), golden_fit["bp_loss_o%d_grad" % order]) < 1e-10
    L0, _ = crit(b, dev(lanes[:, 0]), dev(valid[:, 0] * 0))
    assert float(L0) == 0.0


def test_losses_take_a_lane_of_a_coefficient_tensor_by_stride(lf, golden_fit):
    """Round 6: the loss kernels read a lane's (N, D) coefficient rows out of the fit's (N, K, D) tensor through `beta_stride` instead of
    a copy, and `split_lanes` is one unbind (its backward one stack).  Same loss and gradient, bit for bit, as on contiguous copies
    (reference call pattern: BEV main.py:217-218, BP main.py:297-300 -- one loss call per lane on `beta0 .. beta3`)."""
    from argparse import Namespace
    from lanedetection_end2end_amd import fit
    order = 2
    g = torch.Generator().manual_seed(3)
    beta = (torch.randn(6, 4, order + 1, dtype=torch.float64, generator=g) * 0.1).cuda().requires_grad_(True)
    # Area_Loss (fp32 betas as the BEV tree hands them over), lanes 0 and 1
    gt = torch.rand(6, 2, order + 1, generator=g).cuda()
    crit = lf.losses.Area_Loss(order, "none")
    lanes_ = fit.split_lanes(beta, 4, torch.float32)
    assert not lanes_[1].squeeze(-1).is_contiguous()                       # really the strided path
    L = crit(lanes_[0], gt[:, 0]) + crit(lanes_[1], gt[:, 1])
    L.backward()
    b2 = beta.detach().clone().requires_grad_(True)
    L2 = sum(crit(b2[:, k].float().contiguous().unsqueeze(2), gt[:, k].contiguous()) for k in range(2))
    L2.backward()
    assert float(L) == float(L2) and torch.equal(beta.grad, b2.grad)
    assert float(beta.grad[:, 2:].abs().max()) == 0.0                        # unused lanes: zero rows from the one stack
    # backprojection_loss (fp64 betas), all four lanes
    critb = lf.losses.backprojection_loss(Namespace(resize=256, no_mapping=False, order=order, batch_size=6, no_cuda=False))
    lanes, valid = inputs.bp_targets(6, 4, 256, seed=32)
    lanes, valid = dev(lanes), dev(valid)
    beta.grad = None
    outs = fit.split_lanes(beta, 4, torch.float64)
    Lb = sum(critb(outs[k], lanes[:, k], valid[:, k])[0] for k in range(4)) / 4
    Lb.backward()
    b3 = beta.detach().clone().requires_grad_(True)
    Lc = sum(critb(b3[:, k].contiguous().unsqueeze(2), lanes[:, k], valid[:, k])[0] for k in range(4)) / 4
    Lc.backward()
    assert float(Lb) == float(Lc) and torch.equal(beta.grad, b3.grad) and float(beta.grad.abs().max()) > 0.0


def test_cross_entropy(lf, golden_fit):
    tgt = inputs.seg_targets(2, 8, 16, 3, seed=41)
    z = dev(golden_fit["ce_logits"]).requires_grad_(True)
    crit = lf.losses.CrossEntropyLoss2d(30, seg=True).cuda()
    L = crit(z, dev(tgt).unsqueeze(1))
    L.backward()
    assert abs(float(L) - float(golden_fit["ce_loss"])) < 2e-6 * abs(float(L))
    assert relerr(z.grad.cpu(), golden_fit["ce_grad"]) < 5e-6
    # config-5-like size vs oracle
    N, C, H, W = 2, 3, 128, 256
    zz = np.random.default_rng(4).standard_normal((N, C, H, W)).astype(np.float32) * 3
    tt = inputs.seg_targets(N, H, W, C, seed=42)
    z2 = dev(zz).requires_grad_(True)
    L2 = crit(z2, dev(tt))
    L2.backward()
    Lo, go = fit_oracle.cross_entropy_2d(zz, tt, [1, 30, 30])
    assert abs(float(L2) - Lo) < 1e-5 * Lo and relerr(z2.grad.cpu(), go) < 1e-5
    # a label outside [0, C) raises like nn.NLLLoss does (instead of reading the weight table out of bounds) ...
    bad = tt.copy()
    bad[0, 3, 5] = 255
    # -- asynchronously, like torch's device assert: the call returns, the NEXT call (or flush()) reports it; no host sync
    # inside the step
    crit(dev(zz), dev(bad))
    with pytest.raises(RuntimeError):
        crit.flush()
    crit(dev(zz), dev(bad))
    with pytest.raises(RuntimeError):
        crit(dev(zz), dev(tt))
    crit.check_targets = "always"
    with pytest.raises(RuntimeError):
        crit(dev(zz), dev(bad))
    # ... and with the check off it is counted and carries weight 0 in loss and gradient
    crit.check_targets = False
    z3 = dev(zz).requires_grad_(True)
    L3 = crit(z3, dev(bad))
    L3.backward()
    assert int(lf.ops.CrossEntropy2dFn.last_acc[2]) == 1
    assert float(z3.grad[0, :, 3, 5].abs().max()) == 0.0 and torch.isfinite(L3)


def test_trapezoid_metric(lf, golden_fit):
    b = torch.tensor([[0.1, -0.2, 0.5], [0, 0.1, 0.4]], dtype=torch.float64, device="cuda")
    g = torch.tensor([[0.05, -0.1, 0.45], [0.01, 0.2, 0.5]], dtype=torch.float64, device="cuda")
    tz = lf.losses.polynomial(b.unsqueeze(2)).trapezoidal(lf.losses.polynomial(g))
    assert np.allclose(tz.cpu().numpy(), golden_fit["trapezoid_survey"], atol=1e-12)


def test_gels_function(lf):
    """GELS.apply(A, b) (BP/Networks/gels.py) vs the oracle's restatement and vs torch autograd of the same solve."""
    from lanedetection_end2end_amd.bp.Networks.gels import GELS
    rng = np.random.default_rng(3)
    for D in (2, 3, 4):
        N, P = 3, 5000
        A = rng.standard_normal((N, P, D)).astype(np.float32)
        b = rng.standard_normal((N, P, 1)).astype(np.float32)
        At = dev(A).requires_grad_(True)
        bt = dev(b).requires_grad_(True)
        x = GELS.apply(At, bt)
        go = rng.standard_normal((N, D, 1)).astype(np.float32)
        (x * dev(go)).sum().backward()
        xo, AtA = fit_oracle.gels_forward(A, b)
        gA, gb = fit_oracle.gels_backward(A.astype(np.float64), b.astype(np.float64), xo, AtA, go.astype(np.float64))
        assert x.shape == (N, D, 1)
        assert relerr(x.detach().cpu(), xo) < 1e-6
        assert relerr(At.grad.cpu(), gA) < 1e-5 and relerr(bt.grad.cpu(), gb) < 1e-5
    with pytest.raises(RuntimeError):
        GELS.apply(torch.zeros(1, 64, 3, device="cuda"), torch.zeros(1, 64, 1, device="cuda"))


def test_fused_adam_matches_torch():
    """optim.FusedAdam == torch.optim.Adam (the optimizer the reference builds) over a few steps, with weight decay,
    including a parameter that never receives a gradient."""
    from lanedetection_end2end_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(64, 64, 3, 1), (128,), (13, 3, 3, 3), (5000,), (1,)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    unused_a, unused_b = torch.nn.Parameter(torch.ones(4, device="cuda")), torch.nn.Parameter(torch.ones(4, device="cuda"))
    oa = FusedAdam(pa + [unused_a], lr=1e-2, weight_decay=1e-3)
    ob = torch.optim.Adam(pb + [unused_b], lr=1e-2, weight_decay=1e-3)
    for it in range(5):
        for a, b in zip(pa, pb):
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        assert float((a - b).abs().max()) < 2e-6 * float(b.abs().max())
    assert torch.equal(unused_a, unused_b)
    assert oa.state[pa[0]]["step"] == 5
    # per-parameter step counts (the reference's pretrained schedule switches heads under one optimizer): the unused
    # parameter starts receiving gradients at step 6 and one of the others stops -- torch.optim.Adam restarts the bias
    # correction for the newcomer at 1, so must the fused step
    for it in range(4):
        for i, (a, b) in enumerate(zip(pa, pb)):
            g = torch.randn_like(a)
            a.grad, b.grad = (None, None) if i == 1 else (g.clone(), g.clone())
        g = torch.randn_like(unused_a)
        unused_a.grad, unused_b.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for a, b in zip(pa + [unused_a], pb + [unused_b]):
        assert float((a - b).abs().max()) < 2e-6 * float(b.abs().max())
    assert oa.state[unused_a]["step"] == 4 and oa.state[pa[1]]["step"] == 5 and oa.state[pa[0]]["step"] == 9
    # a torch.optim.Adam state_dict (tensor-valued steps) loads and continues identically
    oc = FusedAdam([torch.nn.Parameter(p.detach().clone()) for p in pb] + [torch.nn.Parameter(unused_b.detach().clone())],
                   lr=1e-2, weight_decay=1e-3)
    import copy
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))      # (load_state_dict keeps same-device tensors by reference)
    pc = oc.param_groups[0]["params"]
    for a, b in zip(pc, pb + [unused_b]):
        g = torch.randn_like(a)
        a.grad, b.grad = g.clone(), g.clone()
    oc.step()
    ob.step()
    for a, b in zip(pc, pb + [unused_b]):
        assert float((a - b).abs().max()) < 2e-6 * float(b.abs().max())


@pytest.mark.parametrize("kind", ["sgd", "rmsprop"])
def test_fused_sgd_rmsprop_match_torch(kind):
    """optim.define_optim('sgd' | 'rmsprop') == the torch optimizers the reference's define_optim builds
    (BEV/Networks/utils.py:414-417: momentum 0.9, weight decay), a parameter without a gradient included."""
    from lanedetection_end2end_amd.optim import define_optim
    torch.manual_seed(1)
    shapes = [(64, 64, 3, 1), (128,), (13, 3, 3, 3), (5000,), (1,)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    ua, ub = torch.nn.Parameter(torch.ones(4, device="cuda")), torch.nn.Parameter(torch.ones(4, device="cuda"))
    oa = define_optim(kind, pa + [ua], 1e-2, 1e-3)
    ob = (torch.optim.SGD(pb + [ub], lr=1e-2, momentum=0.9, weight_decay=1e-3) if kind == "sgd" else
          torch.optim.RMSprop(pb + [ub], lr=1e-2, momentum=0.9, weight_decay=1e-3))
    for it in range(6):
        for a, b in zip(pa, pb):
            g = torch.randn_like(a)
            a.grad, b.grad = g.clone(), g.clone()
        oa.step()
        ob.step()
    for a, b in zip(pa, pb):
        assert float((a - b).abs().max()) < 3e-6 * float(b.abs().max())
    assert torch.equal(ua, ub)
    with pytest.raises(KeyError):
        define_optim("lbfgs", pa, 1e-2, 0.0)


@pytest.mark.parametrize("kind", ["adam", "sgd", "rmsprop"])
def test_fused_optimizers_step_load_step(kind):
    """Mid-run resume / rollback: an optimizer that has ALREADY stepped loads a state_dict (new moment
    buffers, other step counts) and keeps stepping -- the cached device tables must follow the loaded state, i.e. the result
    equals the torch optimizer taken through the same sequence."""
    import copy
    from lanedetection_end2end_amd.optim import define_optim
    torch.manual_seed(2)
    shapes = [(64, 64, 3, 1), (128,), (5000,)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = define_optim(kind, pa, 1e-2, 1e-3)
    ob = {"adam": lambda p: torch.optim.Adam(p, lr=1e-2, weight_decay=1e-3),
          "sgd": lambda p: torch.optim.SGD(p, lr=1e-2, momentum=0.9, weight_decay=1e-3),
          "rmsprop": lambda p: torch.optim.RMSprop(p, lr=1e-2, momentum=0.9, weight_decay=1e-3)}[kind](pb)

    def both_step(n):
        for _ in range(n):
            for a, b in zip(pa, pb):
                g = torch.randn_like(a)
                a.grad, b.grad = g.clone(), g.clone()
            oa.step()
            ob.step()
    both_step(3)
    snap_opt = copy.deepcopy(ob.state_dict())                 # the checkpoint: torch's own format
    snap_par = [b.detach().clone() for b in pb]
    both_step(4)                                              # run on ...
    for a, b, s in zip(pa, pb, snap_par):                     # ... then roll both back to the checkpoint
        a.data.copy_(s)
        b.data.copy_(s)
    oa.load_state_dict(copy.deepcopy(snap_opt))
    ob.load_state_dict(copy.deepcopy(snap_opt))
    both_step(3)
    for a, b in zip(pa, pb):
        assert float((a - b).abs().max()) < 3e-6 * float(b.abs().max())
    if kind == "adam":
        assert oa.state[pa[0]]["step"] == 6


@pytest.mark.parametrize("tree", ["bev", "bp"])
@pytest.mark.parametrize("D,dtype,tol", [(3, torch.float32, 2e-6), (4, torch.float64, 1e-13)])
def test_mse_loss_policy(tree, D, dtype, tol):
    """--loss_policy mse (BEV/Loss_crit.py:52-53,137-150): MSE_Loss through define_loss_crit of either tree == the real
    reference's value and gradient (tests/golden/mse.npz), one launch (lf_mse_loss)."""
    import os
    from argparse import Namespace
    from conftest import GOLDEN
    from oracle.gen_golden_mse import mse_inputs
    mod = __import__("lanedetection_end2end_amd.%s.Loss_crit" % tree, fromlist=["define_loss_crit"])
    opts = Namespace(loss_policy="mse", order=D - 1, weight_funct="none", weight_seg=30, nclasses=2, no_cuda=False, resize=256,
                     no_mapping=False)
    crit, _ = mod.define_loss_crit(opts)
    assert type(crit).__name__ == "MSE_Loss"
    G = np.load(os.path.join(GOLDEN, "mse.npz"))
    p, g = mse_inputs(D)
    pt = torch.from_numpy(p).to(dtype).cuda().requires_grad_(True)
    L = crit(pt, torch.from_numpy(g).to(dtype).cuda())
    (3.0 * L).backward()                                    # the upstream factor reaches the gradient
    tag = "f64"
    assert abs(float(L) - float(G["%s_d%d_%s_loss" % (tree, D, tag)])) < tol * max(float(G["%s_d%d_%s_loss" % (tree, D, tag)]), 1e-30) * 10
    assert relerr(pt.grad.cpu().numpy() / 3.0, G["%s_d%d_%s_grad" % (tree, D, tag)]) < tol * 10
    assert pt.grad.shape == pt.shape
