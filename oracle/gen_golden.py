"""Generate tests/golden/*.npz by running the REAL reference modules on CPU.

Run in the authoring container only (needs /root/reference):
    python -m oracle.gen_golden
Inputs come from ``oracle.inputs`` / ``erfnet_oracle.make_params`` (seeded), so only the
reference's outputs are stored.  fp32 = the reference as shipped; fp64 = the same modules
after ``.double()`` (the "reference formula in fp64" that parity is quoted against).
"""
import os
import sys

import numpy as np
import torch

from . import erfnet_oracle, fit_oracle, inputs, ref_shims

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def gen_homography(ref_bev, ref_bp):
    size, M, M_inv = ref_bev.LSQ_layer.Init_Projective_transform(2, 1, 256)
    out = dict(bev_M_f32=M[0].numpy(), bev_Minv_f32=M_inv[0].numpy())
    for r in (256, 320):
        M, Mi = ref_bp.utils.get_homography(r, False)
        out["bp_M_%d" % r] = M
        out["bp_Minv_%d" % r] = Mi
    return out


def gen_wls_bev(ref, dtype, tag, out):
    """Reference ProjectiveGridGenerator + Weighted_least_squares (+ square, mask) at 64x128."""
    N, K, H, W = 2, 2, 64, 128
    size = torch.Size([N, K, H, W])
    _, M, _ = ref.LSQ_layer.Init_Projective_transform(K, N, H)
    M = M.to(dtype)
    gridgen = ref.LSQ_layer.ProjectiveGridGenerator(size, M, True)
    gridgen.base_grid = gridgen.base_grid.to(dtype)
    grid = gridgen(M)
    if tag == "f32":
        out["bev_grid_64x128_f32"] = grid[0].numpy()
    zero_rows = fit_oracle.zero_rows_of(H, 0.3)
    for order in (0, 1, 2):
        for reg in (0.0, 1e-3):
            ls = ref.LSQ_layer.Weighted_least_squares(size, K, order, True, reg, False)
            ls.tensor_ones = ls.tensor_ones.to(dtype)
            ls.reg_ls = ls.reg_ls.to(dtype)
            o = _t(inputs.lane_like_logits(N, K, H, W, seed=11), dtype).requires_grad_(True)
            masked = (o ** 2).index_fill(2, torch.arange(zero_rows), 0)
            b0, b1, _, _ = ls(masked, grid)
            gb = _t(np.random.default_rng(5).standard_normal((2, N, order + 1, 1)), dtype)
            (b0 * gb[0]).sum().add((b1 * gb[1]).sum()).backward()
            key = "bev_wls_o%d_r%g_%s" % (order, reg, tag)
            out[key + "_beta"] = np.stack([b0.detach().numpy(), b1.detach().numpy()], 1)[..., 0]
            out[key + "_grad"] = o.grad.numpy()


def gen_wls_bp(ref, dtype, tag, out):
    """BP pixel-coordinate grid + WLS (orders 2,3; inverse and GELS paths), 4 lanes, 256x512.

    Full resolution because the literal ``255 - y`` (BP LSQ_layer.py:94) only makes sense at
    resize = 256; the gradient is stored strided ([::8, ::8]) to keep the fixture small.
    """
    N, K, H, W = 1, 4, 256, 512
    size = torch.Size([N, K, H, W])
    M, _ = ref.utils.get_homography(H, False)
    Mt = torch.from_numpy(M).unsqueeze(0).expand(N, 3, 3).to(dtype)
    grid = ref.LSQ_layer.ProjectiveGridGenerator(size, Mt, True)
    if tag == "f32":
        out["bp_grid_256x512_f32_sample"] = grid[0].numpy().reshape(H, W, 2)[::8, ::8].copy()
    zero_rows = fit_oracle.zero_rows_of(H, 0.3)
    for order in (2, 3):
        for chol in (False, True):
            ls = ref.LSQ_layer.Weighted_least_squares(size, K, order, True, 0.0, chol)
            ls.tensor_ones = ls.tensor_ones.to(dtype)
            ls.reg_ls = ls.reg_ls.to(dtype)
            o = _t(inputs.lane_like_logits(N, K, H, W, seed=12), dtype).requires_grad_(True)
            masked = (o ** 2).index_fill(2, torch.arange(zero_rows), 0)
            try:
                betas = ls(masked, grid)
            except RuntimeError as e:   # fp32 Cholesky of a cond~1e8..1e12 matrix: not PD
                print("reference raised for order %d chol %d %s: %s" % (order, chol, tag, str(e)[:80]))
                continue
            gb = _t(np.random.default_rng(6).standard_normal((4, N, order + 1, 1)), torch.float64)
            sum((b * g).sum() for b, g in zip(betas, gb)).backward()
            key = "bp_wls_o%d_c%d_%s" % (order, int(chol), tag)
            out[key + "_beta"] = np.stack([b.detach().numpy() for b in betas], 1)[..., 0]
            out[key + "_grad_sample"] = o.grad.numpy()[:, :, ::8, ::8].copy()


def gen_losses(ref_bev, ref_bp, out):
    rng = np.random.default_rng(21)
    N = 6
    beta = rng.uniform(-0.5, 0.5, (N, 3, 1))
    gt = rng.uniform(-0.5, 0.5, (N, 3))
    gt[1] = 0            # absent lane
    gt[4, 1] = 0         # one zero coefficient also drops the lane
    out["area_beta"] = beta
    out["area_gt"] = gt
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        for order, wf in ((2, "none"), (2, "linear"), (2, "quadratic"), (1, "none")):
            b = _t(beta[:, : order + 1], dtype).requires_grad_(True)
            L = ref_bev.Loss_crit.Area_Loss(order, wf)(b, _t(gt[:, : order + 1], dtype))
            L.backward()
            out["area_o%d_%s_%s_loss" % (order, wf, tag)] = L.detach().numpy()
            out["area_o%d_%s_%s_grad" % (order, wf, tag)] = b.grad.numpy()
    # SURVEY 8c(iv) fixed vectors
    b = torch.tensor([[0.1, -0.2, 0.5], [0, 0.1, 0.4]], dtype=torch.float64).unsqueeze(2)
    g = torch.tensor([[0.05, -0.1, 0.45], [0.01, 0.2, 0.5]], dtype=torch.float64)
    out["area_survey"] = np.array([float(ref_bev.Loss_crit.Area_Loss(2, wf)(b, g))
                                   for wf in ("none", "linear", "quadratic")])
    pa = ref_bev.Loss_crit.polynomial(b.squeeze(2))
    pb = ref_bev.Loss_crit.polynomial(g)
    out["trapezoid_survey"] = pa.trapezoidal(pb).numpy()
    # back-projection loss
    for order in (2, 3):
        args = ref_shims.default_args("bp", batch_size=N, order=order)
        crit = ref_bp.Loss_crit.backprojection_loss(args)
        out["bp_yprime_o%d" % order] = crit.y_prime[0, :, 0].numpy()
        out["bp_Y_o%d" % order] = crit.Y[0].numpy()
        bt = rng.standard_normal((N, order + 1, 1)) * np.array([1e-6, 1e-3, 0.3, 1.0])[-(order + 1):, None]
        bt[:, -1, 0] += 256
        lanes, valid = inputs.bp_targets(N, 1, 256, seed=31)
        b = torch.from_numpy(bt).requires_grad_(True)
        L, xc = crit(b, torch.from_numpy(lanes[:, 0]), torch.from_numpy(valid[:, 0]))
        L.backward()
        out["bp_loss_o%d_beta" % order] = bt
        out["bp_loss_o%d_loss" % order] = L.detach().numpy()
        out["bp_loss_o%d_xcal" % order] = xc.detach().numpy()
        out["bp_loss_o%d_grad" % order] = b.grad.numpy()
    # class-weighted CE (BP define_loss_crit :64-65 builds nn.CrossEntropyLoss(weights))
    z = rng.standard_normal((2, 3, 8, 16)).astype(np.float32)
    tgt = inputs.seg_targets(2, 8, 16, 3, seed=41)
    zt = torch.from_numpy(z).requires_grad_(True)
    L = torch.nn.CrossEntropyLoss(torch.tensor([1.0, 30.0, 30.0]))(zt, torch.from_numpy(tgt))
    L.backward()
    out["ce_logits"] = z
    out["ce_loss"] = L.detach().numpy()
    out["ce_grad"] = zt.grad.numpy()


def _load_params(model, P):
    sd = model.state_dict()
    assert list(sd.keys()) == list(P.keys()), "state_dict key order differs from oracle param_spec"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
    model.load_state_dict(P)


def gen_backbone(ref, out):
    """Reference ERFNet Net, train and eval mode, dropout off, 2x3x64x128, fp32 and fp64."""
    N, H, W, Cout = 2, 64, 128, 2
    x = inputs.images(N, H, W, seed=51)
    gy = np.random.default_rng(52).standard_normal((N, Cout, H, W))
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        P = erfnet_oracle.make_params(seed=3, out_channels=Cout)   # fp32 values, also for the fp64 run
        net = ref.ERFNet.Net(in_channels=3, out_channels=Cout)
        _load_params(net, P)
        net = ref_shims.disable_dropout(net.to(dtype))
        net.train()
        enc, dec = net(_t(x, dtype), True)
        (dec * _t(gy, dtype)).sum().backward()
        out["bb_train_enc_" + tag] = enc.detach().numpy()
        out["bb_train_dec_" + tag] = dec.detach().numpy()
        sd = net.state_dict()
        for k in ("encoder.initial_block.bn.running_mean", "encoder.layers.9.bn2.running_var",
                  "decoder.layers.3.bn.running_var"):
            out["bb_train_%s_%s" % (k, tag)] = sd[k].numpy().copy()
        norms, keys = [], []
        for k, p in net.named_parameters():
            keys.append(k)
            norms.append(float(p.grad.double().norm()) if p.grad is not None else -1.0)
        out["bb_grad_norms_" + tag] = np.array(norms)
        if tag == "f32":
            out["bb_grad_keys"] = np.array(keys)
        for k in ("encoder.initial_block.conv.weight", "encoder.layers.0.conv.bias",
                  "encoder.layers.3.bn1.weight", "encoder.layers.10.conv3x1_2.bias",
                  "encoder.layers.14.conv1x3_2.weight", "decoder.layers.0.conv.bias",
                  "decoder.layers.3.conv.weight", "decoder.layers.5.bn2.bias",
                  "decoder.output_conv.weight", "decoder.output_conv.bias"):
            out["bb_grad_%s_%s" % (k, tag)] = dict(net.named_parameters())[k].grad.numpy().copy()
        net.eval()
        with torch.no_grad():
            enc, dec = net(_t(x, dtype), True)
        out["bb_eval_dec_" + tag] = dec.numpy()


def gen_e2e_bev(ref, out):
    """C1: reference LSQ_layer.Net + Area_Loss, 4x3x256x512, 2 lanes, fp32 and fp64."""
    N, R = 4, 256
    x = inputs.images(N, R, 2 * R, seed=61)
    gt = inputs.bev_gt_params(N, seed=62)
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        args = ref_shims.default_args("bev", batch_size=N)
        model = ref.LSQ_layer.Net(args)
        P = erfnet_oracle.make_params(seed=4, out_channels=2)
        _load_params(model.net, P)
        model = ref_shims.disable_dropout(model.to(dtype))
        model.M = model.M.to(dtype)
        model.project_layer.base_grid = model.project_layer.base_grid.to(dtype)
        model.ls_layer.tensor_ones = model.ls_layer.tensor_ones.to(dtype)
        model.ls_layer.reg_ls = model.ls_layer.reg_ls.to(dtype)
        model.train()
        crit = ref.Loss_crit.Area_Loss(2, "none")
        b0, b1, _, _, masked, M, output, _, _ = model(_t(x, dtype), True)
        output.retain_grad()
        loss = crit(b0, _t(gt[:, 0], dtype)) + crit(b1, _t(gt[:, 1], dtype))
        loss.backward()
        out["e2e_bev_beta_" + tag] = np.stack([b0.detach().numpy(), b1.detach().numpy()], 1)[..., 0]
        out["e2e_bev_loss_" + tag] = loss.detach().numpy()
        out["e2e_bev_logits_sample_" + tag] = output.detach().numpy()[:, :, ::16, ::16].copy()
        out["e2e_bev_dlogits_sample_" + tag] = output.grad.numpy()[:, :, ::16, ::16].copy()
        out["e2e_bev_grad_norms_" + tag] = np.array(
            [float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in model.named_parameters()])
        if tag == "f32":
            out["e2e_bev_grad_keys"] = np.array([k for k, _ in model.named_parameters()])


def gen_e2e_bp(ref, out):
    """BP LSQ_layer.Net + backprojection_loss, 2x3x256x512, 4 lanes, order 2, fp32 trunk."""
    N, R, K = 2, 256, 4
    x = inputs.images(N, R, 2 * R, seed=71)
    lanes, valid = inputs.bp_targets(N, K, R, seed=72)
    args = ref_shims.default_args("bp", batch_size=N, nclasses=K, mask_percentage=0.2)
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        model = ref.LSQ_layer.Net(args)
        P = erfnet_oracle.make_params(seed=5, out_channels=K)
        _load_params(model.net, P)
        model = ref_shims.disable_dropout(model.to(dtype))
        model.grid = model.grid.to(dtype)
        model.ls_layer.tensor_ones = model.ls_layer.tensor_ones.to(dtype)
        model.ls_layer.reg_ls = model.ls_layer.reg_ls.to(dtype)
        model.train()
        crit = ref.Loss_crit.backprojection_loss(args)
        gt_line = torch.zeros(N, K)
        res = model(_t(x, dtype), gt_line, True)
        betas, output = res[:4], res[5]
        output.retain_grad()
        loss = 0
        xcals = []
        for k in range(K):
            l, xc = crit(betas[k], torch.from_numpy(lanes[:, k]), torch.from_numpy(valid[:, k]))
            loss = loss + l
            xcals.append(xc.detach().numpy())
        loss = loss / K
        loss.backward()
        out["e2e_bp_beta_" + tag] = np.stack([b.detach().numpy() for b in betas], 1)[..., 0]
        out["e2e_bp_loss_" + tag] = loss.detach().numpy()
        out["e2e_bp_xcal_" + tag] = np.stack(xcals, 1)
        out["e2e_bp_dlogits_sample_" + tag] = output.grad.numpy()[:, :, ::16, ::16].copy()


def main():
    assert ref_shims.available(), "needs /root/reference"
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    bev = ref_shims.load("bev")
    fit, bb, e2e = {}, {}, {}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        gen_wls_bev(bev, dtype, tag, fit)
    gen_backbone(bev, bb)
    gen_e2e_bev(bev, e2e)
    bev_loss, bev_lsq = bev.Loss_crit, bev.LSQ_layer
    bp = ref_shims.load("bp")
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        gen_wls_bp(bp, dtype, tag, fit)
    gen_e2e_bp(bp, e2e)
    bev2 = type("ns", (), dict(Loss_crit=bev_loss, LSQ_layer=bev_lsq))
    fit.update(gen_homography(bev2, bp))
    gen_losses(bev2, bp, fit)
    for name, d in (("fit_head", fit), ("backbone", bb), ("e2e", e2e)):
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **d)
        print(path, "%.1f KB" % (os.path.getsize(path) / 1024), len(d), "arrays")


if __name__ == "__main__":
    sys.exit(main())
