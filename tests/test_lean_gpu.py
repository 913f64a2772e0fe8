"""The persistent form of the fp32 16 -> 16 channel convolutions (tapgemm_lean_p_kernel; BEV/Networks/ERFNet.py:29-60 at the
decoder's full-resolution stage, 20 launches per step) against tapgemm_lean_kernel, the one-tile-per-workgroup form it replaces:
the arithmetic, the summation order and the statistics rows are the same by construction, so EVERYTHING must agree bit for bit --
kernel by kernel through the C ABI for the variants that have an entry point (plain / ReLU forward, BN+ReLU operand prologue, data
gradient with and without the ReLU mask, the three-tensor epilogue with its BN-backward partial rows) and, for the rest (residual
add, BN forward statistics, recomputed-BN mask), through a whole train-mode forward + backward of the network: logits, every
parameter gradient, every BatchNorm running statistic.  The one-tile form itself is held to the fp64 oracle by the backbone tests."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (N, H, W, axis, dilation): tile counts 64 (8 per XCD range), 90 (ranges of 11 and 12 tiles, row width 192 = 3 x 64), 128; 9 (ranges of 1 and 2)
SHAPES = [(2, 64, 128, 0, 1), (2, 64, 128, 1, 1), (3, 40, 192, 1, 2), (3, 40, 192, 0, 3), (1, 128, 256, 1, 1), (1, 18, 128, 0, 1)]
MODES = [1, 2, 3]          # lf_debug_set_lean_p: one operand register set at 4 workgroups per CU (shipped) / two sets / one set at 3 per CU


@pytest.mark.parametrize("shape", SHAPES)
def test_persistent_16_channel_kernels_bit_identical(shape):
    from lanedetection_end2end_amd import _lib
    lib = _lib.load()
    st = _lib.stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    N, H, W, axis, d = shape
    C = 16
    torch.manual_seed(H + W + axis)
    x = torch.randn(N, H, W, C, device="cuda")
    gy = torch.randn(N, H, W, C, device="cuda")
    mask = torch.randn(N, H, W, C, device="cuda")
    add = torch.randn(N, H, W, C, device="cuda")
    aux = torch.randn(N, H, W, C, device="cuda")
    w = torch.randn(C, C, 3, device="cuda") * (2.0 / (3 * C)) ** 0.5
    b = torch.randn(C, device="cuda")
    sc = torch.rand(C, device="cuda") + 0.5
    sh = torch.randn(C, device="cuda") * 0.5           # relu(0 * sc + sh) != 0: padding must be zero AFTER the transform
    scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096, device="cuda")
    nrows_max = (N * H * W + 255) // 256

    def run(mode):
        lib.lf_debug_set_lean_p(mode)
        nan = lambda: torch.full_like(x, float("nan"))
        y0, y1, yp, g0, g1, g3 = nan(), nan(), nan(), nan(), nan(), nan()
        stats = torch.full((nrows_max, 2, C), float("nan"), device="cuda")
        _lib.check(lib.lf_conv1d_fwd(P(x), P(w), P(b), P(y0), N, H, W, C, axis, d, 0, P(scratch), st), "fwd")
        _lib.check(lib.lf_conv1d_fwd(P(x), P(w), P(b), P(y1), N, H, W, C, axis, d, 1, P(scratch), st), "fwd relu")
        _lib.check(lib.lf_debug_conv1d_fwd_pro(P(x), P(w), P(b), P(sc), P(sh), P(yp), N, H, W, C, axis, d, P(scratch), st), "fwd pro")
        _lib.check(lib.lf_conv1d_bwd_data(P(gy), P(w), None, P(g0), N, H, W, C, axis, d, P(scratch), st), "dgrad")
        _lib.check(lib.lf_conv1d_bwd_data(P(gy), P(w), P(mask), P(g1), N, H, W, C, axis, d, P(scratch), st), "dgrad mask")
        rows = lib.lf_debug_conv1d_bwd_data_epi3(P(gy), P(w), P(mask), P(add), P(aux), P(g3), P(stats), N, H, W, C, axis, d, P(scratch), st)
        assert rows > 0, lib.lf_last_error().decode()
        # the flag sets without an entry point of their own: BN forward sums (8), residual add (4), mask + BN-backward sums (34),
        # recomputed-BN mask + BN-backward sums (48)
        y8, g4, g34, g48 = nan(), nan(), nan(), nan()
        st8, st34, st48 = (torch.full((nrows_max, 2, C), float("nan"), device="cuda") for _ in range(3))
        epi = lambda *args: lib.lf_debug_conv1d_epi(*args, N, H, W, C, axis, d, P(scratch), st)
        r8 = epi(P(x), P(w), P(b), P(y8), 0, 8, None, None, None, None, None, P(st8))
        r4 = epi(P(gy), P(w), None, P(g4), 1, 4, None, P(add), None, None, None, None)
        r34 = epi(P(gy), P(w), None, P(g34), 1, 34, P(mask), None, P(aux), None, None, P(st34))
        r48 = epi(P(gy), P(w), None, P(g48), 1, 48, None, None, P(aux), P(sc), P(sh), P(st48))
        assert r8 == rows and r4 == 0 and r34 == rows and r48 == rows, lib.lf_last_error().decode()
        torch.cuda.synchronize()
        return y0, y1, yp, g0, g1, g3, stats[:rows].clone(), y8, st8[:rows].clone(), g4, g34, st34[:rows].clone(), g48, st48[:rows].clone()

    names = ["fwd", "fwd + relu", "bn-relu prologue", "dgrad", "dgrad * mask", "three-tensor epilogue", "its partial rows",
             "fwd + BN sums", "its rows", "dgrad + add", "dgrad * mask + BN-backward sums", "its rows", "dgrad * recomputed-BN mask + sums",
             "its rows"]
    try:
        ref = run(0)
        assert all(torch.isfinite(t).all() for t in ref)
        # the reference form against torch (loose: the oracle-level checks live in the backbone tests)
        import torch.nn.functional as F
        w4 = (w.view(C, C, 3, 1) if axis == 0 else w.view(C, C, 1, 3)).double()
        pad, dil = ((d, 0), (d, 1)) if axis == 0 else ((0, d), (1, d))
        want = F.conv2d(x.double().permute(0, 3, 1, 2), w4, b.double(), padding=pad, dilation=dil).permute(0, 2, 3, 1)
        assert float((ref[0].double() - want).abs().max()) < 1e-5 * float(want.abs().max())
        assert float((ref[8].double().sum(0)[0] - want.sum((0, 1, 2))).abs().max()) < 1e-4 * float(want.abs().sum((0, 1, 2)).max())
        gwant = torch.nn.grad.conv2d_input(tuple(want.permute(0, 3, 1, 2).shape), w4, gy.double().permute(0, 3, 1, 2), padding=pad,
                                           dilation=dil).permute(0, 2, 3, 1)
        keep = (aux.double() * sc.double() + sh.double()) > 0
        assert float((ref[12].double() - gwant * keep).abs().max()) < 1e-5 * float(gwant.abs().max())
        assert float((ref[9].double() - (gwant + add.double())).abs().max()) < 1e-5 * float(gwant.abs().max())
        for mode in MODES:
            for it in range(3):
                got = run(mode)
                for name, u, v in zip(names, got, ref):
                    assert torch.equal(u, v), "mode %d launch %d: %s differs from the one-tile kernel (max %.3e)" % (
                        mode, it, name, float((u - v).abs().max()))
    finally:
        lib.lf_debug_set_lean_p(1)


def test_whole_step_bit_identical_with_and_without_the_persistent_kernels():
    """Train-mode forward + backward of the network at 4 x 3 x 128 x 256 (16-channel stage 64 x 128, 128 tiles): the variants without
    a kernel-level entry point (residual add, BN statistics, recomputed-BN mask + BN-backward sums) are covered by demanding that the
    logits, all parameter gradients and the BatchNorm running statistics do not change in a single bit."""
    from oracle import erfnet_oracle
    from lanedetection_end2end_amd import _lib, erfnet
    lib = _lib.load()
    P = erfnet_oracle.make_params(seed=9, out_channels=2, pretrained=False)
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.random((4, 3, 128, 256), dtype=np.float32)).cuda()
    gl = torch.from_numpy(rng.standard_normal((4, 2, 128, 256)).astype(np.float32)).cuda()

    def run(mode):
        lib.lf_debug_set_lean_p(mode)
        net = erfnet.Net(in_channels=3, out_channels=2, pretrained=False)
        net.load_state_dict(P)
        net = net.cuda().train()
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout2d):
                m.p = 0
        _, logits = net(x, False)
        (logits * gl).sum().backward()
        torch.cuda.synchronize()
        out = {"logits": logits.detach().clone()}
        for k, p in net.named_parameters():
            if p.grad is not None:
                out["grad " + k] = p.grad.clone()
        for k, v in net.named_buffers():
            out["buffer " + k] = v.clone()
        return out

    try:
        ref = run(0)
        assert len([k for k in ref if k.startswith("grad ")]) > 180
        for mode in MODES:
            got = run(mode)
            assert got.keys() == ref.keys()
            bad = [k for k in ref if not torch.equal(got[k], ref[k])]
            assert not bad, "mode %d: %d tensors differ, first %s" % (mode, len(bad), bad[0])
    finally:
        lib.lf_debug_set_lean_p(1)
