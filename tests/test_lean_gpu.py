"""Every epilogue flag set of the fp32 16 -> 16 channel convolutions (tapgemm_lean_kernel; BEV/Networks/ERFNet.py:29-60 at the decoder's
full-resolution stage, 20 launches per step) ONE LAUNCH AT A TIME through the C ABI against torch fp64: plain / ReLU forward, BN forward
sums, BN+ReLU operand prologue, data gradient plain / times the ReLU mask / plus the residual gradient, mask + BN-backward sums, mask by a
recomputed BatchNorm + BN-backward sums, and the three-tensor epilogue that closes a block's backward -- values to fp32 rounding, the
per-tile partial rows (channel-major [2][C][rows]) summed in fp64 against the column sums of the stored values.  (The whole-network tests cover the same
kernels only through the chain of a full pass; lf_debug_conv1d_epi is the hook that launches one of them.)"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

# (N, H, W, axis, dilation): whole-row waves (W % 64 == 0) and ragged ones; 90 and 9 tiles; one image whose last tile is partial
SHAPES = [(2, 64, 128, 0, 1), (2, 64, 128, 1, 1), (3, 40, 192, 1, 2), (3, 40, 192, 0, 3), (1, 18, 128, 0, 1), (2, 24, 80, 1, 1), (1, 15, 48, 0, 2)]


@pytest.mark.parametrize("shape", SHAPES)
def test_every_epilogue_of_the_16_channel_kernel(shape):
    import torch.nn.functional as F
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
    nrows = (N * H * W + 255) // 256
    nan = lambda: torch.full_like(x, float("nan"))
    rows = lambda: torch.full((2, C, nrows), float("nan"), device="cuda")      # channel-major partial rows: [kind][channel][row]
    epi = lambda *args: lib.lf_debug_conv1d_epi(*args, N, H, W, C, axis, d, P(scratch), st)

    y0, y1, y8, yp, g0, g2, g4, g34, g48, g38 = (nan() for _ in range(10))
    s8, s34, s48, s38 = rows(), rows(), rows(), rows()
    assert epi(P(x), P(w), P(b), P(y0), 0, 0, None, None, None, None, None, None) == 0, lib.lf_last_error().decode()
    assert epi(P(x), P(w), P(b), P(y1), 0, 1, None, None, None, None, None, None) == 0
    assert epi(P(x), P(w), P(b), P(y8), 0, 8, None, None, None, None, None, P(s8)) == nrows, lib.lf_last_error().decode()
    _lib.check(lib.lf_debug_conv1d_fwd_pro(P(x), P(w), P(b), P(sc), P(sh), P(yp), N, H, W, C, axis, d, P(scratch), st), "fwd pro")
    assert epi(P(gy), P(w), None, P(g0), 1, 0, None, None, None, None, None, None) == 0
    assert epi(P(gy), P(w), None, P(g2), 1, 2, P(mask), None, None, None, None, None) == 0
    assert epi(P(gy), P(w), None, P(g4), 1, 4, None, P(add), None, None, None, None) == 0
    assert epi(P(gy), P(w), None, P(g34), 1, 34, P(mask), None, P(aux), None, None, P(s34)) == nrows
    assert epi(P(gy), P(w), None, P(g48), 1, 48, None, None, P(aux), P(sc), P(sh), P(s48)) == nrows
    assert epi(P(gy), P(w), None, P(g38), 1, 38, P(mask), P(add), P(aux), None, None, P(s38)) == nrows
    torch.cuda.synchronize()

    w4 = (w.view(C, C, 3, 1) if axis == 0 else w.view(C, C, 1, 3)).double()
    pad, dil = ((d, 0), (d, 1)) if axis == 0 else ((0, d), (1, d))
    nchw = lambda t: t.double().permute(0, 3, 1, 2)
    nhwc = lambda t: t.permute(0, 2, 3, 1)
    conv = nhwc(F.conv2d(nchw(x), w4, b.double(), padding=pad, dilation=dil))
    xp = torch.relu(x.double() * sc.double() + sh.double())
    convp = nhwc(F.conv2d(nchw(xp), w4, b.double(), padding=pad, dilation=dil))
    dg = nhwc(torch.nn.grad.conv2d_input((N, C, H, W), w4, nchw(gy), padding=pad, dilation=dil))
    m = mask.double() > 0
    mbn = (aux.double() * sc.double() + sh.double()) > 0
    want = {"fwd": (y0, conv), "fwd + relu": (y1, torch.relu(conv)), "fwd + BN sums": (y8, conv), "bn-relu prologue + relu": (yp, torch.relu(convp)),
            "dgrad": (g0, dg), "dgrad * mask": (g2, dg * m), "dgrad + add": (g4, dg + add.double()), "dgrad * mask + sums": (g34, dg * m),
            "dgrad * recomputed-BN mask + sums": (g48, dg * mbn), "(dgrad + add) * mask + sums": (g38, (dg + add.double()) * m)}
    for name, (got, ref) in want.items():
        assert torch.isfinite(got).all(), name
        err = float((got.double() - ref).abs().max()) / float(ref.abs().max())
        assert err < 2e-6, "%s: %.2e" % (name, err)
    # BN forward rows (round 6, centred): [0][c][r] = sum v, [1][c][r] = M2 = sum (v - mean_row)^2 over row r's own 256-pixel tile
    # (the last one ragged) -- each row against fp64 on the stored values, M2 relative to itself (no (mean / sigma)^2 term)
    flat = y8.double().reshape(-1, C)
    for r in range(nrows):
        t = flat[r * 256: (r + 1) * 256]
        m2 = ((t - t.mean(0)) ** 2).sum(0)
        assert float(((s8[0, :, r].double() - t.sum(0)).abs() / t.abs().sum(0)).max()) < 2e-6, ("BN forward sums, row %d" % r)
        assert float(((s8[1, :, r].double() - m2).abs() / m2).max()) < 5e-6, ("BN forward M2, row %d" % r)
    # BN backward rows: [0] = sum v, [1] = sum v * aux (raw), over the values as stored
    for name, st_rows, v, second in (("mask + BN-backward sums", s34, g34, aux),
                                     ("recomputed-BN mask + sums", s48, g48, aux), ("three-tensor epilogue", s38, g38, aux)):
        assert torch.isfinite(st_rows).all(), name
        got = st_rows.double().sum(2)
        s1, s2 = v.double().sum((0, 1, 2)), (v.double() * second.double()).sum((0, 1, 2))
        a1, a2 = v.double().abs().sum((0, 1, 2)), (v.double() * second.double()).abs().sum((0, 1, 2))
        assert float(((got[0] - s1).abs() / a1).max()) < 2e-6 and float(((got[1] - s2).abs() / a2).max()) < 2e-6, name
