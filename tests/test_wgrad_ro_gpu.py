"""The read-once bf16 weight gradient (csrc/lf_wgrad_ro.hip; BEV/Networks/ERFNet.py:29-37,44-60: the 3x1 / 1x3 convolutions of
non_bottleneck_1d at 64 and 128 channels) through the C ABI: against the fp64 weight gradient of the stored bf16 operands (the
products of bf16 values are exact in fp32, so the result is exact to fp32 summation), against tapwgrad_kernel's job form on the same
launch, with the BN+ReLU operand prologue, and -- its LDS ring is ordered by a hand-counted s_waitcnt vmcnt(N) + bare s_barrier --
bit-identical on every launch of a repeat loop at the shapes where such rings failed before (row width 80: a dilation-8 / 16 tap
empties whole DMA instructions; 48: groups next to both row edges; pixel ranges that straddle images)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

# (N, C, H, W, axis, dilation)
SHAPES = [(3, 64, 20, 48, 0, 2), (3, 64, 20, 48, 1, 16), (3, 128, 40, 80, 1, 8), (3, 128, 40, 80, 0, 16), (2, 128, 16, 32, 0, 4),
          (5, 64, 12, 32, 1, 1), (1, 64, 4, 16, 1, 1), (1, 128, 2, 16, 0, 1), (7, 128, 9, 16, 1, 2), (64, 64, 80, 160, 0, 1),
          (64, 128, 40, 80, 1, 16)]


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


@pytest.mark.parametrize("shape", SHAPES)
def test_read_once_weight_gradient(shape):
    import torch.nn.functional as F
    from lanedetection_end2end_amd import _lib
    lib = _lib.load()
    st = _lib.stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    N, C, H, W, axis, d = shape
    torch.manual_seed(C + axis + W)
    x = torch.randn(N, H, W, C, device="cuda").bfloat16()
    gy = torch.randn(N, H, W, C, device="cuda").bfloat16()
    sc = torch.rand(C, device="cuda") + 0.5
    sh = torch.randn(C, device="cuda") * 0.5          # relu(0 * sc + sh) != 0: padding must stay zero AFTER the transform
    scratch = torch.full((lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096,), float("nan"), device="cuda")
    wshape = (C, C, 3, 1) if axis == 0 else (C, C, 1, 3)
    pad, dil = ((d, 0), (d, 1)) if axis == 0 else ((0, d), (1, d))

    def run(mode, caps=(0, 0)):
        lib.lf_debug_set_wgrad_ro(mode, caps[0], caps[1])
        gw, gb, gwp, gbp = (torch.full((C, C, 3), float("nan"), device="cuda"), torch.full((C,), float("nan"), device="cuda"),
                            torch.full((C, C, 3), float("nan"), device="cuda"), torch.full((C,), float("nan"), device="cuda"))
        _lib.check(lib.lf_conv1d_bwd_weight(P(x), P(gy), P(gw), P(gb), N, H, W, C, axis, d, P(scratch), st), "wgrad")
        _lib.check(lib.lf_debug_conv1d_wgrad_pro(P(x), P(gy), P(sc), P(sh), P(gwp), P(gbp), N, H, W, C, axis, d, P(scratch), st), "wgrad pro")
        torch.cuda.synchronize()
        return gw, gb, gwp, gbp

    try:
        lib.lf_debug_set_ops_precision(2)
        old = run(0)
        new = run(1)
        assert all(torch.isfinite(t).all() for t in old + new)
        xn = x.double().permute(0, 3, 1, 2).contiguous()
        gn = gy.double().permute(0, 3, 1, 2).contiguous()
        wref = torch.nn.grad.conv2d_weight(xn, wshape, gn, padding=pad, dilation=dil).view(C, C, 3)
        bref = gn.sum((0, 2, 3))
        e_w, e_b = _rel(new[0], wref), _rel(new[1], bref)
        # prologue: the operand is relu(fma(x, sc, sh)) rounded to bf16; torch's two-rounding multiply-add puts a handful of operands on
        # the other side of a bf16 rounding boundary -> compared in norm against torch, sharply against the job form (same fma)
        xa = torch.relu(x.float() * sc + sh).bfloat16().double().permute(0, 3, 1, 2).contiguous()
        pref = torch.nn.grad.conv2d_weight(xa, wshape, gn, padding=pad, dilation=dil).view(C, C, 3)
        e_p, e_po, e_pb = _rel(new[2], pref), _rel(new[2], old[2]), _rel(new[3], bref)
        print("read-once wgrad %r: vs fp64 gw %.1e gb %.1e | prologue vs torch %.1e, vs job form %.1e (job form vs fp64 %.1e)"
              % (shape, e_w, e_b, e_p, e_po, _rel(old[0], wref)))
        assert e_w < 3e-6 and e_b < 3e-6 and e_pb < 3e-6 and e_po < 3e-6 and e_p < 3e-3
        # every launch equals the first one bit for bit, also with fewer / more workgroups (other pixel ranges per workgroup)
        repeats = 10 if N * H * W > 100000 else 40
        for it in range(repeats):
            again = run(1)
            assert all(torch.equal(u, v) for u, v in zip(again, new)), "launch %d differs from the first" % it
        few = run(1, (64, 64))
        assert _rel(few[0], wref) < 3e-6 and _rel(few[2], old[2]) < 3e-6 and _rel(few[1], bref) < 3e-6
    finally:
        lib.lf_debug_set_ops_precision(0)
        lib.lf_debug_set_wgrad_ro(1, 512, 256)


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("shape", [(3, 16, 20, 48, 1, 2), (3, 16, 37, 16, 0, 1), (2, 16, 32, 64, 1, 1), (32, 16, 128, 256, 0, 1)])
def test_wgrad16_lds_ring_kernels(shape, mode):
    """The 16 x 16 channel weight gradients through per-wave LDS rings (round 5: tapwgrad16_f32_kernel on fp32 tensors; the BN+ReLU
    operand prologue in it and in the bf16 tapwgrad16_tr_kernel -- the block's third convolution, ERFNet.py:53-56): against the fp64
    weight gradient (fp32 tensors: fp32 accuracy; bf16 tensors: exact products of the stored operands; the prologue's operand is
    relu(fma(x, sc, sh)), rounded to bf16 in mode 2), bias gradient, every launch of a repeat loop bit-identical (hand-counted
    vmcnt waits), shapes with an odd number of groups, padding taps that empty whole DMA instructions, ranges straddling images."""
    from lanedetection_end2end_amd import _lib
    lib = _lib.load()
    st = _lib.stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    N, C, H, W, axis, d = shape
    torch.manual_seed(W + axis)
    cast = (lambda t: t.bfloat16()) if mode == 2 else (lambda t: t)
    x, gy = cast(torch.randn(N, H, W, C, device="cuda")), cast(torch.randn(N, H, W, C, device="cuda"))
    sc = torch.rand(C, device="cuda") + 0.5
    sh = torch.randn(C, device="cuda") * 0.5
    scratch = torch.full((lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096,), float("nan"), device="cuda")
    wshape = (C, C, 3, 1) if axis == 0 else (C, C, 1, 3)
    pad, dil = ((d, 0), (d, 1)) if axis == 0 else ((0, d), (1, d))

    def run():
        out = [torch.full((C, C, 3), float("nan"), device="cuda"), torch.full((C,), float("nan"), device="cuda"),
               torch.full((C, C, 3), float("nan"), device="cuda"), torch.full((C,), float("nan"), device="cuda")]
        _lib.check(lib.lf_conv1d_bwd_weight(P(x), P(gy), P(out[0]), P(out[1]), N, H, W, C, axis, d, P(scratch), st), "wgrad")
        _lib.check(lib.lf_debug_conv1d_wgrad_pro(P(x), P(gy), P(sc), P(sh), P(out[2]), P(out[3]), N, H, W, C, axis, d, P(scratch), st), "wgrad pro")
        torch.cuda.synchronize()
        return out

    try:
        lib.lf_debug_set_ops_precision(mode)
        first = run()
        assert all(torch.isfinite(t).all() for t in first)
        xn, gn = x.double().permute(0, 3, 1, 2).contiguous(), gy.double().permute(0, 3, 1, 2).contiguous()
        wref = torch.nn.grad.conv2d_weight(xn, wshape, gn, padding=pad, dilation=dil).view(C, C, 3)
        xa = torch.relu(x.float() * sc + sh)
        xa = (xa.bfloat16() if mode == 2 else xa).double().permute(0, 3, 1, 2).contiguous()
        pref = torch.nn.grad.conv2d_weight(xa, wshape, gn, padding=pad, dilation=dil).view(C, C, 3)
        bref = gn.sum((0, 2, 3))
        e = (_rel(first[0], wref), _rel(first[1], bref), _rel(first[2], pref), _rel(first[3], bref))
        print("16-channel wgrad %r mode %d: gw %.1e gb %.1e | prologue gw %.1e gb %.1e" % ((shape, mode) + e))
        assert e[0] < 3e-6 and e[1] < 3e-6 and e[3] < 3e-6 and e[2] < (3e-3 if mode == 2 else 3e-6)
        for it in range(10 if N * H * W > 100000 else 30):
            assert all(torch.equal(u, v) for u, v in zip(run(), first)), "launch %d differs from the first" % it
    finally:
        lib.lf_debug_set_ops_precision(0)
