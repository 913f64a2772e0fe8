"""Race screen of the LDS-DMA bf16 kernels (tapgemm_bf16_wl_kernel, tapgemm_bf16_ring_kernel and -- round 6 -- the wave-private
64-channel tapgemm_bf16_wv_kernel, whose per-wave rings have no barrier at all): their rings are ordered by
hand-counted s_waitcnt vmcnt(N) + bare s_barrier, so a wrong count shows as rare wrong tiles that come and go with shape and
memory load, not as a failing refcheck.  Every launch of a repeat loop must equal the streaming kernel's result BIT FOR BIT (same
K order: the same MFMA sequence per accumulator), forward (+bias, ReLU) and data gradient (+ReLU mask of a staged tensor), at the
shapes that failed during bring-up: row width 80 (16-pixel groups straddle rows), dilation 8 (a padding tap empties a whole DMA
instruction: it returns without a memory round trip and overtook pending fragment reads before the lgkmcnt(0) in front of the
barrier -- 1 launch in 3 wrong), two workgroups per CU with one or two work items each."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(64, 128, 40, 80, 1, 8), (64, 128, 40, 80, 0, 4), (64, 64, 80, 160, 1, 1), (64, 64, 80, 160, 0, 1), (32, 128, 32, 64, 1, 16),
                                   (3, 128, 20, 48, 1, 16), (5, 64, 12, 32, 0, 2), (3, 64, 20, 48, 1, 2), (2, 64, 8, 16, 0, 1), (7, 64, 10, 80, 1, 1)])
def test_lds_dma_kernels_equal_the_streaming_kernel_on_every_launch(shape):
    from lanedetection_end2end_amd import _lib
    lib = _lib.load()
    st = _lib.stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    N, C, H, W, axis, d = shape
    torch.manual_seed(1)
    x = torch.randn(N, H, W, C, device="cuda").bfloat16()
    gy = torch.randn(N, H, W, C, device="cuda").bfloat16()
    w = torch.randn(C, C, 3, device="cuda") * (2.0 / (3 * C)) ** 0.5
    b = torch.randn(C, device="cuda")
    scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096, device="cuda")

    sc = torch.rand(C, device="cuda") + 0.5
    sh = torch.randn(C, device="cuda") * 0.5          # relu(0 * sc + sh) != 0: padding must stay zero AFTER the transform

    def run(mode):
        lib.lf_debug_set_bf16_lds(mode)
        y, gx, yp = (torch.full_like(x, float("nan")) for _ in range(3))
        _lib.check(lib.lf_conv1d_fwd(P(x), P(w), P(b), P(y), N, H, W, C, axis, d, 1, P(scratch), st), "fwd")
        _lib.check(lib.lf_conv1d_bwd_data(P(gy), P(w), P(x), P(gx), N, H, W, C, axis, d, P(scratch), st), "dgrad")
        # the BN+ReLU operand prologue form (the whole-line kernel transforms the staged pixels in LDS)
        _lib.check(lib.lf_debug_conv1d_fwd_pro(P(x), P(w), P(b), P(sc), P(sh), P(yp), N, H, W, C, axis, d, P(scratch), st), "fwd with prologue")
        torch.cuda.synchronize()
        return y, gx, yp

    try:
        lib.lf_debug_set_ops_precision(2)
        ref = run(0)
        assert all(torch.isfinite(r.float()).all() for r in ref)
        # the prologue reference itself against torch (fp32 math on the bf16 values, bf16-rounded operand, fp32 accumulation)
        import torch.nn.functional as F
        xa = torch.relu(x.float() * sc + sh).bfloat16().double().permute(0, 3, 1, 2)
        w4 = (w.view(C, C, 3, 1) if axis == 0 else w.view(C, C, 1, 3)).bfloat16().double()
        pad, dil = ((d, 0), (d, 1)) if axis == 0 else ((0, d), (1, d))
        want = torch.relu(F.conv2d(xa, w4, b.double(), padding=pad, dilation=dil)).permute(0, 2, 3, 1)
        err = (ref[2].double() - want)
        # (the kernel rounds relu(fma(x, sc, sh)) to bf16, torch a two-rounding multiply-add: a handful of operands land on the other side
        # of a bf16 rounding boundary, so the comparison is in norm, not per element)
        assert float(err.norm() / want.norm()) < 3e-3 and float(err.abs().max()) < 0.03 * float(want.abs().max()), (float(err.norm() / want.norm()), float(err.abs().max()))
        repeats = 12 if N * H * W > 100000 else 40
        # (mode 3 = the whole-line kernel at both channel counts, 4 = shipped: wave-private kernel at 64 channels, whole-line at 128)
        for name, mode in (("ring", 2), ("whole-line", 3), ("wave-private (64 ch) / whole-line (128 ch)", 4)):
            bad = []
            for it in range(repeats):
                y, gx, yp = run(mode)
                if not (torch.equal(y, ref[0]) and torch.equal(gx, ref[1]) and torch.equal(yp, ref[2])):
                    px = ((y != ref[0]) | (gx != ref[1]) | (yp != ref[2])).reshape(-1, C).any(1).nonzero().flatten()
                    bad.append((it, len(px), int(px[0])))
            assert not bad, "%s kernel, shape %r: launches that differ from the streaming kernel (launch, pixels, first pixel): %r" % (name, shape, bad[:6])
    finally:
        lib.lf_debug_set_ops_precision(0)
        lib.lf_debug_set_bf16_lds(4)


@pytest.mark.parametrize("shape", [(64, 64, 80, 160, 1, 1), (64, 64, 80, 160, 0, 1), (16, 128, 40, 80, 0, 8), (3, 64, 20, 48, 1, 2), (5, 64, 12, 32, 0, 1)])
def test_three_tensor_epilogue_whole_line_kernel(shape):
    """The data gradient that closes a non_bottleneck_1d block's backward (ADD + MASK + BN-backward sums: three epilogue tensors staged
    by LDS-DMA in the whole-line kernel, tapgemm_bf16_wl_kernel<2, 38, 0> at 64 channels) at config 3's own shape 64 x 64 x 80 x 160:
    the stored values equal to the streaming kernel's bit for bit on every launch of a repeat loop, the partial sums reproduced
    exactly from launch to launch, and both correct against
    torch: gx = round_bf16((conv^T(gy) + add) * [mask > 0]) within half a bf16 ulp, column sums of gx and gx * aux to fp32 accuracy."""
    import torch.nn.functional as F
    from lanedetection_end2end_amd import _lib
    lib = _lib.load()
    st = _lib.stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    N, C, H, W, axis, d = shape
    torch.manual_seed(3)
    gy = torch.randn(N, H, W, C, device="cuda").bfloat16()
    mask = torch.randn(N, H, W, C, device="cuda").bfloat16()
    add = torch.randn(N, H, W, C, device="cuda").bfloat16()
    aux = torch.randn(N, H, W, C, device="cuda").bfloat16()
    w = torch.randn(C, C, 3, device="cuda") * (2.0 / (3 * C)) ** 0.5
    scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096, device="cuda")
    nrows_max = (N * H * W + 255) // 256

    def run(mode):
        lib.lf_debug_set_bf16_lds(mode)
        gx = torch.full_like(gy, float("nan"))
        stats = torch.full((2, C, nrows_max), float("nan"), device="cuda")      # channel-major partial rows: [kind][channel][row]
        rows = lib.lf_debug_conv1d_bwd_data_epi3(P(gy), P(w), P(mask), P(add), P(aux), P(gx), P(stats), N, H, W, C, axis, d, P(scratch), st)
        assert rows == nrows_max, lib.lf_last_error().decode()
        torch.cuda.synchronize()
        return gx, stats.clone()

    try:
        lib.lf_debug_set_ops_precision(2)
        ref = run(0)
        assert torch.isfinite(ref[0].float()).all() and torch.isfinite(ref[1]).all()
        w4 = (w.view(C, C, 3, 1) if axis == 0 else w.view(C, C, 1, 3)).bfloat16().double()
        pad, dil = ((d, 0), (d, 1)) if axis == 0 else ((0, d), (1, d))
        gn = gy.double().permute(0, 3, 1, 2).contiguous()
        want = (torch.nn.grad.conv2d_input(gn.shape, w4, gn, padding=pad, dilation=dil).permute(0, 2, 3, 1) + add.double()) * (mask.double() > 0)
        err = (ref[0].double() - want).abs()
        assert (err <= want.abs() * 2.0 ** -8 + 1e-6).all(), float(err.max())
        s1 = ref[0].double().sum((0, 1, 2))
        s2 = (ref[0].double() * aux.double()).sum((0, 1, 2))
        got = ref[1].double().sum(2)
        assert float((got[0] - s1).abs().max()) < 2e-5 * float(ref[0].double().abs().sum((0, 1, 2)).max())
        assert float((got[1] - s2).abs().max()) < 2e-5 * float((ref[0].double() * aux.double()).abs().sum((0, 1, 2)).max())
        # (the partial SUMS of the two kernels are taken in different orders -- fp32, per tile -- so they agree to rounding, not in bits;
        # the stored values do, and every launch of the whole-line kernel must reproduce its own sums exactly)
        for mode in (3, 4):          # 3: whole-line kernel at 64 channels too; 4 (shipped): the wave-private kernel at 64 channels
            first = None
            for it in range(10 if N * H * W > 100000 else 30):
                gx, stats = run(mode)
                assert torch.equal(gx, ref[0]), "LDS kernel (mode %d), launch %d: values differ from the streaming kernel" % (mode, it)
                first = stats if first is None else first
                assert torch.equal(stats, first), "LDS kernel (mode %d), launch %d: partial sums differ from its first launch" % (mode, it)
            gotw = first.double().sum(2)
            assert float((gotw[0] - s1).abs().max()) < 2e-5 * float(ref[0].double().abs().sum((0, 1, 2)).max())
            assert float((gotw[1] - s2).abs().max()) < 2e-5 * float((ref[0].double() * aux.double()).abs().sum((0, 1, 2)).max())
    finally:
        lib.lf_debug_set_ops_precision(0)
        lib.lf_debug_set_bf16_lds(4)
