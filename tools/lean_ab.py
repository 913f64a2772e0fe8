#!/usr/bin/env python
"""A/B of the fp32 16 -> 16 channel convolution kernels (csrc/lf_conv.hip): tapgemm_lean_kernel (one 256-pixel tile per workgroup, mode 0)
against the persistent tapgemm_lean_p_kernel in its three forms (lf_debug_set_lean_p 1 / 2 / 3), every epilogue flag set the network
launches, at the headline's 16-channel stage (32 x 128 x 256 pixels) by default.  HIP events around a loop that rotates over buffer
sets larger than the Infinity Cache; every mode's outputs are compared with mode 0's bit for bit.

    python tools/lean_ab.py [N H W]
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanedetection_end2end_amd import _lib  # noqa: E402


def main():
    N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 128, 256)
    C, SETS, ITERS = 16, 3, 24
    cfg = {"axis": 0}
    lib = _lib.load()
    st = _lib.stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    torch.manual_seed(0)
    mk = lambda: [torch.randn(N, H, W, C, device="cuda") for _ in range(SETS)]
    src, mask, add, aux, dst = mk(), mk(), mk(), mk(), mk()
    w = torch.randn(C, C, 3, device="cuda") * (2.0 / (3 * C)) ** 0.5
    b, sc, sh = torch.randn(C, device="cuda"), torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.5
    scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096, device="cuda")
    stats = torch.empty((N * H * W + 255) // 256, 2, C, device="cuda")
    # (name, tensors moved per pixel [x 64 B], launch(i))
    epi = lambda i, tr, flags, bias, m, a, x, s1, s2, stt: lib.lf_debug_conv1d_epi(
        P(src[i]), P(w), bias, P(dst[i]), tr, flags, m, a, x, s1, s2, stt, N, H, W, C, cfg["axis"], 1, P(scratch), st)
    cases = [
        ("fwd + relu            <0,1>", 2, lambda i: epi(i, 0, 1, P(b), None, None, None, None, None, None)),
        ("fwd + BN sums         <0,8>", 2, lambda i: epi(i, 0, 8, P(b), None, None, None, None, None, P(stats))),
        ("bn-relu pro + relu    <1,1>", 2, lambda i: lib.lf_debug_conv1d_fwd_pro(P(src[i]), P(w), P(b), P(sc), P(sh), P(dst[i]), N, H, W, C, cfg["axis"], 1, P(scratch), st)),
        ("dgrad * mask          <0,2>", 3, lambda i: epi(i, 1, 2, None, P(mask[i]), None, None, None, None, None)),
        ("dgrad + add           <0,4>", 3, lambda i: epi(i, 1, 4, None, None, P(add[i]), None, None, None, None)),
        ("dgrad*bn-mask + sums <0,48>", 3, lambda i: epi(i, 1, 48, None, None, None, P(aux[i]), P(sc), P(sh), P(stats))),
        ("dgrad+add*mask+sums  <0,38>", 5, lambda i: epi(i, 1, 38, None, P(mask[i]), P(add[i]), P(aux[i]), None, None, P(stats))),
    ]
    print("16 -> 16 channels, %d x %d x %d pixels (%.0f MB per tensor), axis / mode: us per launch (GB/s algorithmic)" % (N, H, W, N * H * W * 64 / 1e6))
    for axis in (0, 1):
        cfg["axis"] = axis
        for name, ntens, fn in cases:
            line, ref = "axis %d %s |" % (axis, name), None
            for mode in (0, 1, 2, 3):
                lib.lf_debug_set_lean_p(mode)
                for i in range(SETS):
                    assert fn(i) >= 0, lib.lf_last_error().decode()
                torch.cuda.synchronize()
                out = [d.clone() for d in dst]
                if ref is None:
                    ref = out
                same = all(torch.equal(u, v) for u, v in zip(out, ref))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for it in range(ITERS):
                    fn(it % SETS)
                e1.record()
                torch.cuda.synchronize()
                us = 1e3 * e0.elapsed_time(e1) / ITERS
                line += " m%d %6.1f (%4.0f)%s" % (mode, us, ntens * N * H * W * 64 / us / 1e3, "" if same else " DIFF")
            print(line, flush=True)
    lib.lf_debug_set_lean_p(1)


if __name__ == "__main__":
    main()
