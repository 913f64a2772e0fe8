#!/usr/bin/env python
"""Split-copy plumbing check + timing (GPU): fp32 conv on the bf16 matrix cores with the operand split in registers vs
loaded from a split copy its producer wrote; the two must be bitwise identical."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanedetection_end2end_amd import _lib  # noqa: E402
from tools.kbench import P, timeit  # noqa: E402


def unsplit(x48, shape):
    t = x48.view(torch.bfloat16).view(-1, 3, 8).float()
    return ((t[:, 0] + t[:, 1]) + t[:, 2]).reshape(shape)


def main():
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    N = 32
    for terms in (9, 6):
        lib.lf_debug_set_ops_precision(terms)
        for C, H, W, axis, d in [(128, 32, 64, 1, 16), (128, 32, 64, 0, 4), (64, 64, 128, 0, 1), (64, 64, 128, 1, 1)]:
            torch.manual_seed(1)
            x = torch.randn(N, H, W, C, device="cuda") * torch.exp(torch.randn(N, H, W, C, device="cuda"))
            w = torch.randn(C, C, 3, device="cuda") * (2.0 / (3 * C)) ** 0.5
            b = torch.randn(C, device="cuda")
            ya, yb = torch.empty_like(x), torch.empty_like(x)
            x48 = torch.empty(x.numel() * 6, dtype=torch.uint8, device="cuda")
            y48 = torch.zeros(x.numel() * 6, dtype=torch.uint8, device="cuda")
            scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C), device="cuda")
            _lib.check(lib.lf_debug_split_tensor(P(x), P(x48), x.numel(), st), "split")
            exact = bool(torch.equal(unsplit(x48, x.shape), x))
            lib.lf_debug_set_ops_split_copies(None, None)
            fa = lambda: _lib.check(lib.lf_conv1d_fwd(P(x), P(w), P(b), P(ya), N, H, W, C, axis, d, 1, P(scratch), st), "fwd")
            ta = timeit(fa, 20)
            lib.lf_debug_set_ops_split_copies(P(x48), P(y48))
            fb = lambda: _lib.check(lib.lf_conv1d_fwd(P(x), P(w), P(b), P(yb), N, H, W, C, axis, d, 1, P(scratch), st), "fwd48")
            tb = timeit(fb, 20)
            lib.lf_debug_set_ops_split_copies(P(x48), None)
            tc = timeit(fb, 20)
            lib.lf_debug_set_tap_flags(1)       # all pixel loads L1-resident: is the loop bound by operand delivery?
            tcl = timeit(fb, 20)
            lib.lf_debug_set_ops_split_copies(None, None)
            tal = timeit(fa, 20)
            lib.lf_debug_set_tap_flags(0)
            same = bool(torch.equal(ya, yb))
            out_ok = bool(torch.equal(unsplit(y48, yb.shape), yb))
            flops = 2.0 * N * H * W * C * C * 3
            print("x%d C=%3d axis %d dil %2d | split exact %s | in-register %.1f us (%.0f TF) | from split copy %.1f us (%.0f TF), "
                  "+ split copy of the result %.1f us | bitwise equal %s, result copy exact %s"
                  " | L1-resident loads: in-register %.1f, from copy %.1f us"
                  % (terms, C, axis, d, exact, ta * 1e6, flops / ta / 1e12, tc * 1e6, flops / tc / 1e12, tb * 1e6, same, out_ok, tal * 1e6, tcl * 1e6), flush=True)
    lib.lf_debug_set_ops_precision(0)


if __name__ == "__main__":
    main()
