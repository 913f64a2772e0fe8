#!/usr/bin/env python
"""bf16-tensor tap-GEMM: the LDS-staged kernels (round 4: persistent ring, whole lines) against the streaming one on the same
launches -- bit-identical results (same K order: the same MFMA sequence per accumulator) and HIP-event timing, forward (+ReLU) and
data gradient (+mask).

    python tools/bf16_ab.py [--iters 200]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanedetection_end2end_amd import _lib  # noqa: E402


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    lib = _lib.load()
    st = _lib.stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    # (N, C, H, W, axis, dil): the headline geometry at batch 32 and config 3's (320 x 640, batch 64), a ragged one
    shapes = [(32, 128, 32, 64, 0, 4), (32, 128, 32, 64, 1, 16), (32, 64, 64, 128, 0, 1), (32, 64, 64, 128, 1, 1),
              (64, 128, 40, 80, 1, 8), (64, 128, 40, 80, 0, 4), (64, 64, 80, 160, 0, 1), (64, 64, 80, 160, 1, 1), (3, 64, 6, 20, 1, 2),
              (3, 128, 5, 48, 0, 2), (2, 64, 3, 16, 1, 1)]
    lib.lf_debug_set_ops_precision(2)
    try:
        for N, C, H, W, axis, d in shapes:
            torch.manual_seed(C + axis)
            x = torch.randn(N, H, W, C, device="cuda").bfloat16()
            gy = torch.randn(N, H, W, C, device="cuda").bfloat16()
            w = torch.randn(C, C, 3, device="cuda") * (2.0 / (3 * C)) ** 0.5
            b = torch.randn(C, device="cuda")
            scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096, device="cuda")
            sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.5
            res, tim = {}, {}
            # lf_debug_set_bf16_lds: 0 streaming kernel only / 2 the ring for every launch it takes / 3 whole-line kernel at 64 AND 128 channels
            # (round 5's routing) / 4 shipped: the wave-private kernel at 64 channels, the whole-line kernel at 128
            modes = [("streaming", 0), ("ring", 2), ("whole-line", 3), ("shipped", 4)]
            for name, mode in modes:
                lib.lf_debug_set_bf16_lds(mode)
                y, gx, yp = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
                f = lambda: _lib.check(lib.lf_conv1d_fwd(P(x), P(w), P(b), P(y), N, H, W, C, axis, d, 1, P(scratch), st), "fwd")
                g = lambda: _lib.check(lib.lf_conv1d_bwd_data(P(gy), P(w), P(x), P(gx), N, H, W, C, axis, d, P(scratch), st), "dgrad")
                fp = lambda: _lib.check(lib.lf_debug_conv1d_fwd_pro(P(x), P(w), P(b), P(sc), P(sh), P(yp), N, H, W, C, axis, d, P(scratch), st), "fwd + prologue")
                tim[name] = (timeit(f, a.iters), timeit(g, a.iters), timeit(fp, a.iters))
                res[name] = (y.clone(), gx.clone(), yp.clone())
            same = all(torch.equal(res["streaming"][0], r[0]) and torch.equal(res["streaming"][1], r[1]) and torch.equal(res["streaming"][2], r[2]) for r in res.values())
            nbytes = 2 * N * H * W * C * 2
            best = min(t[0] for t in tim.values())
            print("N=%2d C=%3d %3dx%3d axis %d dil %2d | %s | bit-identical %s | best fwd %.2f TB/s algorithmic (launch + pack included)"
                  % (N, C, H, W, axis, d, " | ".join("%s fwd %6.1f dgrad %6.1f pro %6.1f us" % (k, v[0], v[1], v[2]) for k, v in tim.items()), same,
                     nbytes / best / 1e6), flush=True)
            for r in res.values():
                assert torch.isfinite(r[0].float()).all()
    finally:
        lib.lf_debug_set_ops_precision(0)
        lib.lf_debug_set_bf16_lds(4)


if __name__ == "__main__":
    main()
