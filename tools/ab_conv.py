#!/usr/bin/env python
"""A/B timing of the kernel-level conv calls across several builds of liblanefit_hip.so (tuning aid, GPU box only).

    python tools/ab_conv.py tools/ab/liblanefit_r2.so lanedetection_end2end_amd/liblanefit_hip.so ...

For every library: lf_conv1d_fwd (plain, +ReLU), lf_conv1d_bwd_data (plain, +mask), lf_conv1d_bwd_weight at the
network's 64- and 128-channel shapes (batch 32), HIP-event timed over 300 launches each (the weight-pack launch of the
call is included in all of them).  Libraries are dlopen'ed side by side (the C ABI has no global state that matters).
"""
import ctypes
import sys

import torch

SHAPES = [(64, 64, 128, 0, 1), (64, 64, 128, 1, 1), (128, 32, 64, 0, 4), (128, 32, 64, 1, 16)]


def timeit(f, iters=300):
    for _ in range(20 if iters > 50 else 2):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    global SHAPES
    args = sys.argv[1:]
    iters = 300
    if "--c64" in args:          # PMC runs: the 64-channel shapes only, few launches
        args.remove("--c64")
        SHAPES = SHAPES[:2]
        iters = 8
    libs = [(p, ctypes.CDLL(p)) for p in args]
    for _, lib in libs:
        lib.lf_conv1d_scratch_floats.restype = ctypes.c_long
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    N = 32
    torch.manual_seed(0)
    for C, H, W, axis, d in SHAPES:
        x = torch.randn(N, H, W, C, device="cuda")
        gy = torch.randn(N, H, W, C, device="cuda")
        w = torch.randn(C, C, 3, device="cuda") * 0.05
        b = torch.randn(C, device="cuda")
        y, gx = torch.empty_like(x), torch.empty_like(x)
        gw, gb = torch.empty_like(w), torch.empty_like(b)
        ref = None
        for name, lib in libs:
            scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C), device="cuda")
            r = []
            for relu in (0, 1):
                r.append(timeit(lambda: lib.lf_conv1d_fwd(P(x), P(w), P(b), P(y), N, H, W, C, axis, d, relu, P(scratch), st), iters))
            yy = y.clone()
            r.append(timeit(lambda: lib.lf_conv1d_bwd_data(P(gy), P(w), None, P(gx), N, H, W, C, axis, d, P(scratch), st), iters))
            r.append(timeit(lambda: lib.lf_conv1d_bwd_data(P(gy), P(w), P(x), P(gx), N, H, W, C, axis, d, P(scratch), st), iters))
            r.append(timeit(lambda: lib.lf_conv1d_bwd_weight(P(x), P(gy), P(gw), P(gb), N, H, W, C, axis, d, P(scratch), st), iters))
            chk = (float(yy.double().sum()), float(gx.double().sum()), float(gw.double().sum()))
            if ref is None:
                ref = chk
            ok = all(abs(a - b_) <= 1e-6 * max(1.0, abs(b_)) for a, b_ in zip(chk, ref))
            print("C=%3d %3dx%3d axis %d dil %2d | fwd %6.1f  fwd+relu %6.1f | dgrad %6.1f  dgrad+mask %6.1f | wgrad %6.1f us | same=%s | %s"
                  % (C, H, W, axis, d, r[0], r[1], r[2], r[3], r[4], ok, name), flush=True)


if __name__ == "__main__":
    main()
