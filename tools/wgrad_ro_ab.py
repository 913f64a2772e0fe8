#!/usr/bin/env python
"""bf16 weight gradient of the 3-tap 64- / 128-channel convolutions: the read-once kernel (lf_wgrad_ro.hip: one workgroup owns all
taps of a 64-channel x-block against every g-channel; X and G staged once per pixel range by whole-line LDS-DMA) against
tapwgrad_kernel's job form (3 / 12 jobs re-streaming their halves of X and G) on the same launches: HIP-event timing of
lf_conv1d_bwd_weight incl. the split-K reduction, relative difference of the results, with and without the BN+ReLU operand prologue.

    python tools/wgrad_ro_ab.py [--iters 100] [--caps 512,256 256,256 1024,512]
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
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--caps", nargs="*", default=["512,256"], help="workgroups per launch at 64,128 channels (<= the shipped 512,256)")
    a = ap.parse_args()
    lib = _lib.load()
    st = _lib.stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    shapes = [(64, 128, 40, 80, 0, 8), (64, 128, 40, 80, 1, 8), (64, 64, 80, 160, 0, 1), (64, 64, 80, 160, 1, 1),
              (32, 128, 32, 64, 0, 4), (32, 128, 32, 64, 1, 16), (32, 64, 64, 128, 0, 1), (32, 64, 64, 128, 1, 1)]
    lib.lf_debug_set_ops_precision(2)
    try:
        for N, C, H, W, axis, d in shapes:
            torch.manual_seed(C + axis)
            x = torch.randn(N, H, W, C, device="cuda").bfloat16()
            gy = torch.randn(N, H, W, C, device="cuda").bfloat16()
            sc = torch.rand(C, device="cuda") + 0.5
            sh = torch.randn(C, device="cuda") * 0.5
            scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096, device="cuda")
            out = {}
            for name, mode, caps in [("job form", 0, (0, 0))] + [("read-once %s" % c, 1, tuple(int(v) for v in c.split(","))) for c in a.caps]:
                lib.lf_debug_set_wgrad_ro(mode, caps[0], caps[1])
                gw, gb, gwp = torch.empty(C, C, 3, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, C, 3, device="cuda")
                f = lambda: _lib.check(lib.lf_conv1d_bwd_weight(P(x), P(gy), P(gw), P(gb), N, H, W, C, axis, d, P(scratch), st), "wgrad")
                h = lambda: _lib.check(lib.lf_debug_conv1d_wgrad_pro(P(x), P(gy), P(sc), P(sh), P(gwp), None, N, H, W, C, axis, d, P(scratch), st), "wgrad pro")
                out[name] = (timeit(f, a.iters), timeit(h, a.iters), gw.clone(), gb.clone(), gwp.clone())
            ref = out["job form"]
            rel = lambda u, v: float((u - v).norm() / v.norm())
            nbytes = 2 * N * H * W * C * 2
            print("N=%2d C=%3d %3dx%3d axis %d dil %2d | " % (N, C, H, W, axis, d) +
                  " | ".join("%s %6.1f us, prologue %6.1f us (%.2f TB/s algorithmic)%s" %
                             (k, v[0], v[1], nbytes / v[0] / 1e6,
                              "" if k == "job form" else " diff gw %.1e gb %.1e pro %.1e" % (rel(v[2], ref[2]), rel(v[3], ref[3]), rel(v[4], ref[4])))
                             for k, v in out.items()), flush=True)
    finally:
        lib.lf_debug_set_ops_precision(0)
        lib.lf_debug_set_wgrad_ro(1, 512, 256)


if __name__ == "__main__":
    main()
