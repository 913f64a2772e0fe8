#!/usr/bin/env python
"""Is the fp32 weight gradient (tapwgrad_kernel, job form) bound by its fabric-side traffic?  

The job form re-streams X and G once per job of a pixel range: 3 jobs (taps) at 64 channels = 3 x the tensors, 12 jobs (3 taps x 2 x 2
channel blocks, each reading HALF of X and of G) at 128 channels = 6 x -- counters: 5.1 x = 344 MB per 128-channel launch at batch 32
(profiles/traffic.json).  Both launches have the SAME tensors (67 MB each for X and G) and the SAME FLOPs (6.44 G).  If the kernel
were bound by fabric requests, the 128-channel launch (twice the re-reads) would be the slower one and a launch whose operands come
from HBM instead of the Infinity Cache would be slower still.  This tool times

    warm   the same X, G every launch (67 + 67 MB: resident in the 256 MB Infinity Cache after the first launch)
    cold   launches rotating over 10 (X, G) pairs = 1.34 GB: every operand line comes from HBM

with HIP events over `--iters` launches (weight-gradient kernel + its split-K reduction, as lf_conv1d_bwd_weight runs them).

    python tools/wgrad_traffic.py [--iters 200]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanedetection_end2end_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    lib = _lib.load()
    st = _lib.stream()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    lib.lf_debug_set_ops_precision(0)
    for (N, C, H, W, axis, d, jobs, reread) in ((32, 128, 32, 64, 1, 16, 12, 6), (32, 128, 32, 64, 0, 4, 12, 6), (32, 64, 64, 128, 1, 1, 3, 3), (32, 64, 64, 128, 0, 1, 3, 3)):
        torch.manual_seed(0)
        npairs = 10
        xs = [torch.randn(N, H, W, C, device="cuda") for _ in range(npairs)]
        gs = [torch.randn(N, H, W, C, device="cuda") for _ in range(npairs)]
        gw, gb = torch.empty(C, C, 3, device="cuda"), torch.empty(C, device="cuda")
        scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096, device="cuda")
        mb = 2 * N * H * W * C * 4 / 1e6

        def run(k):
            _lib.check(lib.lf_conv1d_bwd_weight(P(xs[k]), P(gs[k]), P(gw), P(gb), N, H, W, C, axis, d, P(scratch), st), "wgrad")

        out = {}
        for name, pick in (("warm", lambda i: 0), ("cold", lambda i: i % npairs)):
            for i in range(20):
                run(pick(i))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.iters):
                run(pick(i))
            e1.record()
            torch.cuda.synchronize()
            out[name] = e0.elapsed_time(e1) / a.iters * 1e3
        flops = 2.0 * N * H * W * C * C * 3
        print("fp32 wgrad C=%3d %3dx%3d axis %d dil %2d: %2d jobs per pixel range (%d x the tensors = %4.0f MB requested of %3.0f MB) | "
              "warm %6.1f us (%5.1f TF/s, %4.2f TB/s of requests) | cold %6.1f us (%5.1f TF/s, %4.2f TB/s of requests)"
              % (C, H, W, axis, d, jobs, reread, reread * mb, mb, out["warm"], flops / out["warm"] / 1e6, reread * mb / out["warm"],
                 out["cold"], flops / out["cold"] / 1e6, reread * mb / out["cold"]), flush=True)


if __name__ == "__main__":
    main()
