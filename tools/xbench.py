#!/usr/bin/env python
"""Micro-benchmarks of the SURVEY 8f rows (HIP-event timed on torch's current stream, which is the stream the
C-ABI calls are launched on):

    python tools/xbench.py pipeline [--batch 32] [--resize 256]   # input pipeline, GB/s vs the 8 TB/s HBM roof
    python tools/xbench.py clas [--batch 32]                      # --clas heads fwd+bwd, TFLOP/s of the conv trunk
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def bench_pipeline(args):
    from lanedetection_end2end_amd.pipeline import InputPipeline
    N, R = args.batch, args.resize
    fr = torch.randint(0, 256, (N, 720, 1280, 3), dtype=torch.uint8, device="cuda")
    lb = torch.randint(0, 5, (N, 720, 1280), dtype=torch.uint8, device="cuda")
    flip = torch.rand(N) > 0.5
    for tree in ("bev", "bp"):
        pipe = InputPipeline(R, tree=tree, nclasses=2 if tree == "bev" else 4)
        pipe(fr, lb, flip)
        t_img = timeit(lambda: pipe(fr, None, flip))
        t_all = timeit(lambda: pipe(fr, lb, flip))
        by_img = N * (640 * 1280 * 3 + 3 * R * 2 * R * 4)            # cropped uint8 rows in, fp32 NCHW out
        by_lab = N * (R * 2 * R * (1 + 8))                            # gathered label bytes in, int64 out
        print("pipeline %s  batch %d  R %d: image %.3f ms = %.0f GB/s algorithmic (%.1f %% of 8 TB/s); image+label %.3f ms"
              " = %.0f GB/s; %.0f frames/s" % (tree, N, R, t_img, by_img / t_img / 1e6, by_img / t_img / 1e6 / 80,
                                              t_all, (by_img + by_lab) / t_all / 1e6, N / t_all * 1e3))


def bench_clas(args):
    from lanedetection_end2end_amd.clas import Classification
    N = args.batch
    x = torch.relu(torch.randn(N, 32, 64, 128, device="cuda")).permute(0, 3, 1, 2).requires_grad_(True)
    flops_fwd = 2.0 * N * 32 * 64 * (128 * 128 + 9 * 128 * 128 + 9 * 128 * 64 + 9 * 64 * 64)
    for ct in ("line", "horizon"):
        m = Classification(ct, size=(32, 64), channels_in=128, resize=256).cuda().train()

        def step():
            m.zero_grad(set_to_none=True)
            x.grad = None
            m(x).sum().backward()
        t = timeit(step, iters=10)
        with torch.no_grad():
            tf = timeit(lambda: m(x), iters=10)
        print("clas %-7s batch %d: fwd %.3f ms, fwd+bwd %.3f ms = %.1f TFLOP/s on the conv trunk (3x fwd FLOPs %.1f GF)"
              % (ct, N, tf, t, 3 * flops_fwd / t / 1e9, 3 * flops_fwd / 1e9))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["pipeline", "clas"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--resize", type=int, default=256)
    a = ap.parse_args()
    (bench_pipeline if a.what == "pipeline" else bench_clas)(a)
