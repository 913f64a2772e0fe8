#!/usr/bin/env python
"""Kernel micro-benchmark: one factorised conv (fwd / dgrad / wgrad) of each ERFNet stage at batch 32,
HIP-event timed, as TFLOP/s against the fp32 MFMA peak.  Also checks each result against torch (on the
GPU, fp32) so that a faster variant that is wrong is caught immediately.

    python tools/kbench.py [--iters 300] [--split 9] [--one C H W AXIS DIL]
    python tools/kbench.py --phases [--wgrad]      per-wave phase stamps, waves paired up by the SIMD they ran on
"""
import argparse
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanedetection_end2end_amd import _lib  # noqa: E402

PEAK = 157.3


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--split", type=int, default=0, choices=[0, 9], help="fp32 from 3-way split operands on the bf16 matrix "
                    "cores (all 9 partial products); checked against the plain fp32 torch result")
    ap.add_argument("--one", type=int, nargs=5, metavar=("C", "H", "W", "AXIS", "DIL"), help="run a single shape (for PMC passes)")
    ap.add_argument("--miopen", action="store_true", help="also time the vendor library on the same problems (torch conv2d forward / "
                    "input gradient / weight gradient through MIOpen, NCHW fp32 as the reference runs it) and print its us")
    a = ap.parse_args()
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    shapes = [(128, 32, 64, 0, 4), (128, 32, 64, 1, 16), (64, 64, 128, 0, 1), (64, 64, 128, 1, 1), (16, 128, 256, 1, 1)]
    N = a.batch
    if a.one:
        shapes = [tuple(a.one)]
    for C, H, W, axis, d in shapes:
        torch.manual_seed(0)
        x = torch.randn(N, H, W, C, device="cuda")
        gy = torch.randn(N, H, W, C, device="cuda")
        w = torch.randn(C, C, 3, device="cuda") * (2.0 / (3 * C)) ** 0.5
        b = torch.randn(C, device="cuda")
        y, gx = torch.empty_like(x), torch.empty_like(x)
        gw, gb = torch.empty_like(w), torch.empty_like(b)
        scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C), device="cuda")
        flops = 2.0 * N * H * W * C * C * 3
        w4 = w.view(C, C, 3, 1) if axis == 0 else w.view(C, C, 1, 3)
        pad, dil = ((d, 0), (d, 1)) if axis == 0 else ((0, d), (1, d))
        xn = x.permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        wn = w4.clone().requires_grad_(True)
        bn = b.clone().requires_grad_(True)
        yr = F.conv2d(xn, wn, bn, padding=pad, dilation=dil)
        yr.backward(gy.permute(0, 3, 1, 2))
        gx_ref = xn.grad
        with torch.no_grad():     # fp64 truth: how far each arithmetic (fp32 cores / split operands) is from exact
            y64 = F.conv2d(xn.detach().double(), w4.double(), b.double(), padding=pad, dilation=dil)
            gx64 = torch.nn.grad.conv2d_input(xn.shape, w4.double(), gy.permute(0, 3, 1, 2).double().contiguous(), padding=pad,
                                              dilation=dil)
            gw64 = torch.nn.grad.conv2d_weight(xn.detach().double(), w4.shape, gy.permute(0, 3, 1, 2).double().contiguous(),
                                               padding=pad, dilation=dil)
        lib.lf_debug_set_ops_precision(a.split)
        for v in (2,):
            f = lambda: _lib.check(lib.lf_conv1d_fwd(P(x), P(w), P(b), P(y), N, H, W, C, axis, d, 0, P(scratch), st), "fwd")
            tf = timeit(f, a.iters)
            e1 = float((y.permute(0, 3, 1, 2) - yr).abs().max() / yr.abs().max())
            d1 = float((y.permute(0, 3, 1, 2).double() - y64).abs().max() / y64.abs().max())
            r1 = float((y.permute(0, 3, 1, 2).double() - y64).pow(2).mean().sqrt() / y64.pow(2).mean().sqrt())
            g = lambda: _lib.check(lib.lf_conv1d_bwd_data(P(gy), P(w), None, P(gx), N, H, W, C, axis, d, P(scratch), st), "dgrad")
            td = timeit(g, a.iters)
            e2 = float((gx.permute(0, 3, 1, 2) - gx_ref).abs().max() / gx_ref.abs().max())
            d2 = float((gx.permute(0, 3, 1, 2).double() - gx64).abs().max() / gx64.abs().max())
            h = lambda: _lib.check(lib.lf_conv1d_bwd_weight(P(x), P(gy), P(gw), P(gb), N, H, W, C, axis, d, P(scratch), st), "wgrad")
            tw = timeit(h, a.iters)
            e3 = float((gw.view_as(wn.grad) - wn.grad).abs().max() / wn.grad.abs().max())
            e4 = float((gb - bn.grad).abs().max() / bn.grad.abs().max())
            d3 = float((gw.view_as(gw64).double() - gw64).abs().max() / gw64.abs().max())
            if a.miopen:      # the vendor library on the same problem (what the reference's nn.Conv2d calls on this GPU)
                gyn = gy.permute(0, 3, 1, 2).contiguous()
                xd, wd = xn.detach(), w4.detach()
                mf = timeit(lambda: F.conv2d(xd, wd, b, padding=pad, dilation=dil), a.iters)
                md = timeit(lambda: torch.nn.grad.conv2d_input(xd.shape, wd, gyn, padding=pad, dilation=dil), a.iters)
                mw = timeit(lambda: torch.nn.grad.conv2d_weight(xd, wd.shape, gyn, padding=pad, dilation=dil), a.iters)
                print("   MIOpen (torch %s, NCHW fp32): fwd %6.1f us %5.1f TF | dgrad %6.1f us %5.1f TF | wgrad %6.1f us %5.1f TF   "
                      "-> HIP / MIOpen speed: fwd %.2fx dgrad %.2fx wgrad %.2fx"
                      % (torch.__version__, mf * 1e6, flops / mf / 1e12, md * 1e6, flops / md / 1e12, mw * 1e6, flops / mw / 1e12,
                         mf / tf, md / td, mw / tw), flush=True)
            print("   vs fp64: fwd max %.2e rms %.2e | dgrad max %.2e | wgrad max %.2e   (torch fp32: fwd %.2e)"
                  % (d1, r1, d2, d3, float((yr.double() - y64).abs().max() / y64.abs().max())), flush=True)
            print("C=%3d %3dx%3d axis %d dil %2d var %d | fwd %6.1f us %5.1f TF (%4.1f%%) err %.1e | dgrad %6.1f us %5.1f TF err %.1e | "
                  "wgrad(+reduce) %6.1f us %5.1f TF err %.1e %.1e"
                  % (C, H, W, axis, d, v, tf * 1e6, flops / tf / 1e12, 100 * flops / tf / 1e12 / PEAK, e1, td * 1e6,
                     flops / td / 1e12, e2, tw * 1e6, flops / tw / 1e12, e3, e4), flush=True)


if __name__ == "__main__" and "--phases" not in sys.argv and "--phases16" not in sys.argv:
    main()


def phases(batch=32, split=0, variant=2, ablate=0, shapes=((128, 32, 64, 1, 16), (64, 64, 128, 1, 1))):
    """Per-wave phase breakdown of the forward tap-GEMM (s_memrealtime stamps, 100 MHz): first operands resident /
    main loop / epilogue + store drain, in microseconds from the first wave's start; launch time by HIP events."""
    import numpy as np
    lib = _lib.load()
    lib.lf_debug_set_ops_precision(split)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for C, H, W, axis, d in shapes:
        N = batch
        x = torch.randn(N, H, W, C, device="cuda")
        w = torch.randn(C, C, 3, device="cuda") * 0.05
        b = torch.randn(C, device="cuda")
        y = torch.empty_like(x)
        scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C), device="cuda")
        nw = ((N * H * W + 255) // 256) * 4 * (C // 64)
        dbg = torch.zeros(max(nw, 2048) * 8 + 4 * 8 * 64, dtype=torch.int64, device="cuda")
        f = lambda: _lib.check(lib.lf_debug_conv1d_fwd_phases(P(x), P(w), P(b), P(y), N, H, W, C, axis, d, P(scratch), P(dbg), st), "phases")
        us = timeit(f, 300) * 1e6          # long enough for the clocks to settle (a cold 1 ms burst runs ~8 % slower)
        dbg.zero_()
        f()
        torch.cuda.synchronize()
        full = dbg.cpu().numpy().astype(np.float64)
        t = full[: nw * 8].reshape(nw, 8)
        t = t[t[:, 0] > 0] * 0.01                      # waves that ran; ticks -> microseconds
        t0 = t[:, 0].min()
        print("fwd C=%3d N=%3d waves %5d | launch+pack %6.1f us | start spread %5.2f | operands ready %5.2f (max %5.2f) | "
              "main loop done %6.2f (min %6.2f max %6.2f) | stores retired %6.2f (max %6.2f)"
              % (C, N, len(t), us, (t[:, 0] - t0).max(), (t[:, 1] - t0).mean(), (t[:, 1] - t0).max(),
                 (t[:, 2] - t0).mean(), (t[:, 2] - t0).min(), (t[:, 2] - t0).max(), (t[:, 3] - t0).mean(), (t[:, 3] - t0).max()),
              flush=True)
        hw = full[: nw * 8].reshape(nw, 8)[:, 4].astype(np.int64)
        _simd_report(t, hw[full[: nw * 8].reshape(nw, 8)[:, 0] > 0], t0)       # pair the waves up by the SIMD they ran on


def _simd_report(t, hw, t0):
    import numpy as np
    key = ((hw >> 32) & 15) * 65536 + ((hw >> 4) & 3) + ((hw >> 8) & 15) * 4 + ((hw >> 12) & 0xf) * 64
    simds = {}
    for i, k in enumerate(key):
        simds.setdefault(int(k), []).append(t[i] - t0)
    cnt = np.bincount([len(v) for v in simds.values()])
    print("     SIMDs used %d; waves per SIMD histogram %s" % (len(simds), dict((i, int(c)) for i, c in enumerate(cnt) if c)))
    ends = np.array([max(w[3] for w in v) for v in simds.values()])
    loops = np.array([max(w[2] for w in v) for v in simds.values()])
    first = np.array([min(w[2] for w in v) for v in simds.values()])
    print("     per SIMD: first wave's loop done %.2f (%.2f..%.2f) | last wave's loop done %.2f (%.2f..%.2f) | last store %.2f (%.2f..%.2f)"
          % (first.mean(), first.min(), first.max(), loops.mean(), loops.min(), loops.max(), ends.mean(), ends.min(), ends.max()))
    for k in sorted(simds)[:3] + sorted(simds)[len(simds) // 2: len(simds) // 2 + 3]:
        print("     simd %06x: %s" % (k, "  ".join("[start %.2f ready %.2f loop %.2f end %.2f]" % tuple(w[:4]) for w in sorted(simds[k], key=lambda w: w[0]))))


def wgrad_phases(batch=32, shapes=((128, 32, 64, 1, 16), (64, 64, 128, 1, 1))):
    """Per-wave phases of the weight-gradient kernel (same stamps as phases())."""
    import numpy as np
    lib = _lib.load()
    lib.lf_debug_set_ops_precision(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for C, H, W, axis, d in shapes:
        N = batch
        x = torch.randn(N, H, W, C, device="cuda")
        gy = torch.randn(N, H, W, C, device="cuda")
        scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C), device="cuda")
        dbg = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
        f = lambda: lib.lf_debug_conv1d_wgrad_phases(P(x), P(gy), N, H, W, C, axis, d, P(scratch), P(dbg), st)
        us = timeit(f, 300) * 1e6
        dbg.zero_()
        nw = f()
        torch.cuda.synchronize()
        full = dbg.cpu().numpy()
        t = full[: nw * 8].reshape(nw, 8)
        ok = t[:, 0] > 0
        hw = t[ok, 4]
        t = t[ok].astype(np.float64) * 0.01
        t0 = t[:, 0].min()
        print("wgrad C=%3d N=%3d waves %5d | launch %6.1f us | start spread %5.2f | first operands %5.2f | main loop done %6.2f (min %6.2f max %6.2f) | "
              "partials stored %6.2f (max %6.2f)" % (C, N, len(t), us, (t[:, 0] - t0).max(), (t[:, 1] - t0).mean(), (t[:, 2] - t0).mean(),
                                                     (t[:, 2] - t0).min(), (t[:, 2] - t0).max(), (t[:, 3] - t0).mean(), (t[:, 3] - t0).max()), flush=True)
        _simd_report(t, hw, t0)


def phases16(shapes=((32, 128, 32, 64, 1, 16), (64, 128, 40, 80, 1, 8), (32, 64, 64, 128, 1, 1), (64, 64, 80, 160, 0, 1))):
    """Per-wave stamps of the persistent bf16 ring kernel (tapgemm_bf16_ring_kernel): when each workgroup started and, per work
    item, when its K loop was done and when its epilogue's stores had retired."""
    import numpy as np
    lib = _lib.load()
    lib.lf_debug_set_ops_precision(2)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    try:
        for N, C, H, W, axis, d in shapes:
            for nh in (1,):
                lib.lf_debug_set_bf16_lds(2)
                x = torch.randn(N, H, W, C, device="cuda").bfloat16()
                w = torch.randn(C, C, 3, device="cuda") * 0.05
                b = torch.randn(C, device="cuda")
                y = torch.empty_like(x)
                scratch = torch.empty(lib.lf_conv1d_scratch_floats(N, H, W, C) + 4096, device="cuda")
                nw = 4096 * 4
                dbg = torch.zeros(nw * 16, dtype=torch.int64, device="cuda")
                f = lambda: _lib.check(lib.lf_debug_conv1d_fwd_phases(P(x), P(w), P(b), P(y), N, H, W, C, axis, d, P(scratch), P(dbg), st), "phases16")
                us = timeit(f, 200) * 1e6
                dbg.zero_()
                f()
                torch.cuda.synchronize()
                t = dbg.cpu().numpy().reshape(nw, 16)
                t = t[t[:, 0] > 0]
                hw = t[:, 1]
                T = t[:, [0] + list(range(2, 16))].astype(np.float64) * 0.01
                t0 = T[:, 0].min()
                T = np.where(T > 0, T - t0, np.nan)
                nit = (~np.isnan(T[:, 1::2])).sum(1)
                print("bf16 ring N=%2d C=%3d %3dx%3d axis %d dil %2d, %d output channels per workgroup | launch+pack %6.1f us | waves %d | kernel span %.2f us | "
                      "items per workgroup %s" % (N, C, H, W, axis, d, 64 * nh, us, len(T), np.nanmax(T), np.bincount(nit).tolist()))
                loop = T[:, 1::2] - np.concatenate([T[:, :1], T[:, 2:-1:2]], 1)       # K loop of item k: from the previous epilogue's end (or the start)
                epi = T[:, 2::2] - T[:, 1::2]
                for k in range(min(4, int(nit.max()))):
                    print("     item %d: K loop (incl. the wait for its first operands) %.2f us (p10 %.2f p90 %.2f) | epilogue + stores retired %.2f us (p10 %.2f p90 %.2f)"
                          % (k, np.nanmean(loop[:, k]), np.nanpercentile(loop[:, k], 10), np.nanpercentile(loop[:, k], 90),
                             np.nanmean(epi[:, k]), np.nanpercentile(epi[:, k], 10), np.nanpercentile(epi[:, k], 90)))
                key = ((hw >> 32) & 15) * 65536 + ((hw >> 8) & 15) * 4 + ((hw >> 12) & 0xf) * 64          # CU
                cus = {}
                for i, k in enumerate(key):
                    cus.setdefault(int(k), []).append(T[i])
                print("     CUs used %d, waves per CU %s" % (len(cus), np.bincount([len(v) for v in cus.values()]).tolist()))
                k0 = sorted(cus)[len(cus) // 2]
                for v in sorted(cus[k0], key=lambda v: v[0])[::4]:
                    print("     CU %06x workgroup: %s" % (k0, " ".join("%.1f" % q for q in v if not np.isnan(q))))
    finally:
        lib.lf_debug_set_ops_precision(0)
        lib.lf_debug_set_bf16_lds(4)


if __name__ == "__main__" and "--phases16" in sys.argv:
    phases16()
    sys.exit(0)

if __name__ == "__main__" and "--phases" in sys.argv:
    sp = int(sys.argv[sys.argv.index("--split") + 1]) if "--split" in sys.argv else 0
    for nb in (32,):
        if "--wgrad" in sys.argv:
            wgrad_phases(nb)
            continue
        phases(nb, sp, variant=2)
