"""Where does the HIP path's extra distance from fp64 come from?  (~1.25x the CPU fp32 leg's.)

CPU-only numerics study: the fp32 oracle leg is re-run with ONE ingredient at a time replaced by an emulation of what the
HIP engine does, and each variant's RMS distance of the logits from the fp64 leg is printed relative to the plain fp32 leg's:

  fold    BatchNorm applied in the folded form y = fma(x, sc, sh), sc = fl32(gamma * rstd), sh = fl32(beta - mean * sc)
          (statistics exact), instead of ((x - mean) * rstd) * gamma + beta;
  stats   BatchNorm batch statistics from fp32 partial sums (sum y, sum y^2 over 256-pixel tiles, fp64 combine,
          var = E[y^2] - mean^2) instead of torch's two-pass fp32 statistics;
  chain   every convolution as ONE sequential K-long fp32 fma chain per output element (tap outer, channel inner) instead
          of oneDNN's kernel.

    python tools/accum_study.py [--n 2] [--r 64] [--seeds 3] [--chain]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import erfnet_oracle, fit_oracle, inputs  # noqa: E402

RULE = None
SEG = 0
RULES = {
    "r:64only1": lambda K: 1 if K == 192 else 0,
    "r:128only1": lambda K: 1 if K > 192 else 0,
    "r:16only1": lambda K: 1 if K < 192 else 0,
    "r:64x2_128x2_16x1": lambda K: 2 if K >= 192 else 1,
    "r:64x2_128x1_16x1": lambda K: 2 if K == 192 else 1,
    "r:64x4_128x2_16x1": lambda K: 4 if K == 192 else 2 if K > 192 else 1,
}
MODE = {"fold": False, "stats": False, "chain": False, "center": False, "chain2": False, "chain4": False, "chain2k": False}


def fma32(a, b, c):
    """fl32(a * b + c) for fp32 tensors (product exact in fp64; one rounding to 53 bits, one to 24: double rounding is rare)."""
    return (a.double() * b.double() + c.double()).float()


def bn_emul(x, P, prefix, training, stats_out):
    w, b = P[prefix + ".weight"], P[prefix + ".bias"]
    if x.dtype == torch.float64 or not training:
        return ORIG_BN(x, P, prefix, training, stats_out)
    if MODE["stats"]:
        # engine: per-256-pixel-tile fp32 partial sums (NHWC pixel order), combined in fp64
        N, C, H, W = x.shape
        xp = x.permute(0, 2, 3, 1).reshape(-1, C)
        npix = xp.shape[0]
        T = 256
        pad = (-npix) % T
        if pad:
            xp = torch.cat([xp, xp.new_zeros(pad, C)], 0)
        xt = xp.reshape(-1, T, C)
        # sequential fp32 accumulation inside a tile is emulated by torch's fp32 sum over 256 (pairwise-ish: optimistic)
        s1 = xt.sum(1, dtype=torch.float32).double().sum(0)
        s2 = (xt * xt).sum(1, dtype=torch.float32).double().sum(0)
        mean = s1 / npix
        var = (s2 / npix - mean * mean).clamp_min(0)
    else:
        mean = x.double().mean(dim=(0, 2, 3))
        var = x.double().var(dim=(0, 2, 3), unbiased=False)
    rstd = torch.rsqrt(var + erfnet_oracle.BN_EPS)
    if MODE["fold"]:
        sc = (w.double() * rstd).float()
        sh = (b.double() - mean * sc.double()).float()
        return fma32(x, sc[None, :, None, None], sh[None, :, None, None])
    if MODE["center"]:
        # candidate fix: y = fma(x - mean32, sc, beta)
        sc = (w.double() * rstd).float()
        m32 = mean.float()
        return fma32(x - m32[None, :, None, None], sc[None, :, None, None], b[None, :, None, None])
    m32, r32 = mean.float(), rstd.float()
    xh = (x - m32[None, :, None, None]) * r32[None, :, None, None]
    return xh * w[None, :, None, None] + b[None, :, None, None]


def conv_chain(x, w, bias, stride=1, padding=0, dilation=1):
    """conv2d as one sequential fp32 fma chain per output (bias first, then tap outer / channel inner)."""
    nacc = 4 if MODE["chain4"] else 2 if MODE["chain2"] else 1
    if RULE is not None:
        nacc = RULE(w.shape[1] * w.shape[2] * w.shape[3])
        if nacc == 0:
            return TF.conv2d(x, w, bias, stride=stride, padding=padding, dilation=dilation)
    if MODE["chain2k"]:      # two accumulator sets only where the contraction is longer than 192 (the 128-channel layers)
        nacc = 2 if w.shape[1] * w.shape[2] * w.shape[3] > 192 else 1
    if x.dtype == torch.float64 or not (SEG or MODE["chain"] or MODE["chain2k"] or nacc > 1 or RULE is not None):
        return TF.conv2d(x, w, bias, stride=stride, padding=padding, dilation=dilation)
    Co, Ci, kh, kw = w.shape
    pad = (padding, padding) if isinstance(padding, int) else padding
    dil = (dilation, dilation) if isinstance(dilation, int) else dilation
    st = (stride, stride) if isinstance(stride, int) else stride
    cols = TF.unfold(x, (kh, kw), dilation=dil, padding=pad, stride=st)          # (N, Ci*kh*kw, L), channel-major
    N, _, L = cols.shape
    cols = cols.reshape(N, Ci, kh * kw, L)
    wk = w.reshape(Co, Ci, kh * kw)
    if SEG:
        # flush form: a short running chain of SEG terms, added into a long-term accumulator at every segment boundary
        run_, long_ = torch.zeros(N, Co, L, dtype=torch.float32), None
        step = 0
        for t in range(kh * kw):
            for c in range(Ci):
                run_ = fma32(wk[:, c, t][None, :, None], cols[:, c, t, :][:, None, :], run_)
                step += 1
                if step % SEG == 0 or step == Ci * kh * kw:
                    long_ = run_ if long_ is None else long_ + run_
                    run_ = torch.zeros_like(run_)
        acc = long_ if bias is None else long_ + bias[None, :, None]
        Ho = (x.shape[2] + 2 * pad[0] - dil[0] * (kh - 1) - 1) // st[0] + 1
        return acc.reshape(N, Co, Ho, -1)
    accs = [torch.zeros(N, Co, L, dtype=torch.float32) for _ in range(nacc)]
    step = 0
    for t in range(kh * kw):
        for c in range(Ci):
            # the engine's K-step = one tap x 16 channels; consecutive K-steps alternate between the accumulator sets
            i = (step // 16) % nacc
            accs[i] = fma32(wk[:, c, t][None, :, None], cols[:, c, t, :][:, None, :], accs[i])
            step += 1
    acc = accs[0]
    if nacc == 2:
        acc = accs[0] + accs[1]
    elif nacc == 4:
        acc = (accs[0] + accs[1]) + (accs[2] + accs[3])
    if bias is not None:
        acc = acc + bias[None, :, None]
    Ho = (x.shape[2] + 2 * pad[0] - dil[0] * (kh - 1) - 1) // st[0] + 1
    return acc.reshape(N, Co, Ho, -1)


class FShim:
    conv2d = staticmethod(conv_chain)
    max_pool2d = staticmethod(TF.max_pool2d)
    conv_transpose2d = staticmethod(TF.conv_transpose2d)
    relu = staticmethod(TF.relu)


ORIG_BN = erfnet_oracle._bn


def run(x, P, dtype):
    Pd = erfnet_oracle.cast_params(P, dtype)
    with torch.no_grad():
        _, dec = erfnet_oracle.erfnet_forward(x.to(dtype), Pd, training=True)
    return dec.double().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--r", type=int, default=64)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--chain", action="store_true")
    a = ap.parse_args()
    erfnet_oracle._bn = bn_emul
    erfnet_oracle.F = FShim
    rms = lambda u, v: float(np.sqrt(np.mean((u - v) ** 2)))
    Mh, _ = fit_oracle.bev_homography()
    grid = fit_oracle.projective_grid(a.r, 2 * a.r, Mh.astype(np.float32), True, np.float32)
    zr = fit_oracle.zero_rows_of(a.r, 0.3)
    beta_of = lambda lg: fit_oracle.wls_forward(lg.astype(np.float32), grid, zr, 2, 0.0, 1.0, "square")["beta"]
    brows = {}
    variants = ["plain", "fold", "stats", "fold+stats"] + (["chain", "seg32", "seg32+fold", "seg32+stats", "seg32+fold+stats"] if a.chain else [])
    rows = {v: [] for v in variants}
    for seed in range(a.seeds):
        P = erfnet_oracle.make_params(seed=40 + seed, out_channels=2)
        x = torch.from_numpy(inputs.images(a.n, a.r, 2 * a.r, seed=500 + seed))
        for k in MODE:
            MODE[k] = False
        r64 = run(x, P, torch.float64)
        base = None
        for v in variants:
            global RULE, SEG
            RULE = RULES.get(v)
            SEG = int(v.split("+")[0][3:]) if v.startswith("seg") else 0
            for k in MODE:
                MODE[k] = k in v.split("+")
            lg = run(x, P, torch.float32)
            d = rms(lg, r64)
            bd = np.abs(beta_of(lg) - beta_of(r64)).max()
            if v == "plain":
                base, bbase = d, bd
            rows[v].append(d / base)
            brows.setdefault(v, []).append(bd / bbase)
        print("seed %d  plain rms %.3e  " % (seed, base) + "  ".join("%s %.2f" % (v, rows[v][-1]) for v in variants[1:]), flush=True)
    for v in variants:
        print("%-18s ratio to the plain fp32 leg: logits rms median %.2f  (min %.2f max %.2f)   lane coefficients (max norm) median %.2f (min %.2f max %.2f)"
              % (v, np.median(rows[v]), min(rows[v]), max(rows[v]), np.median(brows[v]), min(brows[v]), max(brows[v])))


if __name__ == "__main__":
    main()
