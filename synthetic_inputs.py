"""Seeded synthetic inputs shared by bench.py, the tests and the golden generators.

A neutral module on purpose: bench.py's measured path needs synthetic tensors but must import nothing from ``oracle/`` (the
checker package), so the generators live here and ``oracle/inputs.py`` re-exports them for the golden recipes.  numpy
``default_rng`` (PCG64) streams are stable across numpy versions, so golden files store outputs only and the tests regenerate
their inputs.  Data only: no arithmetic of the hot path.
"""
import numpy as np


def images(N, H, W, seed=0, C=3):
    """(N,3,H,W) fp32 in [0,1) -- ToTensor() range (BEV/Dataloader/Load_Data_new.py:104)."""
    return np.random.default_rng(seed).random((N, C, H, W), dtype=np.float32)


def lane_like_logits(N, K, H, W, seed=0, sigma=0.02, noise=0.05):
    """Backbone-output-like logits: one curved Gaussian ridge per lane + uniform noise.

    The WLS weight is logit^4 (square activation, then squared again by the normal
    equations), so a ridge gives the realistic, well-conditioned case and the noise floor
    exercises the rest of the image.  Coordinates are normalised so that any (H,W) works.
    """
    rng = np.random.default_rng(seed)
    ys = (np.arange(H, dtype=np.float64) / H)[None, None, :, None]
    xs = (np.arange(W, dtype=np.float64) / W)[None, None, None, :]
    c = 0.3 + 0.4 * (np.arange(K)[None, :, None, None] + 0.5) / K + rng.uniform(-0.03, 0.03, (N, K, 1, 1))
    b = rng.uniform(-0.4, 0.4, (N, K, 1, 1))
    a = rng.uniform(-0.3, 0.3, (N, K, 1, 1))
    centre = c + b * (ys - 0.6) + a * (ys - 0.6) ** 2
    ridge = np.exp(-0.5 * ((xs - centre) / sigma) ** 2)
    out = ridge + noise * rng.uniform(-1, 1, (N, K, H, W))
    return out.astype(np.float32)


def bev_gt_params(N, seed=0):
    """(N,4,3) fp32: lanes 0,1 present, lanes 2,3 absent (all-zero) -- BEV/main.py:209-210."""
    rng = np.random.default_rng(seed)
    p = np.zeros((N, 4, 3), dtype=np.float32)
    for k, c0 in enumerate((0.45, 0.55)):
        p[:, k, 0] = rng.uniform(-0.1, 0.1, N)
        p[:, k, 1] = rng.uniform(-0.3, 0.3, N)
        p[:, k, 2] = c0 + rng.uniform(0, 0.03, N) * (1 if k else -1)
    return p


def bp_targets(N, K, resize=256, seed=0):
    """(lanes (N,K,56) f64 in [0,2R), valid (N,K,56) f64 in {0,1}, first 8 columns 0).

    BP/Dataloader/Load_Data_new.py:140-141.
    """
    rng = np.random.default_rng(seed)
    lanes = rng.uniform(0, 2 * resize, (N, K, 56))
    valid = (rng.random((N, K, 56)) < 0.8).astype(np.float64)
    valid[:, :, :8] = 0
    return lanes, valid


def seg_targets(N, H, W, nclass, seed=0):
    return np.random.default_rng(seed).integers(0, nclass, (N, H, W)).astype(np.int64)
