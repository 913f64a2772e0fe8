"""CPU restatement of the input pipeline in front of the hot path.

TEST INFRASTRUCTURE ONLY (imported by tests/ and oracle/gen_golden_pipeline.py).

The reference (BEV/Dataloader/Load_Data_new.py:62-117, BP/Dataloader/Load_Data_new.py:110-197) runs, per
sample, on PIL images through torchvision.transforms.functional:
    F.crop(img, h-640, 0, 640, w)                      -> img.crop((0, h-640, w, h))
    F.resize(img, (R, 2R), Image.BILINEAR / NEAREST)   -> img.resize((2R, R), resample)
    F.hflip(img)                                       -> img.transpose(FLIP_LEFT_RIGHT)
    ToTensor()(img)                                    -> uint8 HWC -> float32 CHW, .div(255)
The resampling itself lives in a third-party dependency that is not under /root/reference: Pillow
(libImaging/Resample.c for BILINEAR -- a two-pass separable convolution with an antialiasing triangle filter,
22-bit fixed-point coefficients and a uint8 clip after EACH pass; libImaging/Geometry.c ImagingScaleAffine for
NEAREST -- an incrementally accumulated source coordinate).  torchvision is not installed in this image;
Pillow 12.2.0 is, so the restatement below is pinned bit-exactly against the real ``Image.resize`` by
tests/golden/pipeline.npz (oracle/gen_golden_pipeline.py) and, when Pillow is importable, directly in
tests/test_pipeline_oracle.py.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bilinear_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the triangle filter (support 1.0) over the whole
    axis: returns (bounds (out,2) int32 [first, count], kk (out, ksize) int32 fixed-point weights)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(xmax, dtype=np.float64)
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = 0.0            # Pillow accumulates sequentially in double; keep the order
        for x in range(xmax):
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        fixed = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64),
                         (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
        kk[xx, :xmax] = fixed
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, out_h, out_w):
    """img (H, W, C) uint8 -> (out_h, out_w, C) uint8, as Image.resize((out_w, out_h), BILINEAR): horizontal
    pass, clip to uint8, vertical pass, clip to uint8."""
    H, W, C = img.shape
    bx, kx = bilinear_coeffs(W, out_w)
    by, ky = bilinear_coeffs(H, out_h)
    tmp = np.empty((H, out_w, C), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_w):
        x0, n = bx[xx]
        acc = (src[:, x0:x0 + n, :] * kx[xx, :n, None].astype(np.int64)).sum(1) + (1 << (PRECISION_BITS - 1))
        tmp[:, xx, :] = _clip8(acc)
    out = np.empty((out_h, out_w, C), dtype=np.uint8)
    t = tmp.astype(np.int64)
    for yy in range(out_h):
        y0, n = by[yy]
        acc = (t[y0:y0 + n] * ky[yy, :n, None, None].astype(np.int64)).sum(0) + (1 << (PRECISION_BITS - 1))
        out[yy] = _clip8(acc)
    return out


def nearest_table(in_size, out_size):
    """Source index per output index of ImagingScaleAffine (NEAREST): the coordinate starts at scale/2 and is
    ACCUMULATED in double (xo += scale), truncated toward zero."""
    a = float(in_size) / out_size
    xo = a * 0.5
    tab = np.empty(out_size, dtype=np.int32)
    for x in range(out_size):
        xin = -1 if xo < 0.0 else int(xo)
        tab[x] = min(max(xin, 0), in_size - 1)
        xo += a
    return tab


def resize_nearest_u8(img, out_h, out_w):
    H, W = img.shape[:2]
    return img[nearest_table(H, out_h)][:, nearest_table(W, out_w)]


def totensor_times255_lut():
    """(ToTensor()(gt) * 255).long() per uint8 value: fp32 v/255*255 truncated (not always v)."""
    v = np.arange(256, dtype=np.float32)
    return ((v / np.float32(255)) * np.float32(255)).astype(np.int64)


def preprocess_image(frame, resize, flip):
    """frame (H, W, 3) uint8 -> (3, R, 2R) float32 in [0, 1] (Load_Data_new.py:77-79,87,101)."""
    h = frame.shape[0]
    img = resize_bilinear_u8(frame[h - 640:], resize, 2 * resize)
    if flip:
        img = img[:, ::-1]
    return (img.transpose(2, 0, 1).astype(np.float32) / np.float32(255))


def preprocess_label(label, resize, flip, tree="bev", nclasses=2):
    """label (H, W) uint8 palette indices -> (1, R, 2R) int64, with the class remapping / flipping statements of
    the two loaders replayed literally (BEV :80-92, BP :152-168 -- including BP's use of the PRE-flip masks
    of classes 3/4 on the flipped array)."""
    h = label.shape[0]
    gt = resize_nearest_u8(label[h - 640:], resize, 2 * resize).copy()
    idx3, idx4 = gt == 3, gt == 4
    if tree == "bev" or nclasses < 3:
        gt[idx3] = 0
        gt[idx4] = 0
    if flip:
        gt = gt[:, ::-1].copy()
        idx1, idx2 = gt == 1, gt == 2
        gt[idx1] = 2
        gt[idx2] = 1
        if tree == "bp":
            gt[idx3] = 4
            gt[idx4] = 3
    return totensor_times255_lut()[gt][None]


def bev_horizon(gt_long):
    """gt (1, R, 2R) -> (R,) float32: ones above the first labelled row (BEV :103-105)."""
    rows = np.nonzero(gt_long[0].any(1))[0]
    y_val = rows[0] if len(rows) else gt_long.shape[1]
    hz = np.zeros(gt_long.shape[1], dtype=np.float32)
    hz[:y_val] = 1
    return hz


def bev_flip_params(params):
    """params (4, 3): swap left/right pairs, negate, shift the offset (BEV :93-96)."""
    p = -np.asarray(params, dtype=np.float64)[[1, 0, 3, 2]]
    p[:, -1] = 1 + p[:, -1]
    return p


def mirror_list(lst):
    """Load_Data_new.py mirror_list."""
    middle = len(lst) // 2
    return list(reversed(lst[middle:])) + list(reversed(lst[:middle]))


def synthetic_frame(seed, H=720, W=1280):
    """A decoded-camera-like uint8 frame (smooth shading + texture noise + bright lane stripes) and a palette
    label map with classes 0..4 drawn as lane stripes in the lower part (seeded; shared by goldens and tests)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    frame = np.stack([96 + 60 * np.sin(xx / 97.0 + seed) + 40 * np.cos(yy / 53.0),
                      110 + 50 * np.cos(xx / 131.0) * np.sin(yy / 71.0 + seed),
                      90 + 0.08 * xx + 0.05 * yy], 2)
    frame += rng.normal(0, 12, (H, W, 3))
    label = np.zeros((H, W), dtype=np.uint8)
    top = int(rng.integers(250, 330))
    for cls, (x_top, x_bot) in enumerate(((560, 260), (700, 1010), (520, -200), (760, 1500)), start=1):
        x_top += int(rng.integers(-30, 30))
        for y in range(top + 7 * cls, H):
            t = (y - top) / float(H - top)
            xc = x_top + (x_bot - x_top) * t
            half = 2 + 9 * t
            lo, hi = int(max(0, xc - half)), int(min(W, xc + half))
            if lo < hi:
                label[y, lo:hi] = cls
                frame[y, lo:hi] += 90
    return np.clip(frame, 0, 255).astype(np.uint8), label


def checksum(a):
    """Order-sensitive 64-bit checksum of an integer array (golden fixtures store this instead of full tensors)."""
    v = np.ascontiguousarray(a).astype(np.uint64).reshape(-1)
    idx = np.arange(1, v.size + 1, dtype=np.uint64)
    return np.array([v.sum(dtype=np.uint64), (v * (idx % np.uint64(65521) + np.uint64(1))).sum(dtype=np.uint64)], dtype=np.uint64)
