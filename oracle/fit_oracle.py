"""numpy restatement of the reference's fitting head and losses (CPU oracle).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Every function cites the
reference file:line it follows.  Shorthand:

* ``BEV/`` = /root/reference/Birds_Eye_View_Loss/
* ``BP/``  = /root/reference/Backprojection_Loss/

Pinning: ``oracle/gen_golden.py`` runs the *real* reference modules (imported from
/root/reference in the authoring container) on seeded inputs and stores their
outputs under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks every
function here against those files.  The one boundary that stays unpinned is
``cv2.getPerspectiveTransform`` (OpenCV is a third-party dependency that is not in
/root/reference and not installed here; README.md:32 lists just "opencv", version
unpinned): ``get_perspective_transform`` restates OpenCV's documented algorithm
(8x8 DLT system with h33 = 1, solved in float64) and is checked against the
closed-form matrices quoted in SURVEY.md 8c(i)/(vi).

All arithmetic is float64 unless ``dtype`` says otherwise, i.e. this is the
"reference formula evaluated in fp64" that parity numbers are quoted against.
"""
from math import ceil

import numpy as np

# ----------------------------------------------------------------------------
# homographies
# ----------------------------------------------------------------------------


def get_perspective_transform(src, dst):
    """3x3 homography mapping 4 src points to 4 dst points, h33 = 1.

    Restates OpenCV ``cv::getPerspectiveTransform`` (third-party, absent): for each
    pair (x,y)->(u,v):  [x y 1 0 0 0 -xu -yu] h = u ; [0 0 0 x y 1 -xv -yv] h = v.
    Call sites: BEV/Networks/LSQ_layer.py:28-29, BP/Networks/utils.py:119-120.
    Inputs are float32 arrays (as in the reference); the solve is float64.
    """
    src = np.asarray(src, dtype=np.float32).astype(np.float64)
    dst = np.asarray(dst, dtype=np.float32).astype(np.float64)
    A = np.zeros((8, 8), dtype=np.float64)
    b = np.zeros(8, dtype=np.float64)
    for i in range(4):
        x, y = src[i]
        u, v = dst[i]
        A[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        A[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i] = u
        b[i + 4] = v
    h = np.linalg.solve(A, b)
    return np.append(h, 1.0).reshape(3, 3)


def bev_homography():
    """(M, M_inv) in normalised coordinates -- BEV/Networks/LSQ_layer.py:17-32."""
    y_start, y_stop = 0.3, 1
    xd1, xd2, xd3, xd4 = 0.45, 0.55, 0.45, 0.55
    src = np.float32([[0.45, y_start], [0.55, y_start], [0.1, y_stop], [0.9, y_stop]])
    dst = np.float32([[xd3, y_start], [xd4, y_start], [xd1, y_stop], [xd2, y_stop]])
    return get_perspective_transform(src, dst), get_perspective_transform(dst, src)


def bp_homography(resize=256, no_mapping=False):
    """(M, M_inv) in pixel coordinates -- BP/Networks/utils.py:104-121."""
    if no_mapping:
        return np.identity(3), np.identity(3)
    y_start = 0.20 * resize
    y_stop = resize - 1
    src = np.float32([[0.45 * (2 * resize), y_start], [0.55 * (2 * resize), y_start],
                      [0.02 * (2 * resize), y_stop], [0.97 * (2 * resize), y_stop]])
    dst = np.float32([[0.45 * (2 * resize), y_start], [0.55 * (2 * resize), y_start],
                      [0.45 * (2 * resize), y_stop], [0.55 * (2 * resize), y_stop]])
    return get_perspective_transform(src, dst), get_perspective_transform(dst, src)


# ----------------------------------------------------------------------------
# projective grid
# ----------------------------------------------------------------------------


def projective_grid(H, W, M, normalised, dtype=np.float64):
    """(H*W, 2) grid of (x', y') for every pixel, row-major.

    BEV (normalised=True): base x = linspace(0, 1-1/W, W), y = linspace(0, 1-1/H, H)
    (BEV/Networks/LSQ_layer.py:70-71), g = [x,y,1] M^T, grid = g[:2]/g[2] (:84-87).
    BP (normalised=False): base x = 0..W-1, y = 0..H-1 (BP/Networks/LSQ_layer.py:53-54,
    :64-65).  ``dtype`` = float32 mimics the reference's own precision (M is cast to
    fp32 first: BEV :30, BP :219).
    """
    M = np.asarray(M).astype(dtype)
    if normalised:
        xs = np.linspace(0, 1 - 1 / W, W).astype(dtype)
        ys = np.linspace(0, 1 - 1 / H, H).astype(dtype)
    else:
        xs = np.linspace(0, W - 1, W).astype(dtype)
        ys = np.linspace(0, H - 1, H).astype(dtype)
    base = np.empty((H, W, 3), dtype=dtype)
    base[:, :, 0] = xs[None, :]
    base[:, :, 1] = ys[:, None]
    base[:, :, 2] = 1
    g = base.reshape(H * W, 3) @ M.T
    with np.errstate(divide="ignore", invalid="ignore"):
        return (g[:, 0:2] / g[:, 2:3]).astype(dtype)


# ----------------------------------------------------------------------------
# activation + row mask
# ----------------------------------------------------------------------------

ACTIVATIONS = ("square", "abs", "relu", "sigmoid", "softplus", "none")


def activation(o, kind):
    """BEV/Networks/LSQ_layer.py:43-63 (square :35-36, abs torch.abs, ...)."""
    if kind == "square":
        return o * o
    if kind == "abs":
        return np.abs(o)
    if kind == "relu":
        return np.maximum(o, 0)
    if kind == "sigmoid":
        return 1.0 / (1.0 + np.exp(-o))
    if kind == "softplus":
        # nn.Softplus(beta=1, threshold=20)
        return np.where(o > 20, o, np.log1p(np.exp(np.minimum(o, 20))))
    if kind == "none":
        return o
    raise NotImplementedError(kind)


def activation_grad(o, kind):
    """d act / d o  (2o, sign(o), [o>0], s(1-s), sigmoid(o), 1)."""
    if kind == "square":
        return 2 * o
    if kind == "abs":
        return np.sign(o)
    if kind == "relu":
        return (o > 0).astype(o.dtype)
    if kind == "sigmoid":
        s = 1.0 / (1.0 + np.exp(-o))
        return s * (1 - s)
    if kind == "softplus":
        return np.where(o > 20, 1.0, 1.0 / (1.0 + np.exp(-o)))
    if kind == "none":
        return np.ones_like(o)
    raise NotImplementedError(kind)


def zero_rows_of(resize, mask_percentage):
    """BEV/Networks/LSQ_layer.py:257 -- rows [0, ceil(resize*mask_percentage)) are zeroed."""
    return int(ceil(resize * mask_percentage))


# ----------------------------------------------------------------------------
# weighted least squares layer
# ----------------------------------------------------------------------------


def design_matrix(grid_y, order, y_offset):
    """Y = [y^d ... y 1], y = y_offset - grid_y.

    BEV: y_offset = 1 (LSQ_layer.py:109), orders 0..2 (:110-118).
    BP: y_offset = 255 (BP LSQ_layer.py:94), orders 0..3 (:97-107).
    """
    y = y_offset - grid_y
    return np.stack([y ** (order - j) for j in range(order + 1)], axis=-1)


def wls_forward(logits, grid, zero_rows, order=2, reg=0.0, y_offset=1.0, act="square",
                skip_masked=True):
    """beta for every (image, lane), plus everything backward needs.

    Follows ``Net.forward`` steps 2-4 (BEV/Networks/LSQ_layer.py:310-325) and
    ``Weighted_least_squares.forward`` (:103-167; BP :85-154):
    w = act(o), rows < zero_rows -> 0, Y0 = w*Y, Z = Y0^T Y0 + reg*I,
    X = Y0^T (w*x), beta = Z^-1 X.  The per-pixel weight is therefore s = w^2.

    ``skip_masked``: masked rows contribute exactly 0 in the reference *when the grid
    is finite there*; at 320x640 in BP the grid has a pole on a masked row and the
    reference returns NaN.  Skipping them is the documented
    deviation (the HIP kernel never touches masked rows either).

    logits: (N,K,H,W); grid: (H*W,2).  Returns dict(beta (N,K,d+1), Zinv, w, masked, ...).
    """
    o = np.asarray(logits, dtype=np.float64)
    N, K, H, W = o.shape
    g = np.asarray(grid, dtype=np.float64).reshape(H, W, 2)
    w = activation(o, act)
    w[:, :, :zero_rows, :] = 0
    r0 = zero_rows if skip_masked else 0
    gx = g[r0:, :, 0].reshape(-1)
    Y = design_matrix(g[r0:, :, 1].reshape(-1), order, y_offset)  # (P', d+1)
    wv = w[:, :, r0:, :].reshape(N, K, -1)
    s = wv * wv
    Z = np.einsum("nkp,pi,pj->nkij", s, Y, Y) + reg * np.eye(order + 1)
    X = np.einsum("nkp,p,pi->nki", s, gx, Y)
    Zinv = np.linalg.inv(Z)
    beta = np.einsum("nkij,nkj->nki", Zinv, X)
    return dict(beta=beta, Z=Z, X=X, Zinv=Zinv, w=w, logits=o, grid=g, zero_rows=zero_rows,
                order=order, y_offset=y_offset, act=act, r0=r0)


def wls_backward(cache, grad_beta):
    """d loss / d logits given d loss / d beta (N,K,d+1).

    Derived from the reference forward (autograd through bmm/inverse), SURVEY.md 2.2:
    v = Z^-1 g (Z symmetric), r_i = x_i - Y_i.beta, dL/ds_i = (Y_i.v) r_i,
    dL/dw_i = 2 w_i (Y_i.v) r_i, dL/do_i = act'(o_i) dL/dw_i, 0 on masked rows.
    ``gels.py:17-25`` is the same expression written on A = Y0, b = w*x.
    """
    o, w, g = cache["logits"], cache["w"], cache["grid"]
    N, K, H, W = o.shape
    r0, order = cache["r0"], cache["order"]
    gx = g[r0:, :, 0].reshape(-1)
    Y = design_matrix(g[r0:, :, 1].reshape(-1), order, cache["y_offset"])
    v = np.einsum("nkij,nkj->nki", cache["Zinv"], np.asarray(grad_beta, dtype=np.float64))
    Yv = np.einsum("pi,nki->nkp", Y, v)
    resid = gx[None, None, :] - np.einsum("pi,nki->nkp", Y, cache["beta"])
    wv = w[:, :, r0:, :].reshape(N, K, -1)
    dw = 2 * wv * Yv * resid
    grad = np.zeros_like(o)
    grad[:, :, r0:, :] = dw.reshape(N, K, H - r0, W)
    grad *= activation_grad(o, cache["act"])
    grad[:, :, :cache["zero_rows"], :] = 0
    return grad


def gels_forward(A, b):
    """BP/Networks/gels.py:10-15 -- beta = (A^T A)^-1 A^T b via Cholesky, no reg."""
    A = np.asarray(A, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    AtA = np.swapaxes(A, -1, -2) @ A
    L = np.linalg.cholesky(AtA)
    Atb = np.swapaxes(A, -1, -2) @ b
    y = np.linalg.solve(L, Atb)
    return np.linalg.solve(np.swapaxes(L, -1, -2), y), AtA


def gels_backward(A, b, x, AtA, grad_out):
    """BP/Networks/gels.py:17-25."""
    z = np.linalg.solve(AtA, grad_out)
    xzt = x @ np.swapaxes(z, -1, -2)
    zx_sym = xzt + np.swapaxes(xzt, -1, -2)
    grad_A = -A @ zx_sym + b @ np.swapaxes(z, -1, -2)
    grad_b = A @ z
    return grad_A, grad_b


# ----------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------


def mse_loss(params, gt):
    """``MSE_Loss.forward`` -- BEV/Loss_crit.py:137-150: ``nn.MSELoss()(params.squeeze(-1), gt_params)`` = the mean of the
    squared difference over ALL elements.  Returns (loss, d loss / d params) with the gradient in params' shape."""
    p = np.asarray(params, dtype=np.float64)
    d = p.reshape(len(p), -1) - np.asarray(gt, dtype=np.float64)
    return float((d ** 2).mean()), (2.0 * d / d.size).reshape(p.shape)


def area_loss(beta, gt, order=2, weight_funct="none", t=0.7):
    """``Area_Loss.forward`` -- BEV/Loss_crit.py:98-134.  beta (N,d+1[,1]), gt (N,d+1).

    Returns (loss, d loss / d beta).  Lanes whose gt has any zero coefficient are dropped
    (:131-132); mean over the kept ones, python 0 if none (:133).
    """
    beta = np.asarray(beta, dtype=np.float64).reshape(len(beta), -1)
    gt = np.asarray(gt, dtype=np.float64)
    diff = beta - gt
    a, b = diff[:, 0], diff[:, 1]
    g = np.zeros_like(diff)
    if order == 2:
        c = diff[:, 2]
        if weight_funct == "none":
            L = a**2 * t**5 / 5 + 2 * a * b * t**4 / 4 + (b**2 + c * 2 * a) * t**3 / 3 \
                + 2 * b * c * t**2 / 2 + c**2 * t
            g[:, 0] = 2 * a * t**5 / 5 + b * t**4 / 2 + 2 * c * t**3 / 3
            g[:, 1] = a * t**4 / 2 + 2 * b * t**3 / 3 + c * t**2
            g[:, 2] = 2 * a * t**3 / 3 + b * t**2 + 2 * c * t
        elif weight_funct == "linear":
            L = c**2 * t - t**5 * ((2 * a * b) / 5 - a**2 / 5) + t**2 * (b * c - c**2 / 2) \
                - (a**2 * t**6) / 6 - t**4 * (b**2 / 4 - (a * b) / 2 + (a * c) / 2) \
                + t**3 * (b**2 / 3 - (2 * c * b) / 3 + (2 * a * c) / 3)
            g[:, 0] = -t**5 * (2 * b / 5 - 2 * a / 5) - a * t**6 / 3 \
                - t**4 * (-b / 2 + c / 2) + t**3 * (2 * c / 3)
            g[:, 1] = -t**5 * (2 * a / 5) + t**2 * c - t**4 * (b / 2 - a / 2) \
                + t**3 * (2 * b / 3 - 2 * c / 3)
            g[:, 2] = 2 * c * t + t**2 * (b - c) - t**4 * (a / 2) + t**3 * (-2 * b / 3 + 2 * a / 3)
        elif weight_funct == "quadratic":
            L = t**3 * (1 / 3 * b**2 + 2 / 3 * a * c) - t**(7 / 2) * (2 / 7 * b**2 + 4 / 7 * a * c) \
                + c**2 * t + 0.2 * a**2 * t**5 - 2 / 11 * a**2 * t**(11 / 2) \
                - 2 / 3 * c**2 * t**(3 / 2) + 0.5 * a * b * t**4 - 4 / 9 * a * b * t**(9 / 2) \
                + b * c * t**2 - 0.8 * b * c * t**(5 / 2)
            g[:, 0] = t**3 * (2 / 3 * c) - t**(7 / 2) * (4 / 7 * c) + 0.4 * a * t**5 \
                - 4 / 11 * a * t**(11 / 2) + 0.5 * b * t**4 - 4 / 9 * b * t**(9 / 2)
            g[:, 1] = t**3 * (2 / 3 * b) - t**(7 / 2) * (4 / 7 * b) + 0.5 * a * t**4 \
                - 4 / 9 * a * t**(9 / 2) + c * t**2 - 0.8 * c * t**(5 / 2)
            g[:, 2] = t**3 * (2 / 3 * a) - t**(7 / 2) * (4 / 7 * a) + 2 * c * t \
                - 4 / 3 * c * t**(3 / 2) + b * t**2 - 0.8 * b * t**(5 / 2)
        else:
            raise NotImplementedError(weight_funct)
    elif order == 1:
        L = b**2 * t + a * b * t**2 + (a**2 * t**3) / 3
        g[:, 0] = b * t**2 + 2 * a * t**3 / 3
        g[:, 1] = 2 * b * t + a * t**2
    else:
        raise NotImplementedError(order)
    keep = np.prod(gt != 0, axis=1).astype(bool)
    n = int(keep.sum())
    if n == 0:
        return 0.0, np.zeros_like(diff)
    return float(L[keep].mean()), g * keep[:, None] / n


def trapezoidal(coeffs_a, coeffs_b, lo=0.0, hi=0.7, n=100):
    """``polynomial.trapezoidal`` -- BEV/Loss_crit.py:12-35.  coeffs (N,3[,1]) -> (N,)."""
    pa = np.asarray(coeffs_a, dtype=np.float64).reshape(len(coeffs_a), -1)
    pb = np.asarray(coeffs_b, dtype=np.float64).reshape(len(coeffs_b), -1)

    def pol(p, x):
        return p[:, 0] * x**2 + p[:, 1] * x + p[:, 2]

    h = float(hi - lo) / n
    s = np.abs(pol(pa, lo) / 2.0 - pol(pb, lo) / 2.0)
    for i in range(1, n):
        s = s + np.abs(pol(pa, lo + i * h) - pol(pb, lo + i * h))
    s = s + np.abs(pol(pa, hi) / 2.0 - pol(pb, hi) / 2.0)
    return s * h


def backproj_setup(order, resize=256, no_mapping=False):
    """Constants of ``backprojection_loss.__init__`` -- BP/Loss_crit.py:166-200.

    Returns dict(M, M_inv, y_d, y_prime, Y (56,d+1)).  Note the literal 80, 2.5 and 255
    (they assume resize = 256; reproduced literally).
    """
    M, M_inv = bp_homography(resize, no_mapping)
    y_d = (np.arange(160, 720, 10) - 80).astype(np.float64) / 2.5
    y_prime = (M[1, 1] * y_d + M[1, 2]) / (M[2, 1] * y_d + M[2, 2])
    y_eval = 255 - y_prime
    Y = np.stack([y_eval ** (order - j) for j in range(order + 1)], axis=1)
    return dict(M=M, M_inv=M_inv, y_d=y_d, y_prime=y_prime, Y=Y)


def backproj_loss(beta, x_gt, valid, setup):
    """``backprojection_loss.forward`` -- BP/Loss_crit.py:202-218.

    beta (N,d+1[,1]); x_gt, valid (N,56).  Returns (loss, x_cal*valid, d loss/d beta).
    The normaliser is the batch-wide sum of ``valid`` (:215), loss = 0 when it is 0 (:216-217).
    """
    beta = np.asarray(beta, dtype=np.float64).reshape(len(beta), -1)
    x_gt = np.asarray(x_gt, dtype=np.float64)
    valid = np.asarray(valid, dtype=np.float64)
    Y, Mi, yp = setup["Y"], setup["M_inv"], setup["y_prime"]
    xp = beta @ Y.T                                    # (N,56)
    t0 = Mi[0, 0] * xp + Mi[0, 1] * yp[None] + Mi[0, 2]
    t2 = Mi[2, 0] * xp + Mi[2, 1] * yp[None] + Mi[2, 2]
    x_cal = t0 / t2
    err = (x_gt - x_cal) * valid
    nv = valid.sum()
    if nv == 0:
        return 0.0, x_cal * valid, np.zeros_like(beta)
    loss = float((err**2).sum() / nv)
    dxcal_dxp = (Mi[0, 0] * t2 - Mi[2, 0] * t0) / t2**2
    gx = -2 * err * valid / nv * dxcal_dxp            # (N,56)
    return loss, x_cal * valid, gx @ Y


def cross_entropy_2d(logits, target, weights):
    """Class-weighted pixel CE, weighted-mean reduction.

    BEV ``CrossEntropyLoss2d`` = NLLLoss2d(log_softmax) (BEV/Loss_crit.py:61-75, caller
    passes target[:,0]); BP ``nn.CrossEntropyLoss(weights)`` (BP/Loss_crit.py:64-65).
    logits (N,C,H,W), target (N,H,W) int.  Returns (loss, d loss / d logits).
    """
    z = np.asarray(logits, dtype=np.float64)
    N, C, H, W = z.shape
    wts = np.asarray(weights, dtype=np.float64)
    zmax = z.max(axis=1, keepdims=True)
    lse = zmax + np.log(np.exp(z - zmax).sum(axis=1, keepdims=True))
    logp = z - lse
    tgt = np.asarray(target).astype(np.int64)
    pick = np.take_along_axis(logp, tgt[:, None], axis=1)[:, 0]
    wpix = wts[tgt]
    den = wpix.sum()
    loss = float(-(wpix * pick).sum() / den)
    onehot = np.zeros_like(z)
    np.put_along_axis(onehot, tgt[:, None], 1.0, axis=1)
    grad = (np.exp(logp) - onehot) * wpix[:, None] / den
    return loss, grad
