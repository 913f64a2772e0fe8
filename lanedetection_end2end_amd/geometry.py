"""Host-side constants of the fitting head: the fixed homographies and the projective grid.

These are built once at module construction, exactly where the reference builds them
(BEV/Networks/LSQ_layer.py:17-32,66-82; BP/Networks/utils.py:104-121, BP/Networks/LSQ_layer.py:50-68),
with the same fp32 torch ops, so the device kernels see the reference's own grid values.
``cv2.getPerspectiveTransform`` (OpenCV, not a dependency here) is replaced by its documented
8x8 DLT solve in float64.
"""
import numpy as np
import torch


def perspective_transform(src, dst):
    """3x3 H with H[2,2] = 1 mapping 4 float32 points src -> dst (OpenCV getPerspectiveTransform)."""
    s = np.asarray(src, np.float32).astype(np.float64)
    d = np.asarray(dst, np.float32).astype(np.float64)
    A, b = np.zeros((8, 8)), np.zeros(8)
    for i, ((x, y), (u, v)) in enumerate(zip(s, d)):
        A[i] = (x, y, 1, 0, 0, 0, -x * u, -y * u)
        A[i + 4] = (0, 0, 0, x, y, 1, -x * v, -y * v)
        b[i], b[i + 4] = u, v
    return np.append(np.linalg.solve(A, b), 1.0).reshape(3, 3)


def bev_homography():
    """Normalised-coordinate M, M_inv (float64) -- BEV/Networks/LSQ_layer.py:22-29."""
    top, bottom = 0.3, 1
    src = np.float32([[0.45, top], [0.55, top], [0.1, bottom], [0.9, bottom]])
    dst = np.float32([[0.45, top], [0.55, top], [0.45, bottom], [0.55, bottom]])
    return perspective_transform(src, dst), perspective_transform(dst, src)


def get_homography(resize=256, no_mapping=False):
    """Pixel-coordinate M, M_inv (float64) -- same contract as BP/Networks/utils.py:104-121."""
    if no_mapping:
        return np.identity(3), np.identity(3)
    w = 2 * resize
    top, bottom = 0.20 * resize, resize - 1
    src = np.float32([[0.45 * w, top], [0.55 * w, top], [0.02 * w, bottom], [0.97 * w, bottom]])
    dst = np.float32([[0.45 * w, top], [0.55 * w, top], [0.45 * w, bottom], [0.55 * w, bottom]])
    return perspective_transform(src, dst), perspective_transform(dst, src)


def projective_grid(H, W, M, normalised):
    """(H*W, 2) fp32 CPU tensor of (x', y') per pixel.

    normalised=True: BEV base grid linspace(0, 1-1/W, W) x linspace(0, 1-1/H, H)
    (LSQ_layer.py:70-71); False: pixel indices (BP LSQ_layer.py:53-54).  [x y 1] M^T then
    perspective divide, in fp32 like the reference (:84-87 / :64-65).
    """
    M = torch.as_tensor(np.asarray(M), dtype=torch.float64).float()
    if normalised:
        xs, ys = torch.linspace(0, 1 - 1 / W, W), torch.linspace(0, 1 - 1 / H, H)
    else:
        xs, ys = torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H)
    base = torch.ones(1, H, W, 3)
    base[0, :, :, 0] = xs[None, :]
    base[0, :, :, 1] = ys[:, None]
    g = torch.bmm(base.view(1, H * W, 3), M.t().unsqueeze(0))
    return torch.div(g[0, :, 0:2], g[0, :, 2:]).contiguous()
