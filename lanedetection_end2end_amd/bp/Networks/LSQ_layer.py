"""``Networks.LSQ_layer`` of the BP tree (BP/Networks/LSQ_layer.py): same public names."""
import torch

from lanedetection_end2end_amd import geometry
from lanedetection_end2end_amd.clas import Classification  # noqa: F401
from lanedetection_end2end_amd.fit import WeightedLeastSquares
from lanedetection_end2end_amd.geometry import get_homography  # noqa: F401
from lanedetection_end2end_amd.lsq import BPNet as Net, activation_layer  # noqa: F401


def ProjectiveGridGenerator(size, theta, no_cuda):
    """(N, H*W, 2) pixel-coordinate grid -- BP/Networks/LSQ_layer.py:50-68."""
    N, C, H, W = size
    g = geometry.projective_grid(H, W, theta[0].detach().double().cpu().numpy(), False)
    if not no_cuda:
        g = g.cuda()
    return g.unsqueeze(0).expand(N, -1, -1)


class Weighted_least_squares(WeightedLeastSquares):
    """BP flavour: y = 255 - grid_y, orders 0..3, fp64 betas (BP/Networks/LSQ_layer.py:72-154)."""
    y_offset = 255.0
    max_order = 3
    out_dtype = torch.float64
    cholesky_drops_reg = True    # the --use_cholesky branch is GELS.apply(Y0, W*x): no regulariser (BP/Networks/LSQ_layer.py:112-119, gels.py)
