"""The reference's ``LSQ_layer.Net`` wrappers (backbone -> activation -> row mask -> grid -> WLS),
BEV flavour (normalised coordinates, fp32 betas) and BP flavour (pixel coordinates, fp64 betas).

Reference: BEV/Networks/LSQ_layer.py:231-326, BP/Networks/LSQ_layer.py:210-315.
"""
from math import ceil

import numpy as np
import torch
import torch.nn as nn

from . import erfnet, fit, geometry, ops
from .clas import Classification, ClassificationBEV


def activation_layer(activation='square', no_cuda=False):
    """Returns a callable like the reference's (LSQ_layer.py:43-63); inside ``Net`` the activation is
    fused into the WLS kernel, this exists for API compatibility."""
    table = {'sigmoid': torch.sigmoid, 'relu': torch.relu, 'softplus': nn.functional.softplus,
             'square': lambda x: x ** 2, 'abs': torch.abs, 'none': lambda x: x}
    if activation not in table:
        raise NotImplementedError('Activation type: {} is not implemented'.format(activation))
    return table[activation]


class _LaneFitNet(nn.Module):
    normalised = True
    y_offset = 1.0
    beta_dtype = torch.float32
    max_order = 2
    cholesky_drops_reg = False       # BP only: its --use_cholesky branch is GELS, which has no regulariser
    classification_cls = Classification      # the tree's own --clas head class (the BEV tree's line head differs)

    def _common_init(self, args, M, backbone_cls):
        self.nclasses = args.nclasses
        self.order = args.order
        if self.order < 0 or self.order > self.max_order:
            raise NotImplementedError(
                'Requested order {} for polynomial fit is not implemented'.format(self.order))
        if getattr(args, "no_cuda", False):
            raise RuntimeError("lanefit: no_cuda=True requested but this implementation has no CPU path")
        out_channels = args.nclasses + int(not args.end_to_end)
        self.net = backbone_cls(layers=args.layers, in_channels=args.channels_in, out_channels=out_channels,
                                pretrained=args.pretrained, pool=args.pool)
        self.activation_name = args.activation_layer
        self.activation = activation_layer(args.activation_layer)
        resize = args.resize
        self.resize = resize
        self.zero_rows = ceil(resize * args.mask_percentage)          # LSQ_layer.py:257
        self.reg_ls = float(args.reg_ls)
        self.use_cholesky = bool(args.use_cholesky)
        self.end_to_end = args.end_to_end
        self.pretrained = args.pretrained
        self.classification_branch = bool(getattr(args, "clas", False))
        if self.classification_branch:
            # LSQ_layer.py:247-254 (BP) / :270-277 (BEV): both heads read the (32, 64) encoder output
            cls = self.classification_cls
            self.line_classification = cls('line', size=(32, 64), channels_in=128, resize=resize).cuda()
            self.horizon_estimation = cls('horizon', size=(32, 64), channels_in=128, resize=resize).cuda()
        self.net.export_encoder_output = self.classification_branch
        # precision mode of the backbone (erfnet.Net.precision): "fp32" unless args.precision says otherwise
        self.net.precision = getattr(args, "precision", "fp32")
        if self.net.precision == "bf16" and self.classification_branch:
            raise NotImplementedError("the --clas heads read an fp32 encoder output: use precision 'fp32' or 'fp32x9'")
        self.check_singular = True      # False: skip the per-step D2H status read; inspect self.last_status
        self.return_masked = True
        self.last_status = None
        # constant (H*W,2) grid, computed once on the host with the reference's fp32 ops
        self._grid_cpu = geometry.projective_grid(resize, 2 * resize, M, self.normalised)
        self._grid = None

    def grid_on(self, device):
        if self._grid is None or self._grid.device != device:
            self._grid = self._grid_cpu.to(device)
        return self._grid

    def _heads(self, shared_encoder, end_to_end):
        if end_to_end and self.classification_branch:
            return self.line_classification(shared_encoder), self.horizon_estimation(shared_encoder)
        return None, None

    def _seg_maps(self, output, gt_line=None):
        """Non-end-to-end path: arg-max of the segmentation logits -> per-lane maps valued k at class k (LSQ_layer.py:302-308;
        BP :279-293), the masked rows zeroed (``index_fill``) and, in the BP tree, "Prevent singular matrix" (BP :308-311): lanes
        flagged in ``gt_line`` borrow map [0, 0].  One launch (``lf_seg_maps``), detached, no host read of ``gt_line.sum()``."""
        lanes = 2 if self.nclasses < 3 else 4
        return ops.seg_maps(output, gt_line, self.zero_rows, lanes)

    def _fit(self, output, end_to_end, gt_line=None):
        grid = self.grid_on(output.device)
        reg = 0.0 if (self.use_cholesky and self.cholesky_drops_reg) else self.reg_ls
        if end_to_end:
            beta, masked, status = fit.fit_lanes(output, grid, self.zero_rows, self.order, reg, self.y_offset,
                                                 self.activation_name, self.use_cholesky, self.return_masked,
                                                 self.check_singular)
        else:
            maps = self._seg_maps(output, gt_line)
            # (the masked rows are zeros already; handing zero_rows to the fit keeps its kernels from reading them at all --
            # at 320 x 640 the BP grid has a pole on a masked row, and 0 * inf is NaN)
            beta, _, status = fit.fit_lanes(maps, grid, self.zero_rows, self.order, reg, self.y_offset, "none",
                                            self.use_cholesky, False, self.check_singular)
            masked = maps
        self.last_status = status
        return fit.split_lanes(beta, self.nclasses, self.beta_dtype), masked


class BEVNet(_LaneFitNet):
    """``Net(args)``; ``forward(input, end_to_end) ->
    (beta0, beta1, beta2, beta3, masked, M, output, line, horizon)`` (BEV/Networks/LSQ_layer.py:290-326);
    with ``--clas`` the line output is (N, 3, 4) (four 3-way heads, ``ClassificationBEV``)."""
    classification_cls = ClassificationBEV

    def __init__(self, args):
        super().__init__()
        M, _ = geometry.bev_homography()
        self._common_init(args, M, erfnet.Net)
        self.M = torch.from_numpy(M).unsqueeze(0).expand(args.batch_size, 3, 3).float().cuda()

    def forward(self, input, end_to_end):
        shared_encoder, output = self.net(input, end_to_end * self.pretrained)
        line, horizon = self._heads(shared_encoder, end_to_end)
        (b0, b1, b2, b3), masked = self._fit(output, end_to_end)
        return b0, b1, b2, b3, masked, self.M, output, line, horizon


class _BPBackbone(erfnet.Net):
    three_outputs = True


class BPNet(_LaneFitNet):
    """``Net(args)``; ``forward(input, gt_line, end_to_end, early_return=False, gt=None) ->
    (beta0..3, masked, output, line, horizon, output_seg)`` or bare ``output``
    (BP/Networks/LSQ_layer.py:269-315).  Pixel coordinates, ``255 - y``, orders 0..3, fp64 betas."""
    normalised = False
    y_offset = 255.0
    beta_dtype = torch.float64
    max_order = 3
    cholesky_drops_reg = True

    def __init__(self, args):
        super().__init__()
        M, _ = geometry.get_homography(args.resize, getattr(args, "no_mapping", False))
        self._common_init(args, M, _BPBackbone)
        self.net.export_encoder_output = True        # output_seg IS the encoder output in this tree (BP/Networks/ERFNet.py:143-163)

    def forward(self, input, gt_line, end_to_end, early_return=False, gt=None):
        shared_encoder, output, output_seg = self.net(input, end_to_end * self.pretrained)
        if early_return:
            return output
        line, horizon = self._heads(shared_encoder, end_to_end)
        (b0, b1, b2, b3), masked = self._fit(output, end_to_end, gt_line)
        return b0, b1, b2, b3, masked, output, line, horizon, output_seg
