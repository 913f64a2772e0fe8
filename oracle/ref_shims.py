"""Import the *real* reference modules from /root/reference, unmodified, on CPU.

TEST INFRASTRUCTURE ONLY.  Used by ``oracle/gen_golden.py`` (authoring container) and,
when /root/reference happens to be present, by ``bench.py``'s cpu_baseline leg.  Never
available on the GPU box (no /root/reference there).

Two shims are needed in this image:
1. ``cv2`` is not installed; the hot path uses only ``cv2.getPerspectiveTransform``
   -> inject a stub module backed by ``fit_oracle.get_perspective_transform``.
2. ``Area_Loss`` builds a uint8 mask (BEV/Loss_crit.py:131) that ``torch.masked_select``
   rejects on torch >= 2 -> wrap ``masked_select`` to cast uint8 -> bool.
No reference source is edited or copied.
"""
import importlib
import os
import sys
import types
from argparse import Namespace

import numpy as np
import torch

from . import fit_oracle

REF_ROOT = "/root/reference"
TREES = {"bev": "Birds_Eye_View_Loss", "bp": "Backprojection_Loss"}


def available():
    return os.path.isdir(os.path.join(REF_ROOT, TREES["bev"]))


def _install_cv2_stub():
    if "cv2" in sys.modules and not getattr(sys.modules["cv2"], "_lanefit_stub", False):
        return
    stub = types.ModuleType("cv2")
    stub._lanefit_stub = True
    stub.getPerspectiveTransform = lambda src, dst: fit_oracle.get_perspective_transform(src, dst)
    sys.modules["cv2"] = stub


_orig_masked_select = torch.masked_select


def _masked_select(input, mask, *a, **k):
    if mask.dtype == torch.uint8:
        mask = mask.bool()
    return _orig_masked_select(input, mask, *a, **k)


def load(tree):
    """Return a namespace with the reference's modules for ``tree`` in {'bev','bp'}.

    The two trees use the same top-level package names (``Networks``, ``Loss_crit``), so
    previously imported copies are purged first.
    """
    assert available(), "reference tree not present"
    _install_cv2_stub()
    torch.masked_select = _masked_select
    root = os.path.join(REF_ROOT, TREES[tree])
    for name in [m for m in sys.modules if m == "Networks" or m.startswith("Networks.") or m == "Loss_crit"]:
        del sys.modules[name]
    sys.path[:] = [p for p in sys.path if not p.startswith(REF_ROOT)]
    sys.path.insert(0, root)
    ns = types.SimpleNamespace()
    ns.Networks = importlib.import_module("Networks")
    ns.ERFNet = importlib.import_module("Networks.ERFNet")
    ns.LSQ_layer = importlib.import_module("Networks.LSQ_layer")
    ns.utils = importlib.import_module("Networks.utils")
    ns.Loss_crit = importlib.import_module("Loss_crit")
    if tree == "bp":
        ns.gels = importlib.import_module("Networks.gels")
    return ns


def default_args(tree, **over):
    """The hot-path-relevant subset of ``define_args()`` defaults (BEV/Networks/utils.py:24-92)."""
    a = dict(batch_size=4, nclasses=2, resize=256, end_to_end=True, mod="erfnet", layers=18,
             channels_in=3, pretrained=False, pool=True, activation_layer="square", no_cuda=True,
             order=2, reg_ls=0.0, use_cholesky=False, mask_percentage=0.3, clas=False,
             loss_policy="area", weight_funct="none", weight_seg=30, no_mapping=False)
    if tree == "bp":
        a.update(mask_percentage=0.3, loss_policy="backproject")
    a.update(over)
    return Namespace(**a)


def disable_dropout(model):
    """p -> 0 makes ``non_bottleneck_1d.forward`` skip dropout (ERFNet.py:57)."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0
    return model


def to_numpy(t):
    return None if t is None else t.detach().cpu().numpy()
