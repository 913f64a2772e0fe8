"""lanedetection_end2end_amd -- MI355X-native hot path of LaneDetection_End2End.

ERFNet backbone -> differentiable weighted-least-squares lane fit -> area / back-projection /
segmentation losses, forward and backward, as hand-written HIP kernels for gfx950 behind the C ABI
of include/lanefit.h.  ``bev`` and ``bp`` mirror the reference's two source trees module by module
so that its own main.py can import them unchanged.
"""
from . import _lib  # noqa: F401

__all__ = ["bev", "bp", "fit", "losses", "erfnet", "lsq", "geometry", "ops"]
