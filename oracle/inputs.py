"""Re-export of the seeded synthetic inputs (``synthetic_inputs.py`` at the repository root) under their historical name:
the golden generators and tests say ``from oracle import inputs``.  bench.py's measured path imports ``synthetic_inputs``
directly -- nothing from this package."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synthetic_inputs import bev_gt_params, bp_targets, images, lane_like_logits, seg_targets  # noqa: E402,F401
