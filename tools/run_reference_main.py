#!/usr/bin/env python
"""Run the reference's own main.py (unmodified) on top of the MI355X hot path.

    LANEFIT_REFERENCE_ROOT=/path/to/LaneDetection_End2End python tools/run_reference_main.py {bev|bp} [main.py args…]

Only sys.path is arranged: the mirrored tree (lanedetection_end2end_amd/bev or /bp, which provides Networks/
and Loss_crit.py) goes first, the reference tree second (Dataloader/, eval_lane.py, and -- through the mirrored
package's __path__ extension -- Networks/utils.py).  See INTEGRATION.md.
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) < 2 or sys.argv[1] not in ("bev", "bp"):
        raise SystemExit(__doc__)
    tree = sys.argv[1]
    root = os.environ.get("LANEFIT_REFERENCE_ROOT")
    if not root:
        raise SystemExit("set LANEFIT_REFERENCE_ROOT to the reference checkout")
    ref_dir = os.path.join(root, {"bev": "Birds_Eye_View_Loss", "bp": "Backprojection_Loss"}[tree])
    sys.path.insert(0, HERE)
    import lanedetection_end2end_amd  # noqa: F401  (loads the library early: fail loudly if it is not built)
    sys.path.insert(0, ref_dir)
    sys.path.insert(0, os.path.join(HERE, "lanedetection_end2end_amd", tree))
    sys.argv = [os.path.join(ref_dir, "main.py")] + sys.argv[2:]
    runpy.run_path(sys.argv[0], run_name="__main__")


if __name__ == "__main__":
    main()
