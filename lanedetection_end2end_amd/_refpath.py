"""Optional: let the mirrored ``Networks`` packages fall through to the reference tree for the
host-side plumbing modules this project does not replace (``Networks.utils``: argparse flags,
optimisers, logging -- SURVEY.md section 2, row 10).  Set LANEFIT_REFERENCE_ROOT to the directory
holding ``Birds_Eye_View_Loss/`` and ``Backprojection_Loss/``; see INTEGRATION.md."""
import os


def extend(pkg_path, tree):
    root = os.environ.get("LANEFIT_REFERENCE_ROOT")
    if root:
        cand = os.path.join(root, tree, "Networks")
        if os.path.isdir(cand) and cand not in pkg_path:
            pkg_path.append(cand)
