"""Kernel durations per STEP and per kernel family from a rocprofv3 --kernel-trace run of `bench.py --no-extras` (rocpd sqlite),
written into profiles/kernel_time.json under `key`; bench.py divides the family's algorithmic FLOPs by it
(roofline.frac_by_kernel_durations, next to the HIP-event figure measured live).

    python profiles/summarize_kernel_time.py <results.db> <key> <commit> <digest>

Steps in the run = launches of head_fwd_kernel (one per forward), warm-up steps included: every step launches the same kernels.
"""
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_traffic_step import FAMILIES, short      # noqa: E402  (the same family patterns as the traffic summary)


def main(db, key, commit, digest):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    fam = {name: [0, 0] for name, _ in FAMILIES}
    fam["other (torch glue)"] = [0, 0]
    steps = 0
    for n, s, e in cur.execute("select %s, start, end from kernels" % name_col):
        k = short(n)
        steps += k == "head_fwd_kernel"
        f = next((fn for fn, pat in FAMILIES if re.search(pat, k)), "other (torch glue)")
        fam[f][0] += 1
        fam[f][1] += e - s
    assert steps > 0, "no head_fwd_kernel launch in the trace"
    out = {"commit": commit, "sources_digest": digest, "steps_in_run": steps,
           "families": {f: {"ms_per_step": round(v[1] / steps / 1e6, 4), "launches_per_step": round(v[0] / steps, 1)} for f, v in fam.items()},
           "kernel_ms_per_step": round(sum(v[1] for v in fam.values()) / steps / 1e6, 4)}
    path = os.path.join(root, "profiles", "kernel_time.json")
    try:
        allk = json.load(open(path))
    except (OSError, ValueError):
        allk = {}
    allk[key] = out
    json.dump(allk, open(path, "w"), indent=1)
    print("%s: %d steps, %.3f ms of kernels per step (%s)" % (key, steps, out["kernel_ms_per_step"],
                                                             ", ".join("%s %.2f" % (f.split(" (")[0], v["ms_per_step"]) for f, v in out["families"].items())))


if __name__ == "__main__":
    main(*sys.argv[1:5])
