"""FETCH_SIZE / WRITE_SIZE passes of profiles/collect.sh -> profiles/<tag>_pmc_hbm_conv128.txt + profiles/traffic.json.

    python profiles/summarize_traffic.py gpurun_out/prof_r2 r2 <commit>

Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane)
coalesced streaming reads -> doubled; WRITE_SIZE as reported; values arrive in KB.  Infinity-Cache hits are counted as fetches.
"""
import glob
import json
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*$", "", name)[:60]


def collect(db):
    cur = sqlite3.connect(db).cursor()
    agg = {}
    for name, counter, value in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        a = agg.setdefault((short(name), counter), [])
        a.append(value)
    return agg


def main(out_dir, tag, commit, digest=None):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = ["# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- python tools/kbench.py --variants 2 --iters 3 --one 128 32 64 1 16",
             "# one 1x3 dil-16 conv, C=128, 32x64, batch 32: x, y, gx, gy are 33.5 MB each (NHWC fp32); values in KB as reported",
             "# (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads); commit " + commit]
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        dbs = glob.glob(os.path.join(out_dir, "pmc_" + c, "**", "*_results.db"), recursive=True)
        if not dbs:
            continue
        for (k, counter), vals in sorted(collect(dbs[0]).items()):
            if counter != c:
                continue
            mean = sum(vals) / len(vals)
            lines.append("%-45s %s n=%2d mean=%10.1f KB  min=%10.1f max=%10.1f" % (k, c, len(vals), mean, min(vals), max(vals)))
            tot[(k, c)] = mean
    open(os.path.join(root, "profiles", "%s_pmc_hbm_conv128.txt" % tag), "w").write("\n".join(lines) + "\n")

    def bytes_of(prefix):
        f = [v for (k, c), v in tot.items() if prefix in k and c == "FETCH_SIZE"]
        w = [v for (k, c), v in tot.items() if prefix in k and c == "WRITE_SIZE"]
        if not f or not w:
            return None
        return {"bytes_per_launch": 2 * max(f) * 1e3 + max(w) * 1e3, "fetch_kb_reported": max(f), "write_kb_reported": max(w)}
    out = {"commit": commit, "sources_digest": digest, "source": "profiles/%s_pmc_hbm_conv128.txt" % tag,
           "launch": "128-channel 3-tap conv, 32x64, batch 32 (tools/kbench.py --one 128 32 64 1 16)",
           "algorithmic_bytes_per_launch": 2 * 32 * 32 * 64 * 128 * 4,
           "tapgemm": bytes_of("tapgemm_kernel<"), "tapwgrad": bytes_of("tapwgrad_kernel<")}
    json.dump(out, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "unknown", sys.argv[4] if len(sys.argv) > 4 else None)
