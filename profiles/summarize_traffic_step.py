"""Whole-step HBM-side traffic from counters: two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) over
`bench.py --no-extras --steps 2 ...` -> bytes per STEP per kernel family, written into profiles/traffic_step.json under `key`.

    python profiles/summarize_traffic_step.py <fetch_results.db> <write_results.db> <key> <commit> <digest>

Corrections exactly as MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE reports half the bytes of wide
(16 B / lane) coalesced streaming reads -> doubled (every kernel of the step reads 16 bytes per lane, LDS-DMA included); WRITE_SIZE as
reported; both arrive in KB.  Infinity-Cache hits are counted as fetches (fabric side of the L2s), so the figure is an UPPER bound
on HBM reads.  Steps in the run = launches of head_fwd_kernel (one per forward).
"""
import json
import os
import re
import sqlite3
import sys

FAMILIES = (("conv forward + data gradient (tap-GEMM kernels)", r"^tapgemm"), ("weight gradient (+ its reductions)", r"^tapwgrad|^wgrad_reduce|^rows_reduce|stem_wgrad|head_wgrad"),
            ("BatchNorm / element-wise passes", r"bn_|pool_|stem_fwd|head_fwd|head_bwd|pack_weights"), ("fit + losses", r"wls_|area_loss|backproj|ce_|mse_|seg_maps|trapezoid"))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z_0-9]+?)I", name)       # mangled template instantiations
    return m.group(1) if m else re.sub(r"[<(].*$", "", name)


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    agg = {}
    for name, cn, value in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        if cn != counter:
            continue
        a = agg.setdefault(short(name), [0, 0.0])
        a[0] += 1
        a[1] += value
    return agg


def main(fetch_db, write_db, key, commit, digest):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    steps = f.get("head_fwd_kernel", [0])[0]
    assert steps > 0 and steps == w.get("head_fwd_kernel", [0])[0], "the two passes must cover the same steps"
    fam = {name: dict(fetch=0.0, write=0.0, launches=0) for name, _ in FAMILIES}
    fam["other (torch glue)"] = dict(fetch=0.0, write=0.0, launches=0)
    kern = {}
    for k in sorted(set(f) | set(w)):
        name = next((n for n, pat in FAMILIES if re.search(pat, k)), "other (torch glue)")
        fb, wb = 2e3 * f.get(k, [0, 0.0])[1], 1e3 * w.get(k, [0, 0.0])[1]
        fam[name]["fetch"] += fb; fam[name]["write"] += wb; fam[name]["launches"] += f.get(k, [0])[0]
        kern[k] = {"launches_per_step": round(f.get(k, [0])[0] / steps, 2), "fetch_MB_per_step": round(fb / steps / 1e6, 2), "write_MB_per_step": round(wb / steps / 1e6, 2)}
    out = {"commit": commit, "sources_digest": digest, "steps_in_run": steps,
           "correction": "FETCH_SIZE x 2 (gfx950: half of wide coalesced reads reported), WRITE_SIZE as reported; KB -> bytes; Infinity-Cache hits count as fetches",
           "families": {n: {"bytes_per_step": int((v["fetch"] + v["write"]) / steps), "fetch_bytes_per_step": int(v["fetch"] / steps),
                            "write_bytes_per_step": int(v["write"] / steps), "launches_per_step": round(v["launches"] / steps, 1)} for n, v in fam.items()},
           "total_bytes_per_step": int(sum(v["fetch"] + v["write"] for v in fam.values()) / steps),
           "kernels": dict(sorted(kern.items(), key=lambda kv: -(kv[1]["fetch_MB_per_step"] + kv[1]["write_MB_per_step"]))[:40])}
    path = os.path.join(root, "profiles", "traffic_step.json")
    try:
        allk = json.load(open(path))
    except (OSError, ValueError):
        allk = {}
    allk[key] = out
    json.dump(allk, open(path, "w"), indent=1)
    print("%s: %d steps, %.1f MB per step (%s)" % (key, steps, out["total_bytes_per_step"] / 1e6,
                                                     ", ".join("%s %.0f MB" % (n.split(" (")[0], v["bytes_per_step"] / 1e6) for n, v in out["families"].items())))


if __name__ == "__main__":
    main(*sys.argv[1:6])
