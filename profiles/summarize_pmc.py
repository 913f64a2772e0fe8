"""Summarise a rocprofv3 --pmc ... --kernel-trace run (rocpd sqlite output): mean counter values per kernel, and the
matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) when both were collected.

    python profiles/summarize_pmc.py gpurun_out/pmc_x/p_results.db [min_calls]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main(path, min_calls=3):
    cur = sqlite3.connect(path).cursor()
    agg = {}
    for name, gx, counter, value, dur in cur.execute(
            "select kernel_name, grid_size_x, counter_name, value, duration from counters_collection"):
        a = agg.setdefault((short(name), gx), {}).setdefault(counter, [0, 0.0, 0.0])
        a[0] += 1; a[1] += value; a[2] += dur
    counters = sorted({c for v in agg.values() for c in v})
    print("# %s" % path)
    print("%-72s %9s %5s %9s " % ("kernel", "grid_x", "n", "avg_us") + " ".join("%22s" % c[:22] for c in counters) + "  MFMA busy %")
    for (k, gx), v in sorted(agg.items(), key=lambda kv: -max(a[2] for a in kv[1].values())):
        n = max(a[0] for a in v.values())
        if n < min_calls or "tap" not in k:
            continue
        us = max(a[2] / a[0] for a in v.values()) / 1e3
        row = "%-72s %9d %5d %9.1f " % (k, gx, n, us) + " ".join("%22.4g" % (v[c][1] / v[c][0]) if c in v else "%22s" % "-" for c in counters)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            mf = v["SQ_VALU_MFMA_BUSY_CYCLES"]; gu = v["GRBM_GUI_ACTIVE"]
            row += "  %.1f" % (100.0 * (mf[1] / mf[0]) / ((gu[1] / gu[0]) / 8.0 * 1024.0))
        print(row)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
