"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) into a per-kernel table.

    python profiles/summarize_rocpd.py gpurun_out/prof_r1/bench_results.db > profiles/r1_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for n, s, e in rows:
        k = short(n)
        a = agg.setdefault(k, [0, 0, 10**18, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    span = max(r[2] for r in rows) - min(r[1] for r in rows)
    print("# %s: %d dispatches, kernel time %.3f ms, first-to-last span %.3f ms" % (path, len(rows), total / 1e6, span / 1e6))
    print("%-92s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-92s %8d %12.1f %10.2f %10.2f %10.2f %6.2f%%" % (k, a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3,
                                                               100.0 * a[1] / total))


if __name__ == "__main__":
    main(sys.argv[1])
