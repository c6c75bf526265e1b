#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 'rocpd' sqlite) kernel trace into a --stats style table.

    python tools/rocpd_summary.py gpurun_out/prof_x/x_results.db > profiles/r01/r01_x_kernel_stats.txt
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:110]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        d = (e - s) / 1e3
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f"# source: {path}")
    print(f"# {len(rows)} kernel dispatches, {tot / 1e3:.3f} ms total kernel time")
    print(f"{'kernel':112s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:112s} {a[0]:7d} {a[1]:12.1f} {a[1] / a[0]:10.2f} {a[2]:10.2f} {a[3]:10.2f} {100 * a[1] / tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
