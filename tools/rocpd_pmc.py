#!/usr/bin/env python
"""Per-kernel PMC averages from a rocprofv3 rocpd sqlite file (sums the per-SE/XCD rows of each dispatch).

    python tools/rocpd_pmc.py gpurun_out/pmc_x/pmc_results.db [kernel-substring ...]
"""
import collections
import re
import sqlite3
import sys


def main(path, filt):
    db = sqlite3.connect(path)
    cur = db.cursor()
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, dispatch) -> counter -> sum
    dur = {}
    for name, disp, cname, val, s, e in cur.execute(
            "select kernel_name, dispatch_id, counter_name, value, start, end from counters_collection"):
        k = re.sub(r"\(.*\)$", "", name.replace("(anonymous namespace)::", ""))[:60]
        if filt and not any(f in k for f in filt):
            continue
        per[(k, disp)][cname] += val
        dur[(k, disp)] = (e - s) / 1e3
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for (k, d), cs in per.items():
        cnt[k] += 1
        agg[k]["dur_us"] += dur[(k, d)]
        for c, v in cs.items():
            agg[k][c] += v
    names = sorted({c for a in agg.values() for c in a})
    print("kernel".ljust(40), "n".rjust(5), *[c[-22:].rjust(23) for c in names])
    for k in sorted(agg, key=lambda k: -agg[k]["dur_us"]):
        print(k[:40].ljust(40), str(cnt[k]).rjust(5), *[f"{agg[k][c] / cnt[k]:23.1f}" for c in names])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
