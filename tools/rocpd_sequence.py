#!/usr/bin/env python
"""Per-dispatch durations of the LAST `n` kernel dispatches of a rocprofv3 rocpd sqlite trace, in launch order (one forward of a
launch-chain workload: which level of the network each launch belongs to is its position in the chain).

    python tools/rocpd_sequence.py <results.db> <n>
"""
import re
import sqlite3
import sys


def main(path, n):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()[-n:]
    t0 = rows[0][1]
    print(f"# {len(rows)} dispatches, span {(rows[-1][2] - t0) / 1e3:.1f} us, kernel time {sum(e - s for _, s, e in rows) / 1e3:.1f} us")
    for name, s, e in rows:
        k = re.sub(r"\(.*\)$", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.2f}  {k[:70]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
