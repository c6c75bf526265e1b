#!/usr/bin/env python
"""Occupancy of the step loop's timeline from a rocprofv3 kernel trace (rocpd sqlite): how much of the wall time has 0 / 1 / >= 2 kernels in
flight, and the gap between consecutive kernels of the same queue.  Looks at the steady state: the window between the first and the last
launch of the dominant GEMM symbol in the second half of the trace.

    python tools/rocpd_timeline.py /tmp/prof/x_results.db
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("queue_id", "stream_id", "queue") if c in cols), None)
    sel = f"name, start, end, {qcol}" if qcol else "name, start, end, 0"
    rows = cur.execute(f"select {sel} from kernels order by start").fetchall()
    print(f"# {path}: {len(rows)} dispatches; columns of `kernels`: {cols}")
    gem = [r for r in rows if "gemm_bf16_kernel<3" in r[0] or "gemm_pp_kernel<3" in r[0]]
    if len(gem) < 100:
        print("too few gate+residual GEMM launches")
        return
    half = gem[len(gem) // 2:]
    t0, t1 = half[0][1], half[-1][2]
    win = [r for r in rows if r[1] >= t0 and r[2] <= t1]
    ev = []
    for n, s, e, q in win:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    busy = {0: 0, 1: 0, 2: 0}
    depth, last = 0, t0
    for t, d in ev:
        busy[min(depth, 2)] += t - last
        last = t
        depth += d
    tot = t1 - t0
    ksum = sum(e - s for _, s, e, _ in win)
    print(f"window {tot / 1e6:.3f} ms, {len(win)} kernels, sum of kernel durations {ksum / 1e6:.3f} ms ({ksum / tot:.2f} x the window)")
    print(f"  no kernel running {100 * busy[0] / tot:5.1f} %   one kernel {100 * busy[1] / tot:5.1f} %   two or more {100 * busy[2] / tot:5.1f} %")
    byq = {}
    for n, s, e, q in win:
        byq.setdefault(q, []).append((s, e, n))
    for q, ks in sorted(byq.items(), key=lambda kv: -len(kv[1]))[:4]:
        gaps = sorted((ks[i + 1][0] - ks[i][1]) / 1e3 for i in range(len(ks) - 1))
        if gaps:
            print(f"  queue {q}: {len(ks)} kernels, gap end->next start: median {gaps[len(gaps) // 2]:.2f} us, mean {sum(gaps) / len(gaps):.2f}, p90 {gaps[int(0.9 * len(gaps))]:.2f}, "
                  f"sum {sum(gaps) / 1e3:.3f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
