#!/usr/bin/env python
"""Where one ODE step's time goes OUTSIDE the DiT blocks, from a rocprofv3 --kernel-trace rocpd database of bench.py.

    python tools/rocpd_step_gaps.py <results.db> [step index from the end, default 3]

A step ends with `cfg_euler_kernel`.  For one step of the timed region: the step period (cfg_euler end -> next cfg_euler end), the span of the 22
blocks (first fused QK+V launch -> last gate+residual GEMM), and every launch before the first / after the last block launch with its start, duration
and the gap in front of it -- the per-step prologue (step counter, K = 100 input projection, position convolution, first LayerNorm) and epilogue
(final LayerNorm, output projection, CFG + Euler), and the hand-over between two graph launches."""
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*\)$", "", name.replace("(anonymous namespace)::", "").replace("void ", ""))[:60]


def main(path, back):
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = [(short(n), s, e) for n, s, e in cur.execute(f"select {name_col}, start, end from kernels order by start")]
    ends = [i for i, r in enumerate(rows) if r[0].startswith("cfg_euler_kernel")]
    if len(ends) < back + 2:
        sys.exit("not enough steps in the trace")
    periods = [(rows[ends[i + 1]][2] - rows[ends[i]][2]) / 1e3 for i in range(len(ends) - 1)]
    body = sorted(p for p in periods if p < 3 * sorted(periods)[len(periods) // 2])
    print(f"# {len(ends)} steps in the trace; step period (cfg_euler end -> next cfg_euler end): median {body[len(body) // 2]:.1f} us, min {body[0]:.1f}")
    i0, i1 = ends[-back - 2], ends[-back - 1]
    step = rows[i0 + 1: i1 + 1]
    tprev = rows[i0][2]
    is_block = lambda n: n.startswith("gemm_qkv_fused") or n.startswith("attn_fwd") or (n.startswith("gemm_bf16_kernel<") and not n.startswith("gemm_bf16_kernel<2"))
    blk = [j for j, r in enumerate(step) if is_block(r[0])]
    first, last = blk[0], blk[-1]
    span = (step[last][2] - step[first][1]) / 1e3
    period = (step[-1][2] - tprev) / 1e3
    print(f"# step {len(ends) - back - 1}: period {period:.1f} us, blocks {span:.1f} us ({100 * span / period:.1f} %), outside {period - span:.1f} us; {len(step)} launches")
    print("#   start_us   dur_us  gap_before_us  kernel   (t = 0: the previous step's cfg_euler end)")
    prev_end = tprev
    for j, (n, s, e) in enumerate(step):
        if first < j <= last and not (j == first + 1):
            prev_end = max(prev_end, e)
            continue
        tag = "   <- first block launch" if j == first else "   <- the launch after it" if j == first + 1 else ""
        print(f"{(s - tprev) / 1e3:11.1f} {(e - s) / 1e3:8.2f} {(s - prev_end) / 1e3:10.2f}     {n}{tag}")
        if j == first:
            print(f"        ...  {last - first - 1} block launches ...")
        prev_end = max(prev_end, e) if j != first else e
    ksum = sum((e - s) for j, (n, s, e) in enumerate(step) if j < first or j > last) / 1e3
    print(f"# kernel time outside the blocks {ksum:.1f} us of the {period - span:.1f} us outside")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 3)
