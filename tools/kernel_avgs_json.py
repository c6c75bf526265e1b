#!/usr/bin/env python
"""rocprofv3 kernel-stats text (tools/rocpd_summary.py) of a bench.py run -> the JSON bench.py reads for `avg_launch_us_rocprof`,
keyed by workload and by the kernel symbol names bench.py uses (out-proj and FF2 are ONE instantiation).

    python tools/kernel_avgs_json.py configs1=profiles/r03/r03_kernel_stats.txt [configs3=...] > profiles/r03/r03_kernel_avgs.json
"""
import json
import re
import sys

from traffic_json import symbol


def parse(path):
    acc = {}
    for line in open(path):
        m = re.match(r"(.{112})\s+(\d+)\s+([\d.]+)\s+([\d.]+)", line)
        if not m or line.startswith(("#", "kernel")):
            continue
        sym = symbol(m.group(1))
        if sym is None and "attn_fwd" in m.group(1):
            sym = "attn_fwd_splitkv_kernel"
        if sym is None:
            continue
        a = acc.setdefault(sym, [0, 0.0])
        a[0] += int(m.group(2))
        a[1] += float(m.group(3))
    return {k: {"launches": n, "avg_us": round(t / n, 3)} for k, (n, t) in acc.items()}


def main(args):
    out = {"_source": "rocprofv3 --kernel-trace --stats over bench.py (tools/gpu_session.sh rocprof:<workload>): " + ", ".join(a.split("=")[1] for a in args)}
    for a in args:
        wl, path = a.split("=")
        out[wl] = parse(path)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    main(sys.argv[1:])
