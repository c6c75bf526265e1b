#!/usr/bin/env python
"""PMC text of tools/gpu_session.sh traffic:<workload> (FETCH_SIZE / WRITE_SIZE sections per workload, made by tools/rocpd_pmc.py) -> the JSON bench.py reads
for roofline.traffic, keyed by workload and by the kernel symbol names bench.py uses.

    python tools/traffic_json.py profiles/r02/r02_pmc_traffic.txt > profiles/r02/r02_traffic.json

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half the bytes of wide coalesced reads -> doubled here;
WRITE_SIZE is taken as reported.  Values are per-launch averages over every launch of the symbol in the profiled run."""
import json
import re
import sys

EPI = {"0": "EPI_BIAS_BF16", "1": "EPI_BIAS_GELU", "2": "EPI_BIAS_F32", "3": "EPI_GATE_RES", "4": "EPI_QK_ROPE", "5": "EPI_V_T"}


def symbol(k: str):
    m = re.search(r"gemm_(?:bf16|pp2?)_kernel<(\d)", k)
    if m:
        return f"gemm_bf16_kernel<{EPI.get(m.group(1), m.group(1))}>"
    for s in ("gemm_qkv_fused_kernel", "attn_fwd_splitkv_kernel", "ln_mod_kernel", "convpos_kernel", "gemm_f32_kernel"):
        if s in k:
            return s
    return None


def main(path):
    out, wl, ctr = {"_source": path + " (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py; FETCH_SIZE doubled per the "
                    "gfx950 correction; per-launch averages by kernel symbol)"}, None, None
    acc = {}
    for line in open(path):
        if line.startswith("### workload"):
            wl = line.split()[2]
        elif line.startswith("## "):
            ctr = line.split()[1]
        elif wl and ctr and not line.startswith(("#", "kernel")) and line.strip():
            m = re.match(r"(.{40})\s+(\d+)\s+([\d.]+)\s+([\d.]+)", line)
            if not m:
                continue
            sym = symbol(m.group(1))
            if sym is None:
                continue
            n, kb = int(m.group(2)), float(m.group(3))
            a = acc.setdefault((wl, sym, ctr), [0, 0.0])
            a[0] += n
            a[1] += n * kb
    for (wl, sym, ctr), (n, kbsum) in acc.items():
        e = out.setdefault(wl, {}).setdefault(sym, {})
        e["fetch_kb" if ctr == "FETCH_SIZE" else "write_kb"] = round(kbsum / n, 1)
        e["launches_" + ("fetch" if ctr == "FETCH_SIZE" else "write") + "_pass"] = n
    for wl, syms in out.items():
        if wl.startswith("_"):
            continue
        for sym, e in syms.items():
            if "fetch_kb" in e and "write_kb" in e:
                e["hbm_bytes_per_launch"] = int(2 * e["fetch_kb"] * 1024 + e["write_kb"] * 1024)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
