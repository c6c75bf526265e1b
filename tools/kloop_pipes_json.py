#!/usr/bin/env python
"""K-loop pipe budget of the lock-step 128 x 128 GEMM tile, DERIVED from the raw lines of a committed ablation run (one -DLEMAS_ABLATE build per
variant: bits 1 = no MFMAs, 2 = no refill LDS-DMA, 4 = no fragment reads) -> the JSON bench.py attaches as `roofline.k_loop_pipes`.

    python tools/kloop_pipes_json.py profiles/r04/r04g_kloop_ablations_128x128.txt > profiles/r04/r04g_kloop_pipes.json

Per K-tile time of a variant = (loop stamp at K = 2048 - loop stamp at K = 1024) / 16 K-tiles, tile 17 (8 waves); what is outside the loop comes
from the K = 1024 launch of the unablated build: prologue + loop + epilogue stamps against kbench's launch time."""
import json
import re
import sys


def main(path):
    rows, pending = {}, None
    for line in open(path):
        m = re.match(r"ablate=(\d+)\s+phases gemm_gate M=(\d+) N=(\d+) K=(\d+) grid=(\d+) prologue ([\d.]+) loop ([\d.]+) epilogue ([\d.]+)", line)
        if m:
            pending = dict(abl=int(m.group(1)), K=int(m.group(4)), grid=int(m.group(5)), loop=float(m.group(7)), epilogue=float(m.group(8)))
            continue
        m = re.match(r"ablate=(\d+)\s+gemm_gate M=(\d+) N=(\d+) K=(\d+) tile=(\d+): ([\d.]+) us", line)
        if m and pending and pending["abl"] == int(m.group(1)) and pending["K"] == int(m.group(4)):
            rows[(int(m.group(5)), pending["abl"], pending["K"])] = dict(pending, launch=float(m.group(6)))
            pending = None
    out = {"_source": f"{path} (raw `ablate=` lines; tools/kloop_pipes_json.py)", "tiles": {}}
    names = {0: "full_loop", 2: "mfma_plus_fragment_reads", 4: "mfma_plus_lds_dma", 3: "lds_fragment_reads_alone", 7: "barrier_and_waits_alone"}
    for tile in sorted({k[0] for k in rows}):
        per = {}
        for abl, name in names.items():
            a, b = rows.get((tile, abl, 1024)), rows.get((tile, abl, 2048))
            if a and b:
                per[name] = round((b["loop"] - a["loop"]) / 16.0, 4)
        base = rows.get((tile, 0, 1024))
        e = {"us_per_k_tile": per}
        if base:
            # the loop stamp includes the prologue (the stamps' "prologue" column is 0 in these builds): prologue = loop stamp - 16 K-tiles
            pro = base["loop"] - 16 * per.get("full_loop", 0.0)
            e["outside_the_loop_us"] = {"prologue": round(pro, 2), "epilogue": base["epilogue"],
                                        "launch_ramp_and_drain": round(base["launch"] - base["loop"] - base["epilogue"], 2), "launch": base["launch"]}
        out["tiles"][str(tile)] = e
    # ideal figures (not measured): one K-tile of 128 x 128 x 64 is 512 MFMA clocks per SIMD and 32 KB through the CU's 64 B/clk vector-memory path
    clk_ghz = 2.13
    out["ideal_us_per_k_tile"] = {"mfma_alone": round(512 / clk_ghz / 1e3, 3), "l2_to_lds_64B_per_clk": round(32768 / 64 / clk_ghz / 1e3, 3), "clock_ghz": clk_ghz}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
