#!/usr/bin/env python
"""In-situ timeline of one ODE step (what rocprofv3 cannot show: its kernel trace serialises the two CFG lanes).

A MEASUREMENT build of the library (-DLEMAS_PHASE_TIMESTAMPS) makes every block GEMM and attention launch stamp, per workgroup, the
100 MHz wall clock at entry and after its last store; the captured hipGraph replays those launches every step, so after one utterance
the buffer holds the LAST step's stamps.  Prints, for one DiT block in the middle of the stack, when each launch of either lane started
and ended, and over all blocks the in-situ duration per kernel, the gaps between consecutive launches of a lane and how much of the
step has 0 / 1 / 2 of the stamped launches in flight.

    /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_session.sh timeline:configs1,configs3'      (builds the measurement library, runs this, rebuilds the product)
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench as Bn  # noqa: E402
from lemas_tts_amd import _lib, synth  # noqa: E402
from lemas_tts_amd.model.cfm import CFM  # noqa: E402
from lemas_tts_amd.model.layout import DiTArch  # noqa: E402

NAMES = ["qkv", "attn", "out", "ff1", "ff2"]                 # launches per lane and block when QK+V is one launch
NAMES_UNFUSED = ["qk", "v", "attn", "out", "ff1", "ff2"]       # batched shapes: QK and V separately (engine_dit.hip: qkv_wgs > 250)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="configs1")
    ap.add_argument("--opt", nargs="*", default=[], help="engine options key=value (e.g. ln_fold=1)")
    ap.add_argument("--json", default=None, help="also write the per-block figures as JSON (what bench.py's roofline.latency_batched reads)")
    a = ap.parse_args()
    global NAMES
    w = dict(Bn.WORKLOADS[a.workload])
    B, F, N = w["B"], w["F"], w["N"]
    dev = torch.device("cuda:0")
    arch = DiTArch()
    sd = synth.synth_cfm_state_dict(arch, Bn.VOCAB, 1234)
    cond, text, y0, _ = Bn.build_inputs(w, 1, dev)
    L = _lib.testlib()
    pitch = (N + 127) // 128 * 128
    if ((B * pitch + 255) // 256) * (3 * 1024 // 128) > 250:
        NAMES = NAMES_UNFUSED
    slots = arch.depth * 2 * len(NAMES)
    buf = torch.zeros((slots, 4096), dtype=torch.int64, device=dev)
    m = CFM(arch, Bn.VOCAB, sd, device=dev)
    m.engine.set_option("table_cache", 0)
    for kv in a.opt:
        k, v = kv.split("=")
        m.engine.set_option(k, int(v))

    def run():
        out, _ = m.sample(cond, text, N, steps=Bn.NFE, cfg_strength=Bn.CFG, sway_sampling_coef=Bn.SWAY, y0=y0, use_acc_grl=False)
        torch.cuda.synchronize()

    run()                                      # untimed: tables, allocations
    rc = L.lemas_k_timeline(C.c_void_p(buf.data_ptr()), slots)
    if rc != 0:
        raise SystemExit("not a measurement build: " + L.lemas_last_error().decode())
    m.engine.set_option("dual", 1)             # drops the cached graphs: the next sample captures launches that carry their slot
    run()
    run()
    L.lemas_k_timeline(None, 0)
    t = buf.cpu().numpy().reshape(slots, 1024, 4).astype(np.float64) * 0.01      # us
    rec = []                                   # (block, lane, name, start, end)
    for s in range(slots):
        live = t[s, :, 0] > 0
        if not live.any():
            continue
        l, r = divmod(s, 2 * len(NAMES))
        ln, k = divmod(r, len(NAMES))
        rec.append((l, ln, NAMES[k], t[s, live, 0].min(), t[s, live, 3].max(), int(live.sum())))
    t0 = min(r[3] for r in rec)
    t1 = max(r[4] for r in rec)
    print(f"# {a.workload}: last ODE step, {len(rec)} stamped launches, first start -> last end {t1 - t0:.1f} us ({(t1 - t0) / arch.depth:.1f} us per block)")
    mid = arch.depth // 2
    print(f"# block {mid}: start / end relative to the block's first launch (us), workgroups")
    b0 = min(r[3] for r in rec if r[0] == mid)
    for r in sorted((r for r in rec if r[0] == mid), key=lambda r: r[3]):
        print(f"   lane {r[1]} {r[2]:5s} {r[3] - b0:7.1f} -> {r[4] - b0:7.1f}  ({r[4] - r[3]:5.1f} us, {r[5]} workgroups)")
    for name in NAMES:
        d = [r[4] - r[3] for r in rec if r[2] == name]
        print(f"# {name:5s} in situ: mean {np.mean(d):5.1f} us  min {np.min(d):5.1f}  max {np.max(d):5.1f}")
    for ln in (0, 1):
        seq = sorted((r for r in rec if r[1] == ln), key=lambda r: r[3])
        gaps = {}
        for x, y in zip(seq, seq[1:]):
            gaps.setdefault(f"{x[2]}->{y[2]}", []).append(y[3] - x[4])
        print(f"# lane {ln} gaps end -> next start (us; qkv<-ff2 and ff1<-out contain an LN-mod launch): " +
              "  ".join(f"{k} {np.mean(v):.1f}" for k, v in gaps.items()))
    ev = sorted([(r[3], 1) for r in rec] + [(r[4], -1) for r in rec])
    busy = [0.0, 0.0, 0.0]
    depth, last = 0, t0
    for tt, dlt in ev:
        busy[min(depth, 2)] += tt - last
        last, depth = tt, depth + dlt
    tot = t1 - t0
    print(f"# of the step: no stamped launch in flight {100 * busy[0] / tot:.1f} %, one {100 * busy[1] / tot:.1f} %, two or more {100 * busy[2] / tot:.1f} %")
    # WHICH launch is alone when exactly one is in flight (and what the other lane is doing then: it sits in a gap between two of ITS launches --
    # the LayerNorm launches, which are not stamped, or a launch boundary)
    alone = {n: 0.0 for n in NAMES}
    other_gap = {}
    spans = sorted(rec, key=lambda r: r[3])
    edges = sorted({r[3] for r in rec} | {r[4] for r in rec})
    lane_seq = {ln: sorted((r for r in rec if r[1] == ln), key=lambda r: r[3]) for ln in (0, 1)}
    for x0, x1 in zip(edges, edges[1:]):
        mid_t = 0.5 * (x0 + x1)
        live = [r for r in spans if r[3] <= mid_t < r[4]]
        if len(live) != 1:
            continue
        r = live[0]
        alone[r[2]] += x1 - x0
        oseq = lane_seq[1 - r[1]]
        prev = [q for q in oseq if q[4] <= mid_t]
        nxt = [q for q in oseq if q[3] > mid_t]
        key = f"{prev[-1][2] if prev else 'start'}->{nxt[0][2] if nxt else 'end'}"
        other_gap[key] = other_gap.get(key, 0.0) + (x1 - x0)
    print("# alone in flight, us per block by launch: " + "  ".join(f"{k} {v / arch.depth:.1f}" for k, v in alone.items()))
    print("# ... while the other lane sits between: " + "  ".join(f"{k} {v / arch.depth:.1f}" for k, v in sorted(other_gap.items(), key=lambda kv: -kv[1])))
    if a.json:
        import json
        per_lane_gaps = {}
        for ln in (0, 1):
            seq = lane_seq[ln]
            for x, y in zip(seq, seq[1:]):
                per_lane_gaps.setdefault(f"{x[2]}->{y[2]}", []).append(y[3] - x[4])
        out = {"_source": f"tools/timeline_step.py --workload {a.workload} (measurement build, -DLEMAS_PHASE_TIMESTAMPS): per-workgroup wall-clock stamps of every "
                          "block GEMM / attention launch of the LAST ODE step, replayed from the captured graph",
               "workload": a.workload, "options": a.opt, "blocks": arch.depth, "lanes": 2, "stamped_launches": len(rec),
               "us_per_block": (t1 - t0) / arch.depth,
               "in_situ_span_us": {n: float(np.mean([r[4] - r[3] for r in rec if r[2] == n])) for n in NAMES},
               "sum_in_situ_spans_us_per_block_per_lane": float(sum(np.mean([r[4] - r[3] for r in rec if r[2] == n]) for n in NAMES)),
               "gaps_us_by_site": {k: float(np.mean(v)) for k, v in per_lane_gaps.items()},
               "sum_gaps_us_per_block_per_lane": float(sum(np.mean(v) for v in per_lane_gaps.values())),
               "share_in_flight": {"none": busy[0] / tot, "one": busy[1] / tot, "two_or_more": busy[2] / tot},
               "alone_in_flight_us_per_block": {k: v / arch.depth for k, v in alone.items()},
               "alone_while_other_lane_between_us_per_block": {k: v / arch.depth for k, v in sorted(other_gap.items(), key=lambda kv: -kv[1])},
               "workgroups": {n: int(np.median([r[5] for r in rec if r[2] == n])) for n in NAMES}}
        with open(a.json, "w") as f:
            json.dump(out, f, indent=1)
        print("wrote", a.json)


if __name__ == "__main__":
    main()
