#!/usr/bin/env python
"""Counter-backed bound per big launch: the PMC group passes of tools/gpu_session.sh pmc:<workload> (rocprofv3 --pmc in separate passes over bench.py, summed
per dispatch by tools/rocpd_pmc.py) -> the JSON bench.py reads for `roofline.limited_by` / `roofline_kernels`.

    python tools/pmc_bounds_json.py configs1=profiles/r04/r04_pmc_groups_configs1.txt [configs3=...] > profiles/r04/r04_kernel_bounds.json

Per kernel symbol (launch averages):
  mfma_busy      SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs the launch's workgroups occupy x kernel cycles): share of the matrix pipes' time, on the CUs the
                 launch runs on, that an MFMA is executing (kernel cycles = duration x the clock of the run)
  mfma_busy_chip the same over all 1024 SIMDs of the chip (= the roofline fraction seen from the counters)
  wave_issuing / wave_parked / wave_stalled   SQ_ACTIVE_INST_ANY, SQ_WAIT_ANY (s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue stall) over SQ_WAVE_CYCLES
  valu_active    SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES
  l2_hit         TCC_HIT / (TCC_HIT + TCC_MISS)
  fabric_read_mb / fabric_write_mb   TCC_EA0_RDREQ x 128 B (the gfx950 correction of MI355X_MICROARCH.md: wide reads are tallied at 64 B) / WRREQ x 64 B
  lds_conflict   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  bound          what the numbers say holds the kernel up (see classify())"""
import json
import re
import sys

from traffic_json import symbol

CLOCK_GHZ = 2.2          # the configs[1] solve runs at 2.17-2.25 GHz (bench.py clock_power)
WAVES_PER_WG = {"gemm_bf16_kernel<EPI_GATE_RES>": 8, "gemm_qkv_fused_kernel": 8, "gemm_bf16_kernel<EPI_BIAS_GELU>": 4, "attn_fwd_splitkv_kernel": 8,
                "gemm_bf16_kernel<EPI_QK_ROPE>": 8, "gemm_bf16_kernel<EPI_V_T>": 8}


def parse(path):
    groups, names = {}, None
    for line in open(path):
        if line.startswith("## "):
            names = None
            continue
        if line.startswith("kernel"):
            names = line.split()[2:]
            continue
        m = re.match(r"(.{40})\s+(\d+)\s+(.*)", line)
        if not m or names is None:
            continue
        sym = symbol(m.group(1)) or ("attn_fwd_splitkv_kernel" if "attn_fwd" in m.group(1) else None)
        if sym is None:
            continue
        vals = [float(v) for v in m.group(3).split()]
        n = int(m.group(2))
        e = groups.setdefault(sym, {})
        for k, v in zip(names, vals):
            a = e.setdefault(k, [0, 0.0])
            a[0] += n
            a[1] += n * v
    return {s: {k: t / n for k, (n, t) in e.items()} for s, e in groups.items()}


def get(e, name):
    # the rocpd_pmc header truncates counter names to 22 characters: match by suffix
    for k, v in e.items():
        if k == name or name.endswith(k.lstrip("_")) or k.endswith(name[-22:]):
            return v
    return None


HBM_ACHIEVABLE_TBS = 6.3      # MI355X_MICROARCH.md: float4 copy, 79 % of the 8 TB/s spec


def classify(r):
    """(verdict, sentence) from THIS kernel's counters -- no text is shared between kernels.  verdict: "mfma" (matrix pipe busy > 0.6 on the
    launch's own CUs), "hbm" (fabric bytes per launch at > 0.6 of the achievable HBM rate), "latency" (waves parked at s_waitcnt / s_barrier
    > 0.35 of their life with no pipe saturated), "valu" (VALU active > 0.5), else "issue"."""
    busy, parked, stalled, issuing = r.get("mfma_busy"), r.get("wave_parked"), r.get("wave_stalled"), r.get("wave_issuing")
    valu, l2 = r.get("valu_active"), r.get("l2_hit")
    mb = (r.get("fabric_read_mb") or 0.0) + (r.get("fabric_write_mb") or 0.0)
    dur = r.get("avg_us_under_pmc")
    mem_frac = (mb * 1e6 / (dur * 1e-6) / (HBM_ACHIEVABLE_TBS * 1e12)) if (mb and dur) else None
    if mem_frac is not None:
        r["fabric_frac_of_hbm"] = round(mem_frac, 3)
    facts = []
    chip = r.get("mfma_busy_chip")
    has_mfma = busy is not None or (chip is not None and chip > 0)
    if busy is not None:
        facts.append(f"matrix pipe busy {busy:.2f} of the time on the launch's own CUs ({chip or 0:.2f} chip-wide)")
    elif has_mfma:
        facts.append(f"matrix pipe busy {chip:.2f} chip-wide")
        busy = chip          # (a lower bound of the own-CU figure: the launch's workgroup count per CU is not tabulated for this symbol)
    else:
        facts.append("no MFMA work")
    if parked is not None:
        facts.append(f"waves parked at s_waitcnt / s_barrier {parked:.2f}, issue-stalled {stalled:.2f}, issuing {issuing:.2f}")
    if valu is not None:
        facts.append(f"VALU active {valu:.2f}")
    if mem_frac is not None:
        facts.append(f"{mb:.1f} MB of fabric traffic per launch = {mem_frac:.2f} of the achievable HBM rate" + (f", L2 hit {l2:.2f}" if l2 is not None else ""))
    if r.get("lds_conflict") is not None:
        facts.append(f"LDS conflict share {r['lds_conflict']:.2f}")
    if busy is not None and busy > 0.6:
        v = "mfma"
    elif mem_frac is not None and mem_frac > 0.6:
        v = "hbm"
    elif parked is not None and parked > 0.35:
        v = "latency"
    elif valu is not None and valu > 0.5:
        v = "valu"
    else:
        v = "issue"
    why = {"mfma": "the matrix pipe is the busiest resource", "hbm": "the launch moves its bytes at the memory system's rate",
           "latency": ("no pipe is saturated: the waves wait -- " + ("for operand arrival and the per-K-tile hand-off" if has_mfma
                       else "for one memory round trip in and one write-through round trip out, with too few bytes in flight per CU to cover it")),
           "valu": "the vector ALU is the busiest resource", "issue": "the waves are issuing or issue-stalled; no pipe and no memory level is saturated"}[v]
    return v, f"{v}: {why} ({'; '.join(facts)})"


def main(args):
    out = {"_source": "rocprofv3 --pmc group passes over bench.py (tools/gpu_session.sh pmc:<workload>), per-launch averages: " + ", ".join(a.split("=")[1] for a in args)}
    for a in args:
        wl, path = a.split("=")
        res = {}
        for sym, e in parse(path).items():
            dur = get(e, "dur_us")
            waves = get(e, "SQ_WAVES")
            wc = get(e, "SQ_WAVE_CYCLES")
            r = {"avg_us_under_pmc": round(dur, 2) if dur else None}
            mf = get(e, "SQ_VALU_MFMA_BUSY_CYCLES")
            if mf and dur:
                cyc = dur * 1e-6 * CLOCK_GHZ * 1e9
                r["mfma_busy_chip"] = round(mf / (1024 * cyc), 4)
                if waves and sym in WAVES_PER_WG:
                    simds = min(1024.0, waves / WAVES_PER_WG[sym] * 4)          # one workgroup per CU on every big launch
                    r["mfma_busy"] = round(mf / (simds * cyc), 4)
            if not wc:          # the wave-cycle group was not collected: the three wave states are disjoint and sum to it (MI355X_MICROARCH.md, PMC slots)
                parts = [get(e, c) for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY")]
                if all(v is not None for v in parts):
                    wc = sum(parts)
                    r["wave_cycles_from"] = "sum of the three wave states (SQ_WAVE_CYCLES pass failed)"
            if wc:
                for key, ctr in (("wave_issuing", "SQ_ACTIVE_INST_ANY"), ("wave_parked", "SQ_WAIT_ANY"), ("wave_stalled", "SQ_WAIT_INST_ANY"),
                                 ("valu_active", "SQ_ACTIVE_INST_VALU")):
                    v = get(e, ctr)
                    if v is not None:
                        r[key] = round(v / wc, 4)
            if len(r) <= 1:
                continue        # nothing but a duration: no bound to state
            hit, miss = get(e, "TCC_HIT_sum"), get(e, "TCC_MISS_sum")
            if hit is not None and miss is not None and hit + miss > 0:
                r["l2_hit"] = round(hit / (hit + miss), 4)
            rd, wr = get(e, "TCC_EA0_RDREQ_sum"), get(e, "TCC_EA0_WRREQ_sum")
            if rd is not None:
                r["fabric_read_mb"] = round(rd * 128 / 1e6, 2)
            if wr is not None:
                r["fabric_write_mb"] = round(wr * 64 / 1e6, 2)
            c, ia = get(e, "SQ_LDS_BANK_CONFLICT"), get(e, "SQ_LDS_IDX_ACTIVE")
            if c is not None and ia:
                r["lds_conflict"] = round(c / ia, 4)
            r["verdict"], r["bound"] = classify(r)
            res[sym] = r
        out[wl] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.path.insert(0, __file__.rsplit("/", 1)[0])
    main(sys.argv[1:])
