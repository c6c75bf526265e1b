#!/usr/bin/env python
"""Energy per launch and per utterance of the step-loop kernels: the package sits on its 1.4 kW cap in every workload, so
the quantity that bounds the step time is joules per utterance -- this tool says which kernel spends them.

    python tools/energy_table.py [--workloads configs1 configs3] [--seconds 3] [--out profiles/r06/r06_energy.json]

For each workload, each kernel class of a DiT block (AdaLN LayerNorm, fused QK+V projection or QK / V, attention, out-projection, FF1, FF2) is
looped ALONE at the workload's launch shape for >= `seconds`, in as many concurrent lanes as the engine runs (one per CFG branch: two host
threads, each on its own stream through lemas_k_bench), while a second thread samples `rocm-smi --showclocks --showpower`:
    W (package power while only that kernel runs), sclk, us per launch (per lane, under that concurrency)
    J per launch = W x us / lanes;  J per utterance = J per launch x launches per utterance (depth x NFE x lanes x launches per block)
The workload itself is then run through bench.py (same box, same lease) for its package power and ms per utterance: the table's sum is checked
against W_workload x ms (plus the time the table does not cover -- per-step input projection / ConvPos / final layer, hoists, vocoder -- priced
at the workload's own power).  `idle_w` (no kernel running) splits every figure into a static and a dynamic part.
"""
import argparse
import ctypes as C
import json
import os
import re
import shutil
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402,F401  (one HIP runtime per process: torch's)
from lemas_tts_amd import _lib   # noqa: E402

DEPTH, HEADS = 22, 16
SHAPES = {   # B per CFG branch, frames, NFE  (bench.py WORKLOADS)
    "configs1": dict(B=1, N=1875, nfe=32),
    "configs3": dict(B=8, N=1125, nfe=32),
    "configs4": dict(B=1, N=2814, nfe=48),
}


def smi_sample(exe):
    try:
        out = subprocess.run([exe, "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception:
        return None
    m1 = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    m2 = re.search(r"Package Power \(W\): ([0-9.]+)", out)
    return (int(m1.group(1)), float(m2.group(1))) if m1 and m2 else None


class Sampler:
    def __init__(self, exe):
        self.exe, self.samples, self.stop = exe, [], threading.Event()
        self.th = threading.Thread(target=self.run, daemon=True)

    def run(self):
        time.sleep(0.7)                                   # let the clock settle on the new load
        while not self.stop.is_set():
            s = smi_sample(self.exe)
            if s:
                self.samples.append(s)

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *a):
        self.stop.set()
        self.th.join(timeout=30)

    def median(self, busy_only=True):
        xs = [s for s in self.samples if s[0] > 500] if busy_only else self.samples
        if not xs:
            return None, None, 0
        med = lambda v: sorted(v)[len(v) // 2]
        return med([s[0] for s in xs]), med([s[1] for s in xs]), len(xs)


def kbench(L, what, M, N, K, iters, tile=0):
    us = C.c_double()
    rc = L.lemas_k_bench(what.encode(), M, N, K, iters, tile, C.byref(us))
    if rc != 0:
        raise RuntimeError(f"lemas_k_bench {what} {M} {N} {K}: {L.lemas_last_error().decode()}")
    return us.value


def loop_class(L, exe, what, M, N, K, lanes, seconds):
    """-> (us per launch per lane under `lanes`-way concurrency, sclk MHz, package W, samples)"""
    probe = kbench(L, what, M, N, K, 50)
    iters = max(200, int(0.6 * 1e6 / probe))              # ~0.6 s per call: allocation / fill gaps between calls stay under a few percent
    res = [[] for _ in range(lanes)]
    t_end = time.time() + seconds

    def lane(i):
        while time.time() < t_end:
            res[i].append(kbench(L, what, M, N, K, iters))
    with Sampler(exe) as smp:
        ths = [threading.Thread(target=lane, args=(i,)) for i in range(lanes)]
        [t.start() for t in ths]
        [t.join() for t in ths]
    us = sum(sum(r) / len(r) for r in res) / lanes
    sclk, w, n = smp.median()
    return us, sclk, w, n


def classes_for(w, fused_qkv):
    B, N = w["B"], w["N"]
    pitch = (N + 127) // 128 * 128
    rows = B * pitch                                        # rows of one lane's launches (one CFG branch)
    cl = [("ln_mod", "ln_mod", rows, 1024, 0, 2)]           # (label, kernel, M, N, K, launches per block per lane)
    if fused_qkv:
        cl.append(("gemm_qkv_fused", "gemm_qkv", rows, 3072, 1024, 1))
    else:
        cl += [("gemm_qk_rope", "gemm_qk", rows, 2048, 1024, 1), ("gemm_v_t", "gemm_v", rows, 1024, 1024, 1)]
    cl += [("attention", "attention", N, B * HEADS, 0, 1),
           ("gemm_attn_out", "gemm_gate", rows, 1024, 1024, 1),
           ("gemm_ff1_gelu", "gemm_gelu", rows, 2048, 1024, 1),
           ("gemm_ff2", "gemm_gate", rows, 1024, 2048, 1)]
    return cl, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", nargs="+", default=["configs1", "configs3"])
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "energy.json"))
    ap.add_argument("--no-bench", action="store_true", help="skip the bench.py run of the workload itself")
    args = ap.parse_args()
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    L = _lib.testlib()
    out = {"_source": "tools/energy_table.py: each block kernel looped alone (two lanes) under a rocm-smi sampler, then bench.py on the same lease",
           "seconds_per_class": args.seconds}
    time.sleep(2.0)
    idle = [smi_sample(exe) for _ in range(4)]
    idle = [s for s in idle if s]
    out["idle"] = {"sclk_mhz": sorted(s[0] for s in idle)[len(idle) // 2], "package_w": sorted(s[1] for s in idle)[len(idle) // 2]} if idle else None
    idle_w = out["idle"]["package_w"] if out["idle"] else 0.0
    for name in args.workloads:
        w = SHAPES[name]
        lanes = 2
        pitch = (w["N"] + 127) // 128 * 128
        fused = ((w["B"] * pitch + 255) // 256) * (3 * 1024 // 128) <= 250          # engine_dit.hip: one QK+V launch while it fits one round
        cl, rows = classes_for(w, fused)
        table, tot_j, tot_us = [], 0.0, 0.0
        for label, what, M, N, K, per_block in cl:
            us, sclk, pw, n = loop_class(L, exe, what, M, N, K, lanes, args.seconds)
            launches = DEPTH * w["nfe"] * lanes * per_block
            j_launch = pw * us * 1e-6 / lanes
            row = {"class": label, "kernel": what, "M": M, "N": N, "K": K, "lanes": lanes, "us_per_launch": us, "sclk_mhz": sclk, "package_w": pw,
                   "samples": n, "j_per_launch": j_launch, "j_per_launch_dynamic": max(pw - idle_w, 0.0) * us * 1e-6 / lanes,
                   "launches_per_step_batch": launches, "j_per_step_batch": j_launch * launches,
                   "ms_per_step_batch_if_back_to_back": us * launches / lanes * 1e-3}
            table.append(row)
            tot_j += row["j_per_step_batch"]
            tot_us += us * launches / lanes
            print(f"[{name}] {label:16s} M={M:5d} N={N:4d} K={K:4d}: {us:7.2f} us  {sclk} MHz  {pw:7.1f} W  {j_launch * 1e3:7.3f} mJ/launch  "
                  f"{row['j_per_step_batch']:7.2f} J per step batch", flush=True)
        entry = {"shape": w, "rows_per_lane": rows, "classes": table, "sum_j_block_kernels": tot_j, "sum_ms_block_kernels_back_to_back": tot_us * 1e-3}
        if not args.no_bench:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", name, "--no-cpu-baseline", "--steps", "8", "--warmup", "2"],
                               capture_output=True, text=True, timeout=900)
            line = [x for x in r.stdout.splitlines() if x.startswith("{")]
            if line:
                b = json.loads(line[-1])
                cp = b.get("clock_power") or {}
                ms = b["ms_per_step"]
                measured_j = cp.get("package_w", 0.0) * ms * 1e-3
                rest_ms = max(ms - tot_us * 1e-3, 0.0)
                entry["workload"] = {"ms_per_step_batch": ms, "value": b["value"], "unit": b["unit"], "sclk_mhz": cp.get("sclk_mhz"), "package_w": cp.get("package_w"),
                                     "j_per_step_batch": measured_j, "utterances_per_step_batch": w["B"], "j_per_utterance": measured_j / w["B"]}
                entry["check"] = {"sum_j_block_kernels": tot_j, "rest_ms_not_in_table": rest_ms,
                                  "rest_j_at_workload_power": rest_ms * 1e-3 * cp.get("package_w", 0.0),
                                  "table_plus_rest_j": tot_j + rest_ms * 1e-3 * cp.get("package_w", 0.0), "measured_j": measured_j,
                                  "ratio": (tot_j + rest_ms * 1e-3 * cp.get("package_w", 0.0)) / measured_j if measured_j else None,
                                  "note": "rest = step-batch time the block kernels do not account for when run back to back (per-step input projection, "
                                          "ConvPos, final layer, CFG / Euler, hoists, vocoder, launch gaps), priced at the workload's own package power"}
                print(f"[{name}] workload: {ms:.2f} ms, {cp.get('package_w')} W at {cp.get('sclk_mhz')} MHz -> {measured_j:.1f} J per step batch; "
                      f"table {tot_j:.1f} J + rest {entry['check']['rest_j_at_workload_power']:.1f} J = {entry['check']['ratio']:.3f} of it", flush=True)
            else:
                entry["workload_error"] = r.stderr[-400:]
        for row in table:
            row["share_of_table_j"] = row["j_per_step_batch"] / tot_j
        out[name] = entry
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
