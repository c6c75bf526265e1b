"""Reproducer for the round-1 driver failure (tests/test_gpu_06_configs.py::test_config3..., mel-MSE 0.109 on a fresh box).

Runs the config3 shape (B = 8 ragged, prosody, sway, depth 2, 3 steps) several times in ONE process under different engine
options and compares the results bit for bit against the eager single-stream run; --oracle also checks that run against the
CPU oracle.  Prints, per sample, how many generated frames differ and where.

    python tools/repro_config3.py [--reps 5] [--oracle] [--modes default,nograph,nodual,eager]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from lemas_tts_amd import synth
from lemas_tts_amd.model.cfm import CFM
from lemas_tts_amd.model.layout import DiTArch

VOCAB = 898
Fs = [375, 420, 500, 610, 700, 780, 850, 938]
Ns = [900, 1010, 1130, 1290, 1420, 1560, 1700, 1900]


def inputs(seed, B, nts):
    Fm, Nm, Tm = max(Fs), max(Ns), max(nts)
    cond = torch.zeros(B, Fm, 100)
    text = torch.full((B, Tm), -1, dtype=torch.long)
    y0 = torch.zeros(B, Nm, 100)
    for b in range(B):
        cond[b, : Fs[b]] = torch.from_numpy(synth.synth_cond_mel(seed + b, Fs[b]))
        text[b, : nts[b]] = torch.from_numpy(synth.synth_tokens(seed + b, nts[b], VOCAB))
        y0[b, : Ns[b]] = torch.from_numpy(synth.synth_noise(seed + b, Ns[b]))
    return cond, text, y0


def report(tag, out, base):
    bad_total = 0
    lines = []
    for b in range(out.shape[0]):
        a, r = out[b, Fs[b]:Ns[b]], base[b, Fs[b]:Ns[b]]
        d = (a - r).double()
        rows = (d.abs().amax(dim=1) > 0).nonzero().flatten()
        mse = float((d ** 2).mean())
        nan = int(torch.isnan(a).sum())
        if len(rows) or nan:
            bad_total += 1
            lines.append(f"    sample {b}: mse {mse:.3e} rows differing {len(rows)}/{a.shape[0]} first {int(rows[0]) + Fs[b] if len(rows) else -1} "
                         f"last {int(rows[-1]) + Fs[b] if len(rows) else -1} nan {nan}")
    print(f"  [{tag}] samples differing from baseline: {bad_total}/8")
    for l in lines:
        print(l)
    return bad_total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--modes", default="default,eager,default,nograph,nodual,default")
    ap.add_argument("--tol", type=float, default=0.0, help="0: bitwise; else per-sample mse threshold")
    a = ap.parse_args()
    arch = DiTArch(depth=2)
    sd = synth.synth_cfm_state_dict(arch, VOCAB, 51, prosody=True)
    B = 8
    nts = [round(n * 0.17) for n in Ns]
    cond, text, y0 = inputs(52, B, nts)
    pros = torch.from_numpy(synth.synth_prosody_embed(53, B))
    lens, dur = torch.tensor(Fs), torch.tensor(Ns)

    def run(mode):
        m = CFM(arch, VOCAB, sd, device="cuda:0", use_prosody_encoder=True)
        if mode in ("nograph", "eager"):
            m.engine.set_option("graph", 0)
        if mode in ("nodual", "eager"):
            m.engine.set_option("dual", 0)
        outs = []
        for _ in range(a.reps):
            out, _ = m.sample(cond, text, dur, lens=lens, steps=3, cfg_strength=2.0, sway_sampling_coef=5, y0=y0,
                              use_acc_grl=False, prosody_embeds=pros)
            outs.append(out.cpu())
        m.engine.close()
        return outs

    results = []
    for mode in a.modes.split(","):
        outs = run(mode)
        results.append((mode, outs))
        print(f"mode {mode}: {len(outs)} runs done", flush=True)
    base = None
    for mode, outs in results:
        if mode == "eager":
            base = outs[0]
            break
    if base is None:
        base = results[0][1][0]
    nbad = 0
    for mode, outs in results:
        for i, o in enumerate(outs):
            nbad += report(f"{mode} #{i}", o, base)
    if a.oracle:
        from oracle import lemas_oracle as O
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
        ref, _ = O.OracleCFM(sd, arch).sample(cond, text, dur, y0=y0, lens=lens, steps=3, cfg_strength=2.0,
                                              sway_sampling_coef=5, prosody_embeds=pros)
        for b in range(B):
            d = (base[b, Fs[b]:Ns[b]] - ref[b, Fs[b]:Ns[b]]).double()
            print(f"  baseline vs oracle sample {b}: mse {float((d ** 2).mean()):.3e}")
    print("REPRO_BAD" if nbad else "REPRO_CLEAN", nbad)


if __name__ == "__main__":
    main()
