#!/usr/bin/env python
"""fp8 on activation outliers at a PRODUCTION step count (development aid; needs the MI355X): mel-MSE against the reference's own output
for bf16 / MXFP8 (weights + activations) / weights-only fp8 on tests/golden/configs0_outlier_nfe32.npz (1 % of the residual channels x30
in all 22 blocks, NFE 32) next to the short-solve stress fixtures."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import test_gpu_00_sample as T  # noqa: E402
from lemas_tts_amd import synth  # noqa: E402
from lemas_tts_amd.model.cfm import CFM  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(T.__file__), "golden")
for name in sys.argv[1:] or ["configs0_outlier_nfe32", "full_outlier", "full_plain", "configs0_nfe16"]:
    fx, arch, sd = T._load(GOLDEN, name)
    fx = synth.expand_reference_fixture(fx)
    m = CFM(arch, int(fx["vocab"]), sd, device="cuda:0")
    m.engine.set_option("fp8_outlier_guard", 0)      # the raw cost of quantising (the guard would keep outlier checkpoints on bf16)
    got = {}
    for mode in (0, 1, 2):
        m.engine.set_option("fp8", mode)
        out, _ = m.sample(torch.from_numpy(fx["cond"]), torch.from_numpy(fx["text"]), int(fx["duration"][0]), lens=torch.from_numpy(fx["lens"]),
                          steps=int(fx["steps"]), cfg_strength=float(fx["cfg"]), sway_sampling_coef=float(fx["coef"]), y0=torch.from_numpy(fx["y0"]),
                          use_acc_grl=False)
        got[mode] = T._gen_mse(out.cpu().numpy(), fx["out"], fx)
    print(f"{name:24s} steps {int(fx['steps']):2d}  mel-MSE vs reference: bf16 {got[0]:.3e}   MXFP8 weights+activations {got[1]:.3e}   fp8 weights only {got[2]:.3e}", flush=True)
    del m

# which GEMM site carries the outlier error: one site at a time on fp8, the rest bf16 (option fp8_sites), guard off
for name in ["configs0_outlier_nfe32", "configs0_nfe16"]:
    fx, arch, sd = T._load(GOLDEN, name)
    fx = synth.expand_reference_fixture(fx)
    m = CFM(arch, int(fx["vocab"]), sd, device="cuda:0")
    m.engine.set_option("fp8_outlier_guard", 0)      # the raw cost of quantising (the guard would keep outlier checkpoints on bf16)
    m.engine.set_option("fp8", 1)
    row = []
    for mask in (1, 2, 4, 8, 5, 10, 15):
        m.engine.set_option("fp8_sites", mask)
        out, _ = m.sample(torch.from_numpy(fx["cond"]), torch.from_numpy(fx["text"]), int(fx["duration"][0]), lens=torch.from_numpy(fx["lens"]),
                          steps=int(fx["steps"]), cfg_strength=float(fx["cfg"]), sway_sampling_coef=float(fx["coef"]), y0=torch.from_numpy(fx["y0"]),
                          use_acc_grl=False)
        row.append(f"{mask:2d}: {T._gen_mse(out.cpu().numpy(), fx['out'], fx):.2e}")
    print(f"{name:24s} fp8_sites mask (1 QKV, 2 out-proj, 4 FF1, 8 FF2) -> mel-MSE   " + "   ".join(row), flush=True)
    del m
