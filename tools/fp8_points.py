#!/usr/bin/env python
"""fp8 accuracy points (needs the MI355X): mel-MSE against the REFERENCE's own output with bf16 GEMMs, with MXFP8 GEMMs (e4m3 weights AND
MXFP8 activations, option fp8 = 1) and with weights-only fp8 (e4m3 weights, bf16 activations, option fp8 = 2), on the plain synthetic
weights and on the activation-outlier stress weights (tests/golden/full_outlier.npz: 1 % of the residual channels x30).

    python tools/fp8_points.py [fixture ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lemas_tts_amd import synth  # noqa: E402
from lemas_tts_amd.model.cfm import CFM  # noqa: E402
from lemas_tts_amd.model.layout import DiTArch  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def main(names):
    for name in names:
        fx = synth.expand_reference_fixture(dict(np.load(os.path.join(GOLDEN, name + ".npz"))))
        arch = DiTArch(depth=int(fx["arch_depth"]))
        sd = synth.synth_cfm_state_dict(arch, int(fx["vocab"]), int(fx["wseed"]), prosody=bool(fx["prosody"]),
                                        outlier=tuple(fx["outlier"]) if "outlier" in fx else None)
        m = CFM(arch, int(fx["vocab"]), sd, device="cuda:0")
        coef = None if np.isnan(fx["coef"]) else float(fx["coef"])
        F = int(fx["lens"][0])
        res = {}
        for mode in (0, 1, 2):
            m.engine.set_option("fp8_outlier_guard", 0)      # raw quantisation cost: the guard would keep the outlier fixture on bf16
            m.engine.set_option("fp8", mode)
            out, _ = m.sample(torch.from_numpy(fx["cond"]), torch.from_numpy(fx["text"]), int(fx["duration"][0]), lens=torch.from_numpy(fx["lens"]),
                              steps=int(fx["steps"]), cfg_strength=float(fx["cfg"]), sway_sampling_coef=coef, y0=torch.from_numpy(fx["y0"]),
                              use_acc_grl=False)
            d = (out.cpu().double().numpy() - fx["out"].astype(np.float64))[:, F:]
            res[mode] = float((d ** 2).mean())
        print(f"{name:22s} mel-MSE vs reference: bf16 {res[0]:.3e} | MXFP8 weights+activations {res[1]:.3e} | fp8 weights only {res[2]:.3e}"
              f"   (|out| rms {float(np.sqrt((fx['out'][:, F:] ** 2).mean())):.3f})")
        del m


if __name__ == "__main__":
    main(sys.argv[1:] or ["full_plain", "full_outlier", "configs1_nfe32"])
