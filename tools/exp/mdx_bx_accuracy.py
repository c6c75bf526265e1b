#!/usr/bin/env python
"""Accuracy of the MDX-Net engine's modes against the reference's own outputs (tests/golden/mdxnet_*.npz): exact f32 MFMA, split-bf16 with 3 and
with 4 bf16 MFMAs per product.  Two metrics per case: max|err| / rms(ref) (the tests' bar) and rms(err) / rms(ref)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lemas_tts_amd import _lib, synth   # noqa: E402
from lemas_tts_amd.engine import MdxEngine   # noqa: E402
from lemas_tts_amd.uvr5.arch import KIM_VOCAL_1, MdxArch   # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
MINI = MdxArch(dim_f=32, dim_t=16, num_blocks=5, l=2, g=8, k=3, bn=4, bias=True)
KIM_SAMPLE = (slice(None), slice(None), slice(None, None, 16), slice(None, None, 8))

for name, arch in (("mini", MINI), ("kim", KIM_VOCAL_1)):
    fx = dict(np.load(os.path.join(G, f"mdxnet_{name}.npz")))
    sd = synth.synth_mdx_state_dict(arch, int(fx["seed_weights"][0]))
    x = torch.from_numpy(synth.synth_mdx_input(arch, int(fx["batch"][0]) if "batch" in fx else 1, int(fx["seed_input"][0]))).to("cuda:0")
    for mode, bx, prod in (("f32", False, 3), ("bf16x3", True, 3), ("bf16x4", True, 4)):
        eng = MdxEngine(arch, sd, bf16x3=bx)
        _lib.check(_lib.lib().lemas_mdx_set_option(eng._h, b"bf16x3_products", prod), "products")
        y = eng.forward(x).cpu().numpy()
        ref, got = (fx["output"], y) if name == "mini" else (fx["sample"], y[KIM_SAMPLE])
        rms = float(np.sqrt((ref.astype(np.float64) ** 2).mean())) if name == "mini" else float(fx["rms"][0])
        d = got.astype(np.float64) - ref
        print(f"{name:5s} {mode:7s} max|err|/rms {np.abs(d).max() / rms:.2e}   rms(err)/rms {np.sqrt((d ** 2).mean()) / rms:.2e}", flush=True)
        del eng
_lib.lib().lemas_mdx_set_option(None, b"bf16x3_products", 3)
