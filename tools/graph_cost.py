#!/usr/bin/env python
"""What a NEW sequence length costs (graph capture + instantiate) vs a cached one (development aid)."""
import sys, time
import torch
sys.path.insert(0, ".")
from lemas_tts_amd import synth
from lemas_tts_amd.model.cfm import CFM
from lemas_tts_amd.model.layout import DiTArch
arch = DiTArch()
m = CFM(arch, 898, synth.synth_cfm_state_dict(arch, 898, 1234), device="cuda:0")
F_ = 400
cond = torch.from_numpy(synth.synth_cond_mel(1, F_))[None].cuda()
for graph in (1, 0):
    m.engine.set_option("graph", graph)
    for N in (900, 901, 902, 900, 901):
        text = torch.from_numpy(synth.synth_tokens(2, 100, 898))[None].cuda()
        y0 = torch.from_numpy(synth.synth_noise(3, N))[None].cuda()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m.sample(cond, text, N, steps=32, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False)
        torch.cuda.synchronize()
        print(f"graph={graph} N={N}: {1e3 * (time.perf_counter() - t0):.1f} ms")
