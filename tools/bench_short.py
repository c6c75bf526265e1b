#!/usr/bin/env python
"""Short-utterance regime (what the entry scripts see most): batch 1, N = 500-1000 frames, dual lanes on/off (development aid)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from lemas_tts_amd import synth  # noqa: E402
from lemas_tts_amd.engine import VocosEngine  # noqa: E402
from lemas_tts_amd.model.cfm import CFM  # noqa: E402
from lemas_tts_amd.model.layout import DiTArch  # noqa: E402

dev = torch.device("cuda:0")
arch = DiTArch()
model = CFM(arch, 898, synth.synth_cfm_state_dict(arch, 898, 1234), device=dev)
model.engine.set_option("table_cache", 0)
voc = VocosEngine(synth.synth_vocos_state_dict(1234), device=dev)
for F_, N, nfe in ((188, 375, 32), (375, 750, 16), (375, 750, 32), (469, 1000, 32), (600, 1407, 32)):
    cond = torch.from_numpy(synth.synth_cond_mel(1, F_))[None].to(dev)
    text = torch.from_numpy(synth.synth_tokens(2, round(N * 0.17), 898))[None].to(dev)
    y0 = torch.from_numpy(synth.synth_noise(3, N))[None].to(dev)
    res = {}
    for dual in (1, 0, 1, 0):
        model.engine.set_option("dual", dual)

        def step():
            out, _ = model.sample(cond, text, N, steps=nfe, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False)
            return voc.decode(out[:, F_ - 1:, :].permute(0, 2, 1)).cpu()
        step(); step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        res.setdefault(dual, []).append(round(1e3 * (time.perf_counter() - t0) / 6, 2))
    audio = 256 * (N - F_) / 24000
    print(json.dumps({"frames": N, "ref": F_, "nfe": nfe, "ms_dual": res[1], "ms_single": res[0],
                      "audio_s_per_s_dual": round(audio / (min(res[1]) * 1e-3), 1), "audio_s_per_s_single": round(audio / (min(res[0]) * 1e-3), 1)}))
