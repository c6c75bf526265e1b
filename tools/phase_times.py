#!/usr/bin/env python
"""Where one configs[1] utterance spends its wall time outside the step loop (development aid)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from lemas_tts_amd import synth  # noqa: E402
from lemas_tts_amd.engine import VocosEngine  # noqa: E402
from lemas_tts_amd.model.cfm import CFM, time_grid  # noqa: E402
from lemas_tts_amd.model.layout import DiTArch  # noqa: E402

dev = torch.device("cuda:0")
arch = DiTArch()
m = CFM(arch, 898, synth.synth_cfm_state_dict(arch, 898, 1234), device=dev)
m.engine.set_option("table_cache", 0)
voc = VocosEngine(synth.synth_vocos_state_dict(1234), device=dev)
F_, N = 938, 1875
cond = torch.from_numpy(synth.synth_cond_mel(1, F_))[None].to(dev)
text = torch.from_numpy(synth.synth_tokens(2, round(N * 0.17), 898))[None].to(dev)
y0 = torch.from_numpy(synth.synth_noise(3, N))[None].to(dev)
cm = torch.zeros(1, N, dtype=torch.bool); cm[:, :F_] = True
cpad = torch.nn.functional.pad(cond, (0, 0, 0, N - F_))
tg = time_grid(32, 5).numpy()


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


t_prep = timed(lambda: m.engine.prepare(cpad, cm, text, tg, cond_frames=F_, cfg_strength=2.0))
t_solve = timed(lambda: m.engine.solve(y0))
out, _ = m.sample(cond, text, N, steps=32, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False)
mel = out[:, F_ - 1:, :].permute(0, 2, 1).contiguous()
t_voc = timed(lambda: voc.decode(mel))
t_all = timed(lambda: voc.decode(m.sample(cond, text, N, steps=32, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False)[0][:, F_ - 1:, :].permute(0, 2, 1)).cpu())
print(f"prepare (hoists) {t_prep:.2f} ms | solve (32 steps) {t_solve:.2f} ms = {t_solve / 32:.3f} ms/step | vocoder {t_voc:.2f} ms | whole utterance incl. D2H {t_all:.2f} ms")
