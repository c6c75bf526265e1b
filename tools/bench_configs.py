#!/usr/bin/env python
"""Throughput of the OTHER BASELINE configurations on one MI355X, for information (bench.py measures configs[1] only).

    python tools/bench_configs.py            # cfg4 share (B=8, 4 s prompt + 8 s generated), cfg5 shape (30 s edit, NFE 48; bf16 and fp8)

Same timing discipline as bench.py: inputs resident, warm-up, K timed utterance batches bracketed by synchronize,
vocoder decode + D2H inside; audio-seconds = what the reference's driver would vocode for that call."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from lemas_tts_amd import synth  # noqa: E402
from lemas_tts_amd.engine import VocosEngine  # noqa: E402
from lemas_tts_amd.model.cfm import CFM  # noqa: E402
from lemas_tts_amd.model.layout import DiTArch  # noqa: E402

VOCAB, HOP, SR = 898, 256, 24000


def run(name, B, F_, N, nfe, cfg, coef, fp8, steps=4, warmup=1, vocode_from=None, edit=None):
    dev = torch.device("cuda:0")
    arch = DiTArch()
    model = CFM(arch, VOCAB, synth.synth_cfm_state_dict(arch, VOCAB, 1234), device=dev, fp8_weights=bool(fp8))
    model.engine.set_option("table_cache", 0)
    voc = VocosEngine(synth.synth_vocos_state_dict(1234), device=dev)
    cond = torch.stack([torch.from_numpy(synth.synth_cond_mel(10 + b, F_)) for b in range(B)]).to(dev)
    text = torch.stack([torch.from_numpy(synth.synth_tokens(20 + b, round(N * 0.17), VOCAB)) for b in range(B)]).to(dev)
    y0 = torch.stack([torch.from_numpy(synth.synth_noise(30 + b, N)) for b in range(B)]).to(dev)
    start = F_ - 1 if vocode_from is None else vocode_from
    L = N - start
    host = torch.empty((B, HOP * (L - 1)), dtype=torch.float32).pin_memory()
    kw = dict(edit_mask=edit) if edit is not None else {}

    def step():
        out, _ = model.sample(cond, text, N, steps=nfe, cfg_strength=cfg, sway_sampling_coef=coef, y0=y0, use_acc_grl=False, **kw)
        host.copy_(voc.decode(out[:, start:, :].permute(0, 2, 1)), non_blocking=True)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    audio = B * HOP * (L - 1) / SR
    assert np.isfinite(host.numpy()).all()
    print(json.dumps({"config": name, "batch": B, "ref_frames": F_, "frames": N, "nfe": nfe, "dtype": "fp8" if fp8 else "bf16",
                      "ms_per_batch": round(1e3 * dt, 2), "audio_seconds_per_batch": round(audio, 3), "audio_seconds_per_sec": round(audio / dt, 1)}))
    del model, voc
    torch.cuda.empty_cache()


if __name__ == "__main__":
    run("configs[3] per-GPU share: 8 x (4 s prompt + 8 s generated)", 8, 375, 1125, 32, 2.0, 5, 0)
    run("configs[3] per-GPU share, fp8 GEMMs", 8, 375, 1125, 32, 2.0, 5, 1)
    run("configs[2]-like: 8 x (10 s prompt + 10 s generated), equal lengths", 8, 938, 1875, 32, 2.0, 5, 0, steps=2)
    F5 = 720000 // 256 + 1
    sys.path.insert(0, "lemas_tts_amd")
    from lemas_tts_amd.scripts.speech_edit_multilingual import build_edit_mask
    em = build_edit_mask([(4.0, 6.5), (12.0, 15.0), (22.0, 24.0)], 720000)
    run("configs[4]: 30 s edit, 3 spans, NFE 48, cfg 2, whole utterance vocoded (bf16 GEMMs)", 1, F5, F5 + 1, 48, 2.0, 3.0, 0, steps=2, vocode_from=0, edit=em)
    run("configs[4]: same, fp8 MFMA weights", 1, F5, F5 + 1, 48, 2.0, 3.0, 1, steps=2, vocode_from=0, edit=em)
