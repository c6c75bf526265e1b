#!/usr/bin/env python
"""Roofline lines of the rows either side of the sampler: one JSON line.

    python tools/frontend_bench.py [--prompt-seconds 10] [--iters 20] [--cpu-baseline]

What one prompt costs BEFORE the first ODE step, inputs resident on the device, HIP events on each engine's own stream around `iters`
back-to-back calls:

* ``resample``  lemas_resample_forward (utils_infer.py:494-496 and cfm.py:252-258): 44.1 kHz -> 24 kHz, 16 kHz -> 24 kHz, 24 kHz -> 16 kHz.
  HBM-priced: 4 B in + 4 B out per sample against 8 TB/s (the polyphase bank stays in L2).
* ``mel``       lemas_mel_forward (modules.py:75-143): wav -> [frames, 100] log-mel.  The rDFT is an exact-fp32 MFMA GEMM
  (frames x 1024 x 2 * 513) + the 513 -> 100 filterbank GEMM: priced against the fp32 matrix peak (157.3 TFLOP/s) AND as bytes against HBM.
* ``prosody``   lemas_prosody_fbank + lemas_prosody_encode (prosody_encoder.py:28-361, cfm.py:248-262): kaldi fbank + ECAPA-TDNN at the
  published Pretssel widths; FLOPs counted from the architecture (2 per multiply-add of every TDNN / Linear), against the fp32 matrix peak.
* ``prompt_total``  resample(24 -> 16 kHz) + fbank + encode + mel of one prompt, the way CFM.sample runs them, and what share of the
  headline utterance (BASELINE configs[1], ~100 ms) that is.
``--cpu-baseline``: the oracle restatements (oracle/prosody_oracle.py, oracle/lemas_oracle.py; test infrastructure, torch CPU fp32) on the same
inputs on this box's host cores, timed once.  Synthetic seeded weights and prompts."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lemas_tts_amd import synth   # noqa: E402
from lemas_tts_amd.engine import MelEngine, ResampleEngine   # noqa: E402
from lemas_tts_amd.model.layout import ProsodyArch   # noqa: E402

F32_MFMA_PEAK = 157.3e12
HBM_PEAK = 8.0e12


def ecapa_flops(arch: ProsodyArch, T: int) -> int:
    """2 x multiply-adds of every convolution / linear of the ECAPA-TDNN on T frames (engine_prosody.hip encode())."""
    ch, ks, L = arch.channels, arch.kernel_sizes, len(arch.channels)
    fl = 2 * T * arch.input_dim * ks[0] * ch[0]
    for i in range(1, L - 1):
        c, sub = ch[i], ch[i] // arch.res2net_scale
        fl += 2 * T * ch[i - 1] * c                              # tdnn1
        fl += (arch.res2net_scale - 1) * 2 * T * sub * ks[i] * sub    # Res2Net chunks
        fl += 2 * T * c * c                                      # tdnn2
        fl += 2 * (c * arch.se_channels) * 2                     # SE gate (one row)
        if ch[i - 1] != c:
            fl += 2 * T * ch[i - 1] * c
    cl = ch[L - 1]
    fl += 2 * T * sum(ch[1:L - 1]) * ks[L - 1] * cl              # MFA
    fl += 2 * T * cl * arch.attention_channels + 2 * T * arch.attention_channels * cl     # attention TDNN + conv
    if arch.global_context:
        fl += 2 * 2 * cl * arch.attention_channels
    fl += 2 * 2 * cl * arch.embed_dim
    return fl


def timed(stream, fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(iters):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters, (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt-seconds", type=float, default=10.0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--cpu-baseline", action="store_true")
    ap.add_argument("--api", action="store_true", help="also time the mirrored call surface end to end: infer_batch_process from a RAW prompt "
                    "(rms, log-mel, prosody encoder, 32 ODE steps with CFG, Vocos, D2H, host assembly) on a 22-block model")
    ap.add_argument("--utterance-ms", type=float, default=100.3, help="the headline utterance's time (profiles/r06/r06fin_bench.json) for the share line")
    args = ap.parse_args()
    dev = "cuda:0"
    sec = args.prompt_seconds
    g = torch.Generator().manual_seed(5)

    def prompt(sr):
        n = int(sec * sr)
        t = torch.arange(n) / sr
        return (0.1 * torch.sin(2 * np.pi * 180.0 * t) + 0.02 * torch.randn(n, generator=g))[None].to(dev)

    line = {"metric": "prompt front end (resample, log-mel, prosody encoder)", "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"one {sec:g} s prompt, batch 1, inputs resident in HBM", "iters": args.iters}}

    # ---- resample
    rs = {}
    for a, b in ((44100, 24000), (16000, 24000), (24000, 16000)):
        eng = ResampleEngine(a, b, device=dev)
        w = prompt(a)
        out = eng(w)
        dt, wall = timed(eng.stream, lambda: eng(w), args.iters)
        by = 4 * (w.numel() + out.numel())
        rs[f"{a}->{b}"] = {"us": dt * 1e6, "wall_us": wall * 1e6, "algorithmic_bytes": by, "achieved_GBps": by / dt / 1e9, "frac_of_hbm": by / dt / HBM_PEAK,
                           "audio_seconds_per_second": sec / dt}
    line["resample"] = rs

    # ---- log-mel
    mel = MelEngine(device=dev)
    w24 = prompt(24000)
    m = mel.frames_first(w24)
    frames = m.shape[1]
    dt, wall = timed(mel.stream, lambda: mel.frames_first(w24), args.iters)
    fl = 2 * frames * 1024 * (2 * 513) + 2 * frames * 513 * 100
    by = 4 * (w24.numel() + m.numel())
    line["mel"] = {"frames": frames, "us": dt * 1e6, "wall_us": wall * 1e6, "algorithmic_flops": fl, "algorithmic_bytes": by,
                   "roofline": {"bound": "mfma_f32", "achieved": fl / dt / 1e12, "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": fl / dt / F32_MFMA_PEAK},
                   "hbm_view": {"achieved_GBps": by / dt / 1e9, "frac": by / dt / HBM_PEAK}}
    w24b = w24.repeat(8, 1)
    mel.frames_first(w24b)
    dt8, _ = timed(mel.stream, lambda: mel.frames_first(w24b), args.iters)
    line["mel"]["batch8_us"] = dt8 * 1e6
    line["mel"]["batch8_frac_of_f32_mfma"] = 8 * fl / dt8 / F32_MFMA_PEAK

    # ---- prosody encoder
    from lemas_tts_amd.model.prosody_encoder import ProsodyEncoder
    arch = ProsodyArch()
    psd = synth.synth_prosody_encoder_state_dict(42, arch)
    enc = ProsodyEncoder(state_dict=psd, arch=arch, device=dev)
    w16 = prompt(16000)[0]
    fb = enc.extract_fbank_16k(w16)
    T = fb.shape[0]
    es = enc.engine.stream
    dtf, wallf = timed(es, lambda: enc.extract_fbank_16k(w16), args.iters)
    dte, walle = timed(es, lambda: enc.engine.encode(fb), args.iters)
    ffl = 2 * T * 512 * (2 * 257) + 2 * T * 257 * 80
    efl = ecapa_flops(arch, T)
    wbytes = 4 * sum(int(np.prod(v.shape)) for v in psd.values())
    line["prosody"] = {"frames": T,
                       "fbank": {"us": dtf * 1e6, "wall_us": wallf * 1e6, "algorithmic_flops": ffl, "frac_of_f32_mfma": ffl / dtf / F32_MFMA_PEAK},
                       "encode": {"us": dte * 1e6, "wall_us": walle * 1e6, "algorithmic_flops": efl, "weight_bytes": wbytes,
                                  "roofline": {"bound": "mfma_f32", "achieved": efl / dte / 1e12, "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                                               "frac": efl / dte / F32_MFMA_PEAK},
                                  "graph": int(enc.engine.option("graph")) if hasattr(enc.engine, "option") else None}}

    # ---- the prompt's whole front end as CFM.sample runs it (cfm.py:248-262 + modules.py:130-143)
    def whole():
        enc.embed_prompt(w24, 24000)
        mel.frames_first(w24)
    whole()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        whole()
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / args.iters
    line["prompt_total"] = {"wall_ms": tot * 1e3, "share_of_headline_utterance": tot * 1e3 / args.utterance_ms,
                            "note": "resample 24 -> 16 kHz + kaldi fbank + ECAPA + log-mel, host wall time between synchronisations "
                                    f"(the headline utterance takes {args.utterance_ms:g} ms)"}

    if args.api:
        # utils_infer.py:464-625 as the entry scripts call it: one RAW 24 kHz prompt of `sec` seconds, one text line of the same length, NFE 32,
        # CFG 2, sway 5 (capped), accent-GRL + prosody conditioning -- BASELINE configs[1]'s arithmetic PLUS everything around it
        from lemas_tts_amd.infer.utils_infer import infer_batch_process, load_vocoder
        from lemas_tts_amd.model.cfm import CFM
        from lemas_tts_amd.model.layout import DiTArch
        darch = DiTArch()
        vocab = {f"p{i}": i for i in range(898)}
        model = CFM(darch, 898, synth.synth_cfm_state_dict(darch, 898, 11, prosody=True), vocab_char_map=vocab, device=dev,
                    use_prosody_encoder=True, prosody_encoder=enc)
        vocoder = load_vocoder("vocos", device=dev, state_dict=synth.synth_vocos_state_dict(12))
        ntok = int(10 * sec)
        ref_text = [f"p{i}" for i in synth.synth_tokens(13, ntok, 898)]
        gen = [[f"p{i}" for i in synth.synth_tokens(14, ntok, 898)]]
        wav_cpu = w24.cpu()

        def call():
            return next(infer_batch_process((wav_cpu, 24000), ref_text, gen, model, vocoder, nfe_step=32, cfg_strength=2.0, sway_sampling_coef=5,
                                            use_acc_grl=True, use_prosody_encoder=True, ref_ratio=1, seed=3))
        for _ in range(2):
            wav_out, sr_out, spec_out = call()
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            wav_out, sr_out, spec_out = call()
            ts.append(time.perf_counter() - t0)
        best, med = min(ts), sorted(ts)[len(ts) // 2]
        asec = len(wav_out) / sr_out
        line["api"] = {"call": "infer_batch_process((wav, 24000), ref_text, [gen_text], model, vocoder, nfe_step=32, cfg_strength=2, sway_sampling_coef=5, "
                               "use_acc_grl=True, use_prosody_encoder=True)", "depth": darch.depth, "prompt_seconds": sec, "generated_seconds": asec,
                       "frames_total": int(spec_out.shape[1]) + int(w24.shape[1]) // 256, "median_ms": med * 1e3, "best_ms": best * 1e3,
                       "audio_seconds_per_second": asec / med,
                       "note": "host wall time per call, prompt handed over as a HOST tensor (H2D included), output as a host numpy array"}

    if args.cpu_baseline:
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        from oracle import prosody_oracle as PO          # test infrastructure, timed as the CPU baseline, never shipped
        from oracle.lemas_oracle import vocos_mel_spectrogram as omel
        wc = w16.cpu()
        t0 = time.perf_counter(); fbc = PO.kaldi_fbank_80(wc); tf = time.perf_counter() - t0
        net = PO.OracleECAPA(psd, arch)
        net.forward(fbc[None])
        t0 = time.perf_counter(); net.forward(fbc[None]); te = time.perf_counter() - t0
        t0 = time.perf_counter(); omel(w24.cpu()); tm = time.perf_counter() - t0
        line["cpu_baseline"] = {"kind": "port", "cores": torch.get_num_threads(), "fbank_ms": tf * 1e3, "encode_ms": te * 1e3, "mel_ms": tm * 1e3,
                                "sample": "the same prompt once through the oracle restatements (torch CPU fp32)"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
