#!/usr/bin/env python
"""The roofline line of the UVR5 MDX-Net prompt denoiser: one JSON line.

    python tools/mdx_bench.py [--batch 2] [--iters 10] [--prompt-seconds 10] [--cpu-baseline]

* ``network``: lemas_mdx_forward at the Kim_Vocal_1 shape [batch, 4, 3072, 256], timed with HIP events on the engine's own stream around
  `iters` back-to-back forwards; ``roofline`` = algorithmic FLOPs (lemas_mdx_flops: 2 per multiply-add of every convolution and TDF linear,
  0.759 TFLOP per sample) / time against the exact-fp32 MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md).  batch 2 is what one chunk costs
  with the reference's ``is_denoise`` (+-input pair, multiprocess_cuda_infer.py:269).
* ``denoise``: UVR5.denoise on a synthetic `prompt-seconds` mono 24 kHz prompt, inputs resident on the device: resample to 44.1 kHz,
  chunking, STFT, network (+- pair), inverse STFT -- audio seconds per wall second.
* ``cpu_baseline`` (--cpu-baseline): the fp32 restatement (oracle/mdx_oracle.py, test infrastructure; torch CPU kernels) on one sample
  of the same shape on this box's host cores -- the reference itself ran 2.3 s per forward on 8 threads in the build container
  (tests/golden/mdxnet_kim.npz ref_seconds).
Synthetic seeded weights (lemas_tts_amd/synth.py synth_mdx_state_dict)."""
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
from lemas_tts_amd.engine import MdxEngine   # noqa: E402
from lemas_tts_amd.uvr5.arch import KIM_VOCAL_1, MdxArch, flops   # noqa: E402

F32_MFMA_PEAK = 157.3e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--prompt-seconds", type=float, default=10.0)
    ap.add_argument("--cpu-baseline", action="store_true")
    ap.add_argument("--bf16x3", type=int, default=0, help="engine option bf16x3: the 3x3 convolutions on split-bf16 operands")
    ap.add_argument("--products", type=int, default=0, help="engine option bf16x3_products (3 or 4 bf16 MFMAs per product)")
    ap.add_argument("--conv-chunk", type=int, default=0, help="engine option conv_chunk (4 or 8 channels per K chunk of the exact 3x3 kernel)")
    ap.add_argument("--small", action="store_true", help="dim_f 768, dim_t 64 (a quick run under a profiler)")
    args = ap.parse_args()
    arch = MdxArch(dim_f=768, dim_t=64) if args.small else KIM_VOCAL_1
    sd = synth.synth_mdx_state_dict(arch, 20)
    eng = MdxEngine(arch, sd, bf16x3=bool(args.bf16x3))
    if args.products:
        from lemas_tts_amd import _lib
        _lib.check(_lib.lib().lemas_mdx_set_option(eng._h, b"bf16x3_products", args.products), "bf16x3_products")
    if args.conv_chunk:
        from lemas_tts_amd import _lib
        _lib.check(_lib.lib().lemas_mdx_set_option(eng._h, b"conv_chunk", args.conv_chunk), "conv_chunk")
    x = torch.from_numpy(synth.synth_mdx_input(arch, args.batch, 21)).to("cuda:0")
    for _ in range(2):
        eng.forward(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(eng.stream)
    for _ in range(args.iters):
        eng.forward(x)
    e1.record(eng.stream)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.iters
    dt = e0.elapsed_time(e1) * 1e-3 / args.iters
    fl = eng.flops(args.batch)
    line = {"metric": "MDX-Net forward (UVR5 prompt denoiser)", "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ConvTDFNet [{args.batch}, {arch.dim_c}, {arch.dim_f}, {arch.dim_t}] g{arch.g} l{arch.l} n{arch.n} bn{arch.bn} "
                                   f"({sum(v.size for v in sd.values()) / 1e6:.1f} M values)"},
            "network": {"bf16x3": bool(args.bf16x3), "ms_per_forward": dt * 1e3, "wall_ms_per_forward": wall * 1e3, "batch": args.batch, "iters": args.iters},
            "roofline": {"bound": "mfma", "achieved": fl / dt / 1e12, "peak": F32_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": fl / dt / F32_MFMA_PEAK,
                         "flops_per_forward": fl, "peak_note": "exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), MI355X_MICROARCH.md"}}
    if not args.small and args.prompt_seconds > 0:
        from lemas_tts_amd.uvr5 import MDXConfig, UVR5
        cfg = MDXConfig(is_denoise=True, mdx_batch_size=1)
        uv = UVR5((arch, sd), cfg, device="cuda:0", bf16x3=bool(args.bf16x3))
        n = int(args.prompt_seconds * 24000)
        t = torch.arange(n, device="cuda:0") / 24000.0
        wav = (0.1 * torch.sin(2 * np.pi * 180.0 * t) + 0.01 * torch.randn(n, device="cuda:0"))[None]
        uv.denoise(wav, 24000)
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            out = uv.denoise(wav, 24000)
        torch.cuda.synchronize()
        dd = (time.perf_counter() - t0) / reps
        line["denoise"] = {"prompt_seconds": args.prompt_seconds, "ms": dd * 1e3, "audio_seconds_per_second": args.prompt_seconds / dd,
                           "chunks": int(-(-out.shape[1] // uv.model.gen_size)), "is_denoise": True, "out_samples_44k1": int(out.shape[1])}
    if args.cpu_baseline:
        torch.set_num_threads(min(16, os.cpu_count() or 1))       # (256 threads on the GPU box's host: 50 s per forward, 20x slower than 8 in the build container)
        from oracle import mdx_oracle as MO            # the CPU baseline IS the oracle (test infrastructure), timed, never shipped
        net = MO.MdxOracle(arch, sd)
        xc = synth.synth_mdx_input(arch, 1, 21)
        net.forward(xc)
        t0 = time.perf_counter()
        net.forward(xc)
        dc = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": flops(arch) / dc / 1e12, "unit": "TFLOP/s", "seconds_per_forward": dc, "cores": torch.get_num_threads(),
                                "kind": "port", "sample": "one forward of one sample, the same shape and weights (torch CPU fp32)"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
