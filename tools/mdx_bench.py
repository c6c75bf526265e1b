#!/usr/bin/env python
"""Time the MDX-Net forward (lemas_mdx_forward) at the Kim_Vocal_1 shape and print one JSON line with the roofline of the whole
forward (algorithmic FLOPs / time against the exact-fp32 MFMA peak, 157.3 TFLOP/s: MI355X_MICROARCH.md).

    python tools/mdx_bench.py [--batch 1] [--iters 10] [--small]

Synthetic seeded weights (oracle/mdx_oracle.seeded_state_dict is TEST infrastructure; a tool may use it)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mdx_oracle as MO   # noqa: E402
from lemas_tts_amd.engine import MdxEngine   # noqa: E402

F32_MFMA_PEAK = 157.3e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--small", action="store_true", help="dim_f 768, dim_t 64 (a quick run under a profiler)")
    args = ap.parse_args()
    arch = MO.MdxArch(dim_f=768, dim_t=64) if args.small else MO.KIM_VOCAL_1
    eng = MdxEngine(arch, MO.seeded_state_dict(arch, 20))
    x = torch.from_numpy(MO.seeded_input(arch, args.batch, 21)).to("cuda:0")
    for _ in range(2):
        eng.forward(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        eng.forward(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    fl = eng.flops(args.batch)
    print(json.dumps({"workload": f"mdxnet [{args.batch}, {arch.dim_c}, {arch.dim_f}, {arch.dim_t}] g{arch.g} n{arch.n}", "ms_per_forward": dt * 1e3,
                      "flops": fl, "tflops": fl / dt / 1e12, "frac_of_f32_mfma_peak": fl / dt / F32_MFMA_PEAK,
                      "audio_seconds_per_forward": args.batch * (1024 * (arch.dim_t - 1) - 7680) / 44100.0}))


if __name__ == "__main__":
    main()
