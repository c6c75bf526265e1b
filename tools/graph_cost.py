#!/usr/bin/env python
"""What a NEW sequence length costs the step loop (development aid; needs the MI355X).

The reference's caller produces a different duration for every gen_text line (lemas_tts/infer/utils_infer.py:520-542).  The engine caches
its step graphs per (batch, 128-row pitch) bucket; a never-seen frame count inside a seen bucket re-captures the launches and patches one
of the bucket's instantiated graphs (hipGraphExecUpdate).  Arms: cached length | new length, same bucket (update) | the same with
graph_update = 0 (instantiate) | new bucket | eager (no graph).  Prints wall ms per utterance (hoists + 32 steps, no vocoder)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from lemas_tts_amd import synth  # noqa: E402
from lemas_tts_amd.model.cfm import CFM  # noqa: E402
from lemas_tts_amd.model.layout import DiTArch  # noqa: E402

arch = DiTArch()
m = CFM(arch, 898, synth.synth_cfm_state_dict(arch, 898, 1234), device="cuda:0")
F_ = 400
cond = torch.from_numpy(synth.synth_cond_mel(1, F_))[None].cuda()
text = torch.from_numpy(synth.synth_tokens(2, 100, 898))[None]


def run(N):
    y0 = torch.from_numpy(synth.synth_noise(3, N))[None].cuda()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.sample(cond, text, N, steps=32, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


def stats():
    return {k: m.engine.stat("graph_" + k) for k in ("captures", "instantiates", "updates", "update_failures", "buckets")}


for base in (900, 1800):
    print(f"--- bucket of N = {base} (pitch {(base + 127) // 128 * 128})")
    run(base); run(base + 1)                      # warm: both slots of the bucket instantiated
    cached = min(run(base) for _ in range(3))
    new_upd = [run(base + 2 + i) for i in range(4)]
    m.engine.set_option("graph_update", 0)
    new_inst = [run(base + 10 + i) for i in range(4)]
    m.engine.set_option("graph_update", 1)
    m.engine.set_option("graph", 0)
    eager = min(run(base + 20) for _ in range(3))
    m.engine.set_option("graph", 1)
    nb = run(base + 200)                          # a bucket never seen
    fmt = lambda v: " ".join(f"{x:.1f}" for x in v)
    print(f"cached length            {cached:7.1f} ms")
    print(f"new length, patched      {fmt(new_upd)} ms   (+{100 * (min(new_upd) / cached - 1):.1f} % best, +{100 * (sorted(new_upd)[len(new_upd) // 2] / cached - 1):.1f} % median)")
    print(f"new length, instantiated {fmt(new_inst)} ms   (+{100 * (min(new_inst) / cached - 1):.1f} % best)")
    print(f"eager, no graph          {eager:7.1f} ms   (+{100 * (eager / cached - 1):.1f} %)")
    print(f"new bucket               {nb:7.1f} ms")
    print("counters", stats())
