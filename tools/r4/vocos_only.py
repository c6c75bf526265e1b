#!/usr/bin/env python
"""The vocoder alone, for profiler passes (rocprofv3 --pmc / --kernel-trace): N decodes of one utterance's generated frames (L = 938, the
configs[1] decode).  Prints the average decode time."""
import sys
import time

import torch

sys.path.insert(0, ".")
from lemas_tts_amd import synth  # noqa: E402
from lemas_tts_amd.engine import VocosEngine  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 938
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng = VocosEngine(synth.synth_vocos_state_dict(1234), device="cuda:0")
rows = torch.from_numpy(synth.synth_cond_mel(5, L + 10)).cuda()[None]           # [1, L + 10, 100] frames-first, as the sampler returns it
view = rows[:, 10:, :].permute(0, 2, 1)
for _ in range(3):
    eng.decode(view)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    eng.decode(view)
torch.cuda.synchronize()
print(f"vocoder L={L}: {1e3 * (time.perf_counter() - t0) / N:.3f} ms per decode ({N} decodes, graph replay)")
print(f"DECODES {N + 3}")
