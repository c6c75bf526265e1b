"""Why did 20 back-to-back MelEngine.frames_first calls on one 10 s prompt take 3.8 ms each (r06fe) when 5 take 0.1 ms?  Per-call host times."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lemas_tts_amd.engine import MelEngine

mel = MelEngine(device="cuda:0")
w = (0.1 * torch.randn(1, 240000)).to("cuda:0")
for tag, sync_each in (("async", False), ("sync each", True)):
    for iters in (5, 20, 50):
        mel.frames_first(w); torch.cuda.synchronize()
        ts = []
        t00 = time.perf_counter()
        for _ in range(iters):
            t0 = time.perf_counter()
            mel.frames_first(w)
            if sync_each:
                torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e6)
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t00) * 1e6 / iters
        print(f"{tag:10s} iters {iters:3d}: {tot:8.1f} us per call; host per call min {min(ts):7.1f} median {sorted(ts)[len(ts)//2]:7.1f} max {max(ts):8.1f}; "
              f"reserved {torch.cuda.memory_reserved() >> 20} MiB")
