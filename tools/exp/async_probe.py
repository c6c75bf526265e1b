"""Does CFM.sample return to the host while the GPU still runs the previous utterance?  (bench.py's ms per utterance is 0.7 ms above its
hoists + step loop: GPU idle between utterances would be host work that cannot overlap.)  Host return times of back-to-back calls."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lemas_tts_amd import synth
from lemas_tts_amd.model.cfm import CFM
from lemas_tts_amd.model.layout import DiTArch

dev = "cuda:0"
arch = DiTArch()
model = CFM(arch, 898, synth.synth_cfm_state_dict(arch, 898, 11), device=dev)
F_, N = 938, 1875
cond = torch.from_numpy(synth.synth_cond_mel(1, F_))[None].to(dev)
text = torch.from_numpy(synth.synth_tokens(2, 200, 898))[None].long().pin_memory()
y0 = torch.from_numpy(synth.synth_noise(3, N))[None].to(dev)


def call():
    return model.sample(cond, text, N, steps=32, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False)[0]


for _ in range(2):
    call()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    call(); t1 = time.perf_counter()
    call(); t2 = time.perf_counter()
    call(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f"host return of three back-to-back sample() calls: {1e3 * (t1 - t0):7.2f} {1e3 * (t2 - t1):7.2f} {1e3 * (t3 - t2):7.2f} ms; "
          f"synchronize after them {1e3 * (t4 - t3):7.2f} ms; total {1e3 * (t4 - t0):7.2f} ms = {1e3 * (t4 - t0) / 3:7.2f} per utterance")
# the same three with a synchronize after each: what a fully serial host costs
t0 = time.perf_counter()
for _ in range(3):
    call(); torch.cuda.synchronize()
print(f"with a synchronize after each: {1e3 * (time.perf_counter() - t0) / 3:7.2f} ms per utterance")
