"""cProfile of the mirrored call surface (infer_batch_process from a raw host prompt, 22 blocks, NFE 32) on the GPU box: where the host time of
one call goes (the call is synchronous -- it returns a numpy waveform -- so host work in front of the first launch is on the critical path)."""
import cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lemas_tts_amd import synth
from lemas_tts_amd.infer.utils_infer import infer_batch_process, load_vocoder
from lemas_tts_amd.model.cfm import CFM
from lemas_tts_amd.model.layout import DiTArch, ProsodyArch
from lemas_tts_amd.model.prosody_encoder import ProsodyEncoder

dev = "cuda:0"
enc = ProsodyEncoder(state_dict=synth.synth_prosody_encoder_state_dict(42, ProsodyArch()), arch=ProsodyArch(), device=dev)
darch = DiTArch()
vocab = {f"p{i}": i for i in range(898)}
model = CFM(darch, 898, synth.synth_cfm_state_dict(darch, 898, 11, prosody=True), vocab_char_map=vocab, device=dev, use_prosody_encoder=True, prosody_encoder=enc)
vocoder = load_vocoder("vocos", device=dev, state_dict=synth.synth_vocos_state_dict(12))
n = 240000
t = torch.arange(n) / 24000.0
wav = (0.1 * torch.sin(2 * np.pi * 180.0 * t) + 0.02 * torch.randn(n))[None]
ref_text = [f"p{i}" for i in synth.synth_tokens(13, 100, 898)]
gen = [[f"p{i}" for i in synth.synth_tokens(14, 100, 898)]]


def call():
    return next(infer_batch_process((wav, 24000), ref_text, gen, model, vocoder, nfe_step=32, cfg_strength=2.0, sway_sampling_coef=5,
                                    use_acc_grl=True, use_prosody_encoder=True, ref_ratio=1, seed=3))


for _ in range(3):
    call()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); call(); ts.append(1e3 * (time.perf_counter() - t0))
print("ms per call:", " ".join(f"{v:.2f}" for v in ts), " torch threads", torch.get_num_threads())
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    call()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:40]))
