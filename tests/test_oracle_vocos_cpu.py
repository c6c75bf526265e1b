"""CPU tier: the oracle's Vocos ISTFT head against an independent formulation (explicit inverse rDFT matrix +
overlap-add), i.e. the same algebra the HIP path uses.  Vocos itself is third-party and absent: parity unpinned."""
import numpy as np
import torch

from lemas_tts_amd import synth
from oracle import lemas_oracle as O


def test_istft_head_matches_explicit_dft_overlap_add():
    sd = synth.synth_vocos_state_dict(3)
    v = O.OracleVocos(sd)
    L, nfft, hop = 37, 1024, 256
    g = torch.Generator().manual_seed(0)
    h = torch.randn(1, L, 512, generator=g)
    ref = v.head(h)[0].double()
    o = torch.nn.functional.linear(h, v.p["head.out.weight"], v.p["head.out.bias"])[0].double()
    mag, ph = torch.exp(o[:, :513]).clip(max=100.0), o[:, 513:]
    re, im = mag * torch.cos(ph), mag * torch.sin(ph)
    n = torch.arange(nfft, dtype=torch.float64)
    k = torch.arange(513, dtype=torch.float64)
    ck = torch.full((513,), 2.0, dtype=torch.float64)
    ck[0] = ck[-1] = 1.0
    ang = 2 * np.pi * torch.outer(n, k) / nfft
    w = v.p["head.istft.window"].double()
    frames = (re @ (ck[:, None] * torch.cos(ang).T) - im @ (ck[:, None] * torch.sin(ang).T)) / nfft * w
    T = hop * (L - 1) + nfft
    y, env = torch.zeros(T, dtype=torch.float64), torch.zeros(T, dtype=torch.float64)
    for f in range(L):
        y[f * hop: f * hop + nfft] += frames[f]
        env[f * hop: f * hop + nfft] += w * w
    wav = (y / env)[nfft // 2: nfft // 2 + hop * (L - 1)]
    assert wav.shape == ref.shape
    assert (wav - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


def test_vocos_decode_shapes():
    sd = synth.synth_vocos_state_dict(3)
    mel = torch.from_numpy(synth.synth_cond_mel(1, 20, "m").T[None])
    assert O.OracleVocos(sd).decode(mel).shape == (1, 256 * 19)
