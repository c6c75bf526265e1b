"""CPU tier: the prosody-encoder oracle (oracle/prosody_oracle.py) against the golden vectors the
reference's ECAPA_TDNN produced (oracle/gen_golden.py), and the kaldi-fbank restatement against an explicit DFT and
analytic properties (torchaudio is absent: that part is parity-unpinned, as the oracle header says)."""
import math
import os

import numpy as np
import pytest
import torch

from lemas_tts_amd import synth
from lemas_tts_amd.model.layout import ProsodyArch, prosody_param_shapes
from oracle import prosody_oracle as P


@pytest.mark.parametrize("name", ["prosody_enc_short", "prosody_enc_10s"])
def test_ecapa_oracle_matches_reference_golden(golden_dir, name):
    fx = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    arch = ProsodyArch()
    sd = synth.synth_prosody_encoder_state_dict(int(fx["wseed"]), arch)
    assert abs(synth.checksum(sd) - float(fx["wchecksum"])) < 1e-6 * abs(float(fx["wchecksum"])), "RNG drift"
    enc = P.OracleECAPA(sd, arch)
    for b in range(fx["fbank"].shape[0]):
        emb = enc.forward(torch.from_numpy(fx["fbank"][b: b + 1]))[0].numpy()
        np.testing.assert_allclose(emb, fx["emb"][b], atol=2e-6, rtol=0)
        assert abs(float(np.linalg.norm(emb)) - 1.0) < 1e-5


def test_param_layout_counts():
    shapes = prosody_param_shapes(ProsodyArch())
    assert shapes["blocks.0.conv.weight"] == (512, 80, 5)
    assert shapes["blocks.2.res2net_block.blocks.6.conv.weight"] == (64, 64, 3)
    assert shapes["asp.tdnn.conv.weight"] == (128, 4608, 1) and shapes["fc.weight"] == (512, 3072, 1)
    assert "blocks.1.shortcut.weight" not in shapes            # in == out channels: no projection shortcut (prosody_encoder.py:318-324)
    with pytest.raises(NotImplementedError):
        prosody_param_shapes(ProsodyArch(groups=(1, 2, 1, 1, 1)))


def test_kaldi_fbank_frames_and_explicit_dft():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(16000 + 123, generator=g) * 0.1
    fb = P.kaldi_fbank_80(x)
    assert fb.shape == (1 + (x.numel() - 400) // 160, 80)
    # frame 3 by hand: DC removal, pre-emphasis, povey window, explicit 512-point DFT, mel banks, log
    fr = x[3 * 160: 3 * 160 + 400].double()
    fr = fr - fr.mean()
    fr = fr - 0.97 * torch.cat([fr[:1], fr[:-1]])
    n = torch.arange(400, dtype=torch.float64)
    fr = fr * (0.5 - 0.5 * torch.cos(2 * math.pi * n / 399)).pow(0.85)
    k = torch.arange(257, dtype=torch.float64)[:, None] * torch.arange(400, dtype=torch.float64)[None]
    re, im = (torch.cos(2 * math.pi * k / 512) * fr).sum(1), (torch.sin(2 * math.pi * k / 512) * fr).sum(1)
    ref = torch.log(torch.clamp((re ** 2 + im ** 2) @ P.kaldi_mel_banks().double().T, min=float(torch.finfo(torch.float32).eps)))
    np.testing.assert_allclose(fb[3].double().numpy(), ref.numpy(), atol=2e-4)


def test_kaldi_mel_banks_shape_and_partition():
    fb = P.kaldi_mel_banks()
    assert fb.shape == (80, 257) and float(fb[:, 256].abs().max()) == 0.0
    # triangles overlap so that interior FFT bins (between the first and last centre) are covered with total weight 1
    tot = fb.sum(0)
    inner = (tot > 0.999) & (tot < 1.001)
    assert int(inner.sum()) > 200
    assert float(fb.min()) >= 0.0 and float(fb.max()) <= 1.0


def test_short_audio_is_tiled_and_tone_lands_in_the_right_bin():
    assert P.kaldi_fbank_80(torch.randn(150) * 0.1).shape[1] == 80          # < 400 samples: tiled, still yields frames
    t = torch.arange(16000, dtype=torch.float64) / 16000
    fb = P.kaldi_fbank_80(torch.sin(2 * math.pi * 1000.0 * t).float())
    peak = int(fb.mean(0).argmax())
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    expect = (mel(1000.0) - mel(20.0)) / ((mel(8000.0) - mel(20.0)) / 81) - 1
    assert abs(peak - expect) <= 1.0
