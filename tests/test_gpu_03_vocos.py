"""GPU tier: Vocos decode through the C ABI against the oracle restatement (which itself is checked against
torch.istft on CPU).  Everything is fp32 on the GPU (f32 MFMA), so the waveform tolerance is tight."""
import numpy as np
import pytest
import torch

from lemas_tts_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,L", [(1, 2), (1, 64), (2, 301), (1, 939)])
def test_vocos_decode_vs_oracle(B, L):
    from lemas_tts_amd.engine import VocosEngine
    from oracle import lemas_oracle as O
    sd = synth.synth_vocos_state_dict(7)
    mel = torch.from_numpy(np.stack([synth.synth_cond_mel(100 + b, L, "vmel").T for b in range(B)]))
    ref = O.OracleVocos(sd).decode(mel)
    eng = VocosEngine(sd, device="cuda:0")
    wav = eng.decode(mel).cpu()
    assert wav.shape == ref.shape == (B, 256 * (L - 1))
    err = (wav - ref).abs().max().item()
    print(f"\n[vocos B={B} L={L}] max|err| {err:.3e}  |ref| max {ref.abs().max():.3e}")
    assert err < 1e-4 * max(1.0, ref.abs().max().item())
    half = eng.decode(mel, gain=0.5).cpu()
    np.testing.assert_allclose(half.numpy(), 0.5 * wav.numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("seed", list(range(8)))
def test_randomised_lengths_vocoder_mel_resample_prosody(seed):
    """odd lengths through the launch-bound side kernels: vocoder frames, prompt samples, resample ratios, fbank frames"""
    import numpy as np
    from lemas_tts_amd.engine import MelEngine, ResampleEngine, VocosEngine
    from lemas_tts_amd.model.layout import ProsodyArch
    from lemas_tts_amd.model.prosody_encoder import ProsodyEncoder
    from oracle import lemas_oracle as O
    from oracle import prosody_oracle as P
    rng = np.random.default_rng(700 + seed)
    g = torch.Generator().manual_seed(seed)
    vsd = synth.synth_vocos_state_dict(40 + seed)
    B, L = int(rng.integers(1, 4)), int(rng.integers(2, 700))
    mel = torch.randn(B, 100, L, generator=g) * 2 - 3
    wav = VocosEngine(vsd, device="cuda:0").decode(mel).cpu()
    wref = O.OracleVocos(vsd).decode(mel)
    assert wav.shape == wref.shape == (B, 256 * (L - 1))
    assert float((wav - wref).abs().max()) < 1e-4 * max(1.0, float(wref.abs().max()))
    nw = int(rng.integers(600, 60000))
    x = torch.randn(B, nw, generator=g) * 0.1
    m = MelEngine(device="cuda:0").frames_first(x).cpu()
    mref = O.vocos_mel_spectrogram(x).permute(0, 2, 1)
    assert m.shape == mref.shape and float((m - mref).abs().max()) < 1e-3
    sr = int(rng.choice([8000, 11025, 16000, 22050, 32000, 44100, 48000]))
    r = ResampleEngine(sr, 24000, device="cuda:0")(x).cpu()
    rref = O.resample_sinc_hann(x, sr, 24000)
    assert r.shape == rref.shape and float((r - rref).abs().max()) < 1e-5
    arch = ProsodyArch(channels=(64, 64, 64, 128), kernel_sizes=(5, 3, 3, 1), dilations=(1, 2, 3, 1), attention_channels=16, res2net_scale=4,
                       se_channels=8, groups=(1, 1, 1, 1), embed_dim=32)
    psd = synth.synth_prosody_encoder_state_dict(50 + seed, arch)
    T = int(rng.integers(1, 500))
    fb = torch.from_numpy(synth.synth_fbank(60 + seed, T))[None]
    emb = ProsodyEncoder(state_dict=psd, arch=arch, device="cuda:0")(fb).cpu()
    assert float((emb - P.OracleECAPA(psd, arch).forward(fb)).abs().max()) < 2e-5


def test_vocos_frames_first_view_graph_replay_and_cache():
    """round 4: (1) a permute(0, 2, 1) VIEW of frames-first rows -- how every caller of the reference builds the argument
    (utils_infer.py:546-549) -- is decoded in place (lemas_vocos_decode_rows) and gives the bits of the contiguous [B, C, L] form;
    (2) the hipGraph the backbone + head is replayed from (captured the second time a shape is seen) gives the eager bits;
    (3) the shape cache is bounded and survives eviction."""
    from lemas_tts_amd.engine import VocosEngine
    sd = synth.synth_vocos_state_dict(9)
    eng = VocosEngine(sd, device="cuda:0")
    g = torch.Generator().manual_seed(3)
    out = (torch.randn(2, 400, 100, generator=g) * 2 - 3).cuda()            # sampler output layout [B, N, mel]
    view = out[:, 120:, :].permute(0, 2, 1)                                  # frames 120.. of both samples, not contiguous
    assert not view.is_contiguous()
    eng.set_option("graph", 0)
    eager = eng.decode(view.contiguous()).cpu().numpy()
    np.testing.assert_array_equal(eng.decode(view).cpu().numpy(), eager)     # strided rows == contiguous mel
    one = eng.decode(out[1:2, 120:, :].permute(0, 2, 1)).cpu().numpy()
    np.testing.assert_array_equal(one[0], eager[1])                          # a batch-mate does not change a sample
    eng.set_option("graph", 1)
    for _ in range(3):                                                       # first sight eager, then captured, then replayed
        np.testing.assert_array_equal(eng.decode(view).cpu().numpy(), eager)
    eng.set_option("graph_cache", 2)
    for L in (50, 60, 70, 50, 60, 70):                                       # three shapes, twice each, through a cache of two
        w = eng.decode(out[:1, :L, :].permute(0, 2, 1)).cpu().numpy()
        eng.set_option("graph", 0)
        np.testing.assert_array_equal(w, eng.decode(out[:1, :L, :].permute(0, 2, 1)).cpu().numpy())
        eng.set_option("graph", 1)
    np.testing.assert_array_equal(eng.decode(view).cpu().numpy(), eager)
    with pytest.raises(Exception):
        eng.set_option("no_such_option", 1)
