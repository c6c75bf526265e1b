"""SURVEY.md 8f-4, the UVR5 MDX-Net denoising SHELL (everything around the ONNX network, which is not in the reference tree).

CPU tier: the oracle (oracle/uvr5_oracle.py) against vectors the REFERENCE's own ``Inference`` class produced (oracle/gen_golden_uvr5.py
-> tests/golden/uvr5_shell.npz).  GPU tier: the HIP transforms (lemas_stft_*) and the mirror (lemas_tts_amd/uvr5/mdx.py) against the same
vectors.  The network in all of them is the fixtures' deterministic stand-in."""
import os

import numpy as np
import pytest
import torch

from oracle import uvr5_oracle as U

CASES = ["plain", "denoise_b2", "exact_multiple", "match_mix", "short"]


@pytest.fixture(scope="module")
def fx(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "uvr5_shell.npz")))


def _oracle(fx, name):
    den, bs, match = (int(v) for v in fx[f"{name}_cfg"])
    o = U.ShellOracle(int(fx["n_fft"]), int(fx["dim_f"]), int(fx["dim_t_set"]), bool(den), bs, margin=441)
    o.model_run = U.fake_network(fx["gain"])
    return o, bool(match)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_the_reference_shell(fx, name):
    o, match = _oracle(fx, name)
    out = o.demix_base({0: torch.from_numpy(fx[f"{name}_mix"])}, is_match_mix=match)
    assert out.shape == fx[f"{name}_out"].shape
    np.testing.assert_allclose(out.numpy(), fx[f"{name}_out"], atol=2e-6, rtol=0)


def test_oracle_transforms_and_chunking(fx):
    o, _ = _oracle(fx, "plain")
    waves, pad = o.initialize_mix(torch.from_numpy(fx["plain_mix"]))
    assert pad == int(fx["pad"][0]) and waves.shape[1:] == (2, o.chunk_size)
    np.testing.assert_array_equal(waves[:2].numpy(), fx["stft_in"])
    spek = o.stft(waves[:2])
    np.testing.assert_allclose(spek.numpy(), fx["stft_out"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(o.istft(spek).numpy(), fx["istft_out"], atol=2e-6, rtol=0)
    # a mix that is a whole number of pieces still gets a full extra (all-padding) piece, as the reference does
    w2, pad2 = o.initialize_mix(torch.zeros(2, 2 * o.gen_size))
    assert pad2 == o.gen_size and w2.shape[0] == 3


# ------------------------------------------------------------------------------------------------------------------ GPU
def _mirror(fx, name):
    from lemas_tts_amd.uvr5 import Inference, MDXConfig
    den, bs, match = (int(v) for v in fx[f"{name}_cfg"])
    cfg = MDXConfig(mdx_n_fft_scale_set=int(fx["n_fft"]), mdx_dim_f_set=int(fx["dim_f"]), mdx_dim_t_set=int(fx["dim_t_set"]),
                    compensate=1.0, is_denoise=bool(den), mdx_batch_size=bs, margin=441)
    inf = Inference(cfg, "cuda:0")
    gain = torch.from_numpy(fx["gain"]).to("cuda:0")
    inf.load_model(lambda x: x * gain + 0.3 * torch.tanh(x) + 0.05 * x * x)     # the stand-in network, on the device
    return inf, bool(match)


@pytest.mark.gpu
def test_hip_stft_and_istft_match_the_reference(fx):
    """lemas_stft_forward / lemas_stft_inverse against torch.stft / torch.istft as the reference calls them (:206-223): fp32 GEMMs
    over n_fft = 2048 terms of magnitude <= ~15 -> 2e-3 absolute on spectrogram values of rms ~10; the inverse to 2e-5."""
    inf, _ = _mirror(fx, "plain")
    spek = inf.stft(torch.from_numpy(fx["stft_in"]).to("cuda:0"))
    assert tuple(spek.shape) == fx["stft_out"].shape
    np.testing.assert_allclose(spek.cpu().numpy(), fx["stft_out"], atol=2e-3, rtol=0)
    wav = inf.istft(torch.from_numpy(fx["stft_out"]).to("cuda:0"))
    np.testing.assert_allclose(wav.cpu().numpy(), fx["istft_out"], atol=2e-5, rtol=0)
    # inverse(forward(x)) = x away from the reflected edges
    x = torch.from_numpy(fx["stft_in"]).to("cuda:0")
    rt = inf._stft.inverse(inf._stft.forward(x.reshape(-1, inf.chunk_size)))
    assert float((rt - x.reshape(-1, inf.chunk_size)).abs().max()) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_mirror_matches_the_reference_shell(fx, name):
    inf, match = _mirror(fx, name)
    out = inf.demix_base({0: torch.from_numpy(fx[f"{name}_mix"])}, is_match_mix=match)
    assert tuple(out.shape) == fx[f"{name}_out"].shape
    err = float(np.abs(out.cpu().numpy() - fx[f"{name}_out"]).max())
    print(f"\n[uvr5 shell {name}] max|err| {err:.2e} (|out| rms {float(np.sqrt((fx[f'{name}_out'] ** 2).mean())):.3f})")
    assert err < 5e-5


@pytest.mark.gpu
def test_uvr5_wrapper_denoises_a_file(fx, tmp_path):
    """UVR5.denoise_file (tts_multilingual.py:73-86): mono 24 kHz file -> stereo, 44.1 kHz, shell, 24-bit temporary wav"""
    from lemas_tts_amd.infer import audio_io
    from lemas_tts_amd.uvr5 import MDXConfig, UVR5
    cfg = MDXConfig(mdx_n_fft_scale_set=2048, mdx_dim_f_set=768, mdx_dim_t_set=4, compensate=1.0, margin=441)
    uv = UVR5(lambda x: x, cfg, device="cuda:0")                 # identity network: the shell alone
    t = torch.arange(24000, dtype=torch.float32) / 24000.0
    wav = (0.4 * torch.sin(2 * np.pi * 440.0 * t))[None]
    src = tmp_path / "prompt.wav"
    audio_io.save_wav(str(src), wav.numpy().T, 24000)
    out_path = uv.denoise_file(str(src))
    try:
        y, sr = audio_io.load_wav(out_path)
    finally:
        os.remove(out_path)
    assert sr == 44100 and y.shape == (2, 44100)
    # identity network + band crop at dim_f (768 of 1025 bins = 16.5 kHz) leaves a 440 Hz tone untouched: compare with the resampled input
    from lemas_tts_amd.engine import resampler
    ref = resampler(24000, 44100)(wav.to("cuda:0")).cpu()
    assert float((y[0, 2000:-2000] - ref[0, 2000:-2000]).abs().max()) < 2e-3
    assert torch.equal(y[0], y[1])


def test_path_as_model_is_refused():
    from lemas_tts_amd.uvr5 import mdx
    with pytest.raises(Exception):
        mdx.Inference.load_model(object.__new__(mdx.Inference), "Kim_Vocal_1.onnx")


@pytest.mark.gpu
@pytest.mark.parametrize("n_fft,hop", [(64, 16), (96, 24), (20, 5), (240, 60), (1000, 250), (1536, 384), (2048, 512), (6144, 1024), (7680, 1024),
                                       (8192, 2048), (28, 7), (1792, 448)])
def test_stft_engine_fft_and_gemm_forms_vs_torch(n_fft, hop):
    """lemas_stft_forward / _inverse against torch.stft / torch.istft (the reference's arithmetic, multiprocess_cuda_infer.py:206-223) at
    transform lengths the in-LDS FFT takes (2^a 3^b 5^c <= 8192: radix-4 / 2 / 3 / 5 stages in every mix, Kim_Vocal_1's 7680) and at lengths it
    does not (28, 1792: a factor 7 -- the DFT as a GEMM); an ODD number of rows too (the FFT rides two real rows on one complex transform)."""
    from lemas_tts_amd.engine import StftEngine
    g = torch.Generator().manual_seed(n_fft)
    win = torch.hann_window(n_fft, periodic=True)
    eng = StftEngine(n_fft, hop, win, device="cuda:0")
    for batch in (3, 2):
        frames = 9
        x = torch.randn(batch, hop * (frames - 1), generator=g)
        ref = torch.stft(x, n_fft=n_fft, hop_length=hop, window=win, center=True, return_complex=True)
        spec = eng.forward(x.to("cuda:0")).cpu()
        assert spec.shape == ref.shape
        scale = float(ref.abs().max())
        assert float((spec - ref).abs().max()) < 2e-6 * scale * max(1.0, np.log2(n_fft) / 4), (n_fft, float((spec - ref).abs().max()), scale)
        wav_ref = torch.istft(ref, n_fft=n_fft, hop_length=hop, window=win, center=True)
        wav = eng.inverse(ref.to("cuda:0")).cpu()
        assert wav.shape == wav_ref.shape
        assert float((wav - wav_ref).abs().max()) < 1e-5, (n_fft, float((wav - wav_ref).abs().max()))
        # irfft ignores the imaginary parts of the DC and Nyquist bins: so must both forms
        dirty = ref.clone()
        dirty[:, 0, :] += 0.5j
        dirty[:, -1, :] -= 0.25j
        wav2 = eng.inverse(dirty.to("cuda:0")).cpu()
        assert float((wav2 - wav_ref).abs().max()) < 1e-5


@pytest.mark.gpu
def test_two_stft_engines_of_different_lengths_side_by_side():
    """the FFT kernels' > 64 KB LDS opt-in belongs to the kernel, not to an engine: creating a short-transform engine after a long one must not
    break the long one's launches (and the other way round)"""
    from lemas_tts_amd.engine import StftEngine
    g = torch.Generator().manual_seed(1)
    big = StftEngine(7680, 1024, torch.hann_window(7680, periodic=False), device="cuda:0")
    small = StftEngine(64, 16, torch.hann_window(64, periodic=True), device="cuda:0")
    big2 = StftEngine(8192, 2048, torch.hann_window(8192, periodic=True), device="cuda:0")
    for eng, n_fft, hop, periodic in ((big, 7680, 1024, False), (small, 64, 16, True), (big2, 8192, 2048, True), (big, 7680, 1024, False)):
        win = torch.hann_window(n_fft, periodic=periodic)
        x = torch.randn(2, hop * 8, generator=g)
        ref = torch.stft(x, n_fft=n_fft, hop_length=hop, window=win, center=True, return_complex=True)
        spec = eng.forward(x.to("cuda:0")).cpu()
        assert float((spec - ref).abs().max()) < 1e-5 * float(ref.abs().max())
        rt = eng.inverse(spec.to("cuda:0")).cpu()
        assert float((rt - x).abs().max()) < 2e-5
