"""The north-star entry point end to end (scripts/tts_multilingual.py): files on disk in, a wav file out,
through the mirrored command line -> TTS -> infer_process -> CFM.sample -> Vocos on the MI355X engines."""
import os
import wave

import numpy as np
import pytest
import torch
import yaml

from lemas_tts_amd import synth
from lemas_tts_amd.model.layout import DiTArch

pytestmark = pytest.mark.gpu


def _assets(tmp_path, depth, vocab_size=60):
    from safetensors.torch import save_file
    root = tmp_path / "pretrained_models"
    (root / "ckpts" / "multilingual_grl").mkdir(parents=True)
    (root / "data" / "multilingual_grl").mkdir(parents=True)
    sd = synth.synth_cfm_state_dict(DiTArch(depth=depth), vocab_size, 71)
    save_file({"ema_model." + k: torch.from_numpy(v.copy()).contiguous() for k, v in sd.items()},
              str(root / "ckpts" / "multilingual_grl" / "multilingual_grl.safetensors"))
    (root / "data" / "multilingual_grl" / "vocab.txt").write_text("".join(f"p{i}\n" for i in range(vocab_size)), encoding="utf-8")
    vdir = root / "ckpts" / "vocos-mel-24khz"
    vdir.mkdir()
    torch.save({k: torch.from_numpy(v.copy()) for k, v in synth.synth_vocos_state_dict(72).items()}, str(vdir / "pytorch_model.bin"))
    (vdir / "config.yaml").write_text(yaml.safe_dump({
        "backbone": {"class_path": "vocos.models.VocosBackbone", "init_args": {"input_channels": 100, "dim": 512, "intermediate_dim": 1536, "num_layers": 8}},
        "head": {"class_path": "vocos.heads.ISTFTHead", "init_args": {"dim": 512, "n_fft": 1024, "hop_length": 256, "padding": "center"}}}))
    return root


def test_cli_writes_the_wave_tts_infer_returns(tmp_path, monkeypatch):
    import lemas_tts_amd.api as A
    import lemas_tts_amd.scripts.tts_multilingual as M
    from lemas_tts_amd.infer.audio_io import load_wav, save_wav
    depth = 2
    root = _assets(tmp_path, depth)
    monkeypatch.setattr(M, "PRETRAINED_ROOT", root)
    monkeypatch.setattr(M, "CKPTS_ROOT", root / "ckpts")
    real_cfg = A.load_arch_config
    monkeypatch.setattr(A, "load_arch_config", lambda m: {**real_cfg(m), "arch": {**real_cfg(m)["arch"], "depth": depth}})
    # a 1.5 s stereo 16 kHz prompt: exercises the wav reader, the mono mix-down, the rms lift and the HIP resampler
    t = np.arange(int(1.5 * 16000)) / 16000.0
    rng = np.random.default_rng(73)
    prompt = 0.02 * np.stack([np.sin(2 * np.pi * 220 * t), np.sin(2 * np.pi * 330 * t)], axis=1) + 0.002 * rng.standard_normal((t.size, 2))
    save_wav(tmp_path / "ref.wav", prompt, 16000, "PCM_16")
    ref_ph = "|".join(f"p{i}" for i in synth.synth_tokens(74, 9, 60))
    lines = ["|".join(f"p{i}" for i in synth.synth_tokens(75 + k, 7 + 2 * k, 60)) for k in range(2)]
    out = tmp_path / "out.wav"
    rc = M.main(["--ref_audio", str(tmp_path / "ref.wav"), "--ref_phones", ref_ph, "--phones", "\\n".join(lines), "--output_wave", str(out),
                 "--nfe_step", "3", "--cfg_strength", "2.0", "--sway_sampling_coef", "5", "--seed", "1234", "--use_ema"])
    assert rc == 0
    with wave.open(str(out), "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate()) == (1, 2, 24000)
        n = f.getnframes()
    got, sr = load_wav(out)
    assert sr == 24000 and got.shape == (1, n) and n > 24000 // 4 and np.isfinite(got.numpy()).all() and got.abs().max() > 0

    # the same request through the API directly: the file holds exactly the waveform infer() returns, as 16-bit PCM
    tts = M.build_tts("multilingual_grl", M._resolve_ckpt("multilingual_grl", None), M._resolve_vocab("multilingual_grl", None), None,
                      True, None, False)
    wav, sr2, spec = tts.infer(ref_file=str(tmp_path / "ref.wav"), ref_text=ref_ph.split("|"), gen_text=[x.split("|") for x in lines],
                               nfe_step=3, cfg_strength=2.0, sway_sampling_coef=5.0, use_acc_grl=False, ref_ratio=1.0, seed=1234,
                               use_prosody_encoder=False, file_spec=str(tmp_path / "spec"))
    assert sr2 == 24000 and 100 in spec.shape
    np.testing.assert_array_equal(got.numpy()[0], (np.clip(np.rint(np.asarray(wav, np.float64) * 32768.0), -32768, 32767) / 32768.0).astype(np.float32))
    assert (tmp_path / "spec.npy").is_file() or (tmp_path / "spec").is_file() or (tmp_path / "spec.png").is_file()
    # a wav path and the loaded (audio, sr) pair are the same input
    wav2, _, _ = tts.infer(ref_file=load_wav(tmp_path / "ref.wav"), ref_text=ref_ph.split("|"), gen_text=[x.split("|") for x in lines],
                           nfe_step=3, cfg_strength=2.0, sway_sampling_coef=5.0, use_acc_grl=False, ref_ratio=1.0, seed=1234,
                           use_prosody_encoder=False)
    np.testing.assert_array_equal(wav, wav2)


def test_string_frontend_needs_a_registered_factory(tmp_path, monkeypatch):
    """api.py:140-151 builds TextNorm(dtype=frontend) from the reference's frontend package; the product does not import that
    package: a string frontend asks the registered factory, and without one it is a loud TypeError."""
    import lemas_tts_amd.api as A
    real_cfg = A.load_arch_config
    monkeypatch.setattr(A, "load_arch_config", lambda m: {**real_cfg(m), "arch": {**real_cfg(m)["arch"], "depth": 1}})
    vocab = {f"p{i}": i for i in range(40)}
    kw = dict(model="multilingual_grl", device="cuda:0", state_dict=synth.synth_cfm_state_dict(DiTArch(depth=1), 40, 5),
              vocoder_state_dict=synth.synth_vocos_state_dict(6), vocab_char_map=vocab, frontend="phone")
    monkeypatch.setattr(A, "FRONTEND_FACTORY", None)
    with pytest.raises(TypeError, match="set_frontend_factory"):
        A.TTS(**kw)

    class _FE:
        def __init__(self, dtype):
            self.dtype = dtype
    A.set_frontend_factory(_FE)
    try:
        tts = A.TTS(**kw)
        assert isinstance(tts.frontend, _FE) and tts.frontend.dtype == "phone"
    finally:
        A.set_frontend_factory(None)


def test_cli_default_usage_reaches_the_string_frontend_through_a_named_factory(tmp_path, monkeypatch):
    """The reference's default usage is ``--ref_text ... --text ...`` with ``--frontend phone`` (scripts/tts_multilingual.py:175-296,
    api.py:140-151).  The text frontend is host Python outside this package, so the entry point takes its factory by NAME
    (``--frontend_factory module:callable`` or LEMAS_FRONTEND_FACTORY): with one, plain text in -> wav out; the result is what the same
    phones give through ``--ref_phones/--phones``."""
    import sys
    import lemas_tts_amd.api as A
    import lemas_tts_amd.scripts.tts_multilingual as M
    from lemas_tts_amd.infer.audio_io import load_wav, save_wav
    depth = 1
    root = _assets(tmp_path, depth)
    monkeypatch.setattr(M, "PRETRAINED_ROOT", root)
    monkeypatch.setattr(M, "CKPTS_ROOT", root / "ckpts")
    real_cfg = A.load_arch_config
    monkeypatch.setattr(A, "load_arch_config", lambda m: {**real_cfg(m), "arch": {**real_cfg(m)["arch"], "depth": depth}})
    monkeypatch.setattr(A, "FRONTEND_FACTORY", None)
    # a stand-in frontend in its own module: words -> phone tokens of the synthetic vocabulary ("p<len(word)>" per word, "p1" for '.')
    (tmp_path / "fake_frontend.py").write_text(
        "class Norm:\n"
        "    def __init__(self, dtype):\n"
        "        self.dtype = dtype\n"
        "    def text2phn(self, text):\n"
        "        toks = []\n"
        "        for w in text.replace('.', ' . ').split():\n"
        "            toks.append('p1' if w == '.' else 'p%d' % (2 + len(w) % 50))\n"
        "        return '|'.join(toks)\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    t = np.arange(int(1.2 * 24000)) / 24000.0
    save_wav(tmp_path / "ref.wav", 0.05 * np.sin(2 * np.pi * 200 * t)[:, None], 24000, "PCM_16")
    common = ["--ref_audio", str(tmp_path / "ref.wav"), "--nfe_step", "2", "--cfg_strength", "2.0", "--sway_sampling_coef", "5",
              "--seed", "7", "--use_ema"]
    try:
        # no factory anywhere: the reference's default command line fails loudly, naming the remedy
        monkeypatch.delenv("LEMAS_FRONTEND_FACTORY", raising=False)
        with pytest.raises(TypeError, match="frontend_factory"):
            M.main(common + ["--ref_text", "hello there", "--text", "good morning everyone", "--output_wave", str(tmp_path / "x.wav")])
        rc = M.main(common + ["--ref_text", "hello there", "--text", "good morning everyone", "--output_wave", str(tmp_path / "a.wav"),
                              "--frontend_factory", "fake_frontend:Norm"])
        assert rc == 0
        A.set_frontend_factory(None)
        monkeypatch.setenv("LEMAS_FRONTEND_FACTORY", "fake_frontend:Norm")          # the environment form
        assert M.main(common + ["--ref_text", "hello there", "--text", "good morning everyone", "--output_wave", str(tmp_path / "b.wav")]) == 0
    finally:
        A.set_frontend_factory(None)
        sys.modules.pop("fake_frontend", None)
    import fake_frontend
    fe = fake_frontend.Norm("phone")
    rc = M.main(common + ["--ref_phones", fe.text2phn("hello there. "), "--phones", fe.text2phn("good morning everyone. "),
                          "--output_wave", str(tmp_path / "c.wav")])
    assert rc == 0
    a, b, c = (load_wav(tmp_path / n)[0].numpy() for n in ("a.wav", "b.wav", "c.wav"))
    assert a.shape[1] > 2400 and np.array_equal(a, b) and np.array_equal(a, c)


def test_cli_denoise_runs_the_uvr5_shell_with_a_named_network(tmp_path, monkeypatch):
    """--denoise (tts_multilingual.py:303-314): the prompt goes through the UVR5 shell (lemas_tts_amd/uvr5) before TTS.infer; the network
    here is a user-supplied callable named with --denoise_model_factory (the built-in HIP network from the reference's directory layout:
    tests/test_mdxnet.py::test_cli_denoise_runs_from_the_reference_directory_layout).  With an identity network the shell only band-limits and resamples the prompt, so the output
    exists and differs from the undenoised run only through that; the temporary file is removed."""
    import glob
    import tempfile
    import lemas_tts_amd.api as A
    import lemas_tts_amd.scripts.tts_multilingual as M
    from lemas_tts_amd.infer.audio_io import load_wav, save_wav
    depth = 1
    root = _assets(tmp_path, depth)
    monkeypatch.setattr(M, "PRETRAINED_ROOT", root)
    monkeypatch.setattr(M, "CKPTS_ROOT", root / "ckpts")
    real_cfg = A.load_arch_config
    monkeypatch.setattr(A, "load_arch_config", lambda m: {**real_cfg(m), "arch": {**real_cfg(m)["arch"], "depth": depth}})
    (tmp_path / "fake_mdx.py").write_text("def build():\n    return lambda spek: spek\n")
    (tmp_path / "mdx.json").write_text('{"mdx_n_fft_scale_set": 2048, "mdx_dim_f_set": 768, "mdx_dim_t_set": 5, "compensate": 1.0}')
    monkeypatch.syspath_prepend(str(tmp_path))
    t = np.arange(int(1.2 * 22050)) / 22050.0
    save_wav(tmp_path / "ref.wav", 0.05 * np.sin(2 * np.pi * 300 * t), 22050, "PCM_16")
    ref_ph = "|".join(f"p{i}" for i in synth.synth_tokens(80, 8, 60))
    gen_ph = "|".join(f"p{i}" for i in synth.synth_tokens(81, 9, 60))
    before = set(glob.glob(os.path.join(tempfile.gettempdir(), "*.wav")))
    common = ["--ref_audio", str(tmp_path / "ref.wav"), "--ref_phones", ref_ph, "--phones", gen_ph, "--nfe_step", "2", "--cfg_strength", "2.0",
              "--sway_sampling_coef", "5", "--seed", "7", "--use_ema"]
    with pytest.raises(FileNotFoundError, match="--denoise: .*uvr5 does not exist"):      # no pretrained_models/uvr5, no --denoise_model: named, not worked around
        M.main(common + ["--denoise", "--output_wave", str(tmp_path / "x.wav")])
    assert M.main(common + ["--denoise", "--denoise_model_factory", "fake_mdx:build", "--denoise_config", str(tmp_path / "mdx.json"),
                            "--output_wave", str(tmp_path / "den.wav")]) == 0
    assert M.main(common + ["--output_wave", str(tmp_path / "plain.wav")]) == 0
    den, sr = load_wav(tmp_path / "den.wav")
    plain, _ = load_wav(tmp_path / "plain.wav")
    assert sr == 24000 and den.shape[1] > 2000 and np.isfinite(den.numpy()).all() and den.abs().max() > 0
    assert abs(den.shape[1] - plain.shape[1]) <= 256 * 2          # same prompt length up to the 44.1 kHz round trip
    assert set(glob.glob(os.path.join(tempfile.gettempdir(), "*.wav"))) == before, "the denoised temporary file must be removed"
