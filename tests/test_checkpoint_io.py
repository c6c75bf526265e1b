"""Checkpoint layout: the reference's files must load unchanged.
CPU tier: key surgery of load_checkpoint (utils_infer.py:215-241) and the vocab reader (model/utils.py:121-126).
GPU tier: load_model / load_vocoder from files on disk, strict-load failures."""
import os

import numpy as np
import pytest
import torch

from lemas_tts_amd import synth
from lemas_tts_amd.model.layout import DiTArch, cfm_param_shapes


def _write_ckpt(tmp_path, arch, vocab_size, seed, kind="safetensors", ema=True, prosody=False):
    sd = synth.synth_cfm_state_dict(arch, vocab_size, seed, prosody=prosody)
    tens = {("ema_model." + k if ema else k): torch.from_numpy(v.copy()) for k, v in sd.items()}
    if ema:   # what an EMA training checkpoint carries besides the weights (utils_infer.py:223-235)
        tens["initted"] = torch.tensor(True)
        tens["step"] = torch.tensor(1234)
        tens["ema_model.mel_spec.mel_stft.mel_scale.fb"] = torch.zeros(513, 100)
        tens["ema_model.mel_spec.mel_stft.spectrogram.window"] = torch.zeros(1024)
        tens["ema_model.ctc.proj.0.weight"] = torch.zeros(4, 4)
        tens["ema_model.ctc.proj.0.bias"] = torch.zeros(4)
        tens["ema_model.ctc.ctc_proj.weight"] = torch.zeros(4, 4)
        tens["ema_model.ctc.ctc_proj.bias"] = torch.zeros(4)
    if kind == "safetensors":
        from safetensors.torch import save_file
        path = str(tmp_path / "model.safetensors")
        save_file({k: v.contiguous() for k, v in tens.items()}, path)
    else:
        path = str(tmp_path / "model.pt")
        torch.save({"ema_model_state_dict": tens} if ema else {"model_state_dict": tens}, path)
    return path, sd


@pytest.mark.parametrize("kind", ["safetensors", "pt"])
def test_read_checkpoint_key_surgery(tmp_path, kind):
    from lemas_tts_amd.infer.utils_infer import read_checkpoint
    arch = DiTArch(depth=1, conv_layers=1)
    path, sd = _write_ckpt(tmp_path, arch, 20, 3, kind)
    got = read_checkpoint(path, use_ema=True)
    assert set(got) == set(cfm_param_shapes(arch, 20))
    for k in sd:
        np.testing.assert_array_equal(got[k].numpy(), sd[k])


def test_get_tokenizer_custom(tmp_path):
    from lemas_tts_amd.infer.utils_infer import get_tokenizer
    p = tmp_path / "vocab.txt"
    p.write_text(" \na\n(en)b\n#1\n", encoding="utf-8")
    m, n = get_tokenizer(str(p), "custom")
    assert n == 4 and m == {" ": 0, "a": 1, "(en)b": 2, "#1": 3}


@pytest.mark.gpu
def test_load_model_and_vocoder_from_files(tmp_path):
    import yaml
    from lemas_tts_amd.infer.utils_infer import load_model, load_vocoder
    from oracle import lemas_oracle as O
    arch = DiTArch(depth=1)
    vocab_size = 30
    path, sd = _write_ckpt(tmp_path, arch, vocab_size, 7, "safetensors")
    (tmp_path / "vocab.txt").write_text("".join(f"t{i}\n" for i in range(vocab_size)), encoding="utf-8")
    model = load_model(None, dict(dim=1024, depth=1, heads=16, ff_mult=2, text_dim=512, conv_layers=4), path,
                       vocab_file=str(tmp_path / "vocab.txt"), use_ema=True, device="cuda:0")
    vdir = tmp_path / "vocos-mel-24khz"
    vdir.mkdir()
    vsd = synth.synth_vocos_state_dict(8)
    torch.save({k: torch.from_numpy(v.copy()) for k, v in vsd.items()} | {"feature_extractor.mel_spec.window": torch.zeros(1024)},
               str(vdir / "pytorch_model.bin"))
    (vdir / "config.yaml").write_text(yaml.safe_dump({
        "feature_extractor": {"class_path": "vocos.feature_extractors.MelSpectrogramFeatures", "init_args": {"sample_rate": 24000}},
        "backbone": {"class_path": "vocos.models.VocosBackbone", "init_args": {"input_channels": 100, "dim": 512, "intermediate_dim": 1536, "num_layers": 8}},
        "head": {"class_path": "vocos.heads.ISTFTHead", "init_args": {"dim": 512, "n_fft": 1024, "hop_length": 256, "padding": "center"}}}))
    voc = load_vocoder("vocos", True, str(vdir), "cuda:0")
    F_, N = 30, 80
    cond = torch.from_numpy(synth.synth_cond_mel(9, F_))[None]
    toks = [[f"t{i}" for i in synth.synth_tokens(10, 12, vocab_size)] + ["unknown-token"]]
    y0 = torch.from_numpy(synth.synth_noise(11, N))[None]
    out, _ = model.sample(cond, toks, N, steps=2, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False)
    vmap = {f"t{i}": i for i in range(vocab_size)}
    ref, _ = O.OracleCFM(sd, arch).sample(cond, O.tokens_to_idx(toks, vmap), N, y0=y0, steps=2, cfg_strength=2.0, sway_sampling_coef=5)
    assert float(((out.cpu() - ref)[:, F_:] ** 2).mean()) <= 1e-4
    wav = voc.decode(out[:, F_:, :].permute(0, 2, 1)).cpu()
    wref = O.OracleVocos(vsd).decode(out.cpu()[:, F_:, :].permute(0, 2, 1))
    assert (wav - wref).abs().max().item() < 1e-4 * max(1.0, wref.abs().max().item())


@pytest.mark.gpu
def test_strict_load_errors():
    """utils_infer.py:237 loads strictly: unexpected, mis-shaped and missing tensors must fail loudly."""
    from lemas_tts_amd import _lib
    from lemas_tts_amd.engine import DiTEngine
    arch = DiTArch(depth=1)
    sd = synth.synth_cfm_state_dict(arch, 30, 5)
    with pytest.raises(_lib.LemasError, match="unexpected tensor"):
        DiTEngine(arch, 30, {**sd, "transformer.bogus.weight": np.zeros((2, 2), np.float32)}, device="cuda:0")
    bad = dict(sd)
    bad["transformer.proj_out.weight"] = np.zeros((100, 512), np.float32)
    with pytest.raises(_lib.LemasError, match="wrong shape"):
        DiTEngine(arch, 30, bad, device="cuda:0")
    miss = {k: v for k, v in sd.items() if k != "transformer.norm_out.linear.bias"}
    with pytest.raises(_lib.LemasError, match="missing tensor"):
        DiTEngine(arch, 30, miss, device="cuda:0")
    with pytest.raises(_lib.LemasError, match="prosody"):
        DiTEngine(arch, 30, {**sd, "prosody_to_mel.weight": np.zeros((100, 512), np.float32)}, device="cuda:0")
    # accent-classifier tensors of the reference checkpoint are accepted and ignored (cfm.py:171)
    DiTEngine(arch, 30, sd, device="cuda:0").close()
