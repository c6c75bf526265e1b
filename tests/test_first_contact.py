"""tools/first_contact.py (the checklist to run against REAL checkpoints, which no build of this repository has seen) on synthetic files laid
out like the real ones: it must load them, say PASS where they match the layouts this build declares, FAIL by name where they do not, and
call the RoPE probe INCONCLUSIVE on untrained weights.  CPU tier: the device section reports SKIP; GPU tier: it builds the engines from the files."""
import json
import os
import sys

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.dirname(__file__))

import first_contact as FC   # noqa: E402
from lemas_tts_amd import synth   # noqa: E402
from lemas_tts_amd.model.layout import DiTArch, ProsodyArch   # noqa: E402
from oracle import mdx_oracle as MO   # noqa: E402


@pytest.fixture()
def assets(tmp_path):
    from safetensors.torch import save_file
    depth, vocab = 2, 50
    sd = synth.synth_cfm_state_dict(DiTArch(depth=depth), vocab, 3)
    full = {"ema_model." + k: torch.from_numpy(v.copy()).contiguous() for k, v in sd.items()}
    full["ema_model.mel_spec.mel_stft.spectrogram.window"] = torch.hann_window(1024)          # dropped on load by the reference's loader
    full["ema_model.mel_spec.mel_stft.mel_scale.fb"] = torch.zeros(513, 100)
    save_file(full, str(tmp_path / "model.safetensors"))
    (tmp_path / "vocab.txt").write_text("".join(f"p{i}\n" for i in range(vocab)), encoding="utf-8")
    vdir = tmp_path / "vocos"
    vdir.mkdir()
    vsd = {k: torch.from_numpy(v.copy()) for k, v in synth.synth_vocos_state_dict(4).items()}
    vsd["head.istft.window"] = torch.hann_window(1024)
    vsd["feature_extractor.mel_spec.spectrogram.window"] = torch.hann_window(1024)
    torch.save(vsd, str(vdir / "pytorch_model.bin"))
    (vdir / "config.yaml").write_text(yaml.safe_dump({
        "backbone": {"class_path": "vocos.models.VocosBackbone", "init_args": {"input_channels": 100, "dim": 512, "intermediate_dim": 1536, "num_layers": 8}},
        "head": {"class_path": "vocos.heads.ISTFTHead", "init_args": {"dim": 512, "n_fft": 1024, "hop_length": 256, "padding": "center"}}}))
    pa = ProsodyArch()
    (tmp_path / "pretssel_cfg.json").write_text(json.dumps({"model": {
        "prosody_channels": list(pa.channels), "prosody_kernel_sizes": list(pa.kernel_sizes), "prosody_dilations": list(pa.dilations),
        "prosody_attention_channels": pa.attention_channels, "prosody_res2net_scale": pa.res2net_scale, "prosody_se_channels": pa.se_channels,
        "prosody_global_context": pa.global_context, "prosody_groups": list(pa.groups), "prosody_embed_dim": pa.embed_dim, "input_feat_per_channel": pa.input_dim}}))
    torch.save({"prosody_encoder." + k: torch.from_numpy(v.copy()) for k, v in synth.synth_prosody_encoder_state_dict(5).items()}, str(tmp_path / "prosody.pt"))
    from onnx_writer import convtdfnet_onnx
    uv = tmp_path / "uvr5"
    uv.mkdir()
    arch = MO.MdxArch(dim_f=64, dim_t=16, num_blocks=5, l=2, g=8, k=3, bn=4, bias=False)
    convtdfnet_onnx(str(uv / "Kim_Vocal_1.onnx"), arch, MO.seeded_state_dict(arch, 3))
    (uv / "MDX-Net-Kim-Vocal1.json").write_text(json.dumps({"mdx_dim_f_set": 64, "mdx_dim_t_set": 4, "mdx_n_fft_scale_set": 2048}))
    return tmp_path, depth


def _run(tmp, depth, capsys, **over):
    args = {"--ckpt": str(tmp / "model.safetensors"), "--vocab": str(tmp / "vocab.txt"), "--vocos": str(tmp / "vocos"),
            "--pretssel-cfg": str(tmp / "pretssel_cfg.json"), "--prosody-ckpt": str(tmp / "prosody.pt"), "--uvr5": str(tmp / "uvr5"), "--depth": str(depth)}
    args.update(over)
    rc = FC.main([x for kv in args.items() for x in kv if kv[1] is not None])
    return rc, capsys.readouterr().out


@pytest.mark.skipif(torch.cuda.is_available(), reason="the device section would build engines; this is the CPU-tier run of the checklist")
def test_checklist_on_synthetic_files(assets, capsys):
    tmp, depth = assets
    rc, out = _run(tmp, depth, capsys)
    assert rc == 0, out
    status = dict(FC.RESULTS)
    assert status["1 vocab"] == "PASS" and status["1 checkpoint strict load"] == "PASS" and status["1 rotary inv_freq buffer"] == "PASS"
    assert status["2 rope convention"] == "INCONCLUSIVE"                     # untrained weights: no locality under either convention
    assert status["3 vocos config.yaml"] == "PASS" and status["3 vocos strict load"] == "PASS" and status["3 vocos ISTFT window"] == "PASS"
    assert status["4 pretssel_cfg.json"] == "PASS" and status["4 prosody encoder strict load"] == "PASS"
    assert status["5 uvr5 onnx graph"] == "PASS" and status["5 uvr5 network hyper-parameters"] == "INCONCLUSIVE"    # a mini network, not the Kim shape
    assert status["6 device"] == "SKIP"
    assert "Conv" in out and "BatchNormalization" in out


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-tier run of the checklist")
def test_checklist_names_what_is_wrong(assets, capsys):
    from safetensors.torch import load_file, save_file
    tmp, depth = assets
    sd = load_file(str(tmp / "model.safetensors"))
    del sd["ema_model.transformer.transformer_blocks.1.attn.to_k.bias"]
    sd["ema_model.transformer.transformer_blocks.0.ff.ff.2.weight"] = torch.zeros(1024, 1024)
    sd["ema_model.transformer.surprise"] = torch.zeros(3)
    save_file(sd, str(tmp / "broken.safetensors"))
    (tmp / "uvr5" / "MDX-Net-Kim-Vocal1.json").write_text(json.dumps({"mdx_dim_f_set": 128, "mdx_dim_t_set": 4, "mdx_n_fft_scale_set": 2048}))
    rc, out = _run(tmp, depth, capsys, **{"--ckpt": str(tmp / "broken.safetensors")})
    assert rc == 1
    line = [x for x in out.splitlines() if "1 checkpoint strict load" in x][0]
    assert "FAIL" in line and "attn.to_k.bias" in line and "transformer.surprise" in line and "ff.ff.2.weight" in line
    assert dict(FC.RESULTS)["5 uvr5 network vs configuration"] == "FAIL"
    assert dict(FC.RESULTS)["2 rope convention"] == "SKIP"


def test_probe_conventions_are_the_two_in_question():
    """The probe's 'interleaved' rotation is the one the oracle's x_transformers stand-in (and the HIP epilogue) implements; 'half_split' is not."""
    from oracle import ref_shims as R
    t = torch.randn(2, 7, 64, generator=torch.Generator().manual_seed(1))
    freqs, _ = R.RotaryEmbedding(64).forward_from_seq_len(7)
    want = R.apply_rotary_pos_emb(t[None], freqs)[0]
    assert torch.allclose(FC._rope(t, "interleaved"), want, atol=1e-6)
    assert not torch.allclose(FC._rope(t, "half_split"), want, atol=1e-3)
    # both are rotations (norm-preserving) and both reduce to the identity at position 0
    for conv in ("interleaved", "half_split"):
        r = FC._rope(t, conv)
        assert torch.allclose(r.norm(dim=-1), t.norm(dim=-1), atol=1e-5) and torch.allclose(r[:, 0], t[:, 0], atol=1e-6)


@pytest.mark.gpu
def test_checklist_device_section(assets, capsys):
    """With a GPU the kit also builds the engines from the files and runs a 2-step synthesis and one denoiser forward."""
    tmp, depth = assets
    rc, out = _run(tmp, depth, capsys)
    assert rc == 0, out
    status = dict(FC.RESULTS)
    assert status["6 device: 2-step synthesis on the real weights"] == "PASS" and status["6 device: MDX-Net forward on the real weights"] == "PASS", out
