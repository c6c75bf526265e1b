"""GPU tier: the prosody encoder through the C ABI -- the HIP ECAPA-TDNN against the golden vectors the
reference's own class produced, the kaldi-fbank kernel chain against the oracle restatement, and the row as CFM.sample
uses it (raw prompt -> 16 kHz -> fbank -> embedding -> conditioning).  Everything is exact fp32: tolerance 2e-5."""
import os

import numpy as np
import pytest
import torch

from lemas_tts_amd import synth
from lemas_tts_amd.model.layout import DiTArch, ProsodyArch

pytestmark = pytest.mark.gpu


def _encoder(wseed, arch=ProsodyArch()):
    from lemas_tts_amd.model.prosody_encoder import ProsodyEncoder
    return ProsodyEncoder(state_dict=synth.synth_prosody_encoder_state_dict(wseed, arch), arch=arch, device="cuda:0")


@pytest.mark.parametrize("name", ["prosody_enc_short", "prosody_enc_10s"])
def test_ecapa_matches_reference_golden(golden_dir, name):
    fx = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    enc = _encoder(int(fx["wseed"]))
    emb = enc(torch.from_numpy(fx["fbank"])).cpu().numpy()
    err = float(np.abs(emb - fx["emb"]).max())
    print(f"\n[{name}] max|emb - reference| {err:.2e}")
    assert err < 2e-5
    np.testing.assert_allclose(np.linalg.norm(emb, axis=-1), 1.0, atol=1e-5)


def test_ecapa_other_architecture_vs_oracle():
    """a second, smaller architecture (projection shortcut, no global context, other kernels / dilations) against the oracle"""
    from oracle.prosody_oracle import OracleECAPA
    arch = ProsodyArch(channels=(96, 128, 128, 256), kernel_sizes=(3, 5, 3, 1), dilations=(2, 1, 3, 1), attention_channels=32,
                       res2net_scale=4, se_channels=16, global_context=False, groups=(1, 1, 1, 1), embed_dim=64, input_dim=40)
    sd = synth.synth_prosody_encoder_state_dict(7, arch)
    assert "blocks.1.shortcut.weight" in sd
    from lemas_tts_amd.model.prosody_encoder import ProsodyEncoder
    enc = ProsodyEncoder(state_dict=sd, arch=arch, device="cuda:0")
    fb = torch.from_numpy(synth.synth_fbank(8, 77, bins=40))[None]
    ref = OracleECAPA(sd, arch).forward(fb)
    assert float((enc(fb).cpu() - ref).abs().max()) < 2e-5


def test_strict_load_and_prefix_stripping():
    from lemas_tts_amd.model.prosody_encoder import ProsodyEncoder
    sd = synth.synth_prosody_encoder_state_dict(9)
    pre = {"prosody_encoder." + k: v for k, v in sd.items()}
    pre["unrelated.weight"] = np.zeros(3, np.float32)                  # ignored once a prefixed key exists (prosody_encoder.py:410-418)
    ProsodyEncoder(state_dict=pre, device="cuda:0")
    bad = dict(sd); bad.pop("fc.bias")
    with pytest.raises(RuntimeError, match="missing keys"):
        ProsodyEncoder(state_dict=bad, device="cuda:0")
    bad = dict(sd); bad["extra"] = np.zeros(1, np.float32)
    with pytest.raises(RuntimeError, match="unexpected keys"):
        ProsodyEncoder(state_dict=bad, device="cuda:0")


@pytest.mark.parametrize("T", [1, 2, 15, 16, 17, 33, 63, 65, 130])
def test_ecapa_published_widths_at_tile_edges_vs_oracle(T):
    """the published architecture (the one-launch Res2Net chunks: 16-row tiles with a dilation halo of up to 4 rows; column statistics on
    64 row lanes; the wave-per-output one-row Linears) at frame counts around every tile edge, against the oracle restatement"""
    from oracle.prosody_oracle import OracleECAPA
    arch = ProsodyArch()
    sd = synth.synth_prosody_encoder_state_dict(17, arch)
    enc = _encoder(17)
    fb = torch.from_numpy(synth.synth_fbank(T + 100, T))[None]
    ref = OracleECAPA(sd, arch).forward(fb)
    out = enc(fb).cpu()
    assert torch.isfinite(out).all()
    assert float((out - ref).abs().max()) < 2e-5, float((out - ref).abs().max())


def test_ecapa_is_deterministic_and_independent_of_what_ran_before():
    """the same prompt twice, with a longer one in between (workspaces regrown, LDS tiles reused): bit-identical embeddings"""
    enc = _encoder(17)
    a = torch.from_numpy(synth.synth_fbank(3, 200))[None]
    b = torch.from_numpy(synth.synth_fbank(4, 1500))[None]
    e1 = enc(a).cpu().numpy()
    enc(b)
    e2 = enc(a).cpu().numpy()
    np.testing.assert_array_equal(e1, e2)


@pytest.mark.parametrize("n", [400, 16000, 160123, 150])
def test_kaldi_fbank_vs_oracle(n):
    from oracle.prosody_oracle import kaldi_fbank_80
    g = torch.Generator().manual_seed(n)
    wav = torch.randn(n, generator=g) * 0.1 + 0.01
    enc = _encoder(3)
    out = enc.extract_fbank_16k(wav).cpu()
    ref = kaldi_fbank_80(wav)
    assert out.shape == ref.shape
    # log-mel energies: fp32 DFT-as-GEMM vs torch.fft; compare in the log domain
    assert float((out - ref).abs().max()) < 2e-3, float((out - ref).abs().max())


def test_sampler_computes_the_embedding_from_the_raw_prompt():
    """cfm.py:248-262 + :313-318: a raw 24 kHz prompt through resample -> fbank -> ECAPA gives the conditioning embedding;
    sampling with it equals sampling with the same embedding passed in explicitly, and it matches the oracle chain."""
    from lemas_tts_amd.model.cfm import CFM
    from oracle import prosody_oracle as P
    arch = DiTArch(depth=2)
    sd = synth.synth_cfm_state_dict(arch, 898, 41, prosody=True)
    enc = _encoder(42)
    m = CFM(arch, 898, sd, device="cuda:0", use_prosody_encoder=True, prosody_encoder=enc)
    g = torch.Generator().manual_seed(5)
    wav = torch.randn(1, 24000, generator=g) * 0.05
    text = torch.from_numpy(synth.synth_tokens(6, 20, 898))[None]
    F_ = 24000 // 256 + 1
    y0 = torch.from_numpy(synth.synth_noise(7, 200))[None]
    emb = enc.embed_prompt(wav, 24000)
    ref_emb = P.OracleECAPA(synth.synth_prosody_encoder_state_dict(42), ProsodyArch()).forward(P.prosody_features_from_24k(wav[0])[None])
    assert float((emb.cpu() - ref_emb).abs().max()) < 1e-3          # fbank differences (1e-3 log units) pushed through the encoder
    a, _ = m.sample(wav, text, 200, steps=2, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False)
    b, _ = m.sample(wav, text, 200, steps=2, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False, prosody_embeds=emb)
    np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    c, _ = m.sample(wav, text, 200, steps=2, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False, use_prosody_encoder=False)
    assert not np.array_equal(a.cpu().numpy()[:, F_:], c.cpu().numpy()[:, F_:])
