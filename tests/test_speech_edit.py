"""The speech-edit entry point (BASELINE config 5; scripts/speech_edit_multilingual.py): its edit-mask builder against the
masks the REFERENCE function produced (tests/golden/edit_masks.npz, made by oracle/gen_golden.py running the reference's
gen_wav_multilingual with a stand-in tts), CPU tier; the whole mirrored routine against the oracle pipeline, GPU tier."""
import os

import numpy as np
import pytest
import torch

from oracle import lemas_oracle as O


def _cases(golden_dir):
    fx = dict(np.load(os.path.join(golden_dir, "edit_masks.npz")))
    for name in sorted({k.split("/")[0] for k in fx}):
        yield name, int(fx[name + "/nw"]), [tuple(x) for x in fx[name + "/spans"]], fx[name + "/edit_mask"], int(fx[name + "/duration"])


def test_edit_mask_builders_match_the_reference(golden_dir):
    from lemas_tts_amd.scripts.speech_edit_multilingual import build_edit_mask
    n = 0
    for name, nw, spans, ref, dur in _cases(golden_dir):
        assert dur == nw // 256, name
        np.testing.assert_array_equal(build_edit_mask(spans, nw).numpy(), ref, err_msg=f"host mirror: {name}")
        np.testing.assert_array_equal(O.build_edit_mask(nw, spans).numpy(), ref, err_msg=f"oracle: {name}")
        n += 1
    assert n == 6


def test_tokens_from_text_dispatch():
    from lemas_tts_amd.scripts.speech_edit_multilingual import build_tokens_from_text

    class _T:
        frontend = None
    assert build_tokens_from_text(_T, " ab ") == [["a", "b", "."]]
    assert build_tokens_from_text(_T, "ok?") == [["o", "k", "?"]]

    class _FP:
        dtype = "phone"

        @staticmethod
        def text2phn(s):
            return "(cmn)|n|i||h|"
    _T.frontend = _FP
    assert build_tokens_from_text(_T, "x") == [["(zh)", "n", "i", "h"]]

    class _FC:
        dtype = "char"

        @staticmethod
        def text2norm(s):
            return "cmn", "ab"
    _T.frontend = _FC
    assert build_tokens_from_text(_T, "x") == [["(zh)", "a", "b"]]


@pytest.mark.gpu
def test_gen_wav_multilingual_vs_oracle_pipeline():
    """three edit spans in a 6 s utterance, raw audio in, waveform out; defaults of the entry point except NFE"""
    import types
    from lemas_tts_amd import synth
    from lemas_tts_amd.engine import VocosEngine
    from lemas_tts_amd.model.cfm import CFM
    from lemas_tts_amd.model.layout import DiTArch
    from lemas_tts_amd.scripts.speech_edit_multilingual import gen_wav_multilingual
    arch = DiTArch(depth=2)
    vocab = {c: i for i, c in enumerate(" abcdefghijklmnopqrstuvwxyz.")}
    sd = synth.synth_cfm_state_dict(arch, len(vocab), 91)
    vsd = synth.synth_vocos_state_dict(92)
    model = CFM(arch, len(vocab), sd, vocab_char_map=vocab, device="cuda:0")
    voc = types.SimpleNamespace(engine=VocosEngine(vsd, device="cuda:0"))
    tts = types.SimpleNamespace(ema_model=model, vocoder=voc, frontend=None, device="cuda:0", mel_spec_type="vocos")
    g = torch.Generator().manual_seed(93)
    nw = 144000
    audio = torch.randn(nw, generator=g) * 0.02                   # rms < 0.1: exercises the rescale in and out
    spans = [(0.8, 1.4), (2.5, 3.0), (4.6, 5.2)]
    F_ = nw // 256 + 1
    y0 = torch.from_numpy(synth.synth_noise(94, F_ + 1))[None]
    wav, mel = gen_wav_multilingual(tts, audio, 24000, "the quick brown fox", spans, nfe_step=3, cfg_strength=5.0,
                                    sway_sampling_coef=3.0, y0=y0)
    assert mel.shape == (1, 100, F_ + 1) and wav.shape == (256 * F_,)

    rms = float(audio.pow(2).mean().sqrt())
    a = audio[None] * (0.1 / rms)
    cond = O.vocos_mel_spectrogram(a).permute(0, 2, 1)
    text = O.tokens_to_idx([list("the quick brown fox.")], vocab)
    edit = O.build_edit_mask(nw, spans)
    ref, _ = O.OracleCFM(sd, arch).sample(cond, text, nw // 256, y0=y0, steps=3, cfg_strength=5.0, sway_sampling_coef=3.0, edit_mask=edit)
    keep = torch.nn.functional.pad(edit, (0, 1), value=False)[0]
    mse = float(((mel.cpu().permute(0, 2, 1)[0, ~keep] - ref[0, ~keep]).double() ** 2).mean())
    print(f"\n[gen_wav_multilingual] mel-MSE over regenerated frames {mse:.3e} ({int((~keep).sum())} frames)")
    assert mse <= 1e-4
    wref = O.OracleVocos(vsd).decode(mel.cpu()) * (rms / 0.1)
    assert float((wav.cpu() - wref[0]).abs().max()) < 1e-4 * max(1.0, float(wref.abs().max()))


def test_edit_mask_builders_agree_on_random_spans():
    """the host mirror's run-length builder and the oracle's restatement are two independent writings of
    scripts/speech_edit_multilingual.py:124-158; both reproduce the six reference-produced masks above, and they must keep
    agreeing on thousands of random span layouts (overlapping margins, spans clipped at either end, odd lengths, other
    sample rates / hops)"""
    from lemas_tts_amd.scripts.speech_edit_multilingual import build_edit_mask
    rng = np.random.default_rng(0)
    for it in range(3000):
        sr, hop = (24000, 256) if it % 5 else (int(rng.choice([16000, 22050, 44100])), int(rng.choice([160, 200, 512])))
        nw = int(rng.integers(hop * 3, sr * 12))
        total = nw / sr
        k = int(rng.integers(0, 5))
        pts = np.sort(rng.uniform(0, total, size=2 * k))
        spans = [(float(pts[2 * i]), float(pts[2 * i + 1])) for i in range(k)]
        a = build_edit_mask(spans, nw, sr, hop)
        b = O.build_edit_mask(nw, spans, sr, hop)
        assert a.shape == b.shape and torch.equal(a, b), (it, nw, sr, hop, spans)
        assert a.shape[1] >= nw // hop + 1


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(6)))
def test_gen_wav_multilingual_random_spans_and_rates(seed):
    """random utterance lengths, span layouts, prompt sample rates (resampled on the way in) and loudness (rms rescale in / out)"""
    import types
    from lemas_tts_amd import synth
    from lemas_tts_amd.engine import VocosEngine
    from lemas_tts_amd.model.cfm import CFM
    from lemas_tts_amd.model.layout import DiTArch
    from lemas_tts_amd.scripts.speech_edit_multilingual import gen_wav_multilingual
    rng = np.random.default_rng(40 + seed)
    arch = DiTArch(depth=2)
    vocab = {c: i for i, c in enumerate(" abcdefghijklmnopqrstuvwxyz.")}
    sd = synth.synth_cfm_state_dict(arch, len(vocab), 191)
    vsd = synth.synth_vocos_state_dict(192)
    model = CFM(arch, len(vocab), sd, vocab_char_map=vocab, device="cuda:0")
    tts = types.SimpleNamespace(ema_model=model, vocoder=types.SimpleNamespace(engine=VocosEngine(vsd, device="cuda:0")), frontend=None,
                                device="cuda:0", mel_spec_type="vocos")
    sr = int(rng.choice([24000, 16000, 44100]))
    secs = float(rng.uniform(1.5, 5.0))
    g = torch.Generator().manual_seed(seed)
    audio = torch.randn(int(secs * sr), generator=g) * float(rng.choice([0.02, 0.3]))
    k = int(rng.integers(1, 4))
    pts = np.sort(rng.uniform(0.1, secs - 0.1, size=2 * k))
    spans = [(float(pts[2 * i]), float(pts[2 * i + 1])) for i in range(k)]
    rms = float(audio.pow(2).mean().sqrt())
    a = audio[None] * (0.1 / rms) if rms < 0.1 else audio[None]
    a24 = O.resample_sinc_hann(a, sr, 24000) if sr != 24000 else a
    nw = a24.shape[-1]
    F_ = nw // 256 + 1
    y0 = torch.from_numpy(synth.synth_noise(200 + seed, F_ + 1))[None]
    wav, mel = gen_wav_multilingual(tts, audio, sr, "some new words", spans, nfe_step=2, cfg_strength=5.0, sway_sampling_coef=3.0, y0=y0)
    assert mel.shape == (1, 100, F_ + 1) and wav.shape == (256 * F_,)
    cond = O.vocos_mel_spectrogram(a24).permute(0, 2, 1)
    edit = O.build_edit_mask(nw, spans)
    text = O.tokens_to_idx([list("some new words.")], vocab)
    ref, _ = O.OracleCFM(sd, arch).sample(cond, text, nw // 256, y0=y0, steps=2, cfg_strength=5.0, sway_sampling_coef=3.0, edit_mask=edit)
    keep = torch.nn.functional.pad(edit, (0, 1), value=False)[0]
    if int((~keep).sum()):
        mse = float(((mel.cpu().permute(0, 2, 1)[0, ~keep] - ref[0, ~keep]).double() ** 2).mean())
        assert mse <= 1e-4, (seed, sr, secs, spans, mse)
    gain = rms / 0.1 if rms < 0.1 else 1.0
    wref = O.OracleVocos(vsd).decode(mel.cpu()) * gain
    assert float((wav.cpu() - wref[0]).abs().max()) < 1e-4 * max(1.0, float(wref.abs().max()))
