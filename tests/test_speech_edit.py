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


# ---------------------------------------------------------------- file-level entry point (run_edit_for_pair / main, :210-466)
def _alignment():
    return {"interval": [0.5, 4.5], "display_text": "the quick brown fox jumps", "modified_text": ["brown fox", "red cat"],
            "modified_index": [2, 4],
            "words": [{"interval": [0.6, 0.9]}, {"interval": [1.0, 1.5]}, {"interval": [1.7, 2.2]}, {"interval": [2.3, 2.9]},
                      {"interval": [3.1, 3.9]}]}


def test_edit_request_from_alignment_and_pairs(tmp_path):
    from lemas_tts_amd.scripts.speech_edit_multilingual import build_parser, collect_pairs, edit_request_from_alignment
    (utt0, utt1), spans, text = edit_request_from_alignment(_alignment())
    assert (utt0, utt1) == (0.5, 4.5) and text == "the quick red cat jumps"
    assert spans == [(pytest.approx(1.7 - 0.5 - 0.1), pytest.approx(2.9 - 0.5))]           # :246-247
    d = _alignment()
    d["modified_index"] = [-3, 99]                                                          # clipped to the word list (:238-239)
    d["words"][0]["interval"][0] = 0.52                                                     # start - 0.1 s would be negative
    _, spans, _ = edit_request_from_alignment(d)
    assert spans == [(0.0, pytest.approx(3.9 - 0.5))]
    d["modified_index"] = [3, 3]
    with pytest.raises(AssertionError, match="empty"):
        edit_request_from_alignment(d)
    for n in ("b.wav", "a.WAV", "c.txt", "d.mp3"):
        (tmp_path / n).write_bytes(b"")
    pairs = collect_pairs(None, str(tmp_path), "/al", "/out")
    assert pairs == [(str(tmp_path / "a.WAV"), "/al/a.json", "/out/a.wav"), (str(tmp_path / "b.wav"), "/al/b.json", "/out/b.wav")]
    assert collect_pairs("/x/y.wav", "ignored", "/al", "/out") == [("/x/y.wav", "/al/y.json", "/out/y.wav")]
    a = build_parser().parse_args([])
    assert (a.model, a.frontend, a.nfe_step, a.cfg_strength, a.sway_sampling_coef, a.ref_ratio, a.seed) == ("multilingual", "phone", 64, 5.0, 3.0, 1.0, -1)


@pytest.mark.gpu
def test_speech_edit_main_files_in_files_out(tmp_path):
    """wav (stereo, 16 kHz) + alignment json in, edited 24 kHz wav out; equals the in-memory routine on the same segment"""
    import json
    import types
    from lemas_tts_amd import synth
    from lemas_tts_amd.engine import VocosEngine
    from lemas_tts_amd.infer.audio_io import load_wav, save_wav
    from lemas_tts_amd.model.cfm import CFM
    from lemas_tts_amd.model.layout import DiTArch
    import lemas_tts_amd.scripts.speech_edit_multilingual as M
    arch = DiTArch(depth=2)
    vocab = {c: i for i, c in enumerate(" abcdefghijklmnopqrstuvwxyz.")}
    model = CFM(arch, len(vocab), synth.synth_cfm_state_dict(arch, len(vocab), 101), vocab_char_map=vocab, device="cuda:0")
    voc = types.SimpleNamespace(engine=VocosEngine(synth.synth_vocos_state_dict(102), device="cuda:0"))
    tts = types.SimpleNamespace(ema_model=model, vocoder=voc, frontend=None, device="cuda:0", mel_spec_type="vocos", target_sample_rate=24000)
    rng = np.random.default_rng(103)
    x = 0.05 * rng.standard_normal((5 * 16000, 2))
    (tmp_path / "in").mkdir(); (tmp_path / "al").mkdir()
    save_wav(tmp_path / "in" / "utt.wav", x, 16000, "PCM_16")
    (tmp_path / "al" / "utt.json").write_text(json.dumps(_alignment()))
    save_wav(tmp_path / "in" / "orphan.wav", x[:100], 16000, "PCM_16")            # no alignment: warned about and skipped
    rc = M.main(["--wav_dir", str(tmp_path / "in"), "--align_dir", str(tmp_path / "al"), "--save_dir", str(tmp_path / "out"),
                 "--nfe_step", "2", "--seed", "77", "--frontend", "none"], tts=tts)
    assert rc == 0 and not (tmp_path / "out" / "orphan.wav").exists()
    got, sr = load_wav(tmp_path / "out" / "utt.wav")
    wav, sr_in = M.load_wav_mono(str(tmp_path / "in" / "utt.wav"), 24000)
    assert sr == 24000 and sr_in == 24000 and wav.ndim == 1 and abs(wav.numel() - 5 * 24000) <= 1 and float(wav.abs().max()) <= 0.999
    seg = wav[int(round(0.5 * 24000)):int(round(4.5 * 24000))]
    (_, _), spans, text = M.edit_request_from_alignment(_alignment())
    ref, mel = M.gen_wav_multilingual(tts, seg, 24000, text, spans, nfe_step=2, cfg_strength=5.0, sway_sampling_coef=3.0, seed=77)
    assert got.shape == (1, ref.numel()) and mel.shape[2] == seg.numel() // 256 + 2   # F = nw // hop + 1 prompt frames, +1 (cfm.py:300-302)
    np.testing.assert_array_equal(got.numpy()[0], ref.reshape(-1).cpu().numpy())   # 32-bit float WAV: bit-exact
