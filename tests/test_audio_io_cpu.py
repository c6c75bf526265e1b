"""WAV reader / writer of the host mirror (stand-ins for torchaudio.load and soundfile.write, utils_infer.py:425, api.py:162-166)
and the command-line surface of the north-star entry script (scripts/tts_multilingual.py:175-296).  CPU tier."""
import struct
import wave

import numpy as np
import pytest
import torch

from lemas_tts_amd.infer.audio_io import load_wav, save_wav


def test_pcm16_matches_the_stdlib_reader(tmp_path):
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=(1000, 2))
    p = tmp_path / "a.wav"
    save_wav(p, x, 22050, "PCM_16")
    with wave.open(str(p), "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (2, 2, 22050, 1000)
        pcm = np.frombuffer(f.readframes(1000), dtype="<i2").reshape(1000, 2)
    np.testing.assert_array_equal(pcm, np.clip(np.rint(x * 32768.0), -32768, 32767).astype(np.int16))
    y, sr = load_wav(p)
    assert sr == 22050 and y.shape == (2, 1000) and y.dtype == torch.float32
    np.testing.assert_array_equal(y.numpy(), pcm.T.astype(np.float32) / 32768.0)     # torchaudio.load's normalisation


@pytest.mark.parametrize("subtype,tol", [("PCM_16", 1 / 32768), ("PCM_24", 1 / 8388608), ("FLOAT", 0.0)])
def test_round_trip(tmp_path, subtype, tol):
    rng = np.random.default_rng(1)
    x = rng.uniform(-0.999, 0.999, size=777).astype(np.float32)     # odd length: 24-bit body needs the pad byte
    p = tmp_path / "r.wav"
    save_wav(p, x, 24000, subtype)
    y, sr = load_wav(p)
    assert sr == 24000 and y.shape == (1, 777)
    # PCM_16 rounds to nearest; PCM_24 keeps the top 3 bytes of the rounded 32-bit value (libsndfile): one step of truncation
    assert np.abs(y.numpy()[0] - x).max() <= tol * (1.0 if subtype == "PCM_24" else 0.5) + 1e-9


def test_pcm24_known_values_and_saturation(tmp_path):
    p = tmp_path / "k.wav"
    save_wav(p, np.array([0.0, 0.5, -0.5, 1.5, -1.5, 1 / 8388608]), 8000, "PCM_24")
    y, _ = load_wav(p)
    np.testing.assert_array_equal(y.numpy()[0], np.array([0.0, 0.5, -0.5, 8388607 / 8388608, -1.0, 1 / 8388608], np.float32))


def _raw_wav(path, fmt_body, data, extra_chunks=b""):
    body = b"WAVE" + extra_chunks + b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body + b"data" + struct.pack("<I", len(data)) + data
    path.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_extensible_header_other_widths_and_extra_chunks(tmp_path):
    # WAVE_FORMAT_EXTENSIBLE float32 stereo with a LIST chunk (odd size: padded) in front of fmt
    x = np.array([[0.25, -0.25], [0.5, -1.0], [0.0, 1.0]], dtype="<f4")
    guid = struct.pack("<H", 3) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    fmt = struct.pack("<HHIIHH", 0xFFFE, 2, 16000, 16000 * 8, 8, 32) + struct.pack("<HHI", 22, 32, 3) + guid
    p = tmp_path / "e.wav"
    _raw_wav(p, fmt, x.tobytes(), extra_chunks=b"LIST" + struct.pack("<I", 3) + b"abc\x00")
    y, sr = load_wav(p)
    assert sr == 16000
    np.testing.assert_array_equal(y.numpy(), x.T)
    # 8-bit offset binary, 32-bit PCM, 64-bit float
    _raw_wav(p, struct.pack("<HHIIHH", 1, 1, 8000, 8000, 1, 8), bytes([0, 128, 255]))
    np.testing.assert_array_equal(load_wav(p)[0].numpy()[0], np.array([-1.0, 0.0, 127 / 128], np.float32))
    _raw_wav(p, struct.pack("<HHIIHH", 1, 1, 8000, 32000, 4, 32), np.array([-2 ** 31, 2 ** 30], "<i4").tobytes())
    np.testing.assert_array_equal(load_wav(p)[0].numpy()[0], np.array([-1.0, 0.5], np.float32))
    _raw_wav(p, struct.pack("<HHIIHH", 3, 1, 8000, 64000, 8, 64), np.array([0.125, -0.75], "<f8").tobytes())
    np.testing.assert_array_equal(load_wav(p)[0].numpy()[0], np.array([0.125, -0.75], np.float32))


def test_rejects_what_it_cannot_read(tmp_path):
    p = tmp_path / "bad.wav"
    p.write_bytes(b"not a wave file at all")
    with pytest.raises(ValueError, match="RIFF"):
        load_wav(p)
    _raw_wav(p, struct.pack("<HHIIHH", 0x0055, 1, 8000, 8000, 1, 8), b"\x00\x00")        # MP3-in-WAV
    with pytest.raises(ValueError, match="format tag"):
        load_wav(p)
    _raw_wav(p, struct.pack("<HHIIHH", 1, 2, 8000, 8000, 3, 16), b"\x00" * 12)           # block align != channels * width
    with pytest.raises(ValueError, match="inconsistent"):
        load_wav(p)
    with pytest.raises(ValueError, match="subtype"):
        save_wav(p, np.zeros(4), 8000, "ULAW")


# ---------------------------------------------------------------- entry-script surface
def test_cli_defaults_are_the_reference_scripts():
    """scripts/tts_multilingual.py:175-296 (NOT TTS.infer's defaults: the script asks for NFE 64, cfg 5, sway 3)."""
    from lemas_tts_amd.scripts.tts_multilingual import build_parser
    a = build_parser().parse_args(["--ref_audio", "r.wav", "--ref_text", "a", "--text", "b"])
    assert (a.model, a.ckpt_file, a.vocab_file, a.frontend, a.output_wave) == ("multilingual_grl", "", "", "phone", "output.wav")
    assert (a.nfe_step, a.cfg_strength, a.sway_sampling_coef, a.ref_ratio, a.speed, a.seed) == (64, 5.0, 3.0, 1.0, 1.0, -1)
    assert not any([a.use_ema, a.enable_prosody_encoder, a.denoise, a.no_ref_audio, a.separate_langs, a.use_acc_grl])
    with pytest.raises(SystemExit):
        build_parser().parse_args(["--ref_audio", "r.wav", "--frontend", "bytes"])


def test_cli_resolution_and_refusals(tmp_path, monkeypatch):
    import lemas_tts_amd.scripts.tts_multilingual as M
    root = tmp_path / "pretrained_models"
    (root / "ckpts" / "multilingual_grl").mkdir(parents=True)
    (root / "data" / "multilingual_grl").mkdir(parents=True)
    monkeypatch.setattr(M, "PRETRAINED_ROOT", root)
    monkeypatch.setattr(M, "CKPTS_ROOT", root / "ckpts")
    with pytest.raises(FileNotFoundError, match="No ckpt found"):
        M._resolve_ckpt("multilingual_grl", None)
    with pytest.raises(FileNotFoundError, match="Vocab file not found"):
        M._resolve_vocab("multilingual_grl", None)
    for n in ("model_1000.pt", "model_2000.safetensors", "model_1500.safetensors"):
        (root / "ckpts" / "multilingual_grl" / n).write_bytes(b"")
    (root / "data" / "multilingual_grl" / "vocab.txt").write_text("a\n")
    # the reference sorts the safetensors list and the pt list together and takes the last (:93-104)
    assert M._resolve_ckpt("multilingual_grl", None).endswith("model_2000.safetensors")
    assert M._resolve_ckpt("multilingual_grl", "/x/y.pt") == "/x/y.pt"
    assert M._resolve_vocab("multilingual_grl", None) == str(root / "data" / "multilingual_grl" / "vocab.txt")
    (root / "ckpts" / "multilingual_prosody.safetensors").write_bytes(b"")           # fallback: directly under ckpts/
    assert M._resolve_ckpt("multilingual_prosody", None).endswith("multilingual_prosody.safetensors")
    # --denoise needs the reference's pretrained_models/uvr5 directory (or --denoise_model): a missing one is named, not worked around
    (root / "data" / "multilingual_prosody").mkdir(parents=True)
    (root / "data" / "multilingual_prosody" / "vocab.txt").write_text("a\n")
    (tmp_path / "r.wav").write_bytes(b"")
    with pytest.raises(FileNotFoundError, match="--denoise: .*uvr5 does not exist"):
        M.main(["--model", "multilingual_prosody", "--ref_audio", str(tmp_path / "r.wav"), "--ref_text", "a", "--text", "b", "--denoise"])
    (root / "uvr5").mkdir()
    with pytest.raises(FileNotFoundError, match="no Kim_Vocal_1"):
        M.main(["--model", "multilingual_prosody", "--ref_audio", str(tmp_path / "r.wav"), "--ref_text", "a", "--text", "b", "--denoise"])
    with pytest.raises(SystemExit, match="go together"):
        M.main(["--ref_audio", "r.wav", "--ref_phones", "a|b"])
    with pytest.raises(FileNotFoundError, match="Prosody encoder assets"):
        M.build_tts("multilingual_prosody", "c", "v", None, False, None, True)
    assert M._phone_lines("a|b||c\\nd|e\n\n") == [["a", "b", "c"], ["d", "e"]]


def test_pretrained_root_env_override(tmp_path, monkeypatch):
    import lemas_tts_amd.api as A
    monkeypatch.setenv("LEMAS_PRETRAINED_ROOT", str(tmp_path))
    assert A._find_pretrained_root() == tmp_path
    monkeypatch.setenv("LEMAS_PRETRAINED_ROOT", str(tmp_path / "missing"))
    assert A._find_pretrained_root().name == "pretrained_models"


def test_against_scipy_wavfile(tmp_path):
    """an independent RIFF implementation on both sides: scipy writes -> load_wav reads, save_wav writes -> scipy reads"""
    from scipy.io import wavfile
    rng = np.random.default_rng(5)
    for dtype, scale in ((np.int16, 32768.0), (np.int32, 2147483648.0), (np.uint8, None), (np.float32, 1.0), (np.float64, 1.0)):
        if dtype == np.uint8:
            x = rng.integers(0, 256, size=(501, 2), dtype=np.uint8)
            want = (x.astype(np.float32) - 128.0) / 128.0
        elif np.issubdtype(dtype, np.integer):
            info = np.iinfo(dtype)
            x = rng.integers(info.min, info.max, size=(501, 2), dtype=dtype)
            want = (x.astype(np.float64) / scale).astype(np.float32)
        else:
            x = rng.uniform(-1, 1, size=(501, 2)).astype(dtype)
            want = x.astype(np.float32)
        p = tmp_path / f"s_{np.dtype(dtype).name}.wav"
        wavfile.write(str(p), 44100, x)
        y, sr = load_wav(p)
        assert sr == 44100 and y.shape == (2, 501)
        np.testing.assert_array_equal(y.numpy(), want.T)
    z = rng.uniform(-1, 1, size=(333, 2))
    save_wav(tmp_path / "p16.wav", z, 24000, "PCM_16")
    sr, got = wavfile.read(str(tmp_path / "p16.wav"))
    assert sr == 24000 and got.dtype == np.int16
    np.testing.assert_array_equal(got, np.clip(np.rint(z * 32768.0), -32768, 32767).astype(np.int16))
    save_wav(tmp_path / "pf.wav", z, 24000, "FLOAT")
    sr, got = wavfile.read(str(tmp_path / "pf.wav"))
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, z.astype(np.float32))
    save_wav(tmp_path / "p24.wav", z, 24000, "PCM_24")
    sr, got = wavfile.read(str(tmp_path / "p24.wav"))             # scipy returns 24-bit data left-justified in int32
    want = (np.clip(np.rint(z.astype(np.float64) * 2147483648.0), -2147483648.0, 2147483647.0).astype(np.int64) >> 8).astype(np.int32)
    np.testing.assert_array_equal(got >> 8, want)                  # libsndfile: double -> int32 (x 0x80000000, clipped), top 3 bytes
