"""WAV file I/O for the host mirror (the reference calls ``torchaudio.load`` / ``soundfile.write``, neither of which
exists in this image: ``utils_infer.py:425``, ``api.py:162-166``, ``scripts/tts_multilingual.py:84``).

``load_wav`` returns what ``torchaudio.load(path)`` returns for a RIFF/WAVE file: a float32 tensor ``[channels, frames]``
in [-1, 1) (integer PCM divided by ``2**(bits-1)``; 8-bit is offset binary) and the sample rate.  Supported encodings:
PCM 8/16/24/32-bit, IEEE float 32/64-bit, plain or WAVE_FORMAT_EXTENSIBLE headers.  Anything else raises.
"""
from __future__ import annotations

import struct

import numpy as np
import torch

_PCM, _FLOAT, _EXTENSIBLE = 0x0001, 0x0003, 0xFFFE


def _chunks(buf: bytes):
    pos = 12
    while pos + 8 <= len(buf):
        cid, size = buf[pos:pos + 4], struct.unpack_from("<I", buf, pos + 4)[0]
        yield cid, buf[pos + 8:pos + 8 + size]
        pos += 8 + size + (size & 1)          # chunks are word aligned


def load_wav(path) -> tuple[torch.Tensor, int]:
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 12 or buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    fmt = data = None
    for cid, body in _chunks(buf):
        if cid == b"fmt " and fmt is None:
            fmt = body
        elif cid == b"data" and data is None:
            data = body
    if fmt is None or data is None or len(fmt) < 16:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, channels, rate, _, block, bits = struct.unpack_from("<HHIIHH", fmt, 0)
    if tag == _EXTENSIBLE:
        if len(fmt) < 40:
            raise ValueError(f"{path}: truncated WAVE_FORMAT_EXTENSIBLE header")
        tag = struct.unpack_from("<H", fmt, 24)[0]           # first two bytes of the sub-format GUID
    if channels <= 0 or block != channels * (bits // 8):
        raise ValueError(f"{path}: inconsistent header (channels {channels}, block {block}, bits {bits})")
    n = len(data) // block
    raw = data[:n * block]
    if tag == _PCM:
        if bits == 8:
            x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v & 0x800000, v - 0x1000000, v)
            x = v.astype(np.float32) / 8388608.0
        elif bits == 32:
            x = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == _FLOAT:
        if bits == 32:
            x = np.frombuffer(raw, dtype="<f4").astype(np.float32)
        elif bits == 64:
            x = np.frombuffer(raw, dtype="<f8").astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported float width {bits}")
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag 0x{tag:04x}")
    return torch.from_numpy(np.ascontiguousarray(x.reshape(n, channels).T)), int(rate)


def save_wav(path, wav, sample_rate: int, subtype: str = "PCM_16") -> None:
    """``soundfile.write(path, wav, sr)`` for mono/multi-channel float data: ``wav`` is ``[frames]`` or ``[frames, channels]``.
    The float -> integer step restates libsndfile's, as python-soundfile drives it (it switches SFC_SET_CLIPPING on, so the
    ``*_clip_array`` converters of libsndfile's pcm.c run): ``PCM_16`` (soundfile's default for .wav) = ``lrint(x * 0x8000)``
    saturated to int16; ``PCM_24`` (the denoised prompt, ``tts_multilingual.py:84``) = ``lrint(x * 0x80000000)`` saturated to
    int32, of which the top three bytes are written (an arithmetic ``>> 8``, not a second rounding); ``FLOAT`` stores float32.
    Neither package is installed here, so this is restated from the published sources, not pinned."""
    a = np.asarray(wav, dtype=np.float64)
    if a.ndim == 1:
        a = a[:, None]
    frames, channels = a.shape
    if subtype == "PCM_16":
        q = np.clip(np.rint(a * 32768.0), -32768, 32767).astype("<i2")
        body, tag, bits = q.tobytes(), _PCM, 16
    elif subtype == "PCM_24":
        q = np.clip(np.rint(a * 2147483648.0), -2147483648.0, 2147483647.0).astype(np.int64) >> 8
        u = (q & 0xFFFFFF).astype(np.uint32).reshape(-1)
        body = np.stack([u & 0xFF, (u >> 8) & 0xFF, (u >> 16) & 0xFF], axis=1).astype(np.uint8).tobytes()
        tag, bits = _PCM, 24
    elif subtype == "FLOAT":
        body, tag, bits = a.astype("<f4").tobytes(), _FLOAT, 32
    else:
        raise ValueError(f"unsupported subtype {subtype!r}")
    block = channels * bits // 8
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, tag, channels, int(sample_rate), int(sample_rate) * block, block, bits) + b"data" + struct.pack("<I", len(body))
    with open(path, "wb") as f:
        f.write(hdr + body + (b"\x00" if len(body) & 1 else b""))
