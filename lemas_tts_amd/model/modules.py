"""Host-side mirror of the one ``lemas_tts/model/modules.py`` symbol callers touch directly: ``MelSpec``.

``CFM.sample`` calls ``self.mel_spec(cond)`` when handed raw audio (cfm.py:232-236) and the speech-edit script reads
``mel_spec.target_sample_rate`` / ``.hop_length`` (scripts/speech_edit_multilingual.py:100-105).  Only the "vocos" mel
type of the shipped configs is built; the arithmetic runs in liblemas_hip.so (``lemas_mel_forward``).
"""
from __future__ import annotations

import torch

from ..engine import MelEngine


class MelSpec:
    def __init__(self, n_fft=1024, hop_length=256, win_length=1024, n_mel_channels=100, target_sample_rate=24_000,
                 mel_spec_type="vocos", device="cuda:0"):
        if mel_spec_type != "vocos":
            raise NotImplementedError("only the 'vocos' mel of the shipped configs is built (bigvgan branch out of scope)")
        if win_length != n_fft:
            raise NotImplementedError("win_length != n_fft is not used by the shipped configs")
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self.n_mel_channels, self.target_sample_rate = n_mel_channels, target_sample_rate
        self.engine = MelEngine(device, n_fft, hop_length, n_mel_channels, target_sample_rate)

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        """wav [B, nw] (or [B, 1, nw]) -> mel [B, n_mels, frames] like the reference (modules.py:94-101)."""
        if wav.ndim == 3:
            wav = wav.squeeze(1)
        assert wav.ndim == 2
        return self.engine.frames_first(wav).permute(0, 2, 1)

    forward = __call__
