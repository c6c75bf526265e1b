"""Host-side mirror of ``lemas_tts.model.backbones.prosody_encoder`` (inference only): ``ProsodyEncoder`` and
``extract_fbank_16k`` with the reference's names, argument meaning and error behaviour; the arithmetic runs on the
MI355X engine (``lemas_prosody_*``), there is no CPU path."""
from __future__ import annotations

import json
from pathlib import Path
from typing import Optional

import torch

from ..engine import ProsodyEngine, ResampleEngine
from .layout import ProsodyArch, prosody_param_shapes

AUDIO_SAMPLE_RATE = 16_000


def _strip_state(state: dict) -> dict:
    """``_load_prosody_encoder_state`` (prosody_encoder.py:406-424): keep and strip the ``prosody_encoder.`` /
    ``prosody_encoder_model.`` prefixed keys when the checkpoint has them."""
    if any(k.startswith(("prosody_encoder.", "prosody_encoder_model.")) for k in state):
        state = {k.replace("prosody_encoder_model.", "", 1).replace("prosody_encoder.", "", 1): v for k, v in state.items()
                 if k.startswith(("prosody_encoder.", "prosody_encoder_model."))}
    return state


class ProsodyEncoder:
    """``ProsodyEncoder(cfg_path, ckpt_path, freeze=True)`` (prosody_encoder.py:364-432).  ``state_dict`` / ``arch`` bypass
    the two files (synthetic weights; the reference's ``pretssel_cfg.json`` is not in its tree)."""

    def __init__(self, cfg_path: Optional[Path] = None, ckpt_path: Optional[Path] = None, freeze: bool = True, *,
                 state_dict: Optional[dict] = None, arch: Optional[ProsodyArch] = None, device="cuda:0"):
        if arch is None:
            if cfg_path:
                cfg = json.loads(Path(cfg_path).read_text())
                if "model" not in cfg:
                    raise ValueError(f"{cfg_path} does not contain a top-level 'model' key.")      # :385-386
                arch = ProsodyArch.from_pretssel_cfg(cfg["model"])
            else:
                arch = ProsodyArch()
        if state_dict is None:
            state_dict = torch.load(ckpt_path, map_location="cpu")
        state_dict = _strip_state(dict(state_dict))
        want = prosody_param_shapes(arch)
        missing, unexpected = [k for k in want if k not in state_dict], [k for k in state_dict if k not in want]
        if missing or unexpected:                                                                   # :420-424
            raise RuntimeError(f"Error loading checkpoint {ckpt_path}: missing keys={missing}, unexpected keys={unexpected}")
        self.arch = arch
        self.engine = ProsodyEngine(arch, state_dict, device=device)
        self._to16k = {}

    @property
    def device(self):
        return self.engine.device

    def extract_fbank_16k(self, audio_16k: torch.Tensor) -> torch.Tensor:
        """prosody_encoder.py:334-361: [T] or [1, T] at 16 kHz -> [frames, 80]; prompts shorter than one 25 ms window are tiled"""
        a = audio_16k.reshape(1, -1)
        if a.shape[-1] < 400:
            a = a.repeat(1, 400 // a.shape[-1] + 1)
        return self.engine.fbank(a[0])

    def resample_to_16k(self, audio: torch.Tensor, src_sr: int) -> torch.Tensor:
        """cfm.py:252-258"""
        if src_sr == AUDIO_SAMPLE_RATE:
            return audio.reshape(-1)
        if src_sr not in self._to16k:
            self._to16k[src_sr] = ResampleEngine(src_sr, AUDIO_SAMPLE_RATE, device=self.device)
        return self._to16k[src_sr](audio.reshape(1, -1))[0]

    def forward(self, fbank: torch.Tensor, padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """fbank [B, T, 80] -> [B, 512].  The sampler only ever calls this with one sample and ``padding_mask=None``
        (cfm.py:259); masked batches are not built."""
        if padding_mask is not None:
            raise NotImplementedError("padding_mask is not used on the inference path (cfm.py:259 passes None)")
        return torch.stack([self.engine.encode(fbank[b]) for b in range(fbank.shape[0])])

    __call__ = forward

    def embed_prompt(self, raw_audio: torch.Tensor, src_sr: int = 24000) -> torch.Tensor:
        """cfm.py:248-262 for a batch of raw prompts [B, nw] at ``src_sr``: per sample resample -> fbank -> encoder"""
        return torch.stack([self.forward(self.extract_fbank_16k(self.resample_to_16k(raw_audio[b], src_sr))[None])[0]
                            for b in range(raw_audio.shape[0])])
