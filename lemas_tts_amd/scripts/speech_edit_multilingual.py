"""Host-side mirror of the speech-edit entry point ``lemas_tts/scripts/speech_edit_multilingual.py`` (BASELINE config 5,
SURVEY.md 3.2 / 8a row a-E): same function names, arguments, defaults and return values; the sampler and the vocoder
behind it are the MI355X engines.  File I/O (``load_wav_mono``, ``run_edit_for_pair``: soundfile / torchaudio.load) and
the text frontend stay with the caller (frontend is host Python by north_star)."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def build_tokens_from_text(tts, text: str) -> List[List[str]]:
    """scripts/speech_edit_multilingual.py:29-63: sentence-final punctuation is ensured, then the TTS object's frontend
    decides: ``dtype == "phone"`` -> ``text2phn`` split on '|' (with (cmn) -> (zh)); ``"char"`` -> language tag + characters;
    no frontend -> plain characters."""
    t = text.strip()
    if not t.endswith((".", "。", "!", "？", "?", "！")):
        t += "."
    fe = getattr(tts, "frontend", None)
    if fe is None:
        return [list(t)]
    kind = getattr(fe, "dtype", "phone")
    if kind == "phone":
        phones = fe.text2phn(t + " ").replace("(cmn)", "(zh)")
        return [[p for p in phones.split("|") if p]]
    if kind == "char":
        lang, norm = fe.text2norm(t + " ")
        return [["(" + lang.replace("cmn", "zh") + ")"] + list(norm)]
    return [list(t)]


def build_edit_mask(parts_to_edit: Sequence[Tuple[float, float]], n_samples: int, target_sr: int = 24000,
                    hop_length: int = 256) -> torch.Tensor:
    """The mask ``gen_wav_multilingual`` builds (:124-158): bool [1, n_samples // hop + 1], True = keep the original frame.
    Every span is widened by 0.1 s on both sides (clipped to the utterance); frame counts are rounded per run, measured
    from the END of the previous widened span, so rounding does not accumulate across spans."""
    runs = []                      # (value, length)
    cursor = 0.0                   # samples consumed so far (end of the previous widened span)
    total_sec = n_samples / target_sr
    for start, end in parts_to_edit:
        lo, hi = max(start - 0.1, 0.0), min(end + 0.1, total_sec)
        lo_samples = int(round(lo * target_sr))
        keep = int(round((lo_samples - cursor) / hop_length))
        edit = int(round(int(round((hi - lo) * target_sr)) / hop_length))
        if keep > 0:
            runs.append((True, keep))
        if edit > 0:
            runs.append((False, edit))
        cursor = hi * target_sr
    mask = np.concatenate([np.full(n, v, dtype=bool) for v, n in runs]) if runs else np.zeros(0, dtype=bool)
    frames = n_samples // hop_length + 1
    if mask.shape[0] < frames:
        mask = np.concatenate([mask, np.ones(frames - mask.shape[0], dtype=bool)])
    return torch.from_numpy(mask)[None]


def gen_wav_multilingual(tts, segment_audio: torch.Tensor, sr: int, target_text, parts_to_edit: List[Tuple[float, float]],
                         nfe_step: int = 64, cfg_strength: float = 5.0, sway_sampling_coef: float = 3.0, ref_ratio: float = 1.0,
                         no_ref_audio: bool = False, use_acc_grl: bool = False, use_prosody_encoder_flag: bool = False,
                         seed: Optional[int] = None, *, y0: Optional[torch.Tensor] = None,
                         prosody_embeds: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """:67-207.  ``target_text`` is the full text after editing (a string for the frontend, or an already tokenised
    ``list[str]``); returns ``(wav [T], mel [1, 100, frames])``.  ``y0`` / ``prosody_embeds`` are the explicit-input
    extensions of the mirrored sampler."""
    model, vocoder = tts.ema_model, tts.vocoder
    mel_spec = getattr(model, "mel_spec", None)
    if mel_spec is None:
        raise RuntimeError("CFM model has no attached MelSpec; check your checkpoint.")
    target_sr, hop = int(mel_spec.target_sample_rate), int(mel_spec.hop_length)
    target_rms = 0.1
    audio = segment_audio[None] if segment_audio.dim() == 1 else segment_audio
    audio = audio.to(torch.float32)
    rms = torch.sqrt(torch.mean(torch.square(audio)))                         # :112-115
    if rms < target_rms:
        audio = audio * target_rms / rms
    if sr != target_sr:                                                       # :118-120
        from ..engine import ResampleEngine
        audio = ResampleEngine(int(sr), target_sr, device=model.device)(audio).cpu()
    edit_mask = build_edit_mask(parts_to_edit, audio.shape[-1], target_sr, hop)
    duration = audio.shape[-1] // hop                                         # :161 (the sampler raises it to frames + 1)
    tokens = [list(target_text)] if isinstance(target_text, (list, tuple)) else build_tokens_from_text(tts, target_text)
    if hasattr(tts, "process_phone_list") and len(tokens) > 0 and not isinstance(target_text, (list, tuple)):
        tokens = [tts.process_phone_list(tokens[0])]                          # :171-172
    generated, _ = model.sample(cond=audio, text=tokens, duration=duration, steps=nfe_step, cfg_strength=cfg_strength,
                                sway_sampling_coef=sway_sampling_coef, seed=seed, edit_mask=edit_mask, use_acc_grl=use_acc_grl,
                                use_prosody_encoder=use_prosody_encoder_flag, ref_ratio=ref_ratio, no_ref_audio=no_ref_audio,
                                y0=y0, prosody_embeds=prosody_embeds)
    mel = generated.to(torch.float32).permute(0, 2, 1)                        # [B, C, T_mel]  :191-192
    if tts.mel_spec_type != "vocos":
        raise ValueError(f"Unsupported vocoder type: {tts.mel_spec_type}")
    gain = float(rms / target_rms) if rms < target_rms else 1.0               # :203-204
    wav = vocoder.engine.decode(mel, gain=gain) if hasattr(vocoder, "engine") else vocoder.decode(mel) * gain
    return wav.squeeze(0), mel
