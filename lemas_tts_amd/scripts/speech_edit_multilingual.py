"""Host-side mirror of the speech-edit entry point ``lemas_tts/scripts/speech_edit_multilingual.py`` (BASELINE config 5,
SURVEY.md 3.2 / 8a row a-E): same function names, arguments, defaults and return values; the sampler and the vocoder
behind it are the MI355X engines, the wav files go through ``infer/audio_io.py`` (no torchaudio / soundfile here) and the
prompt resampling through the HIP polyphase resampler.  The text frontend stays host Python (north_star): ``--frontend
phone|char`` asks the factory registered with ``lemas_tts_amd.api.set_frontend_factory``, ``--frontend none`` feeds characters.

    python -m lemas_tts_amd.scripts.speech_edit_multilingual --wav_dir in/ --align_dir align/ --save_dir out/ --ckpt_file ... --vocab_file ...
"""
from __future__ import annotations

import argparse
import json
import os
import time
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def load_wav_mono(path: str, target_sr: int, device="cuda:0") -> Tuple[torch.Tensor, int]:
    """:17-26: load, average the channels, resample to ``target_sr`` (``torchaudio.functional.resample`` = the sinc/Hann
    polyphase filter of ``lemas_resample_*``), clamp to +-0.999; returns ``(wav [T], sr)``."""
    from ..infer.audio_io import load_wav
    wav, sr = load_wav(path)
    if wav.dim() > 1 and wav.shape[0] > 1:
        wav = wav.mean(dim=0, keepdim=True)
    if sr != target_sr:
        from ..engine import resampler
        wav = resampler(int(sr), int(target_sr), device=device)(wav).cpu()
        sr = target_sr
    return torch.clip(wav, -0.999, 0.999).squeeze(0), sr


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
        from ..engine import resampler
        audio = resampler(int(sr), target_sr, device=model.device)(audio).cpu()
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


def edit_request_from_alignment(data: dict) -> Tuple[Tuple[float, float], List[Tuple[float, float]], str]:
    """The bookkeeping of ``run_edit_for_pair`` (:230-262) on one alignment record: the utterance interval [s], the span to
    regenerate relative to it (the edited words, widened by 0.1 s at the front and clipped at utterance end + 0.1 s) and the
    text after the edit."""
    utt_start, utt_end = data["interval"]
    start_idx, end_idx = data["modified_index"]
    words = data["words"]
    start_idx, end_idx = max(0, start_idx), min(len(words), end_idx)
    assert start_idx < end_idx, "modified_index range is empty."
    w0, w1 = words[start_idx]["interval"][0], words[end_idx - 1]["interval"][1]
    span = (max(0.0, w0 - utt_start - 0.1), min(w1 - utt_start, utt_end - utt_start + 0.1))
    orig_phrase, new_phrase = data["modified_text"]
    return (utt_start, utt_end), [span], data["display_text"].replace(orig_phrase, new_phrase)


def run_edit_for_pair(tts, wav_path: str, json_path: str, save_path: str, *, nfe_step: int, cfg_strength: float,
                      sway_sampling_coef: float, ref_ratio: float, no_ref_audio: bool, use_acc_grl: bool,
                      use_prosody_encoder_flag: bool, seed: Optional[int]) -> None:
    """:210-283: one (wav, alignment json) pair -> edited wav on disk (32-bit float WAV, what ``torchaudio.save`` writes for
    a float tensor)."""
    from ..infer.audio_io import save_wav
    os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
    wav, sr = load_wav_mono(wav_path, tts.target_sample_rate, device=tts.device)
    with open(json_path, "r", encoding="utf-8") as f:
        data = json.load(f)
    (utt_start, utt_end), parts_to_edit, target_text = edit_request_from_alignment(data)
    segment = wav[int(round(utt_start * sr)):int(round(utt_end * sr))]
    print(f"\n[EDIT] {os.path.basename(wav_path)}")
    print(f"  target_text : {target_text}")
    print(f"  edit_span    : {parts_to_edit} (sec, relative to utterance)")
    t0 = time.time()
    gen_wav, _ = gen_wav_multilingual(tts=tts, segment_audio=segment, sr=sr, target_text=target_text, parts_to_edit=parts_to_edit,
                                      nfe_step=nfe_step, cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef,
                                      ref_ratio=ref_ratio, no_ref_audio=no_ref_audio, use_acc_grl=use_acc_grl,
                                      use_prosody_encoder_flag=use_prosody_encoder_flag, seed=seed)
    save_wav(save_path, gen_wav.reshape(-1).cpu().numpy(), sr, "FLOAT")
    print(f"  saved: {save_path}  ({time.time() - t0:.3f} s)")


def collect_pairs(wav: Optional[str], wav_dir: str, align_dir: str, save_dir: str) -> List[Tuple[str, str, str]]:
    """:286-314: (wav, json, save) triples for one file or for every .wav of ``wav_dir`` (sorted); the reference also lists
    .mp3, which nothing here can decode."""
    if wav is not None:
        wav_paths = [wav]
    else:
        wav_paths = sorted(os.path.join(wav_dir, f) for f in os.listdir(wav_dir) if f.lower().endswith(".wav"))
    pairs = []
    for wp in wav_paths:
        base = os.path.splitext(os.path.basename(wp))[0]
        pairs.append((wp, os.path.join(align_dir, base + ".json"), os.path.join(save_dir, base + ".wav")))
    return pairs


def build_parser() -> argparse.ArgumentParser:
    """:317-384, same names, types and defaults (``bpe`` is not a frontend the TTS facade knows: api.py:199-213)."""
    p = argparse.ArgumentParser(description="Multilingual speech editing on MI355X.")
    p.add_argument("--wav", type=str)
    p.add_argument("--wav_dir", type=str)
    p.add_argument("--align_dir", type=str)
    p.add_argument("--save_dir", type=str)
    p.add_argument("--model", type=str, default="multilingual")
    p.add_argument("--ckpt_file", type=str, default="")
    p.add_argument("--vocab_file", type=str, default="")
    p.add_argument("--device", type=str, default=None)
    p.add_argument("--frontend", type=str, default="phone", choices=["phone", "char", "bpe", "none"])
    p.add_argument("--use_ema", action="store_true")
    p.add_argument("--enable_prosody_encoder", default=False, action="store_true")
    p.add_argument("--prosody_cfg_path", type=str, default="")
    p.add_argument("--prosody_ckpt_path", type=str, default="")
    p.add_argument("--nfe_step", type=int, default=64)
    p.add_argument("--speed", type=float, default=1.0)
    p.add_argument("--cfg_strength", type=float, default=5.0)
    p.add_argument("--sway_sampling_coef", type=float, default=3.0)
    p.add_argument("--ref_ratio", type=float, default=1.0)
    p.add_argument("--no_ref_audio", action="store_true")
    p.add_argument("--use_acc_grl", action="store_true")
    p.add_argument("--use_prosody_encoder", default=False, action="store_true")
    p.add_argument("--seed", type=int, default=-1)
    p.add_argument("--vocoder_local_path", type=str, default="", help="addition: vocos directory (config.yaml + pytorch_model.bin)")
    p.add_argument("--frontend_factory", type=str, default=None,
                   help="addition: 'package.module:callable' building the text frontend for --frontend phone|char (called as "
                        "callable(dtype=...)); default: the LEMAS_FRONTEND_FACTORY environment variable")
    return p


def main(argv=None, tts=None) -> int:
    """:386-466.  ``tts`` (addition) lets a caller hand over an already built ``TTS``."""
    args = build_parser().parse_args(argv)
    if tts is None:
        from ..api import CKPTS_ROOT, TTS
        if args.frontend_factory:
            from ..api import resolve_frontend_factory, set_frontend_factory
            set_frontend_factory(resolve_frontend_factory(args.frontend_factory))
        tts = TTS(model=args.model, ckpt_file=args.ckpt_file, vocab_file=args.vocab_file, device=args.device, use_ema=args.use_ema,
                  frontend=None if args.frontend == "none" else args.frontend, use_prosody_encoder=args.enable_prosody_encoder,
                  prosody_cfg_path=args.prosody_cfg_path, prosody_ckpt_path=args.prosody_ckpt_path,
                  vocoder_local_path=args.vocoder_local_path or CKPTS_ROOT / "vocos-mel-24khz")
    seed = None if args.seed == -1 else args.seed
    pairs = collect_pairs(args.wav, args.wav_dir, args.align_dir, args.save_dir)
    os.makedirs(args.save_dir, exist_ok=True)
    for wav_path, json_path, save_path in pairs:
        if not os.path.exists(wav_path):
            print(f"[WARN] wav not found: {wav_path}")
            continue
        if not os.path.exists(json_path):
            print(f"[WARN] json not found: {json_path}")
            continue
        run_edit_for_pair(tts=tts, wav_path=wav_path, json_path=json_path, save_path=save_path, nfe_step=args.nfe_step,
                          cfg_strength=args.cfg_strength, sway_sampling_coef=args.sway_sampling_coef, ref_ratio=args.ref_ratio,
                          no_ref_audio=args.no_ref_audio, use_acc_grl=args.use_acc_grl,
                          use_prosody_encoder_flag=args.use_prosody_encoder and args.enable_prosody_encoder, seed=seed)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
