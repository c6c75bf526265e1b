"""Host-side mirror of ``lemas_tts/infer/utils_infer.py`` for the acoustic path.

Same names, argument meaning and return values as the reference for ``load_vocoder`` (:120-159),
``load_checkpoint`` (:204-246), ``load_model`` (:252-303), ``infer_process`` (:399-458) and
``infer_batch_process`` (:464-625); the arithmetic (sampling, vocoding) runs in liblemas_hip.so.
Differences that are stated rather than hidden:

* compute dtype: the reference picks fp16 on "cuda" (:205-213); this build uses bf16 MFMA operands with fp32
  accumulation / residual stream / ODE state (BASELINE.json), whatever ``dtype`` says;
* reference audio arrives as a ``(tensor[channels, samples], sample_rate)`` pair (what ``torchaudio.load`` returns
  at :422; any sample rate: it is resampled to 24 kHz on the device, :494-496) or as a ready mel ``[F, 100]``;
* ``ref_text`` / ``gen_text`` are phone-token lists (the text frontend stays host Python and is out of scope);
  the ``str`` branch (:509-515) needs ``convert_char_to_pinyin`` and raises ``NotImplementedError``.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
import yaml

from ..engine import VocosEngine
from ..model.cfm import CFM, pad_rows
from ..model.layout import DROPPED_ON_LOAD, DiTArch, VocosArch

# module-level defaults of the reference (:68-81)
target_sample_rate = 24000
n_mel_channels = 100
hop_length = 256
win_length = 1024
n_fft = 1024
mel_spec_type = "vocos"
target_rms = 0.1
cross_fade_duration = 0.15
ode_method = "euler"
nfe_step = 32
cfg_strength = 3.0
sway_sampling_coef = 1
speed = 1.0
fix_duration = None

device = "cuda" if torch.cuda.is_available() else "cpu"


def get_tokenizer(vocab_file: str, tokenizer: str = "custom"):
    """``model/utils.py:121-126`` ("custom" only: the path the inference driver uses, utils_infer.py:267-273)."""
    if tokenizer != "custom":
        raise NotImplementedError("only the 'custom' vocab-file tokenizer is used at inference")
    vocab_char_map = {}
    with open(vocab_file, "r", encoding="utf-8") as f:
        for i, char in enumerate(f):
            vocab_char_map[char[:-1]] = i
    return vocab_char_map, len(vocab_char_map)


class _Vocoder:
    """What callers hold as ``vocoder``: ``decode(mel[B,100,L]) -> wav[B,256(L-1)]`` (call site :549)."""

    def __init__(self, engine: VocosEngine):
        self.engine = engine

    def decode(self, mel: torch.Tensor) -> torch.Tensor:
        return self.engine.decode(mel)

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self


def load_vocoder(vocoder_name="vocos", is_local=False, local_path="", device=device, hf_cache_dir=None, state_dict=None):
    """Reads ``{local_path}/config.yaml`` + ``pytorch_model.bin`` exactly as :123-143.  ``state_dict`` lets a caller
    hand over an in-memory Vocos state dict (synthetic weights in tests/bench)."""
    if vocoder_name != "vocos":
        raise NotImplementedError("the bigvgan branch (:144-158) is out of scope; mel_spec_type is 'vocos' in both shipped configs")
    arch = VocosArch()
    if state_dict is None:
        if not is_local:
            raise RuntimeError("no network in this build: pass a local vocos directory (config.yaml + pytorch_model.bin)")
        with open(f"{local_path}/config.yaml") as f:
            cfg = yaml.safe_load(f)
        bb = cfg["backbone"]["init_args"]
        hd = cfg["head"]["init_args"]
        arch = VocosArch(input_channels=bb["input_channels"], dim=bb["dim"], intermediate_dim=bb["intermediate_dim"],
                         num_layers=bb["num_layers"], n_fft=hd["n_fft"], hop_length=hd["hop_length"])
        state_dict = torch.load(f"{local_path}/pytorch_model.bin", map_location="cpu", weights_only=True)
    return _Vocoder(VocosEngine(state_dict, device=_cuda(device), arch=arch))


def _cuda(dev) -> str:
    d = str(dev)
    return "cuda:0" if d == "cuda" else d


def read_checkpoint(ckpt_path: str, use_ema: bool = True) -> dict:
    """The key surgery of ``load_checkpoint`` (:215-241): safetensors or .pt, strip ``ema_model.``, drop
    ``initted``/``step`` and the mel_spec / ctc keys."""
    ckpt_type = ckpt_path.split(".")[-1]
    if ckpt_type == "safetensors":
        from safetensors.torch import load_file
        checkpoint = load_file(ckpt_path, device="cpu")
    else:
        checkpoint = torch.load(ckpt_path, map_location="cpu", weights_only=True)
    if use_ema:
        if ckpt_type == "safetensors":
            checkpoint = {"ema_model_state_dict": checkpoint}
        sd = {k.replace("ema_model.", ""): v for k, v in checkpoint["ema_model_state_dict"].items()
              if k not in ["initted", "step"]}
        for key in DROPPED_ON_LOAD:
            sd.pop(key, None)
    else:
        if ckpt_type == "safetensors":
            checkpoint = {"model_state_dict": checkpoint}
        sd = checkpoint["model_state_dict"]
    return sd


def load_checkpoint(model_args: dict, ckpt_path: str, device: str, dtype=None, use_ema=True) -> CFM:
    """Strict load (:237): an unexpected or missing tensor raises, as ``load_state_dict`` does."""
    sd = read_checkpoint(ckpt_path, use_ema)
    return CFM(state_dict=sd, device=_cuda(device), **model_args)


def load_model(model_cls, model_cfg, ckpt_path, mel_spec_type=mel_spec_type, vocab_file="", ode_method=ode_method,
               use_ema=True, device=device, use_prosody_encoder=False, prosody_cfg_path="", prosody_ckpt_path="",
               state_dict: Optional[dict] = None, vocab_char_map: Optional[dict] = None):
    """``model_cls`` is accepted for signature parity (the reference passes ``DiT``); ``model_cfg`` is ``model.arch``
    of the yaml.  ``state_dict`` + ``vocab_char_map`` bypass the files (synthetic weights)."""
    arch = DiTArch.from_yaml_arch(dict(model_cfg))
    if vocab_char_map is None:
        if vocab_file == "":
            raise FileNotFoundError("vocab_file is required (the reference's bundled default does not ship)")
        vocab_char_map, vocab_size = get_tokenizer(vocab_file, "custom")
    else:
        vocab_size = len(vocab_char_map)
    args = dict(arch=arch, vocab_size=vocab_size, vocab_char_map=vocab_char_map, use_prosody_encoder=use_prosody_encoder,
                num_channels=n_mel_channels, odeint_kwargs=dict(method=ode_method))
    if use_prosody_encoder:                                            # cfm.py:139-145: the Pretssel ECAPA encoder next to the CFM
        from ..api import CKPTS_ROOT
        if not prosody_cfg_path:                                       # :276-280: default assets under CKPTS_ROOT/prosody_encoder
            prosody_cfg_path = str(CKPTS_ROOT / "prosody_encoder" / "pretssel_cfg.json")
        if not prosody_ckpt_path:
            prosody_ckpt_path = str(CKPTS_ROOT / "prosody_encoder" / "prosody_encoder_UnitY2.pt")
        have = os.path.exists(prosody_cfg_path) and os.path.exists(prosody_ckpt_path)
        if have:
            from ..model.prosody_encoder import ProsodyEncoder
            args["prosody_encoder"] = ProsodyEncoder(prosody_cfg_path, prosody_ckpt_path, device=_cuda(device))
        elif state_dict is None:
            # a real checkpoint load: the reference would fail opening these files (prosody_encoder.py:390-424)
            raise FileNotFoundError(f"use_prosody_encoder needs {prosody_cfg_path} and {prosody_ckpt_path}")
        # synthetic-weights bypass without encoder assets: CFM.sample then requires explicit prosody_embeds for raw audio
    if state_dict is not None:
        return CFM(state_dict=state_dict, device=_cuda(device), **args)
    return load_checkpoint(args, ckpt_path, device, use_ema=use_ema)


def load_arch_config(model: str) -> dict:
    """``lemas_tts/configs/<model>.yaml`` -> the ``model`` section (api.py:99-105); the two shipped configs are bundled."""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(here, "configs", f"{model}.yaml")) as f:
        return yaml.safe_load(f)["model"]


def infer_process(ref_audio, ref_text, gen_text, model_obj, vocoder, mel_spec_type=mel_spec_type, show_info=print,
                  progress=None, target_rms=target_rms, cross_fade_duration=cross_fade_duration, nfe_step=nfe_step,
                  cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, use_acc_grl=True,
                  use_prosody_encoder=True, ref_ratio=None, no_ref_audio=False, speed=speed, fix_duration=fix_duration,
                  device=device, **extra):
    """:399-458.  ``ref_audio`` is a wav path (``torchaudio.load`` at :425 -> ``audio_io.load_wav``), the loaded
    ``(audio, sr)`` pair, or a mel (see module docstring)."""
    if isinstance(ref_audio, (str, os.PathLike)):
        from .audio_io import load_wav
        ref_audio = load_wav(ref_audio)
    if isinstance(ref_text, str):
        raise NotImplementedError("str ref_text needs chunk_text/convert_char_to_pinyin (frontend, out of scope); pass phone lists")
    gen_text_batches = gen_text
    show_info(f"Generating audio in {len(gen_text_batches)} batches...")
    return next(infer_batch_process(
        ref_audio, ref_text, gen_text_batches, model_obj, vocoder, mel_spec_type=mel_spec_type, progress=progress,
        target_rms=target_rms, cross_fade_duration=cross_fade_duration, nfe_step=nfe_step, cfg_strength=cfg_strength,
        sway_sampling_coef=sway_sampling_coef, use_acc_grl=use_acc_grl, use_prosody_encoder=use_prosody_encoder,
        ref_ratio=ref_ratio, no_ref_audio=no_ref_audio, speed=speed, fix_duration=fix_duration, device=device, **extra))


def cross_fade_concat(waves, cross_fade_duration: float, sr: int = target_sample_rate) -> np.ndarray:
    """:581-617 linear cross-fade between consecutive generated lines."""
    if cross_fade_duration <= 0:
        return np.concatenate(waves)
    final = waves[0]
    for nxt in waves[1:]:
        n = min(int(cross_fade_duration * sr), len(final), len(nxt))
        if n <= 0:
            final = np.concatenate([final, nxt])
            continue
        mix = final[-n:] * np.linspace(1, 0, n) + nxt[:n] * np.linspace(0, 1, n)
        final = np.concatenate([final[:-n], mix, nxt[n:]])
    return final


def infer_batch_process(ref_audio, ref_text, gen_text_batches, model_obj, vocoder, mel_spec_type="vocos", progress=None,
                        target_rms=0.1, cross_fade_duration=0.15, nfe_step=32, cfg_strength=2.0, sway_sampling_coef=-1,
                        use_acc_grl=True, use_prosody_encoder=True, ref_ratio=None, no_ref_audio=False, speed=1,
                        fix_duration=None, device=None, streaming=False, chunk_size=2048, seed=None,
                        prosody_embeds=None, noise=None, batch_lines=1, skip_padding_blocks=False):
    """:464-625 generator.  Yields ``(final_wave, 24000, combined_mel)`` (or chunks when ``streaming``).
    ``noise`` (list of y0 tensors, one per line) and ``prosody_embeds`` are explicit inputs the reference draws /
    computes on its own device.  ``batch_lines`` > 1 runs up to that many lines of ``gen_text`` as ONE
    ``CFM.sample`` batch (one captured graph per step for all of them) instead of the reference's serial loop
    (:572-579); lines of unequal length then follow the reference's own B > 1 semantics (``lens`` / duration masks,
    cfm.py:336-339).  The attention half of every block always skips the 128-row blocks that lie wholly in a shorter line's padding (exact:
    the reference zeroes that half's output there).  ``skip_padding_blocks`` (with ``batch_lines`` > 1; engine option ``skip_dead``,
    DESIGN.md section 8) lets the FF half skip them too: ``True`` / 2 keeps one padding block alive behind every line (the rows the
    reference's position-embedding conv reads: results at the reference's own error level, ~+2 % on a ragged batch of 8), 1 skips
    them all (~+4 %; a line's last ~30 frames then differ from the B > 1 reference's by 1e-5 instead of 2e-6 mel-MSE)."""
    rms = None
    if isinstance(ref_audio, tuple):
        audio, sr = ref_audio
        if audio.shape[0] > 1:
            audio = torch.mean(audio, dim=0, keepdim=True)                      # :488-489
        rms = torch.sqrt(torch.mean(torch.square(audio)))                       # :491
        if rms < target_rms:
            audio = audio * target_rms / rms                                    # :492-493
        if sr != target_sample_rate:                                            # :494-496
            from ..engine import resampler
            audio = resampler(int(sr), target_sample_rate, device=getattr(model_obj, "device", device) or "cuda:0")(audio).cpu()
        cond = audio
        ref_audio_len = audio.shape[-1] // hop_length                           # :520
    else:
        cond = torch.as_tensor(ref_audio, dtype=torch.float32)
        if cond.ndim == 2:
            cond = cond[None]
        ref_audio_len = cond.shape[1] - 1                                       # F = nw // hop + 1

    generated_waves, spectrograms = [], []

    def line_duration(gen_text):
        if fix_duration is not None:
            return int(fix_duration * target_sample_rate / hop_length)          # :522
        return ref_audio_len + int(ref_audio_len / len(ref_text) * len(gen_text) / speed)   # :525-527

    gain = float(rms / target_rms) if (rms is not None and rms < target_rms) else 1.0      # :552-553

    def vocode(gen_mel):
        wave = vocoder.engine.decode(gen_mel, gain=gain) if hasattr(vocoder, "engine") else vocoder.decode(gen_mel) * gain
        return wave.squeeze().cpu().numpy()                                      # :557

    lines = list(gen_text_batches)
    group = max(1, int(batch_lines)) if not streaming else 1
    # skip_padding_blocks: False / 0 = the reference's arithmetic, True / 2 = engine option skip_dead 2 (all but one padding block behind every
    # line skipped: the reference's error level), the integer 1 = skip_dead 1 (all of them).  The engine's setting is restored when this call
    # ends, however it ends: the option belongs to this batch, not to the model.
    engine = getattr(model_obj, "engine", None) if group > 1 else None
    if engine is not None:
        level = 0 if not skip_padding_blocks else 2 if skip_padding_blocks is True else int(skip_padding_blocks)
        skip_dead_before = engine.option("skip_dead", 0)
        engine.set_option("skip_dead", level)
    try:
        yield from _infer_lines(lines, group, model_obj, cond, ref_text, line_duration, nfe_step, cfg_strength, sway_sampling_coef, use_acc_grl,
                                use_prosody_encoder, ref_ratio, no_ref_audio, seed, noise, prosody_embeds, ref_audio_len, vocode, streaming, chunk_size,
                                generated_waves, spectrograms)
    finally:
        if engine is not None:
            engine.set_option("skip_dead", skip_dead_before)
    if streaming:
        return
    if generated_waves:
        final_wave = cross_fade_concat(generated_waves, cross_fade_duration)
        yield np.clip(final_wave, -0.999, 0.999), target_sample_rate, np.concatenate(spectrograms, axis=1)   # :620-622
    else:
        yield None, target_sample_rate, None


def _infer_lines(lines, group, model_obj, cond, ref_text, line_duration, nfe_step, cfg_strength, sway_sampling_coef, use_acc_grl,
                 use_prosody_encoder, ref_ratio, no_ref_audio, seed, noise, prosody_embeds, ref_audio_len, vocode, streaming, chunk_size,
                 generated_waves, spectrograms):
    """the per-line loop of ``infer_batch_process`` (utils_infer.py:506-565 per line, :572-618 around it); yields only when streaming"""
    for g0 in range(0, len(lines), group):
        chunk = lines[g0: g0 + group]
        if len(chunk) == 1:                                                      # the reference's path: one line, B = 1
            gen_text = chunk[0]
            generated, _ = model_obj.sample(
                cond=cond, text=[list(ref_text) + list(gen_text)], duration=line_duration(gen_text), steps=nfe_step,
                cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, use_acc_grl=use_acc_grl,
                use_prosody_encoder=use_prosody_encoder, ref_ratio=ref_ratio, no_ref_audio=no_ref_audio, seed=seed,
                y0=None if noise is None else noise[g0], prosody_embeds=prosody_embeds)
            mels = [generated.to(torch.float32)[:, ref_audio_len:, :].permute(0, 2, 1)]    # :545-547
        else:                                                                    # several lines as one batch
            nb = len(chunk)
            durs = torch.tensor([line_duration(t) for t in chunk], dtype=torch.long)
            cond_b = cond if cond.ndim == 2 and cond.shape[0] == nb else cond.expand(nb, *cond.shape[1:])
            y0 = None
            if noise is not None:
                y0 = pad_rows([noise[g0 + j][0] for j in range(nb)], 0)
            pros = prosody_embeds
            if pros is not None and pros.shape[0] == 1:
                pros = pros.expand(nb, -1)
            generated, _ = model_obj.sample(
                cond=cond_b, text=[list(ref_text) + list(t) for t in chunk], duration=durs, steps=nfe_step,
                cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, use_acc_grl=use_acc_grl,
                use_prosody_encoder=use_prosody_encoder, ref_ratio=ref_ratio, no_ref_audio=no_ref_audio, seed=seed,
                y0=y0, prosody_embeds=pros)
            generated = generated.to(torch.float32)
            # frames of line j: the sampler raises a duration below max(tokens, ref frames) + 1 (cfm.py:300-302)
            f_ref = cond.shape[1] if cond.ndim == 3 else ref_audio_len + 1
            eff = [max(max(len(ref_text) + len(t), f_ref) + 1, int(d)) for t, d in zip(chunk, durs)]
            mels = [generated[j: j + 1, ref_audio_len: eff[j], :].permute(0, 2, 1).contiguous() for j in range(nb)]
        for gen_mel in mels:
            wave = vocode(gen_mel)
            if streaming:
                for j in range(0, len(wave), chunk_size):
                    yield wave[j: j + chunk_size], target_sample_rate
            else:
                generated_waves.append(wave)
                spectrograms.append(gen_mel[0].cpu().numpy())
