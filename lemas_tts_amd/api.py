"""Host-side mirror of ``lemas_tts/api.py``: the ``TTS`` facade over the MI355X engines.

Same constructor arguments and the same ``infer(...)`` keyword surface and return value as ``api.py:83-97`` /
``:171-249``.  What differs is stated, not hidden:

* ``frontend`` is an OBJECT with ``text2phn(str) -> 'p1|p2|...'`` and ``dtype == "phone"`` (the reference builds
  ``TextNorm`` from espeak/jieba/langid, none of which exist here and all of which are out of scope); with
  ``frontend=None`` the caller passes phone-token lists directly;
* ``frontend="phone"`` / ``"char"`` (the reference's spelling) asks the factory registered with
  :func:`set_frontend_factory` for the object (the integrator registers ``lambda d: TextNorm(dtype=d)``; the product itself
  never imports the reference package) and raises ``TypeError`` when none is registered;
* ``ref_file`` is a wav path (as in the reference), a loaded ``(audio, sr)`` pair, or a ready mel ``[F, 100]``;
* ``state_dict`` / ``vocoder_state_dict`` / ``vocab_char_map`` allow in-memory (synthetic) weights.
"""
from __future__ import annotations

import random
import sys
from pathlib import Path

import numpy as np
import torch

from .infer.audio_io import save_wav
from .infer.utils_infer import infer_process, load_arch_config, load_model, load_vocoder


def _find_pretrained_root() -> Path:
    """``api.py:39-75``: LEMAS_PRETRAINED_ROOT, then a /models/<id>/pretrained_models mount, then ``pretrained_models`` next
    to an ancestor of this file or in the working directory (falling back to the latter path even when absent)."""
    import os
    env_root = os.environ.get("LEMAS_PRETRAINED_ROOT")
    if env_root and Path(env_root).is_dir():
        return Path(env_root)
    models_dir = Path("/models")
    if models_dir.is_dir():
        specific = models_dir / "LEMAS-Project__LEMAS-TTS" / "pretrained_models"
        if specific.is_dir():
            return specific
        for child in sorted(models_dir.iterdir()):
            if child.is_dir() and (child / "pretrained_models").is_dir():
                return child / "pretrained_models"
    for parent in Path(__file__).resolve().parents:
        if (parent / "pretrained_models").is_dir():
            return parent / "pretrained_models"
    return Path.cwd() / "pretrained_models"


PRETRAINED_ROOT = _find_pretrained_root()
CKPTS_ROOT = PRETRAINED_ROOT / "ckpts"


def seed_everything(seed=0):
    """``model/utils.py:18-25``."""
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


FRONTEND_FACTORY = None


def set_frontend_factory(factory) -> None:
    """``factory(dtype: str) -> frontend object`` for ``TTS(frontend="phone" | "char")`` (what api.py:140-151 does with the
    reference's ``TextNorm``).  The text frontend stays host Python outside this package (north star); this is the seam."""
    global FRONTEND_FACTORY
    FRONTEND_FACTORY = factory


def resolve_callable(spec: str, what: str = "factory"):
    """``"package.module:callable"`` -> the callable object it names (how the command-line entry points take host-side plug-ins -- the
    text frontend, the prompt denoiser's network -- without this package importing them)."""
    import importlib
    mod, sep, attr = spec.partition(":")
    if not sep or not mod or not attr:
        raise ValueError(f"{what} {spec!r}: expected 'package.module:callable'")
    obj = importlib.import_module(mod)
    for part in attr.split("."):
        obj = getattr(obj, part)
    if not callable(obj):
        raise TypeError(f"{what} {spec!r} is not callable")
    return obj


def resolve_frontend_factory(spec: str):
    """``"package.module:callable"`` -> the frontend factory (e.g. ``lemas_tts.infer.frontend:TextNorm`` called as ``TextNorm(dtype=...)``).
    This is how the command-line entry points and the ``LEMAS_FRONTEND_FACTORY`` environment variable name a frontend without this
    package importing one."""
    obj = resolve_callable(spec, "frontend factory")
    return lambda dtype, _f=obj: _f(dtype=dtype)


def _registered_frontend_factory():
    """the factory set with :func:`set_frontend_factory`, else the one ``LEMAS_FRONTEND_FACTORY=module:callable`` names"""
    import os
    if FRONTEND_FACTORY is not None:
        return FRONTEND_FACTORY
    spec = os.environ.get("LEMAS_FRONTEND_FACTORY")
    return resolve_frontend_factory(spec) if spec else None


class TTS:
    def __init__(self, model="multilingual_grl", ckpt_file="", vocab_file="", ode_method="euler", use_ema=False,
                 vocoder_local_path=None, use_prosody_encoder=False, prosody_cfg_path="", prosody_ckpt_path="",
                 device=None, hf_cache_dir=None, frontend=None, *, state_dict=None, vocoder_state_dict=None,
                 vocab_char_map=None):
        cfg = load_arch_config(model)                                   # api.py:99-105
        self.mel_spec_type = cfg["mel_spec"]["mel_spec_type"]
        self.target_sample_rate = cfg["mel_spec"]["target_sample_rate"]
        self.ode_method, self.use_ema = ode_method, use_ema
        self.langs = {"cmn": "zh", "zh": "zh", "en": "en-us", "it": "it", "es": "es", "pt": "pt-br", "fr": "fr-fr",
                      "de": "de", "ru": "ru", "id": "id", "vi": "vi", "th": "th"}
        if device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("lemas_tts_amd has no CPU path: an MI355X ('cuda') device is required")
            device = "cuda:0"
        self.device = device
        is_local = vocoder_local_path is not None and Path(str(vocoder_local_path)).is_dir()
        self.vocoder = load_vocoder(self.mel_spec_type, is_local, vocoder_local_path, self.device, hf_cache_dir,
                                    state_dict=vocoder_state_dict)          # api.py:136
        if isinstance(frontend, str):
            # api.py:140-151 builds TextNorm(dtype=frontend) from the reference's own host-side frontend package (espeak / jieba /
            # langid).  The text frontend is out of this build's scope and the product does not import the reference package:
            # the integrator registers a factory once, or hands the OBJECT over.
            factory = _registered_frontend_factory()
            if factory is None:
                raise TypeError(f"frontend={frontend!r}: no text frontend is registered.  Call lemas_tts_amd.api.set_frontend_factory("
                                "lambda dtype: TextNorm(dtype=dtype)) once, set LEMAS_FRONTEND_FACTORY=package.module:callable (the "
                                "command-line entry points take --frontend_factory), or pass a frontend OBJECT with text2phn() / "
                                "text2norm() and a .dtype of 'phone' or 'char', or frontend=None with phone-token lists")
            frontend = factory(frontend)
        self.frontend = frontend
        self.ema_model = load_model(None, cfg["arch"], ckpt_file, self.mel_spec_type, vocab_file, self.ode_method,
                                    self.use_ema, self.device, use_prosody_encoder=use_prosody_encoder,
                                    prosody_cfg_path=prosody_cfg_path, prosody_ckpt_path=prosody_ckpt_path,
                                    state_dict=state_dict, vocab_char_map=vocab_char_map)   # api.py:154
        self.seed = None

    def export_wav(self, wav, file_wave, remove_silence=False):
        """api.py:162-166: ``soundfile.write(file_wave, wav, sr)`` (16-bit PCM for .wav); silence removal (pydub) is out of scope."""
        if remove_silence:
            raise NotImplementedError("remove_silence needs pydub (utils_infer.py:629-640): not on the MI355X path")
        save_wav(file_wave, wav, self.target_sample_rate, "PCM_16")

    def export_spectrogram(self, spec, file_spec):
        """api.py:168-169 renders the mel with matplotlib; without it the array itself is stored (``.npy``)."""
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except ImportError:
            np.save(str(file_spec) if str(file_spec).endswith(".npy") else str(file_spec) + ".npy", np.asarray(spec))
            return
        plt.figure(figsize=(12, 4))
        plt.imshow(spec, origin="lower", aspect="auto")
        plt.colorbar()
        plt.savefig(file_spec)
        plt.close()

    def infer(self, ref_file, ref_text, gen_text, show_info=print, progress=None, target_rms=0.1,
              cross_fade_duration=0.15, use_acc_grl=False, ref_ratio=None, no_ref_audio=False, cfg_strength=2,
              nfe_step=32, speed=1.0, sway_sampling_coef=5, separate_langs=False, fix_duration=None,
              use_prosody_encoder=True, file_wave=None, file_spec=None, seed=None, **extra):
        if seed is None:
            seed = random.randint(0, sys.maxsize)                           # api.py:194-197
        seed_everything(seed)
        self.seed = seed
        if self.frontend is not None and isinstance(ref_text, str):
            kind = getattr(self.frontend, "dtype", "phone")
            if kind == "phone":                                             # api.py:201-204
                ref_text = self.frontend.text2phn(ref_text + ". ").replace("(cmn)", "(zh)").split("|")
                gen_text = [self.frontend.text2phn(x + ". ").replace("(cmn)", "(zh)").split("|") for x in gen_text.split("\n")]
            elif kind == "char":                                            # api.py:206-211: language tag + characters
                lang, norm = self.frontend.text2norm(ref_text + ". ")
                ref_text = ["(" + lang.replace("cmn", "zh") + ")"] + list(norm)
                pairs = [self.frontend.text2norm(x + ". ") for x in gen_text.split("\n")]
                gen_text = [["(" + lg.replace("cmn", "zh") + ")"] + list(tx) for lg, tx in pairs]
            else:
                raise NotImplementedError(f"frontend dtype {kind!r}")
        if separate_langs:
            ref_text = self.process_phone_list(ref_text)                    # api.py:214-216
            gen_text = [self.process_phone_list(x) for x in gen_text]
        wav, sr, spec = infer_process(
            ref_file, ref_text, gen_text, self.ema_model, self.vocoder, self.mel_spec_type, show_info=show_info,
            progress=progress, target_rms=target_rms, cross_fade_duration=cross_fade_duration, nfe_step=nfe_step,
            cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, use_prosody_encoder=use_prosody_encoder,
            use_acc_grl=use_acc_grl, ref_ratio=ref_ratio, no_ref_audio=no_ref_audio, speed=speed,
            fix_duration=fix_duration, device=self.device, **extra)   # no per-line re-seeding: like api.py:194-197 + utils_infer.py:531-542,
        # every line draws its noise from the generator seed_everything() just seeded, in order
        if file_wave is not None:
            self.export_wav(wav, file_wave)
        if file_spec is not None:
            self.export_spectrogram(spec, file_spec)
        return wav, sr, spec

    def process_phone_list(self, parts):
        """api.py:252-276: prefix every non-punctuation phone with the current language id."""
        puncs = {"#1", "#2", "#3", "#4", "_", "!", ",", ".", "?", '"', "'", "^", "。", "，", "？", "！"}
        processed, current_lang = [], ""
        for part in parts:
            if part.startswith("(") and part.endswith(")") and part[1:-1] in self.langs:
                current_lang = part
            elif part in puncs:
                if processed and processed[-1] == "_":
                    processed.pop()
                elif processed and processed[-1] in puncs and part == "_":
                    continue
                processed.append(part)
            elif current_lang is not None:
                processed.append(f"{current_lang}{part}")
        return processed
