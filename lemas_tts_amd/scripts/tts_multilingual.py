"""Host-side mirror of the north-star entry point ``lemas_tts/scripts/tts_multilingual.py``: the same
command line, defaults and call order -- resolve checkpoint and vocab, build ``TTS``, ``tts.infer(..., file_wave=...)`` --
with the sampler and vocoder behind it running on the MI355X engines.

    python -m lemas_tts_amd.scripts.tts_multilingual --ref_audio ref.wav --ref_text "..." --text "..." --output_wave out.wav

What differs from the reference script, and why:

* ``--denoise`` (UVR5 MDX-Net, ``tts_multilingual.py:38-86,303-314``): the reference runs ``pretrained_models/uvr5/Kim_Vocal_1.onnx``
  through onnxruntime on the CPU; here the same directory (``--denoise_model``, default ``PRETRAINED_ROOT/uvr5``) feeds the HIP
  implementation of the network the ONNX file was exported from (``lemas_tts_amd/uvr5``, ``lemas_mdx_*``).
  ``--denoise_model`` may also name a network file directly (``.onnx`` or a state dict of the module), ``--denoise_config`` the reference's
  json files, and ``--denoise_model_factory package.module:callable`` a function returning ``model_run(spek) -> spec_pred`` to use instead;
* there is no CPU retry (``:332-336``): this build has no CPU path, a missing GPU is an error;
* checkpoints resolve locally only (``:89-119`` falls back to a Hugging Face download; no network here);
* the text frontend comes from the factory registered with ``lemas_tts_amd.api.set_frontend_factory``; ``--ref_phones`` / ``--phones``
  (an extension) take text that is ALREADY phonemised -- ``p1|p2|...``, one generated line per ``\\n`` -- and skip the
  frontend, which is how the entry point is exercised on a box without espeak.
"""
from __future__ import annotations

import argparse
import os
from typing import List, Optional

from ..api import CKPTS_ROOT, PRETRAINED_ROOT, TTS


def _resolve_ckpt(model_name: str, ckpt_file: Optional[str]) -> str:
    """``:89-119`` without the remote fallback: explicit path, else the newest ``*.safetensors`` / ``*.pt`` under
    ``CKPTS_ROOT/<model>``, else ``CKPTS_ROOT/<model>*``."""
    if ckpt_file:
        return ckpt_file
    ckpt_dir = CKPTS_ROOT / model_name
    found = sorted(list(ckpt_dir.glob("*.safetensors")) + list(ckpt_dir.glob("*.pt")))
    if not found:
        found = sorted(list(CKPTS_ROOT.glob(f"{model_name}*.safetensors")) + list(CKPTS_ROOT.glob(f"{model_name}*.pt")))
    if not found:
        raise FileNotFoundError(f"No ckpt found for model '{model_name}' under {CKPTS_ROOT} (no network: pass --ckpt_file)")
    return str(found[-1])


def _resolve_vocab(model_name: str, vocab_file: Optional[str]) -> str:
    """``:122-128``."""
    if vocab_file:
        return vocab_file
    vf = PRETRAINED_ROOT / "data" / model_name / "vocab.txt"
    if not vf.is_file():
        raise FileNotFoundError(f"Vocab file not found: {vf}")
    return str(vf)


def build_tts(model_name: str, ckpt_file: str, vocab_file: str, device: Optional[str], use_ema: bool, frontend,
              enable_prosody: bool, prosody_cfg_path: str = "", prosody_ckpt_path: str = "", vocoder_local_path=None) -> TTS:
    """``:131-172``: the ``*grl`` models never use the prosody encoder, the others when it is enabled and its files exist."""
    prosody_cfg = prosody_cfg_path or str(CKPTS_ROOT / "prosody_encoder" / "pretssel_cfg.json")
    prosody_ckpt = prosody_ckpt_path or str(CKPTS_ROOT / "prosody_encoder" / "prosody_encoder_UnitY2.pt")
    if enable_prosody and not (os.path.isfile(prosody_cfg) and os.path.isfile(prosody_ckpt)):
        raise FileNotFoundError(f"Prosody encoder assets not found: {prosody_cfg}, {prosody_ckpt}")
    use_prosody_encoder = False if model_name.endswith("grl") else enable_prosody
    return TTS(model=model_name, ckpt_file=ckpt_file, vocab_file=vocab_file, device=device, use_ema=use_ema, frontend=frontend,
               use_prosody_encoder=use_prosody_encoder, prosody_cfg_path=prosody_cfg if use_prosody_encoder else "",
               prosody_ckpt_path=prosody_ckpt if use_prosody_encoder else "",
               vocoder_local_path=vocoder_local_path if vocoder_local_path else CKPTS_ROOT / "vocos-mel-24khz")


def build_parser() -> argparse.ArgumentParser:
    """The reference's argument list (``:175-296``), same names, types and defaults; ``--ref_phones`` / ``--phones`` /
    ``--vocoder_local_path`` / ``--device`` are additions."""
    p = argparse.ArgumentParser(prog="python -m lemas_tts_amd.scripts.tts_multilingual",
                                description="Multilingual zero-shot TTS with LEMAS-TTS on MI355X.")
    p.add_argument("--model", type=str, default="multilingual_grl")
    p.add_argument("--ckpt_file", type=str, default="")
    p.add_argument("--vocab_file", type=str, default="")
    p.add_argument("--frontend", type=str, default="phone", choices=["phone", "char"])
    p.add_argument("--use_ema", action="store_true")
    p.add_argument("--enable_prosody_encoder", action="store_true")
    p.add_argument("--prosody_cfg_path", type=str, default="")
    p.add_argument("--prosody_ckpt_path", type=str, default="")
    p.add_argument("--ref_audio", type=str, required=True)
    p.add_argument("--ref_text", type=str, default=None)
    p.add_argument("--text", type=str, default=None)
    p.add_argument("--output_wave", type=str, default="output.wav")
    p.add_argument("--denoise", action="store_true")
    p.add_argument("--nfe_step", type=int, default=64)
    p.add_argument("--cfg_strength", type=float, default=5.0)
    p.add_argument("--sway_sampling_coef", type=float, default=3.0)
    p.add_argument("--ref_ratio", type=float, default=1.0)
    p.add_argument("--no_ref_audio", action="store_true")
    p.add_argument("--separate_langs", action="store_true")
    p.add_argument("--speed", type=float, default=1.0)
    p.add_argument("--use_acc_grl", action="store_true")
    p.add_argument("--seed", type=int, default=-1)
    # additions
    p.add_argument("--ref_phones", type=str, default=None, help="reference transcript, already phonemised: 'p1|p2|...'")
    p.add_argument("--phones", type=str, default=None, help="text to generate, already phonemised; '\\n' separates lines")
    p.add_argument("--vocoder_local_path", type=str, default="", help="vocos directory (config.yaml + pytorch_model.bin)")
    p.add_argument("--device", type=str, default=None, help="cuda:N (default cuda:0)")
    p.add_argument("--frontend_factory", type=str, default=None,
                   help="'package.module:callable' building the text frontend for --frontend phone|char, called as callable(dtype=...): "
                        "e.g. lemas_tts.infer.frontend:TextNorm.  Default: the LEMAS_FRONTEND_FACTORY environment variable.  The text "
                        "frontend (espeak / jieba / langid) is host Python outside this package")
    p.add_argument("--denoise_model", type=str, default=None,
                   help="--denoise: directory laid out like the reference's pretrained_models/uvr5 (Kim_Vocal_1.onnx + MDX-Net-Kim-Vocal1.json "
                        "[+ model_data.json]) or a network file (.onnx, or a ConvTDFNet state dict as .safetensors/.pt/.ckpt/.npz); "
                        "default PRETRAINED_ROOT/uvr5")
    p.add_argument("--denoise_model_factory", type=str, default=None,
                   help="'package.module:callable' returning the MDX-Net as model_run(spek[b, 4, dim_f, dim_t]) -> spec_pred for --denoise")
    p.add_argument("--denoise_config", type=str, nargs="*", default=None,
                   help="the reference's MDX-Net json files (model_data entry, MDX-Net-Kim-Vocal1.json); default: published Kim_Vocal_1 values")
    return p


def _phone_lines(s: str) -> List[List[str]]:
    return [[p for p in line.split("|") if p] for line in s.replace("\\n", "\n").split("\n") if line.strip()]


def main(argv=None) -> int:
    args = build_parser().parse_args(argv)
    phones_given = args.ref_phones is not None or args.phones is not None
    if phones_given and (args.ref_phones is None or args.phones is None):
        raise SystemExit("--ref_phones and --phones go together")
    if not phones_given and (args.ref_text is None or args.text is None):
        raise SystemExit("--ref_text and --text are required (or --ref_phones and --phones)")

    ckpt_file = _resolve_ckpt(args.model, args.ckpt_file or None)
    vocab_file = _resolve_vocab(args.model, args.vocab_file or None)
    if not os.path.isfile(args.ref_audio):
        raise FileNotFoundError(f"Reference audio not found: {args.ref_audio}")

    if args.frontend_factory:
        from ..api import resolve_frontend_factory, set_frontend_factory
        set_frontend_factory(resolve_frontend_factory(args.frontend_factory))
    ref_audio, tmp_denoised = args.ref_audio, None
    if args.denoise:             # :303-314: the prompt goes through UVR5 first, the temporary file is removed at the end
        from ..uvr5 import MDXConfig, UVR5
        cfg = MDXConfig.from_json(*args.denoise_config) if args.denoise_config else None
        if args.denoise_model_factory:
            from ..api import resolve_callable
            network = resolve_callable(args.denoise_model_factory, "denoise model factory")()
        else:
            network = args.denoise_model or str(PRETRAINED_ROOT / "uvr5")
            if not os.path.exists(network):
                raise FileNotFoundError(f"--denoise: {network} does not exist (the reference keeps Kim_Vocal_1.onnx and its json files "
                                        "there); point --denoise_model at the directory or at a network file")
        tmp_denoised = UVR5(network, cfg, device=args.device or "cuda:0").denoise_file(ref_audio)
        ref_audio = tmp_denoised
    try:                         # everything after the temporary file exists runs under the finally that removes it
        tts = build_tts(args.model, ckpt_file, vocab_file, args.device, args.use_ema, None if phones_given else args.frontend,
                        args.enable_prosody_encoder, args.prosody_cfg_path, args.prosody_ckpt_path, args.vocoder_local_path)
        if phones_given:
            ref_text, gen_text = _phone_lines(args.ref_phones)[0], _phone_lines(args.phones)
        else:
            ref_text, gen_text = args.ref_text.strip(), args.text.strip()
        seed = None if args.seed == -1 else args.seed
        tts.infer(ref_file=ref_audio, ref_text=ref_text, gen_text=gen_text, nfe_step=int(args.nfe_step),
                  cfg_strength=float(args.cfg_strength), sway_sampling_coef=float(args.sway_sampling_coef),
                  use_acc_grl=bool(args.use_acc_grl), ref_ratio=float(args.ref_ratio), no_ref_audio=bool(args.no_ref_audio),
                  separate_langs=bool(args.separate_langs), speed=float(args.speed),
                  use_prosody_encoder=args.enable_prosody_encoder, file_wave=args.output_wave, seed=seed)
    finally:
        if tmp_denoised is not None and os.path.isfile(tmp_denoised):
            os.remove(tmp_denoised)
    print(f"Saved synthesized audio to: {args.output_wave}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
