"""MI355X mirror of the reference's UVR5 MDX-Net prompt denoiser (``uvr5/multiprocess_cuda_infer.py:181-301`` ``Inference``; wrapper
``lemas_tts/scripts/tts_multilingual.py:38-86`` ``UVR5``), same method names and argument meaning:

* the shell AROUND the separation network -- stereo chunking with trimmed overlaps, STFT to the network's [batch, 4, dim_f, dim_t] layout,
  the +-input averaging ("denoise") trick, inverse STFT, margin handling.  The two transforms run in ``liblemas_hip.so`` (``lemas_stft_*``:
  fp32 MFMA GEMMs against precomputed bases); slicing, padding and concatenation are tensor plumbing; resampling to 44.1 kHz uses
  ``lemas_resample_*``;
* the network itself.  The reference runs ``Kim_Vocal_1.onnx`` through onnxruntime (:225-238); that file is an export of ``ConvTDFNet``
  (``uvr5/lib_v5/mdxnet.py:36-127``), and ``Inference.load_model`` here builds that architecture on the HIP engine (``lemas_mdx_*``,
  csrc/engine_mdx.hip) from the ONNX file's initializers or from the module's state dict -- no onnxruntime involved.  A callable is still
  accepted in its place (tests use it to exercise the shell alone).
"""
from __future__ import annotations

import os
import tempfile
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import numpy as np
import torch

from ..engine import StftEngine, resampler

MODEL_RATE = 44100


@dataclass
class MDXConfig:
    """The fields of the reference's ``ModelData`` that ``Inference`` reads (multiprocess_cuda_infer.py:183-204).  The reference fills
    them from ``MDX-Net-Kim-Vocal1.json`` / ``model_data.json`` in its pretrained directory; neither file is in the tree, so the defaults
    here are the values UVR publishes for Kim_Vocal_1 -- an assumption, override from the real json (``from_json``)."""
    mdx_n_fft_scale_set: int = 7680
    mdx_dim_f_set: int = 3072
    mdx_dim_t_set: int = 8            # dim_t = 2 ** this
    compensate: float = 1.043
    is_normalization: bool = False
    is_denoise: bool = False
    mdx_batch_size: int = 1
    chunks: int = 0
    margin: int = 44100
    save_background: bool = False

    @classmethod
    def from_json(cls, *paths) -> "MDXConfig":
        """Merge the reference's json files (later wins), keeping the keys this class knows."""
        import json
        cfg = cls()
        for p in paths:
            with open(p, "r", encoding="utf-8") as f:
                for k, v in json.load(f).items():
                    if hasattr(cfg, k):
                        setattr(cfg, k, type(getattr(cfg, k))(v))
        return cfg


class Inference:
    def __init__(self, model_data: MDXConfig, device="cuda:0"):
        self.device = torch.device(device)
        self.n_fft = int(model_data.mdx_n_fft_scale_set)
        self.is_normalization = model_data.is_normalization
        self.compensate = model_data.compensate
        self.dim_f, self.dim_t = int(model_data.mdx_dim_f_set), 2 ** int(model_data.mdx_dim_t_set)
        self.mdx_batch_size = int(model_data.mdx_batch_size)
        self.is_denoise = bool(model_data.is_denoise)
        self.hop = 1024
        self.dim_c = 4
        self.chunks = model_data.chunks
        self.margin = int(model_data.margin)
        self.adjust = 1
        self.n_bins = self.n_fft // 2 + 1
        self.trim = self.n_fft // 2
        self.chunk_size = self.hop * (self.dim_t - 1)
        self.gen_size = self.chunk_size - 2 * self.trim
        if self.gen_size <= 0:
            raise ValueError("chunk (hop * (dim_t - 1)) must be longer than n_fft")
        self.window = torch.hann_window(self.n_fft, periodic=False)
        self.save_background = model_data.save_background
        self._stft = StftEngine(self.n_fft, self.hop, self.window, device=self.device)
        self.model_run: Optional[Callable] = None

    # ---- transforms (multiprocess_cuda_infer.py:206-223) ---------------------------------------------------------
    def stft(self, x: torch.Tensor) -> torch.Tensor:
        """[b, 2, chunk] -> [b, 4, dim_f, dim_t]; channel order (left re, left im, right re, right im)."""
        x = x.reshape(-1, self.chunk_size)
        spec = self._stft.forward(x)                                         # [b*2, n_bins, dim_t] complex
        planes = torch.stack((spec.real, spec.imag), dim=1)                  # [b*2, 2, n_bins, dim_t]
        return planes.reshape(-1, self.dim_c, self.n_bins, self.dim_t)[:, :, :self.dim_f]

    def istft(self, x: torch.Tensor, freq_pad: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[b, 4, dim_f, dim_t] -> [b, 2, chunk]; bins above dim_f are zero (or ``freq_pad``)."""
        b = x.shape[0]
        if freq_pad is None:
            freq_pad = torch.zeros((b, self.dim_c, self.n_bins - self.dim_f, self.dim_t), device=x.device, dtype=x.dtype)
        full = torch.cat((x, freq_pad), dim=-2).reshape(b * 2, 2, self.n_bins, self.dim_t)
        wav = self._stft.inverse(torch.complex(full[:, 0].contiguous(), full[:, 1].contiguous()))
        return wav.reshape(b, 2, self.chunk_size)

    # ---- the network ---------------------------------------------------------------------------------------------
    def load_model(self, model, threads: int = 1, device=None, bf16x3: bool = False):
        """The reference builds an onnxruntime session from ``model_path`` here (:225-240).  This mirror runs the network the ONNX file
        was exported from -- ConvTDFNet, uvr5/lib_v5/mdxnet.py:36-127 -- on the HIP engine (``lemas_mdx_*``).  ``model`` is
          * a path: ``*.onnx`` (initializers read by ``onnx_weights.load_onnx``, no onnx / onnxruntime package involved), or a state dict
            of the module as ``*.safetensors`` / ``*.pt`` / ``*.ckpt`` / ``*.npz`` (hyper-parameters inferred from the tensor shapes);
          * ``(arch, state_dict)``: the ConvTDFNet constructor arguments (object or dict, see ``MdxEngine``) and its state dict;
          * a callable ``model_run(spek[b, 4, dim_f, dim_t]) -> spec_pred`` (tensor or numpy), used as is.
        ``threads`` is the reference's onnxruntime intra-op thread count: meaningless here, accepted.  ``bf16x3`` (an addition): the engine's
        split-bf16 3x3 convolutions (``MdxEngine``), ~2^-16 relative precision per product instead of exact fp32, about twice the speed."""
        if callable(model):
            def run(spek: torch.Tensor) -> torch.Tensor:
                out = model(spek)
                if isinstance(out, np.ndarray):
                    out = torch.from_numpy(out)
                return out.to(self.device, torch.float32)
            self.model_run = run
            return
        from ..engine import MdxEngine
        from . import onnx_weights
        if isinstance(model, (str, os.PathLike)):
            arch, sd = onnx_weights.load_network_file(os.fspath(model), dim_t=self.dim_t)
        elif isinstance(model, (tuple, list)) and len(model) == 2:
            arch, sd = model
        else:
            raise TypeError("load_model: a path, an (arch, state_dict) pair or a callable")
        get = (lambda k: arch[k]) if isinstance(arch, dict) else (lambda k: getattr(arch, k))
        if (int(get("dim_c")), int(get("dim_f")), int(get("dim_t"))) != (self.dim_c, self.dim_f, self.dim_t):
            raise ValueError(f"the network is built for [b, {get('dim_c')}, {get('dim_f')}, {get('dim_t')}] but the configuration cuts "
                             f"[b, {self.dim_c}, {self.dim_f}, {self.dim_t}] spectrograms")
        self.network = MdxEngine(arch, sd, device=self.device, bf16x3=bf16x3)
        self.model_run = self.network.forward

    # ---- chunking (multiprocess_cuda_infer.py:243-258) -------------------------------------------------------------
    def initialize_mix(self, mix: torch.Tensor):
        """[2, n] -> ([chunks, 2, chunk_size], pad): windows of chunk_size every gen_size samples over the mix framed by `trim` zeros
        and padded to a whole number of gen_size pieces (a FULL extra piece when n is already a multiple, as the reference)."""
        n = mix.shape[1]
        pad = self.gen_size - n % self.gen_size
        framed = torch.nn.functional.pad(mix, (self.trim, pad + self.trim))
        starts = range(0, n + pad, self.gen_size)
        waves = torch.stack([framed[:, i:i + self.chunk_size] for i in starts], dim=0)
        return waves.to(self.device), pad

    def run_model(self, mix: torch.Tensor, is_match_mix: bool = False) -> torch.Tensor:
        """[b, 2, chunk] -> [2, b * gen_size]: the network's output (or, is_match_mix, the band-limited input itself) back in time."""
        spek = self.stft(mix.to(self.device)) * self.adjust
        spek[:, :, :3, :] = 0                                   # the three lowest bins are always dropped (:264)
        if is_match_mix:
            pred = spek
        else:
            if self.model_run is None:
                raise RuntimeError("load_model() has not been given a network")
            if self.is_denoise and getattr(self, "network", None) is not None:
                both = self.model_run(torch.cat((spek, -spek), dim=0))          # the +- pair of :269 as one batch of the engine
                pred = (both[:spek.shape[0]] - both[spek.shape[0]:]) * 0.5
            else:
                pred = (self.model_run(spek) - self.model_run(-spek)) * 0.5 if self.is_denoise else self.model_run(spek)
        wav = self.istft(pred)[:, :, self.trim:-self.trim]
        return wav.transpose(0, 1).reshape(2, -1)

    def demix_base(self, mix: Dict[int, torch.Tensor], is_match_mix: bool = False, device=None) -> torch.Tensor:
        """{slice id: [2, n]} -> [2, n'].  Margins: every slice but the first drops `margin` samples at its start, every slice but the
        last at its end.  As in the reference (:278-301) the returned tensor is that of the LAST slice of the dict -- the callers on
        the path pass a single slice {0: mix}."""
        keys = list(mix.keys())
        result = None
        for key in keys:
            waves, pad = self.initialize_mix(mix[key].to(self.device))
            with torch.no_grad():
                parts = [self.run_model(w, is_match_mix=is_match_mix) for w in waves.split(self.mdx_batch_size)]
            tar = torch.cat(parts, dim=-1)[:, :-pad]
            start = 0 if key == 0 else self.margin
            end = None if key == keys[-1] or self.margin == 0 else -self.margin
            result = tar[:, start:end] * (1 / self.adjust)
        return result


def normalize_two_stem(wave: torch.Tensor, mix: torch.Tensor, is_normalize: bool = False):
    """multiprocess_cuda_infer.py:337-350: scale both stems by the primary stem's peak when it clips and normalisation is on."""
    peak = wave.abs().max()
    if peak > 1.0 and is_normalize:
        wave, mix = wave / peak, mix / peak
    return wave, mix


MODEL_STEM, UI_CONFIG = "Kim_Vocal_1", "MDX-Net-Kim-Vocal1.json"      # tts_multilingual.py:55-56
MODEL_EXTS = (".onnx", ".safetensors", ".pt", ".ckpt", ".npz")


def model_hash(model_path: str) -> str:
    """``ModelData.get_model_hash`` (multiprocess_cuda_infer.py:165-178): md5 of the file's last 10 000 KiB, of the whole file if shorter."""
    import hashlib
    with open(model_path, "rb") as f:
        try:
            f.seek(-10000 * 1024, 2)
        except OSError:
            f.seek(0)
        return hashlib.md5(f.read()).hexdigest()


def resolve_model_dir(model_dir: str):
    """The reference's ``pretrained_models/uvr5`` layout (tts_multilingual.py:55-69) -> (network file, MDXConfig): ``Kim_Vocal_1.onnx`` (or
    the same stem as a state-dict file), the UI options in ``MDX-Net-Kim-Vocal1.json`` and the model's entry of ``model_data.json``
    (keyed by ``model_hash``; ModelData :108-121), each optional -- absent files leave the published Kim_Vocal_1 defaults."""
    import json
    path = next((os.path.join(model_dir, MODEL_STEM + e) for e in MODEL_EXTS if os.path.isfile(os.path.join(model_dir, MODEL_STEM + e))), None)
    if path is None:
        raise FileNotFoundError(f"{model_dir}: no {MODEL_STEM}{{{','.join(MODEL_EXTS)}}} (the UVR5 MDX-Net weights the reference keeps in "
                                "pretrained_models/uvr5)")
    ui = os.path.join(model_dir, UI_CONFIG)
    cfg = MDXConfig.from_json(ui) if os.path.isfile(ui) else MDXConfig()
    table = os.path.join(model_dir, "model_data.json")
    if os.path.isfile(table):
        with open(table, "r", encoding="utf-8") as f:
            entry = json.load(f).get(model_hash(path))
        if entry:
            for k in ("compensate", "mdx_dim_f_set", "mdx_dim_t_set", "mdx_n_fft_scale_set"):
                if k in entry:
                    setattr(cfg, k, type(getattr(cfg, k))(entry[k]))
    return path, cfg


class UVR5:
    """``tts_multilingual.py:38-86``: denoise a prompt file.  ``model``: a directory laid out like the reference's ``pretrained_models/uvr5``
    (``resolve_model_dir``), or anything ``Inference.load_model`` takes (a network file, an ``(arch, state_dict)`` pair, a callable)."""

    def __init__(self, model, config: Optional[MDXConfig] = None, device: str = "cuda:0", bf16x3: bool = False) -> None:
        self.device = device
        if isinstance(model, (str, os.PathLike)) and os.path.isdir(model):
            model, dir_config = resolve_model_dir(os.fspath(model))
            config = config or dir_config
        self.model = Inference(config or MDXConfig(), device)
        self.model.load_model(model, 1, bf16x3=bf16x3)

    def denoise(self, wav: torch.Tensor, sr: int) -> torch.Tensor:
        """wav [channels, n] at `sr` -> vocal stem [2, n'] at 44.1 kHz."""
        wav = wav.to(self.device, torch.float32)
        if wav.shape[0] == 1:
            wav = torch.cat((wav, wav), dim=0)          # mono -> stereo
        if sr != MODEL_RATE:
            wav = resampler(sr, MODEL_RATE, device=self.device)(wav)
        return self.model.demix_base({0: wav}, is_match_mix=False, device=self.device)

    def denoise_file(self, wav_path: str) -> str:
        """Denoise a wav file and return the path of a temporary 24-bit wav holding the vocal stem."""
        from ..infer import audio_io
        wav, sr = audio_io.load_wav(wav_path)
        out = self.denoise(wav, sr).to("cpu").numpy().T            # [T, 2]
        with tempfile.NamedTemporaryFile(delete=False, suffix=".wav") as f:
            name = f.name
        audio_io.save_wav(name, out, MODEL_RATE, subtype="PCM_24")
        return name
