"""Python handles over the C ABI objects ``lemas_dit`` and ``lemas_vocos``.

Host code here is plumbing only: it moves weights to the library, owns input/output torch tensors
and a non-default HIP stream (a captured hipGraph cannot live on the legacy NULL stream).  All
arithmetic of the acoustic path happens inside liblemas_hip.so.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np
import torch

from . import _lib
from .model.layout import DiTArch, VocosArch


def _as_f32_host(v) -> np.ndarray:
    if isinstance(v, torch.Tensor):
        v = v.detach().to("cpu", torch.float32).numpy()
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32))


def _load(fn, handle, name: str, arr, fn_device=None) -> None:
    """Hand one tensor to the library: host fp32 array, or -- when ``arr`` is a CUDA tensor and the engine has a device-side
    loader -- the device pointer itself (weights that arrived by RCCL broadcast stay on the device)."""
    if fn_device is not None and isinstance(arr, torch.Tensor) and arr.is_cuda:
        t = arr.detach().to(torch.float32).contiguous()
        torch.cuda.current_stream(t.device).synchronize()      # the loader copies on the NULL stream
        shape = (C.c_int64 * t.ndim)(*t.shape)
        _lib.check(fn_device(handle, name.encode(), C.c_void_p(t.data_ptr()), shape, t.ndim), f"load_weight_device({name})")
        return
    a = _as_f32_host(arr)
    shape = (C.c_int64 * a.ndim)(*a.shape)
    _lib.check(fn(handle, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim), f"load_weight({name})")


def _aux_tables(arch: DiTArch) -> dict:
    """Tables the reference holds as non-persistent buffers or recomputes per call, built with the very
    same torch expressions so they are bit-identical to the reference's:
    ``precompute_freqs_cis(text_dim, 4096)`` (modules.py:196-207) and the sinusoid frequencies of
    ``SinusPositionEmbedding`` (modules.py:156-158)."""
    td = arch.text_dim
    freqs = 1.0 / (10000.0 ** (torch.arange(0, td, 2)[: (td // 2)].float() / td))
    ang = torch.outer(torch.arange(4096), freqs).float()
    half = arch.time_freq_dim // 2
    tf = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
    return {"transformer.text_embed.freqs_cis": torch.cat([ang.cos(), ang.sin()], dim=-1),
            "transformer.time_embed.freqs": tf}


class _Streamed:
    """Run library calls on a private non-default stream, ordered after/before torch's current stream."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.LemasError("the MI355X engine needs a HIP device ('cuda:N'); there is no CPU path")
        self.stream = torch.cuda.Stream(self.device)

    def _enter(self, *tensors):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        for t in tensors:
            if t is not None:
                t.record_stream(self.stream)
        return C.c_void_p(self.stream.cuda_stream)

    def _exit(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)


class DiTEngine(_Streamed):
    """Owns one ``lemas_dit`` (weights, workspaces, hipGraphs) on one device."""

    def __init__(self, arch: DiTArch, vocab_size: int, state_dict: dict, device="cuda:0", prosody: bool = False):
        super().__init__(device)
        self.arch, self.vocab_size, self.prosody = arch, vocab_size, prosody
        L = _lib.lib()
        cfg = _lib.DitConfig(arch.dim, arch.depth, arch.heads, arch.dim_head, arch.ff_mult, arch.text_dim,
                             arch.conv_layers, arch.mel_dim, vocab_size + 1, arch.conv_pos_kernel,
                             arch.conv_pos_groups, arch.time_freq_dim, int(prosody))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.lemas_dit_create(C.byref(cfg), C.byref(self._h)), "lemas_dit_create")
            for name, v in state_dict.items():
                if name.startswith("prosody_encoder.") or name.startswith("mel_spec."):
                    continue  # the prosody ENCODER is a "next" row; mel_spec keys are dropped by the reference loader
                if not prosody and (name.startswith("prosody_to_mel.") or ".prosody_text_proj." in name):
                    raise _lib.LemasError(f"checkpoint has prosody tensor '{name}' but the model was built without it")
                _load(L.lemas_dit_load_weight, self._h, name, v, L.lemas_dit_load_weight_device)
            for name, v in _aux_tables(arch).items():
                _load(L.lemas_dit_load_weight, self._h, name, v)
            _lib.check(L.lemas_dit_finalize(self._h), "lemas_dit_finalize")

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().lemas_dit_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key: str, value: int):
        _lib.check(_lib.lib().lemas_dit_set_option(self._h, key.encode(), int(value)), f"set_option({key})")
        self.__dict__.setdefault("_options", {})[key] = int(value)

    def option(self, key: str, default: int = 0) -> int:
        """the value this object last set for ``key`` (``default``: the library's own default when it was never set here)"""
        return self.__dict__.get("_options", {}).get(key, default)

    def stat(self, key: str) -> int:
        """counters of the engine's step-graph cache etc. (``lemas_dit_get_stat``)"""
        v = C.c_int64()
        _lib.check(_lib.lib().lemas_dit_get_stat(self._h, key.encode(), C.byref(v)), f"get_stat({key})")
        return int(v.value)

    def check_health(self):
        """Synchronise and raise if a device-side wait of this engine gave up (results since then would be invalid)."""
        _lib.check(_lib.lib().lemas_dit_health(self._h), "lemas_dit_health")

    # ------------------------------------------------------------------------------------------
    def _args(self, cond, cond_mask, text, seq_len, prosody, t_grid, cfg_strength, cond_frames, y, out, traj, step_cond=None,
              prosody_text_only=False, y_init=None):
        B, N = cond_mask.shape                          # cond may hold fewer rows than N (cond_rows): the library pads
        tg = np.ascontiguousarray(np.asarray(t_grid, dtype=np.float32))
        self._tg_keep = tg
        a = _lib.SampleArgs()
        a.batch, a.frames, a.cond_frames, a.text_len, a.steps = B, N, int(cond_frames), text.shape[1], tg.shape[0] - 1
        a.cfg_strength = float(cfg_strength)
        a.cond, a.cond_mask, a.text = cond.data_ptr(), cond_mask.data_ptr(), text.data_ptr()
        a.seq_len = seq_len.data_ptr() if seq_len is not None else None
        a.prosody = prosody.data_ptr() if prosody is not None else None
        a.t_grid = tg.ctypes.data_as(C.POINTER(C.c_float))
        a.y = y.data_ptr() if y is not None else None
        a.out = out.data_ptr() if out is not None else None
        a.trajectory = traj.data_ptr() if traj is not None else None
        a.step_cond = step_cond.data_ptr() if step_cond is not None else None
        a.prosody_text_only = 1 if prosody_text_only else 0
        a.cond_rows = cond.shape[1]
        if step_cond is not None:
            assert step_cond.shape[1] == cond.shape[1], "cond and step_cond share cond_rows"
        a.y_init = y_init.data_ptr() if y_init is not None else None
        return a

    def _canon(self, cond, cond_mask, text, seq_len, prosody):
        dev = self.device
        cond = cond.to(dev, torch.float32).contiguous()
        cond_mask = cond_mask.to(torch.uint8).to(dev).contiguous()      # (converted where it lives -- the host, in the mirror's flow -- not by a torch kernel on the device)
        text = text.to(dev, torch.int64).contiguous()
        seq_len = None if seq_len is None else seq_len.to(dev, torch.int32).contiguous()
        prosody = None if prosody is None else prosody.to(dev, torch.float32).contiguous()
        return cond, cond_mask, text, seq_len, prosody

    def sample(self, cond, cond_mask, text, t_grid, y0, *, cond_frames: int, cfg_strength: float,
               seq_len: Optional[torch.Tensor] = None, prosody: Optional[torch.Tensor] = None,
               want_trajectory: bool = False, step_cond: Optional[torch.Tensor] = None, prosody_text_only: bool = False):
        """cond [B,R,mel] reference mel, R <= N rows per sample (rows R..N-1 are the reference's zero padding, applied by the
        library); cond_mask [B,N] bool; text [B,Nt] int64 (-1 pad); y0 [B,N,mel] (left untouched);
        step_cond [B,R,mel] or None: the accent-GRL conditioning (cfm.py:387-388) when it differs from cond.
        Returns (out, y_final, trajectory|None) as device tensors."""
        cond, cond_mask, text, seq_len, prosody = self._canon(cond, cond_mask, text, seq_len, prosody)
        step_cond = None if step_cond is None else step_cond.to(self.device, torch.float32).contiguous()
        with torch.cuda.device(self.device):
            y_init = y0.to(self.device, torch.float32).contiguous()     # no copy when the caller's noise is already a device tensor
            y = torch.empty_like(y_init)
            out = torch.empty_like(y)
            S = len(t_grid) - 1
            traj = torch.empty((S + 1,) + tuple(y.shape), device=self.device, dtype=torch.float32) if want_trajectory else None
            a = self._args(cond, cond_mask, text, seq_len, prosody, t_grid, cfg_strength, cond_frames, y, out, traj, step_cond,
                           prosody_text_only, y_init=y_init)
            s = self._enter(cond, cond_mask, text, seq_len, prosody, y, y_init, out, traj, step_cond)
            _lib.check(_lib.lib().lemas_dit_sample(self._h, C.byref(a), s), "lemas_dit_sample")
            self._exit()
        return out, y, traj

    def prepare(self, cond, cond_mask, text, t_grid, *, cond_frames: int, cfg_strength: float,
                seq_len=None, prosody=None):
        cond, cond_mask, text, seq_len, prosody = self._canon(cond, cond_mask, text, seq_len, prosody)
        self._prep_keep = (cond, cond_mask, text, seq_len, prosody)
        with torch.cuda.device(self.device):
            a = self._args(cond, cond_mask, text, seq_len, prosody, t_grid, cfg_strength, cond_frames, None, None, None)
            s = self._enter(cond, cond_mask, text, seq_len, prosody)
            _lib.check(_lib.lib().lemas_dit_prepare(self._h, C.byref(a), s), "lemas_dit_prepare")
            self._exit()
        self._prep_args = a

    def solve(self, y0, want_out: bool = True):
        """Step loop on the prepared state (bench hot region).  Returns (out|None, y_final)."""
        cond, cond_mask, *_ = self._prep_keep
        with torch.cuda.device(self.device):
            y_init = y0.to(self.device, torch.float32).contiguous()
            y = torch.empty_like(y_init)
            out = torch.empty_like(y) if want_out else None
            a = self._prep_args
            a.y, a.out, a.trajectory, a.y_init = y.data_ptr(), (out.data_ptr() if out is not None else None), None, y_init.data_ptr()
            s = self._enter(y, y_init, out)
            _lib.check(_lib.lib().lemas_dit_solve(self._h, C.byref(a), s), "lemas_dit_solve")
            self._exit()
        return out, y

    def forward(self, x, step_index: int):
        """One DiT forward (both CFG branches) at step ``step_index`` of the prepared grid -> [BB, N, mel]."""
        cond, cond_mask = self._prep_keep[0], self._prep_keep[1]
        B, N = cond_mask.shape          # cond may hold fewer rows than N (cond_rows): the library works on N = frames
        md = cond.shape[2]
        assert N == self._prep_args.frames
        if tuple(x.shape) != (B, N, md):
            raise ValueError(f"forward: x must be [{B}, {N}, {md}] (the prepared batch / frames / mel), got {tuple(x.shape)}")
        with torch.cuda.device(self.device):
            x = x.to(self.device, torch.float32).contiguous()
            bb = 2 * B if self._prep_args.cfg_strength >= 1e-5 else B
            pred = torch.empty((bb, N, md), device=self.device, dtype=torch.float32)
            s = self._enter(x, pred)
            _lib.check(_lib.lib().lemas_dit_forward(self._h, x.data_ptr(), int(step_index), pred.data_ptr(), s),
                       "lemas_dit_forward")
            self._exit()
        return pred

    def profile_read(self) -> dict:
        names = (C.c_char * 32 * 16)()
        ms = (C.c_double * 16)()
        cnt = (C.c_int64 * 16)()
        n = _lib.lib().lemas_dit_profile_read(self._h, C.cast(names, C.c_void_p), ms, cnt, 16)
        if n < 0:
            _lib.check(n, "lemas_dit_profile_read")
        return {bytes(names[i]).split(b"\0")[0].decode(): (ms[i], cnt[i]) for i in range(n)}


class VocosEngine(_Streamed):
    """``vocoder.decode`` replacement (lemas_tts/infer/utils_infer.py:549)."""

    def __init__(self, state_dict: dict, device="cuda:0", arch: VocosArch = VocosArch()):
        super().__init__(device)
        self.arch = arch
        L = _lib.lib()
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.lemas_vocos_create(arch.input_channels, arch.dim, arch.intermediate_dim, arch.num_layers,
                                            arch.n_fft, arch.hop_length, C.byref(self._h)), "lemas_vocos_create")
            for name, v in state_dict.items():
                _load(L.lemas_vocos_load_weight, self._h, name, v, L.lemas_vocos_load_weight_device)
            _lib.check(L.lemas_vocos_finalize(self._h), "lemas_vocos_finalize")

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().lemas_vocos_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key: str, value: int):
        _lib.check(_lib.lib().lemas_vocos_set_option(self._h, key.encode(), int(value)), f"vocos set_option({key})")

    def decode(self, mel: torch.Tensor, gain: float = 1.0) -> torch.Tensor:
        """mel [B, 100, L] -> wav [B, 256 (L-1)] on the engine's device.  A ``permute(0, 2, 1)`` VIEW of frames-first rows -- how every
        caller of the reference builds the argument (utils_infer.py:546-549, speech_edit_multilingual.py:196-198) -- is decoded in
        place through ``lemas_vocos_decode_rows``: no permute copy."""
        with torch.cuda.device(self.device):
            mel = mel.to(self.device, torch.float32)
            B, Cc, L = mel.shape
            wav = torch.empty((B, self.arch.hop_length * (L - 1)), device=self.device, dtype=torch.float32)
            rows_view = mel.stride(1) == 1 and mel.stride(2) == Cc and (B == 1 or mel.stride(0) >= Cc * L) and not mel.is_contiguous()
            if not rows_view:
                mel = mel.contiguous()
            s = self._enter(mel, wav)
            if rows_view:
                _lib.check(_lib.lib().lemas_vocos_decode_rows(self._h, mel.data_ptr(), B, L, mel.stride(0), float(gain), wav.data_ptr(), s),
                           "lemas_vocos_decode_rows")
            else:
                _lib.check(_lib.lib().lemas_vocos_decode(self._h, mel.data_ptr(), B, L, float(gain), wav.data_ptr(), s),
                           "lemas_vocos_decode")
            self._exit()
        return wav


class MelEngine(_Streamed):
    """``MelSpec.forward`` replacement (lemas_tts/model/modules.py:130-143): wav [B, nw] -> log-mel."""

    def __init__(self, device="cuda:0", n_fft=1024, hop_length=256, n_mel_channels=100, target_sample_rate=24000):
        super().__init__(device)
        self.n_fft, self.hop_length, self.n_mel_channels, self.target_sample_rate = n_fft, hop_length, n_mel_channels, target_sample_rate
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().lemas_mel_create(n_fft, hop_length, n_mel_channels, target_sample_rate, C.byref(self._h)),
                       "lemas_mel_create")

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().lemas_mel_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def frames_first(self, wav: torch.Tensor) -> torch.Tensor:
        """wav [B, nw] -> [B, nw // hop + 1, n_mels] (the layout CFM.sample works in)."""
        with torch.cuda.device(self.device):
            wav = wav.to(self.device, torch.float32).contiguous()
            B, nw = wav.shape
            mel = torch.empty((B, nw // self.hop_length + 1, self.n_mel_channels), device=self.device, dtype=torch.float32)
            s = self._enter(wav, mel)
            _lib.check(_lib.lib().lemas_mel_forward(self._h, wav.data_ptr(), B, nw, mel.data_ptr(), s), "lemas_mel_forward")
            self._exit()
        return mel


class ResampleEngine(_Streamed):
    """``torchaudio.transforms.Resample(orig_freq, new_freq)`` replacement for the prompt (utils_infer.py:494-496):
    wav [B, nw] at ``orig_freq`` -> [B, ceil(new * nw / orig)] at ``new_freq``."""

    def __init__(self, orig_freq: int, new_freq: int = 24000, device="cuda:0"):
        super().__init__(device)
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().lemas_resample_create(self.orig_freq, self.new_freq, C.byref(self._h)), "lemas_resample_create")

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().lemas_resample_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        with torch.cuda.device(self.device):
            wav = wav.to(self.device, torch.float32).contiguous()
            B, nw = wav.shape
            n_out = int(_lib.lib().lemas_resample_out_len(self._h, nw))
            out = torch.empty((B, n_out), device=self.device, dtype=torch.float32)
            s = self._enter(wav, out)
            _lib.check(_lib.lib().lemas_resample_forward(self._h, wav.data_ptr(), B, nw, out.data_ptr(), s), "lemas_resample_forward")
            self._exit()
        return out


class StftEngine(_Streamed):
    """``torch.stft`` / ``torch.istft`` (center=True, onesided) as the UVR5 denoiser uses them (uvr5/multiprocess_cuda_infer.py:206-223).
    ``forward``: wav [B, nw] -> complex spectrogram [B, n_fft // 2 + 1, nw // hop + 1] (torch.stft's layout);
    ``inverse``: that layout -> wav [B, hop * (frames - 1)]."""

    def __init__(self, n_fft: int, hop_length: int, window: torch.Tensor, device="cuda:0"):
        super().__init__(device)
        self.n_fft, self.hop_length = int(n_fft), int(hop_length)
        w = window.detach().to("cpu", torch.float32).contiguous()
        assert w.numel() == self.n_fft
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().lemas_stft_create(self.n_fft, self.hop_length, w.data_ptr(), C.byref(self._h)), "lemas_stft_create")
        self.ld = int(_lib.lib().lemas_stft_ld(self._h))
        self.n_bins = self.n_fft // 2 + 1

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().lemas_stft_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, wav: torch.Tensor) -> torch.Tensor:
        with torch.cuda.device(self.device):
            wav = wav.to(self.device, torch.float32).contiguous()
            B, nw = wav.shape
            F = nw // self.hop_length + 1
            spec = torch.empty((B, F, self.ld), device=self.device, dtype=torch.float32)
            s = self._enter(wav, spec)
            _lib.check(_lib.lib().lemas_stft_forward(self._h, wav.data_ptr(), B, nw, spec.data_ptr(), s), "lemas_stft_forward")
            self._exit()
        nb = self.n_bins
        return torch.complex(spec[:, :, :nb], spec[:, :, nb:2 * nb]).permute(0, 2, 1)      # [B, n_bins, frames]

    def inverse(self, spec: torch.Tensor) -> torch.Tensor:
        with torch.cuda.device(self.device):
            B, nb, F = spec.shape
            assert nb == self.n_bins
            buf = torch.zeros((B, F, self.ld), device=self.device, dtype=torch.float32)
            buf[:, :, :nb] = spec.real.permute(0, 2, 1)
            buf[:, :, nb:2 * nb] = spec.imag.permute(0, 2, 1)
            wav = torch.empty((B, self.hop_length * (F - 1)), device=self.device, dtype=torch.float32)
            s = self._enter(buf, wav)
            _lib.check(_lib.lib().lemas_stft_inverse(self._h, buf.data_ptr(), B, F, wav.data_ptr(), s), "lemas_stft_inverse")
            self._exit()
        return wav


_RESAMPLERS: dict = {}


class MdxEngine(_Streamed):
    """Owns one ``lemas_mdx``: the MDX-Net separation network of the UVR5 prompt denoiser (the reference's onnxruntime session,
    ``uvr5/multiprocess_cuda_infer.py:225-238``; architecture ``uvr5/lib_v5/mdxnet.py:36-127``).

    ``arch``: any object / mapping with the ConvTDFNet constructor arguments ``dim_c, dim_f, dim_t, num_blocks, l, g, k, bn, bias,
    optimizer`` (``bn`` None = no TDF branch; ``optimizer`` 'rmsprop' -> BatchNorm2d, 'adamw' -> GroupNorm(2, c)).
    ``state_dict``: the module's state dict (key names of the reference class; strict)."""

    def __init__(self, arch, state_dict: dict, device="cuda:0", bf16x3: bool = False):
        """``bf16x3``: the 3x3 convolutions on split-bf16 operands (three bf16 MFMAs per product, ~2^-16 relative precision, 5.3x the f32
        matrix rate) instead of the exact f32-input MFMA."""
        super().__init__(device)
        get = (lambda k: arch[k]) if isinstance(arch, dict) else (lambda k: getattr(arch, k))
        opt = get("optimizer")
        if opt not in ("rmsprop", "adamw"):
            raise ValueError(f"optimizer {opt!r}: the reference defines a norm only for 'rmsprop' and 'adamw' (mdxnet.py:51-55)")
        bn = get("bn")
        self.dim_c, self.dim_f, self.dim_t = int(get("dim_c")), int(get("dim_f")), int(get("dim_t"))
        cfg = _lib.MdxConfig(self.dim_c, self.dim_f, self.dim_t, int(get("num_blocks")), int(get("l")), int(get("g")), int(get("k")),
                             -1 if bn is None else int(bn), int(bool(get("bias"))), 1 if opt == "adamw" else 0)
        L = _lib.lib()
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.lemas_mdx_create(C.byref(cfg), C.byref(self._h)), "lemas_mdx_create")
            if bf16x3:
                _lib.check(L.lemas_mdx_set_option(self._h, b"bf16x3", 1), "lemas_mdx_set_option")
            for name, v in state_dict.items():
                _load(L.lemas_mdx_load_weight, self._h, name, v)
            _lib.check(L.lemas_mdx_finalize(self._h), "lemas_mdx_finalize")
        self._taps = {}

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().lemas_mdx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def flops(self, batch: int = 1) -> int:
        return int(_lib.lib().lemas_mdx_flops(self._h, batch))

    def tap(self, name: str, buf: Optional[torch.Tensor]) -> None:
        """Verification hook: copy the activation after stage ``name`` into ``buf`` on every following forward (None removes it)."""
        self._taps[name] = buf
        _lib.check(_lib.lib().lemas_mdx_tap(self._h, name.encode(), C.c_void_p(buf.data_ptr()) if buf is not None else None), "lemas_mdx_tap")

    def forward(self, spek: torch.Tensor) -> torch.Tensor:
        """[b, dim_c, dim_f, dim_t] fp32 -> the same shape (ConvTDFNet.forward, mdxnet.py:103-127)."""
        if spek.ndim != 4 or tuple(spek.shape[1:]) != (self.dim_c, self.dim_f, self.dim_t):
            raise ValueError(f"expected [b, {self.dim_c}, {self.dim_f}, {self.dim_t}], got {tuple(spek.shape)}")
        with torch.cuda.device(self.device):
            x = spek.to(self.device, torch.float32).contiguous()
            out = torch.empty_like(x)
            s = self._enter(x, out, *[t for t in self._taps.values() if t is not None])
            _lib.check(_lib.lib().lemas_mdx_forward(self._h, x.data_ptr(), x.shape[0], out.data_ptr(), s), "lemas_mdx_forward")
            self._exit()
        return out

    __call__ = forward


def resampler(orig_freq: int, new_freq: int = 24000, device="cuda:0") -> "ResampleEngine":
    """One ResampleEngine per (orig, new, device): building one costs a fp64 kernel bank on the host, a hipMalloc and an
    H2D copy, which does not belong on the per-utterance latency path."""
    key = (int(orig_freq), int(new_freq), str(torch.device(device)))
    eng = _RESAMPLERS.get(key)
    if eng is None:
        eng = _RESAMPLERS[key] = ResampleEngine(orig_freq, new_freq, device=device)
    return eng


class ProsodyEngine(_Streamed):
    """Owns one ``lemas_prosody`` (ECAPA-TDNN weights + workspaces + the kaldi-fbank constants) on one device."""

    def __init__(self, arch, state_dict: dict, device="cuda:0"):
        super().__init__(device)
        self.arch = arch
        L = _lib.lib()
        n = len(arch.channels)
        pad = lambda t: (C.c_int32 * 8)(*(list(t) + [0] * (8 - n)))
        cfg = _lib.ProsodyConfig(n, pad(arch.channels), pad(arch.kernel_sizes), pad(arch.dilations), pad(arch.groups),
                                 arch.attention_channels, arch.res2net_scale, arch.se_channels, int(arch.global_context),
                                 arch.embed_dim, arch.input_dim)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.lemas_prosody_create(C.byref(cfg), C.byref(self._h)), "lemas_prosody_create")
            for name, v in state_dict.items():
                _load(L.lemas_prosody_load_weight, self._h, name, v)
            _lib.check(L.lemas_prosody_finalize(self._h), "lemas_prosody_finalize")

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().lemas_prosody_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fbank(self, wav16k: torch.Tensor) -> torch.Tensor:
        """16 kHz audio [n] (n >= 400) -> kaldi fbank [frames, 80]"""
        with torch.cuda.device(self.device):
            w = wav16k.reshape(-1).to(self.device, torch.float32).contiguous()
            frames = int(_lib.lib().lemas_prosody_fbank_frames(w.numel()))
            out = torch.empty((max(frames, 0), 80), device=self.device, dtype=torch.float32)
            s = self._enter(w, out)
            _lib.check(_lib.lib().lemas_prosody_fbank(self._h, w.data_ptr(), w.numel(), out.data_ptr(), s), "lemas_prosody_fbank")
            self._exit()
        return out

    def encode(self, fbank: torch.Tensor) -> torch.Tensor:
        """fbank [T, input_dim] -> L2-normalised embedding [embed_dim] (one sample, no padding mask: cfm.py:259)"""
        with torch.cuda.device(self.device):
            f = fbank.to(self.device, torch.float32).contiguous()
            emb = torch.empty((self.arch.embed_dim,), device=self.device, dtype=torch.float32)
            s = self._enter(f, emb)
            _lib.check(_lib.lib().lemas_prosody_encode(self._h, f.data_ptr(), f.shape[0], emb.data_ptr(), s), "lemas_prosody_encode")
            self._exit()
        return emb
