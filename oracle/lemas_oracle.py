"""CPU oracle: fp32 PyTorch restatement of the LEMAS-TTS acoustic-generation hot path.

TEST INFRASTRUCTURE -- NOT THE PRODUCT.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the checker / the timed CPU
baseline.  The product path (``lemas_tts_amd``) never routes through it and has no CPU fallback.

What is restated (every function cites the reference file:line it follows, paths relative to
/root/reference):  CFM.sample (cfm.py:206-473), DiT.forward (backbones/dit.py:194-254) and the
building blocks in model/modules.py, plus the three third-party pieces the reference calls:
fixed-grid Euler (torchdiffeq), rotary embedding (x_transformers) and Vocos decode (vocos).

Pinning status
  * In-tree arithmetic (CFM.sample, DiT, all modules.py blocks): PINNED -- ``oracle/gen_golden.py``
    runs the real reference here and ``tests/test_oracle_golden.py`` checks this file against the
    committed vectors in ``tests/golden``.
  * torchdiffeq Euler and x_transformers RoPE: the golden vectors were produced with arithmetic
    stand-ins written from the published algorithms (``oracle/ref_shims.py``) because those packages
    are not installed and not vendored: PARITY UNPINNED at those two boundaries.
  * Vocos decode (pip ``vocos``, unpinned in requirements.txt:179; weights charactr/vocos-mel-24khz):
    package absent, no reference tests or vectors exist: PARITY UNPINNED.  This restatement follows
    the published VocosBackbone / ConvNeXtBlock / ISTFTHead(padding="center") definitions and is
    cross-checked against ``torch.istft`` in tests.
  * torchaudio (==2.3.1, requirements.txt:166; absent): the wav -> log-mel front edge (MelSpectrogram) and the prompt
    resampler (``resample_sinc_hann``) are restated from the published algorithms and cross-checked against explicit
    DFT / analytic properties in tests: PARITY UNPINNED.
  * The fp8 GEMM variant (``OracleDiT(fp8=True)``, oracle/mxfp8.py) is not a reference feature: the oracle defines it.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# helpers (model/utils.py)
# ----------------------------------------------------------------------------------------------
def lens_to_mask(lens: Tensor, length: Optional[int] = None) -> Tensor:
    """model/utils.py:42-47."""
    if length is None:
        length = int(lens.amax())
    return torch.arange(length)[None, :] < lens[:, None]


def tokens_to_idx(tokens: list, vocab: dict, pad: int = -1) -> Tensor:
    """model/utils.py:87-94 -- unknown token -> 0, right-pad with -1."""
    rows = [[vocab.get(c, 0) for c in t] for t in tokens]
    n = max(len(r) for r in rows)
    return torch.tensor([r + [pad] * (n - len(r)) for r in rows], dtype=torch.long)


def sway_max(steps: int, min_ratio: float = 1e-9, safety: float = 0.7, t_start: float = 0.0) -> float:
    """cfm.py:343-373 as called at :447 (t_start=0 except in the duplicate_test corner)."""
    dt = (1.0 - t_start) / max(1, steps)
    p_max = 11.0 if dt >= 0.9 else math.log(min_ratio) / math.log(dt)
    return max(0.0, p_max - 1.0) * safety


def time_grid(steps: int, sway_sampling_coef: Optional[float], t_start: float = 0.0) -> Tensor:
    """cfm.py:445-453: power-law warp t**(1+s), s capped by ``sway_max`` (fp32 throughout)."""
    t = torch.linspace(t_start, 1, steps + 1, dtype=torch.float32)
    smax = torch.tensor(sway_max(steps, t_start=t_start), dtype=torch.float32)
    # the reference takes python ``min(tensor, number)``: the exponent is either the fp32 tensor cap or
    # the caller's python number (then ``t ** float``) -- kept distinct so the grid matches bit-for-bit
    s = smax if sway_sampling_coef is None else min(smax, sway_sampling_coef)
    return t ** (1 + s)


def clip_and_shuffle(mel: Tensor, mel_len: int, ratio=None, sample_rate: int = 24000, hop_length: int = 256) -> Tensor:
    """cfm.py:39-84 (accent-GRL conditioning segment); consumes Python's ``random`` exactly as the reference does:
    [randint crop length unless ratio], randint crop start, shuffle(chunks), choice(chunks) per top-up chunk."""
    import random as _random
    fps = int(sample_rate / hop_length)
    total = mel_len
    seg_len = int(total * ratio) if ratio else _random.randint(int(0.25 * total), int(0.75 * total))
    start = _random.randint(0, max(0, total - seg_len))
    seg = mel[:, start:start + seg_len]
    chunks = [seg[:, i * fps:(i + 1) * fps] for i in range((seg.size(1) + fps - 1) // fps)]
    _random.shuffle(chunks)
    out = torch.cat(chunks, dim=1)
    if out.size(1) < total:
        rep = []
        while sum(c.size(1) for c in rep) < total:
            rep.append(_random.choice(chunks))
        out = torch.cat([out] + rep, dim=1)
    return out[:, :total]


def build_edit_mask(n_samples: int, spans, sr: int = 24000, hop: int = 256) -> Tensor:
    """scripts/speech_edit_multilingual.py:125-158 -- True = keep original frame."""
    offset = 0.0
    parts = []
    for (start, end) in spans:
        start = max(start - 0.1, 0.0)
        end = min(end + 0.1, n_samples / sr)
        dur_samples = int(round((end - start) * sr))
        start_samples = int(round(start * sr))
        keep = int(round((start_samples - offset) / hop))
        edit = int(round(dur_samples / hop))
        if keep > 0:
            parts.append(torch.ones(keep, dtype=torch.bool))
        if edit > 0:
            parts.append(torch.zeros(edit, dtype=torch.bool))
        offset = end * sr
    m = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.bool)
    total = n_samples // hop
    if m.shape[0] < total + 1:
        m = F.pad(m, (0, total + 1 - m.shape[0]), value=True)
    return m[None, :]


# ----------------------------------------------------------------------------------------------
# third-party pieces (published algorithms; see module docstring for pin status)
# ----------------------------------------------------------------------------------------------
def rope_freqs(n: int, dim_head: int, inv_freq: Optional[Tensor] = None) -> Tensor:
    """x_transformers RotaryEmbedding.forward_from_seq_len (call site dit.py:236): [n, dim_head]
    with every frequency duplicated into adjacent slots."""
    if inv_freq is None:
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim_head, 2).float() / dim_head))
    ang = torch.arange(n).float()[:, None] * inv_freq[None, :]
    return ang.repeat_interleave(2, dim=-1)


def rope_apply(x: Tensor, freqs: Tensor) -> Tensor:
    """x_transformers apply_rotary_pos_emb with scale 1 (call sites modules.py:479-480);
    rotation acts on adjacent pairs (x0,x1) -> (x0 c - x1 s, x1 c + x0 s)."""
    x0, x1 = x[..., 0::2], x[..., 1::2]
    rot = torch.stack((-x1, x0), dim=-1).flatten(-2)
    return x * freqs.cos() + rot * freqs.sin()


def euler_solve(fn, y0: Tensor, t: Tensor) -> Tensor:
    """torchdiffeq odeint(method='euler') on the caller's grid (call site cfm.py:456).  torchdiffeq refuses a grid that is
    not strictly monotone -- which is what `sway_sampling_coef = -1` (the default of infer_batch_process, utils_infer.py:476)
    produces with this sampler's power warp: t ** (1 + min(sway_max, -1)) = t ** 0 = 1 everywhere."""
    assert bool((t[1:] > t[:-1]).all()), "t must be strictly increasing or decreasing"
    ys = [y0]
    y = y0
    for k in range(t.shape[0] - 1):
        y = y + (t[k + 1] - t[k]) * fn(t[k], y)
        ys.append(y)
    return torch.stack(ys)


# ----------------------------------------------------------------------------------------------
# DiT (backbones/dit.py + model/modules.py)
# ----------------------------------------------------------------------------------------------
class OracleDiT:
    def __init__(self, sd: dict, arch, prefix: str = "transformer.", fp8: bool = False, fp8_qk: Optional[bool] = None):
        """``fp8=True`` emulates the build's fp8 GEMM variant (BASELINE config 5; not a reference feature): the q/k/v/out
        and ff linears of every DiTBlock see MXFP8-quantised inputs and per-channel e4m3 weights (oracle/mxfp8.py).
        ``fp8_qk`` (default: follows ``fp8``; engine option attn_f8qk): the rotated q (already multiplied by
        softmax_scale * log2(e)) and k are rounded to bf16 and quantised to MXFP8, one scale per 32-wide half of a head,
        before Q K^T -- what csrc/attention.hip's fp8 QK^T path consumes (round 6)."""
        self.fp8 = fp8
        self.fp8_qk = fp8 if fp8_qk is None else fp8_qk
        self._w8 = {}
        self.a = arch
        self.p = {k[len(prefix):]: torch.as_tensor(v, dtype=torch.float32) for k, v in sd.items()
                  if k.startswith(prefix)}
        self.has_prosody = "prosody_text_proj.weight" in self.p
        self._text_cache = {}

    def lin(self, name: str, x: Tensor) -> Tensor:
        if self.fp8 and name.startswith("transformer_blocks.") and "attn_norm" not in name:
            from .mxfp8 import mx_quant, w_quant
            if name not in self._w8:
                self._w8[name] = w_quant(self.p[name + ".weight"])[2]
            xq = mx_quant(x.reshape(-1, x.shape[-1]))[2].reshape(x.shape)
            return F.linear(xq, self._w8[name], self.p[name + ".bias"])
        return F.linear(x, self.p[name + ".weight"], self.p[name + ".bias"])

    # modules.py:149-161, 721-731
    def time_embed(self, t: Tensor) -> Tensor:
        half = self.a.time_freq_dim // 2
        e = torch.exp(torch.arange(half).float() * -(math.log(10000) / (half - 1)))
        e = 1000 * t[:, None] * e[None, :]
        h = torch.cat((e.sin(), e.cos()), dim=-1)
        return self.lin("time_embed.time_mlp.2", F.silu(self.lin("time_embed.time_mlp.0", h)))

    # modules.py:241-269 (GRN :225-234)
    def _convnext_v2(self, i: int, x: Tensor) -> Tensor:
        p, q = self.p, f"text_embed.text_blocks.{i}."
        h = F.conv1d(x.transpose(1, 2), p[q + "dwconv.weight"], p[q + "dwconv.bias"], padding=3,
                     groups=x.shape[-1]).transpose(1, 2)
        h = F.layer_norm(h, (h.shape[-1],), p[q + "norm.weight"], p[q + "norm.bias"], eps=1e-6)
        h = F.gelu(self.lin(q + "pwconv1", h))
        g = torch.norm(h, p=2, dim=1, keepdim=True)
        h = p[q + "grn.gamma"] * (h * (g / (g.mean(dim=-1, keepdim=True) + 1e-6))) + p[q + "grn.beta"] + h
        return x + self.lin(q + "pwconv2", h)

    # dit.py:51-81 (freqs table modules.py:196-207, indices :210-219)
    def text_embed(self, text: Tensor, n: int, drop_text: bool) -> Tensor:
        td = self.a.text_dim
        tok = (text + 1)[:, :n]
        tok = F.pad(tok, (0, n - tok.shape[1]), value=0)
        pad_mask = tok == 0                       # computed BEFORE the cfg drop (dit.py:56-60)
        if drop_text:
            tok = torch.zeros_like(tok)
        h = F.embedding(tok, self.p["text_embed.text_embed.weight"])
        if self.a.conv_layers > 0:
            inv = 1.0 / (10000.0 ** (torch.arange(0, td, 2)[: td // 2].float() / td))
            pos = torch.arange(n).clamp(max=4095).float()
            ang = torch.outer(pos, inv)
            h = h + torch.cat([ang.cos(), ang.sin()], dim=-1)[None]
            h = h.masked_fill(pad_mask[..., None], 0.0)
            for i in range(self.a.conv_layers):
                h = self._convnext_v2(i, h).masked_fill(pad_mask[..., None], 0.0)
        return h

    # modules.py:167-190 (no mask is passed from dit.py:98)
    def conv_pos(self, x: Tensor) -> Tensor:
        p, g, k = self.p, self.a.conv_pos_groups, self.a.conv_pos_kernel
        h = x.transpose(1, 2)
        for j in (0, 2):
            q = f"input_embed.conv_pos_embed.conv1d.{j}."
            h = F.mish(F.conv1d(h, p[q + "weight"], p[q + "bias"], padding=k // 2, groups=g))
        return h.transpose(1, 2)

    # dit.py:93-99
    def input_embed(self, x: Tensor, cond: Tensor, text_embed: Tensor, drop_audio_cond: bool) -> Tensor:
        if drop_audio_cond:
            cond = torch.zeros_like(cond)
        h = self.lin("input_embed.proj", torch.cat((x, cond, text_embed), dim=-1))
        return self.conv_pos(h) + h

    # modules.py:442-503
    def attention(self, i: int, h: Tensor, mask: Optional[Tensor], freqs: Tensor) -> Tensor:
        b, n, _ = h.shape
        H, Dh = self.a.heads, self.a.dim_head
        q = f"transformer_blocks.{i}.attn."
        qh = self.lin(q + "to_q", h).view(b, n, H, Dh).transpose(1, 2)
        kh = self.lin(q + "to_k", h).view(b, n, H, Dh).transpose(1, 2)
        vh = self.lin(q + "to_v", h).view(b, n, H, Dh).transpose(1, 2)
        qh, kh = rope_apply(qh, freqs), rope_apply(kh, freqs)
        if self.fp8_qk:
            from .mxfp8 import mx_quant
            c = float(np.float32(1.0 / math.sqrt(Dh)) * np.float32(1.4426950408889634))      # the QK epilogue's q_scale (fp32)

            def mx(x):
                xb = x.to(torch.bfloat16).float()
                return mx_quant(xb.reshape(-1, Dh))[2].reshape(x.shape)
            s = (mx(qh * c) @ mx(kh).transpose(-1, -2)) * math.log(2.0)
        else:
            s = (qh @ kh.transpose(-1, -2)) * (1.0 / math.sqrt(Dh))
        if mask is not None:
            s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
        o = (torch.softmax(s, dim=-1) @ vh).transpose(1, 2).reshape(b, n, H * Dh)
        o = self.lin(q + "to_out.0", o)
        if mask is not None:
            o = o.masked_fill(~mask[..., None], 0.0)
        return o

    # modules.py:627-641 with AdaLayerNorm :310-315
    def block(self, i: int, x: Tensor, t: Tensor, mask: Optional[Tensor], freqs: Tensor) -> Tensor:
        d = self.a.dim
        q = f"transformer_blocks.{i}."
        emb = self.lin(q + "attn_norm.linear", F.silu(t))
        shift_a, scale_a, gate_a, shift_m, scale_m, gate_m = emb.chunk(6, dim=1)
        h = F.layer_norm(x, (d,), eps=1e-6) * (1 + scale_a[:, None]) + shift_a[:, None]
        x = x + gate_a[:, None] * self.attention(i, h, mask, freqs)
        h = F.layer_norm(x, (d,), eps=1e-6) * (1 + scale_m[:, None]) + shift_m[:, None]
        h = self.lin(q + "ff.ff.2", F.gelu(self.lin(q + "ff.ff.0.0", h), approximate="tanh"))
        return x + gate_m[:, None] * h

    # dit.py:194-254
    def forward(self, x, cond, text, time, drop_audio_cond, drop_text, mask=None, cache=False,
                prosody_text=None) -> Tensor:
        b, n, _ = x.shape
        if time.ndim == 0:
            time = time.repeat(b)
        t = self.time_embed(time)
        key = bool(drop_text)
        if cache and key in self._text_cache:
            te = self._text_cache[key]
        else:
            te = self.text_embed(text, n, drop_text)
            if cache:
                self._text_cache[key] = te
        if prosody_text is not None and self.has_prosody:
            pt = self.lin("prosody_text_proj", prosody_text)
            pt = F.pad(pt, (0, 0, 0, n - pt.shape[1])) if pt.shape[1] < n else pt[:, :n]
            te = te + pt
        h = self.input_embed(x, cond, te, drop_audio_cond)
        freqs = rope_freqs(n, self.a.dim_head, self.p.get("rotary_embed.inv_freq"))
        for i in range(self.a.depth):
            h = self.block(i, h, t, mask, freqs)
        emb = self.lin("norm_out.linear", F.silu(t))       # modules.py:331-336: order scale, shift
        scale, shift = emb.chunk(2, dim=1)
        h = F.layer_norm(h, (self.a.dim,), eps=1e-6) * (1 + scale)[:, None] + shift[:, None]
        return self.lin("proj_out", h)

    def clear_cache(self):
        self._text_cache = {}


# ----------------------------------------------------------------------------------------------
# CFM.sample (cfm.py:206-473) -- inference sampler only
# ----------------------------------------------------------------------------------------------
class OracleCFM:
    def __init__(self, sd: dict, arch, fp8: bool = False, fp8_qk: Optional[bool] = None):
        self.dit = OracleDiT(sd, arch, fp8=fp8, fp8_qk=fp8_qk)
        self.a = arch
        self.prosody_to_mel = None
        if "prosody_to_mel.weight" in sd:
            self.prosody_to_mel = (torch.as_tensor(sd["prosody_to_mel.weight"]).float(),
                                   torch.as_tensor(sd["prosody_to_mel.bias"]).float())

    @torch.no_grad()
    def sample(self, cond: Tensor, text: Tensor, duration, *, y0: Tensor, lens: Optional[Tensor] = None,
               steps: int = 32, cfg_strength: float = 1.0, sway_sampling_coef: Optional[float] = None,
               max_duration: int = 4096, edit_mask: Optional[Tensor] = None,
               prosody_embeds: Optional[Tensor] = None, t_grid: Optional[Tensor] = None,
               no_ref_audio: bool = False, cond_noise: Optional[Tensor] = None,
               use_acc_grl: bool = False, ref_ratio: float = 1, duplicate_test: bool = False, t_inter: float = 0.1):
        """``cond`` is a mel [B,F,100]; ``text`` int64 [B,Nt] padded with -1; ``y0`` [B,N,100] is the
        explicit ODE start (the reference draws it at cfm.py:430-435).  ``prosody_embeds`` [B,512]
        stands for the prosody-encoder output (cfm.py:248-265, a "next" row).  ``no_ref_audio`` (cfm.py:320-324, 464-466)
        replaces the conditioning by noise around the prompt's mean; ``cond_noise`` [B,N,100] is the ``randn_like`` draw."""
        cond = cond.float()
        b, f = cond.shape[:2]
        cond_mean = cond.mean(dim=1, keepdim=True)                         # :239
        cond_grl = None
        if use_acc_grl:                                                    # :266-283 (grad_reverse is the identity forward)
            cond_grl = cond if ref_ratio >= 1 else clip_and_shuffle(cond[0].T, f, ratio=ref_ratio).T[None]
        if lens is None:
            lens = torch.full((b,), f, dtype=torch.long)
        cond_mask = lens_to_mask(lens)                                     # :293
        if edit_mask is not None:
            cond_mask = cond_mask & edit_mask                              # :294-295
        if isinstance(duration, int):
            duration = torch.full((b,), duration, dtype=torch.long)
        duration = torch.maximum(torch.maximum((text != -1).sum(-1), lens) + 1, duration)   # :300-302
        duration = duration.clamp(max=max_duration)
        n = int(duration.amax())
        test_cond = F.pad(cond, (0, 0, f, n - 2 * f)) if duplicate_test else None   # :307-309
        cond = F.pad(cond, (0, 0, 0, n - f))                               # :311
        prosody_text = None
        if prosody_embeds is not None and self.prosody_to_mel is not None:
            pm = F.pad(prosody_embeds[:, None, :].expand(-1, f, -1), (0, 0, 0, n - f))     # :265,:314-316
            cond = cond + F.linear(pm, *self.prosody_to_mel)               # :317-318
            prosody_text = prosody_embeds[:, None, :].expand(-1, text.shape[1], -1)        # :376-378
        if no_ref_audio:                                                   # :320-324
            rc = cond_noise.float() * 0.1 + cond_mean
            cond = rc / rc.mean(dim=1, keepdim=True) * cond_mean
        cond_mask = F.pad(cond_mask, (0, n - cond_mask.shape[-1]), value=False)[..., None]  # :326-327
        step_cond = torch.where(cond_mask, cond, torch.zeros_like(cond))   # :388-390
        if cond_grl is not None:                                           # :329-330, :387-388
            step_cond = torch.where(cond_mask, F.pad(cond_grl, (0, 0, 0, n - f)), torch.zeros_like(cond))
        mask = lens_to_mask(duration) if b > 1 else None                   # :336-339

        def fn(t, x):                                                      # :382-425
            pred = self.dit.forward(x, step_cond, text, t, False, False, mask, True, prosody_text)
            if cfg_strength < 1e-5:
                return pred
            null = self.dit.forward(x, step_cond, text, t, True, True, mask, True, prosody_text)
            return (pred + (pred - null) * (cfg_strength * (1 - t) ** 2)).clamp(-20, 20)

        assert y0.shape == (b, n, self.a.mel_dim), (y0.shape, (b, n))
        y0 = y0.float()
        t_start = 0.0
        if duplicate_test:                                                 # :438-443
            t_start = t_inter
            y0 = (1 - t_start) * y0 + t_start * test_cond
            steps = int(steps * (1 - t_start))
        t = time_grid(steps, sway_sampling_coef, t_start) if t_grid is None else t_grid
        traj = euler_solve(fn, y0, t)                                      # :456
        self.dit.clear_cache()                                             # :457
        out = torch.where(cond_mask, cond, traj[-1])                       # :459-461
        if no_ref_audio:                                                   # :464-466
            out = out.clone()
            out[:, f:, :] = out[:, f:, :] - (out[:, f:, :].mean(dim=1, keepdim=True) - cond_mean)
        return out, traj


# ----------------------------------------------------------------------------------------------
# wav -> log-mel front edge (model/modules.py:75-101; torchaudio MelSpectrogram is third-party: PARITY UNPINNED)
# ----------------------------------------------------------------------------------------------
def resample_sinc_hann(wav: Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
                       rolloff: float = 0.99) -> Tensor:
    """Prompt resampling, ``torchaudio.transforms.Resample(sr, 24000)`` at utils_infer.py:494-496 (and cfm.py:254).
    torchaudio is third party, not in the tree and not installed: PARITY UNPINNED.  Restated from the published
    ``torchaudio.functional.resample`` (``sinc_interp_hann``): reduce the rates by their gcd, build one windowed-sinc
    kernel per output phase in float64, zero-pad (width, width + orig), strided correlation, truncate to
    ceil(new * len / orig).  wav [B, len] -> [B, ceil(new * len / orig)]."""
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    if o == n:
        return wav.clone()
    base = min(o, n) * rolloff
    width = int(math.ceil(lowpass_filter_width * o / base))
    idx = torch.arange(-width, width + o, dtype=torch.float64)[None, None] / o
    t = torch.arange(0, -n, -1, dtype=torch.float64)[:, None, None] / n + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / o)
    kernels = kernels.to(torch.float32)                                   # [n, 1, 2 width + o]
    length = wav.shape[-1]
    x = F.pad(wav.float(), (width, width + o))
    y = F.conv1d(x[:, None], kernels, stride=o)                           # [B, n, frames]
    y = y.transpose(1, 2).reshape(wav.shape[0], -1)
    return y[..., : int(math.ceil(n * length / o))]


def htk_filterbank(n_freqs: int = 513, n_mels: int = 100, sample_rate: int = 24000) -> Tensor:
    """torchaudio.functional.melscale_fbanks(n_freqs, 0, sr/2, n_mels, sr, norm=None, mel_scale='htk') -> [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    m_max = 2595.0 * math.log10(1.0 + (sample_rate / 2) / 700.0)
    m_pts = torch.linspace(0.0, m_max, n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0).float()


def vocos_mel_spectrogram(wav: Tensor, n_fft: int = 1024, hop: int = 256, n_mels: int = 100, sample_rate: int = 24000) -> Tensor:
    """get_vocos_mel_spectrogram (modules.py:75-101): wav [B, nw] -> log-mel [B, n_mels, nw // hop + 1]."""
    spec = torch.stft(wav.float(), n_fft, hop_length=hop, win_length=n_fft, window=torch.hann_window(n_fft), center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs()      # power = 1
    mel = torch.matmul(spec.transpose(1, 2), htk_filterbank(n_fft // 2 + 1, n_mels, sample_rate)).transpose(1, 2)
    return mel.clamp(min=1e-5).log()


# ----------------------------------------------------------------------------------------------
# Vocos decode (third-party; call site infer/utils_infer.py:549)
# ----------------------------------------------------------------------------------------------
class OracleVocos:
    """VocosBackbone(100->512, 8 ConvNeXt blocks, intermediate 1536) + ISTFTHead(n_fft 1024, hop 256,
    padding='center') as published in the ``vocos`` package (models.py / modules.py / heads.py /
    spectral_ops.py)."""

    def __init__(self, sd: dict, num_layers: int = 8, n_fft: int = 1024, hop: int = 256):
        self.p = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in sd.items()}
        self.num_layers, self.n_fft, self.hop = num_layers, n_fft, hop

    def backbone(self, mel: Tensor) -> Tensor:
        p = self.p
        h = F.conv1d(mel, p["backbone.embed.weight"], p["backbone.embed.bias"], padding=3)
        d = h.shape[1]
        h = F.layer_norm(h.transpose(1, 2), (d,), p["backbone.norm.weight"], p["backbone.norm.bias"], 1e-6)
        h = h.transpose(1, 2)
        for i in range(self.num_layers):
            q = f"backbone.convnext.{i}."
            r = h
            h = F.conv1d(h, p[q + "dwconv.weight"], p[q + "dwconv.bias"], padding=3, groups=d).transpose(1, 2)
            h = F.layer_norm(h, (d,), p[q + "norm.weight"], p[q + "norm.bias"], 1e-6)
            h = F.linear(F.gelu(F.linear(h, p[q + "pwconv1.weight"], p[q + "pwconv1.bias"])),
                         p[q + "pwconv2.weight"], p[q + "pwconv2.bias"])
            h = r + (p[q + "gamma"] * h).transpose(1, 2)
        return F.layer_norm(h.transpose(1, 2), (d,), p["backbone.final_layer_norm.weight"],
                            p["backbone.final_layer_norm.bias"], 1e-6)          # [B, L, 512]

    def head(self, h: Tensor) -> Tensor:
        p = self.p
        o = F.linear(h, p["head.out.weight"], p["head.out.bias"]).transpose(1, 2)   # [B, 1026, L]
        mag, ph = o.chunk(2, dim=1)
        mag = torch.exp(mag).clip(max=1e2)
        spec = mag * (torch.cos(ph) + 1j * torch.sin(ph))
        return torch.istft(spec, self.n_fft, self.hop, self.n_fft, p["head.istft.window"], center=True)

    @torch.no_grad()
    def decode(self, mel: Tensor) -> Tensor:
        """mel [B,100,L] fp32 -> wav [B, 256*(L-1)]."""
        return self.head(self.backbone(mel.float()))
