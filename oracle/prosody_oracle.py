"""CPU oracle for the prosody-encoder row: fp32 PyTorch restatement of the reference's ECAPA-TDNN
(``lemas_tts/model/backbones/prosody_encoder.py``) and of the feature front end it is fed with at ``cfm.py:248-263``.

TEST INFRASTRUCTURE -- NOT THE PRODUCT (same import rule as lemas_oracle.py).

Pinning status
  * ECAPA-TDNN arithmetic (in-tree reference code): PINNED by ``tests/golden/prosody_enc_*.npz``, produced by running the
    reference class here (``oracle/gen_golden.py``) on synthetic weights.  The ARCHITECTURE numbers come from
    ``pretssel_cfg.json``, which is not in the tree: ``ProsodyArch`` defaults are the published Pretssel values, an assumption.
  * kaldi fbank (``torchaudio.compliance.kaldi.fbank`` with ``num_mel_bins=80, sample_frequency=16000`` and otherwise the
    defaults, prosody_encoder.py:356-360) and the 24 k -> 16 k resample (cfm.py:254): third party, absent: PARITY UNPINNED,
    restated from the published Kaldi feature pipeline (povey window, pre-emphasis 0.97, DC removal, 512-point power
    spectrum, 80 triangular bins on the 1127 ln(1 + f / 700) mel scale between 20 Hz and Nyquist, log with float eps floor).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F
from torch import Tensor

from .lemas_oracle import resample_sinc_hann


# ---- prosody_encoder.py:334-361 extract_fbank_16k (+ kaldi.fbank defaults) ------------------------------------
def kaldi_mel_banks(num_bins: int = 80, padded: int = 512, sample_rate: float = 16000.0, low: float = 20.0, high: float = 0.0) -> Tensor:
    nyq = 0.5 * sample_rate
    if high <= 0:
        high += nyq
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    n_fft_bins = padded // 2
    width = sample_rate / padded
    ml, mh = mel(low), mel(high)
    delta = (mh - ml) / (num_bins + 1)
    fb = torch.zeros(num_bins, n_fft_bins + 1, dtype=torch.float64)       # last column (Nyquist) stays 0, as torchaudio pads it
    for b in range(num_bins):
        left, center, right = ml + b * delta, ml + (b + 1) * delta, ml + (b + 2) * delta
        for i in range(n_fft_bins):
            m = mel(width * i)
            if left < m < right:
                fb[b, i] = (m - left) / (center - left) if m <= center else (right - m) / (right - center)
    return fb.float()


def kaldi_fbank_80(wav16k: Tensor) -> Tensor:
    """wav16k [T] or [1, T] fp32 at 16 kHz -> [frames, 80] log mel energies; short inputs are tiled to >= 400 samples
    first (prosody_encoder.py:349-354)."""
    x = wav16k.reshape(1, -1).float()
    if x.shape[-1] < 400:
        x = x.repeat(1, 400 // x.shape[-1] + 1)
    x = x[0]
    win, shift, padded = 400, 160, 512
    m = 1 + (x.numel() - win) // shift                                   # snip_edges
    frames = x.unfold(0, win, shift)[:m].clone()                          # [m, 400]
    frames = frames - frames.mean(dim=1, keepdim=True)                    # remove_dc_offset
    prev = torch.cat([frames[:, :1], frames[:, :-1]], dim=1)              # replicate-pad on the left
    frames = frames - 0.97 * prev                                         # preemphasis
    n = torch.arange(win, dtype=torch.float64)
    povey = (0.5 - 0.5 * torch.cos(2 * math.pi * n / (win - 1))).pow(0.85).float()
    frames = F.pad(frames * povey, (0, padded - win))
    power = torch.fft.rfft(frames, dim=1).abs().pow(2.0)                  # [m, 257]
    mel = power @ kaldi_mel_banks().T
    return torch.clamp(mel, min=torch.finfo(torch.float32).eps).log()


def prosody_features_from_24k(wav24k: Tensor) -> Tensor:
    """cfm.py:250-260: 24 kHz prompt [nw] -> 16 kHz -> fbank [frames, 80]"""
    a16 = resample_sinc_hann(wav24k.reshape(1, -1), 24000, 16000)[0]
    return kaldi_fbank_80(a16)


# ---- prosody_encoder.py:28-331 ECAPA_TDNN ------------------------------------------------------------------------
class OracleECAPA:
    def __init__(self, sd: dict, arch):
        self.p = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in sd.items()}
        self.a = arch

    def tdnn(self, q: str, x: Tensor, k: int, dil: int) -> Tensor:       # :136-161  conv -> ReLU -> LayerNorm(eps 1e-12) over channels
        y = F.relu(F.conv1d(x, self.p[q + "conv.weight"], self.p[q + "conv.bias"], dilation=dil, padding=dil * (k - 1) // 2))
        c = y.shape[1]
        return F.layer_norm(y.transpose(1, 2), (c,), self.p[q + "norm.weight"], self.p[q + "norm.bias"], eps=1e-12).transpose(1, 2)

    def res2net(self, q: str, x: Tensor, k: int, dil: int) -> Tensor:    # :164-202
        ys, prev = [], None
        for i, xi in enumerate(torch.chunk(x, self.a.res2net_scale, dim=1)):
            if i == 0:
                yi = xi
            else:
                yi = self.tdnn(f"{q}blocks.{i - 1}.", xi if i == 1 else xi + prev, k, dil)
            prev = yi
            ys.append(yi)
        return torch.cat(ys, dim=1)

    def se(self, q: str, x: Tensor) -> Tensor:                           # :205-230 (padding_mask=None branch)
        s = x.mean(dim=2, keepdim=True)
        s = F.relu(F.conv1d(s, self.p[q + "conv1.weight"], self.p[q + "conv1.bias"]))
        s = torch.sigmoid(F.conv1d(s, self.p[q + "conv2.weight"], self.p[q + "conv2.bias"]))
        return s * x

    def se_res2net(self, q: str, x: Tensor, k: int, dil: int) -> Tensor:  # :283-331
        res = x
        if q + "shortcut.weight" in self.p:
            res = F.conv1d(x, self.p[q + "shortcut.weight"], self.p[q + "shortcut.bias"])
        y = self.tdnn(q + "tdnn1.", x, 1, 1)
        y = self.res2net(q + "res2net_block.", y, k, dil)
        y = self.tdnn(q + "tdnn2.", y, 1, 1)
        return self.se(q + "se_block.", y) + res

    def asp(self, x: Tensor) -> Tensor:                                   # :233-280 (no mask)
        n, c, L = x.shape
        eps = 1e-12

        def stats(w):
            mean = (w * x).sum(2)
            std = torch.sqrt((w * (x - mean.unsqueeze(2)).pow(2)).sum(2).clamp(eps))
            return mean, std

        if self.a.global_context:
            mean, std = stats(torch.full((n, 1, L), 1.0 / L))
            attn = torch.cat([x, mean.unsqueeze(2).repeat(1, 1, L), std.unsqueeze(2).repeat(1, 1, L)], dim=1)
        else:
            attn = x
        attn = F.conv1d(torch.tanh(self.tdnn("asp.tdnn.", attn, 1, 1)), self.p["asp.conv.weight"], self.p["asp.conv.bias"])
        attn = F.softmax(attn, dim=2)
        mean, std = stats(attn)
        return torch.cat((mean, std), dim=1).unsqueeze(2)

    @torch.no_grad()
    def forward(self, fbank: Tensor) -> Tensor:                           # :103-133; fbank [B, T, 80] -> [B, embed_dim]
        a = self.a
        x = fbank.float().transpose(1, 2)
        x = self.tdnn("blocks.0.", x, a.kernel_sizes[0], a.dilations[0])
        outs = []
        for i in range(1, len(a.channels) - 1):
            x = self.se_res2net(f"blocks.{i}.", x, a.kernel_sizes[i], a.dilations[i])
            outs.append(x)
        x = self.tdnn("mfa.", torch.cat(outs, dim=1), a.kernel_sizes[-1], a.dilations[-1])
        x = self.asp(x)
        c2 = x.shape[1]
        x = F.layer_norm(x.transpose(1, 2), (c2,), self.p["asp_norm.weight"], self.p["asp_norm.bias"], eps=1e-12).transpose(1, 2)
        x = F.conv1d(x, self.p["fc.weight"], self.p["fc.bias"])
        return F.normalize(x.transpose(1, 2).squeeze(1), dim=-1)
