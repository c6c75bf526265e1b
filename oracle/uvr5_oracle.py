"""CPU restatement of the reference's UVR5 MDX-Net denoising SHELL -- TEST INFRASTRUCTURE, never imported by the product
(``lemas_tts_amd/``); only ``tests/`` may use it.

Restates ``uvr5/multiprocess_cuda_infer.py:181-301`` (class ``Inference``) without the ONNX network, which is a callable here:
  :206-212 stft           torch.stft(n_fft, hop 1024, hann(n_fft, periodic=False), center=True) -> [b, 4, dim_f, dim_t]
  :214-223 istft          zero bins above dim_f, torch.istft(center=True) -> [b, 2, chunk]
  :243-258 initialize_mix trim zeros | mix | pad to whole gen_size pieces (+ a full piece when already whole) | trim zeros
  :261-273 run_model      three lowest bins dropped, network (or +-input average), inverse, centre trim
  :276-301 demix_base     per slice: chunks in batches, drop the padding, margins between slices; the LAST slice's result is returned
Pinned by ``tests/golden/uvr5_shell.npz``, which ``oracle/gen_golden_uvr5.py`` produced by running the reference class itself.
"""
from __future__ import annotations

import numpy as np
import torch

HOP = 1024


class ShellOracle:
    def __init__(self, n_fft: int, dim_f: int, dim_t_set: int, is_denoise: bool = False, mdx_batch_size: int = 1, margin: int = 44100):
        self.n_fft, self.dim_f, self.dim_t = n_fft, dim_f, 2 ** dim_t_set
        self.is_denoise, self.mdx_batch_size, self.margin = is_denoise, mdx_batch_size, margin
        self.n_bins, self.trim = n_fft // 2 + 1, n_fft // 2
        self.chunk_size = HOP * (self.dim_t - 1)
        self.gen_size = self.chunk_size - 2 * self.trim
        self.window = torch.hann_window(n_fft, periodic=False)
        self.model_run = None

    def stft(self, x):
        x = x.reshape(-1, self.chunk_size)
        c = torch.stft(x, n_fft=self.n_fft, hop_length=HOP, window=self.window, center=True, return_complex=True)   # [b2, bins, t]
        planes = torch.stack((c.real, c.imag), dim=1)
        return planes.reshape(-1, 4, self.n_bins, self.dim_t)[:, :, :self.dim_f]

    def istft(self, x):
        b = x.shape[0]
        full = torch.cat((x, torch.zeros(b, 4, self.n_bins - self.dim_f, self.dim_t)), dim=-2).reshape(b * 2, 2, self.n_bins, self.dim_t)
        c = torch.complex(full[:, 0], full[:, 1])
        return torch.istft(c, n_fft=self.n_fft, hop_length=HOP, window=self.window, center=True).reshape(b, 2, self.chunk_size)

    def initialize_mix(self, mix):
        n = mix.shape[1]
        pad = self.gen_size - n % self.gen_size
        framed = torch.cat((torch.zeros(2, self.trim), mix, torch.zeros(2, pad), torch.zeros(2, self.trim)), dim=1)
        waves = [framed[:, i:i + self.chunk_size] for i in range(0, n + pad, self.gen_size)]
        return torch.stack(waves), pad

    def run_model(self, mix, is_match_mix=False):
        spek = self.stft(mix)
        spek[:, :, :3, :] *= 0
        if is_match_mix:
            pred = spek
        elif self.is_denoise:
            pred = torch.as_tensor(-np.asarray(self.model_run(-spek)) * 0.5 + np.asarray(self.model_run(spek)) * 0.5)
        else:
            pred = torch.as_tensor(np.asarray(self.model_run(spek)))
        return self.istft(pred)[:, :, self.trim:-self.trim].transpose(0, 1).reshape(2, -1)

    def demix_base(self, mix: dict, is_match_mix=False):
        keys, result = list(mix.keys()), None
        for key in keys:
            waves, pad = self.initialize_mix(mix[key])
            parts = [self.run_model(w, is_match_mix) for w in waves.split(self.mdx_batch_size)]
            tar = torch.cat(parts, dim=-1)[:, :-pad]
            start = 0 if key == 0 else self.margin
            end = None if key == keys[-1] or self.margin == 0 else -self.margin
            result = tar[:, start:end]
        return result


def fake_network(gain: np.ndarray):
    """The stand-in network of the fixtures (gen_golden_uvr5.py): frequency-dependent gain + an odd and an even nonlinearity."""
    def run(spek):
        x = spek.detach().cpu().numpy() if isinstance(spek, torch.Tensor) else np.asarray(spek, dtype=np.float32)
        return (x * gain + 0.3 * np.tanh(x) + 0.05 * x * x).astype(np.float32)
    return run
