#!/usr/bin/env python
"""Golden vectors for the UVR5 MDX-Net denoising SHELL from the REAL reference code.  Build container only: reads
/root/reference, which never travels to the GPU box; the vectors it writes (tests/golden/uvr5_shell.npz) do.

    python oracle/gen_golden_uvr5.py

Imports ``uvr5/multiprocess_cuda_infer.py`` as it lies (its ``Inference`` class: stft / istft / initialize_mix / run_model /
demix_base, :181-301) behind import-only stubs for the packages that are not installed and are not reached on this path (torchaudio,
onnx, onnxruntime), builds ``Inference`` on the CPU from a plain namespace holding the ``ModelData`` fields it reads, and replaces the
one thing that is not in the tree -- the ONNX network -- by a small deterministic function of the spectrogram (odd, even and
frequency-dependent terms, so that the "denoise" +-input averaging and the band crop both matter).  torch.stft / torch.istft are
torch's own, so every number here is the reference's.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/uvr5"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "uvr5_shell.npz")


def fake_network(dim_f):
    gain = (0.25 + 0.75 * np.cos(np.linspace(0.0, 3.0, dim_f)) ** 2).astype(np.float32)[None, None, :, None]

    def run(spek):                       # numpy in, numpy out, like onnxruntime's session.run
        x = np.asarray(spek, dtype=np.float32)
        return (x * gain + 0.3 * np.tanh(x) + 0.05 * x * x).astype(np.float32)
    return run, gain


def main():
    def never(*a, **k):
        raise RuntimeError("import-only stub was called")
    for name in ("torchaudio", "onnx", "onnxruntime"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torchaudio"].load = never
    sys.modules["onnx"].load_model = never
    sys.path.insert(0, REF)
    import multiprocess_cuda_infer as M

    torch.manual_seed(0)
    cases = {}
    base = dict(mdx_n_fft_scale_set=2048, mdx_dim_f_set=768, mdx_dim_t_set=4, is_normalization=False, compensate=1.0,
                chunks=0, margin=441, save_background=False)
    specs = {                              # name: (samples, is_denoise, mdx_batch_size, is_match_mix)
        "plain": (30000, False, 1, False),
        "denoise_b2": (30000, True, 2, False),
        "exact_multiple": (2 * 13312, False, 4, False),     # n % gen_size == 0: the reference pads a whole extra piece
        "match_mix": (20000, False, 1, True),
        "short": (1500, True, 1, False),
    }
    out = {"n_fft": 2048, "dim_f": 768, "dim_t_set": 4}
    for name, (n, den, bs, match) in specs.items():
        md = types.SimpleNamespace(**base, is_denoise=den, mdx_batch_size=bs)
        inf = M.Inference(md, "cpu")
        run, gain = fake_network(inf.dim_f)
        inf.model_run = run
        mix = (torch.randn(2, n) * 0.3).float()
        y = inf.demix_base({0: mix}, is_match_mix=match, device="cpu")
        out[f"{name}_mix"] = mix.numpy()
        out[f"{name}_out"] = y.numpy().astype(np.float32)
        out[f"{name}_cfg"] = np.array([int(den), bs, int(match)], dtype=np.int64)
        print(name, tuple(mix.shape), "->", tuple(y.shape), "rms", float(y.pow(2).mean().sqrt()))
        if name == "plain":                # the two transforms on their own
            waves, pad = inf.initialize_mix(mix)
            spek = inf.stft(waves[:2])
            out["stft_in"] = waves[:2].numpy()
            out["stft_out"] = spek.numpy()
            out["istft_out"] = inf.istft(spek).numpy()
            out["pad"] = np.array([pad])
    out["gain"] = gain
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
