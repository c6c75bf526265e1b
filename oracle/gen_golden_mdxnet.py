#!/usr/bin/env python
"""Golden vectors for the MDX-Net separation network from the REFERENCE's own class.  Build container only: reads
/root/reference, which never travels to the GPU box; the vectors it writes (tests/golden/mdxnet_*.npz) do.

    python oracle/gen_golden_mdxnet.py [--skip-kim]

Imports ``uvr5/lib_v5/mdxnet.py`` (ConvTDFNet, :36-127) and ``uvr5/lib_v5/modules.py`` (TFC / TFC_TDF, :5-74) as they lie, behind a stub
of the one package that is not installed and contributes no arithmetic (``pytorch_lightning.LightningModule`` -> ``torch.nn.Module`` with a
no-op ``save_hyperparameters``), loads the seeded weights of ``oracle/mdx_oracle.seeded_state_dict`` with ``strict=True`` over every key the
forward reads (so the key names and shapes of ``mdx_oracle.schema`` are the reference's), puts the module in eval mode and runs ``forward``.
Weights and inputs are regenerated from their seeds on the GPU box (numpy Generator), so only outputs are stored:
  mdxnet_mini*.npz   small architectures (BatchNorm two-linear TDF with bias / single-linear TDF / no TDF / GroupNorm): input, output and the
                     intermediate activations after every stage
  mdxnet_kim.npz     the Kim_Vocal_1 shape [1, 4, 3072, 256] (17.5 M parameters, 0.74 TFLOP): a strided sample of the output plus its
                     moments (the full tensors are 12.6 MB each)
"""
import argparse
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mdx_oracle as MO   # noqa: E402

REF = "/root/reference/uvr5"
GOLD = os.path.join(ROOT, "tests", "golden")
KIM_SAMPLE = MO.KIM_SAMPLE


def reference_class():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass
    pl.LightningModule = LightningModule
    sys.modules["pytorch_lightning"] = pl
    sys.path.insert(0, REF)
    from lib_v5.mdxnet import ConvTDFNet
    return ConvTDFNet


def build(ConvTDFNet, arch: MO.MdxArch, seed: int):
    net = ConvTDFNet("vocals", 1e-4, arch.optimizer, arch.dim_c, arch.dim_f, arch.dim_t, 2 * (arch.dim_f + 8), 1024,
                     arch.num_blocks, arch.l, arch.g, arch.k, arch.bn, arch.bias, 0)
    sd = MO.seeded_state_dict(arch, seed)
    own = net.state_dict()
    unused = {k for k in own if k in ("window", "freq_pad") or k.endswith("num_batches_tracked")}
    assert set(own) - unused == set(sd), sorted(set(own) - unused ^ set(sd))[:8]
    for k, v in sd.items():
        assert tuple(own[k].shape) == v.shape, (k, tuple(own[k].shape), v.shape)
    full = {k: own[k] for k in unused}
    full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    net.load_state_dict(full, strict=True)
    return net.eval(), sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-kim", action="store_true")
    args = ap.parse_args()
    ConvTDFNet = reference_class()
    torch.set_num_threads(os.cpu_count() or 1)
    minis = {"mini": (MO.MINI, 2), "mini_wide": (MO.MINI_WIDE, 1), "mini_notdf": (MO.MINI_NOTDF, 3), "mini_gn": (MO.MINI_GN, 2)}
    for name, (arch, batch) in minis.items():
        net, sd = build(ConvTDFNet, arch, seed=10)
        x = MO.seeded_input(arch, batch, seed=11)
        with torch.no_grad():
            y = net(torch.from_numpy(x))
        taps = {}
        yo = MO.MdxOracle(arch, sd).forward(x, taps)
        err = float((yo - y).abs().max())
        print(f"{name}: out {tuple(y.shape)} rms {float(y.pow(2).mean().sqrt()):.4f}  restatement-vs-reference max|d| {err:.2e}")
        assert err < 1e-5
        out = {"output": y.numpy(), "input": x, "batch": np.array([batch]), "seed_weights": np.array([10]), "seed_input": np.array([11])}
        out.update({f"tap_{k}": v.numpy() for k, v in taps.items()})
        np.savez_compressed(os.path.join(GOLD, f"mdxnet_{name}.npz"), **out)
    if not args.skip_kim:
        arch = MO.KIM_VOCAL_1
        net, sd = build(ConvTDFNet, arch, seed=20)
        x = MO.seeded_input(arch, 1, seed=21)
        t0 = time.time()
        with torch.no_grad():
            y = net(torch.from_numpy(x))
        dt = time.time() - t0
        print(f"kim: out {tuple(y.shape)} rms {float(y.pow(2).mean().sqrt()):.4f}  reference forward {dt:.1f} s on {torch.get_num_threads()} threads "
              f"= {MO.flops(arch) / dt / 1e9:.0f} GFLOP/s")
        yn = y.numpy()
        np.savez_compressed(os.path.join(GOLD, "mdxnet_kim.npz"), sample=yn[KIM_SAMPLE].copy(), mean=np.array([yn.mean(dtype=np.float64)]),
                            rms=np.array([np.sqrt((yn.astype(np.float64) ** 2).mean())]),
                            channel_rms=np.sqrt((yn.astype(np.float64) ** 2).mean(axis=(0, 2, 3))),
                            row_sums=yn.astype(np.float64).sum(axis=(0, 1, 3)).astype(np.float32),          # [3072]: every frequency bin is touched
                            col_sums=yn.astype(np.float64).sum(axis=(0, 1, 2)).astype(np.float32),          # [256]: every frame
                            seed_weights=np.array([20]), seed_input=np.array([21]), ref_seconds=np.array([dt]),
                            ref_threads=np.array([torch.get_num_threads()]))
    for f in sorted(os.listdir(GOLD)):
        if f.startswith("mdxnet_"):
            print(f, os.path.getsize(os.path.join(GOLD, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
