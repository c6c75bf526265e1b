"""CPU restatement of the reference's MDX-Net separation network -- TEST INFRASTRUCTURE, never imported by the
product (``lemas_tts_amd/``); only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use it.

The network the reference runs through ONNX Runtime (``uvr5/multiprocess_cuda_infer.py:225-238`` session, ``:262-272`` run_model) is an
export of ``ConvTDFNet``, whose definition IS in the reference tree:
  uvr5/lib_v5/mdxnet.py:36-101   ConvTDFNet.__init__  first 1x1 conv, n x [TFC_TDF, 2x2 stride-2 conv], bottleneck TFC_TDF,
                                                      n x [2x2 stride-2 transposed conv, TFC_TDF], last 1x1 conv; n = num_blocks // 2,
                                                      channels g, 2g, ... (n+1)g, frequency bins dim_f, dim_f/2, ...
  uvr5/lib_v5/mdxnet.py:103-127  ConvTDFNet.forward   transpose to [b, c, t, f]; skip connections MULTIPLY (:117); transpose back
  uvr5/lib_v5/modules.py:5-21    TFC                  l x [k x k conv (pad k//2) -> norm -> ReLU]
  uvr5/lib_v5/modules.py:43-74   TFC_TDF              x = tfc(x); x + tdf(x) with tdf = Linear(f -> f/bn) -> norm -> ReLU ->
                                                      Linear(f/bn -> f) -> norm -> ReLU over the LAST (frequency) axis; bn == 0: one
                                                      Linear(f -> f); bn is None: no tdf
``norm`` is BatchNorm2d (optimizer 'rmsprop'; inference = running statistics, eps 1e-5) or GroupNorm(2, c) (optimizer 'adamw').

Pinned by ``tests/golden/mdxnet_*.npz``, which ``oracle/gen_golden_mdxnet.py`` produced by running the reference's own class (behind a
one-line stub of ``pytorch_lightning.LightningModule``) on the seeded weights of ``seeded_state_dict`` below.
What stays UNPINNED: the ONNX-initializer -> state-dict name map of the real ``Kim_Vocal_1.onnx`` (file not in the tree) and its
hyper-parameters (``KIM_VOCAL_1`` below = the values UVR publishes for that model; 16.8 M parameters = the 66.8 MB of the ONNX file).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
# variance gain of the synthetic kernels: at 2.0 (He) the multiplicative skips of an n = 5 network overflow (output rms 1e8), at 1.0 the
# output barely depends on the input (4 %); 1.5 keeps the Kim_Vocal_1 shape at rms ~5 with half of that input-dependent
KERNEL_GAIN = 1.5


@dataclass(frozen=True)
class MdxArch:
    """The constructor arguments of ConvTDFNet that shape the network (mdxnet.py:37-49)."""
    dim_c: int = 4
    dim_f: int = 3072
    dim_t: int = 256
    num_blocks: int = 11
    l: int = 3
    g: int = 48
    k: int = 3
    bn: Optional[int] = 8          # None: no TDF branch; 0: a single Linear(f, f)
    bias: bool = False
    optimizer: str = "rmsprop"     # 'rmsprop' -> BatchNorm2d, 'adamw' -> GroupNorm(2, c)

    @property
    def n(self) -> int:
        return self.num_blocks // 2


KIM_SAMPLE = (slice(None), slice(None), slice(None, None, 16), slice(None, None, 8))     # the stored part of the Kim-shape output: [1, 4, 192, 32]
KIM_VOCAL_1 = MdxArch()                                     # dim_f 3072, dim_t 256, n_fft 7680: MDXConfig's defaults
MINI = MdxArch(dim_f=32, dim_t=16, num_blocks=5, l=2, g=8, k=3, bn=4, bias=True)
MINI_WIDE = MdxArch(dim_f=48, dim_t=8, num_blocks=3, l=3, g=12, k=3, bn=0, bias=False)
MINI_NOTDF = MdxArch(dim_f=16, dim_t=8, num_blocks=3, l=1, g=16, k=3, bn=None, bias=False)
MINI_GN = MdxArch(dim_f=32, dim_t=16, num_blocks=5, l=2, g=8, k=3, bn=4, bias=True, optimizer="adamw")


def _norm_entries(prefix: str, c: int, arch: MdxArch) -> List[Tuple[str, Tuple[int, ...]]]:
    e = [(prefix + "weight", (c,)), (prefix + "bias", (c,))]
    if arch.optimizer == "rmsprop":
        e += [(prefix + "running_mean", (c,)), (prefix + "running_var", (c,))]
    return e


def _tfc_tdf_entries(prefix: str, c: int, f: int, arch: MdxArch) -> List[Tuple[str, Tuple[int, ...]]]:
    e = []
    for j in range(arch.l):
        e += [(f"{prefix}tfc.H.{j}.0.weight", (c, c, arch.k, arch.k)), (f"{prefix}tfc.H.{j}.0.bias", (c,))]
        e += _norm_entries(f"{prefix}tfc.H.{j}.1.", c, arch)
    if arch.bn is None:
        return e
    h = f if arch.bn == 0 else f // arch.bn
    e.append((f"{prefix}tdf.0.weight", (h, f)))
    if arch.bias:
        e.append((f"{prefix}tdf.0.bias", (h,)))
    e += _norm_entries(f"{prefix}tdf.1.", c, arch)
    if arch.bn != 0:
        e.append((f"{prefix}tdf.3.weight", (f, h)))
        if arch.bias:
            e.append((f"{prefix}tdf.3.bias", (f,)))
        e += _norm_entries(f"{prefix}tdf.4.", c, arch)
    return e


def schema(arch: MdxArch) -> List[Tuple[str, Tuple[int, ...]]]:
    """(state-dict key, shape) of every tensor the forward reads, in construction order (mdxnet.py:62-101).  Not listed: ``window`` /
    ``freq_pad`` (AbstractMDXNet's STFT constants, unused by forward) and BatchNorm's ``num_batches_tracked``."""
    g, n = arch.g, arch.n
    e = [("first_conv.0.weight", (g, arch.dim_c, 1, 1)), ("first_conv.0.bias", (g,))] + _norm_entries("first_conv.1.", g, arch)
    f, c = arch.dim_f, g
    for i in range(n):
        e += _tfc_tdf_entries(f"encoding_blocks.{i}.", c, f, arch)
        e += [(f"ds.{i}.0.weight", (c + g, c, 2, 2)), (f"ds.{i}.0.bias", (c + g,))] + _norm_entries(f"ds.{i}.1.", c + g, arch)
        f, c = f // 2, c + g
    e += _tfc_tdf_entries("bottleneck_block.", c, f, arch)
    for i in range(n):
        e += [(f"us.{i}.0.weight", (c, c - g, 2, 2)), (f"us.{i}.0.bias", (c - g,))] + _norm_entries(f"us.{i}.1.", c - g, arch)
        f, c = f * 2, c - g
        e += _tfc_tdf_entries(f"decoding_blocks.{i}.", c, f, arch)
    e += [("final_conv.0.weight", (arch.dim_c, c, 1, 1)), ("final_conv.0.bias", (arch.dim_c,))]
    return e


def seeded_state_dict(arch: MdxArch, seed: int = 0) -> Dict[str, np.ndarray]:
    """Deterministic synthetic weights (numpy Generator: the same bits on the build container and the GPU box, whatever the torch
    version).  Scales keep activations O(1) through the ReLU chain: normal kernels of variance KERNEL_GAIN / fan_in, norm gains around 1, running variances in
    [0.5, 1.5], small biases / running means -- every term of the inference BatchNorm matters in the output."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shape in schema(arch):
        leaf = name.rsplit(".", 1)[1]
        is_norm = _is_norm_module(name, arch)
        if leaf == "running_var":
            v = rng.uniform(0.5, 1.5, shape)
        elif leaf == "running_mean":
            v = rng.normal(0.0, 0.1, shape)
        elif is_norm and leaf == "weight":
            v = rng.uniform(0.8, 1.2, shape)
        elif leaf == "bias":
            v = rng.normal(0.0, 0.05, shape)
        else:                                                   # conv / linear kernels
            if name.startswith("us."):
                fan_in = shape[0]                               # transposed 2x2 stride 2: each output sees ONE tap per input channel
            else:
                fan_in = int(np.prod(shape[1:]))
            v = rng.normal(0.0, np.sqrt(KERNEL_GAIN / fan_in), shape)
        sd[name] = v.astype(np.float32)
    return sd


def _is_norm_module(name: str, arch: MdxArch) -> bool:
    parts = name.split(".")
    if parts[0] in ("first_conv", "ds", "us"):
        return parts[-2] == "1"
    if "tfc" in parts:
        return parts[-2] == "1"
    if "tdf" in parts:
        return parts[-2] in ("1", "4")
    return False


def seeded_input(arch: MdxArch, batch: int, seed: int = 1) -> np.ndarray:
    """A spectrogram-like input [b, dim_c, dim_f, dim_t]: unit-normal with a 1/(1 + f/64) roll-off, the three lowest bins zero as
    Inference.run_model leaves them (multiprocess_cuda_infer.py:264)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((batch, arch.dim_c, arch.dim_f, arch.dim_t)).astype(np.float32)
    x *= (4.0 / (1.0 + np.arange(arch.dim_f, dtype=np.float32) / 64.0))[None, None, :, None]
    x[:, :, :3, :] = 0
    return x


class MdxOracle:
    """fp32 functional restatement of ConvTDFNet.forward (eval mode) on a state dict."""

    def __init__(self, arch: MdxArch, state_dict: Dict[str, np.ndarray]):
        self.arch = arch
        self.sd = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in state_dict.items()}
        missing = [k for k, _ in schema(arch) if k not in self.sd]
        if missing:
            raise KeyError(f"state dict lacks {missing[:4]}{' ...' if len(missing) > 4 else ''}")

    # modules.py: ``norm(c)`` -- BatchNorm2d in eval mode or GroupNorm(2, c); both act on dim 1 of [b, c, t, f]
    def _norm(self, x, p):
        if self.arch.optimizer == "rmsprop":
            return F.batch_norm(x, self.sd[p + "running_mean"], self.sd[p + "running_var"], self.sd[p + "weight"], self.sd[p + "bias"],
                                training=False, eps=BN_EPS)
        return F.group_norm(x, 2, self.sd[p + "weight"], self.sd[p + "bias"], eps=1e-5)

    def _tfc_tdf(self, x, p):                                   # modules.py:5-21, 43-74
        a = self.arch
        for j in range(a.l):
            x = F.conv2d(x, self.sd[f"{p}tfc.H.{j}.0.weight"], self.sd[f"{p}tfc.H.{j}.0.bias"], padding=a.k // 2)
            x = F.relu(self._norm(x, f"{p}tfc.H.{j}.1."))
        if a.bn is None:
            return x
        y = F.linear(x, self.sd[p + "tdf.0.weight"], self.sd.get(p + "tdf.0.bias"))
        y = F.relu(self._norm(y, p + "tdf.1."))
        if a.bn != 0:
            y = F.linear(y, self.sd[p + "tdf.3.weight"], self.sd.get(p + "tdf.3.bias"))
            y = F.relu(self._norm(y, p + "tdf.4."))
        return x + y

    @torch.no_grad()
    def forward(self, x, taps: Optional[dict] = None) -> torch.Tensor:
        """x [b, dim_c, dim_f, dim_t] -> same shape (mdxnet.py:103-127).  ``taps``: filled with intermediate activations."""
        a, sd = self.arch, self.sd
        x = torch.as_tensor(np.asarray(x) if not isinstance(x, torch.Tensor) else x, dtype=torch.float32)
        x = F.relu(self._norm(F.conv2d(x, sd["first_conv.0.weight"], sd["first_conv.0.bias"]), "first_conv.1."))
        x = x.transpose(-1, -2)                                 # [b, g, t, f]
        if taps is not None:
            taps["first"] = x.clone()
        skips = []
        for i in range(a.n):
            x = self._tfc_tdf(x, f"encoding_blocks.{i}.")
            if taps is not None:
                taps[f"enc{i}"] = x.clone()
            skips.append(x)
            x = F.relu(self._norm(F.conv2d(x, sd[f"ds.{i}.0.weight"], sd[f"ds.{i}.0.bias"], stride=2), f"ds.{i}.1."))
            if taps is not None:
                taps[f"ds{i}"] = x.clone()
        x = self._tfc_tdf(x, "bottleneck_block.")
        if taps is not None:
            taps["bottleneck"] = x.clone()
        for i in range(a.n):
            x = F.relu(self._norm(F.conv_transpose2d(x, sd[f"us.{i}.0.weight"], sd[f"us.{i}.0.bias"], stride=2), f"us.{i}.1."))
            x = x * skips[-i - 1]
            if taps is not None:
                taps[f"us{i}"] = x.clone()
            x = self._tfc_tdf(x, f"decoding_blocks.{i}.")
            if taps is not None:
                taps[f"dec{i}"] = x.clone()
        x = x.transpose(-1, -2)
        return F.conv2d(x, sd["final_conv.0.weight"], sd["final_conv.0.bias"])

    __call__ = forward


def flops(arch: MdxArch, batch: int = 1) -> int:
    """Multiply-add FLOPs (2 per MAC) of one forward: the algorithmic work figure of the roofline line."""
    g, n, k = arch.g, arch.n, arch.k
    T, f, c = arch.dim_t, arch.dim_f, g
    total = 2 * arch.dim_c * g * T * f                                   # first 1x1

    def block(c, T, f):
        w = arch.l * 2 * c * c * k * k * T * f
        if arch.bn is not None:
            h = f if arch.bn == 0 else f // arch.bn
            w += 2 * c * T * f * h * (1 if arch.bn == 0 else 2)
        return w
    for _ in range(n):
        total += block(c, T, f)
        total += 2 * c * (c + g) * 4 * (T // 2) * (f // 2)               # 2x2 stride 2
        T, f, c = T // 2, f // 2, c + g
    total += block(c, T, f)
    for _ in range(n):
        total += 2 * c * (c - g) * 4 * T * f                             # transposed 2x2 stride 2: 4 taps per input position
        T, f, c = T * 2, f * 2, c - g
        total += block(c, T, f)
    total += 2 * c * arch.dim_c * T * f                                  # last 1x1
    return total * batch
