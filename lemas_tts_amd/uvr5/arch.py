"""The MDX-Net's architecture description: the constructor arguments of the reference's ``ConvTDFNet`` (``uvr5/lib_v5/mdxnet.py:37-49``)
that shape the network, the state-dict schema they imply (``mdxnet.py:57-101``, ``modules.py:43-70``) and the algorithmic work of one forward.
Shared by the engine wrapper (``MdxEngine``), the weight readers (``onnx_weights.py``), the synthetic-weight generator (``synth.py``) and the
test oracle."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple


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


# Kim_Vocal_1: dim_f 3072, dim_t 256 (n_fft 7680) are MDXConfig's defaults; g 48, l 3, 11 blocks, bn 8, no TDF bias are the values UVR publishes
# for that model -- 16.7 M parameters = the 66.8 MB of the ONNX file.  An assumption until the real file is read (tools/first_contact.py).
KIM_VOCAL_1 = MdxArch()

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
