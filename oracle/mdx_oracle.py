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
one-line stub of ``pytorch_lightning.LightningModule``) on the seeded weights of ``lemas_tts_amd/synth.py synth_mdx_state_dict``.
What stays UNPINNED: the ONNX-initializer -> state-dict name map of the real ``Kim_Vocal_1.onnx`` (file not in the tree) and its
hyper-parameters (``lemas_tts_amd/uvr5/arch.py KIM_VOCAL_1`` = the values UVR publishes for that model; 16.8 M parameters = the 66.8 MB of the ONNX file).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

# the architecture description, its state-dict schema and the seeded weights / inputs live in the product package (the engine wrapper, the
# weight readers and the bench tool need them too); re-exported here under the names the tests use
from lemas_tts_amd.synth import synth_mdx_input as seeded_input, synth_mdx_state_dict as seeded_state_dict   # noqa: F401
from lemas_tts_amd.uvr5.arch import KIM_VOCAL_1, MdxArch, flops, schema   # noqa: F401

BN_EPS = 1e-5


KIM_SAMPLE = (slice(None), slice(None), slice(None, None, 16), slice(None, None, 8))     # the stored part of the Kim-shape output: [1, 4, 192, 32]
MINI = MdxArch(dim_f=32, dim_t=16, num_blocks=5, l=2, g=8, k=3, bn=4, bias=True)
MINI_WIDE = MdxArch(dim_f=48, dim_t=8, num_blocks=3, l=3, g=12, k=3, bn=0, bias=False)
MINI_NOTDF = MdxArch(dim_f=16, dim_t=8, num_blocks=3, l=1, g=16, k=3, bn=None, bias=False)
MINI_GN = MdxArch(dim_f=32, dim_t=16, num_blocks=5, l=2, g=8, k=3, bn=4, bias=True, optimizer="adamw")


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


