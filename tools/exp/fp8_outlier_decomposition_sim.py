#!/usr/bin/env python
"""CPU study (oracle arithmetic, no GPU): which mixed-precision decomposition brings the MXFP8 block GEMMs back under the 1e-4 mel-MSE target on
weights with outlier residual channels (tests/golden/full_outlier.npz: 22 blocks, 8-step solve, 1 % of the channels x30)?

    python tools/exp/fp8_outlier_decomposition_sim.py [variant ...]

Variants (sites: q = QKV, o = out-projection, 1 = FF1, 2 = FF2; every listed site runs MXFP8 activations x per-channel e4m3 weights):
  plain        all four sites, no decomposition (the engine's unguarded fp8 path)
  kside        QKV / FF1: the flagged K columns of the LayerNorm output leave the MXFP8 image (zeroed) and go through a bf16 side product
  nside        out-proj / FF2: the flagged OUTPUT channels (weight rows) are computed from the bf16 activations and bf16 weights instead
  both         kside + nside
  nside2       out-proj / FF2: the flagged output channels keep their MXFP8 activations but get TWO-TERM e4m3 weights (W = W_hi + W_lo, each e4m3
               with its own per-row scale: ~bf16 precision from two fp8 MFMA passes over 1 % of the rows) -- what the engine implements
  <v>@<sites>  any of the above restricted to the sites named (e.g. both@q1: only QKV and FF1 on fp8, the others bf16)
  <v>+wbf / <v>+abf   ... with bf16 WEIGHTS (only the activations MXFP8) / bf16 ACTIVATIONS (only the weights e4m3) at the fp8 sites
  attn_qk8     (bf16 GEMMs) attention logits from MXFP8 q and k (one E8M0 scale per 32 of the 64 head dimensions), softmax and P.V unchanged
  attn_pv8     (bf16 GEMMs) P.V from MXFP8 P (one scale per query and 32 keys) and MXFP8 V^T (per head dimension and 32 keys), logits unchanged
  attn8        both: the whole attention on the fp8 MFMA (what VERDICT r04 item 4 proposes for the fp8 = 1 path)
Flagged channels: per-output-channel weight scale of attn.to_out.0 / ff.ff.2 above 8x the median (the engine's guard criterion)."""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lemas_tts_amd import synth  # noqa: E402
from lemas_tts_amd.model.layout import DiTArch  # noqa: E402
from oracle import lemas_oracle as O  # noqa: E402
from oracle.mxfp8 import mx_quant, w_quant  # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).float()


class SimDiT(O.OracleDiT):
    def __init__(self, sd, arch, variant, flagged):
        super().__init__(sd, arch, fp8=False)
        self.wbf, self.abf = "+wbf" in variant, "+abf" in variant
        variant = variant.replace("+wbf", "").replace("+abf", "")
        self.sites = "qo12"
        if "@" in variant:
            variant, self.sites = variant.split("@")
        self.variant, self.flagged = variant, flagged
        self._cache = {}

    def attention(self, i, h, mask, freqs):
        if not self.variant.startswith("attn"):
            return super().attention(i, h, mask, freqs)
        import math
        b, n, _ = h.shape
        H, Dh = self.a.heads, self.a.dim_head
        q = f"transformer_blocks.{i}.attn."
        qh = self.lin(q + "to_q", h).view(b, n, H, Dh).transpose(1, 2)
        kh = self.lin(q + "to_k", h).view(b, n, H, Dh).transpose(1, 2)
        vh = self.lin(q + "to_v", h).view(b, n, H, Dh).transpose(1, 2)
        qh, kh = O.rope_apply(qh, freqs), O.rope_apply(kh, freqs)
        c = (1.0 / math.sqrt(Dh)) * 1.4426950408889634           # the QK epilogue hands q over multiplied by scale * log2(e)
        qs = qh * c
        if self.variant in ("attn_qk8", "attn8"):
            qs = mx_quant(qs.reshape(-1, Dh))[2].reshape(qs.shape)
            kq = mx_quant(kh.reshape(-1, Dh))[2].reshape(kh.shape)
        else:
            qs, kq = bf(qs), bf(kh)
        s2 = qs @ kq.transpose(-1, -2)                            # base-2 logits
        if mask is not None:
            s2 = s2.masked_fill(~mask[:, None, None, :], float("-inf"))
        pm = torch.exp2(s2 - s2.amax(-1, keepdim=True))           # (the kernel has no running max; the offset does not change relative precision)
        l = pm.sum(-1, keepdim=True)
        if self.variant in ("attn_pv8", "attn8"):
            npad = (n + 31) // 32 * 32
            pq = mx_quant(F.pad(pm, (0, npad - n)).reshape(-1, npad))[2].reshape(b, H, n, npad)[..., :n]
            vt = F.pad(vh.transpose(-1, -2), (0, npad - n))       # [b, H, Dh, keys]
            vq = mx_quant(vt.reshape(-1, npad))[2].reshape(b, H, Dh, npad)[..., :n]
            o = (pq @ vq.transpose(-1, -2)) / l
        else:
            o = (bf(pm) @ bf(vh)) / l
        o = o.transpose(1, 2).reshape(b, n, H * Dh)
        o = self.lin(q + "to_out.0", bf(o))
        if mask is not None:
            o = o.masked_fill(~mask[..., None], 0.0)
        return o

    def lin(self, name, x):
        if self.variant.startswith("attn"):                       # bf16 GEMMs: the attention study isolates the attention operands
            if name.startswith("transformer_blocks.") and "attn_norm" not in name:
                return F.linear(bf(x), bf(self.p[name + ".weight"]), self.p[name + ".bias"])
            return F.linear(x, self.p[name + ".weight"], self.p[name + ".bias"])
        if not name.startswith("transformer_blocks.") or "attn_norm" in name:
            return F.linear(x, self.p[name + ".weight"], self.p[name + ".bias"])
        w, b = self.p[name + ".weight"], self.p[name + ".bias"]
        site = "q" if (".to_q" in name or ".to_k" in name or ".to_v" in name) else "o" if ".to_out.0" in name else "1" if ".ff.ff.0.0" in name else "2"
        if site not in self.sites:                    # this site stays on bf16 operands
            return F.linear(bf(x), bf(w), b)
        kside = self.variant in ("kside", "both") and (".to_q" in name or ".to_k" in name or ".to_v" in name or ".ff.ff.0.0" in name)
        nside = self.variant in ("nside", "both", "nside2") and (".to_out.0" in name or ".ff.ff.2" in name)
        fl = self.flagged
        key = (name, kside, nside)
        if key not in self._cache:
            wq = w.clone()
            if kside:
                wq[:, fl] = 0.0                       # those K columns leave the e4m3 image
            if nside:
                wq[fl, :] = 0.0                       # those output rows leave it
            w2 = None
            if self.variant == "nside2":                # two-term e4m3 image of the flagged rows
                hi = w_quant(w[fl, :])[2]
                w2 = hi + w_quant(w[fl, :] - hi)[2]
            self._cache[key] = (bf(wq) if self.wbf else w_quant(wq)[2], bf(w), w2)
        w8, wb, w2 = self._cache[key]
        x2 = x.reshape(-1, x.shape[-1])
        xm = x2.clone()
        if kside:
            xm[:, fl] = 0.0
        y = F.linear(bf(xm) if self.abf else mx_quant(xm)[2], w8)
        if kside:                                     # bf16 side product over the flagged K columns
            y = y + F.linear(bf(x2[:, fl]), wb[:, fl])
        if nside and w2 is not None:                  # flagged output channels: the same MXFP8 activations, two-term e4m3 weights
            y[:, fl] = F.linear(mx_quant(x2)[2], w2)
        elif nside:                                   # flagged output channels from the bf16 operands
            y[:, fl] = F.linear(bf(x2), wb[fl, :])
        return (y + b).reshape(*x.shape[:-1], -1)


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    name = os.environ.get("FIXTURE", "full_outlier")
    fx = dict(np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")))
    arch = DiTArch(depth=int(fx["arch_depth"]))
    sd = synth.synth_cfm_state_dict(arch, int(fx["vocab"]), int(fx["wseed"]), outlier=tuple(fx["outlier"]) if "outlier" in fx else None)
    chan = np.zeros(arch.dim, np.float32)
    for k, v in sd.items():
        if k.endswith(".attn.to_out.0.weight") or k.endswith(".ff.ff.2.weight"):
            chan = np.maximum(chan, np.abs(v).max(axis=1) / 448.0)
    flagged = torch.from_numpy(np.nonzero(chan > 8.0 * np.median(chan))[0])
    print(f"{name}: {len(flagged)} flagged channels of {arch.dim}: {flagged.tolist()}", flush=True)
    F_, N = int(fx["F"]), int(fx["N"])
    for variant in (sys.argv[1:] or ["plain", "kside", "nside", "both"]):
        cfm = O.OracleCFM(sd, arch)
        cfm.dit = SimDiT(sd, arch, variant, flagged) if variant != "fp32" else O.OracleDiT(sd, arch)
        t0 = time.time()
        out, _ = cfm.sample(torch.from_numpy(fx["cond"]), torch.from_numpy(fx["text"]), int(fx["duration"][0]), y0=torch.from_numpy(fx["y0"]),
                            steps=int(fx["steps"]), cfg_strength=float(fx["cfg"]), sway_sampling_coef=int(fx["coef"]))
        ref = fx["out"]
        d = out.numpy()[:, F_:N] - ref[:, F_:N]
        print(f"  {variant:8s} mel-MSE vs the reference {float((d.astype(np.float64) ** 2).mean()):.3e}   ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
