"""Checkpoint layout of the LEMAS-TTS acoustic model and the Vocos vocoder.

The build reads the reference's checkpoint files unchanged, so the key names and shapes
below are the reference's (enumerated by instantiating it):

* CFM / DiT:  ``lemas_tts/model/cfm.py:86-172`` (``transformer.*``, ``accent_classifier.*``,
  ``prosody_to_mel.*``) and ``lemas_tts/model/backbones/dit.py:105-169``,
  ``lemas_tts/model/modules.py`` (block internals).
* Vocos (third-party ``vocos`` package, ``charactr/vocos-mel-24khz``): loaded by
  ``lemas_tts/infer/utils_infer.py:120-143`` from ``config.yaml`` + ``pytorch_model.bin``.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field


@dataclass(frozen=True)
class DiTArch:
    """``model.arch`` of ``lemas_tts/configs/multilingual_grl.yaml:48-58``."""
    dim: int = 1024
    depth: int = 22
    heads: int = 16
    dim_head: int = 64
    ff_mult: int = 2
    text_dim: int = 512
    conv_layers: int = 4
    mel_dim: int = 100
    text_mask_padding: bool = True
    qk_norm: object = None
    pe_attn_head: object = None
    checkpoint_activations: bool = False
    conv_pos_kernel: int = 31
    conv_pos_groups: int = 16
    time_freq_dim: int = 256

    @staticmethod
    def from_yaml_arch(arch: dict) -> "DiTArch":
        known = {k: arch[k] for k in arch if k in DiTArch.__dataclass_fields__}
        a = DiTArch(**known)
        if a.qk_norm is not None or a.pe_attn_head is not None:
            raise NotImplementedError("qk_norm / pe_attn_head are null in both shipped configs")
        return a

    def reference_kwargs(self) -> dict:
        """kwargs for the reference ``DiT(**arch)`` (used by the golden generator only)."""
        return dict(dim=self.dim, depth=self.depth, heads=self.heads, dim_head=self.dim_head,
                    ff_mult=self.ff_mult, text_dim=self.text_dim, conv_layers=self.conv_layers,
                    text_mask_padding=True, qk_norm=None, pe_attn_head=None,
                    checkpoint_activations=False)


@dataclass(frozen=True)
class VocosArch:
    """``charactr/vocos-mel-24khz`` config.yaml (backbone + ISTFT head)."""
    input_channels: int = 100
    dim: int = 512
    intermediate_dim: int = 1536
    num_layers: int = 8
    n_fft: int = 1024
    hop_length: int = 256


@dataclass(frozen=True)
class ProsodyArch:
    """ECAPA-TDNN prosody encoder, ``lemas_tts/model/backbones/prosody_encoder.py:390-403``.  The reference reads these
    numbers from ``pretssel_cfg.json`` (``model.prosody_*``), a file that is NOT in the tree: the defaults below are the
    published Pretssel / SeamlessExpressive values and are an assumption until that file is available."""
    channels: tuple = (512, 512, 512, 512, 1536)
    kernel_sizes: tuple = (5, 3, 3, 3, 1)
    dilations: tuple = (1, 2, 3, 4, 1)
    attention_channels: int = 128
    res2net_scale: int = 8
    se_channels: int = 128
    global_context: bool = True
    groups: tuple = (1, 1, 1, 1, 1)
    embed_dim: int = 512
    input_dim: int = 80

    @staticmethod
    def from_pretssel_cfg(model_cfg: dict) -> "ProsodyArch":
        """``_build_prosody_encoder`` key mapping (prosody_encoder.py:390-403)."""
        return ProsodyArch(channels=tuple(model_cfg["prosody_channels"]), kernel_sizes=tuple(model_cfg["prosody_kernel_sizes"]),
                           dilations=tuple(model_cfg["prosody_dilations"]), attention_channels=model_cfg["prosody_attention_channels"],
                           res2net_scale=model_cfg["prosody_res2net_scale"], se_channels=model_cfg["prosody_se_channels"],
                           global_context=model_cfg["prosody_global_context"], groups=tuple(model_cfg["prosody_groups"]),
                           embed_dim=model_cfg["prosody_embed_dim"], input_dim=model_cfg["input_feat_per_channel"])

    def reference_kwargs(self) -> dict:
        return dict(channels=list(self.channels), kernel_sizes=list(self.kernel_sizes), dilations=list(self.dilations),
                    attention_channels=self.attention_channels, res2net_scale=self.res2net_scale, se_channels=self.se_channels,
                    global_context=self.global_context, groups=list(self.groups), embed_dim=self.embed_dim, input_dim=self.input_dim)


def prosody_param_shapes(a: ProsodyArch) -> "OrderedDict[str, tuple]":
    """state-dict names / shapes of ``ECAPA_TDNN`` (prosody_encoder.py:36-101, 136-331), the names the reference's own
    loader expects after stripping ``prosody_encoder.`` (:406-424)."""
    if any(g != 1 for g in a.groups):
        raise NotImplementedError("grouped TDNN convolutions (groups != 1) are not built")
    s: "OrderedDict[str, tuple]" = OrderedDict()

    def tdnn(p, cin, cout, k):
        s[p + "conv.weight"] = (cout, cin, k)
        s[p + "conv.bias"] = (cout,)
        s[p + "norm.weight"] = (cout,)
        s[p + "norm.bias"] = (cout,)

    ch = a.channels
    tdnn("blocks.0.", a.input_dim, ch[0], a.kernel_sizes[0])
    for i in range(1, len(ch) - 1):
        p = f"blocks.{i}."
        tdnn(p + "tdnn1.", ch[i - 1], ch[i], 1)
        sub = ch[i] // a.res2net_scale
        for j in range(a.res2net_scale - 1):
            tdnn(p + f"res2net_block.blocks.{j}.", sub, sub, a.kernel_sizes[i])
        tdnn(p + "tdnn2.", ch[i], ch[i], 1)
        s[p + "se_block.conv1.weight"] = (a.se_channels, ch[i], 1)
        s[p + "se_block.conv1.bias"] = (a.se_channels,)
        s[p + "se_block.conv2.weight"] = (ch[i], a.se_channels, 1)
        s[p + "se_block.conv2.bias"] = (ch[i],)
        if ch[i - 1] != ch[i]:
            s[p + "shortcut.weight"] = (ch[i], ch[i - 1], 1)
            s[p + "shortcut.bias"] = (ch[i],)
    tdnn("mfa.", ch[-1], ch[-1], a.kernel_sizes[-1])
    tdnn("asp.tdnn.", ch[-1] * (3 if a.global_context else 1), a.attention_channels, 1)
    s["asp.conv.weight"] = (ch[-1], a.attention_channels, 1)
    s["asp.conv.bias"] = (ch[-1],)
    s["asp_norm.weight"] = (2 * ch[-1],)
    s["asp_norm.bias"] = (2 * ch[-1],)
    s["fc.weight"] = (a.embed_dim, 2 * ch[-1], 1)
    s["fc.bias"] = (a.embed_dim,)
    return s


def cfm_param_shapes(a: DiTArch, vocab_size: int, prosody: bool = False) -> "OrderedDict[str, tuple]":
    """name -> shape for every tensor of the (stripped) CFM state dict the loader must accept."""
    d, td, inner = a.dim, a.text_dim, a.heads * a.dim_head
    s: "OrderedDict[str, tuple]" = OrderedDict()
    T = "transformer."
    s[T + "time_embed.time_mlp.0.weight"] = (d, a.time_freq_dim)
    s[T + "time_embed.time_mlp.0.bias"] = (d,)
    s[T + "time_embed.time_mlp.2.weight"] = (d, d)
    s[T + "time_embed.time_mlp.2.bias"] = (d,)
    s[T + "text_embed.text_embed.weight"] = (vocab_size + 1, td)
    for i in range(a.conv_layers):
        p = f"{T}text_embed.text_blocks.{i}."
        s[p + "dwconv.weight"] = (td, 1, 7)
        s[p + "dwconv.bias"] = (td,)
        s[p + "norm.weight"] = (td,)
        s[p + "norm.bias"] = (td,)
        s[p + "pwconv1.weight"] = (2 * td, td)
        s[p + "pwconv1.bias"] = (2 * td,)
        s[p + "grn.gamma"] = (1, 1, 2 * td)
        s[p + "grn.beta"] = (1, 1, 2 * td)
        s[p + "pwconv2.weight"] = (td, 2 * td)
        s[p + "pwconv2.bias"] = (td,)
    if prosody:
        s[T + "prosody_text_proj.weight"] = (td, 512)
        s[T + "prosody_text_proj.bias"] = (td,)
    s[T + "input_embed.proj.weight"] = (d, 2 * a.mel_dim + td)
    s[T + "input_embed.proj.bias"] = (d,)
    for j in (0, 2):
        s[f"{T}input_embed.conv_pos_embed.conv1d.{j}.weight"] = (d, d // a.conv_pos_groups, a.conv_pos_kernel)
        s[f"{T}input_embed.conv_pos_embed.conv1d.{j}.bias"] = (d,)
    s[T + "rotary_embed.inv_freq"] = (a.dim_head // 2,)
    for i in range(a.depth):
        p = f"{T}transformer_blocks.{i}."
        s[p + "attn_norm.linear.weight"] = (6 * d, d)
        s[p + "attn_norm.linear.bias"] = (6 * d,)
        for n in ("to_q", "to_k", "to_v"):
            s[p + f"attn.{n}.weight"] = (inner, d)
            s[p + f"attn.{n}.bias"] = (inner,)
        s[p + "attn.to_out.0.weight"] = (d, inner)
        s[p + "attn.to_out.0.bias"] = (d,)
        s[p + "ff.ff.0.0.weight"] = (a.ff_mult * d, d)
        s[p + "ff.ff.0.0.bias"] = (a.ff_mult * d,)
        s[p + "ff.ff.2.weight"] = (d, a.ff_mult * d)
        s[p + "ff.ff.2.bias"] = (d,)
    s[T + "norm_out.linear.weight"] = (2 * d, d)
    s[T + "norm_out.linear.bias"] = (2 * d,)
    s[T + "proj_out.weight"] = (a.mel_dim, d)
    s[T + "proj_out.bias"] = (a.mel_dim,)
    if prosody:
        s["prosody_to_mel.weight"] = (a.mel_dim, 512)
        s["prosody_to_mel.bias"] = (a.mel_dim,)
    # loaded, unused at inference (cfm.py:171)
    s["accent_classifier.net.0.weight"] = (d, a.mel_dim)
    s["accent_classifier.net.0.bias"] = (d,)
    s["accent_classifier.net.3.weight"] = (12, d)
    s["accent_classifier.net.3.bias"] = (12,)
    return s


def vocos_param_shapes(v: VocosArch = VocosArch()) -> "OrderedDict[str, tuple]":
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["backbone.embed.weight"] = (v.dim, v.input_channels, 7)
    s["backbone.embed.bias"] = (v.dim,)
    s["backbone.norm.weight"] = (v.dim,)
    s["backbone.norm.bias"] = (v.dim,)
    for i in range(v.num_layers):
        p = f"backbone.convnext.{i}."
        s[p + "dwconv.weight"] = (v.dim, 1, 7)
        s[p + "dwconv.bias"] = (v.dim,)
        s[p + "norm.weight"] = (v.dim,)
        s[p + "norm.bias"] = (v.dim,)
        s[p + "pwconv1.weight"] = (v.intermediate_dim, v.dim)
        s[p + "pwconv1.bias"] = (v.intermediate_dim,)
        s[p + "pwconv2.weight"] = (v.dim, v.intermediate_dim)
        s[p + "pwconv2.bias"] = (v.dim,)
        s[p + "gamma"] = (v.dim,)
    s["backbone.final_layer_norm.weight"] = (v.dim,)
    s["backbone.final_layer_norm.bias"] = (v.dim,)
    s["head.out.weight"] = (v.n_fft + 2, v.dim)
    s["head.out.bias"] = (v.n_fft + 2,)
    s["head.istft.window"] = (v.n_fft,)
    return s


# keys dropped by the reference loader before the strict load (utils_infer.py:223-235)
DROPPED_ON_LOAD = ("initted", "step", "mel_spec.mel_stft.mel_scale.fb", "mel_spec.mel_stft.spectrogram.window",
                   "ctc.proj.0.weight", "ctc.proj.0.bias", "ctc.ctc_proj.weight", "ctc.ctc_proj.bias")
