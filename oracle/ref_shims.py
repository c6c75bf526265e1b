"""Import harness for the UPSTREAM reference (/root/reference) -- build container only.

TEST INFRASTRUCTURE.  Nothing in the product path imports this file.  It exists so that
``oracle/gen_golden.py`` can run the real reference code (``lemas_tts.model.cfm.CFM`` /
``lemas_tts.model.backbones.dit.DiT``) on this CPU-only container and dump golden
input/output vectors into ``tests/golden``.  The reference never travels to the GPU box.

The reference depends on third-party packages that are not installed here (no network):
torchaudio, torchdiffeq, x_transformers, librosa, jieba, pypinyin.  Two kinds of stand-ins:

* import-only stubs (torchaudio, librosa, jieba, pypinyin): never called on the golden path,
  because ``cond`` is handed to ``CFM.sample`` as a 3-D mel (cfm.py:232) so ``MelSpec`` is
  never touched.
* ARITHMETIC stand-ins, written from the published upstream algorithm, for
    - ``torchdiffeq.odeint(method="euler")``  (pin torchdiffeq==0.2.4, requirements.txt:167;
      call site cfm.py:456): fixed-grid Euler on the caller's grid,
      y[k+1] = y[k] + (t[k+1]-t[k]) * f(t[k], y[k]);
    - ``x_transformers.x_transformers.RotaryEmbedding / apply_rotary_pos_emb``
      (pin x-transformers>=1.31.14, requirements.txt:180; call sites dit.py:143,236 and
      modules.py:470-480): inv_freq = 10000^(-2j/d); freqs interleaved-duplicated
      [f0,f0,f1,f1,...]; rotate_half on adjacent pairs (x0,x1)->(-x1,x0).
  These two are third-party semantics that the reference tree does not contain:
  PARITY UNPINNED at those two boundaries; everything in-tree is pinned.
"""
from __future__ import annotations

import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _module(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


# --- arithmetic stand-in: torchdiffeq fixed-grid Euler ---------------------------------------
def odeint(func, y0, t, method="euler", **_):
    assert method == "euler", "only the fixed-grid Euler solver is restated"
    assert bool((t[1:] > t[:-1]).all()), "t must be strictly increasing"
    sol = [y0]
    y = y0
    for k in range(t.shape[0] - 1):
        t0, t1 = t[k], t[k + 1]
        dt = t1 - t0
        y = y + dt * func(t0.to(y.dtype), y)
        sol.append(y)
    return torch.stack(sol, dim=0)


# --- arithmetic stand-in: x_transformers rotary embedding -------------------------------------
class RotaryEmbedding(torch.nn.Module):
    def __init__(self, dim, base=10000.0):
        super().__init__()
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq)

    def forward_from_seq_len(self, seq_len):
        t = torch.arange(seq_len, device=self.inv_freq.device)
        return self.forward(t)

    def forward(self, t):
        if t.ndim == 1:
            t = t[None, :]
        freqs = torch.einsum("bi,j->bij", t.type_as(self.inv_freq), self.inv_freq)
        freqs = torch.stack((freqs, freqs), dim=-1).flatten(-2)  # '... d r -> ... (d r)'
        return freqs, 1.0


def _rotate_half(x):
    x = x.unflatten(-1, (-1, 2))
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def apply_rotary_pos_emb(t, freqs, scale=1):
    rot_dim, seq_len, orig_dtype = freqs.shape[-1], t.shape[-2], t.dtype
    freqs = freqs[:, -seq_len:, :]
    if t.ndim == 4 and freqs.ndim == 3:
        freqs = freqs[:, None]
    t, t_unrot = t[..., :rot_dim], t[..., rot_dim:]
    t = (t * freqs.cos() * scale) + (_rotate_half(t) * freqs.sin() * scale)
    return torch.cat((t, t_unrot), dim=-1).type(orig_dtype)


_installed = False


def install():
    """Register the stand-ins and make ``lemas_tts.model.*`` importable from the reference tree
    without executing ``lemas_tts/__init__.py`` (which pulls api.py -> soundfile/hydra/vocos)."""
    global _installed
    if _installed:
        return
    pkg = types.ModuleType("lemas_tts")
    pkg.__path__ = [f"{REFERENCE_ROOT}/lemas_tts"]
    sys.modules["lemas_tts"] = pkg

    def _never(*a, **k):  # import-only stubs must not be reached on the golden path
        raise RuntimeError("import-only stub was called: golden path must not reach this")

    ta = _module("torchaudio", load=_never)
    ta.transforms = _module("torchaudio.transforms", MelSpectrogram=_never, Resample=_never)
    ta.functional = _module("torchaudio.functional", resample=_never)
    ta.compliance = _module("torchaudio.compliance")
    ta.compliance.kaldi = _module("torchaudio.compliance.kaldi", fbank=_never)
    lib = _module("librosa")
    lib.filters = _module("librosa.filters", mel=_never)
    _module("jieba")
    _module("pypinyin", lazy_pinyin=_never, Style=object)
    _module("torchdiffeq", odeint=odeint)
    xt = _module("x_transformers")
    xt.x_transformers = _module(
        "x_transformers.x_transformers",
        RotaryEmbedding=RotaryEmbedding,
        apply_rotary_pos_emb=apply_rotary_pos_emb,
    )
    _installed = True


def build_reference_cfm(arch: dict, vocab_size: int, state_dict: dict, use_prosody: bool = False):
    """Instantiate the reference CFM(DiT(**arch)) (utils_infer.py:281-298) and strict-load weights."""
    install()
    from lemas_tts.model.backbones.dit import DiT
    from lemas_tts.model.cfm import CFM

    dit = DiT(**arch, text_num_embeds=vocab_size, mel_dim=100, use_prosody_encoder=use_prosody)
    cfm = CFM(
        transformer=dit,
        mel_spec_kwargs=dict(n_fft=1024, hop_length=256, win_length=1024, n_mel_channels=100,
                             target_sample_rate=24000, mel_spec_type="vocos"),
        odeint_kwargs=dict(method="euler"),
        vocab_char_map=None,
    )
    if use_prosody:
        # cfm.py:139-145 builds these only together with the (unavailable) Pretssel encoder files;
        # attach the two projection layers by hand so the a-P arithmetic (cfm.py:313-318) is exercised.
        cfm.prosody_to_mel = torch.nn.Linear(512, 100)
    missing, unexpected = cfm.load_state_dict(state_dict, strict=False)
    missing = [k for k in missing if not k.startswith("mel_spec")]
    assert not missing and not unexpected, (missing, unexpected)
    return cfm.eval()
