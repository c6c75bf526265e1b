"""Host-side mirror of the reference sampler object: ``lemas_tts.model.cfm.CFM`` (inference only).

Same constructor role and the same ``sample(...)`` signature, argument meaning, defaults and error
behaviour as ``lemas_tts/model/cfm.py:206-227``; the body only does the integer/boolean bookkeeping
of ``cfm.py:284-339`` (token ids, lengths, masks, duration) and the t-grid (``:445-453``) on the
host, then hands the batch to the MI355X engine, which runs the hoists, the NFE-step Euler loop with
CFG folded into the batch, and the final ``where``.  There is no CPU path.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
import torch.nn.functional as F

from ..engine import DiTEngine
from .layout import DiTArch


def lens_to_mask(lens: torch.Tensor, length: Optional[int] = None) -> torch.Tensor:
    """``model/utils.py:42-47``."""
    if length is None:
        length = int(lens.amax())
    return torch.arange(length, device=lens.device)[None, :] < lens[:, None]


def pad_rows(rows, padding_value=0) -> torch.Tensor:
    """``torch.nn.utils.rnn.pad_sequence(rows, padding_value, batch_first=True)`` written out.  The library call hands ONE [1875, 100] row to its
    thread pool: 20 ms on 8 host threads, 0.07 ms on one (measured in the build container; round 6) -- per utterance, on the API path, before
    the first launch.  A single row is returned as a view."""
    if len(rows) == 1:
        return rows[0][None]
    n = max(r.shape[0] for r in rows)
    out = rows[0].new_full((len(rows), n) + tuple(rows[0].shape[1:]), padding_value)
    for i, r in enumerate(rows):
        out[i, :r.shape[0]] = r
    return out


def list_str_to_idx(text, vocab_char_map: dict, padding_value: int = -1) -> torch.Tensor:
    """``model/utils.py:87-94``: unknown token -> 0, right-pad with -1."""
    rows = [torch.tensor([vocab_char_map.get(c, 0) for c in t], dtype=torch.long) for t in text]
    return pad_rows(rows, padding_value)


def list_str_to_tensor(text, padding_value: int = -1) -> torch.Tensor:
    """``model/utils.py:81-84`` (byte tokenizer)."""
    rows = [torch.tensor([*bytes(t, "UTF-8")], dtype=torch.long) for t in text]
    return pad_rows(rows, padding_value)


def clip_and_shuffle(mel: torch.Tensor, mel_len: int, sample_rate: int = 24000, hop_length: int = 256, ratio=None) -> torch.Tensor:
    """``cfm.py:39-84``: the accent-GRL conditioning segment -- a crop of the prompt mel [n_mels, T] cut into ~1 s pieces,
    shuffled, and topped up with randomly chosen pieces to the original length.  Draws come from Python's ``random`` in the
    reference's order (crop length unless ``ratio``, crop start, shuffle, then one ``choice`` per top-up piece), so a seeded
    interpreter reproduces the reference's segment."""
    import random
    fps = int(sample_rate / hop_length)
    seg = int(mel_len * ratio) if ratio else random.randint(int(0.25 * mel_len), int(0.75 * mel_len))
    begin = random.randint(0, max(0, mel_len - seg))
    piece = mel[:, begin: begin + seg]
    pieces = [piece[:, j: j + fps] for j in range(0, piece.size(1), fps)]
    random.shuffle(pieces)
    order, have = list(pieces), piece.size(1)
    if have < mel_len:
        extra = 0
        while extra < mel_len:
            pick = random.choice(pieces)
            order.append(pick)
            extra += pick.size(1)
    out = torch.cat(order, dim=1)[:, :mel_len]
    assert out.shape == mel.shape, (out.shape, mel.shape)
    return out


def compute_sway_max(steps: int, t_start: float = 0.0, min_ratio: float = 1e-9, safety_factor: float = 0.7) -> float:
    """``cfm.py:343-373`` with the arguments of the call at ``:447``."""
    assert 0.0 <= t_start < 1.0
    dt = (1.0 - t_start) / max(1, steps)
    p_max = 11.0 if dt >= 0.9 else math.log(min_ratio) / math.log(dt)
    return max(0.0, p_max - 1.0) * float(safety_factor)


def time_grid(steps: int, sway_sampling_coef, t_start: float = 0.0) -> torch.Tensor:
    """``cfm.py:445-453``: linspace(t_start,1,steps+1) ** (1 + min(sway_max, coef)), fp32, python ``min`` semantics (``t_start`` > 0 only in
    the ``duplicate_test`` corner, :438-443)."""
    t = torch.linspace(t_start, 1, int(steps + 1), dtype=torch.float32)
    sway_max = torch.tensor(compute_sway_max(steps, t_start=t_start), dtype=torch.float32)
    if sway_sampling_coef is not None:
        return t ** (1 + min(sway_max, sway_sampling_coef))
    return t ** (1 + sway_max)


class CFM:
    """Inference-time stand-in for ``lemas_tts.model.cfm.CFM``: ``transformer`` is the weight holder
    (arch + state dict) and the compute lives in a :class:`DiTEngine`."""

    def __init__(self, arch: DiTArch, vocab_size: int, state_dict: dict, *, vocab_char_map: Optional[dict] = None,
                 device="cuda:0", use_prosody_encoder: bool = False, num_channels: int = 100,
                 odeint_kwargs: dict = dict(method="euler"), mel_spec_module=None, fp8_weights: bool = False,
                 prosody_encoder=None):
        if odeint_kwargs.get("method", "euler") != "euler":
            raise NotImplementedError("only the fixed-grid Euler solver of the shipped configs is built")
        self.arch = arch
        self.num_channels = num_channels
        self.vocab_char_map = vocab_char_map
        self.use_prosody_encoder = use_prosody_encoder
        if mel_spec_module is None:              # wav -> mel front edge, cfm.py:113
            from .modules import MelSpec
            mel_spec_module = MelSpec(n_mel_channels=num_channels, device=device)
        self.mel_spec = mel_spec_module
        self.prosody_encoder = prosody_encoder   # model.prosody_encoder.ProsodyEncoder (cfm.py:139-145), or None: embeds are inputs
        self.odeint_kwargs = odeint_kwargs
        self.engine = DiTEngine(arch, vocab_size, state_dict, device=device, prosody=use_prosody_encoder)
        if fp8_weights:      # BASELINE config 5: block GEMMs on fp8-e4m3 MFMA (MXFP8 activations, per-channel weight scales); 2 = the
            self.engine.set_option("fp8", 2 if fp8_weights == 2 else 1)     # weights-only accuracy point (bf16 activations)

    @property
    def device(self):
        return self.engine.device

    def eval(self):
        return self

    @torch.no_grad()
    def sample(self, cond, text, duration, *, lens=None, steps=32, cfg_strength=1.0, sway_sampling_coef=None,
               seed=None, max_duration=4096, vocoder: Optional[Callable] = None, no_ref_audio=False,
               duplicate_test=False, t_inter=0.1, edit_mask=None, use_acc_grl=True, use_prosody_encoder=True,
               ref_ratio=1, y0: Optional[torch.Tensor] = None, prosody_embeds: Optional[torch.Tensor] = None,
               return_trajectory: bool = False, cond_noise: Optional[torch.Tensor] = None):
        """Extra keyword-only inputs over the reference: ``y0`` (explicit ODE start; the reference draws it on
        its own device, cfm.py:430-435), ``prosody_embeds`` [B,512] (what the prosody encoder would return,
        cfm.py:261-263) and ``return_trajectory`` (the reference always returns it; its callers discard it)."""
        if use_acc_grl and ref_ratio is None:
            raise TypeError("'<' not supported between instances of 'NoneType' and 'int'")  # cfm.py:273 hazard
        dev = self.device
        if cond.ndim == 2:
            raw_audio = cond
            if prosody_embeds is None and use_prosody_encoder and self.use_prosody_encoder:
                if self.prosody_encoder is None:
                    # the reference's model always owns its encoder when built with use_prosody_encoder (cfm.py:139-145, default
                    # assets utils_infer.py:276-280): skipping the conditioning here would diverge from it without a word
                    raise RuntimeError("use_prosody_encoder is set and raw audio was given, but this model has no prosody encoder "
                                       "(pass prosody_ckpt_path to load_model, or prosody_embeds=...)")
                prosody_embeds = self.prosody_encoder.embed_prompt(raw_audio, self.mel_spec.target_sample_rate)   # cfm.py:248-262
            cond = self.mel_spec(cond).permute(0, 2, 1)              # cfm.py:232-236
        assert cond.shape[-1] == self.num_channels
        cond = cond.to(dev, torch.float32)
        batch, cond_seq_len = cond.shape[:2]
        cond_mean = cond.mean(dim=1, keepdim=True) if no_ref_audio else None    # cfm.py:239 (only the no_ref_audio branch reads it)
        cond_grl = None
        if use_acc_grl:                                                      # cfm.py:266-283: built from the RAW prompt mel
            if ref_ratio < 1:
                if cond.shape[0] != 1:
                    raise RuntimeError("ref_ratio < 1 needs a single prompt (the reference squeezes the batch dimension, cfm.py:274)")
                cond_grl = clip_and_shuffle(cond[0].T.cpu(), cond.shape[1], ratio=ref_ratio).T[None].to(dev)
            else:
                cond_grl = cond
        if lens is None:
            lens = torch.full((batch,), cond_seq_len, dtype=torch.long)
        lens = lens.to("cpu", torch.long)

        if isinstance(text, list):
            text = list_str_to_idx(text, self.vocab_char_map) if self.vocab_char_map is not None else list_str_to_tensor(text)
            assert text.shape[0] == batch
        text = text.to("cpu", torch.long)
        if text.numel() and (int(text.min()) < -1 or int(text.max()) >= self.engine.vocab_size):
            # nn.Embedding(vocab + 1, ...) on text + 1 (dit.py:37,53-61) raises for these ids
            raise IndexError("index out of range in self")

        cond_mask = lens_to_mask(lens)
        if edit_mask is not None:
            cond_mask = cond_mask & edit_mask.to("cpu")
        if isinstance(duration, int):
            duration = torch.full((batch,), duration, dtype=torch.long)
        duration = duration.to("cpu", torch.long)
        duration = torch.maximum(torch.maximum((text != -1).sum(dim=-1), lens) + 1, duration).clamp(max=max_duration)
        n = int(duration.amax())

        test_cond = None
        if duplicate_test:
            # cfm.py:307-309 ("duplicate test corner for inner time step observation"): the RAW prompt mel shifted behind itself, rows
            # [F, 2F) of an n-row tensor (a negative right pad crops, as F.pad does there)
            test_cond = F.pad(cond, (0, 0, cond_seq_len, n - 2 * cond_seq_len), value=0.0)
        # cfm.py:311 right-pads the prompt mel with zeros up to n (and CROPS it when `lens` lets the duration fall below the prompt
        # length: negative pad).  The padding is the library's (lemas_sample_args.cond_rows): no pad copy here
        if cond_seq_len > n:
            cond = cond[:, :n].contiguous()
        pros = prosody_embeds if (use_prosody_encoder and self.use_prosody_encoder) else None
        if no_ref_audio:
            # cfm.py:320-324: the conditioning becomes noise around the prompt's mean (the draw is an explicit input here,
            # ``cond_noise``; the reference takes it from the global RNG).  It OVERWRITES cond after the prosody-to-mel projection
            # was added (:313-318): with the prosody encoder on, the embedding then acts through the text side only
            # (dit.py:225-233) -- the engine is told not to add its mel projection (prosody_text_only).
            if cond_noise is None:
                cond_noise = torch.randn(batch, n, self.num_channels, dtype=torch.float32)   # host generator, like the y0 draw below
            rc = cond_noise.to(dev, torch.float32) * 0.1 + cond_mean
            cond = rc / rc.mean(dim=1, keepdim=True) * cond_mean
        cond_mask = F.pad(cond_mask, (0, n - cond_mask.shape[-1]), value=False)
        seq_len = duration.to(torch.int32) if batch > 1 else None          # cfm.py:336-339

        if y0 is None:
            # cfm.py:430-435.  Drawn with the HOST generator, as the reference's CPU path does (its `self.device` is the CPU
            # there): the same `seed` then gives the same noise, which is what "identical noise seed" parity needs.
            ys = []
            for dur in duration:
                if seed is not None:
                    torch.manual_seed(seed)
                ys.append(torch.randn(int(dur), self.num_channels, dtype=torch.float32))
            y0 = pad_rows(ys, 0)
        assert tuple(y0.shape) == (batch, n, self.num_channels), (tuple(y0.shape), (batch, n, self.num_channels))

        t_start = 0.0
        if duplicate_test:                                                   # cfm.py:438-443: start the solve at t_inter from a blend of
            t_start = t_inter                                                # the noise and the shifted prompt, on a shortened grid
            y0 = (1 - t_start) * y0.to(torch.float32) + t_start * test_cond.to(y0.device)
            steps = int(steps * (1 - t_start))
        t = time_grid(steps, sway_sampling_coef, t_start)
        step_cond = None
        if cond_grl is not None and (pros is not None or no_ref_audio or ref_ratio < 1):
            # the flow is conditioned on cond_grl, not on the (prosody-shifted / replaced) cond (cfm.py:329-330, 387-388);
            # when the two coincide the engine's default is already right
            step_cond = cond_grl[:, :n].contiguous() if cond_seq_len > n else cond_grl
            if step_cond.shape[1] != cond.shape[1]:          # no_ref_audio replaced cond by a full-length tensor
                step_cond = F.pad(step_cond, (0, 0, 0, cond.shape[1] - step_cond.shape[1]), value=0.0)
        # what the engine was handed, for measurement tools that time its prepare() / solve() halves on the same inputs (bench.py)
        self.last_engine_call = dict(cond=cond, cond_mask=cond_mask, text=text, t_grid=t.numpy(), y0=y0, cond_frames=min(cond_seq_len, n),
                                     cfg_strength=float(cfg_strength), seq_len=seq_len, prosody=pros)
        out, y_final, traj = self.engine.sample(
            cond, cond_mask, text, t.numpy(), y0, cond_frames=min(cond_seq_len, n), cfg_strength=float(cfg_strength),   # F.pad above CROPS the
            # prompt when `lens` lets the duration fall below the prompt length (negative pad, as cfm.py:311 does)
            seq_len=seq_len, prosody=pros, want_trajectory=return_trajectory, step_cond=step_cond,
            prosody_text_only=bool(no_ref_audio and pros is not None))
        if no_ref_audio:                                                     # cfm.py:464-466: re-centre the generated part
            gen = out[:, cond_seq_len:, :]
            out[:, cond_seq_len:, :] = gen - (gen.mean(dim=1, keepdim=True) - cond_mean)
        if vocoder is not None:
            out = vocoder(out.permute(0, 2, 1))
        return out, traj
