"""Generate the golden vectors in tests/golden by RUNNING THE REFERENCE (build container only).

    python -m oracle.gen_golden            # needs /root/reference; writes tests/golden/*.npz

Each fixture is data only: the case parameters, the explicit inputs (cond mel, token ids, y0,
masks), the weight seed + checksum (weights are regenerated from ``lemas_tts_amd.synth``), and
the reference's outputs (``out`` and the Euler ``trajectory`` of ``CFM.sample``, cfm.py:206-473).
Nothing of the reference's source travels.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lemas_tts_amd.model.layout import DiTArch  # noqa: E402
from lemas_tts_amd import synth  # noqa: E402
from oracle import ref_shims  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
VOCAB = 898

MINI = DiTArch(depth=2)
FULL = DiTArch()


def _reference_y0(seed: int, durations) -> torch.Tensor:
    """What cfm.py:430-435 draws when ``seed`` is given (re-seeded per sample, zero right-pad)."""
    ys = []
    for d in durations:
        torch.manual_seed(seed)
        ys.append(torch.randn(int(d), 100))
    return torch.nn.utils.rnn.pad_sequence(ys, padding_value=0, batch_first=True)


def run_case(name, arch, *, wseed, B, F, lens, Nt, duration, steps, cfg, coef, noise_seed,
             edit_spans=None, prosody=False, use_acc_grl=False, no_ref_audio=False, ref_ratio=1, pyseed=None, store_traj=True,
             outlier=None, duplicate_test=False, t_inter=0.1):
    if ONLY is not None and name not in ONLY:
        return
    sd_np = synth.synth_cfm_state_dict(arch, VOCAB, wseed, prosody=prosody, outlier=outlier)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    cfm = ref_shims.build_reference_cfm(arch.reference_kwargs(), VOCAB, sd, use_prosody=prosody)

    cond = np.stack([synth.synth_cond_mel(wseed + 1, F, f"cond{b}") for b in range(B)])
    text = np.full((B, max(Nt)), -1, dtype=np.int64)
    for b in range(B):
        text[b, : Nt[b]] = synth.synth_tokens(wseed + 2, Nt[b], VOCAB, f"tok{b}")
    lens_t = None if lens is None else torch.tensor(lens, dtype=torch.long)
    dur_arg = duration if isinstance(duration, int) else torch.tensor(duration, dtype=torch.long)

    edit_mask = None
    if edit_spans is not None:
        from oracle.lemas_oracle import build_edit_mask
        # the mask builder itself (speech_edit_multilingual.py:125-158) is script-level code with no
        # importable function; the golden pins the *sampler's* use of the mask (cfm.py:293-295).
        edit_mask = build_edit_mask(F * 256 - 256 + 100, edit_spans)
        assert edit_mask.shape[1] == F, (edit_mask.shape, F)

    kw = dict(steps=steps, cfg_strength=cfg, sway_sampling_coef=coef, seed=noise_seed,
              edit_mask=edit_mask, use_acc_grl=use_acc_grl, ref_ratio=ref_ratio, lens=lens_t)
    if duplicate_test:              # cfm.py:307-309, 438-443: the solve starts at t_inter from a blend of the noise and the shifted prompt
        kw["duplicate_test"] = True
        kw["t_inter"] = t_inter
    pros = None
    cond_in = torch.from_numpy(cond)
    if prosody:
        # reach cfm.py:248-265,313-318,376-380 without the Pretssel encoder files: raw audio in,
        # mel / resample / fbank / encoder replaced by lookups keyed on the first audio sample.
        import lemas_tts.model.cfm as cfm_mod
        pros = synth.synth_prosody_embed(wseed + 3, B)
        audio = torch.zeros(B, 4000)
        audio[:, 0] = torch.arange(B).float()
        class _MelLookup(torch.nn.Module):
            target_sample_rate = 24000

            def forward(self, wav):
                return torch.from_numpy(cond).permute(0, 2, 1)

        cfm.mel_spec = _MelLookup()
        cfm_mod.torchaudio.functional.resample = lambda a, s, d: a
        cfm_mod.extract_fbank_16k = lambda a: a[:1]
        cfm.prosody_encoder = lambda fb, padding_mask=None: [torch.from_numpy(pros[int(fb.flatten()[0])])]
        cond_in = audio
        kw["use_prosody_encoder"] = True

    cond_noise = None
    if no_ref_audio:
        # cfm.py:321 draws randn_like(cond) from the global RNG before the (re-seeded) y0 draws: pin that draw
        kw["no_ref_audio"] = True
        n_pad = int(max(max(Nt[b], F) + 1 for b in range(B)) if isinstance(duration, int) else max(duration))
        n_pad = max(n_pad, duration if isinstance(duration, int) else 0)
        torch.manual_seed(noise_seed + 7)
        cond_noise = torch.randn(B, n_pad, 100)
        torch.manual_seed(noise_seed + 7)
    if pyseed is not None:          # clip_and_shuffle (cfm.py:39-84) draws from Python's random
        import random as _pyrandom
        _pyrandom.seed(pyseed)
    t0 = time.time()
    out, traj = cfm.sample(cond=cond_in, text=torch.from_numpy(text), duration=dur_arg, **kw)
    dt = time.time() - t0
    if cond_noise is not None:
        assert cond_noise.shape == out.shape, (cond_noise.shape, out.shape)
    N = out.shape[1]
    lens_eff = [F] * B if lens is None else lens
    durs = torch.maximum(torch.maximum(torch.tensor([int((text[b] != -1).sum()) for b in range(B)]),
                                       torch.tensor(lens_eff)) + 1,
                         torch.full((B,), duration) if isinstance(duration, int) else torch.tensor(duration))
    y0 = _reference_y0(noise_seed, durs.tolist())
    if duplicate_test:              # traj[0] is the blend; the fixture stores the NOISE (the sampler's input), like every other case
        tc = torch.nn.functional.pad(torch.from_numpy(cond), (0, 0, F, N - 2 * F))
        assert y0.shape == out.shape and torch.equal((1 - t_inter) * y0 + t_inter * tc, traj[0]), "y0 replication drifted"
    else:
        assert y0.shape == out.shape and torch.equal(y0, traj[0]), "y0 replication drifted"

    if use_acc_grl is False and not prosody and edit_spans is None and B == 1 and not no_ref_audio and ref_ratio >= 1 and not duplicate_test:
        out2, _ = cfm.sample(cond=cond_in, text=torch.from_numpy(text), duration=dur_arg,
                             **{**kw, "use_acc_grl": True})
        assert torch.equal(out, out2), "accent-GRL flag must be a forward no-op at ref_ratio>=1"

    fx = dict(
        arch_depth=arch.depth, vocab=VOCAB, wseed=wseed, wchecksum=synth.checksum(sd_np),
        prosody=int(prosody), B=B, F=F, N=N, steps=steps, cfg=cfg,
        coef=np.float32(np.nan if coef is None else coef),
        cond=cond, text=text, y0=y0.numpy(), out=out.numpy(),
        duration=np.asarray(durs.tolist(), dtype=np.int64),
        lens=np.asarray(lens_eff, dtype=np.int64),
    )
    if not store_traj and B > 1:
        # cfm.py:430-435 re-seeds per sample, so every sample's noise is a PREFIX of the longest sample's draw (zero-padded
        # beyond its duration): keep that one draw (loaders cut and pad it).  The full-size fixtures also blank `out` outside the
        # generated frames (conditioning frames are a copy of `cond`, frames past a sample's duration are never compared).
        j = int(np.argmax(durs.numpy()))
        ok = all(torch.equal(y0[b, : int(durs[b])], y0[j, : int(durs[b])]) and not y0[b, int(durs[b]):].any() for b in range(B))
        if ok:
            fx["y0"] = y0[j: j + 1].numpy()
            fx["y0_shared"] = np.int64(1)
        else:          # ragged durations: the draws are not prefixes of one another; the loader re-draws them from the seed
            del fx["y0"]
            fx["noise_seed"] = np.int64(noise_seed)
        o = out.numpy().copy()
        for b in range(B):
            o[b, : lens_eff[b]] = 0
            o[b, int(durs[b]):] = 0
        fx["out"] = o
        fx["out_generated_only"] = np.int64(1)
    if store_traj:           # the full-size cases keep only `out`: a [33, 1, 1875, 100] trajectory is 25 MB
        fx["trajectory"] = traj.numpy()
    fx["ref_seconds"] = np.float64(dt)
    if edit_mask is not None:
        fx["edit_mask"] = edit_mask.numpy()
    if pros is not None:
        fx["prosody_embeds"] = pros
    if cond_noise is not None:
        fx["cond_noise"] = cond_noise.numpy()
    if use_acc_grl:
        fx["use_acc_grl"] = np.int64(1)
        fx["ref_ratio"] = np.float64(ref_ratio)
    if pyseed is not None:
        fx["pyseed"] = np.int64(pyseed)
    if outlier is not None:
        fx["outlier"] = np.asarray(outlier, dtype=np.float64)
    if duplicate_test:
        fx["duplicate_test"] = np.int64(1)
        fx["t_inter"] = np.float64(t_inter)
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **fx)
    print(f"{name}: N={N} steps={steps} ref {dt:.1f}s |out| mean {out.abs().mean():.4f} "
          f"traj[-1] std {traj[-1].std():.4f}")


def run_prosody_case(name, wseed, frames):
    """ECAPA-TDNN prosody encoder: the reference class on synthetic weights and features.  The
    architecture numbers are ProsodyArch's defaults (pretssel_cfg.json is not in the tree)."""
    ref_shims.install()
    from lemas_tts.model.backbones.prosody_encoder import ECAPA_TDNN
    from lemas_tts_amd.model.layout import ProsodyArch
    arch = ProsodyArch()
    sd = synth.synth_prosody_encoder_state_dict(wseed, arch)
    model = ECAPA_TDNN(**arch.reference_kwargs()).eval()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    fbank = np.stack([synth.synth_fbank(wseed + 1 + b, frames) for b in range(2)])
    with torch.no_grad():
        emb = torch.stack([model(torch.from_numpy(fbank[b: b + 1]), padding_mask=None)[0] for b in range(2)])   # per sample, cfm.py:248-262
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), wseed=wseed, wchecksum=synth.checksum(sd), fbank=fbank, emb=emb.numpy())
    print(f"{name}: frames={frames} emb norm {emb.norm(dim=-1).tolist()}")


def run_full_size_cases(which=None):
    """The BASELINE configurations at FULL size, FULL depth (22 blocks) and FULL NFE against the reference's own CFM.sample
    (cfm.py:206-473, CPU fp32): "mel MSE <= 1e-4 vs reference" measured where it is claimed.  The reference needs 2-10 min of host
    time per case, so each is run once here and its output committed; bench.py and tests/test_gpu_06_configs.py compare against
    them on every run.
      configs0_nfe16        configs[0]: one sentence, 4 s prompt (F = 375), N = 750, NFE 16
      configs1_nfe32        configs[1]: batch 1, 10 s + 10 s (F = 938, N = 1875), NFE 32            <- the headline
      configs2_prosody_b8   configs[2]: multilingual_prosody, batch 8 of mixed lengths (ragged lens / durations), sway, NFE 32
      configs3_share_nfe32  configs[3]: the per-GPU share of the 64-utterance batch, 8 x (4 s + 8 s) (N = 1125), NFE 32
      configs4_edit_nfe48   configs[4]: speech-edit infill of a 30 s source (F = 2813), 3 edit spans, NFE 48, sway 3
      configs0_outlier_nfe32  configs[0]'s shape at NFE 32 on weights with activation outliers (fp8 stress at a production step count)"""
    cases = {
        "configs0_nfe16": dict(arch=FULL, wseed=1234, B=1, F=375, lens=None, Nt=[128], duration=750, steps=16, cfg=2.0, coef=5, noise_seed=1230),
        "configs1_nfe32": dict(arch=FULL, wseed=1234, B=1, F=938, lens=None, Nt=[319], duration=1875, steps=32, cfg=2.0, coef=5, noise_seed=1234),
        "configs2_prosody_b8": dict(arch=FULL, wseed=1235, B=8, F=938, lens=[375, 420, 500, 610, 700, 780, 850, 938],
                                    Nt=[153, 172, 192, 219, 241, 265, 289, 323], duration=[900, 1010, 1130, 1290, 1420, 1560, 1700, 1900],
                                    steps=32, cfg=2.0, coef=5, noise_seed=1232, prosody=True),
        "configs3_share_nfe32": dict(arch=FULL, wseed=1234, B=8, F=375, lens=None, Nt=[191] * 8, duration=1125, steps=32, cfg=2.0, coef=5,
                                     noise_seed=4321),
        "configs4_edit_nfe48": dict(arch=FULL, wseed=1234, B=1, F=2813, lens=None, Nt=[400], duration=2813, steps=48, cfg=2.0, coef=3.0,
                                    noise_seed=1236, edit_spans=[(4.0, 6.5), (12.0, 15.0), (22.0, 24.0)]),
        # round 4: the activation-outlier stress of `full_outlier` (1 % of the residual channels x30 in every block) on a PRODUCTION solve --
        # full depth, NFE 32 -- so that the fp8 path is judged on outliers at the step count it ships with, not on a short solve whose
        # coarse steps integrate the e4m3 flow error undamped with or without outliers (full_plain: 3 steps, full_outlier: 8)
        "configs0_outlier_nfe32": dict(arch=FULL, wseed=24, B=1, F=375, lens=None, Nt=[128], duration=750, steps=32, cfg=2.0, coef=5,
                                       noise_seed=1237, outlier=(0.01, 30.0)),
    }
    for name, kw in cases.items():
        if which and name not in which:
            continue
        arch = kw.pop("arch")
        run_case(name, arch, store_traj=False, **kw)


ONLY = None      # --only name ...: regenerate just these sampler fixtures


def main():
    global ONLY
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(8)
    if "--only" in sys.argv:
        ONLY = set(sys.argv[sys.argv.index("--only") + 1:])
    if "--full-size" in sys.argv:
        run_full_size_cases([a for a in sys.argv[1:] if not a.startswith("--")] or None)
        return
    run_case("mini_plain", MINI, wseed=11, B=1, F=60, lens=None, Nt=[30], duration=160, steps=4,
             cfg=2.0, coef=5, noise_seed=101)
    run_case("mini_nocfg_nosway", MINI, wseed=12, B=1, F=40, lens=None, Nt=[20], duration=96, steps=3,
             cfg=0.0, coef=None, noise_seed=102)
    run_case("mini_batch", MINI, wseed=13, B=2, F=70, lens=[50, 70], Nt=[24, 31], duration=[120, 150],
             steps=4, cfg=2.0, coef=5, noise_seed=103)
    run_case("mini_edit", MINI, wseed=14, B=1, F=200, lens=None, Nt=[40], duration=199, steps=4,
             cfg=2.0, coef=3.0, noise_seed=104, edit_spans=[(0.4, 0.7), (1.3, 1.6)])
    run_case("mini_prosody", DiTArch(depth=2), wseed=15, B=2, F=64, lens=None, Nt=[20, 26],
             duration=[130, 144], steps=3, cfg=2.0, coef=5, noise_seed=105, prosody=True)
    run_case("mini_noref", MINI, wseed=19, B=1, F=50, lens=None, Nt=[22], duration=140, steps=3,
             cfg=2.0, coef=5, noise_seed=107, no_ref_audio=True)
    run_case("mini_grl_prosody", DiTArch(depth=2), wseed=20, B=2, F=48, lens=None, Nt=[18, 25], duration=[120, 131], steps=3,
             cfg=2.0, coef=5, noise_seed=108, prosody=True, use_acc_grl=True)
    # no_ref_audio together with the prosody encoder: the random conditioning OVERWRITES the prosody-shifted mel (cfm.py:313-324), so
    # the prosody embedding acts through the text side only (dit.py:225-233)
    run_case("mini_noref_prosody", DiTArch(depth=2), wseed=22, B=1, F=56, lens=None, Nt=[24], duration=150, steps=3,
             cfg=2.0, coef=5, noise_seed=110, prosody=True, no_ref_audio=True)
    run_case("mini_grl_shuffle", MINI, wseed=21, B=1, F=230, lens=None, Nt=[40], duration=400, steps=3,
             cfg=2.0, coef=5, noise_seed=109, use_acc_grl=True, ref_ratio=0.5, pyseed=4242)
    # the duplicate_test corner (cfm.py:307-309, 438-443): steps = int(10 * 0.75) = 7 on a grid that starts at 0.25; batch of 2 with ragged
    # durations so that the shifted prompt is cropped for neither and zero-padded differently for each
    run_case("mini_duplicate", MINI, wseed=25, B=2, F=40, lens=None, Nt=[18, 22], duration=[110, 96], steps=10,
             cfg=2.0, coef=5, noise_seed=112, duplicate_test=True, t_inter=0.25)
    if ONLY is None:
        run_edit_mask_cases()
        run_prosody_case("prosody_enc_short", 17, 41)
        run_prosody_case("prosody_enc_10s", 18, 998)
    # activation-outlier stress for the fp8 path (synth.synth_cfm_state_dict outlier=): 1 % of the residual channels x30, all 22 blocks
    run_case("full_outlier", FULL, wseed=23, B=1, F=150, lens=None, Nt=[60], duration=400, steps=8,
             cfg=2.0, coef=5, noise_seed=111, outlier=(0.01, 30.0), store_traj=False)
    run_case("full_plain", FULL, wseed=16, B=1, F=150, lens=None, Nt=[60], duration=400, steps=3,
             cfg=2.0, coef=5, noise_seed=106)


def run_edit_mask_cases():
    """Edit-mask builder of the speech-edit entry point (scripts/speech_edit_multilingual.py:125-161, SURVEY.md 8a row a-E):
    the reference's own ``gen_wav_multilingual`` is run with a stand-in ``tts`` whose sampler records what it is handed."""
    import types
    ref_shims.install()
    api_stub = types.ModuleType("lemas_tts.api")
    api_stub.TTS = object
    sys.modules.setdefault("lemas_tts.api", api_stub)
    from lemas_tts.scripts.speech_edit_multilingual import gen_wav_multilingual

    class _Model:
        mel_spec = types.SimpleNamespace(target_sample_rate=24000, hop_length=256)

        def sample(self, cond, text, duration, **kw):
            self.got = dict(edit_mask=kw["edit_mask"].clone(), duration=duration, text=text, nw=cond.shape[-1], kw={k: v for k, v in kw.items() if k != "edit_mask"})
            return torch.zeros(1, duration + 1, 100), None

    class _Voc:
        def decode(self, mel):
            return torch.zeros(1, 256 * (mel.shape[-1] - 1))

    cases = {
        "one_span": (72000, [(1.0, 1.5)]),
        "three_spans_30s": (720000, [(4.0, 6.5), (12.0, 15.0), (22.0, 24.0)]),
        "span_at_start": (48000, [(0.0, 0.4)]),
        "span_to_the_end": (50000, [(1.7, 2.0833)]),
        "touching_margins": (96000, [(1.0, 1.2), (1.35, 1.6)]),
        "odd_length": (61111, [(0.33, 0.77), (1.9, 2.2)]),
    }
    out = {}
    for name, (nw, spans) in cases.items():
        tts = types.SimpleNamespace(device="cpu", ema_model=_Model(), vocoder=_Voc(), frontend=None, mel_spec_type="vocos")
        g = torch.Generator().manual_seed(nw)
        audio = torch.randn(nw, generator=g) * 0.3          # rms > 0.1: no rescale
        gen_wav_multilingual(tts, audio, 24000, "ab", spans, nfe_step=2, cfg_strength=2.0, sway_sampling_coef=3.0, seed=0)
        got = tts.ema_model.got
        out[name + "/nw"] = np.int64(nw)
        out[name + "/spans"] = np.asarray(spans, dtype=np.float64)
        out[name + "/edit_mask"] = got["edit_mask"].numpy()
        out[name + "/duration"] = np.int64(got["duration"])
        print(f"edit mask {name}: frames {got['edit_mask'].shape[-1]} regenerated {int((~got['edit_mask']).sum())} duration {got['duration']} text {got['text']}")
    np.savez_compressed(os.path.join(GOLDEN, "edit_masks.npz"), **out)


if __name__ == "__main__":
    main()
