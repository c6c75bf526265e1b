"""GPU tier: the BASELINE.json configurations as parity cases (bench.py only times configs[1]).

Shapes follow SURVEY.md section 8d.  Depth is reduced to 2 where the fp32 oracle would otherwise take minutes on the
host (the 22 blocks are identical code; full depth is covered at full size by the one-step test below and at small
size by the reference golden vector ``full_plain``).  Tolerance: mel-MSE <= 1e-4 over generated frames.
"""
import numpy as np
import pytest
import torch

from lemas_tts_amd import synth
from lemas_tts_amd.model.layout import DiTArch

pytestmark = pytest.mark.gpu
VOCAB = 898


def _mse_generated(out, ref, lens, durs):
    se, cnt = 0.0, 0
    for b in range(out.shape[0]):
        d = (out[b, lens[b]:durs[b]] - ref[b, lens[b]:durs[b]]).double()
        se += float((d ** 2).sum())
        cnt += d.numel()
    return se / cnt


def _inputs(seed, B, Fs, Ns, nts):
    Fm, Nm, Tm = max(Fs), max(Ns), max(nts)
    cond = torch.zeros(B, Fm, 100)
    text = torch.full((B, Tm), -1, dtype=torch.long)
    y0 = torch.zeros(B, Nm, 100)
    for b in range(B):
        cond[b, : Fs[b]] = torch.from_numpy(synth.synth_cond_mel(seed + b, Fs[b]))
        text[b, : nts[b]] = torch.from_numpy(synth.synth_tokens(seed + b, nts[b], VOCAB))
        y0[b, : Ns[b]] = torch.from_numpy(synth.synth_noise(seed + b, Ns[b]))
    return cond, text, y0


def test_config3_batch8_mixed_lengths_prosody_sway():
    """configs[2]: multilingual_prosody, batch 8 of mixed lengths (mask on), prosody embedding given, NFE steps, sway."""
    from lemas_tts_amd.model.cfm import CFM
    from oracle import lemas_oracle as O
    arch = DiTArch(depth=2)
    sd = synth.synth_cfm_state_dict(arch, VOCAB, 51, prosody=True)
    B = 8
    Fs = [375, 420, 500, 610, 700, 780, 850, 938]
    Ns = [900, 1010, 1130, 1290, 1420, 1560, 1700, 1900]
    nts = [round(n * 0.17) for n in Ns]
    cond, text, y0 = _inputs(52, B, Fs, Ns, nts)
    pros = torch.from_numpy(synth.synth_prosody_embed(53, B))
    lens, dur = torch.tensor(Fs), torch.tensor(Ns)
    steps = 3   # the oracle needs ~10 s per Euler step here; step count only repeats the same loop body
    m = CFM(arch, VOCAB, sd, device="cuda:0", use_prosody_encoder=True)
    out, _ = m.sample(cond, text, dur, lens=lens, steps=steps, cfg_strength=2.0, sway_sampling_coef=5, y0=y0,
                      use_acc_grl=False, prosody_embeds=pros)
    ref, _ = O.OracleCFM(sd, arch).sample(cond, text, dur, y0=y0, lens=lens, steps=steps, cfg_strength=2.0,
                                          sway_sampling_coef=5, prosody_embeds=pros)
    mse = _mse_generated(out.cpu(), ref, Fs, Ns)
    print(f"\n[config3 B=8 mixed] mel-MSE {mse:.3e}")
    assert mse <= 1e-4


def test_config4_batch8_equal_length():
    """configs[3] per-GPU share: 8 utterances of 4 s reference + 8 s generated (F=375, N=1125), no mask needed."""
    from lemas_tts_amd.model.cfm import CFM
    from oracle import lemas_oracle as O
    arch = DiTArch(depth=2)
    sd = synth.synth_cfm_state_dict(arch, VOCAB, 61)
    B, F_, N = 8, 375, 1125
    cond, text, y0 = _inputs(62, B, [F_] * B, [N] * B, [round(N * 0.17)] * B)
    m = CFM(arch, VOCAB, sd, device="cuda:0")
    out, _ = m.sample(cond, text, N, steps=2, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False)
    ref, _ = O.OracleCFM(sd, arch).sample(cond, text, N, y0=y0, steps=2, cfg_strength=2.0, sway_sampling_coef=5)
    mse = _mse_generated(out.cpu(), ref, [F_] * B, [N] * B)
    print(f"\n[config4 B=8 N=1125] mel-MSE {mse:.3e}")
    assert mse <= 1e-4
    # the data-parallel split relies on this: each utterance alone gives the same bits as inside the batch of 8
    one, _ = m.sample(cond[3:4], text[3:4], N, steps=2, cfg_strength=2.0, sway_sampling_coef=5, y0=y0[3:4], use_acc_grl=False)
    np.testing.assert_array_equal(one[0].cpu().numpy(), out[3].cpu().numpy())


def test_config5_speech_edit_30s_three_spans():
    """configs[4]: 30 s source, 3 edit spans, the LAST two steps of the NFE-48 grid (the first steps of a sway-warped
    grid have dt ~ 1e-7 and would make any comparison vacuous), whole utterance vocoded.  bf16 operands here; the fp8-weights
    variant of the same case is tests/test_gpu_02_fp8.py::test_config5_fp8_speech_edit_30s_three_spans."""
    from lemas_tts_amd.engine import VocosEngine
    from lemas_tts_amd.model.cfm import CFM
    from oracle import lemas_oracle as O
    arch = DiTArch(depth=2)
    sd = synth.synth_cfm_state_dict(arch, VOCAB, 71)
    nw = 720000
    F_ = nw // 256 + 1                      # 2813 mel frames
    edit = O.build_edit_mask(nw, [(4.0, 6.5), (12.0, 15.0), (22.0, 24.0)])
    assert edit.shape == (1, F_)
    cond, text, _ = _inputs(72, 1, [F_], [F_ + 1], [400])
    dur = nw // 256                          # -> bumped to F + 1 = 2814 by the sampler (cfm.py:300-302)
    y0 = torch.from_numpy(synth.synth_noise(73, F_ + 1))[None]
    tg = O.time_grid(48, 3.0)[-3:]
    m = CFM(arch, VOCAB, sd, device="cuda:0")
    # two steps of the 48-step grid: drive the engine with the explicit grid through its lower-level entry
    cmask = torch.nn.functional.pad(edit, (0, 1), value=False)
    cpad = torch.nn.functional.pad(cond, (0, 0, 0, 1))
    out, y, _ = m.engine.sample(cpad, cmask, text, tg.numpy(), y0, cond_frames=F_, cfg_strength=2.0)
    ref, _ = O.OracleCFM(sd, arch).sample(cond, text, dur, y0=y0, steps=2, cfg_strength=2.0, edit_mask=edit, t_grid=tg)
    keep = ~cmask[0]
    mse = float(((out.cpu()[0, keep] - ref[0, keep]).double() ** 2).mean())
    print(f"\n[config5 edit N={F_ + 1}] mel-MSE over regenerated frames {mse:.3e} ({int(keep.sum())} frames)")
    assert mse <= 1e-4
    np.testing.assert_array_equal(out.cpu().numpy()[0, ~keep.numpy()], cpad.numpy()[0, ~keep.numpy()])   # kept frames are copied
    vsd = synth.synth_vocos_state_dict(74)
    wav = VocosEngine(vsd, device="cuda:0").decode(out.permute(0, 2, 1)).cpu()
    wref = O.OracleVocos(vsd).decode(out.cpu().permute(0, 2, 1))
    assert wav.shape == (1, 256 * F_)
    assert (wav - wref).abs().max().item() < 1e-4 * max(1.0, wref.abs().max().item())


def test_config2_full_size_last_two_euler_steps():
    """configs[1] at FULL size (22 blocks, F=938, N=1875, CFG on): the last two Euler steps of the NFE-32 grid (dt = 0.12
    and 0.13; the first steps of the warped grid have dt ~ 1e-7) and one flow evaluation at t = 0 against the oracle."""
    from lemas_tts_amd.model.cfm import CFM
    from oracle import lemas_oracle as O
    arch = DiTArch()
    sd = synth.synth_cfm_state_dict(arch, VOCAB, 1234)
    F_, N = 938, 1875
    cond, text, y0 = _inputs(1234, 1, [F_], [N], [round(N * 0.17)])
    grid = O.time_grid(32, 5)
    tg = grid[-3:]
    m = CFM(arch, VOCAB, sd, device="cuda:0")
    cmask = torch.zeros(1, N, dtype=torch.bool)
    cmask[:, :F_] = True
    cpad = torch.nn.functional.pad(cond, (0, 0, 0, N - F_))
    out, y, _ = m.engine.sample(cpad, cmask, text, tg.numpy(), y0, cond_frames=F_, cfg_strength=2.0)
    torch.set_num_threads(16)
    oc = O.OracleCFM(sd, arch)
    ref, _ = oc.sample(cond, text, N, y0=y0, steps=2, cfg_strength=2.0, t_grid=tg)
    mse = float(((out.cpu()[0, F_:] - ref[0, F_:]).double() ** 2).mean())
    # flow at t = 0 (full CFG amplification, 1 + 2 = 3x): relative rms error of pred + (pred - null) * cfg
    m.engine.prepare(cpad, cmask, text, grid.numpy(), cond_frames=F_, cfg_strength=2.0)
    pred = m.engine.forward(y0, 0).cpu()
    sc = torch.where(cmask[..., None], cpad, torch.zeros_like(cpad))
    rc = oc.dit.forward(y0, sc, text, grid[0], False, False, None, False)
    ru = oc.dit.forward(y0, sc, text, grid[0], True, True, None, False)
    fh, fr = pred[0] + (pred[0] - pred[1]) * 2.0, rc[0] + (rc[0] - ru[0]) * 2.0
    rel = float((fh - fr).pow(2).mean().sqrt() / fr.pow(2).mean().sqrt())
    print(f"\n[config2 full size] last-2-steps mel-MSE {mse:.3e}; CFG flow relative rms error at t=0 {rel:.3e}")
    assert mse <= 1e-4
    assert rel < 1e-2


def test_config1_plumbing_sentence_nfe16():
    """configs[0]: one pre-phonemised English sentence through the token-list path, 4 s prompt
    (F = 375), N = 750, NFE = 16, cfg 2, coef 5, vocoder included.  The reference runs this case on its CPU path; the build
    has no CPU path (tier rule), so it is a GPU parity case against the oracle like the others."""
    from lemas_tts_amd.engine import VocosEngine
    from lemas_tts_amd.model.cfm import CFM
    from oracle import lemas_oracle as O
    phones = ("(en) ð ə _ k w ɪ k _ b ɹ aʊ n _ f ɑ k s _ dʒ ʌ m p s _ oʊ v ɚ _ ð ə _ l eɪ z i _ d ɔ ɡ .").split()
    ref_phones = "(en) h ə l oʊ _ w ɜ l d .".split()
    symbols = sorted(set(phones) | set(ref_phones))
    vocab = {s: i for i, s in enumerate(symbols)}
    arch = DiTArch(depth=2)
    sd = synth.synth_cfm_state_dict(arch, len(vocab), 131)
    vsd = synth.synth_vocos_state_dict(132)
    F_, N, S = 375, 750, 16
    cond = torch.from_numpy(synth.synth_cond_mel(133, F_))[None]
    y0 = torch.from_numpy(synth.synth_noise(134, N))[None]
    m = CFM(arch, len(vocab), sd, vocab_char_map=vocab, device="cuda:0")
    out, _ = m.sample(cond, [ref_phones + phones + ["<unk>"]], N, steps=S, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False)
    text = O.tokens_to_idx([ref_phones + phones + ["<unk>"]], vocab)          # unknown token -> 0 (model/utils.py:87-94)
    assert int(text[0, -1]) == 0
    ref, _ = O.OracleCFM(sd, arch).sample(cond, text, N, y0=y0, steps=S, cfg_strength=2.0, sway_sampling_coef=5)
    mse = _mse_generated(out.cpu(), ref, [F_], [N])
    print(f"\n[config1 N=750 NFE=16] mel-MSE {mse:.3e}")
    assert mse <= 1e-4
    mel = out[:, F_ - 1:, :].permute(0, 2, 1)
    wav = VocosEngine(vsd, device="cuda:0").decode(mel).cpu()
    wref = O.OracleVocos(vsd).decode(mel.cpu())
    assert wav.shape == (1, 256 * (N - F_))
    assert (wav - wref).abs().max().item() < 1e-4 * max(1.0, wref.abs().max().item())


FULL_SIZE = ["configs0_nfe16", "configs1_nfe32", "configs2_prosody_b8", "configs3_share_nfe32", "configs4_edit_nfe48"]


@pytest.mark.parametrize("name", FULL_SIZE)
def test_baseline_config_full_size_full_nfe_vs_the_reference(golden_dir, name):
    """THE parity numbers: every BASELINE configuration at FULL size, FULL depth (22 blocks) and FULL NFE, bf16 MFMA operands,
    against the output of the REFERENCE's own CFM.sample (cfm.py:206-473, CPU fp32) on identical inputs -- fixtures made once by
    `python -m oracle.gen_golden --full-size` (the reference needs 20 s ... 10 min per case on the build box).  Target: mel-MSE
    <= 1e-4 over the generated frames (BASELINE.json).  configs[1] and configs[4] are also run with fp8 MFMA weights (configs[4]'s
    precision).
      configs0_nfe16        one sentence, F = 375, N = 750, NFE 16
      configs1_nfe32        batch 1, 10 s + 10 s (F = 938, N = 1875), NFE 32: what bench.py times
      configs2_prosody_b8   batch 8 of mixed lengths (ragged lens / durations), prosody conditioning, sway, NFE 32
      configs3_share_nfe32  configs[3]'s per-GPU share: 8 x (4 s + 8 s), N = 1125, NFE 32 (256x256 GEMM tiles, BH = 128 attention)
      configs4_edit_nfe48   speech-edit infill of a 30 s source (N = 2814), 3 edit spans, NFE 48, sway 3"""
    import os
    import test_gpu_00_sample as T
    if not os.path.exists(os.path.join(golden_dir, name + ".npz")):
        pytest.fail(f"{name}.npz is missing: run `python -m oracle.gen_golden --full-size {name}` in the build container (needs /root/reference)")
    fx, arch, sd = T._load(golden_dir, name)
    fx = synth.expand_reference_fixture(fx)
    assert arch.depth == 22
    out, _ = T._run_case(fx, arch, sd, graph=True, traj=False)
    mse = T._gen_mse(out, fx["out"], fx)
    B = int(fx["B"])
    print(f"\n[{name}: 22 blocks, B={B}, N={out.shape[1]}, NFE={int(fx['steps'])}] mel-MSE vs the reference {mse:.3e} "
          f"(the reference's CPU run took {float(fx['ref_seconds']):.0f} s)")
    assert mse <= 1e-4, mse
    if "edit_mask" in fx:      # kept frames are copied from the source mel, exactly (cfm.py:459-461)
        em = np.pad(fx["edit_mask"][0], (0, out.shape[1] - fx["edit_mask"].shape[1]))
        np.testing.assert_array_equal(out[0, em][:, :], np.pad(fx["cond"][0], ((0, out.shape[1] - fx["cond"].shape[1]), (0, 0)))[em])
    elif "prosody_embeds" not in fx:
        for b in range(B):
            L = int(fx["lens"][b])
            np.testing.assert_array_equal(out[b, :L], fx["cond"][b, :L])
    if name in ("configs1_nfe32", "configs4_edit_nfe48"):
        m = T._model(arch, int(fx["vocab"]), int(fx["wseed"]), bool(fx["prosody"]), sd)
        m.engine.set_option("fp8", 1)
        try:
            out8, _ = T._run_case(fx, arch, sd, graph=True, traj=False)
        finally:
            m.engine.set_option("fp8", 0)
        mse8 = T._gen_mse(out8, fx["out"], fx)
        print(f"[{name}, fp8 GEMM operands] mel-MSE vs the reference {mse8:.3e}")
        assert mse8 <= 1e-4, mse8


def test_config4_full_form_64_utterances_over_8_ranks_on_one_gpu():
    """BASELINE configs[3] in its FULL form -- 64 utterances dealt to 8 ranks (lemas_tts_amd.parallel.shard_utterances, longest first),
    each rank running its 8 as one CFM.sample batch -- with the eight ranks played one after the other on the one GPU of the box
    (full depth, full NFE; the 8-GPU run itself is the driver's).  Checks the deal (every utterance exactly once, 8 per rank) and
    that what a rank computes for an utterance inside its batch is bit-for-bit what that utterance gives alone: with configs[3]'s
    equal-length utterances the result of the 64 does not depend on how they were dealt.  (Utterances of DIFFERENT lengths in one batch
    follow the reference's B > 1 semantics -- filler-token text embeddings beyond an utterance's own end reach its last frames through
    the text ConvNeXt blocks -- so there a batch-mate changes the result, in the reference as here; test_config3 covers that form
    against the oracle's batch.)"""
    from lemas_tts_amd.model.cfm import CFM
    from lemas_tts_amd.parallel import shard_utterances
    arch = DiTArch(depth=22)
    sd = synth.synth_cfm_state_dict(arch, VOCAB, 1234)
    m = CFM(arch, VOCAB, sd, device="cuda:0")
    rng = np.random.default_rng(64)
    U, world = 64, 8
    Fs, Ns = [375] * U, [1125] * U                                   # configs[3]: 4 s prompt + 8 s generated, every utterance
    nts = [int(v) for v in rng.integers(150, 192, U)]                # token counts differ (padded with fillers up to N either way)
    shards = shard_utterances(Ns, world)
    assert sorted(i for sh in shards for i in sh) == list(range(U)) and all(len(sh) == U // world for sh in shards)
    results = {}
    for rank in range(world):
        idx = shards[rank]
        cond = torch.zeros(len(idx), max(Fs[i] for i in idx), 100)
        text = torch.full((len(idx), max(nts[i] for i in idx)), -1, dtype=torch.long)
        y0 = torch.zeros(len(idx), max(Ns[i] for i in idx), 100)
        for b, i in enumerate(idx):
            cond[b, : Fs[i]] = torch.from_numpy(synth.synth_cond_mel(1000 + i, Fs[i]))
            text[b, : nts[i]] = torch.from_numpy(synth.synth_tokens(1000 + i, nts[i], VOCAB))
            y0[b, : Ns[i]] = torch.from_numpy(synth.synth_noise(1000 + i, Ns[i]))
        out, _ = m.sample(cond, text, torch.tensor([Ns[i] for i in idx]), lens=torch.tensor([Fs[i] for i in idx]), steps=32, cfg_strength=2.0,
                          sway_sampling_coef=5, y0=y0, use_acc_grl=False)
        for b, i in enumerate(idx):
            results[i] = out[b, : Ns[i]].cpu().numpy()
    assert len(results) == U and all(np.isfinite(v).all() for v in results.values())
    for i in (shards[0][0], shards[3][5], shards[7][7]):
        cond1, text1, y01 = _inputs(1000 + i, 1, [Fs[i]], [Ns[i]], [nts[i]])
        one, _ = m.sample(cond1, text1, Ns[i], steps=32, cfg_strength=2.0, sway_sampling_coef=5, y0=y01, use_acc_grl=False)
        np.testing.assert_array_equal(one[0, : Ns[i]].cpu().numpy(), results[i], err_msg=f"utterance {i}")
