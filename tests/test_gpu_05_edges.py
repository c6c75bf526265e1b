"""GPU tier: edge cases of the sampler the reference's code paths imply (a-S duration rules, a-X text
truncation / padding, a-Q masks) -- tiny, ragged, maximum-size -- against the fp32 oracle.  Tolerance mel-MSE <= 1e-4."""
import numpy as np
import pytest
import torch

from lemas_tts_amd import synth
from lemas_tts_amd.model.layout import DiTArch

pytestmark = pytest.mark.gpu
VOCAB = 898
_cache = {}


def _pair(depth=2, seed=101, prosody=False):
    from lemas_tts_amd.model.cfm import CFM
    from oracle import lemas_oracle as O
    key = (depth, seed, prosody)
    if key not in _cache:
        _cache.clear()
        arch = DiTArch(depth=depth)
        sd = synth.synth_cfm_state_dict(arch, VOCAB, seed, prosody=prosody)
        _cache[key] = (CFM(arch, VOCAB, sd, device="cuda:0", use_prosody_encoder=prosody), O.OracleCFM(sd, arch))
    return _cache[key]


def _mse(out, ref, lens, durs):
    se = cnt = 0.0
    for b in range(out.shape[0]):
        d = (out[b, lens[b]:durs[b]] - ref[b, lens[b]:durs[b]]).double()
        se += float((d ** 2).sum()); cnt += d.numel()
    return se / max(cnt, 1)


@pytest.mark.parametrize("F_,N,nt", [(1, 2, 1), (3, 5, 2), (7, 64, 30), (63, 65, 9), (100, 129, 140), (127, 128, 20), (128, 257, 64)])
def test_tiny_and_tile_boundary_lengths(F_, N, nt):
    """lengths around the 64 / 128 tile sizes, 1-frame prompts, more tokens than requested frames (duration is raised to
    tokens + 1, cfm.py:300-302)"""
    m, o = _pair()
    cond = torch.from_numpy(synth.synth_cond_mel(F_ + N, F_))[None]
    text = torch.from_numpy(synth.synth_tokens(N, nt, VOCAB))[None]
    y0 = torch.from_numpy(synth.synth_noise(N + 1, max(N, nt + 1, F_ + 1)))[None]
    kw = dict(steps=4, cfg_strength=2.0, sway_sampling_coef=5)
    out, _ = m.sample(cond, text, N, y0=y0, use_acc_grl=False, **kw)
    ref, _ = o.sample(cond, text, N, y0=y0, **kw)
    assert out.shape == ref.shape
    mse = _mse(out.cpu(), ref, [F_], [ref.shape[1]])
    assert mse <= 1e-4, mse
    np.testing.assert_array_equal(out.cpu().numpy()[0, :F_], cond.numpy()[0])          # conditioning frames are copied


def test_duration_is_raised_to_text_and_prompt_length():
    """cfm.py:297-305: duration < max(tokens, lens) + 1 is raised; the output has the raised length"""
    m, o = _pair()
    F_, nt = 40, 70
    cond = torch.from_numpy(synth.synth_cond_mel(7, F_))[None]
    text = torch.from_numpy(synth.synth_tokens(8, nt, VOCAB))[None]
    y0 = torch.from_numpy(synth.synth_noise(9, nt + 1))[None]
    out, _ = m.sample(cond, text, 10, y0=y0, steps=2, cfg_strength=2.0, sway_sampling_coef=5, use_acc_grl=False)
    ref, _ = o.sample(cond, text, 10, y0=y0, steps=2, cfg_strength=2.0, sway_sampling_coef=5)
    assert out.shape == ref.shape == (1, nt + 1, 100)
    assert _mse(out.cpu(), ref, [F_], [nt + 1]) <= 1e-4


def test_maximum_duration_4096_frames():
    """max_duration clamp (cfm.py:305): a 4096-frame utterance (43.7 s), the largest the sampler accepts"""
    m, o = _pair()
    F_, N = 1500, 5000                      # asks for 5000, gets 4096
    cond = torch.from_numpy(synth.synth_cond_mel(11, F_))[None]
    text = torch.from_numpy(synth.synth_tokens(12, 700, VOCAB))[None]
    y0 = torch.from_numpy(synth.synth_noise(13, 4096))[None]
    from oracle import lemas_oracle as O
    tg = O.time_grid(32, 5)[-2:]            # one (large) Euler step of the NFE-32 grid
    cm = torch.zeros(1, 4096, dtype=torch.bool); cm[:, :F_] = True
    cpad = torch.nn.functional.pad(cond, (0, 0, 0, 4096 - F_))
    out, _, _ = m.engine.sample(cpad, cm, text, tg.numpy(), y0, cond_frames=F_, cfg_strength=2.0)
    ref, _ = o.sample(cond, text, N, y0=y0, steps=1, cfg_strength=2.0, t_grid=tg)
    assert ref.shape == (1, 4096, 100) and out.shape == ref.shape
    mse = _mse(out.cpu(), ref, [F_], [4096])
    print(f"\n[N=4096] mel-MSE {mse:.3e}")
    assert mse <= 1e-4
    # and through the mirrored sampler: the clamp itself
    out2, _ = m.sample(cond, text, N, y0=y0, steps=1, cfg_strength=2.0, sway_sampling_coef=5, use_acc_grl=False)
    assert out2.shape == (1, 4096, 100)


def test_ragged_batch_with_one_frame_generation_and_full_mask_rows():
    """batch of 5 with very different lengths: one sample generates a single frame, one fills the whole batch length"""
    m, o = _pair()
    Fs, Ns = [10, 30, 64, 5, 90], [11, 200, 65, 131, 260]
    nts = [4, 60, 10, 40, 33]
    B = len(Fs)
    cond = torch.zeros(B, max(Fs), 100); text = torch.full((B, max(nts)), -1, dtype=torch.long); y0 = torch.zeros(B, max(Ns), 100)
    for b in range(B):
        cond[b, :Fs[b]] = torch.from_numpy(synth.synth_cond_mel(200 + b, Fs[b]))
        text[b, :nts[b]] = torch.from_numpy(synth.synth_tokens(210 + b, nts[b], VOCAB))
        y0[b, :Ns[b]] = torch.from_numpy(synth.synth_noise(220 + b, Ns[b]))
    lens, dur = torch.tensor(Fs), torch.tensor(Ns)
    kw = dict(steps=3, cfg_strength=2.0, sway_sampling_coef=5)
    out, _ = m.sample(cond, text, dur, lens=lens, y0=y0, use_acc_grl=False, **kw)
    ref, _ = o.sample(cond, text, dur, y0=y0, lens=lens, **kw)
    assert _mse(out.cpu(), ref, Fs, Ns) <= 1e-4


@pytest.mark.parametrize("dual,big", [(1, False), (0, False), (1, True)])
def test_ragged_batch_with_padding_blocks_left_uncomputed(dual, big):
    """Option skip_dead (off by default): the 128-row blocks that lie wholly in a sample's padding are skipped by every kernel of the block
    chain.  Lengths straddle the 128-row blocks in every way (1, 2, 3 live blocks of 4; one sample fills the batch; one ends exactly on a
    block edge).  Against the fp32 oracle the generated frames stay inside 1e-4; against the default path every valid frame more than 30
    before its sample's end (the reach of the reference's unmasked position-embedding conv into the padding, modules.py:167-190) agrees to
    bf16 noise, and nothing in the output -- padding rows included -- is non-finite."""
    m, o = _pair()
    # pitch 640 = five 128-row blocks: the 256-row tiles straddle samples and live / dead blocks
    Fs, Ns = [10, 40, 64, 5, 90, 100], [11, 200, 256, 131, 600, 385]
    nts = [4, 60, 10, 40, 33, 120]
    if big:        # 12 x 640 rows per lane: the 256 x 256 tile for the N = 2048 GEMMs (gemm_bf16.hip pick_tile)
        Fs, Ns, nts = Fs * 2, Ns + [129, 513, 70, 640, 300, 257], nts * 2
    B = len(Fs)
    cond = torch.zeros(B, max(Fs), 100); text = torch.full((B, max(nts)), -1, dtype=torch.long); y0 = torch.zeros(B, max(Ns), 100)
    for b in range(B):
        cond[b, :Fs[b]] = torch.from_numpy(synth.synth_cond_mel(300 + b, Fs[b]))
        text[b, :nts[b]] = torch.from_numpy(synth.synth_tokens(310 + b, nts[b], VOCAB))
        y0[b, :Ns[b]] = torch.from_numpy(synth.synth_noise(320 + b, Ns[b]))
    lens, dur = torch.tensor(Fs), torch.tensor(Ns)
    kw = dict(steps=4, cfg_strength=2.0, sway_sampling_coef=5)
    ref, _ = o.sample(cond, text, dur, y0=y0, lens=lens, **kw)
    outs = {}
    try:
        m.engine.set_option("dual", dual)
        for skip in (0, 1, 2):
            m.engine.set_option("skip_dead", skip)
            for graph in (0, 1):
                m.engine.set_option("graph", graph)
                outs[(skip, graph)] = m.sample(cond, text, dur, lens=lens, y0=y0, use_acc_grl=False, **kw)[0].cpu()
    finally:
        m.engine.set_option("skip_dead", 0); m.engine.set_option("graph", 1); m.engine.set_option("dual", 1)
    m.engine.check_health()
    # the attention half of a block skips padding blocks by default (exact: the reference zeroes that half's output there): not a bit of the
    # output, padding rows included, may differ from computing them
    try:
        m.engine.set_option("dual", dual); m.engine.set_option("skip_masked", 0)
        full = m.sample(cond, text, dur, lens=lens, y0=y0, use_acc_grl=False, **kw)[0].cpu()
    finally:
        m.engine.set_option("skip_masked", 1); m.engine.set_option("dual", 1)
    assert torch.equal(full, outs[(0, 0)])
    for skip in (0, 1, 2):
        assert torch.equal(outs[(skip, 0)], outs[(skip, 1)])                # eager == replayed graph, with and without the switch
        assert torch.isfinite(outs[(skip, 0)]).all()
        assert _mse(outs[(skip, 0)], ref, Fs, Ns) <= 1e-4
    for b in range(B):
        far = max(Fs[b], Ns[b] - 30)
        if far > Fs[b]:
            d = (outs[(1, 0)][b, Fs[b]:far] - outs[(0, 0)][b, Fs[b]:far]).abs().max().item()
            assert d < 5e-2, (b, d)
        assert torch.equal(outs[(1, 0)][b, :Fs[b]], outs[(0, 0)][b, :Fs[b]])     # the prompt frames are the prompt, either way


def test_no_cfg_and_no_sway_paths():
    """cfg_strength < 1e-5 skips the unconditional branch (cfm.py:404-405); sway None uses the plain power warp (:452-453)"""
    m, o = _pair()
    F_, N = 50, 170
    cond = torch.from_numpy(synth.synth_cond_mel(31, F_))[None]
    text = torch.from_numpy(synth.synth_tokens(32, 25, VOCAB))[None]
    y0 = torch.from_numpy(synth.synth_noise(33, N))[None]
    for cfg, coef in ((0.0, 5), (2.0, None), (0.0, None), (3.5, 0.5)):
        out, _ = m.sample(cond, text, N, y0=y0, steps=3, cfg_strength=cfg, sway_sampling_coef=coef, use_acc_grl=False)
        ref, _ = o.sample(cond, text, N, y0=y0, steps=3, cfg_strength=cfg, sway_sampling_coef=coef)
        assert _mse(out.cpu(), ref, [F_], [N]) <= 1e-4, (cfg, coef)
    # coef = -1 (infer_batch_process's own default!) warps the grid to t ** 0 = 1 everywhere: torchdiffeq refuses such a grid, and
    # so do the oracle's restatement and the engine
    from lemas_tts_amd._lib import LemasError
    with pytest.raises(AssertionError):
        o.sample(cond, text, N, y0=y0, steps=3, cfg_strength=2.0, sway_sampling_coef=-1.0)
    with pytest.raises(LemasError, match="monotone"):
        m.sample(cond, text, N, y0=y0, steps=3, cfg_strength=2.0, sway_sampling_coef=-1.0, use_acc_grl=False)


def test_shape_churn_reuses_one_engine_correctly():
    """one engine, a sequence of different shapes / batch sizes / option flips (graph cache keyed by shape, workspaces that
    only grow, AdaLN-table cache keyed by the t-grid): every result must equal what a fresh run of that case gives"""
    m, o = _pair()

    def case(B, F_, N, nt, steps, seed):
        cond = torch.stack([torch.from_numpy(synth.synth_cond_mel(seed + b, F_)) for b in range(B)])
        text = torch.stack([torch.from_numpy(synth.synth_tokens(seed + 10 + b, nt, VOCAB)) for b in range(B)])
        y0 = torch.stack([torch.from_numpy(synth.synth_noise(seed + 20 + b, N)) for b in range(B)])
        return cond, text, y0, dict(steps=steps, cfg_strength=2.0, sway_sampling_coef=5)

    plan = [(1, 40, 200, 30, 3, 300), (2, 64, 333, 50, 2, 310), (1, 40, 200, 30, 3, 300), (4, 30, 129, 20, 3, 320), (1, 100, 640, 90, 2, 330),
            (1, 40, 200, 30, 4, 300), (1, 40, 200, 30, 3, 300)]
    first = {}
    for i, (B, F_, N, nt, steps, seed) in enumerate(plan):
        if i == 3:
            m.engine.set_option("dual", 0)
        if i == 4:
            m.engine.set_option("graph", 0)
        if i == 5:
            m.engine.set_option("dual", 1); m.engine.set_option("graph", 1)
        cond, text, y0, kw = case(B, F_, N, nt, steps, seed)
        out, _ = m.sample(cond, text, N, y0=y0, use_acc_grl=False, **kw)
        out = out.cpu()
        key = (B, F_, N, nt, steps, seed)
        if key in first:
            np.testing.assert_array_equal(out.numpy(), first[key].numpy(), err_msg=f"step {i}: same case, different bits")
        else:
            first[key] = out
            ref, _ = o.sample(cond, text, N, y0=y0, **kw)
            assert _mse(out, ref, [F_] * B, [N] * B) <= 1e-4, (i, key)
    m.engine.set_option("dual", 1); m.engine.set_option("graph", 1)


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("LEMAS_FUZZ", "24")))))
def test_randomised_small_configurations_vs_oracle(seed):
    """seeded random draws over batch size, prompt / total lengths, token counts, ragged `lens`, edit masks, step counts,
    CFG on / off and the three sway settings -- every combination the sampler's bookkeeping has to get right at once"""
    rng = np.random.default_rng(1000 + seed)
    m, o = _pair()
    B = int(rng.integers(1, 4))
    Fm = int(rng.integers(2, 90))
    lens = [int(rng.integers(1, Fm + 1)) for _ in range(B)]
    lens[int(rng.integers(0, B))] = Fm if rng.random() < 0.7 else lens[0]
    durs = [int(rng.integers(l + 1, l + 140)) for l in lens]
    nts = [int(rng.integers(1, max(2, d // 2))) for d in durs]
    steps = int(rng.integers(1, 4))
    cfg = float(rng.choice([0.0, 2.0, 3.5]))
    coef = [None, 0.5, 5][int(rng.integers(0, 3))]
    cond = torch.zeros(B, Fm, 100); text = torch.full((B, max(nts)), -1, dtype=torch.long)
    for b in range(B):
        cond[b] = torch.from_numpy(synth.synth_cond_mel(seed * 10 + b, Fm))
        text[b, :nts[b]] = torch.from_numpy(synth.synth_tokens(seed * 10 + b, nts[b], VOCAB))
    edit = None
    if rng.random() < 0.4:
        edit = torch.from_numpy(rng.random((B, Fm)) < 0.7)
    kw = dict(steps=steps, cfg_strength=cfg, sway_sampling_coef=coef, lens=torch.tensor(lens))
    if edit is not None:
        kw["edit_mask"] = edit
    dur_arg = durs[0] if B == 1 else torch.tensor(durs)
    eff = [max(max(nts[b], lens[b]) + 1, durs[b]) for b in range(B)]
    y0 = torch.zeros(B, max(eff), 100)
    for b in range(B):
        y0[b, :eff[b]] = torch.from_numpy(synth.synth_noise(seed * 10 + b, eff[b]))
    if edit is not None and max(lens) < Fm:
        # lens_to_mask(lens) is max(lens) wide and cannot be and-ed with an F-wide edit mask: the reference raises
        # (cfm.py:293-295), and so must the mirror and the oracle
        with pytest.raises(RuntimeError):
            m.sample(cond, text, dur_arg, y0=y0, use_acc_grl=False, **kw)
        with pytest.raises(RuntimeError):
            o.sample(cond, text, dur_arg, y0=y0, **kw)
        return
    out, _ = m.sample(cond, text, dur_arg, y0=y0, use_acc_grl=False, **kw)
    ref, _ = o.sample(cond, text, dur_arg, y0=y0, **kw)
    assert out.shape == ref.shape
    keep = torch.zeros(B, ref.shape[1], dtype=torch.bool)
    for b in range(B):
        keep[b, :lens[b]] = True
        if edit is not None:
            keep[b, :Fm] &= edit[b]
        keep[b, eff[b]:] = True                     # beyond a sample's own duration nothing is compared
    d = (out.cpu() - ref)[~keep]
    mse = float((d.double() ** 2).mean()) if d.numel() else 0.0
    assert mse <= 1e-4, (seed, B, Fm, lens, durs, nts, steps, cfg, coef, edit is not None, mse)


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("LEMAS_FUZZ2", "16")))))
def test_randomised_conditioning_flags_vs_oracle(seed):
    """the conditioning switches of CFM.sample in random combination: prosody embedding, accent-GRL conditioning
    (use_acc_grl with ref_ratio 1 or a clip-and-shuffle ratio), no_ref_audio -- against the oracle's restatement of
    cfm.py:266-283, 313-324, 387-388, 464-466 (itself pinned by the mini_grl_* / mini_noref / mini_prosody goldens)"""
    import random
    rng = np.random.default_rng(5000 + seed)
    prosody = bool(rng.random() < 0.5)
    m, o = _pair(depth=2, seed=141, prosody=prosody)
    B = 1 if rng.random() < 0.6 else int(rng.integers(2, 4))
    F_ = int(rng.integers(30, 260))
    durs = [int(rng.integers(F_ + 2, F_ + 160)) for _ in range(B)]
    nts = [int(rng.integers(2, 40)) for _ in range(B)]
    use_grl = bool(rng.random() < 0.6)
    ref_ratio = float(rng.choice([1.0, 0.5, 0.3])) if (use_grl and B == 1) else 1.0
    no_ref = bool((not prosody) and rng.random() < 0.3)
    cond = torch.stack([torch.from_numpy(synth.synth_cond_mel(seed * 7 + b, F_)) for b in range(B)])
    text = torch.full((B, max(nts)), -1, dtype=torch.long)
    for b in range(B):
        text[b, :nts[b]] = torch.from_numpy(synth.synth_tokens(seed * 7 + b, nts[b], VOCAB))
    N = max(durs)
    y0 = torch.zeros(B, N, 100)
    for b in range(B):
        y0[b, :durs[b]] = torch.from_numpy(synth.synth_noise(seed * 7 + b, durs[b]))
    kw = dict(steps=int(rng.integers(1, 4)), cfg_strength=2.0, sway_sampling_coef=5, use_acc_grl=use_grl, ref_ratio=ref_ratio)
    if prosody:
        kw["prosody_embeds"] = torch.from_numpy(synth.synth_prosody_embed(seed + 900, B))
    if no_ref:
        kw.update(no_ref_audio=True, cond_noise=torch.from_numpy(synth.synth_noise(seed + 901, B * N).reshape(B, N, 100)))
    dur_arg = durs[0] if B == 1 else torch.tensor(durs)
    random.seed(seed)
    out, _ = m.sample(cond, text, dur_arg, y0=y0, **kw)
    random.seed(seed)
    ref, _ = o.sample(cond, text, dur_arg, y0=y0, **kw)
    mse = _mse(out.cpu(), ref, [F_] * B, durs)
    assert mse <= 1e-4, (seed, prosody, B, F_, durs, use_grl, ref_ratio, no_ref, mse)
    if not no_ref:       # kept frames are the (prosody-shifted) prompt itself
        assert float((out.cpu()[:, :F_] - ref[:, :F_]).abs().max()) <= 1e-5


def test_graph_cache_is_bounded_and_survives_many_lengths():
    """40 different lengths through one engine (more than the 32-entry graph cache): results stay right, the first length
    still reproduces its bits afterwards"""
    m, o = _pair()
    F_ = 20
    cond = torch.from_numpy(synth.synth_cond_mel(400, F_))[None]
    text = torch.from_numpy(synth.synth_tokens(401, 12, VOCAB))[None]
    outs = {}
    for i, N in enumerate(list(range(60, 100)) + [60]):
        y0 = torch.from_numpy(synth.synth_noise(402, N))[None]
        out, _ = m.sample(cond, text, N, y0=y0, steps=2, cfg_strength=2.0, sway_sampling_coef=5, use_acc_grl=False)
        if N in outs:
            np.testing.assert_array_equal(out.cpu().numpy(), outs[N])
        outs[N] = out.cpu().numpy()
    ref, _ = o.sample(cond, text, 99, y0=torch.from_numpy(synth.synth_noise(402, 99))[None], steps=2, cfg_strength=2.0, sway_sampling_coef=5)
    assert _mse(torch.from_numpy(outs[99]), ref, [F_], [99]) <= 1e-4
