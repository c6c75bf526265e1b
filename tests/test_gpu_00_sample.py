"""GPU tier, path level: the HIP sampler (through the C ABI) against the committed golden vectors that the real
reference produced (tests/golden, see oracle/gen_golden.py) and against the fp32 oracle on fresh seeded inputs.

Tolerance: north_star asks for mel-MSE <= 1e-4 vs the fp32 reference on identical (tokens, ref-mel, noise, NFE)
inputs; the step loop uses bf16 MFMA operands with fp32 accumulation / residual stream / ODE state."""
import os

import numpy as np
import pytest
import torch

from conftest import needs_measurement_build
from lemas_tts_amd import synth
from lemas_tts_amd.model.layout import DiTArch

pytestmark = pytest.mark.gpu

MSE_TOL = 1e-4
GOLDEN_CASES = ["mini_plain", "mini_nocfg_nosway", "mini_batch", "mini_edit", "mini_prosody", "mini_noref", "mini_noref_prosody", "mini_grl_prosody", "mini_grl_shuffle", "mini_duplicate", "full_plain", "full_outlier"]


def _load(golden_dir, name):
    fx = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    arch = DiTArch(depth=int(fx["arch_depth"]))
    sd = synth.synth_cfm_state_dict(arch, int(fx["vocab"]), int(fx["wseed"]), prosody=bool(fx["prosody"]),
                                    outlier=tuple(fx["outlier"]) if "outlier" in fx else None)
    assert abs(synth.checksum(sd) - float(fx["wchecksum"])) < 1e-6 * abs(float(fx["wchecksum"])), "RNG drift"
    return fx, arch, sd


_models = {}


def _model(arch, vocab, wseed, prosody, sd):
    from lemas_tts_amd.model.cfm import CFM
    key = (arch.depth, vocab, wseed, prosody)
    if key not in _models:
        _models.clear()  # one resident model at a time
        _models[key] = CFM(arch, vocab, sd, device="cuda:0", use_prosody_encoder=prosody)
    return _models[key]


def _run_case(fx, arch, sd, graph=True, traj=True):
    m = _model(arch, int(fx["vocab"]), int(fx["wseed"]), bool(fx["prosody"]), sd)
    m.engine.set_option("graph", 1 if graph else 0)
    coef = None if np.isnan(fx["coef"]) else float(fx["coef"])
    if coef is not None and coef == int(coef):
        coef = int(coef)
    B = int(fx["B"])
    kw = {}
    if "edit_mask" in fx:
        kw["edit_mask"] = torch.from_numpy(fx["edit_mask"])
    if "prosody_embeds" in fx:
        kw["prosody_embeds"] = torch.from_numpy(fx["prosody_embeds"])
    if "cond_noise" in fx:
        kw.update(no_ref_audio=True, cond_noise=torch.from_numpy(fx["cond_noise"]))
    grl = "use_acc_grl" in fx
    if grl:
        kw["ref_ratio"] = float(fx["ref_ratio"])
    if "duplicate_test" in fx:              # cfm.py:307-309, 438-443
        kw.update(duplicate_test=True, t_inter=float(fx["t_inter"]))
    if "pyseed" in fx:                      # clip_and_shuffle draws from Python's random (cfm.py:39-84)
        import random
        random.seed(int(fx["pyseed"]))
    dur = fx["duration"]
    out, tr = m.sample(torch.from_numpy(fx["cond"]), torch.from_numpy(fx["text"]),
                       int(dur[0]) if B == 1 else torch.from_numpy(dur), lens=torch.from_numpy(fx["lens"]),
                       steps=int(fx["steps"]), cfg_strength=float(fx["cfg"]), sway_sampling_coef=coef,
                       y0=torch.from_numpy(fx["y0"]), use_acc_grl=grl, return_trajectory=traj, **kw)
    torch.cuda.synchronize()
    return out.cpu().numpy(), None if tr is None else tr.cpu().numpy()


def _gen_mse(out, ref, fx):
    """MSE over the generated (non-conditioning) frames of every sample, as SURVEY.md 8d defines mel-MSE."""
    se, cnt = 0.0, 0
    for b in range(int(fx["B"])):
        L, D = int(fx["lens"][b]), int(fx["duration"][b])
        keep = np.ones(out.shape[1], bool)
        keep[:L] = False
        if "edit_mask" in fx:
            keep = ~(np.pad(fx["edit_mask"][b], (0, out.shape[1] - fx["edit_mask"].shape[1])) & (np.arange(out.shape[1]) < L))
        keep &= np.arange(out.shape[1]) < D
        d = out[b, keep] - ref[b, keep]
        se += float((d.astype(np.float64) ** 2).sum())
        cnt += d.size
    return se / max(cnt, 1)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_sampler_matches_reference_golden(golden_dir, name):
    fx, arch, sd = _load(golden_dir, name)
    out, traj = _run_case(fx, arch, sd, graph=False, traj=True)
    mse = _gen_mse(out, fx["out"], fx)
    mx = float(np.abs(out - fx["out"]).max())
    terr = f"{np.abs(traj - fx['trajectory']).max():.3e}" if "trajectory" in fx else "n/a (fixture stores `out` only)"
    print(f"\n[{name}] mel-MSE {mse:.3e}  max|err| {mx:.3e}  traj max|err| {terr}")
    # the solve starts from the caller's noise -- or, in the duplicate_test corner, from its blend with the shifted prompt (same host arithmetic)
    assert np.array_equal(traj[0], fx["trajectory"][0] if "duplicate_test" in fx else fx["y0"])
    assert mse <= MSE_TOL, (mse, mx)
    # conditioning frames of `out` are copied, not computed (cfm.py:461): exact
    if "edit_mask" not in fx and "prosody_embeds" not in fx and "cond_noise" not in fx:
        for b in range(int(fx["B"])):
            L = int(fx["lens"][b])
            np.testing.assert_array_equal(out[b, :L], fx["cond"][b, :L])


@pytest.mark.parametrize("name", ["mini_plain", "mini_batch"])
def test_graph_replay_is_bit_identical_to_eager(golden_dir, name):
    fx, arch, sd = _load(golden_dir, name)
    eager, _ = _run_case(fx, arch, sd, graph=False, traj=False)
    graph, _ = _run_case(fx, arch, sd, graph=True, traj=False)
    graph2, _ = _run_case(fx, arch, sd, graph=True, traj=False)
    np.testing.assert_array_equal(eager, graph)
    np.testing.assert_array_equal(graph, graph2)


@pytest.mark.parametrize("name", ["mini_plain", "mini_batch"])
def test_dual_lane_is_bit_identical_to_single_lane(golden_dir, name):
    """The two CFG branches as concurrent lanes (two streams / graph branches) must not change a single bit."""
    fx, arch, sd = _load(golden_dir, name)
    m = _model(arch, int(fx["vocab"]), int(fx["wseed"]), bool(fx["prosody"]), sd)
    outs = {}
    for dual in (0, 1):
        m.engine.set_option("dual", dual)
        for graph in (False, True):
            outs[(dual, graph)] = _run_case(fx, arch, sd, graph=graph, traj=False)[0]
    m.engine.set_option("dual", 1)
    ref = outs[(0, False)]
    for k, v in outs.items():
        np.testing.assert_array_equal(v, ref, err_msg=str(k))


@pytest.mark.parametrize("name", ["mini_plain", "mini_batch", "full_plain"])
def test_default_dispatch_equals_the_tiles_without_loader_waves(golden_dir, name):
    """Round 6: for bf16 operands the dispatch picks the LOADER-WAVE forms of the lock-step GEMM tiles (gemm_bf16.hip pick_tile / pick_qkv_tile:
    tiles 28 / 30 / 31, four waves that only request and await the ring's LDS-DMA pieces behind the compute waves).  Same ring, LDS image, barriers
    and MFMA order: forcing the plain tiles (engine measurement options) must reproduce the default's output bit for bit -- the block GEMMs of
    both widths and the fused QK + V launch, on short rows (64 x 64 / 128 x 64 tiles) and at full depth."""
    fx, arch, sd = _load(golden_dir, name)
    m = _model(arch, int(fx["vocab"]), int(fx["wseed"]), bool(fx["prosody"]), sd)
    outs = {}
    arms = {"default": {}, "plain small": {"tile_n1024": 19, "tile_n2048": 18, "tile_qkv": 18}, "plain 128": {"tile_n1024": 17, "tile_n2048": 26, "tile_qkv": 26},
            "loaders 128": {"tile_n1024": 28, "tile_n2048": 28, "tile_qkv": 28}, "loaders small": {"tile_n1024": 31, "tile_n2048": 30, "tile_qkv": 30}}
    for arm, opts in arms.items():
        for k in ("tile_n1024", "tile_n2048", "tile_qkv"):
            m.engine.set_option(k, opts.get(k, 0))
        outs[arm] = _run_case(fx, arch, sd, graph=True, traj=False)[0]      # (_model caches: the same engine)
    for k in ("tile_n1024", "tile_n2048", "tile_qkv"):
        m.engine.set_option(k, 0)
    for arm, v in outs.items():
        np.testing.assert_array_equal(v, outs["default"], err_msg=arm)


@needs_measurement_build
@pytest.mark.parametrize("name", ["mini_plain", "mini_batch", "full_plain"])
def test_fused_layernorm_tail_is_bit_identical_to_separate_launches(golden_dir, name):
    """Option ln_fused: the AdaLN LayerNorms behind the gated residual updates as the tail of the out-projection / FF2 launches
    (gemm_bf16.hip ln_tail) -- the same row arithmetic as the stand-alone kernel (csrc/ln_core.h), so not a bit may change, with one
    lane or two, eager or replayed."""
    fx, arch, sd = _load(golden_dir, name)
    m = _model(arch, int(fx["vocab"]), int(fx["wseed"]), bool(fx["prosody"]), sd)
    outs = {}
    for ln in (0, 1):
        m.engine.set_option("ln_fused", ln)
        for dual in (0, 1):
            m.engine.set_option("dual", dual)
            for graph in (False, True):
                outs[(ln, dual, graph)] = _run_case(fx, arch, sd, graph=graph, traj=False)[0]
    m.engine.set_option("dual", 1)
    m.engine.set_option("ln_fused", 0)
    m.engine.check_health()
    ref = outs[(0, 0, False)]
    for k, v in outs.items():
        np.testing.assert_array_equal(v, ref, err_msg=str(k))


def test_dit_forward_vs_oracle(golden_dir):
    """One DiT forward (both CFG branches) against the fp32 oracle: localises step-loop errors."""
    from oracle import lemas_oracle as O
    fx, arch, sd = _load(golden_dir, "mini_batch")
    m = _model(arch, int(fx["vocab"]), int(fx["wseed"]), False, sd)
    B, N = int(fx["B"]), int(fx["N"])
    cond = torch.from_numpy(fx["cond"])
    lens, dur = torch.from_numpy(fx["lens"]), torch.from_numpy(fx["duration"])
    text = torch.from_numpy(fx["text"])
    F_ = cond.shape[1]
    cond_pad = torch.nn.functional.pad(cond, (0, 0, 0, N - F_))
    cmask = torch.nn.functional.pad(O.lens_to_mask(lens), (0, N - F_), value=False)
    t = O.time_grid(4, 5)
    m.engine.prepare(cond_pad, cmask, text, t.numpy(), cond_frames=F_, cfg_strength=2.0, seq_len=dur.to(torch.int32))
    x = torch.from_numpy(fx["trajectory"][2])
    pred = m.engine.forward(x, 2).cpu()
    oc = O.OracleCFM(sd, arch)
    step_cond = torch.where(cmask[..., None], cond_pad, torch.zeros_like(cond_pad))
    mask = O.lens_to_mask(dur)
    ref_c = oc.dit.forward(x, step_cond, text, t[2], False, False, mask, True)
    ref_u = oc.dit.forward(x, step_cond, text, t[2], True, True, mask, True)
    ref = torch.cat([ref_c, ref_u])
    valid = torch.cat([mask, mask])
    err = ((pred - ref)[valid]).abs()
    print(f"\n[dit_forward] max|err| {err.max():.3e} rms {err.pow(2).mean().sqrt():.3e}  |ref| rms {ref[valid].pow(2).mean().sqrt():.3e}")
    assert err.pow(2).mean().sqrt() < 2e-2


def test_utterance_shards_are_independent(golden_dir):
    """Property the multi-GPU split relies on: a sample's result does not depend on its batch mates
    when lengths are equal (no mask, no padding leak) -- B=2 of identical items == 2 x B=1, bit for bit."""
    fx, arch, sd = _load(golden_dir, "mini_plain")
    m = _model(arch, int(fx["vocab"]), int(fx["wseed"]), False, sd)
    cond, text, y0 = (torch.from_numpy(fx[k]) for k in ("cond", "text", "y0"))
    one, _ = m.sample(cond, text, int(fx["duration"][0]), steps=4, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False)
    two, _ = m.sample(cond.repeat(2, 1, 1), text.repeat(2, 1), int(fx["duration"][0]), steps=4, cfg_strength=2.0,
                      sway_sampling_coef=5, y0=y0.repeat(2, 1, 1), use_acc_grl=False)
    # B=2 takes the masked path (seq_len given) while B=1 does not: same math, full-length mask
    np.testing.assert_array_equal(two[0].cpu().numpy(), two[1].cpu().numpy())
    np.testing.assert_allclose(two[0].cpu().numpy(), one[0].cpu().numpy(), atol=0, rtol=0)


def test_seed_draws_the_reference_cpu_noise(golden_dir):
    """north_star: parity on identical (phone-seq, ref-mel, NOISE SEED, NFE).  With `seed` given and no explicit y0 the mirror
    draws what the reference's CPU path draws at cfm.py:430-435 (host generator, re-seeded per sample, zero right-pad): the
    fixtures' y0 came out of the reference with seed 101 (mini_plain) / 103 (mini_batch, oracle/gen_golden.py)."""
    for name, seed in (("mini_plain", 101), ("mini_batch", 103)):
        fx, arch, sd = _load(golden_dir, name)
        m = _model(arch, int(fx["vocab"]), int(fx["wseed"]), False, sd)
        B = int(fx["B"])
        dur = fx["duration"]
        out, tr = m.sample(torch.from_numpy(fx["cond"]), torch.from_numpy(fx["text"]), int(dur[0]) if B == 1 else torch.from_numpy(dur),
                           lens=torch.from_numpy(fx["lens"]), steps=int(fx["steps"]), cfg_strength=float(fx["cfg"]),
                           sway_sampling_coef=float(fx["coef"]), seed=seed, use_acc_grl=False, return_trajectory=True)
        np.testing.assert_array_equal(tr[0].cpu().numpy(), fx["y0"])
        assert _gen_mse(out.cpu().numpy(), fx["out"], fx) <= MSE_TOL


@pytest.mark.parametrize("name", ["mini_plain", "mini_batch", "mini_prosody", "full_plain", "full_outlier"])
def test_ln_fold_option_meets_the_reference_goldens(golden_dir, name):
    """Option "ln_fold" (off by default, csrc/common.h GemmParams): the block chain's LayerNorms folded across the GEMMs on either side --
    no LayerNorm launches after a step's first, per-step c1 / c2 rows in the AdaLN table.  It is a different rounding of the same
    arithmetic (bf16 of x (1 + s) instead of LN(x) (1 + s) + b), so the bar is the reference's golden at the sampler's tolerance; the
    single-lane, two-lane, eager and replayed runs of the fold stay bit-identical among themselves (the row statistics are added up in
    one fixed tree whatever tile produces or consumes them)."""
    fx, arch, sd = _load(golden_dir, name)
    m = _model(arch, int(fx["vocab"]), int(fx["wseed"]), bool(fx["prosody"]), sd)
    base, _ = _run_case(fx, arch, sd, graph=True, traj=False)
    outs = {}
    try:
        m.engine.set_option("ln_fold", 1)
        for dual in (1, 0):
            m.engine.set_option("dual", dual)
            for graph in (False, True):
                outs[(dual, graph)] = _run_case(fx, arch, sd, graph=graph, traj=False)[0]
    finally:
        m.engine.set_option("dual", 1)
        m.engine.set_option("ln_fold", 0)
    m.engine.check_health()
    out = outs[(1, True)]
    mse, dm = _gen_mse(out, fx["out"], fx), _gen_mse(out, base, fx)
    print(f"\n[{name}] ln_fold: mel-MSE vs reference {mse:.3e} (default path {_gen_mse(base, fx['out'], fx):.3e}), vs default path {dm:.3e}")
    assert mse <= MSE_TOL, mse
    for k, v in outs.items():
        np.testing.assert_array_equal(v, out, err_msg=str(k))


@needs_measurement_build
@pytest.mark.parametrize("name", ["mini_plain", "full_plain"])
def test_persistent_ff_half_is_bit_identical_to_separate_launches(golden_dir, name):
    """Measurement option block_persist: out-projection -> ff_norm -> FF1 -> FF2 of a lane as ONE persistent launch with grid barriers between
    the stages (gemm_bf16.hip gemm_chain_ffhalf_kernel).  The stages are the bodies of the separate launches: not a bit may change, eager or replayed."""
    fx, arch, sd = _load(golden_dir, name)
    m = _model(arch, int(fx["vocab"]), int(fx["wseed"]), bool(fx["prosody"]), sd)
    outs = {}
    for persist in (0, 1, 2):
        m.engine.set_option("block_persist", persist)
        for graph in (False, True):
            outs[(persist, graph)] = _run_case(fx, arch, sd, graph=graph, traj=False)[0]
    m.engine.set_option("block_persist", 0)
    m.engine.check_health()
    ref = outs[(0, False)]
    for k, v in outs.items():
        np.testing.assert_array_equal(v, ref, err_msg=str(k))


@pytest.mark.parametrize("name", ["mini_batch", "mini_prosody", "mini_duplicate"])
def test_lane_split_is_bit_identical(golden_dir, name):
    """Option lane_split: each CFG branch of a batch cut into k groups of samples, every group its own chain of launches on its own stream /
    graph branch (2 k lanes).  Rows are independent and the per-sample lengths travel with their group: not a bit may change, ragged batch
    (mini_batch: lens 50 / 70, durations 120 / 150) included, eager or replayed."""
    fx, arch, sd = _load(golden_dir, name)
    m = _model(arch, int(fx["vocab"]), int(fx["wseed"]), bool(fx["prosody"]), sd)
    outs = {}
    for split in (1, 2):
        m.engine.set_option("lane_split", split)
        for graph in (False, True):
            outs[(split, graph)] = _run_case(fx, arch, sd, graph=graph, traj=False)[0]
    m.engine.set_option("lane_split", 0)
    ref = outs[(1, False)]
    for k, v in outs.items():
        np.testing.assert_array_equal(v, ref, err_msg=str(k))
