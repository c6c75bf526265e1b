"""CPU tier: the oracle (oracle/lemas_oracle.py) against vectors produced by the real reference.

The fixtures in tests/golden were written by oracle/gen_golden.py, which imports /root/reference
and runs ``CFM.sample`` (cfm.py:206-473).  Weights are regenerated from the stored seed and guarded
by a checksum.  Tolerance: both sides are fp32 on CPU; only op order differs.
"""
import os

import numpy as np
import pytest
import torch

from lemas_tts_amd import synth
from lemas_tts_amd.model.layout import DiTArch
from oracle import lemas_oracle as O

CASES = ["mini_plain", "mini_nocfg_nosway", "mini_batch", "mini_edit", "mini_prosody", "mini_noref", "mini_noref_prosody", "mini_grl_prosody", "mini_grl_shuffle", "mini_duplicate", "full_plain", "full_outlier"]
ATOL = 5e-5   # measured max |err| 3.7e-6 (fp32 vs fp32, different summation order); |out| ~ 1.8


def load_case(golden_dir, name):
    fx = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    arch = DiTArch(depth=int(fx["arch_depth"]))
    sd = synth.synth_cfm_state_dict(arch, int(fx["vocab"]), int(fx["wseed"]), prosody=bool(fx["prosody"]),
                                    outlier=tuple(fx["outlier"]) if "outlier" in fx else None)
    assert abs(synth.checksum(sd) - float(fx["wchecksum"])) < 1e-6 * abs(float(fx["wchecksum"])), "RNG drift"
    return fx, arch, sd


def oracle_sample(fx, arch, sd):
    cfm = O.OracleCFM(sd, arch)
    coef = None if np.isnan(fx["coef"]) else float(fx["coef"])
    if coef is not None and coef == int(coef):
        coef = int(coef)
    kw = {}
    if "edit_mask" in fx:
        kw["edit_mask"] = torch.from_numpy(fx["edit_mask"])
    if "prosody_embeds" in fx:
        kw["prosody_embeds"] = torch.from_numpy(fx["prosody_embeds"])
    if "cond_noise" in fx:
        kw.update(no_ref_audio=True, cond_noise=torch.from_numpy(fx["cond_noise"]))
    if "use_acc_grl" in fx:
        kw.update(use_acc_grl=True, ref_ratio=float(fx["ref_ratio"]))
    if "duplicate_test" in fx:              # cfm.py:307-309, 438-443
        kw.update(duplicate_test=True, t_inter=float(fx["t_inter"]))
    if "pyseed" in fx:                      # clip_and_shuffle draws from Python's random (cfm.py:39-84)
        import random
        random.seed(int(fx["pyseed"]))
    B = int(fx["B"])
    dur = fx["duration"]
    return cfm.sample(torch.from_numpy(fx["cond"]), torch.from_numpy(fx["text"]),
                      int(dur[0]) if B == 1 else torch.from_numpy(dur),
                      y0=torch.from_numpy(fx["y0"]), lens=torch.from_numpy(fx["lens"]),
                      steps=int(fx["steps"]), cfg_strength=float(fx["cfg"]), sway_sampling_coef=coef, **kw)


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(golden_dir, name):
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    fx, arch, sd = load_case(golden_dir, name)
    out, traj = oracle_sample(fx, arch, sd)
    assert out.shape == fx["out"].shape
    if "trajectory" in fx:                  # the longer cases store `out` only
        np.testing.assert_allclose(traj.numpy(), fx["trajectory"], atol=ATOL, rtol=0)
    np.testing.assert_allclose(out.numpy(), fx["out"], atol=ATOL, rtol=0)
    # the conditioning region of ``out`` is the (prosody-shifted) cond itself (cfm.py:461)
    if "edit_mask" not in fx and "prosody_embeds" not in fx and "cond_noise" not in fx:   # (also true for mini_grl_shuffle: out keeps cond)
        for b in range(int(fx["B"])):
            L = int(fx["lens"][b])
            np.testing.assert_array_equal(out.numpy()[b, :L], fx["cond"][b, :L])


def test_oracle_matches_the_reference_at_full_depth_and_production_length(golden_dir):
    """configs[0] as the reference itself ran it (22 blocks, F = 375, N = 750, all 16 Euler steps; tests/golden/configs0_nfe16.npz):
    the oracle is pinned at the real depth and a production sequence length over a whole solve, not only on the 2-block minis and
    the 3-step full_plain case.  ~30 s of host time."""
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    fx, arch, sd = load_case(golden_dir, "configs0_nfe16")
    assert arch.depth == 22
    out, _ = oracle_sample(fx, arch, sd)
    err = np.abs(out.numpy() - fx["out"]).max()
    print(f"\n[oracle vs reference, configs0 full size, NFE 16] max|err| {err:.3e}")
    np.testing.assert_allclose(out.numpy(), fx["out"], atol=ATOL, rtol=0)


def test_sway_cap_values():
    # values the survey measured by running the reference's closure (cfm.py:343-373)
    for steps, want in ((16, 4.532), (32, 3.486), (48, 3.047), (64, 2.788)):
        assert abs(O.sway_max(steps) - want) < 1e-3


def test_time_grid_endpoints_and_monotone():
    for steps in (4, 16, 32, 48):
        for coef in (None, 1, 5, 3.0):
            t = O.time_grid(steps, coef)
            assert t[0] == 0 and t[-1] == 1 and bool((t[1:] > t[:-1]).all())
