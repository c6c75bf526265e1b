"""GPU tier: the engine's step-graph cache (csrc/engine_dit.hip step_graph) -- buckets keyed on (batch, 128-row pitch, switches), a new
frame count inside a bucket patches an instantiated graph (hipGraphExecUpdate) instead of instantiating another, buckets are evicted
least-recently-used.  Whatever the cache does, the results must be the bits the eager path produces.  Caller in the reference: every
gen_text line has its own duration (lemas_tts/infer/utils_infer.py:520-542)."""
import numpy as np
import pytest
import torch

from lemas_tts_amd import synth
from lemas_tts_amd.model.layout import DiTArch

pytestmark = pytest.mark.gpu
VOCAB = 898


def _case(seed, F_, N):
    cond = torch.from_numpy(synth.synth_cond_mel(seed, F_))[None]
    text = torch.from_numpy(synth.synth_tokens(seed + 1, 40, VOCAB))[None]
    y0 = torch.from_numpy(synth.synth_noise(seed + 2, N))[None]
    return cond, text, y0


def _run(m, case, N):
    cond, text, y0 = case
    out, _ = m.sample(cond, text, N, steps=4, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_new_length_in_a_seen_bucket_patches_the_graph_and_matches_eager():
    from lemas_tts_amd.model.cfm import CFM
    arch = DiTArch(depth=2)
    m = CFM(arch, VOCAB, synth.synth_cfm_state_dict(arch, VOCAB, 7), device="cuda:0")
    lengths = [300, 301, 320, 300, 384, 301, 385, 300]          # pitch 384 for all but 385 (pitch 512)
    cases = {N: _case(100 + N, 120, N) for N in set(lengths)}
    m.engine.set_option("graph", 0)
    eager = {N: _run(m, cases[N], N) for N in cases}
    m.engine.set_option("graph", 1)
    base = {k: m.engine.stat(k) for k in ("graph_captures", "graph_instantiates", "graph_updates", "graph_update_failures")}
    for N in lengths:
        np.testing.assert_array_equal(_run(m, cases[N], N), eager[N], err_msg=f"N={N}")
    d = {k: m.engine.stat(k) - v for k, v in base.items()}
    print(f"\n[graph cache] {d}, buckets {m.engine.stat('graph_buckets')}")
    assert m.engine.stat("graph_buckets") == 2
    # bucket 384 saw 300, 301, 320, 300 (cached), 384, 301, 300: two instantiated graphs, everything else patched (or, if the runtime
    # refuses an update, re-instantiated: still correct, and counted)
    assert d["graph_instantiates"] + d["graph_updates"] == d["graph_captures"]
    assert d["graph_instantiates"] == 3 + d["graph_update_failures"]        # 2 slots of bucket 384 + 1 of bucket 512


def test_lru_eviction_bounds_the_cache():
    from lemas_tts_amd.model.cfm import CFM
    arch = DiTArch(depth=1)
    m = CFM(arch, VOCAB, synth.synth_cfm_state_dict(arch, VOCAB, 8), device="cuda:0")
    m.engine.set_option("graph_cache", 2)
    e0 = m.engine.stat("graph_evictions")
    outs = {}
    # largest first: a growing length re-allocates the engine's workspaces, which drops every cached graph (their addresses are stale)
    for N in (460, 200, 330, 200, 460):       # three buckets through a cache of two: 330 evicts 460, the second 460 evicts 330
        outs.setdefault(N, []).append(_run(m, _case(N, 100, N), N))
        assert m.engine.stat("graph_buckets") <= 2
    assert m.engine.stat("graph_evictions") - e0 == 2
    np.testing.assert_array_equal(outs[200][0], outs[200][1])
    np.testing.assert_array_equal(outs[460][0], outs[460][1])
    with pytest.raises(Exception):
        m.engine.set_option("graph_cache", 0)
