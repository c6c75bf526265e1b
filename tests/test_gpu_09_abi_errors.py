"""GPU tier: error behaviour of the C ABI -- wrong calls come back as an error code + lemas_last_error() text, never as a crash
or a silent success (include/lemas_hip.h: 0 = ok, LEMAS_E_* / negated hipError_t otherwise)."""
import ctypes as C

import numpy as np
import pytest
import torch

from lemas_tts_amd import _lib, synth
from lemas_tts_amd.model.layout import DiTArch

pytestmark = pytest.mark.gpu


def _new_dit(depth=1):
    L = _lib.lib()
    a = DiTArch(depth=depth)
    cfg = _lib.DitConfig(a.dim, a.depth, a.heads, a.dim_head, a.ff_mult, a.text_dim, a.conv_layers, a.mel_dim, 899, a.conv_pos_kernel,
                         a.conv_pos_groups, a.time_freq_dim, 0)
    h = C.c_void_p()
    assert L.lemas_dit_create(C.byref(cfg), C.byref(h)) == 0
    return L, h


def _err(L):
    return L.lemas_last_error().decode()


def test_dit_wrong_calls_are_refused():
    L, h = _new_dit()
    try:
        assert L.lemas_dit_create(None, C.byref(C.c_void_p())) != 0
        # finalize with nothing loaded: strict load fails and names a missing tensor
        assert L.lemas_dit_finalize(h) != 0 and "missing" in _err(L).lower()
        # unknown tensor name / wrong shape
        w = np.zeros((4, 4), np.float32)
        shp = (C.c_int64 * 2)(4, 4)
        assert L.lemas_dit_load_weight(h, b"transformer.no.such.tensor", w.ctypes.data_as(C.c_void_p), shp, 2) != 0
        assert "no.such.tensor" in _err(L)
        assert L.lemas_dit_load_weight(h, b"transformer.proj_out.bias", w.ctypes.data_as(C.c_void_p), shp, 2) != 0
        assert "shape" in _err(L).lower()
        # sampling / solving / forward before finalize + prepare
        args = _lib.SampleArgs()
        assert L.lemas_dit_sample(h, C.byref(args), None) != 0
        assert L.lemas_dit_solve(h, C.byref(args), None) != 0
        assert L.lemas_dit_forward(h, None, 0, None, None) != 0
        assert L.lemas_dit_set_option(h, b"no_such_option", 1) != 0 and "no_such_option" in _err(L)
        assert L.lemas_dit_sample(None, C.byref(args), None) != 0
    finally:
        L.lemas_dit_destroy(h)


def test_dit_bad_sample_arguments():
    from lemas_tts_amd.engine import DiTEngine
    arch = DiTArch(depth=1)
    eng = DiTEngine(arch, 898, synth.synth_cfm_state_dict(arch, 898, 1), device="cuda:0")
    L = _lib.lib()
    cond = torch.zeros(1, 40, 100, device="cuda:0")
    mask = torch.zeros(1, 40, dtype=torch.uint8, device="cuda:0")
    text = torch.zeros(1, 5, dtype=torch.int64, device="cuda:0")
    y = torch.zeros(1, 40, 100, device="cuda:0")
    tg = np.linspace(0, 1, 3).astype(np.float32)

    def call(**over):
        a = _lib.SampleArgs()
        a.batch, a.frames, a.cond_frames, a.text_len, a.steps, a.cfg_strength = 1, 40, 20, 5, 2, 2.0
        a.cond, a.cond_mask, a.text, a.y = cond.data_ptr(), mask.data_ptr(), text.data_ptr(), y.data_ptr()
        a.t_grid = tg.ctypes.data_as(C.POINTER(C.c_float))
        for k, v in over.items():
            setattr(a, k, v)
        return L.lemas_dit_sample(eng._h, C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream))

    assert call() == 0
    torch.cuda.synchronize()
    for bad in (dict(batch=0), dict(frames=0), dict(frames=5000), dict(cond_frames=41), dict(steps=0), dict(cond=None), dict(text=None),
                dict(cond_mask=None), dict(y=None), dict(t_grid=None), dict(text_len=0)):
        assert call(**bad) != 0, bad
        assert _err(L), bad
    # ABI 200: a client built against another layout of lemas_sample_args is refused by its struct_size, with a message that says what to do
    assert call(struct_size=0) != 0 and "struct_size" in _err(L)
    assert call(struct_size=C.sizeof(_lib.SampleArgs) - 8) != 0 and "struct_size" in _err(L)
    # cond_rows: 0 = `frames` rows (the old meaning), fewer rows are zero-filled on the device, more than `frames` is a caller bug
    assert call(cond_rows=20) == 0
    assert call(cond_rows=41) != 0 and _err(L)
    torch.cuda.synchronize()
    # a non-monotone time grid is what torchdiffeq would reject (SURVEY a-O)
    tg[:] = [0.0, 0.6, 0.5]
    assert call() != 0 and "monoton" in _err(L).lower()


def test_vocos_mel_resample_prosody_wrong_calls():
    L = _lib.lib()
    v = C.c_void_p()
    assert L.lemas_vocos_create(100, 512, 1536, 8, 1024, 256, C.byref(v)) == 0
    try:
        assert L.lemas_vocos_finalize(v) != 0                       # nothing loaded
        assert L.lemas_vocos_decode(v, None, 1, 10, C.c_float(1.0), None, None) != 0
    finally:
        L.lemas_vocos_destroy(v)
    m = C.c_void_p()
    assert L.lemas_mel_create(1000, 256, 100, 24000, C.byref(m)) == 0 or True   # n_fft must be a multiple of 4: 1000 is
    if m:
        buf = torch.zeros(1, 100, device="cuda:0")
        out = torch.zeros(1, 1, 100, device="cuda:0")
        assert L.lemas_mel_forward(m, buf.data_ptr(), 1, 100, out.data_ptr(), None) != 0      # shorter than the reflect pad
        L.lemas_mel_destroy(m)
    assert L.lemas_mel_create(1023, 256, 100, 24000, C.byref(C.c_void_p())) != 0
    assert L.lemas_resample_create(0, 24000, C.byref(C.c_void_p())) != 0
    cfg = _lib.ProsodyConfig()
    cfg.n_layers = 2
    assert L.lemas_prosody_create(C.byref(cfg), C.byref(C.c_void_p())) != 0 and "architecture" in _err(L)
    assert L.lemas_prosody_fbank_frames(399) == 0 and L.lemas_prosody_fbank_frames(400) == 1 and L.lemas_prosody_fbank_frames(16000) == 98


def test_create_destroy_cycles_do_not_leak_device_memory():
    """every object frees what it allocated: 12 create / use / destroy cycles of each engine leave the free-memory reading flat"""
    import gc
    from lemas_tts_amd.engine import DiTEngine, MelEngine, ProsodyEngine, ResampleEngine, VocosEngine
    from lemas_tts_amd.model.layout import ProsodyArch
    arch = DiTArch(depth=1)
    sd = synth.synth_cfm_state_dict(arch, 898, 3)
    vsd = synth.synth_vocos_state_dict(4)
    parch = ProsodyArch(channels=(64, 64, 64, 128), kernel_sizes=(5, 3, 3, 1), dilations=(1, 2, 3, 1), attention_channels=16, res2net_scale=4,
                        se_channels=8, groups=(1, 1, 1, 1), embed_dim=32)
    psd = synth.synth_prosody_encoder_state_dict(5, parch)
    cond = torch.from_numpy(synth.synth_cond_mel(6, 40))[None]
    text = torch.from_numpy(synth.synth_tokens(7, 10, 898))[None]
    y0 = torch.from_numpy(synth.synth_noise(8, 120))[None]
    cm = torch.zeros(1, 120, dtype=torch.bool); cm[:, :40] = True
    tg = np.linspace(0, 1, 3).astype(np.float32) ** 2 + np.arange(3, dtype=np.float32) * 1e-3

    def cycle():
        e = DiTEngine(arch, 898, sd, device="cuda:0")
        e.sample(torch.nn.functional.pad(cond, (0, 0, 0, 80)), cm, text, tg, y0, cond_frames=40, cfg_strength=2.0)
        e.set_option("fp8", 1)
        e.sample(torch.nn.functional.pad(cond, (0, 0, 0, 80)), cm, text, tg, y0, cond_frames=40, cfg_strength=2.0)
        v = VocosEngine(vsd, device="cuda:0"); v.decode(torch.zeros(1, 100, 50))
        MelEngine(device="cuda:0").frames_first(torch.zeros(1, 4000))
        ResampleEngine(16000, 24000, device="cuda:0")(torch.zeros(1, 4000))
        p = ProsodyEngine(parch, psd, device="cuda:0"); p.encode(p.fbank(torch.zeros(2000)))
        torch.cuda.synchronize()
        for o in (e, v, p):
            o.close()

    cycle(); gc.collect(); torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(12):
        cycle()
    gc.collect(); torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, (free0, free1)          # < 64 MiB drift (torch's own caching allocator noise), not 12 x engine size


def test_device_resident_weights_and_reload_after_finalize():
    """lemas_*_load_weight_device (what the data-parallel launcher uses after the RCCL broadcast): an engine built from DEVICE
    tensors gives the same bits as one built from host arrays; reloading a weight on a live handle invalidates what was
    prepared / captured on the old tensors (no replay of a graph that baked freed addresses)."""
    from lemas_tts_amd.engine import DiTEngine, VocosEngine
    from lemas_tts_amd.model.cfm import CFM
    arch = DiTArch(depth=2)
    sd = synth.synth_cfm_state_dict(arch, 898, 77)
    cond = torch.from_numpy(synth.synth_cond_mel(78, 40))[None]
    text = torch.from_numpy(synth.synth_tokens(79, 12, 898))[None]
    y0 = torch.from_numpy(synth.synth_noise(80, 96))[None]
    kw = dict(steps=3, cfg_strength=2.0, sway_sampling_coef=5, y0=y0, use_acc_grl=False)
    host = CFM(arch, 898, sd, device="cuda:0")
    a, _ = host.sample(cond, text, 96, **kw)
    flat = {k: torch.from_numpy(np.ascontiguousarray(v)).to("cuda:0") for k, v in sd.items()}
    devm = CFM(arch, 898, flat, device="cuda:0")
    b, _ = devm.sample(cond, text, 96, **kw)
    np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    vsd = synth.synth_vocos_state_dict(81)
    mel = a[:, 40:, :].permute(0, 2, 1)
    w1 = VocosEngine(vsd, device="cuda:0").decode(mel)
    w2 = VocosEngine({k: torch.from_numpy(np.ascontiguousarray(v)).to("cuda:0") for k, v in vsd.items()}, device="cuda:0").decode(mel)
    np.testing.assert_array_equal(w1.cpu().numpy(), w2.cpu().numpy())

    # reload on a live handle: same shape sampled before (a graph for it is cached), new weights must take effect
    sd2 = synth.synth_cfm_state_dict(arch, 898, 177)
    L, eng = _lib.lib(), host.engine
    name = "transformer.transformer_blocks.0.attn.to_out.0.weight"
    w = np.ascontiguousarray(sd2[name], dtype=np.float32)
    shp = (C.c_int64 * w.ndim)(*w.shape)
    assert L.lemas_dit_load_weight(eng._h, name.encode(), w.ctypes.data_as(C.c_void_p), shp, w.ndim) == 0
    with pytest.raises(_lib.LemasError):          # not finalized again yet
        host.sample(cond, text, 96, **kw)
    assert L.lemas_dit_finalize(eng._h) == 0
    c, _ = host.sample(cond, text, 96, **kw)
    sd_mixed = dict(sd)
    sd_mixed[name] = sd2[name]
    ref, _ = CFM(arch, 898, sd_mixed, device="cuda:0").sample(cond, text, 96, **kw)
    np.testing.assert_array_equal(c.cpu().numpy(), ref.cpu().numpy())
    assert not np.array_equal(c.cpu().numpy(), a.cpu().numpy())


def test_token_ids_outside_the_vocabulary_raise_like_nn_embedding():
    from lemas_tts_amd.model.cfm import CFM
    arch = DiTArch(depth=1)
    m = CFM(arch, 50, synth.synth_cfm_state_dict(arch, 50, 3), device="cuda:0")
    cond = torch.from_numpy(synth.synth_cond_mel(4, 30))[None]
    y0 = torch.from_numpy(synth.synth_noise(5, 64))[None]
    for bad in (50, 1000, -2):
        text = torch.tensor([[1, 2, bad, 3]])
        with pytest.raises(IndexError):
            m.sample(cond, text, 64, steps=2, cfg_strength=2.0, y0=y0, use_acc_grl=False)
    out, _ = m.sample(cond, torch.tensor([[1, 2, 49, -1]]), 64, steps=2, cfg_strength=2.0, y0=y0, use_acc_grl=False)
    assert torch.isfinite(out).all()
