"""SURVEY.md 8f-4, the MDX-Net separation network itself (ConvTDFNet: uvr5/lib_v5/mdxnet.py:36-127, uvr5/lib_v5/modules.py:5-74), which the
reference runs as an onnxruntime session (uvr5/multiprocess_cuda_infer.py:225-238,262-272).

CPU tier: the restatement (oracle/mdx_oracle.py) against outputs of the REFERENCE'S OWN CLASS on seeded weights
(oracle/gen_golden_mdxnet.py -> tests/golden/mdxnet_*.npz).  GPU tier: the HIP engine (lemas_mdx_*, through ctypes -> C ABI) against the
same vectors, at four small architectures (every stage's activation) and at the Kim_Vocal_1 shape, plus the denoising shell around it.

Tolerance: everything is fp32 on both sides; the HIP path folds the inference BatchNorm into the weights and sums in MFMA order, so
results differ from the reference by rounding only.  Measured on MI355X: <= 3e-6 of the tensor's rms at the small shapes, <= 1e-5 at the
Kim shape (K up to 2592 terms per output, 60 layers deep); asserted at 1e-4 of the rms (``REL``), the bar VERDICT r5 set.
"""
import os

import numpy as np
import pytest
import torch

from oracle import mdx_oracle as MO

REL = 1e-4
MINIS = {"mini": MO.MINI, "mini_wide": MO.MINI_WIDE, "mini_notdf": MO.MINI_NOTDF, "mini_gn": MO.MINI_GN}


def _fx(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, f"mdxnet_{name}.npz")))


def _rel_rms(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).mean()) / max(np.sqrt((b ** 2).mean()), 1e-30))


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.sqrt((b ** 2).mean()), 1e-30))


# ------------------------------------------------------------------------------------------------------------------ CPU
@pytest.mark.parametrize("name", sorted(MINIS))
def test_oracle_matches_the_reference_class(golden_dir, name):
    fx, arch = _fx(golden_dir, name), MINIS[name]
    sd = MO.seeded_state_dict(arch, int(fx["seed_weights"][0]))
    x = MO.seeded_input(arch, int(fx["batch"][0]), int(fx["seed_input"][0]))
    np.testing.assert_array_equal(x, fx["input"])                  # the numpy Generator reproduces the fixture's input bit for bit
    taps = {}
    y = MO.MdxOracle(arch, sd).forward(x, taps).numpy()
    assert y.shape == fx["output"].shape
    assert _rel(y, fx["output"]) < 2e-6
    for k, v in taps.items():
        assert _rel(v.numpy(), fx[f"tap_{k}"]) < 2e-6, k


def test_schema_and_flops():
    a = MO.KIM_VOCAL_1
    names = [k for k, _ in MO.schema(a)]
    assert len(names) == len(set(names)) and "us.4.0.weight" in names and "encoding_blocks.0.tdf.0.bias" not in names
    params = sum(int(np.prod(s)) for k, s in MO.schema(a) if not k.endswith(("running_mean", "running_var")))
    assert 16.5e6 < params < 17.6e6                                # the 66.8 MB Kim_Vocal_1.onnx holds 16.7 M fp32 values
    assert 0.72e12 < MO.flops(a) < 0.76e12
    assert MO.flops(a, 2) == 2 * MO.flops(a)


def test_oracle_kim_shape_matches_the_reference_class(golden_dir):
    """2-4 s of CPU work on 8 threads (0.74 TFLOP)."""
    fx, arch = _fx(golden_dir, "kim"), MO.KIM_VOCAL_1
    sd = MO.seeded_state_dict(arch, int(fx["seed_weights"][0]))
    y = MO.MdxOracle(arch, sd).forward(MO.seeded_input(arch, 1, int(fx["seed_input"][0]))).numpy()
    _check_kim(y, fx, 2e-5)


def _check_kim(y, fx, rel, rel_sums=None):
    KIM_SAMPLE = MO.KIM_SAMPLE
    rms = float(fx["rms"][0])
    rel_sums = rel if rel_sums is None else rel_sums
    assert np.abs(y[KIM_SAMPLE] - fx["sample"]).max() / rms < rel
    assert abs(np.sqrt((y.astype(np.float64) ** 2).mean()) - rms) / rms < rel
    # sums over [b, c, t] per frequency bin (768 values each) and over [b, c, f] per frame (12288 values each): every output element is in one
    er = np.abs(y.astype(np.float64).sum(axis=(0, 1, 3)) - fx["row_sums"]).max() / (rms * np.sqrt(768))
    ec = np.abs(y.astype(np.float64).sum(axis=(0, 1, 2)) - fx["col_sums"]).max() / (rms * np.sqrt(12288))
    print(f"[mdxnet kim] row sums {er:.2e}, column sums {ec:.2e} (of rms * sqrt(count))")
    assert er < rel_sums and ec < rel_sums, (er, ec)


# ------------------------------------------------------------------------------------------------------------------ GPU
def _engine(arch, seed):
    from lemas_tts_amd.engine import MdxEngine
    return MdxEngine(arch, MO.seeded_state_dict(arch, seed))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MINIS))
def test_hip_network_matches_the_reference_class(golden_dir, name):
    fx, arch = _fx(golden_dir, name), MINIS[name]
    eng = _engine(arch, int(fx["seed_weights"][0]))
    bufs = {}
    for k in fx:
        if k.startswith("tap_"):
            bufs[k[4:]] = torch.zeros(fx[k].shape, device="cuda:0")
            eng.tap(k[4:], bufs[k[4:]])
    y = eng.forward(torch.from_numpy(fx["input"]).to("cuda:0"))
    torch.cuda.synchronize()
    worst = 0.0
    for k, b in bufs.items():
        e = _rel(b.cpu().numpy(), fx[f"tap_{k}"])
        worst = max(worst, e)
        assert e < REL, f"stage {k}: {e:.2e}"
    e = _rel(y.cpu().numpy(), fx["output"])
    print(f"\n[mdxnet {name}] output rel err {e:.2e}, worst stage {worst:.2e}")
    assert e < REL
    assert eng.flops(int(fx['batch'][0])) == MO.flops(arch, int(fx['batch'][0]))


@pytest.mark.gpu
def test_hip_network_kim_shape_matches_the_reference_class(golden_dir):
    fx, arch = _fx(golden_dir, "kim"), MO.KIM_VOCAL_1
    eng = _engine(arch, int(fx["seed_weights"][0]))
    x = torch.from_numpy(MO.seeded_input(arch, 1, int(fx["seed_input"][0]))).to("cuda:0")
    y = eng.forward(x).cpu().numpy()
    _check_kim(y, fx, REL)
    # batch of 2 (the "denoise" +- pair of Inference.run_model): each sample equals its own single forward
    y2 = eng.forward(torch.cat((x, -x))).cpu().numpy()
    np.testing.assert_array_equal(y2[0], y[0])
    assert np.abs(y2[1] - y[0]).max() > 0.1 * float(fx["rms"][0])


@pytest.mark.gpu
@pytest.mark.parametrize("arch,batch", [
    (MO.MdxArch(dim_f=12, dim_t=6, num_blocks=3, l=2, g=20, bn=None), 2),          # f = 12, 6: scalar loads; t = 6, 3: ragged row tiles
    (MO.MdxArch(dim_f=200, dim_t=20, num_blocks=3, l=1, g=52, bn=5, bias=True), 1),   # ragged column tiles, channel padding 52 -> 56 / 96
    (MO.MdxArch(dim_f=256, dim_t=64, num_blocks=7, l=1, g=16, bn=2, bias=False), 3),  # n = 3, wide tiles at the top levels
    (MO.MdxArch(dim_f=64, dim_t=8, num_blocks=1, l=2, g=48, bn=4, bias=False), 1),    # no encoder / decoder at all
    (MO.MdxArch(dim_f=48, dim_t=8, num_blocks=3, l=1, g=10, bn=0, bias=True, optimizer="adamw"), 2),   # GroupNorm + single-linear TDF
])
def test_hip_network_edge_shapes_vs_oracle(arch, batch):
    sd = MO.seeded_state_dict(arch, 5)
    x = MO.seeded_input(arch, batch, 6)
    ref = MO.MdxOracle(arch, sd).forward(x).numpy()
    from lemas_tts_amd.engine import MdxEngine
    y = MdxEngine(arch, sd).forward(torch.from_numpy(x).to("cuda:0")).cpu().numpy()
    e = _rel(y, ref)
    print(f"\n[mdxnet edge {arch.dim_f}x{arch.dim_t} g{arch.g} n{arch.n}] rel err {e:.2e}")
    assert e < REL


# The split-bf16 mode (engine option bf16x3, OFF by default): x = hi + lo in bf16, three bf16 MFMAs per product, ~2^-16 relative per product
# with fp32 accumulation -- ten times coarser than the exact path, thirty times finer than the TF32 convolutions of the reference's CUDA
# provider.  Measured against the reference class's outputs (tools/exp/mdx_bx_accuracy.py, profiles/r06/r06i_mdx_modes.txt): rms(err) / rms
# 2.0e-5 (exact path: 9e-7 ... 2e-6), max|err| / rms 1.6e-4 ... 3.9e-4 (exact: 7e-6 ... 3.9e-5).  It does NOT meet the 1e-4-of-rms max-error
# bar, which is why it is an option; what is asserted for it: rms-relative error <= 1e-4, max error <= 1e-3 of the rms.
BX_RMS, BX_MAX = 1e-4, 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mini", "mini_wide", "mini_notdf"])
def test_hip_network_split_bf16_vs_the_reference_class(golden_dir, name):
    from lemas_tts_amd.engine import MdxEngine
    fx, arch = _fx(golden_dir, name), MINIS[name]
    eng = MdxEngine(arch, MO.seeded_state_dict(arch, int(fx["seed_weights"][0])), bf16x3=True)
    bufs = {k[4:]: torch.zeros(fx[k].shape, device="cuda:0") for k in fx if k.startswith("tap_")}
    for k, b in bufs.items():
        eng.tap(k, b)
    y = eng.forward(torch.from_numpy(fx["input"]).to("cuda:0"))
    torch.cuda.synchronize()
    worst = max(_rel(b.cpu().numpy(), fx[f"tap_{k}"]) for k, b in bufs.items())
    worst_rms = max(_rel_rms(b.cpu().numpy(), fx[f"tap_{k}"]) for k, b in bufs.items())
    e, er = _rel(y.cpu().numpy(), fx["output"]), _rel_rms(y.cpu().numpy(), fx["output"])
    print(f"\n[mdxnet {name}, bf16x3] output: max err / rms {e:.2e}, rms err / rms {er:.2e}; worst stage {worst:.2e} / {worst_rms:.2e}")
    assert er < BX_RMS and worst_rms < BX_RMS and e < BX_MAX and worst < BX_MAX


@pytest.mark.gpu
def test_hip_network_split_bf16_kim_shape_and_edges(golden_dir):
    from lemas_tts_amd import _lib
    from lemas_tts_amd.engine import MdxEngine
    fx, arch = _fx(golden_dir, "kim"), MO.KIM_VOCAL_1
    eng = MdxEngine(arch, MO.seeded_state_dict(arch, int(fx["seed_weights"][0])), bf16x3=True)
    x = torch.from_numpy(MO.seeded_input(arch, 1, int(fx["seed_input"][0]))).to("cuda:0")
    y = eng.forward(x).cpu().numpy()
    rms = float(fx["rms"][0])
    d = y[MO.KIM_SAMPLE].astype(np.float64) - fx["sample"]
    print(f"\n[mdxnet kim, bf16x3] sample: max err / rms {np.abs(d).max() / rms:.2e}, rms err / rms {np.sqrt((d ** 2).mean()) / rms:.2e}")
    assert np.sqrt((d ** 2).mean()) / rms < BX_RMS
    # the checksums get their own bar in this mode: a split-bf16 product drops the lo x lo term and rounds the WEIGHTS' hi / lo parts once, so the
    # error of an output element has a part that is common to everything the same weights produced -- the 768 errors of a frequency bin (same
    # TDF rows, every frame) add coherently instead of as a random walk: measured 1.8e-3 / 1.1e-3 of rms * sqrt(count) (exact mode: 2e-5)
    _check_kim(y, fx, BX_MAX, rel_sums=5e-3)
    with pytest.raises(_lib.LemasError, match="before finalize"):          # the option decides the weight layout
        _lib.check(_lib.lib().lemas_mdx_set_option(eng._h, b"bf16x3", 0), "set_option")
    del eng
    for a2, batch in [(MO.MdxArch(dim_f=12, dim_t=6, num_blocks=3, l=2, g=20, bn=None), 2),       # scalar loads, ragged tiles, 20 -> 32 channel padding
                      (MO.MdxArch(dim_f=200, dim_t=20, num_blocks=3, l=1, g=52, bn=5, bias=True), 1),
                      (MO.MdxArch(dim_f=256, dim_t=64, num_blocks=7, l=1, g=16, bn=2, bias=False), 3)]:
        sd = MO.seeded_state_dict(a2, 5)
        xi = MO.seeded_input(a2, batch, 6)
        ref = MO.MdxOracle(a2, sd).forward(xi).numpy()
        got = MdxEngine(a2, sd, bf16x3=True).forward(torch.from_numpy(xi).to("cuda:0")).cpu().numpy()
        print(f"[mdxnet edge {a2.dim_f}x{a2.dim_t} g{a2.g} n{a2.n}, bf16x3] max err / rms {_rel(got, ref):.2e}, rms err / rms {_rel_rms(got, ref):.2e}")
        assert _rel_rms(got, ref) < BX_RMS and _rel(got, ref) < BX_MAX


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [4, 8])
def test_hip_network_conv_chunk_options_agree(golden_dir, chunk):
    """Engine option conv_chunk: 4 (default) or 8 input channels per K chunk of the exact 3x3 kernel -- same products, the partial sums of a
    chunk grouped differently: both within the bar, and equal to each other to rounding."""
    from lemas_tts_amd import _lib
    fx, arch = _fx(golden_dir, "mini"), MINIS["mini"]
    eng = _engine(arch, int(fx["seed_weights"][0]))
    _lib.check(_lib.lib().lemas_mdx_set_option(eng._h, b"conv_chunk", chunk), "conv_chunk")
    try:
        y = eng.forward(torch.from_numpy(fx["input"]).to("cuda:0")).cpu().numpy()
    finally:
        _lib.check(_lib.lib().lemas_mdx_set_option(eng._h, b"conv_chunk", 4), "conv_chunk")
    assert _rel(y, fx["output"]) < REL


@pytest.mark.gpu
def test_hip_network_strict_loading():
    from lemas_tts_amd import _lib
    from lemas_tts_amd.engine import MdxEngine
    sd = MO.seeded_state_dict(MO.MINI, 1)
    with pytest.raises(_lib.LemasError, match="missing tensor"):
        MdxEngine(MO.MINI, {k: v for k, v in sd.items() if k != "ds.1.1.running_var"})
    with pytest.raises(_lib.LemasError, match="unexpected tensor"):
        MdxEngine(MO.MINI, dict(sd, **{"encoding_blocks.9.tdf.0.weight": np.zeros((2, 2), np.float32)}))
    with pytest.raises(_lib.LemasError, match="wrong shape"):
        MdxEngine(MO.MINI, dict(sd, **{"first_conv.0.weight": np.zeros((8, 4), np.float32)}))
    with pytest.raises(_lib.LemasError, match="only 3"):
        MdxEngine(MO.MdxArch(dim_f=32, dim_t=16, num_blocks=5, l=2, g=8, k=5, bn=4), {})
    # module state the forward never reads is accepted, as load_state_dict(strict=True) would
    extra = dict(sd, window=np.zeros(64, np.float32), freq_pad=np.zeros((1, 4, 1, 16), np.float32))
    extra["first_conv.1.num_batches_tracked"] = np.zeros((), np.float32)
    MdxEngine(MO.MINI, extra)


@pytest.mark.gpu
@pytest.mark.parametrize("denoise", [False, True])
def test_denoiser_shell_with_the_hip_network(denoise):
    """Inference.demix_base (multiprocess_cuda_infer.py:276-301) with the HIP network inside against the shell oracle with the oracle
    network inside; is_denoise = the +-input average of :269."""
    from oracle.uvr5_oracle import ShellOracle
    from lemas_tts_amd.uvr5 import Inference, MDXConfig
    arch = MO.MdxArch(dim_f=64, dim_t=16, num_blocks=5, l=2, g=8, k=3, bn=4, bias=False)
    sd = MO.seeded_state_dict(arch, 3)
    net = MO.MdxOracle(arch, sd)
    o = ShellOracle(2048, 64, 4, is_denoise=denoise, mdx_batch_size=2, margin=441)
    o.model_run = lambda spek: net.forward(spek).numpy()
    mix = torch.from_numpy(np.random.default_rng(9).standard_normal((2, 30000)).astype(np.float32) * 0.3)
    ref = o.demix_base({0: mix}).numpy()
    cfg = MDXConfig(mdx_n_fft_scale_set=2048, mdx_dim_f_set=64, mdx_dim_t_set=4, compensate=1.0, is_denoise=denoise, mdx_batch_size=2, margin=441)
    inf = Inference(cfg, "cuda:0")
    inf.load_model((arch, sd))
    out = inf.demix_base({0: mix}).cpu().numpy()
    assert out.shape == ref.shape
    e = _rel(out, ref)
    print(f"\n[uvr5 shell + hip network, denoise {denoise}] rel err {e:.2e}")
    assert e < REL


@pytest.mark.gpu
def test_cli_denoise_runs_from_the_reference_directory_layout(tmp_path, monkeypatch):
    """``tts_multilingual --denoise`` (tts_multilingual.py:303-314) with no user-supplied callable: pretrained_models/uvr5 holds
    Kim_Vocal_1.onnx + MDX-Net-Kim-Vocal1.json + model_data.json (keyed by the reference's model hash), the prompt goes through the HIP
    denoiser, and the synthesis that follows reads the denoised temporary file (removed afterwards)."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from onnx_writer import convtdfnet_onnx
    from test_gpu_08_cli import _assets
    import lemas_tts_amd.api as A
    import lemas_tts_amd.scripts.tts_multilingual as M
    from lemas_tts_amd import synth
    from lemas_tts_amd.infer.audio_io import load_wav, save_wav
    from lemas_tts_amd.uvr5 import UVR5, mdx
    root = _assets(tmp_path, 1)
    monkeypatch.setattr(M, "PRETRAINED_ROOT", root)
    monkeypatch.setattr(M, "CKPTS_ROOT", root / "ckpts")
    real_cfg = A.load_arch_config
    monkeypatch.setattr(A, "load_arch_config", lambda m: {**real_cfg(m), "arch": {**real_cfg(m)["arch"], "depth": 1}})
    arch = MO.MdxArch(dim_f=64, dim_t=16, num_blocks=5, l=2, g=8, k=3, bn=4, bias=False)
    sd = MO.seeded_state_dict(arch, 3)
    uv = root / "uvr5"
    uv.mkdir()
    convtdfnet_onnx(str(uv / "Kim_Vocal_1.onnx"), arch, sd)
    (uv / "MDX-Net-Kim-Vocal1.json").write_text(json.dumps({"is_denoise": True, "mdx_batch_size": 2, "margin": 441, "chunks": 0, "model_name": "Kim_Vocal_1"}))
    (uv / "model_data.json").write_text(json.dumps({mdx.model_hash(str(uv / "Kim_Vocal_1.onnx")): {
        "compensate": 1.0, "mdx_dim_f_set": 64, "mdx_dim_t_set": 4, "mdx_n_fft_scale_set": 2048, "primary_stem": "Vocals"}}))
    t = np.arange(int(1.2 * 16000)) / 16000.0
    prompt = 0.05 * np.sin(2 * np.pi * 220 * t)[:, None] + 0.004 * np.random.default_rng(7).standard_normal((t.size, 1))
    save_wav(tmp_path / "ref.wav", prompt, 16000, "PCM_16")

    # the wrapper alone: resolves the directory, is_denoise on, output = shell oracle around the oracle network on the same 44.1 kHz input
    u = UVR5(str(uv), device="cuda:0")
    assert (u.model.dim_f, u.model.dim_t, u.model.n_fft, u.model.is_denoise, u.model.mdx_batch_size) == (64, 16, 2048, True, 2)
    wav, sr = load_wav(tmp_path / "ref.wav")
    from lemas_tts_amd.engine import resampler
    stereo = resampler(sr, 44100)(torch.cat((wav, wav)).to("cuda:0"))
    got = u.denoise(wav, sr).cpu().numpy()
    from oracle.uvr5_oracle import ShellOracle
    net = MO.MdxOracle(arch, sd)
    o = ShellOracle(2048, 64, 4, is_denoise=True, mdx_batch_size=2, margin=441)
    o.model_run = lambda spek: net.forward(spek).numpy()
    ref = o.demix_base({0: stereo.cpu()}).numpy()
    assert got.shape == ref.shape and _rel(got, ref) < REL

    seen = {}
    real_infer = A.TTS.infer

    def spy(self, **kw):
        seen["ref"] = kw["ref_file"]
        seen["audio"] = load_wav(kw["ref_file"])
        return real_infer(self, **kw)
    monkeypatch.setattr(A.TTS, "infer", spy)
    ref_ph = "|".join(f"p{i}" for i in synth.synth_tokens(74, 9, 60))
    out = tmp_path / "out.wav"
    rc = M.main(["--ref_audio", str(tmp_path / "ref.wav"), "--ref_phones", ref_ph, "--phones", ref_ph, "--output_wave", str(out), "--nfe_step", "2",
                 "--seed", "1", "--use_ema", "--denoise"])
    assert rc == 0 and out.is_file()
    assert seen["ref"] != str(tmp_path / "ref.wav") and not os.path.exists(seen["ref"])          # a temporary file, gone again
    den, dsr = seen["audio"]
    assert dsr == 44100 and den.shape == (2, got.shape[1])
    assert np.abs(den.numpy() - np.clip(got, -1, 1)).max() < 2.0 ** -22                           # the 24-bit file of the denoiser's output
