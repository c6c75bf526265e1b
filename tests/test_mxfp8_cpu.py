"""CPU tier: the MXFP8 quantisation scheme of the fp8 GEMM variant as the oracle restates it (oracle/mxfp8.py).
The GPU tier checks the HIP quantisers bit-for-bit against these functions (tests/test_gpu_02_fp8.py)."""
import numpy as np
import torch

from oracle.mxfp8 import mx_dequant, mx_exponent, mx_quant, w_quant


def test_exponent_is_smallest_power_of_two_that_fits():
    amax = torch.tensor([448.0, 448.0001, 224.0, 224.1, 1.0, 0.0, 1e-38, 3e38, 447.99997])
    e = mx_exponent(amax)
    assert e.tolist()[:5] == [0, 1, -1, 0, -8]
    assert e[5] == -120 and e[6] == -120 and e[7] == 120          # clamped: 2^e and 2^-e stay normal fp32
    ok = amax[:5] * torch.exp2(-e[:5].float()) <= 448.0
    assert bool(ok.all())
    assert bool((amax[:5] * torch.exp2(-(e[:5] - 1).float()) > 448.0).all())   # one less would overflow e4m3


def test_quant_dequant_error_bound_and_layout():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 256, generator=g) * torch.exp2(torch.randint(-10, 10, (37, 8, 1), generator=g).float()).expand(37, 8, 32).reshape(37, 256)
    q, mx, deq = mx_quant(x)
    assert q.shape == (37, 256) and q.dtype == torch.uint8 and mx.shape == (37, 8) and mx.dtype == torch.uint8
    assert torch.equal(mx_dequant(q, mx), deq)
    scale = torch.exp2(mx.float() - 127)[..., None].expand(37, 8, 32).reshape(37, 256)
    # half an e4m3 ulp: 2^-4 relative for normals, 2^-10 of the block scale in the subnormal range
    assert bool(((deq - x).abs() <= 0.0625 * x.abs() + scale * 2.0 ** -10).all())
    assert bool((deq.reshape(37, 8, 32).abs().amax(-1) <= 448.0 * scale.reshape(37, 8, 32)[..., 0]).all())
    # values exactly representable survive unchanged
    y = torch.tensor([[1.0, -2.0, 0.5, 448.0] * 8])
    assert torch.equal(mx_quant(y)[2], y)


def test_weight_quant_per_channel():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(16, 128, generator=g) * (1 + torch.arange(16)[:, None])
    w[3] = 0
    q, sc, deq = w_quant(w)
    assert sc[3] == 1.0 and bool((deq[3] == 0).all())
    np.testing.assert_allclose(sc.numpy(), (w.abs().amax(-1) * np.float32(1 / 448)).numpy().clip(min=0) + (sc == 1.0).float().numpy() * (w.abs().amax(-1) == 0).float().numpy(), rtol=1e-6)
    assert bool(((deq - w).abs() <= 0.0625 * w.abs() + sc[:, None] * 2.0 ** -10 + 1e-7).all())


def test_oracle_fp8_variant_is_close_to_fp32_and_not_equal():
    from lemas_tts_amd import synth
    from lemas_tts_amd.model.layout import DiTArch
    from oracle import lemas_oracle as O
    arch, vocab = DiTArch(depth=2), 898
    sd = synth.synth_cfm_state_dict(arch, vocab, 3)
    cond = torch.from_numpy(synth.synth_cond_mel(4, 24))[None]
    text = torch.from_numpy(synth.synth_tokens(5, 10, vocab))[None]
    y0 = torch.from_numpy(synth.synth_noise(6, 64))[None]
    kw = dict(y0=y0, steps=4, cfg_strength=2.0, sway_sampling_coef=5)
    a, _ = O.OracleCFM(sd, arch).sample(cond, text, 64, **kw)
    b, _ = O.OracleCFM(sd, arch, fp8=True).sample(cond, text, 64, **kw)
    mse = float(((a - b)[:, 24:] ** 2).mean())
    assert 0 < mse <= 1e-4, mse
