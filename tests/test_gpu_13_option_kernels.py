"""GPU tier, LAST file on purpose: kernel tests of the engine's NON-DEFAULT options -- the LayerNorm tail of the gate + residual GEMM (`ln_fused`)
and the LayerNorm folded across the GEMMs (`ln_fold`); DESIGN.md section 8b: both measured, neither ships.  They run after every product-path
test so that `pytest -x` reaches those first.  (One unexplained failure of `test_ln_fold_every_consumer_epilogue_and_tile[4-18-2048]` was seen
in the first pytest process of a fresh box and never again in 30 runs of the file / 900 repetitions of the case: LABLOG.md section 10b.)"""
import math

import numpy as np
import pytest
import torch

from conftest import needs_measurement_build
from test_gpu_01_kernels import EPI_GATE, _bf, _lib  # helpers of the kernel tests

pytestmark = pytest.mark.gpu


def _gate_ln_case(tile, batch, frames, K, seed, concurrent=1, ragged=False):
    L, lib = _lib()
    dev = "cuda:0"
    N = 1024
    pitch = (frames + 127) // 128 * 128
    M = batch * pitch
    g = torch.Generator(device=dev).manual_seed(seed)
    A = torch.randn(M, K, generator=g, device=dev)
    W = torch.randn(N, K, generator=g, device=dev) * 0.05
    bias = torch.randn(N, generator=g, device=dev)
    gate = torch.randn(N, generator=g, device=dev)
    sc, sh = torch.randn(N, generator=g, device=dev) * 0.3, torch.randn(N, generator=g, device=dev) * 0.3
    x0 = torch.randn(M, N, generator=g, device=dev) * 2 + 0.25
    lens_d = None
    if ragged:
        lens_d = torch.randint(1, frames + 1, (batch,), generator=g, device=dev, dtype=torch.int32)
        lens_d[-1] = frames
    lp = lens_d.data_ptr() if lens_d is not None else None
    # (a) the fused launch
    x, h = x0.clone(), torch.zeros(M, N, device=dev)
    L.check(lib.lemas_k_gemm_gate_ln(tile, A.data_ptr(), W.data_ptr(), bias.data_ptr(), gate.data_ptr(), sc.data_ptr(), sh.data_ptr(), lp,
                                     x.data_ptr(), h.data_ptr(), batch, pitch, frames, K, concurrent, None), f"gate_ln tile {tile}")
    # (b) the same tile without the tail, then the stand-alone LayerNorm kernel on its result
    x2, h2 = x0.clone(), torch.zeros(M, N, device=dev)
    L.check(lib.lemas_k_gemm_epi(EPI_GATE, tile, A.data_ptr(), W.data_ptr(), bias.data_ptr(), gate.data_ptr(), lp, x2.data_ptr(),
                                 batch, pitch, frames, N, K, None))
    L.check(lib.lemas_k_ln_mod(x2.data_ptr(), sc.data_ptr(), sh.data_ptr(), h2.data_ptr(), M, N, None))
    assert torch.equal(x, x2), (tile, float((x - x2).abs().max()))
    assert torch.equal(h, h2), (tile, float((h - h2).abs().max()), int((h != h2).any(dim=1).sum()))
    # (c) fp32 torch on the bf16-rounded operands
    acc = (_bf(A).double() @ _bf(W).double().T).float() + bias
    pos, sample = torch.arange(M, device=dev) % pitch, torch.arange(M, device=dev) // pitch
    live = pos < frames
    if lens_d is not None:
        live = live & (pos < lens_d[sample])
    xr = torch.where(live[:, None], x0 + gate * acc, x0)
    assert float((x - xr).abs().max()) < 4e-3 * math.sqrt(K / 64)
    hr = torch.nn.functional.layer_norm(x, (N,), eps=1e-6) * (1 + sc) + sh       # on the kernel's own x: isolates the tail
    assert float(((h - _bf(hr)).abs() / hr.abs().clamp(min=1.0)).max()) < 1.01 / 128


@needs_measurement_build
@pytest.mark.parametrize("tile", [17, 18, 19, 26])
@pytest.mark.parametrize("K", [1024, 2048])
def test_gemm_gate_with_layernorm_tail_every_tile(tile, K):
    """one CFG lane of configs[1] (1875 frames in a 1920-row pitch): out-projection (K = 1024) and FF2 (K = 2048)"""
    _gate_ln_case(tile, 1, 1875, K, seed=tile * 7 + K)


@needs_measurement_build
@pytest.mark.parametrize("tile,batch,frames", [(17, 1, 1875), (19, 1, 750), (18, 1, 750), (17, 2, 900)])
def test_gemm_gate_with_layernorm_tail_two_concurrent_lanes(tile, batch, frames):
    """two launches at once on two streams, as the CFG lanes run them: each waits only for its own panels; identical results"""
    _gate_ln_case(tile, batch, frames, 2048, seed=tile + frames, concurrent=2, ragged=batch > 1)


@needs_measurement_build
def test_gemm_gate_with_layernorm_tail_repeated_under_load():
    """the tail's hand-off (write-through stores, drained, one arrival per workgroup, relaxed poll, L1-bypassing loads) repeated with
    fresh data while a second pair of launches keeps the chip busy: a stale row would show as a bit difference against the
    stand-alone kernel"""
    for it in range(12):
        _gate_ln_case(17, 1, 1875, 1024 if it % 2 else 2048, seed=1000 + it, concurrent=2 + (it % 2))


def test_gemm_gate_layernorm_tail_refuses_tiles_without_it():
    """(in the product build: refuses every tile -- the tail's device code is not there)"""
    L, lib = _lib()
    dev = "cuda:0"
    a, w = torch.zeros(256, 1024, device=dev), torch.zeros(1024, 1024, device=dev)
    x, h = torch.zeros(256, 1024, device=dev), torch.zeros(256, 1024, device=dev)
    v = torch.zeros(1024, device=dev)
    for tile in (16, 22):
        rc = lib.lemas_k_gemm_gate_ln(tile, a.data_ptr(), w.data_ptr(), v.data_ptr(), v.data_ptr(), v.data_ptr(), v.data_ptr(), None,
                                      x.data_ptr(), h.data_ptr(), 1, 256, 256, 1024, 1, None)
        assert rc != 0, tile


def _ln_fold_case(prod_tile, cons_epi, cons_tile, batch, frames, Kp, Nc, seed, use_prep=False, ragged=False, mean_shift=0.25, tol_b=6e-2):
    """ln fold (csrc/common.h GemmParams): producer GEMM -> table rows -> consumer GEMM against (a) an fp32 emulation of the SAME
    arithmetic (tight: isolates implementation errors) and (b) the textbook LayerNorm-modulate -> Linear of modules.py:627-641 on the
    kernel's own x (loose: bf16 rounding of a different intermediate)."""
    L, lib = _lib()
    dev = "cuda:0"
    D = 1024
    pitch = (frames + 127) // 128 * 128
    M = batch * pitch
    g = torch.Generator(device=dev).manual_seed(seed)
    A = torch.randn(M, Kp, generator=g, device=dev)
    Wp = torch.randn(D, Kp, generator=g, device=dev) * 0.05
    bias_p = torch.randn(D, generator=g, device=dev)
    gate = torch.randn(D, generator=g, device=dev)
    sc, sh = torch.randn(D, generator=g, device=dev) * 0.3, torch.randn(D, generator=g, device=dev) * 0.3
    Wc = torch.randn(Nc, D, generator=g, device=dev) * 0.05
    bias_c = torch.randn(Nc, generator=g, device=dev)
    x0 = torch.randn(M, D, generator=g, device=dev) * 2 + mean_shift
    pos_t = torch.arange(frames, device=dev, dtype=torch.float32)[:, None] * (10000.0 ** (-torch.arange(32, device=dev) / 32.0))[None]
    rope = torch.cat([pos_t.cos(), pos_t.sin()]).contiguous()
    lens_d = None
    if ragged:
        lens_d = torch.randint(1, frames + 1, (batch,), generator=g, device=dev, dtype=torch.int32)
        lens_d[-1] = frames
    lp = lens_d.data_ptr() if lens_d is not None else None
    x = x0.clone()
    y = torch.zeros(M * Nc, device=dev)
    L.check(lib.lemas_k_ln_fold_pair(prod_tile, cons_epi, cons_tile, A.data_ptr(), Wp.data_ptr(), bias_p.data_ptr(), gate.data_ptr(), sc.data_ptr(),
                                     sh.data_ptr(), Wc.data_ptr(), bias_c.data_ptr(), rope.data_ptr(), lp, x.data_ptr(), y.data_ptr(), batch, pitch,
                                     frames, Kp, Nc, 1 if use_prep else 0, None), f"ln_fold_pair {prod_tile}/{cons_epi}/{cons_tile}")
    pos, sample = torch.arange(M, device=dev) % pitch, torch.arange(M, device=dev) // pitch
    valid = pos < frames
    if use_prep:
        assert torch.equal(x, x0)
    else:       # the residual update itself is the plain gate + residual epilogue
        acc = (_bf(A).double() @ _bf(Wp).double().T).float() + bias_p
        live = valid if lens_d is None else valid & (pos < lens_d[sample])
        xr = torch.where(live[:, None], x0 + gate * acc, x0)
        assert float((x - xr).abs().max()) < 4e-3 * math.sqrt(Kp / 64)
    xd = x.double()
    mu, var = xd.mean(1, keepdim=True), xd.var(1, unbiased=False, keepdim=True)
    r = (var + 1e-6).rsqrt()
    Wcb = _bf(Wc).double()
    # (a) the fold's own arithmetic: bf16(x (1 + s)) . W^T scaled by r, minus r mu c1, plus c2
    xs = _bf(x * (1 + sc)).double()
    c1 = ((1 + sc).double()[None] @ Wcb.T)[0]
    c2 = (sh.double()[None] @ Wcb.T)[0] + bias_c.double()
    pre_a = (r * (xs @ Wcb.T) - r * mu * c1 + c2).float()
    # (b) the textbook form
    h = ((xd - mu) * r * (1 + sc.double()) + sh.double()).float()
    pre_b = (_bf(h).double() @ Wcb.T).float() + bias_c

    def finish(pre):
        if cons_epi == 1:
            return torch.nn.functional.gelu(pre, approximate="tanh")[valid]
        H = Nc // (128 if cons_epi == 4 else 64)
        if cons_epi == 5:      # v^T [batch][H][64][pitch]
            return pre.view(batch, pitch, H, 64).permute(0, 2, 3, 1)[..., :frames]
        qk = pre.view(batch, pitch, 2, H, 32, 2)[:, :frames]
        cs, sn = rope[:frames, None, None, :].unsqueeze(0), rope[frames:, None, None, :].unsqueeze(0)
        a, b = qk[..., 0], qk[..., 1]
        rot = torch.stack([a * cs - b * sn, b * cs + a * sn], dim=-1).reshape(batch, frames, 2, H, 64)
        return rot.permute(2, 0, 3, 1, 4)          # [q|k][batch][H][frames][64]

    if cons_epi == 1:
        got = y.view(M, Nc)[valid]
    elif cons_epi == 5:
        got = y.view(batch, Nc // 64, 64, pitch)[..., :frames]
    else:
        got = y.view(2, batch, Nc // 128, pitch, 64)[:, :, :, :frames]
    ra, rb = finish(pre_a), finish(pre_b)
    scale_ref = rb.abs().clamp(min=1.0)
    ea = float(((got - ra).abs() / scale_ref).max())
    eb = float(((got - rb).abs() / scale_ref).max())
    assert ea < 1.2 / 128, (prod_tile, cons_epi, cons_tile, ea)         # one bf16 output rounding (2^-8 relative) + fp32 order effects
    assert eb < tol_b, (prod_tile, cons_epi, cons_tile, eb)             # bf16 rounding of x (1 + s) instead of LN(x) (1 + s) + b, K = 1024
    return ea, eb


@pytest.mark.parametrize("prod_tile", [16, 17, 18, 19, 22, 26, 28, 30, 31])
@pytest.mark.parametrize("Kp", [1024, 2048])
def test_ln_fold_every_producer_tile(prod_tile, Kp):
    """the gate + residual GEMM writing the scaled bf16 rows + row partial sums, on every tile the sampler can pick for it"""
    _ln_fold_case(prod_tile, 1, 26, 1, 1875, Kp, 2048, seed=prod_tile * 3 + Kp)


@pytest.mark.parametrize("cons_epi,cons_tile,Nc", [(1, 16, 2048), (1, 17, 2048), (1, 18, 2048), (1, 19, 2048), (1, 22, 2048), (1, 26, 2048), (1, 28, 2048), (1, 30, 2048), (4, 28, 2048), (4, 30, 2048), (5, 28, 1024), (5, 31, 1024),
                                                   (4, 16, 2048), (4, 17, 2048), (4, 18, 2048), (4, 22, 2048), (4, 26, 2048),
                                                   (5, 16, 1024), (5, 17, 1024), (5, 18, 1024), (5, 19, 1024), (5, 22, 1024), (5, 26, 1024)])
def test_ln_fold_every_consumer_epilogue_and_tile(cons_epi, cons_tile, Nc):
    """FF1 (GELU), QK (+RoPE) and V^T epilogues applying the row statistics, on every tile"""
    _ln_fold_case(17, cons_epi, cons_tile, 1, 1875, 1024, Nc, seed=cons_epi * 100 + cons_tile)


@pytest.mark.parametrize("batch,frames", [(1, 750), (3, 700), (8, 1125)])
def test_ln_fold_batched_ragged_and_chain_entry(batch, frames):
    _ln_fold_case(0, 1, 0, batch, frames, 2048, 2048, seed=frames, ragged=batch > 1)
    _ln_fold_case(0, 5, 0, batch, frames, 1024, 1024, seed=frames + 1, ragged=batch > 1)
    _ln_fold_case(0, 4, 0, batch, frames, 1024, 2048, seed=frames + 2, use_prep=True)


def test_ln_fold_rows_with_a_large_mean():
    """rows whose mean is 10x their spread: the statistics (E[x^2] - mean^2 in fp32) lose ~7 bits and stay inside the bf16 budget (tight
    check unchanged); against the textbook form the bf16 rounding of x (1 + s) is sqrt(1 + (mean/std)^2) ~ 10x that of LN(x) (1 + s) + b --
    the known price of the fold for rows with a DC offset"""
    ea, eb = _ln_fold_case(17, 1, 26, 1, 512, 1024, 2048, seed=5, mean_shift=20.0, tol_b=0.5)
    print(f"\n[ln fold, mean/std = 10] emulation err {ea:.2e}, textbook err {eb:.2e}")
