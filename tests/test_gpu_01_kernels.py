"""GPU tier, kernel level: each hand-written HIP kernel through its C-ABI entry point (lemas_k_*) against a
plain fp32 torch reference of the same op.  Tolerances are stated per test; bf16-operand kernels are compared
against fp32 math on bf16-ROUNDED inputs so that only accumulation order / output rounding remain."""
import math

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from lemas_tts_amd import _lib as L
    return L, L.testlib()


def _dev(t):
    return t.to("cuda:0", torch.float32).contiguous()


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("M,N,K,act", [(128, 128, 64, 0), (300, 384, 1024, 0), (517, 2048, 1024, 1), (1875, 100, 1024, 0),
                                        (130, 1024, 2048, 0)])
def test_linear_bf16(M, N, K, act):
    L, lib = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    # asymmetric, non-uniform weights: catches row<->col swaps in the MFMA C-fragment mapping
    W = torch.randn(N, K, generator=g) * 0.05 + (torch.arange(N)[:, None] % 7 - 3) * 0.01
    b = torch.randn(N, generator=g)
    ref = _bf(A) @ _bf(W).T + b
    if act == 1:
        ref = _bf(torch.nn.functional.gelu(ref, approximate="tanh"))
    Ad, Wd, bd = _dev(A), _dev(W), _dev(b)
    out = torch.empty(M, N, device="cuda:0")
    L.check(lib.lemas_k_linear_bf16(Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), out.data_ptr(), M, N, K, act, None))
    err = (out.cpu() - ref).abs().max().item()
    tol = 2e-2 if act == 1 else 2e-3 * math.sqrt(K / 64)   # fp32 accumulate of bf16 products, |ref| ~ sqrt(K)*0.05
    assert err < tol, err


@pytest.mark.parametrize("M,N,K,act", [(64, 64, 16, 0), (100, 1026, 512, 0), (333, 1024, 700, 1), (32, 6144, 1024, 2),
                                        (1875, 1024, 100, 0)])
def test_linear_f32(M, N, K, act):
    L, lib = _lib()
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
    ref = (A.double() @ W.double().T + b.double())
    if act == 1:
        ref = torch.nn.functional.gelu(ref)
    if act == 2:
        ref = torch.nn.functional.silu(ref)
    Ad, Wd, bd = _dev(A), _dev(W), _dev(b)
    out = torch.empty(M, N, device="cuda:0")
    L.check(lib.lemas_k_linear_f32(Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), out.data_ptr(), M, N, K, act, None))
    err = (out.cpu().double() - ref).abs().max().item()
    assert err < 2e-5 * math.sqrt(K / 16), err          # exact-fp32 MFMA == fmaf chain


@pytest.mark.parametrize("B,H,N,lens", [(1, 2, 64, None), (2, 16, 200, None), (2, 4, 333, [333, 210]), (1, 16, 1875, None),
                                         (3, 2, 130, [1, 64, 130]), (1, 1, 65, None), (2, 2, 192, [129, 192])])
def test_attention(B, H, N, lens):
    L, lib = _lib()
    g = torch.Generator().manual_seed(N + H)
    q, k, v = (torch.randn(B, H, N, 64, generator=g) for _ in range(3))
    k[0, 0, N // 2] *= 4.0        # a spiky key: exercises the online-softmax rescale
    qb, kb, vb = _bf(q), _bf(k), _bf(v)
    s = (qb @ kb.transpose(-1, -2)) / 8.0
    lens_d = None
    if lens is not None:
        m = torch.arange(N)[None, :] < torch.tensor(lens)[:, None]
        s = s.masked_fill(~m[:, None, None, :], float("-inf"))
        lens_d = torch.tensor(lens, dtype=torch.int32, device="cuda:0")
    ref = (torch.softmax(s, -1) @ vb).transpose(1, 2).reshape(B, N, H * 64)
    out = torch.empty(B, N, H * 64, device="cuda:0")
    qd, kd, vd = _dev(q), _dev(k), _dev(v)
    L.check(lib.lemas_k_attention(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), lens_d.data_ptr() if lens_d is not None else None,
                                  out.data_ptr(), B, H, N, None))
    err = (out.cpu() - ref).abs().max().item()
    assert err < 2e-2, err          # P and O rounded to bf16 (8 mantissa bits) on |v| ~ 1..4


def test_ln_mod():
    L, lib = _lib()
    g = torch.Generator().manual_seed(5)
    M, D = 777, 1024
    x = torch.randn(M, D, generator=g) * 3 + 0.5
    sc, sh = torch.randn(D, generator=g) * 0.3, torch.randn(D, generator=g) * 0.3
    ref = _bf(torch.nn.functional.layer_norm(x, (D,), eps=1e-6) * (1 + sc) + sh)
    out = torch.empty(M, D, device="cuda:0")
    xd, scd, shd = _dev(x), _dev(sc), _dev(sh)
    L.check(lib.lemas_k_ln_mod(xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), out.data_ptr(), M, D, None))
    err = (out.cpu() - ref).abs().max().item()
    assert err < 4e-2, err          # one bf16 ulp at |x| ~ 4 is 3.1e-2


@pytest.mark.parametrize("B,N", [(1, 100), (2, 333), (1, 1875)])
def test_convpos(B, N):
    L, lib = _lib()
    g = torch.Generator().manual_seed(N)
    C_, G, T = 1024, 16, 31
    x = torch.randn(B, N, C_, generator=g)
    w1, w2 = (torch.randn(C_, C_ // G, T, generator=g) * 0.02 for _ in range(2))
    b1, b2 = (torch.randn(C_, generator=g) * 0.02 for _ in range(2))
    F = torch.nn.functional
    h = _bf(F.mish(F.conv1d(_bf(x).transpose(1, 2), _bf(w1), b1, padding=T // 2, groups=G)))
    ref = F.mish(F.conv1d(h, _bf(w2), b2, padding=T // 2, groups=G)).transpose(1, 2) + x
    out = torch.empty(B, N, C_, device="cuda:0")
    args = [_dev(t) for t in (x, w1, b1, w2, b2)]
    L.check(lib.lemas_k_convpos(*[a.data_ptr() for a in args], out.data_ptr(), B, N, C_, G, T, None))
    err = (out.cpu() - ref).abs().max().item()
    assert err < 5e-3, err


@pytest.mark.parametrize("seed", list(range(20)))
def test_randomised_gemm_and_attention_shapes(seed):
    """tile-edge hunting: random M (incl. 1 and just around multiples of 32 / 128 / 256), N, K for the bf16 GEMM; random
    sequence lengths, head counts and ragged key lengths for the attention kernel"""
    import numpy as np
    L, lib = _lib()
    rng = np.random.default_rng(300 + seed)
    edge = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 511, 513]
    M = int(rng.choice(edge)) if rng.random() < 0.5 else int(rng.integers(1, 2500))
    act = int(rng.integers(0, 2))
    N = int(rng.choice([4, 12, 100, 128, 132, 252, 256, 1024, 1100])) if act == 0 else int(rng.choice([8, 24, 104, 128, 136, 248, 256, 1024, 1096]))
    K = 64 * int(rng.integers(1, 33))
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) * 0.05 + (torch.arange(N)[:, None] % 5 - 2) * 0.01
    b = torch.randn(N, generator=g)
    ref = _bf(A) @ _bf(W).T + b
    if act == 1:
        ref = _bf(torch.nn.functional.gelu(ref, approximate="tanh"))
    Ad, Wd, bd = _dev(A), _dev(W), _dev(b)
    out = torch.empty(M, N, device="cuda:0")
    L.check(lib.lemas_k_linear_bf16(Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), out.data_ptr(), M, N, K, act, None))
    err = (out.cpu() - ref).abs().max().item()
    assert err < (2e-2 if act == 1 else 2e-3 * math.sqrt(K / 64)) * max(1.0, float(ref.abs().max()) / 4), (M, N, K, act, err)

    # widths the 16-byte-chunk epilogues cannot store are refused, not mangled
    bad = torch.empty(M, 7, device="cuda:0")
    assert lib.lemas_k_linear_bf16(Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), bad.data_ptr(), M, 7, K, 0, None) != 0

    B, H = int(rng.integers(1, 4)), int(rng.choice([1, 2, 16]))
    Ns = int(rng.choice(edge[5:])) if rng.random() < 0.5 else int(rng.integers(2, 700))
    lens = None if rng.random() < 0.5 else [int(rng.integers(1, Ns + 1)) for _ in range(B)]
    q, k, v = (torch.randn(B, H, Ns, 64, generator=g) for _ in range(3))
    qb, kb, vb = _bf(q), _bf(k), _bf(v)
    mask = None
    if lens is not None:
        mask = (torch.arange(Ns)[None, :] < torch.tensor(lens)[:, None])[:, None, None, :]
    refa = torch.nn.functional.scaled_dot_product_attention(qb, kb, vb, attn_mask=mask).transpose(1, 2).reshape(B, Ns, H * 64)
    qd, kd, vd = _dev(q), _dev(k), _dev(v)
    ld = None if lens is None else torch.tensor(lens, dtype=torch.int32, device="cuda:0")
    o = torch.empty(B, Ns, H * 64, device="cuda:0")
    L.check(lib.lemas_k_attention(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), ld.data_ptr() if ld is not None else None, o.data_ptr(), B, H, Ns, None))
    d = (o.cpu() - refa).abs()
    if lens is not None:                        # rows past a sample's length are unspecified
        for bi in range(B):
            d[bi, lens[bi]:] = 0
    assert float(d.max()) < 3e-2, (B, H, Ns, lens, float(d.max()))


# ---------------------------------------------------------------------------------------------------------------------
# Every (tile shape, epilogue) pair the sampler's dispatch can pick, at the row counts where it picks them: 128x64 / 128x128 /
# 256x128 at batch 1, 256x256 for batched shapes (M >= 8192 rows per lane: config3's 15360, 8 x configs[1]'s 30720).  The
# round-1 suite never reached the 256x256 tile from a unit test.  References are fp64 matmuls of the bf16-ROUNDED operands on
# the device (exact to fp32 rounding), then the epilogue in fp32 torch.
EPI_BF16, EPI_GELU, EPI_F32, EPI_GATE, EPI_QK, EPI_VT = 0, 1, 2, 3, 4, 5


def _gemm_epi_case(tile, epi, batch, frames, N, K, seed, ragged=False):
    L, lib = _lib()
    dev = "cuda:0"
    pitch = (frames + 127) // 128 * 128
    M = batch * pitch
    g = torch.Generator(device=dev).manual_seed(seed)
    A = torch.randn(M, K, generator=g, device=dev)
    W = torch.randn(N, K, generator=g, device=dev) * 0.05 + ((torch.arange(N, device=dev)[:, None] % 7) - 3) * 0.01
    bias = torch.randn(N, generator=g, device=dev)
    acc = (_bf(A).double() @ _bf(W).double().T).float() + bias                     # [M, N]
    pos = torch.arange(M, device=dev) % pitch
    sample = torch.arange(M, device=dev) // pitch
    valid = pos < frames
    aux, lens_d, out = None, None, None
    if epi == EPI_F32:
        out = torch.zeros(M, N, device=dev)
        ref, tol = acc, 2e-3 * math.sqrt(K / 64)
    elif epi == EPI_BF16:
        out = torch.zeros(M, N, device=dev)
        ref, tol = _bf(acc), 1.01 / 128
    elif epi == EPI_GELU:
        out = torch.zeros(M, N, device=dev)
        ref, tol = _bf(torch.nn.functional.gelu(acc, approximate="tanh")), 1.5 / 128   # + the v_exp / v_rcp activation
    elif epi == EPI_GATE:
        aux = torch.randn(N, generator=g, device=dev)
        x0 = torch.randn(M, N, generator=g, device=dev)
        out = x0.clone()
        live = valid
        if ragged:
            lens = torch.randint(1, frames + 1, (batch,), generator=g, device=dev, dtype=torch.int32)
            lens[-1] = frames
            lens_d = lens
            live = valid & (pos < lens[sample])
        ref = torch.where(live[:, None], x0 + aux * acc, x0)
        tol = 4e-3 * math.sqrt(K / 64)
    elif epi == EPI_QK:
        H = N // 128
        inv = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, device=dev).float() / 64))
        ang = torch.arange(frames, device=dev).float()[:, None] * inv[None, :]      # [frames, 32]
        aux = torch.cat([ang.cos().reshape(-1), ang.sin().reshape(-1)]).contiguous()
        out = torch.zeros(2, batch, H, pitch, 64, device=dev)
        x = acc.view(batch, pitch, 2, H, 32, 2)[:, :frames]                         # (.., pair index, 2)
        c, s_ = ang.cos()[None, :, None, None, :], ang.sin()[None, :, None, None, :]
        rot = torch.stack([x[..., 0] * c - x[..., 1] * s_, x[..., 1] * c + x[..., 0] * s_], dim=-1)   # interleaved pairs
        ref = torch.zeros_like(out)
        ref[:, :, :, :frames] = _bf(rot.reshape(batch, frames, 2, H, 64)).permute(2, 0, 3, 1, 4)
        tol = 1.5 / 128
    elif epi == EPI_VT:
        H = N // 64
        out = torch.zeros(batch, H, 64, pitch, device=dev)
        ref = _bf(acc.view(batch, pitch, H, 64)).permute(0, 2, 3, 1).contiguous()
        tol = 1.01 / 128
    rc = lib.lemas_k_gemm_epi(epi, tile, A.data_ptr(), W.data_ptr(), bias.data_ptr(), aux.data_ptr() if aux is not None else None,
                              lens_d.data_ptr() if lens_d is not None else None, out.data_ptr(), batch, pitch, frames, N, K, None)
    L.check(rc, f"lemas_k_gemm_epi(epi {epi}, tile {tile})")
    d = (out - ref).abs()
    if epi not in (EPI_F32, EPI_GATE):
        d = d / ref.abs().clamp(min=1.0)       # bf16 outputs: the tolerance is in units of one bf16 ulp of the value (2^-7 relative)
    if epi in (EPI_F32, EPI_BF16, EPI_GELU):
        d = d[valid]                       # padding rows of the row space are never stored
    elif epi == EPI_QK:
        d = d[:, :, :, :frames]
    elif epi == EPI_VT:
        d = d[..., :frames]                # padding COLUMNS of v^T may hold anything finite (masked keys)
        assert torch.isfinite(out).all()
    err = float(d.max())
    assert err < tol, (tile, epi, batch, frames, N, K, err)


@pytest.mark.parametrize("tile", [16, 17, 18, 19, 22, 26, 27, 28, 29, 30, 31])
@pytest.mark.parametrize("epi,N,K", [(EPI_F32, 1024, 1024), (EPI_GELU, 2048, 1024), (EPI_GATE, 1024, 2048), (EPI_QK, 2048, 1024),
                                     (EPI_VT, 1024, 1024), (EPI_BF16, 1024, 1024)])
def test_gemm_every_tile_and_epilogue_batch1(tile, epi, N, K):
    """one utterance (frames 1875 in a 1920-row pitch: a ragged last tile for every tile height)"""
    _gemm_epi_case(tile, epi, 1, 1875, N, K, seed=tile * 10 + epi)


@pytest.mark.parametrize("tile", [16, 22])
@pytest.mark.parametrize("epi,N,K", [(EPI_GELU, 2048, 1024), (EPI_GATE, 1024, 2048), (EPI_QK, 2048, 1024), (EPI_VT, 1024, 1024),
                                     (EPI_F32, 1024, 1024)])
@pytest.mark.parametrize("batch,frames", [(8, 1900), (16, 1875), (5, 1601)])
def test_gemm_batched_row_counts(tile, epi, N, K, batch, frames):
    """M = 15360 (config3's lane), 30720 (8 x configs[1]), 8320 (>= 8192, odd tile count): the shapes where the dispatch
    switches to 256x256 tiles; gate/residual with ragged per-sample lengths"""
    _gemm_epi_case(tile, epi, batch, frames, N, K, seed=batch * 100 + tile + epi, ragged=True)


@pytest.mark.parametrize("plain,loaders", [(17, 27), (26, 28), (26, 29), (18, 30), (19, 31)])
@pytest.mark.parametrize("epi,N,K,batch,frames", [(EPI_GATE, 1024, 1024, 1, 1875), (EPI_GATE, 1024, 2048, 3, 700), (EPI_GELU, 2048, 1024, 1, 1875),
                                                   (EPI_QK, 2048, 1024, 2, 333), (EPI_VT, 1024, 1024, 1, 130), (EPI_F32, 1024, 64, 1, 100)])
def test_gemm_loader_wave_tiles_are_bit_identical(plain, loaders, epi, N, K, batch, frames):
    """tiles 27 / 28 = tiles 17 / 26 with four loader waves behind the compute waves (gemm_bf16.hip gemm_body NL): the same ring, LDS image, barriers
    and MFMA order, so every output bit must be the same -- incl. K = 64 (one K-tile: the loaders' prologue only) and ragged batches"""
    L, lib = _lib()
    dev = "cuda:0"
    pitch = (frames + 127) // 128 * 128
    M = batch * pitch
    g = torch.Generator(device=dev).manual_seed(plain * 1000 + epi * 10 + batch)
    A = torch.randn(M, K, generator=g, device=dev)
    W = torch.randn(N, K, generator=g, device=dev) * 0.05
    bias = torch.randn(N, generator=g, device=dev)
    aux = lens = None
    if epi == EPI_GATE:
        aux = torch.randn(N, generator=g, device=dev)
        lens = torch.randint(1, frames + 1, (batch,), generator=g, device=dev, dtype=torch.int32)
    elif epi == EPI_QK:
        ang = torch.arange(frames, device=dev).float()[:, None] * (1.0 / (10000.0 ** (torch.arange(0, 64, 2, device=dev).float() / 64)))[None, :]
        aux = torch.cat([ang.cos().reshape(-1), ang.sin().reshape(-1)]).contiguous()
    x0 = torch.randn(M, N, generator=g, device=dev)
    outs = []
    for tile in (plain, loaders):
        out = x0.clone() if epi in (EPI_GATE, EPI_F32, EPI_GELU) else torch.zeros(2 * M * N if epi == EPI_QK else M * N, device=dev)
        L.check(lib.lemas_k_gemm_epi(epi, tile, A.data_ptr(), W.data_ptr(), bias.data_ptr(), aux.data_ptr() if aux is not None else None,
                                     lens.data_ptr() if lens is not None else None, out.data_ptr(), batch, pitch, frames, N, K, None), f"tile {tile}")
        outs.append(out)
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())


def test_gemm_production_choice_matches_explicit_tiles():
    """tile 0 (the dispatch heuristic) must give the same numbers as the tile it picks: bit-identical to one of them"""
    _gemm_epi_case(0, EPI_GELU, 8, 1900, 2048, 1024, seed=7)
    _gemm_epi_case(0, EPI_GATE, 1, 375, 1024, 1024, seed=8)


# ---------------------------------------------------------------------------------------------------------------------
# The gate + residual GEMM with its LayerNorm-modulate tail (gemm_bf16.hip ln_tail): x must equal the plain epilogue-3 launch bit for
# bit, h must equal the stand-alone ln_mod kernel run on that x bit for bit (same row arithmetic, ln_core.h), and both must match fp32
# torch.  `concurrent` copies run at once on separate streams (the two CFG lanes) and must agree with each other.
@pytest.mark.parametrize("variant", [0, 4113])
@pytest.mark.parametrize("B,H,N", [(8, 16, 1900), (16, 16, 1875), (3, 16, 2814), (2, 16, 130), (1, 16, 64), (5, 16, 257)])
def test_attention_large_ragged_batches(B, H, N, variant):
    """BH = 128 / 256 per launch with ragged key lengths at N ~ 1900 (config3's lane) and config 5's N = 2814; short sequences around
    the 256-query workgroup of the q64 kernel (one key tile, one and a bit query blocks)"""
    L, lib = _lib()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(B * 1000 + N)
    q, k, v = (torch.randn(B, H, N, 64, generator=g, device=dev) for _ in range(3))
    k[:, :, N // 3] *= 4.0
    lens = torch.randint(N // 3, N + 1, (B,), generator=g, device=dev, dtype=torch.int32)
    lens[0], lens[-1] = N, max(1, N // 2 + 1)
    out = torch.empty(B, N, H * 64, device=dev)
    L.check(lib.lemas_k_attention_variant(q.data_ptr(), k.data_ptr(), v.data_ptr(), lens.data_ptr(), out.data_ptr(), B, H, N, variant, None))
    pres = bool(variant & 16)          # "prescaled q" variants round q * scale * log2(e) to bf16 and work in base 2
    c = float(np.float32(0.125) * np.float32(1.4426950408889634))
    qb, kb, vb = (_bf(q * c) if pres else _bf(q)), _bf(k), _bf(v)
    worst = 0.0
    for b in range(B):      # one sample at a time: the score matrix of a sample is H x N x N fp32
        n = int(lens[b])
        s = (qb[b, :, :n] @ kb[b, :, :n].transpose(-1, -2)) * (math.log(2.0) if pres else 0.125)
        ref = (torch.softmax(s, -1) @ vb[b, :, :n]).transpose(0, 1).reshape(n, H * 64)
        worst = max(worst, float((out[b, :n] - ref).abs().max()))
    assert worst < 3e-2, worst


# ---------------------------------------------------------------------------------------------------------------------
# Attention schedule variants (csrc/attention.hip VAR): same contract as the original kernel, plus inputs that FORCE the rare paths of
# the sum-checked softmax (variant bit 1): a key whose score jumps far above the running max in a late tile (the overflow guard must
# send that tile through the classical path), scores so large that exp2 overflows to inf, and rows whose later scores sit far BELOW
# the first tile's maximum (everything after underflows: must equal the reference, not NaN).
def _attn_ref(q, k, v, lens, prescaled=False):
    """fp64 softmax attention on the operands as the kernel sees them: q, k, v rounded to bf16 -- for the "prescaled q" variants q is
    multiplied by softmax_scale * log2(e) in fp32 BEFORE the rounding (what the QK GEMM epilogue does), and the scores are base-2"""
    kb, vb = _bf(k), _bf(v)
    B, H, N, _ = q.shape
    out = torch.zeros(B, N, H * 64, device=q.device)
    c = float(np.float32(0.125) * np.float32(1.4426950408889634))
    qb = _bf(q * c) if prescaled else _bf(q)
    for b in range(B):
        n = int(lens[b]) if lens is not None else N
        s = (qb[b, :, :, :].double() @ kb[b, :, :n].double().transpose(-1, -2))
        s = s * math.log(2.0) if prescaled else s / 8.0
        out[b] = (torch.softmax(s, -1) @ vb[b, :, :n].double()).float().transpose(0, 1).reshape(N, H * 64)
    return out


# 4096 + 16 + s: the 64-queries-per-wave kernel (csrc/attention_q64.hip), s = schedule bits
@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 7, 17, 19, 17 + 1024, 19 + 1024, 4112, 4113, 4114, 4115])
@pytest.mark.parametrize("case", ["plain", "late_spike", "overflow", "early_peak", "ragged", "deep_negative"])
def test_attention_variants(variant, case):
    L, lib = _lib()
    dev = "cuda:0"
    B, H, N = (3, 16, 777) if case == "ragged" else (1, 16, 1000)
    g = torch.Generator(device=dev).manual_seed(("plain", "late_spike", "overflow", "early_peak", "ragged", "deep_negative").index(case) * 100 + variant)
    q, k, v = (torch.randn(B, H, N, 64, generator=g, device=dev) for _ in range(3))
    lens = None
    if case == "late_spike":        # one key in a late tile scores ~40 nats above everything before it, for a quarter of the queries
        k[:, :, 900] = q[:, :, 100] * 5.0
        q[:, :, 100:350] = q[:, :, 100:101] + 0.05 * q[:, :, 100:350]
    elif case == "overflow":        # |score| in the hundreds: exp2 of the difference to the first tile's max overflows fp32
        q[:, :, :64] *= 12.0
        k[:, :, 700:708] *= 12.0
    elif case == "early_peak":      # the first tile holds a huge score, everything later underflows against it
        k[:, :, 3] = q[:, :, 500] * 6.0
    elif case == "ragged":
        lens = torch.tensor([777, 64, 391], device=dev, dtype=torch.int32)
    elif case == "deep_negative":   # some rows score around -100 nats everywhere: exp2 of the raw score underflows (the no-max variant must notice)
        k[:, :, :, 0] = 0.0
        q[:, :, 200:232, 0] = -1.0e4
        k[:, :, :, 0] = 0.125 * (1.0 + 0.01 * torch.randn(B, H, N, generator=g, device=dev))
    out = torch.empty(B, N, H * 64, device=dev)
    L.check(lib.lemas_k_attention_variant(q.data_ptr(), k.data_ptr(), v.data_ptr(), lens.data_ptr() if lens is not None else None,
                                          out.data_ptr(), B, H, N, variant, None))
    ref = _attn_ref(q, k, v, lens, prescaled=bool(variant & 16))
    assert torch.isfinite(out).all()
    for b in range(B):
        n = int(lens[b]) if lens is not None else N
        # bf16 output: half an ulp is 2^-9 relative (1.6e-2 absolute in [4, 8), where a row that locks onto one key can land)
        err = float(((out[b, :n] - ref[b, :n]).abs() / ref[b, :n].abs().clamp(min=1.0)).max())
        assert err < 2e-2, (variant, case, b, err)
