"""GPU tier, kernel level: each hand-written HIP kernel through its C-ABI entry point (lemas_k_*) against a
plain fp32 torch reference of the same op.  Tolerances are stated per test; bf16-operand kernels are compared
against fp32 math on bf16-ROUNDED inputs so that only accumulation order / output rounding remain."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from lemas_tts_amd import _lib as L
    return L, L.lib()


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


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("B,H,N,lens", [(1, 2, 64, None), (2, 16, 200, None), (2, 4, 333, [333, 210]), (1, 16, 1875, None),
                                         (3, 2, 130, [1, 64, 130]), (1, 1, 65, None), (2, 2, 192, [129, 192])])
def test_attention(B, H, N, lens, variant):
    L, lib = _lib()
    lib.lemas_k_set_attention_variant(variant)      # 1 = four-wave kernel, 2 = split-KV eight-wave kernel
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
    lib.lemas_k_set_attention_variant(0)
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
    sequence lengths, head counts and ragged key lengths for both attention kernels"""
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
    for variant in (1, 2):
        lib.lemas_k_set_attention_variant(variant)
        qd, kd, vd = _dev(q), _dev(k), _dev(v)
        ld = None if lens is None else torch.tensor(lens, dtype=torch.int32, device="cuda:0")
        o = torch.empty(B, Ns, H * 64, device="cuda:0")
        L.check(lib.lemas_k_attention(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), ld.data_ptr() if ld is not None else None, o.data_ptr(), B, H, Ns, None))
        d = (o.cpu() - refa).abs()
        if lens is not None:                        # rows past a sample's length are unspecified
            for bi in range(B):
                d[bi, lens[bi]:] = 0
        assert float(d.max()) < 3e-2, (variant, B, H, Ns, lens, float(d.max()))
    lib.lemas_k_set_attention_variant(0)
