"""GPU tier: the fp8 (MXFP8) GEMM path of BASELINE config 5 ("fp8 MFMA weights") at kernel level, through the C ABI.

Quantisers are compared bit-for-bit with the torch restatement in oracle/mxfp8.py; the fp8 MFMA GEMM
(v_mfma_scale_f32_32x32x64_f8f6f4, hardware block scales) is compared with fp32 math on the DEQUANTISED operands, so
only accumulation order remains (tolerances stated per test)."""
import math

import numpy as np
import pytest
import torch

from conftest import needs_measurement_build
from oracle.mxfp8 import mx_dequant, mx_quant, w_quant

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lib():
    from lemas_tts_amd import _lib as L
    return L, L.testlib()


def _dev(t):
    return t.to(DEV, torch.float32).contiguous()


def _mixed_scale_rows(M, K, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, K, generator=g)
    x = x * torch.exp2(torch.randint(-12, 9, (M, K // 32, 1), generator=g).float()).expand(M, K // 32, 32).reshape(M, K)
    x[0, :32] = 0.0                 # an all-zero block
    x[1, 32:64] = 448.0             # exactly the e4m3 maximum
    x[2, 64:96] = 1e-30             # below the clamped exponent range
    return x


def test_mx_quant_bit_exact():
    L, lib = _lib()
    M, K = 257, 1024
    x = _mixed_scale_rows(M, K, 5)
    q_ref, mx_ref, _ = mx_quant(x)
    q = torch.empty(M, K, dtype=torch.uint8, device=DEV)
    mx = torch.empty(M, K // 32, dtype=torch.uint8, device=DEV)
    xd = _dev(x)
    L.check(lib.lemas_k_mx_quant(xd.data_ptr(), M, K, q.data_ptr(), mx.data_ptr(), None))
    assert torch.equal(mx.cpu(), mx_ref)
    assert torch.equal(q.cpu(), q_ref)


def test_w_quant_bit_exact():
    L, lib = _lib()
    N, K = 384, 1024
    g = torch.Generator().manual_seed(6)
    w = torch.randn(N, K, generator=g) * 0.02 * (1 + torch.arange(N)[:, None] % 5)
    w[7] = 0.0
    q_ref, sc_ref, _ = w_quant(w)
    q = torch.empty(N, K, dtype=torch.uint8, device=DEV)
    sc = torch.empty(N, device=DEV)
    wd = _dev(w)
    L.check(lib.lemas_k_w_quant_f8(wd.data_ptr(), N, K, q.data_ptr(), sc.data_ptr(), None))
    assert torch.equal(sc.cpu(), sc_ref)
    assert torch.equal(q.cpu(), q_ref)


def _fp8_close(deq, ref, mx):
    """every element within half an e4m3 ulp (2^-4 relative; subnormal spacing 2^-9 of the block scale) of ref,
    plus 1e-5 for fp32 differences in how ref itself was computed"""
    scale = torch.exp2(mx.float() - 127)[..., None].expand(*mx.shape, 32).reshape(ref.shape)
    bound = 0.0626 * ref.abs() + scale * 2.0 ** -10 + 1e-5 * (1 + ref.abs())
    return bool(((deq - ref).abs() <= bound).all())


def test_ln_mod_f8():
    L, lib = _lib()
    M, D = 300, 1024
    g = torch.Generator().manual_seed(7)
    x = torch.randn(M, D, generator=g) * 3 + 0.5
    x[:, 17] *= 40                                            # an outlier channel: block scaling keeps its neighbours precise
    sc, sh = torch.randn(D, generator=g) * 0.3, torch.randn(D, generator=g) * 0.3
    ref = torch.nn.functional.layer_norm(x, (D,), eps=1e-6) * (1 + sc) + sh
    q = torch.empty(M, D, dtype=torch.uint8, device=DEV)
    mx = torch.empty(M, D // 32, dtype=torch.uint8, device=DEV)
    xd, scd, shd = _dev(x), _dev(sc), _dev(sh)        # keep the device tensors alive across the call
    L.check(lib.lemas_k_ln_mod_f8(xd.data_ptr(), scd.data_ptr(), shd.data_ptr(), q.data_ptr(), mx.data_ptr(), M, D, None))
    q, mx = q.cpu(), mx.cpu()
    _, mx_ref, _ = mx_quant(ref)
    assert (mx.int() - mx_ref.int()).abs().max() <= 1 and (mx != mx_ref).float().mean() < 0.01   # amax at a power-of-two edge
    assert _fp8_close(mx_dequant(q, mx), ref, mx)


@pytest.mark.parametrize("M,N,K,act", [(128, 128, 128, 0), (300, 384, 1024, 0), (517, 2048, 1024, 1), (1875, 100, 1024, 0),
                                        (130, 1024, 2048, 0), (1920, 2048, 1024, 2), (3750, 2048, 1024, 2), (200, 128, 1024, 2)])
def test_linear_f8(M, N, K, act):
    L, lib = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-3, 4, (M, 1), generator=g).float())
    W = torch.randn(N, K, generator=g) * 0.05 + (torch.arange(N)[:, None] % 7 - 3) * 0.01     # asymmetric: catches row<->col swaps
    b = torch.randn(N, generator=g)
    _, _, Adq = mx_quant(A)
    _, _, Wdq = w_quant(W)
    ref = (Adq.double() @ Wdq.double().T + b.double()).float()
    out = torch.empty(M, N, device=DEV)
    o8 = torch.zeros(M, max(N, 128), dtype=torch.uint8, device=DEV)
    omx = torch.zeros(M, max(N, 128) // 32, dtype=torch.uint8, device=DEV)
    Ad, Wd, bd = _dev(A), _dev(W), _dev(b)
    L.check(lib.lemas_k_linear_f8(Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), out.data_ptr(), M, N, K, act,
                                  o8.data_ptr(), omx.data_ptr(), None))
    mag = float(ref.abs().max())
    if act == 0:
        err = (out.cpu() - ref).abs().max().item()
        assert err < 2e-6 * mag * math.sqrt(K), (err, mag)          # fp32 accumulation of exact fp8 products
    elif act == 1:
        gref = torch.nn.functional.gelu(ref, approximate="tanh")
        err = (out.cpu() - gref).abs().max().item()
        assert err < 2 ** -8 * mag + 1e-3, (err, mag)               # bf16 output rounding
    else:
        gref = torch.nn.functional.gelu(ref, approximate="tanh")
        mx = omx.cpu()
        _, mx_ref, _ = mx_quant(gref)
        assert (mx.int() - mx_ref.int()).abs().max() <= 1 and (mx != mx_ref).float().mean() < 0.01
        deq = mx_dequant(o8.cpu(), mx)
        bound = 0.0626 * gref.abs() + torch.exp2(mx.float() - 127)[..., None].expand(M, N // 32, 32).reshape(M, N) * 2.0 ** -10 \
            + 2e-3 * (1 + gref.abs())                               # + v_exp/v_rcp GELU vs torch (1e-3 class)
        assert bool(((deq - gref).abs() <= bound).all()), float(((deq - gref).abs() - bound).max())


# ------------------------------------------------------------------------------------------- path level
def _fp8_model(arch, vocab, sd, prosody=False):
    from lemas_tts_amd.model.cfm import CFM
    return CFM(arch, vocab, sd, device=DEV, use_prosody_encoder=prosody, fp8_weights=True)


def _golden_args(fx):
    coef = None if np.isnan(fx["coef"]) else float(fx["coef"])
    B = int(fx["B"])
    kw = {"edit_mask": torch.from_numpy(fx["edit_mask"])} if "edit_mask" in fx else {}
    dur = fx["duration"]
    args = (torch.from_numpy(fx["cond"]), torch.from_numpy(fx["text"]), int(dur[0]) if B == 1 else torch.from_numpy(dur))
    kw.update(lens=torch.from_numpy(fx["lens"]), steps=int(fx["steps"]), cfg_strength=float(fx["cfg"]), sway_sampling_coef=coef,
              y0=torch.from_numpy(fx["y0"]))
    return args, kw


@pytest.mark.parametrize("name", ["mini_plain", "mini_batch", "mini_edit", "full_plain"])
def test_fp8_sampler_vs_emulation_and_reference_golden(golden_dir, name):
    """The HIP fp8 path against (1) the fp32 oracle running the SAME quantisation scheme (OracleDiT(fp8=True),
    oracle/mxfp8.py) and (2) the vectors the real reference produced; north_star's tolerance mel-MSE <= 1e-4 for both.
    (1) is not tighter than (2): the bf16-class differences upstream of each quantiser (attention probabilities, LN)
    move ~2 % of the elements across an e4m3 rounding boundary, i.e. by a whole fp8 ulp, so two correct implementations
    of the scheme decorrelate at the level of the quantisation noise itself; exact-operand correctness of the fp8 kernels
    is what the kernel-level tests above pin.  `full_plain` is a 3-step solve of the 22-block net: its two large Euler
    steps (dt 0.3 / 0.6) carry the ~1 % e4m3 flow error almost undamped (the emulation itself sits at 3.6e-4 from the
    reference), so it gets 1e-3 here and the NFE-32 case below is the one the stated tolerance applies to."""
    from oracle import lemas_oracle as O
    import test_gpu_00_sample as T
    fx, arch, sd = T._load(golden_dir, name)
    m = _fp8_model(arch, int(fx["vocab"]), sd, bool(fx["prosody"]))
    args, kw = _golden_args(fx)
    out, _ = m.sample(*args, use_acc_grl=False, **kw)
    emu, _ = O.OracleCFM(sd, arch, fp8=True).sample(*args, **kw)
    mse_emu = T._gen_mse(out.cpu().numpy(), emu.numpy(), fx)
    mse_ref = T._gen_mse(out.cpu().numpy(), fx["out"], fx)
    print(f"\n[fp8 golden {name}] mel-MSE vs fp8 emulation {mse_emu:.3e}; vs reference {mse_ref:.3e}")
    tol = 1e-3 if name == "full_plain" else 1e-4
    assert mse_emu <= tol, mse_emu
    assert mse_ref <= tol, mse_ref


@pytest.mark.parametrize("name,bounds", [
    # (bf16, MXFP8 weights + activations, fp8 weights only): stated bounds = ~2.5x the measured values in profiles/r03/r03_fp8_points.txt
    ("full_plain", (4e-5, 1e-3, 2.5e-4)),            # measured 1.3e-5 / 3.7e-4 / 8.4e-5 (3-step solve: see above)
    ("full_outlier", (2e-5, 1.6e-3, 6e-4)),          # measured 6.3e-6 / 6.4e-4 / 2.4e-4
])
def test_fp8_cost_of_quantising_activations_and_outlier_stress(golden_dir, name, bounds):
    """BASELINE configs[4] says "fp8 MFMA weights"; the fp8 path of this build also quantises the ACTIVATIONS (MXFP8) because that is
    what the fp8 MFMA wants.  Both points against the REFERENCE's own output, on the plain synthetic weights and on activation-outlier
    stress weights (1 % of the residual channels x30 in all 22 blocks, synth.synth_cfm_state_dict(outlier=...), fixture generated
    through the reference with the same weights): option fp8 = 1 (weights + activations) and fp8 = 2 (weights only, bf16
    activations; an accuracy point computed by the bf16 kernels on the dequantised weights).  With outlier channels NEITHER meets
    the 1e-4 target of north_star -- stated in DESIGN.md section 2; the bf16 path is untouched by them."""
    import test_gpu_00_sample as T
    from lemas_tts_amd.model.cfm import CFM
    fx, arch, sd = T._load(golden_dir, name)
    m = CFM(arch, int(fx["vocab"]), sd, device=DEV)
    m.engine.set_option("fp8_outlier_guard", 0)       # the raw cost of the two quantisations (the guard's effect: test_fp8_outlier_guard below)
    args, kw = _golden_args(fx)
    got = {}
    for mode in (0, 1, 2):
        m.engine.set_option("fp8", mode)
        out, _ = m.sample(*args, use_acc_grl=False, **kw)
        got[mode] = T._gen_mse(out.cpu().numpy(), fx["out"], fx)
    print(f"\n[fp8 points {name}] mel-MSE vs reference: bf16 {got[0]:.3e}  MXFP8 weights+activations {got[1]:.3e}  fp8 weights only {got[2]:.3e}")
    assert got[0] <= bounds[0] and got[1] <= bounds[1] and got[2] <= bounds[2], got
    assert got[0] < got[2] < got[1], got          # each step of quantisation costs accuracy: bf16 < weights-only < weights + activations


@pytest.mark.parametrize("name", ["mini_plain", "mini_batch", "full_plain"])
def test_fp8_attn_f8qk_epilogue_equals_side_launch(golden_dir, name):
    """Engine option attn_f8qk (default 1 on the fp8 path: attention's QK^T on the fp8 MFMA from MXFP8 q / k).  Mode 1 = the fp8 QK GEMM's
    epilogue writes the MXFP8 rows; mode 2 = it writes bf16 and a side launch (launch_qk_mx8, pinned against oracle/mxfp8.py by
    tests/test_gpu_14_attention_f8qk.py) quantises them: the same bytes, so the whole output tensor must be bit-identical.  Mode 0 (bf16
    q / k) must differ and every mode must hold the fp8 path's tolerance against the reference's own output."""
    import test_gpu_00_sample as T
    fx, arch, sd = T._load(golden_dir, name)
    m = _fp8_model(arch, int(fx["vocab"]), sd, bool(fx["prosody"]))
    args, kw = _golden_args(fx)
    outs, mse = {}, {}
    for mode in (1, 2, 0, 1):
        m.engine.set_option("attn_f8qk", mode)
        out, _ = m.sample(*args, use_acc_grl=False, **kw)
        o = out.cpu().numpy()
        if mode in outs:
            assert np.array_equal(outs[mode], o), "replay of mode 1 after the other modes differs"
        outs[mode], mse[mode] = o, T._gen_mse(o, fx["out"], fx)
    print(f"\n[attn_f8qk {name}] mel-MSE vs reference: bf16 q/k {mse[0]:.3e}, MXFP8 q/k {mse[1]:.3e}")
    assert np.array_equal(outs[1], outs[2]), float(np.abs(outs[1] - outs[2]).max())
    assert not np.array_equal(outs[0], outs[1])
    tol = 1e-3 if name == "full_plain" else 1e-4          # (3-step solve: see test_fp8_sampler_vs_emulation_and_reference_golden)
    assert mse[0] <= tol and mse[1] <= tol, mse


def test_attn_f8qk_is_inert_on_the_bf16_path(golden_dir):
    """BASELINE's bf16 configurations must not see it: with the block GEMMs on bf16 operands the option changes nothing (bit-identical),
    unless bit 2 (measurement) forces it"""
    import test_gpu_00_sample as T
    from lemas_tts_amd.model.cfm import CFM
    fx, arch, sd = T._load(golden_dir, "mini_plain")
    m = CFM(arch, int(fx["vocab"]), sd, device=DEV)
    args, kw = _golden_args(fx)
    outs = {}
    for mode in (0, 1, 2, 6):
        m.engine.set_option("attn_f8qk", mode)
        outs[mode] = m.sample(*args, use_acc_grl=False, **kw)[0].cpu().numpy()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert not np.array_equal(outs[0], outs[6])
    assert T._gen_mse(outs[6], fx["out"], fx) <= 1e-4


@needs_measurement_build
def test_outlier_rows_kernel():
    """csrc/outlier_rows.hip: the flagged output channels of a residual-writing projection from bf16 operands, gated and added in place --
    against fp64 on the bf16-rounded operands; ragged lengths (rows at or past a sample's length and padding rows must not be touched)."""
    L, lib = _lib()
    g = torch.Generator(device=DEV).manual_seed(77)
    for (batch, frames, K, nf, ragged) in ((1, 1875, 1024, 10, False), (3, 700, 2048, 32, True), (2, 130, 1024, 1, True), (1, 2814, 2048, 10, False)):
        pitch = (frames + 127) // 128 * 128
        M, d = batch * pitch, 1024
        A = torch.randn(M, K, generator=g, device=DEV)
        W = torch.randn(nf, K, generator=g, device=DEV) * 0.6
        bias = torch.randn(nf, generator=g, device=DEV)
        gate = torch.randn(d, generator=g, device=DEV)
        chan = torch.sort(torch.randperm(d, generator=g, device=DEV)[:nf]).values.to(torch.int32)
        x0 = torch.randn(M, d, generator=g, device=DEV)
        lens = None
        if ragged:
            lens = torch.randint(1, frames + 1, (batch,), generator=g, device=DEV, dtype=torch.int32)
            lens[-1] = frames
        x = x0.clone()
        a8, amx = torch.zeros(M, K, dtype=torch.uint8, device=DEV), torch.zeros(M, K // 32, dtype=torch.uint8, device=DEV)
        L.check(lib.lemas_k_outlier_rows(A.data_ptr(), W.data_ptr(), bias.data_ptr(), chan.data_ptr(), nf, gate.data_ptr(),
                                         lens.data_ptr() if lens is not None else None, x.data_ptr(), batch, frames, pitch, K, d,
                                         a8.data_ptr(), amx.data_ptr(), None), "outlier_rows")
        # the MXFP8 image it writes on the way: bit-exact against the oracle's quantiser applied to the bf16-rounded rows
        from oracle import mxfp8 as MX
        q_ref, mx_ref, _ = MX.mx_quant(A.to(torch.bfloat16).float().cpu())
        assert torch.equal(a8.cpu(), q_ref) and torch.equal(amx.cpu(), mx_ref), "MXFP8 image differs from oracle/mxfp8.py"
        y = (A.to(torch.bfloat16).double() @ W.to(torch.bfloat16).double().T + bias.double()) * gate[chan.long()].double()
        ref = x0.double().clone()
        pos = torch.arange(M, device=DEV) % pitch
        lim = torch.full((M,), frames, device=DEV) if lens is None else torch.minimum(lens.long().repeat_interleave(pitch), torch.tensor(frames, device=DEV))
        live = pos < lim
        ref[live[:, None] & torch.zeros(M, d, dtype=torch.bool, device=DEV).index_fill_(1, chan.long(), True)] += y[live].reshape(-1)
        err = float((x.double() - ref).abs().max())
        print(f"\n[outlier_rows B={batch} frames={frames} K={K} nf={nf}] max|err| {err:.3e}")
        assert err <= 2e-5 * math.sqrt(K) * 3.0, err
        untouched = ~live[:, None] | ~torch.zeros(M, d, dtype=torch.bool, device=DEV).index_fill_(1, chan.long(), True)
        assert torch.equal(x[untouched], x0[untouched]), "a row or column outside the flagged live set was written"


def _outlier_case(golden_dir, name):
    import test_gpu_00_sample as T
    from lemas_tts_amd import synth
    from lemas_tts_amd.model.cfm import CFM
    fx, arch, sd = T._load(golden_dir, name)
    fx = synth.expand_reference_fixture(fx)
    m = CFM(arch, int(fx["vocab"]), sd, device=DEV)
    args, kw = _golden_args(fx)

    def run(fp8, guard=1, mode=1):
        m.engine.set_option("fp8_outlier_guard", guard)
        m.engine.set_option("fp8_outlier_mode", mode)
        m.engine.set_option("fp8", fp8)
        out, _ = m.sample(*args, use_acc_grl=False, **kw)
        return out.cpu().numpy(), T._gen_mse(out.cpu().numpy(), fx["out"], fx)
    return m, fx, run


def test_fp8_outlier_guard(golden_dir):
    """Activation outliers at a PRODUCTION step count (tests/golden/configs0_outlier_nfe32.npz: the reference's own output at full depth,
    NFE 32, on weights whose residual-writing projections scale 1 % of the channels x30 -- oracle/gen_golden.py --full-size).  Unguarded,
    the fp8 path misses the 1e-4 target there (2.9e-4).  The guard (default) sees the outlier channels in the per-channel weight scales and
    runs every block GEMM on its bf16 operands: the bf16 path's bits.  (The mixed-precision decomposition of round 5, engine option
    fp8_outlier_mode = 1, lives in measurement builds only since round 6 -- it met 1e-4 but ran slower than this fallback:
    test_fp8_outlier_decomposition below.)  On weights without such channels nothing changes."""
    m, fx, run = _outlier_case(golden_dir, "configs0_outlier_nfe32")
    out_u, unguarded = run(1, guard=0, mode=0)
    assert m.engine.stat("fp8_gemms_kept_bf16") == 0
    n_out = m.engine.stat("fp8_outlier_channels")
    out_g, legacy = run(1, guard=1, mode=0)
    assert m.engine.stat("fp8_gemms_kept_bf16") == 4
    out_b, bf16 = run(0, mode=0)
    print(f"\n[fp8 on outlier weights, NFE 32, {n_out} outlier channels] mel-MSE vs reference: bf16 {bf16:.3e}  fp8 unguarded {unguarded:.3e}  "
          f"guard with every GEMM on bf16 {legacy:.3e}")
    assert 5 <= n_out <= 20                   # 1 % of 1024 channels
    assert bf16 <= 1e-4 and legacy <= 1e-4, (bf16, legacy)      # THE tolerance (BASELINE.json), not a multiple of what was measured
    assert unguarded > 1e-4                   # what the guard is there for (if this ever passes unguarded, the guard can go)
    np.testing.assert_array_equal(out_g, out_b)
    out_u2, _ = run(1, guard=0, mode=0)
    np.testing.assert_array_equal(out_u, out_u2)
    if not __import__("conftest").measurement_build():          # the product refuses the measurement option by name
        from lemas_tts_amd import _lib as L
        with pytest.raises(L.LemasError, match="measurement option"):
            m.engine.set_option("fp8_outlier_mode", 1)
    del m
    # no outlier channels: the guard does not trip and the fp8 path is bit-for-bit what it was
    import test_gpu_00_sample as T
    fx2, arch2, sd2 = T._load(golden_dir, "mini_plain")
    m2 = _fp8_model(arch2, int(fx2["vocab"]), sd2)
    a2, k2 = _golden_args(fx2)
    outs = []
    for guard in (0, 1):
        m2.engine.set_option("fp8_outlier_guard", guard)
        o, _ = m2.sample(*a2, use_acc_grl=False, **k2)
        outs.append(o.cpu().numpy())
        assert m2.engine.stat("fp8_gemms_kept_bf16") == 0 and m2.engine.stat("fp8_outlier_channels") == 0
    np.testing.assert_array_equal(outs[0], outs[1])
    m2.engine.set_option("fp8_outlier_guard", 1)


@needs_measurement_build
def test_fp8_outlier_decomposition(golden_dir):
    """MEASUREMENT BUILDS ONLY: engine option fp8_outlier_mode = 1 -- QKV, out-projection and FF2 stay on fp8 operands (three of the four
    GEMM sites), FF1 runs on bf16 operands and the flagged OUTPUT channels of out-projection / FF2 are computed from bf16 operands by
    csrc/outlier_rows.hip.  Measured 8.4e-5 / 9.6e-5 against the 1e-4 target (a 4 % margin on the eight-step fixture: a different reduction
    order or lease can flip it, which is one more reason it is not product code); -8 % throughput against the bf16 fallback."""
    m, fx, run = _outlier_case(golden_dir, "configs0_outlier_nfe32")
    _, unguarded = run(1, guard=0, mode=0)
    out_s, split = run(1, guard=1, mode=1)
    assert m.engine.stat("fp8_gemms_kept_bf16") == 1          # FF1 only: fp8 on 3 of the 4 sites
    _, bf16 = run(0, mode=0)
    assert split <= 1e-4 and bf16 < split < unguarded, (bf16, split, unguarded)
    m.engine.set_option("graph", 0)
    out_s2, _ = run(1, guard=1, mode=1)
    m.engine.set_option("graph", 1)
    np.testing.assert_array_equal(out_s, out_s2)


@needs_measurement_build
def test_fp8_outlier_decomposition_on_the_eight_step_fixture(golden_dir):
    """tests/golden/full_outlier.npz: the same stress on an EIGHT-step solve (F = 150, N = 400), whose coarse steps integrate any flow error
    almost undamped: fp8 unguarded 6.9e-4.  The decomposition meets the 1e-4 target here too -- at 9.6e-5, a 4 % margin (see above)."""
    m, fx, run = _outlier_case(golden_dir, "full_outlier")
    _, unguarded = run(1, guard=0)
    _, split = run(1, guard=1, mode=1)
    assert m.engine.stat("fp8_gemms_kept_bf16") == 1
    _, bf16 = run(0)
    print(f"\n[fp8 on outlier weights, 8-step solve] mel-MSE vs reference: bf16 {bf16:.3e}  fp8 unguarded {unguarded:.3e}  fp8 decomposition {split:.3e}")
    assert bf16 <= 1e-4 and split <= 1e-4 and unguarded > 1e-4, (bf16, split, unguarded)


# The full-depth, NFE-32 tolerance of the fp8 path is checked at FULL SIZE against the reference's own output in
# tests/test_gpu_06_configs.py::test_configs1_full_size_full_nfe_vs_the_reference (mel-MSE 4.2e-5 against the 1e-4 target); the
# round-1 form of that check (N = 300 against a 2-minute oracle run on the GPU box's host cores) was dropped for it.


def test_fp8_graph_eager_dual_bit_identical(golden_dir):
    import test_gpu_00_sample as T
    fx, arch, sd = T._load(golden_dir, "mini_plain")
    m = _fp8_model(arch, int(fx["vocab"]), sd)
    outs = []
    for graph, dual in ((1, 1), (0, 1), (1, 0), (0, 0)):
        m.engine.set_option("graph", graph)
        m.engine.set_option("dual", dual)
        o, _ = m.sample(torch.from_numpy(fx["cond"]), torch.from_numpy(fx["text"]), int(fx["duration"][0]), lens=torch.from_numpy(fx["lens"]),
                        steps=int(fx["steps"]), cfg_strength=float(fx["cfg"]), sway_sampling_coef=float(fx["coef"]),
                        y0=torch.from_numpy(fx["y0"]), use_acc_grl=False)
        outs.append(o.cpu().numpy())
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0], o)


def test_fp8_switch_is_reversible_and_changes_the_numbers(golden_dir):
    """option "fp8" really switches the GEMM path (outputs differ from bf16) and switching back restores the bf16 bits"""
    import test_gpu_00_sample as T
    from lemas_tts_amd.model.cfm import CFM
    fx, arch, sd = T._load(golden_dir, "mini_plain")
    m = CFM(arch, int(fx["vocab"]), sd, device=DEV)
    run = lambda: m.sample(torch.from_numpy(fx["cond"]), torch.from_numpy(fx["text"]), int(fx["duration"][0]), lens=torch.from_numpy(fx["lens"]),
                           steps=int(fx["steps"]), cfg_strength=float(fx["cfg"]), sway_sampling_coef=float(fx["coef"]),
                           y0=torch.from_numpy(fx["y0"]), use_acc_grl=False)[0].cpu().numpy()
    a = run()
    m.engine.set_option("fp8", 1)
    b = run()
    m.engine.set_option("fp8", 0)
    c = run()
    assert not np.array_equal(a, b)
    np.testing.assert_array_equal(a, c)


def test_config5_fp8_speech_edit_30s_three_spans():
    """configs[4] as BASELINE states it: 30 s source, 3 edit spans, NFE-48 grid (its last two steps), fp8 MFMA weights."""
    from oracle import lemas_oracle as O
    from lemas_tts_amd import synth
    from lemas_tts_amd.model.layout import DiTArch
    import test_gpu_06_configs as Cf
    arch = DiTArch(depth=2)
    sd = synth.synth_cfm_state_dict(arch, Cf.VOCAB, 71)
    nw = 720000
    F_ = nw // 256 + 1
    edit = O.build_edit_mask(nw, [(4.0, 6.5), (12.0, 15.0), (22.0, 24.0)])
    cond, text, _ = Cf._inputs(72, 1, [F_], [F_ + 1], [400])
    y0 = torch.from_numpy(synth.synth_noise(73, F_ + 1))[None]
    tg = O.time_grid(48, 3.0)[-3:]
    m = _fp8_model(arch, Cf.VOCAB, sd)
    cmask = torch.nn.functional.pad(edit, (0, 1), value=False)
    cpad = torch.nn.functional.pad(cond, (0, 0, 0, 1))
    out, _, _ = m.engine.sample(cpad, cmask, text, tg.numpy(), y0, cond_frames=F_, cfg_strength=2.0)
    ref, _ = O.OracleCFM(sd, arch).sample(cond, text, nw // 256, y0=y0, steps=2, cfg_strength=2.0, edit_mask=edit, t_grid=tg)
    keep = ~cmask[0]
    mse = float(((out.cpu()[0, keep] - ref[0, keep]).double() ** 2).mean())
    print(f"\n[config5 fp8 edit N={F_ + 1}] mel-MSE over regenerated frames {mse:.3e}")
    assert mse <= 1e-4


def test_fp8_flow_error_full_size():
    """configs[1] shape, 22 blocks: one flow evaluation at t = 0 with fp8 GEMMs vs the fp32 oracle.  Relative error of
    the flow is the honest per-step figure (bf16 path: 0.3 %); 3 % is the e4m3 budget (2^-4 half-ulp, sqrt-averaged)."""
    from oracle import lemas_oracle as O
    from lemas_tts_amd import synth
    from lemas_tts_amd.model.layout import DiTArch
    import test_gpu_06_configs as Cf
    arch = DiTArch()
    sd = synth.synth_cfm_state_dict(arch, Cf.VOCAB, 1234)
    F_, N = 938, 1875
    cond, text, y0 = Cf._inputs(1234, 1, [F_], [N], [round(N * 0.17)])
    m = _fp8_model(arch, Cf.VOCAB, sd)
    tg = O.time_grid(32, 5)
    cm = torch.zeros(1, N, dtype=torch.bool)
    cm[:, :F_] = True
    cpad = torch.nn.functional.pad(cond, (0, 0, 0, N - F_))
    m.engine.prepare(cpad, cm, text, tg.numpy(), cond_frames=F_, cfg_strength=2.0)
    pred = m.engine.forward(y0, 31).cpu()
    odit = O.OracleDiT(sd, arch)
    step_cond = torch.where(cm[..., None], cpad, torch.zeros_like(cpad))
    pc = odit.forward(y0, step_cond, text, tg[31], False, False)
    rel = float((pred[0] - pc[0]).norm() / pc[0].norm())
    print(f"\n[fp8 flow N={N}] relative error of the conditional flow at step 31: {rel:.4f}")
    assert rel < 0.03


@pytest.mark.parametrize("seed", list(range(10)))
def test_fp8_randomised_small_configurations(seed):
    """the fp8 variant on random batch sizes / lengths / masks / step counts (2 blocks): finite, within the reference tolerance
    of the fp32 oracle, kept frames exact, and run-to-run deterministic"""
    from lemas_tts_amd import synth
    from lemas_tts_amd.model.layout import DiTArch
    from oracle import lemas_oracle as O
    rng = np.random.default_rng(9000 + seed)
    arch = DiTArch(depth=2)
    sd = synth.synth_cfm_state_dict(arch, 898, 171)
    m = _fp8_model(arch, 898, sd)
    B = int(rng.integers(1, 4))
    Fm = int(rng.integers(3, 120))
    lens = [Fm] + [int(rng.integers(1, Fm + 1)) for _ in range(B - 1)]
    durs = [int(rng.integers(l + 1, l + 200)) for l in lens]
    nts = [int(rng.integers(1, max(2, d // 3))) for d in durs]
    cond = torch.stack([torch.from_numpy(synth.synth_cond_mel(seed * 5 + b, Fm)) for b in range(B)])
    text = torch.full((B, max(nts)), -1, dtype=torch.long)
    for b in range(B):
        text[b, :nts[b]] = torch.from_numpy(synth.synth_tokens(seed * 5 + b, nts[b], 898))
    eff = [max(max(nts[b], lens[b]) + 1, durs[b]) for b in range(B)]
    y0 = torch.zeros(B, max(eff), 100)
    for b in range(B):
        y0[b, :eff[b]] = torch.from_numpy(synth.synth_noise(seed * 5 + b, eff[b]))
    kw = dict(steps=int(rng.integers(2, 5)), cfg_strength=2.0, sway_sampling_coef=5, lens=torch.tensor(lens))
    dur_arg = durs[0] if B == 1 else torch.tensor(durs)
    out, _ = m.sample(cond, text, dur_arg, y0=y0, use_acc_grl=False, **kw)
    out2, _ = m.sample(cond, text, dur_arg, y0=y0, use_acc_grl=False, **kw)
    np.testing.assert_array_equal(out.cpu().numpy(), out2.cpu().numpy())
    ref, _ = O.OracleCFM(sd, arch).sample(cond, text, dur_arg, y0=y0, **kw)
    assert bool(torch.isfinite(out).all())
    se = cnt = 0.0
    for b in range(B):
        d = (out.cpu()[b, lens[b]:eff[b]] - ref[b, lens[b]:eff[b]]).double()
        se += float((d ** 2).sum()); cnt += d.numel()
        np.testing.assert_array_equal(out.cpu().numpy()[b, :lens[b]], cond.numpy()[b, :lens[b]])
    assert se / max(cnt, 1) <= 1e-4, (seed, B, Fm, lens, durs, se / max(cnt, 1))
