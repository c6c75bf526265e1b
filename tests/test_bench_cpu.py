"""CPU tier: the host-side pieces of bench.py that need no GPU -- the rocm-smi reader behind the ``clock_power`` field and the FLOP
bookkeeping the roofline fields are priced with."""
import os
import stat
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAKE_SMI = """#!/bin/sh
cat <<'TXT'
============================ ROCm System Management Interface ============================
GPU[0]\t\t: fclk clock level: 0: (1250Mhz)
GPU[0]\t\t: mclk clock level: 0: (2000Mhz)
GPU[0]\t\t: sclk clock level: 1: (2184Mhz)
=================================== Power Consumption ====================================
GPU[0]\t\t: Current Socket Graphics Package Power (W): 1231.0
TXT
"""


def test_clock_power_reads_rocm_smi(tmp_path, monkeypatch):
    """``clock_power`` runs the timed work again (untimed) while a thread reads rocm-smi: with a stand-in rocm-smi on PATH and a
    stand-in step it must report the averages of the busy samples and say where they came from."""
    import bench
    exe = tmp_path / "rocm-smi"
    exe.write_text(FAKE_SMI)
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ.get("PATH", ""))
    monkeypatch.setattr(bench.torch.cuda, "synchronize", lambda *a, **k: None)
    calls = []
    cp = bench.clock_power(lambda: calls.append(1) or __import__("time").sleep(0.05), seconds=2.5)
    assert cp is not None and calls
    assert cp["sclk_mhz"] == 2184 and cp["package_w"] == 1231.0 and cp["samples"] >= 1
    assert cp["sclk_ceiling_mhz"] == 2400 and "rocm-smi" in cp["source"]


def test_clock_power_without_rocm_smi_is_none(tmp_path, monkeypatch):
    import bench
    monkeypatch.setenv("PATH", str(tmp_path))
    monkeypatch.setattr(bench.os.path, "exists", lambda p: False)
    assert bench.clock_power(lambda: None, seconds=0.1) is None


def test_forward_flops_formula():
    """F_fwd(B, N) = B (378 888 192 N + 90 112 N^2); the block GEMM classes add up to the linear part's share"""
    import bench
    assert bench.fwd_flops(1, 1875) == pytest.approx(378_888_192 * 1875 + 90_112 * 1875 ** 2)
    assert bench.fwd_flops(8, 1125) == pytest.approx(8 * (378_888_192 * 1125 + 90_112 * 1125 ** 2))


def test_frontend_bench_counts_the_encoder_flops_of_the_weights_it_loads():
    """tools/frontend_bench.py prices the ECAPA encode at 2 flops per multiply-add of every convolution / linear: the count from the layer
    list must equal the one from the state-dict shapes (every weight element is used once per frame, or once per prompt for the one-row
    Linears: SE gates, the global-context columns of the attention TDNN, the final fc), for the published widths and for a second architecture."""
    import importlib.util
    import numpy as np
    from lemas_tts_amd.model.layout import ProsodyArch, prosody_param_shapes
    spec = importlib.util.spec_from_file_location("frontend_bench", os.path.join(ROOT, "tools", "frontend_bench.py"))
    fb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fb)
    other = ProsodyArch(channels=(96, 128, 128, 256), kernel_sizes=(3, 5, 3, 1), dilations=(2, 1, 3, 1), attention_channels=32,
                        res2net_scale=4, se_channels=16, global_context=False, groups=(1, 1, 1, 1), embed_dim=64, input_dim=40)
    for arch, T in ((ProsodyArch(), 998), (other, 77)):
        want = 0
        for name, shape in prosody_param_shapes(arch).items():
            if not name.endswith("weight") or len(shape) != 3:
                continue
            n = int(np.prod(shape))
            if "se_block" in name or name == "fc.weight":
                want += 2 * n
            elif name == "asp.tdnn.conv.weight" and arch.global_context:
                want += 2 * T * (n // 3) + 2 * (2 * n // 3)
            else:
                want += 2 * T * n
        assert fb.ecapa_flops(arch, T) == want, (fb.ecapa_flops(arch, T), want)
    assert fb.ecapa_flops(ProsodyArch(), 998) == 9562013696     # the figure in DESIGN.md section 3 / profiles/r06/r06fk_frontend_bench.json
