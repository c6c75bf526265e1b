"""CPU tier: the C-ABI library builds, loads, exports every symbol include/*.h declares, and refuses
to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest
import torch

from lemas_tts_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    return set(re.findall(r"\b(lemas_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", header)).read()))


def test_header_symbols_are_exported():
    """include/lemas_hip.h is the product library's surface, include/lemas_hip_test.h the test library's; neither library exports
    the other's entry points (no lemas_k_* -- and no process-global dispatch switch -- in the product)."""
    product, tests = _declared("lemas_hip.h"), _declared("lemas_hip_test.h")
    assert {"lemas_dit_sample", "lemas_vocos_decode", "lemas_dit_health"} <= product and "lemas_k_gemm_epi" in tests, "declarations not parsed"
    assert not any(n.startswith("lemas_k_") for n in product), "test entry points belong in lemas_hip_test.h"
    assert all(n.startswith("lemas_k_") for n in tests)
    P = C.CDLL(_lib.LIB_PATH)
    _lib.lib()
    T = C.CDLL(_lib.TEST_LIB_PATH)
    for name in sorted(product):
        assert hasattr(P, name), f"{name} declared in lemas_hip.h but not exported by liblemas_hip.so"
    for name in sorted(tests):
        assert hasattr(T, name), f"{name} declared in lemas_hip_test.h but not exported by liblemas_hip_test.so"
    exported = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "lemas_k_" not in exported and "force_tiles" not in exported, "the product library must not carry test hooks"
    undefined = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    own = [l for l in undefined.splitlines() if "launch_" in l or "lemas" in l]
    assert not own, f"the product library references its own symbols without defining them (a source compiled out by a build flag?): {own}"
    assert set(_lib.EXPORTED) == product, (set(_lib.EXPORTED) ^ product)
    assert set(_lib.EXPORTED_TEST) == tests, (set(_lib.EXPORTED_TEST) ^ tests)


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a GPU")
def test_no_cpu_fallback():
    L = _lib.lib()
    cfg = _lib.DitConfig(1024, 2, 16, 64, 2, 512, 4, 100, 899, 31, 16, 256, 0)
    h = C.c_void_p()
    rc = L.lemas_dit_create(C.byref(cfg), C.byref(h))
    assert rc != 0 and b"no HIP device" in L.lemas_last_error()
    from lemas_tts_amd.engine import DiTEngine
    from lemas_tts_amd.model.layout import DiTArch
    with pytest.raises(_lib.LemasError):
        DiTEngine(DiTArch(depth=1), 10, {}, device="cpu")


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under lemas_tts_amd/ may import it."""
    pkg = os.path.join(ROOT, "lemas_tts_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"


def test_measurement_hooks_are_compiled_out_of_the_product_build():
    """the in-situ timeline (tools/timeline_step.sh) exists only in a -DLEMAS_PHASE_TIMESTAMPS build: the in-tree library refuses it"""
    L = _lib.testlib()
    assert L.lemas_k_timeline(None, 0) != 0
    assert b"not a measurement build" in L.lemas_last_error()
