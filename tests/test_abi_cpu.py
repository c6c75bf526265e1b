"""CPU tier: the C-ABI library builds, loads, exports every symbol include/*.h declares, and refuses
to run without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

from lemas_tts_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    declared = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if h.endswith(".h"):
            declared |= set(re.findall(r"\b(lemas_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", h)).read()))
    assert {"lemas_dit_sample", "lemas_vocos_decode", "lemas_k_gemm_epi"} <= declared, "declarations not parsed"
    product = set(re.findall(r"\b(lemas_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "lemas_hip.h")).read()))
    assert not any(n.startswith("lemas_k_") for n in product), "test entry points belong in lemas_hip_test.h"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/*.h but not exported"
    assert set(_lib.EXPORTED) == declared, (set(_lib.EXPORTED) ^ declared)


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
    L = _lib.lib()
    assert L.lemas_k_timeline(None, 0) != 0
    assert b"not a measurement build" in L.lemas_last_error()
