import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def measurement_build() -> bool:
    """True when the in-tree libraries were built with -DLEMAS_MEASUREMENT_BUILD: the kept-reproducible experiments (engine options ln_fused /
    lane_skew / xcd_runs, the LayerNorm tail of the gate + residual GEMM) exist only there; the product build refuses them."""
    from lemas_tts_amd import _lib
    return bool(_lib.testlib().lemas_k_build_flags() & 1)


needs_measurement_build = pytest.mark.skipif("not __import__('conftest').measurement_build()",
                                             reason="measurement-only code: build with LEMAS_EXTRA_HIPCC_FLAGS=-DLEMAS_MEASUREMENT_BUILD")
