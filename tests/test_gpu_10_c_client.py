"""The drop-in boundary is a C ABI: a gcc-compiled, torch-free client (examples/c_abi_smoke.c) drives the vocoder entry points (the
operators behind vocos.decode, utils_infer.py:601-608) AND the sampler (lemas_dit_create / load_weight under the checkpoint's keys /
finalize / sample on a depth-1 model: CFM.sample, cfm.py:206-473) through liblemas_hip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_plain_c_client_runs():
    exe = os.path.join(ROOT, "examples", "c_abi_smoke")
    if not os.path.exists(exe):
        from lemas_tts_amd.build import build_c_client
        exe = build_c_client()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "c_abi_smoke:" in r.stdout and "dit depth 1" in r.stdout and "repeatable 1, conditioning frames copied 1" in r.stdout
