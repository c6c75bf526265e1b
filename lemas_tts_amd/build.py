"""Build liblemas_hip.so (the product: engines + kernels behind include/lemas_hip.h) and liblemas_hip_test.so (the lemas_k_* test and
measurement entry points of include/lemas_hip_test.h: csrc/engine_ktests.hip linked AGAINST the product library, so the process holds
one copy of every kernel) in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m lemas_tts_amd.build            # incremental
    python -m lemas_tts_amd.build --force

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "liblemas_hip.so")
TEST_LIB = os.path.join(LIBDIR, "liblemas_hip_test.so")
TEST_ONLY = {"engine_ktests.hip", "attention_q64.hip"}     # sources that go into the test library, not the product (test entry points,
                                                           # measurement-only kernels)
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-I", INCLUDE, "-I", CSRC]
FLAGS += os.environ.get("LEMAS_EXTRA_HIPCC_FLAGS", "").split()     # e.g. -DLEMAS_PHASE_TIMESTAMPS for a measurement build


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, src[:-4] + ".o")
    spath = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(spath), _headers_mtime()):
        return obj
    cmd = [HIPCC, *FLAGS, "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def _link(out: str, objs, extra=()) -> None:
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, *extra, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")


def build_library(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = dict(zip(srcs, ex.map(lambda s: _compile(s, force), srcs)))
    prod = [o for s, o in objs.items() if s not in TEST_ONLY]
    test = [o for s, o in objs.items() if s in TEST_ONLY]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in prod):
        _link(LIB, prod)
    if force or not os.path.exists(TEST_LIB) or os.path.getmtime(LIB) > os.path.getmtime(TEST_LIB) or \
            any(os.path.getmtime(o) > os.path.getmtime(TEST_LIB) for o in test):
        _link(TEST_LIB, test, ["-L", LIBDIR, "-llemas_hip", "-Wl,-rpath,$ORIGIN"])
    return LIB


def build_c_client() -> str:
    """gcc-compile the plain-C client of the ABI (examples/c_abi_smoke.c): a compile-time proof that
    include/lemas_hip.h is C, and the binary a GPU test runs."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "examples", "c_abi_smoke.c")
    out = os.path.join(root, "examples", "c_abi_smoke")
    if os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(src), os.path.getmtime(LIB), _headers_mtime()):
        return out
    cmd = ["gcc", "-O2", "-std=c11", "-Wall", "-D_GNU_SOURCE", "-I", INCLUDE, src, "-o", out, "-L", LIBDIR, "-llemas_hip",
           "-Wl,-rpath,$ORIGIN/../lemas_tts_amd/lib", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"gcc failed for c_abi_smoke.c:\n{r.stdout}\n{r.stderr}")
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
    print(build_c_client())
