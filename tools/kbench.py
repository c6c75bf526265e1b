#!/usr/bin/env python
"""Micro-benchmark sweep of step-loop kernels through lemas_k_bench (development aid).
    python tools/kbench.py [variants...]"""
import ctypes as C
import sys

sys.path.insert(0, ".")
import torch  # noqa
from lemas_tts_amd import _lib

L = _lib.lib()
import os
M = int(os.environ.get("KB_M", "3840"))
variants = [int(v) for v in sys.argv[1:]] or [2, 3, 4, 5, 6, 7, 10, 11, 12]
shapes = [("gemm_qk", 2048, 1024), ("gemm_v", 1024, 1024), ("gemm_gate", 1024, 1024), ("gemm_gelu", 2048, 1024), ("gemm_gate", 1024, 2048),
          ("gemm_gelu", 2048, 2048)]
print(f"M={M}; columns = variants {variants}; cells = us (TFLOP/s)")
for what, N, K in shapes:
    row = []
    for v in variants:
        us = C.c_double()
        rc = L.lemas_k_bench(what.encode(), M, N, K, 50, v, C.byref(us))
        if rc != 0:
            row.append(f"err{rc}")
            continue
        tf = 2.0 * M * N * K / (us.value * 1e-6) / 1e12
        row.append(f"{us.value:6.1f} ({tf:4.0f})")
    print(f"{what:10s} N={N:4d} K={K:4d} | " + " | ".join(row))
us = C.c_double()
for n, bh in ((1875, 32), (1125, 256)):
    rc = L.lemas_k_bench(b"attention", n, bh, 0, 50, 0, C.byref(us))
    tf = 4.0 * n * n * 64 * bh / (us.value * 1e-6) / 1e12
    print(f"attention N={n} BH={bh}: {us.value:.1f} us ({tf:.0f} TF)" if rc == 0 else f"attention err {rc}")
