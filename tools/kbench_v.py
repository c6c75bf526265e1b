#!/usr/bin/env python
"""GEMM variant sweep incl. fp8: python tools/kbench_v.py M v1 v2 ...   (development aid)"""
import ctypes as C
import sys

sys.path.insert(0, ".")
import torch  # noqa
from lemas_tts_amd import _lib

L = _lib.lib()
M = int(sys.argv[1])
variants = [int(v) for v in sys.argv[2:]]
shapes = [("gemm_none", 2048, 1024), ("gemm_qk", 2048, 1024), ("gemm_gelu", 2048, 1024), ("gemm_v", 1024, 1024), ("gemm_gate", 1024, 1024),
          ("gemm_gate", 1024, 2048), ("f8_gemm_none", 2048, 1024), ("f8_gemm_qk", 2048, 1024), ("f8_gemm_gelu8", 2048, 1024),
          ("f8_gemm_v", 1024, 1024), ("f8_gemm_gate", 1024, 1024), ("f8_gemm_gate", 1024, 2048)]
print(f"M={M}; columns = variants {variants}; cells = us (TFLOP/s)")
for what, N, K in shapes:
    row = []
    for v in variants:
        us = C.c_double()
        rc = L.lemas_k_bench(what.encode(), M, N, K, 50, v, C.byref(us))
        row.append(f"{us.value:6.1f} ({2.0 * M * N * K / (us.value * 1e-6) / 1e12:4.0f})" if rc == 0 else f"   err{rc:4d}   ")
    print(f"{what:13s} N={N:4d} K={K:4d} | " + " | ".join(row))
