#!/usr/bin/env python
"""bf16 vs fp8 (MX) GEMM timings through lemas_k_bench (development aid).   KB_M=1920 python tools/kbench_f8.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, ".")
import torch  # noqa
from lemas_tts_amd import _lib

L = _lib.lib()
shapes = [("gemm_qk", 2048, 1024), ("gemm_v", 1024, 1024), ("gemm_gate", 1024, 1024), ("gemm_gelu", 2048, 1024), ("gemm_gelu8", 2048, 1024),
          ("gemm_gate", 1024, 2048), ("gemm_none", 2048, 1024), ("gemm_none", 1024, 2048)]
for M in [int(v) for v in os.environ.get("KB_M", "3840,1920").split(",")]:
    print(f"M={M}; cells = us (TFLOP/s); columns: bf16 auto | fp8 auto | fp8 v4 | fp8 v6 | fp8 v10 | fp8 v11")
    for what, N, K in shapes:
        row = []
        for pre, v in (("", 0), ("f8_", 0), ("f8_", 4), ("f8_", 6), ("f8_", 10), ("f8_", 11)):
            if pre == "" and what == "gemm_gelu8":
                row.append("      -      ")
                continue
            us = C.c_double()
            rc = L.lemas_k_bench((pre + what).encode(), M, N, K, 50, v, C.byref(us))
            row.append(f"{us.value:6.1f} ({2.0 * M * N * K / (us.value * 1e-6) / 1e12:4.0f})" if rc == 0 else f"err{rc}")
        print(f"{what:10s} N={N:4d} K={K:4d} | " + " | ".join(row))
