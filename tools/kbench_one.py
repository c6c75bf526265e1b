"""Run ONE kernel micro-benchmark (for rocprofv3 --pmc passes): python tools/kbench_one.py <what> <M> <N> <K> [variant]"""
import ctypes as C, sys
sys.path.insert(0, ".")
import torch
from lemas_tts_amd import _lib
L = _lib.lib()
what, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
v = int(sys.argv[5]) if len(sys.argv) > 5 else 0
us = C.c_double()
rc = L.lemas_k_bench(what.encode(), M, N, K, 20, v, C.byref(us))
print(f"{what} M={M} N={N} K={K} v={v}: rc={rc} {us.value:.1f} us")
