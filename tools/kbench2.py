import ctypes as C, sys
sys.path.insert(0, ".")
import torch
from lemas_tts_amd import _lib
L = _lib.lib()
M = 3840
def run(what, N, K, v, iters=100):
    us = C.c_double()
    rc = L.lemas_k_bench(what.encode(), M, N, K, iters, v, C.byref(us))
    return us.value if rc == 0 else float("nan")
for v in (10, 6):
    for what, N in (("gemm_none", 1024), ("gemm_v", 1024), ("gemm_gate", 1024), ("gemm_none", 2048), ("gemm_gelu", 2048), ("gemm_qk", 2048)):
        print(f"v{v} {what:12s} N={N}: " + "  ".join(f"K={K}: {run(what, N, K, v):6.1f}us" for K in (64, 128, 256, 512, 1024, 2048)))
