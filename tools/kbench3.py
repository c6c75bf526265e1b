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
for v in (6, 10, 12, 11):
    for what in ("gemm_none", "gemm_nodma"):
        print(f"v{v} {what:11s} N=2048: " + "  ".join(f"K={K}: {run(what, 2048, K, v):6.1f}us" for K in (1024, 2048, 4096)))
