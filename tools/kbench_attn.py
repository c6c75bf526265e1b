import ctypes as C, sys
sys.path.insert(0, ".")
import torch
from lemas_tts_amd import _lib
L = _lib.lib()
us = C.c_double()
for n, bh in ((1875, 32), (1125, 256), (750, 32)):
    for v in (1, 2, 3):
        rc = L.lemas_k_bench(b"attention", n, bh, 0, 20, v, C.byref(us))
        print(f"attention N={n} BH={bh} variant {v}: {us.value:.1f} us ({4.0 * n * n * 64 * bh / (us.value * 1e-6) / 1e12:.0f} TF)")
