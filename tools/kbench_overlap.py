import ctypes as C, sys
sys.path.insert(0, ".")
import torch
from lemas_tts_amd import _lib
L = _lib.lib()
us = C.c_double()
for wide, narrow in ((6, 10), (2, 2), (2, 5), (7, 5)):
    r = []
    for mode in (0, 1):
        rc = L.lemas_k_bench_overlap(mode, 50, wide, narrow, C.byref(us))
        r.append(us.value if rc == 0 else float("nan"))
    print(f"gemm variants wide={wide} narrow={narrow}: serial {r[0]:.1f} us  concurrent {r[1]:.1f} us  ratio {r[1]/r[0]:.2f}")

for mode in (3, 2):
    rc = L.lemas_k_bench_overlap(mode, 50, 6, 10, C.byref(us))
    print(f"control: two half-size attentions, mode {mode} ({'two streams' if mode == 2 else 'one stream'}): {us.value:.1f} us")
