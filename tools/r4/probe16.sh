#!/bin/bash
# round 4, probe 16: in-kernel phase stamps of the GEMM launches after the row-window epilogues (measurement build prebuilt in tools/_alt/phases)
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT/tools/_alt/phases
python - > $O/r04p16_phases.txt 2>&1 <<'PY'
import ctypes as C, sys
sys.path.insert(0, '.')
from lemas_tts_amd import _lib
L = _lib.testlib()
G = [("gemm_qk", 2048, 1024), ("gemm_v", 1024, 1024), ("gemm_gate", 1024, 1024), ("gemm_gelu", 2048, 1024), ("gemm_gate", 1024, 2048)]
for M, tiles in ((1875, (0,)), (9216, (22,)), (30720, (22,))):
    for what, N, K in G:
        for t in tiles:
            us = C.c_double()
            L.lemas_k_bench(what.encode(), M, N, K, 20, t, C.byref(us))
            print(f"M={M} {what} N={N} K={K} tile {t}: {us.value:.1f} us", flush=True)
PY
grep -v amdgpu.ids $O/r04p16_phases.txt
