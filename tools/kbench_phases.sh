# Phase timestamps of the GEMM launches (profiles/r02_kbench_phases.txt): a MEASUREMENT build of the library, on the GPU box only.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/kbench_phases.sh'
cd $GRAFT_REPO_ROOT
LEMAS_EXTRA_HIPCC_FLAGS=-DLEMAS_PHASE_TIMESTAMPS python -c "from lemas_tts_amd import build; build.build_library(force=True)"
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import ctypes as C, sys
sys.path.insert(0, '.')
from lemas_tts_amd import _lib
L = _lib.lib()
G = [("gemm_qk", 2048, 1024), ("gemm_v", 1024, 1024), ("gemm_gate", 1024, 1024), ("gemm_gelu", 2048, 1024), ("gemm_gate", 1024, 2048)]
for M, tiles in ((750, (18, 19)), (1875, (16, 17, 18)), (9216, (22,)), (30720, (22,))):
    for what, N, K in G:
        for t in tiles:
            us = C.c_double()
            L.lemas_k_bench(what.encode(), M, N, K, 20, t, C.byref(us))
            print(f"M={M} {what} N={N} K={K} tile {t}: {us.value:.1f} us", flush=True)
for n, bh in ((750, 16), (1875, 16), (1875, 32), (1125, 128), (1875, 256)):
    us = C.c_double()
    L.lemas_k_bench(b"attention", n, bh, 0, 20, 0, C.byref(us))
    print(f"attention N={n} BH={bh}: {us.value:.1f} us", flush=True)
PY
python -c "from lemas_tts_amd import build; build.build_library(force=True)"      # back to the product build
