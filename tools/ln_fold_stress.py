"""Stress one ln-fold producer -> consumer case many times (a rare wrong result was seen once in the suite): counts failures per case."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_gpu_13_option_kernels as T

cases = [(17, 4, 18, 2048), (17, 4, 17, 2048), (17, 1, 18, 2048), (17, 5, 18, 1024), (17, 4, 16, 2048), (17, 4, 22, 2048)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for (pt, ce, ct, nc) in cases:
    bad = 0
    worst = 0.0
    for it in range(n):
        try:
            T._ln_fold_case(pt, ce, ct, 1, 1875, 1024, nc, seed=ce * 100 + ct)
        except AssertionError as e:
            bad += 1
            try:
                worst = max(worst, float(e.args[0][-1]))
            except Exception:
                pass
    print(f"prod {pt} cons_epi {ce} cons_tile {ct}: {bad} / {n} failed (worst {worst:.3g})", flush=True)
