#!/usr/bin/env python
"""Stress of the loader-wave GEMM tiles (gemm_bf16.hip gemm_body NL; tiles 28 / 30 / 31): the bit-identity cases of
tests/test_gpu_01_kernels.py::test_gemm_loader_wave_tiles_are_bit_identical repeated many times, alone and beside a second stream that keeps the
chip busy with the other lane's kind of work -- a race between a loader's refill and a slow compute wave's fragment reads would show as a
differing bit in SOME repetition.

    python tools/exp/loader_tiles_stress.py [reps]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch  # noqa: E402

import test_gpu_01_kernels as T  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    cases = [(26, 28, T.EPI_GATE, 1024, 2048, 1, 1875), (26, 28, T.EPI_GATE, 1024, 1024, 3, 700), (19, 31, T.EPI_GATE, 1024, 2048, 1, 750),
             (18, 30, T.EPI_GELU, 2048, 1024, 1, 750), (26, 28, T.EPI_QK, 2048, 1024, 2, 333), (19, 31, T.EPI_VT, 1024, 1024, 1, 130)]
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda:0", dtype=torch.bfloat16)
    bad = 0
    for load in (False, True):
        for r in range(reps):
            if load:
                with torch.cuda.stream(side):
                    for _ in range(2):
                        a @ a
            for c in cases:
                try:
                    T.test_gemm_loader_wave_tiles_are_bit_identical.__wrapped__(*c) if hasattr(T.test_gemm_loader_wave_tiles_are_bit_identical, "__wrapped__") \
                        else T.test_gemm_loader_wave_tiles_are_bit_identical(*c)
                except AssertionError as e:
                    bad += 1
                    print("MISMATCH", load, r, c, e)
        torch.cuda.synchronize()
        print(f"beside a busy stream: {load}; {reps} x {len(cases)} cases, mismatches so far: {bad}")
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
