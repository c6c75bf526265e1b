#!/usr/bin/env python
"""What the vendor GEMM library reaches on this box at the DiT block's shapes -- a practical ceiling to read lemas_k_bench against.

torch.matmul(bf16) = hipBLASLt/rocBLAS on ROCm: plain C = A W^T with NO bias / GELU / RoPE / gate+residual epilogue and a bf16 store,
i.e. strictly less work than the product kernels do per launch.  Measurement only: nothing in the product links these libraries.

    python tools/exp/library_gemm_ceiling.py            (on an MI355X box)
"""
import torch

SHAPES = [("QK   ", 2048, 1024), ("V/out", 1024, 1024), ("FF1  ", 2048, 1024), ("FF2  ", 1024, 2048)]


def bench(M, N, K, iters=30):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.matmul(a, w.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    print(torch.cuda.get_device_name(0), torch.__version__)
    for M in (1875, 3750, 9216, 18432, 30720):
        cells = []
        for name, N, K in SHAPES:
            us = bench(M, N, K)
            cells.append(f"{name} N={N} K={K}: {us:7.1f} us ({2.0 * M * N * K / us / 1e6:5.0f} TF)")
        print(f"M={M:6d}  " + "   ".join(cells), flush=True)
    # a big square GEMM: the library's own best case on this box
    for n in (4096, 8192):
        us = bench(n, n, n, 10)
        print(f"square {n}: {us:8.1f} us ({2.0 * n ** 3 / us / 1e6:5.0f} TF)")


if __name__ == "__main__":
    main()
