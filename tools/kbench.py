#!/usr/bin/env python
"""Micro-benchmarks of the step-loop kernels through lemas_k_bench (development aid; needs the MI355X).

    python tools/kbench.py gemm [--M 3840] [--tiles 0 16 17 18 22] [--f8]   # the five block GEMM shapes x tile shapes
    python tools/kbench.py attn [--N 1875 750] [--BH 16 256]                # attention launches
    python tools/kbench.py one <what> <M> <N> <K> [tile] [--iters 20]       # ONE kernel (for rocprofv3 --pmc passes)

Cells are average launch microseconds (algorithmic TFLOP/s).  Tiles: 0 = production choice, 16 = 256x128, 17 = 128x128,
18 = 128x64, 19 = 64x64, 22 = 256x256 (bf16 only)."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lemas_tts_amd import _lib  # noqa: E402

GEMMS = [("gemm_qk", 2048, 1024), ("gemm_v", 1024, 1024), ("gemm_gate", 1024, 1024), ("gemm_gelu", 2048, 1024), ("gemm_gate", 1024, 2048)]


def bench(L, what, M, N, K, iters, tile):
    us = C.c_double()
    rc = L.lemas_k_bench(what.encode(), M, N, K, iters, tile, C.byref(us))
    return None if rc != 0 else us.value


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    g = sub.add_parser("gemm")
    g.add_argument("--M", type=int, nargs="+", default=[3840])
    g.add_argument("--tiles", type=int, nargs="+", default=[0, 16, 17, 18, 22])
    g.add_argument("--f8", action="store_true")
    g.add_argument("--iters", type=int, default=50)
    a = sub.add_parser("attn")
    a.add_argument("--N", type=int, nargs="+", default=[1875])
    a.add_argument("--BH", type=int, nargs="+", default=[16, 32, 256])
    a.add_argument("--iters", type=int, default=20)
    a.add_argument("--variants", type=int, nargs="+", default=[19], help="attention schedule variants (csrc/attention.hip; 4112 + s = the "
                   "64-queries-per-wave kernel of attention_q64.hip, s = 0..3)")
    o = sub.add_parser("one")
    o.add_argument("what"); o.add_argument("M", type=int); o.add_argument("N", type=int); o.add_argument("K", type=int)
    o.add_argument("tile", type=int, nargs="?", default=0)
    o.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    L = _lib.testlib()
    if args.cmd == "gemm":
        for M in args.M:
            print(f"M={M}; columns = tiles {args.tiles}; cells = us (TFLOP/s)")
            for what, N, K in GEMMS:
                cells = []
                for t in args.tiles:
                    us = bench(L, ("f8_" if args.f8 else "") + what, M, N, K, args.iters, t)
                    cells.append("     n/a      " if us is None else f"{us:7.1f} ({2.0 * M * N * K / us / 1e6:5.0f})")
                print(f"  {what:10s} N={N:4d} K={K:4d}: " + "  ".join(cells))
    elif args.cmd == "attn":
        for n in args.N:
            for bh in args.BH:
                cells = []
                for v in args.variants:
                    us = bench(L, "attention", n, bh, 0, args.iters, v)
                    cells.append(f"v{v}:      n/a      " if us is None else f"v{v}: {us:6.1f} us ({4.0 * n * n * 64 * bh / us / 1e6:4.0f} TF)")
                    if v & 8192 and us is not None:      # attention_f8qk.h: the launch that quantises q and k to MXFP8, timed alone
                        qus = bench(L, "attention_qkquant", n, bh, 0, args.iters, v)
                        cells.append(f"(+ q,k -> MXFP8 launch {qus:5.1f} us)" if qus is not None else "(quant n/a)")
                print(f"attention N={n} BH={bh}: " + "   ".join(cells))
    else:
        us = bench(L, args.what, args.M, args.N, args.K, args.iters, args.tile)
        if us is None:
            raise SystemExit(_lib.testlib().lemas_last_error().decode())
        print(f"{args.what} M={args.M} N={args.N} K={args.K} tile={args.tile}: {us:.2f} us")


if __name__ == "__main__":
    main()
