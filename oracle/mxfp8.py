"""Torch restatement of the MXFP8 quantisers of the fp8 GEMM path (test infrastructure, part of the oracle: only
tests/, smoke() and bench.py's cpu_baseline leg may import it; the product never does).

There is no reference code for this variant -- the reference runs fp32/fp16 only; BASELINE config 5 asks for "fp8 MFMA
weights" -- so this file DEFINES the quantisation scheme the HIP path implements, and OracleDiT(fp8=True) applies it
to the fp32 oracle so that kernel correctness (HIP vs this emulation) and quantisation loss (emulation vs reference)
can be told apart.

Activations: OCP microscaling -- e4m3 elements, one E8M0 (power of two) scale per 32 consecutive K: the smallest
2^e with amax * 2^-e <= 448.  Weights: e4m3 with one fp32 scale per output channel (amax / 448).  The arithmetic is
written to be bit-identical to lemas_tts_amd/csrc/common.h (mx_exponent / pack_fp8x4) and
norm_elementwise.hip (w_quant_f8_kernel)."""
import numpy as np
import torch

_INV448 = torch.tensor(np.float32(1.0) / np.float32(448.0))


def mx_exponent(amax: torch.Tensor) -> torch.Tensor:
    b = (amax.float() * _INV448).contiguous().view(torch.int32)
    e = ((b >> 23) & 255) - 127 + ((b & 0x7FFFFF) != 0).to(torch.int32)
    return e.clamp(-120, 120)


def mx_quant(x: torch.Tensor):
    """x [M,K] fp32 -> (bytes uint8 [M,K], scales uint8 [M,K/32], dequantised fp32 [M,K])"""
    M, K = x.shape
    xb = x.float().reshape(M, K // 32, 32)
    e = mx_exponent(xb.abs().amax(-1))
    one = torch.ones((), dtype=torch.float32)
    q = (xb * torch.ldexp(one, -e)[..., None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    deq = q.float() * torch.ldexp(one, e)[..., None]
    return q.view(torch.uint8).reshape(M, K), (e + 127).to(torch.uint8), deq.reshape(M, K)


def mx_dequant(q8: torch.Tensor, mx: torch.Tensor) -> torch.Tensor:
    M, K = q8.shape
    v = q8.contiguous().view(torch.float8_e4m3fn).float().reshape(M, K // 32, 32)
    return (v * torch.ldexp(torch.ones((), dtype=torch.float32), mx.to(torch.int32) - 127)[..., None]).reshape(M, K)


def w_quant(w: torch.Tensor):
    """w [N,K] fp32 -> (bytes uint8, scale fp32 [N], dequantised fp32)"""
    amax = w.float().abs().amax(-1)
    sc = torch.where(amax > 0, amax * _INV448, torch.ones_like(amax))
    q = (w.float() / sc[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), sc, q.float() * sc[:, None]
