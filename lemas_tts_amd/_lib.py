"""ctypes binding of liblemas_hip.so (the C ABI declared in include/lemas_hip.h) and, for tests and measurement tools only, of
liblemas_hip_test.so (the lemas_k_* entry points of include/lemas_hip_test.h; it links against the product library).

There is NO CPU fallback: if the shared library is missing or no HIP device is present the
product path raises.  PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblemas_hip.so")
TEST_LIB_PATH = os.path.join(_HERE, "lib", "liblemas_hip_test.so")

_lib = None
_testlib = None
ABI_VERSION = 200          # include/lemas_hip.h: lemas_sample_args opens with struct_size


class DitConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dim", "depth", "heads", "dim_head", "ff_mult", "text_dim", "conv_layers", "mel_dim", "vocab_rows",
        "conv_pos_kernel", "conv_pos_groups", "time_freq_dim", "has_prosody")]


class ProsodyConfig(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("channels", C.c_int32 * 8), ("kernel_sizes", C.c_int32 * 8), ("dilations", C.c_int32 * 8),
                ("groups", C.c_int32 * 8), ("attention_channels", C.c_int32), ("res2net_scale", C.c_int32), ("se_channels", C.c_int32),
                ("global_context", C.c_int32), ("embed_dim", C.c_int32), ("input_dim", C.c_int32)]


class MdxConfig(C.Structure):
    """lemas_mdx_config (include/lemas_hip.h): the constructor arguments of the reference's ConvTDFNet (uvr5/lib_v5/mdxnet.py:37-49)."""
    _fields_ = [(n, C.c_int32) for n in ("dim_c", "dim_f", "dim_t", "num_blocks", "l", "g", "k", "bn", "bias", "norm")]


class SampleArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("batch", C.c_int32), ("frames", C.c_int32), ("cond_frames", C.c_int32), ("text_len", C.c_int32),
        ("steps", C.c_int32), ("cfg_strength", C.c_float),
        ("cond", C.c_void_p), ("cond_mask", C.c_void_p), ("text", C.c_void_p), ("seq_len", C.c_void_p),
        ("prosody", C.c_void_p), ("t_grid", C.POINTER(C.c_float)),
        ("y", C.c_void_p), ("out", C.c_void_p), ("trajectory", C.c_void_p), ("step_cond", C.c_void_p),
        ("prosody_text_only", C.c_int32), ("cond_rows", C.c_int32), ("y_init", C.c_void_p),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = C.sizeof(SampleArgs)      # ABI 200: checked by the library


class LemasError(RuntimeError):
    pass


def _bind(L, sig):
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args


def lib():
    """Load the PRODUCT library once.  Import torch first so the process-wide HIP runtime is torch's."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (must come first: one libamdhip64 per process)
    if not os.path.exists(LIB_PATH):
        raise LemasError(f"{LIB_PATH} is missing: build it with `python -m lemas_tts_amd.build` "
                         "(there is no CPU fallback for the acoustic path)")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)      # the test library resolves its references against this one
    L.lemas_version.restype = C.c_int
    if L.lemas_version() != ABI_VERSION:
        raise LemasError(f"{LIB_PATH} reports ABI {L.lemas_version()}, this binding is written against {ABI_VERSION}: rebuild with "
                         "`python -m lemas_tts_amd.build --force`")
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    sig = {
        "lemas_last_error": (C.c_char_p, []),
        "lemas_version": (C.c_int, []),
        "lemas_dit_create": (C.c_int, [C.POINTER(DitConfig), C.POINTER(vp)]),
        "lemas_dit_destroy": (None, [vp]),
        "lemas_dit_load_weight": (C.c_int, [vp, C.c_char_p, vp, C.POINTER(i64), i32]),
        "lemas_dit_load_weight_device": (C.c_int, [vp, C.c_char_p, vp, C.POINTER(i64), i32]),
        "lemas_dit_finalize": (C.c_int, [vp]),
        "lemas_dit_set_option": (C.c_int, [vp, C.c_char_p, i64]),
        "lemas_dit_health": (C.c_int, [vp]),
        "lemas_dit_get_stat": (C.c_int, [vp, C.c_char_p, C.POINTER(i64)]),
        "lemas_dit_sample": (C.c_int, [vp, C.POINTER(SampleArgs), vp]),
        "lemas_dit_prepare": (C.c_int, [vp, C.POINTER(SampleArgs), vp]),
        "lemas_dit_solve": (C.c_int, [vp, C.POINTER(SampleArgs), vp]),
        "lemas_dit_forward": (C.c_int, [vp, vp, i32, vp, vp]),
        "lemas_dit_profile_read": (C.c_int, [vp, vp, C.POINTER(C.c_double), C.POINTER(i64), i32]),
        "lemas_vocos_create": (C.c_int, [i32, i32, i32, i32, i32, i32, C.POINTER(vp)]),
        "lemas_vocos_destroy": (None, [vp]),
        "lemas_vocos_load_weight": (C.c_int, [vp, C.c_char_p, vp, C.POINTER(i64), i32]),
        "lemas_vocos_load_weight_device": (C.c_int, [vp, C.c_char_p, vp, C.POINTER(i64), i32]),
        "lemas_vocos_finalize": (C.c_int, [vp]),
        "lemas_vocos_decode": (C.c_int, [vp, vp, i32, i32, f32, vp, vp]),
        "lemas_vocos_decode_rows": (C.c_int, [vp, vp, i32, i32, i64, f32, vp, vp]),
        "lemas_vocos_set_option": (C.c_int, [vp, C.c_char_p, i64]),
        "lemas_mel_create": (C.c_int, [i32, i32, i32, i32, C.POINTER(vp)]),
        "lemas_mel_destroy": (None, [vp]),
        "lemas_mel_forward": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "lemas_prosody_create": (C.c_int, [vp, C.POINTER(vp)]),
        "lemas_prosody_destroy": (None, [vp]),
        "lemas_prosody_load_weight": (C.c_int, [vp, C.c_char_p, vp, C.POINTER(C.c_int64), i32]),
        "lemas_prosody_finalize": (C.c_int, [vp]),
        "lemas_prosody_fbank_frames": (C.c_int64, [C.c_int64]),
        "lemas_prosody_fbank": (C.c_int, [vp, vp, i32, vp, vp]),
        "lemas_prosody_encode": (C.c_int, [vp, vp, i32, vp, vp]),
        "lemas_resample_create": (C.c_int, [i32, i32, C.POINTER(vp)]),
        "lemas_resample_destroy": (None, [vp]),
        "lemas_resample_out_len": (C.c_int64, [vp, C.c_int64]),
        "lemas_resample_forward": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "lemas_stft_create": (C.c_int, [i32, i32, vp, C.POINTER(vp)]),
        "lemas_stft_destroy": (None, [vp]),
        "lemas_stft_ld": (C.c_int32, [vp]),
        "lemas_stft_frames": (C.c_int64, [vp, C.c_int64]),
        "lemas_stft_forward": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "lemas_stft_inverse": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "lemas_mdx_create": (C.c_int, [C.POINTER(MdxConfig), C.POINTER(vp)]),
        "lemas_mdx_destroy": (None, [vp]),
        "lemas_mdx_load_weight": (C.c_int, [vp, C.c_char_p, vp, C.POINTER(i64), i32]),
        "lemas_mdx_finalize": (C.c_int, [vp]),
        "lemas_mdx_set_option": (C.c_int, [vp, C.c_char_p, i64]),
        "lemas_mdx_forward": (C.c_int, [vp, vp, i32, vp, vp]),
        "lemas_mdx_tap": (C.c_int, [vp, C.c_char_p, vp]),
        "lemas_mdx_flops": (C.c_int64, [vp, i32]),
    }
    _bind(L, sig)
    _lib = L
    return L


class _WithTests:
    """The product library plus the lemas_k_* entry points of the TEST library under one handle, for tests and tools: attribute
    lookups starting with ``lemas_k_`` go to liblemas_hip_test.so, everything else to liblemas_hip.so (whose error buffer the test
    entry points share: the test library links against it)."""

    def __init__(self, prod, test):
        self._prod, self._test = prod, test

    def __getattr__(self, name):
        return getattr(self._test if name.startswith("lemas_k_") else self._prod, name)


def testlib():
    """Product + test entry points.  The product package never calls this; tests and tools/ do."""
    global _testlib
    if _testlib is not None:
        return _testlib
    prod = lib()
    if not os.path.exists(TEST_LIB_PATH):
        raise LemasError(f"{TEST_LIB_PATH} is missing: build it with `python -m lemas_tts_amd.build`")
    T = C.CDLL(TEST_LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int32
    sig = {
        "lemas_k_linear_bf16": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
        "lemas_k_linear_f32": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
        "lemas_k_attention": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp]),
        "lemas_k_attention_variant": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
        "lemas_k_mx_quant": (C.c_int, [vp, i32, i32, vp, vp, vp]),
        "lemas_k_w_quant_f8": (C.c_int, [vp, i32, i32, vp, vp, vp]),
        "lemas_k_ln_mod_f8": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, vp]),
        "lemas_k_outlier_rows": (C.c_int, [vp, vp, vp, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]),
        "lemas_k_linear_f8": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
        "lemas_k_ln_mod": (C.c_int, [vp, vp, vp, vp, i32, i32, vp]),
        "lemas_k_convpos": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "lemas_k_timeline": (C.c_int, [C.c_void_p, i32]),
        "lemas_k_build_flags": (C.c_int, []),
        "lemas_k_gemm_epi": (C.c_int, [i32, i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "lemas_k_gemm_gate_ln": (C.c_int, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "lemas_k_ln_fold_pair": (C.c_int, [i32, i32, i32] + [vp] * 12 + [i32] * 6 + [vp]),
        "lemas_k_bench": (C.c_int, [C.c_char_p, i32, i32, i32, i32, i32, C.POINTER(C.c_double)]),
    }
    _bind(T, sig)
    _testlib = _WithTests(prod, T)
    return _testlib


EXPORTED = [      # include/lemas_hip.h: the product library
    "lemas_last_error", "lemas_version", "lemas_dit_create", "lemas_dit_destroy", "lemas_dit_load_weight", "lemas_dit_load_weight_device", "lemas_vocos_load_weight_device",
    "lemas_dit_finalize", "lemas_dit_set_option", "lemas_dit_get_stat", "lemas_dit_health", "lemas_dit_sample", "lemas_dit_prepare", "lemas_dit_solve",
    "lemas_dit_forward", "lemas_dit_profile_read", "lemas_vocos_create", "lemas_vocos_destroy",
    "lemas_vocos_load_weight", "lemas_vocos_finalize", "lemas_vocos_decode", "lemas_vocos_decode_rows", "lemas_vocos_set_option", "lemas_mel_create", "lemas_mel_destroy",
    "lemas_mel_forward", "lemas_resample_create", "lemas_resample_destroy", "lemas_resample_out_len", "lemas_resample_forward",
    "lemas_prosody_create", "lemas_prosody_destroy", "lemas_prosody_load_weight", "lemas_prosody_finalize", "lemas_prosody_fbank_frames",
    "lemas_prosody_fbank", "lemas_prosody_encode",
    "lemas_stft_create", "lemas_stft_destroy", "lemas_stft_ld", "lemas_stft_frames", "lemas_stft_forward", "lemas_stft_inverse",
    "lemas_mdx_create", "lemas_mdx_destroy", "lemas_mdx_load_weight", "lemas_mdx_finalize", "lemas_mdx_set_option", "lemas_mdx_forward", "lemas_mdx_tap", "lemas_mdx_flops",
]
EXPORTED_TEST = [  # include/lemas_hip_test.h: the test library
    "lemas_k_linear_bf16", "lemas_k_linear_f32", "lemas_k_attention", "lemas_k_attention_variant", "lemas_k_ln_mod", "lemas_k_convpos", "lemas_k_bench", "lemas_k_gemm_epi",
    "lemas_k_gemm_gate_ln", "lemas_k_ln_fold_pair", "lemas_k_timeline", "lemas_k_build_flags", "lemas_k_mx_quant", "lemas_k_w_quant_f8", "lemas_k_ln_mod_f8", "lemas_k_outlier_rows", "lemas_k_linear_f8",
]


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().lemas_last_error()
        raise LemasError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
