"""ctypes binding of libpm_mi355x.so (include/pm_mi355x.h).

The library is the product: there is NO fallback.  ``lib()`` raises if the shared object is
missing, and every ``ops.*`` wrapper raises ``RuntimeError`` on a non-zero status.
"""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_PKG_ROOT = os.path.dirname(os.path.dirname(_HERE))  # .../pytorch-models_amd
LIB_PATH = os.environ.get("PM_MI355X_LIB") or os.path.join(REPO_PKG_ROOT, "csrc", "build", "libpm_mi355x.so")
HEADER_PATH = os.path.join(os.path.dirname(REPO_PKG_ROOT), "include", "pm_mi355x.h")

PM_BF16, PM_F32 = 0, 1
ACT = dict(none=0, gelu=1, approximate_gelu=2, relu=3, silu=4)

_p, _i, _l, _f = c_void_p, c_int, c_int64, c_float
# name -> argtypes; must list every function declared in include/pm_mi355x.h (tests/test_abi.py checks)
SIGNATURES = {
    "pm_abi_version": ([], c_int),
    "pm_strerror": ([_i], c_char_p),
    "pm_linear_bf16": ([_p, _l, _p, _l, _p, _p, _l, _i, _p, _l, _i, _l, _l, _l, _i, _p], c_int),
    "pm_linear_bf16_ex": ([_p, _l, _l, _l, _p, _l, _p, _p, _l, _i, _l, _p, _l, _i, _l, _l, _l, _i, _p], c_int),
    "pm_linear_bf16_ln": ([_p, _l, _l, _l, _p, _l, _p, _p, _l, _i, _l, _p, _l, _i, _l, _l, _l, _i, _p, _p, _p, _p], c_int),
    "pm_linear_f32": ([_p, _l, _l, _l, _p, _l, _p, _p, _l, _l, _p, _l, _l, _l, _l, _i, _p], c_int),
    "pm_attention_generic_f32": ([_p, _l, _l, _p, _l, _l, _p, _l, _l, _p, _l, _l, _l, _l, _l, _l, _l, _i, _p, _l, _l, _l, _p], c_int),
    "pm_linear_bf16_ws": ([_p, _l, _l, _l, _p, _l, _p, _p, _l, _i, _l, _p, _l, _i, _l, _l, _l, _i, _p, _p, _p, _p, _l, _p], c_int),
    "pm_linear_ws_bytes": ([], c_int64),
    "pm_ln_stats_finalize": ([_p, _p, _l, _l, _f, _p], c_int),
    "pm_linear_ln_supported": ([_l, _l, _l, _i, _i], c_int),
    "pm_stft_mel": ([_p, _l, _l, _l, _p, _p, _l, _l, _l, _i, _p, _p, _p, _l, _p, _p, _p], c_int),
    "pm_stft_mel_folded": ([_p, _l, _l, _l, _p, _p, _l, _l, _l, _i, _p, _p, _p, _l, _p, _p, _p], c_int),
    "pm_logmel_finalize": ([_p, _p, _l, _l, _p], c_int),
    "pm_whisper_stem1": ([_p, _p, _p, _p, _l, _l, _l, _l, _l, _p], c_int),
    "pm_embed_tokens": ([_p, _p, _p, _p, _i, _l, _l, _l, _l, _l, _p], c_int),
    "pm_embed_tokens_f32": ([_p, _p, _p, _p, _l, _l, _l, _l, _l, _p], c_int),
    "pm_dec_embed": ([_p, _p, _p, _p, _p, _l, _l, _l, _p], c_int),
    "pm_conv2d_nhwc_bf16": ([_p, _l, _l, _l, _l, _p, _p, _p, _p, _l, _l, _l, _l, _l, _l, _i, _p], c_int),
    "pm_mean_rows_bf16": ([_p, _p, _l, _l, _l, _p], c_int),
    "pm_dec_linear": ([_p, _l, _p, _p, _f, _p, _l, _p, _p, _l, _p, _l, _l, _l, _l, _i, _i, _p, _p, _l, _l, _l, _p, _p, _p, _p], c_int),
    "pm_dec_argmax_tile": ([_l], c_int),
    "pm_dec_attention": ([_p, _p, _p, _l, _l, _l, _p, _l, _l, _p, _l, _l, _p], c_int),
    "pm_dec_linear_ksplit": ([_p, _l, _p, _l, _p, _p, _l, _p, _l, _l, _l, _l, _i, _l, _p, _p, _p], c_int),
    "pm_dec_attention_fused": ([_p, _l, _p, _p, _f, _p, _p, _p, _p, _l, _l, _l, _p, _l, _l, _p, _l, _l, _i, _p], c_int),
    "pm_dec_attention_fused_kv32": ([_p, _l, _p, _p, _f, _p, _p, _p, _p, _l, _l, _l, _p, _l, _l, _p, _l, _l, _i, _p], c_int),
    "pm_dec_attention_chain": ([_p, _l, _p, _p, _f, _p, _p, _p, _p, _l, _l, _l, _p, _l, _l, _l, _l, _i, _i, _p, _l, _l, _l, _p, _p, _p, _p, _p, _p], c_int),
    "pm_dec_linear_kparts": ([_p, _l, _p, _l, _p, _l, _l, _l, _l, _l, _l, _p], c_int),
    "pm_dec_argmax_reduce": ([_p, _p, _l, _p, _p, _l, _p, _p, _l, _p, _l, _p], c_int),
    "pm_dec_advance": ([_p, _p], c_int),
    "pm_dec_whisper_rules": ([_p, _l, _l, _p, _l, _p, _l, _l, _l, _l, _l, _p, _l, _p, _l, _l, _p], c_int),
    "pm_dec_sample_topk": ([_p, _l, _l, _l, ctypes.c_uint64, _p, _p, _l, _p, _p, _l, _p, _p, _p, _l, _p, _l, _p], c_int),
    "pm_dec_next_token": ([_p, _p, _l, _p, _p, _l, _p, _p, _l, _p, _p, _p, _p, _l, _l, _p, _l, _p], c_int),
    "pm_layernorm": ([_p, _l, _i, _p, _p, _f, _p, _l, _i, _l, _l, _p], c_int),
    "pm_layernorm_ex": ([_p, _l, _i, _p, _p, _f, _i, _p, _l, _i, _p, _l, _i, _l, _l, _p], c_int),
    "pm_rmsnorm": ([_p, _l, _i, _p, _f, _p, _l, _i, _l, _l, _p], c_int),
    "pm_geglu": ([_p, _l, _p, _l, _l, _l, _p], c_int),
    "pm_w2v_stem0_scratch_floats": ([_l, _l], c_int64),
    "pm_w2v_stem0": ([_p, _p, _p, _i, _p, _p, _f, _p, _p, _p, _l, _l, _l, _l, _l, _p], c_int),
    "pm_group_windows": ([_p, _l, _i, _p, _l, _l, _l, _l, _l, _l, _l, _p], c_int),
    "pm_grouped_conv_supported": ([_l, _l], c_int),
    "pm_grouped_conv_bf16": ([_p, _p, _p, _p, _l, _p, _l, _l, _l, _l, _l, _l, _l, _l, _l, _i, _p], c_int),
    "pm_avgpool_time2": ([_p, _p, _l, _l, _l, _p], c_int),
    "pm_attention_bf16": ([_p, _l, _l, _p, _l, _l, _p, _l, _l, _p, _l, _l, _l, _l, _l, _l, _i, _p], c_int),
    "pm_attention_bias_bf16": ([_p, _l, _l, _p, _l, _l, _p, _l, _l, _p, _l, _l, _l, _l, _l, _l, _i, _p, _l, _l, _l, _p], c_int),
    "pm_attention_generic_bf16": ([_p, _l, _l, _p, _l, _l, _p, _l, _l, _p, _l, _l, _l, _l, _l, _l, _l, _i, _p, _l, _l, _l, _p], c_int),
    "pm_vit_tokens": ([_p, _p, _p, _p, _p, _p, _l, _l, _l, _l, _l, _p], c_int),
    "pm_vit_tokens_generic": ([_p, _p, _l, _p, _p, _p, _p, _l, _l, _l, _l, _l, _p], c_int),
}

# Entry points of the experiment kernels (include/pm_mi355x_experiments.h; csrc/experiments/): present only in
# build/libpm_mi355x_exp.so (`make experiments`, loaded through PM_MI355X_LIB).  Bound when the loaded library has them.
EXPERIMENT_SIGNATURES = {
    "pm_dec_attention_fused_v2": ([_p, _l, _p, _p, _f, _p, _p, _p, _p, _l, _l, _l, _p, _l, _l, _p, _l, _l, _i, _p], c_int),
    "pm_dec_layers": ([_p, _l, _l, _l, _l, _l, _l, _l, _i, _l, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p], c_int),
    "pm_dec_layers_grid": ([], c_int),
    "pm_gemm8ph_bench": ([_p, _l, _p, _l, _p, _l, _l, _l, _l, _p], c_int),
}


class DecLayer(ctypes.Structure):
    """pm_dec_layer_t of include/pm_mi355x.h (field order is the ABI)."""
    _fields_ = [(n, c_void_p) for n in (
        "sa_g", "sa_b", "w_qkv", "b_qkv", "kc", "vc", "w_so", "b_so", "ca_g", "ca_b", "w_q", "b_q", "cross_kv", "w_co", "b_co",
        "mlp_g", "mlp_b", "w1", "b1", "w2", "b2")] + [(n, c_float) for n in ("sa_eps", "ca_eps", "mlp_eps", "reserved_")]


_lib = None


def header_functions() -> list[str]:
    """Names of the functions include/pm_mi355x.h declares."""
    src = open(HEADER_PATH).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", src)))


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the MI355X HIP library has not been built "
                "(run `make -C pytorch-models_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "This package has no CPU or eager fallback."
            )
        L = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
            fn.argtypes = argtypes
            fn.restype = restype
        for name, (argtypes, restype) in EXPERIMENT_SIGNATURES.items():
            if hasattr(L, name):
                fn = getattr(L, name)
                fn.argtypes = argtypes
                fn.restype = restype
        if L.pm_abi_version() != 1:
            raise RuntimeError(f"libpm_mi355x ABI {L.pm_abi_version()} != binding ABI 1: rebuild the library")
        _lib = L
    return _lib


def has_experiments() -> bool:
    """True when the loaded library is the experiments build (csrc/experiments/ linked in)."""
    return hasattr(lib(), "pm_dec_layers")


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what}: pm_mi355x error {code}: {lib().pm_strerror(code).decode()}")
