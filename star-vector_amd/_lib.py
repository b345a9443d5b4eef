"""ctypes binding of libstarvector_hip.so (include/starvector_hip.h: the product ABI; include/starvector_hip_debug.h: the test and
measurement surface of the same library).

The product path has NO fallback: if the HIP library is missing or fails to load this module raises,
and every public entry point of the package goes through it.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libstarvector_hip.so")
HEADER_PATH = os.path.normpath(os.path.join(HERE, "..", "include", "starvector_hip.h"))
DEBUG_HEADER_PATH = os.path.normpath(os.path.join(HERE, "..", "include", "starvector_hip_debug.h"))

ABI_VERSION = 8
SV_DTYPE_BF16, SV_DTYPE_F32 = 0, 1
SV_NORM_LAYER, SV_NORM_BATCH = 0, 1
SV_ARCH_V1, SV_ARCH_V2 = 0, 1
ACT = {"none": 0, "quickgelu": 1, "swish": 2, "gelu_tanh": 3}


class SvConfig(C.Structure):
    _fields_ = [
        ("image_size", C.c_int32), ("patch_size", C.c_int32), ("vit_width", C.c_int32),
        ("vit_layers", C.c_int32), ("vit_heads", C.c_int32), ("adapter_norm", C.c_int32),
        ("hidden", C.c_int32), ("n_layer", C.c_int32), ("n_head", C.c_int32), ("n_inner", C.c_int32),
        ("vocab", C.c_int32), ("n_positions", C.c_int32), ("max_batch", C.c_int32),
        ("max_seq_len", C.c_int32), ("ln_eps", C.c_float), ("device", C.c_int32),
        ("arch", C.c_int32), ("n_kv_head", C.c_int32), ("rope_theta", C.c_float), ("vit_mlp", C.c_int32),
        ("vit_eps", C.c_float), ("sliding_window", C.c_int32), ("weight_dtype", C.c_int32), ("exclusive_device", C.c_int32),
    ]


class SvSampling(C.Structure):
    _fields_ = [
        ("do_sample", C.c_int32), ("temperature", C.c_float), ("top_p", C.c_float),
        ("max_length", C.c_int32), ("eos_token_id", C.c_int32), ("pad_token_id", C.c_int32),
        ("n_stop", C.c_int32), ("stop_ids", C.POINTER(C.c_int32)), ("seed", C.c_uint64),
        ("sync_every", C.c_int32), ("repetition_penalty", C.c_float),
        ("num_beams", C.c_int32), ("length_penalty", C.c_float), ("early_stopping", C.c_int32),
        ("top_k", C.c_int32), ("on_tokens", C.c_void_p), ("user_data", C.c_void_p),
        ("min_new_tokens", C.c_int32),
    ]


TOKEN_CALLBACK = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_int32)


class SvCbRequest(C.Structure):
    _fields_ = [
        ("do_sample", C.c_int32), ("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_int32),
        ("seed", C.c_uint64), ("max_new_tokens", C.c_int32), ("eos_token_id", C.c_int32), ("pad_token_id", C.c_int32),
        ("min_new_tokens", C.c_int32), ("repetition_penalty", C.c_float), ("n_stop", C.c_int32),
        ("stop_ids", C.c_int32 * 16),
    ]


class SvBeamConfig(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("num_beams", C.c_int32), ("vocab", C.c_int32), ("max_new", C.c_int32),
        ("eos_token_id", C.c_int32), ("pad_token_id", C.c_int32), ("early_stopping", C.c_int32),
        ("length_penalty", C.c_float), ("repetition_penalty", C.c_float), ("n_stop", C.c_int32),
        ("stop_ids", C.POINTER(C.c_int32)), ("do_sample", C.c_int32), ("temperature", C.c_float),
        ("top_p", C.c_float), ("top_k", C.c_int32), ("seed", C.c_uint64), ("min_new_tokens", C.c_int32),
    ]


_P = C.c_void_p
_I = C.c_int32
_F = C.c_float

# name -> (restype, argtypes): exactly the prototypes of include/starvector_hip.h
PRODUCT_PROTOTYPES = {
    "sv_abi_version": (_I, []),
    "sv_last_error": (C.c_char_p, []),
    "sv_config_default_1b": (None, [C.POINTER(SvConfig)]),
    "sv_config_default_8b": (None, [C.POINTER(SvConfig)]),
    "sv_create": (_I, [C.POINTER(SvConfig), C.POINTER(_P)]),
    "sv_destroy": (_I, [_P]),
    "sv_load_weight": (_I, [_P, C.c_char_p, _P, _I, _I, C.POINTER(C.c_int64), _P]),
    "sv_weights_complete": (_I, [_P]),
    "sv_weight_count": (_I, [_P]),
    "sv_weight_info": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(C.c_int64), C.POINTER(_I), C.POINTER(_I)]),
    "sv_encode_image": (_I, [_P, _P, _I, _P, _P]),
    "sv_adapter": (_I, [_P, _P, _I, _P, _P]),
    "sv_embed_tokens": (_I, [_P, _P, _I, _P, _P]),
    "sv_adapter_into": (_I, [_P, _P, _I, _P, _I, _P]),
    "sv_embed_tokens_into": (_I, [_P, _P, _I, _I, _P, _I, _I, _P]),
    "sv_preprocess_image": (_I, [_P, _I, _I, _I, _I, _I, C.POINTER(_F), C.POINTER(_F), _P, _P]),
    "sv_preprocess_workspace_bytes": (C.c_int64, [C.POINTER(_I), C.POINTER(_I), _I, _I, _I]),
    "sv_preprocess_images": (_I, [C.POINTER(_P), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), _I, _I, _I, C.POINTER(_F),
                                  C.POINTER(_F), _P, _P, C.c_int64, _P]),
    "sv_prefill": (_I, [_P, _P, _I, _I, _P, _P]),
    "sv_forward_logits": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "sv_decode_step": (_I, [_P, _P, _I, _P, _P]),
    "sv_generate": (_I, [_P, _P, _I, _I, C.POINTER(SvSampling), _P, C.POINTER(_I), _P]),
    "sv_cb_admit": (_I, [_P, _P, _I, _I, C.POINTER(SvCbRequest), C.POINTER(_I), _P]),
    "sv_cb_step": (_I, [_P, _I, C.POINTER(_I), _P]),
    "sv_cb_poll": (_I, [_P, C.POINTER(_I), C.POINTER(_I), _I]),
    "sv_cb_read": (_I, [_P, _I, _I, _I, C.POINTER(C.c_int64)]),
    "sv_cb_release": (_I, [_P, _I]),
    "sv_cb_reset": (_I, [_P]),
    "sv_beam_create": (_I, [C.POINTER(SvBeamConfig), C.POINTER(_P)]),
    "sv_beam_destroy": (_I, [_P]),
    "sv_beam_step": (_I, [_P, _P, _I, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_F), _P]),
    "sv_beam_finalize": (_I, [_P, C.POINTER(C.c_int64), C.POINTER(_I), C.POINTER(_F), _P]),
    "sv_beam_history": (_I, [_P, C.POINTER(_I), C.POINTER(_I), _I, C.POINTER(_I), C.POINTER(_I)]),
    "sv_last_timing": (_I, [_P, C.POINTER(C.c_double)]),
}

# the test / measurement surface: include/starvector_hip_debug.h (same library)
DEBUG_PROTOTYPES = {
    "sv_debug_resample_coeffs": (_I, [_I, _I, C.POINTER(_I), C.POINTER(_I), _I]),
    "sv_debug_gemm_plan": (_I, [_I, _I, _I, _I, C.POINTER(_I)]),
    "sv_debug_skinny_plan": (_I, [_I, _I, _I, _I, _I, C.POINTER(_I)]),
    "sv_debug_set_exp": (_I, [_P, _I]),
    "sv_debug_set_col_tiles": (_I, [_I]),
    "sv_debug_set_skinny_form": (_I, [_I]),
    "sv_debug_set_gemm_form": (_I, [_I]),
    "sv_debug_set_linear_seq_rows": (_I, [_I]),
    "sv_debug_gemm_seq_form": (_I, [_I, _I, _I, _I]),
    "sv_debug_attn_plan": (_I, [_I, _I, _I, C.POINTER(_I)]),
    "sv_debug_rowln_plan": (_I, [_I, _I, _I, _I, _I, C.POINTER(_I)]),
    "sv_debug_rowln_occupancy": (_I, [_I, C.POINTER(_I)]),
    "sv_debug_step_plan": (_I, [_P, C.POINTER(_I)]),
    "sv_debug_attn_trace": (_I, [_P, C.POINTER(C.c_int64), _I]),
    "sv_debug_mlp_trace": (_I, [_P, C.POINTER(C.c_int64), _I]),
    "sv_debug_xcc_map": (_I, [_P, _I, _I, C.POINTER(_I)]),
    "sv_debug_occupy_cus": (_I, [_P, _I, _I, _I]),
    "sv_debug_gemm_trace": (_I, [_I, _I, _I, _I, _I, C.POINTER(C.c_int64), _I]),
    "sv_debug_kv_load": (_I, [_P, _I, _P, _I, _I, _P, _P]),
    "sv_debug_attn_decode": (_I, [_P, _I, _P, _I, _P, _I, _P]),
    "sv_debug_decode_plan": (_I, [_I, _I, _I, _I, _I, _I, C.POINTER(_I)]),
    "sv_profile_decode_step": (_I, [_P, _I, _I, C.POINTER(C.c_double), _P]),
    "sv_profile_ttft": (_I, [_P, _P, _I, _P, _I, _I, C.POINTER(C.c_double), _P]),
    "sv_op_layernorm": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "sv_op_linear": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sv_op_linear_skinny": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sv_op_linear_skinny_fp8": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "sv_op_linear_skinny_epi": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sv_op_lm_head_argmax": (_I, [_P, _P, _P, C.POINTER(_I), _I, _I, _I, _P]),
    "sv_op_decode_proj_fold": (_I, [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sv_bench_linear": (_I, [_I, _I, _I, _I, _I, _I, C.POINTER(C.c_double), _P]),
    "sv_bench_decode_linear": (_I, [_I, _I, _I, _I, _I, _I, C.POINTER(C.c_double), _P]),
    "sv_op_cvt_bf16_hw": (_I, [_P, _P, C.c_int64, _P]),
    "sv_op_attention": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P]),
    "sv_op_plane_layernorm": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "sv_op_argmax": (_I, [_P, _I, _I, _I, _P, _P]),
    "sv_op_sample_top_p": (_I, [_P, _I, _I, _I, _F, _F, C.c_uint64, _I, _P, _P]),
    "sv_op_sample": (_I, [_P, _I, _I, _I, _F, _I, _F, C.c_uint64, _I, _P, _P]),
}

PROTOTYPES = {**PRODUCT_PROTOTYPES, **DEBUG_PROTOTYPES}

_lib = None


class StarVectorHipError(RuntimeError):
    pass


class StarVectorBusy(StarVectorHipError):
    """sv_cb_admit: no free slot / KV pages right now (SV_EBUSY); release finished requests and retry."""


def load() -> C.CDLL:
    """Load the HIP library or raise.  Never falls back to another implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise StarVectorHipError(
            f"{LIB_PATH} is missing: the HIP engine has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `python star-vector_amd/build.py`); "
            "there is no CPU/PyTorch fallback for this path.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64 missing
        raise StarVectorHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise StarVectorHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.sv_abi_version() != ABI_VERSION:
        raise StarVectorHipError(f"ABI version mismatch: library {lib.sv_abi_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    """Map C return codes onto the exceptions the reference's callers already catch
    (serve/model_worker.py:183-207 handles ValueError / RuntimeError)."""
    if rc == 0:
        return
    msg = load().sv_last_error().decode("utf-8", "replace")
    text = f"{what}: {msg}" if what else msg
    if rc == -22:
        raise ValueError(text)
    if rc == -2:
        raise KeyError(text)
    if rc == -95:
        raise NotImplementedError(text)
    if rc == -16:
        raise StarVectorBusy(text)
    raise StarVectorHipError(f"{text} (code {rc})")
