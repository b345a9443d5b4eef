"""CPU: the C-ABI library builds, loads and exports every symbol declared in include/*.h; argument
validation paths that need no GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from starvector_amd import _lib
    return _lib.load()


HEADERS = ("starvector_hip.h", "starvector_hip_debug.h")          # product ABI | test and measurement surface


def _header_text(name):
    text = open(os.path.join(ROOT, "include", name)).read()
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def _declared_functions(name):
    return sorted(set(re.findall(r"\b(sv_[a-z0-9_]+)\s*\(", _header_text(name))))


def test_header_symbols_exported(lib):
    for h in HEADERS:
        names = _declared_functions(h)
        assert len(names) >= 20
        for n in names:
            assert hasattr(lib, n), f"{n} declared in include/{h} but not exported"


def test_binding_matches_header(lib):
    from starvector_amd import _lib
    assert sorted(_lib.PRODUCT_PROTOTYPES) == _declared_functions("starvector_hip.h")
    assert sorted(_lib.DEBUG_PROTOTYPES) == _declared_functions("starvector_hip_debug.h")


def test_product_header_carries_no_test_surface():
    """the product header is what a reference-side binding binds (SURVEY.md 8b's export list + continuous
    batching, the beam scorer, pre-processing); plans, traces, micro-benchmarks, single operators and the A/B switches live in
    the debug header, which includes the product one."""
    prod = _declared_functions("starvector_hip.h")
    assert not [n for n in prod if n.startswith(("sv_debug_", "sv_op_", "sv_bench_", "sv_profile_"))], prod
    for need in ("sv_create", "sv_destroy", "sv_load_weight", "sv_encode_image", "sv_adapter", "sv_embed_tokens", "sv_prefill",
                 "sv_decode_step", "sv_generate", "sv_cb_admit", "sv_beam_step", "sv_preprocess_images", "sv_last_error"):
        assert need in prod, need
    dbg = _declared_functions("starvector_hip_debug.h")
    assert all(n.startswith(("sv_debug_", "sv_op_", "sv_bench_", "sv_profile_")) for n in dbg), dbg
    assert '#include "starvector_hip.h"' in open(os.path.join(ROOT, "include", "starvector_hip_debug.h")).read()


def test_binding_arity_and_struct_fields_match_header():
    """Every ctypes prototype has as many arguments as the C declaration, and the ctypes mirrors of the three structs list
    the header's fields in the header's order (an ABI drift here corrupts arguments silently)."""
    from starvector_amd import _lib
    text = _header_text("starvector_hip.h") + _header_text("starvector_hip_debug.h")
    seen = set()
    for name, params in re.findall(r"\b(sv_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        if name not in _lib.PROTOTYPES:
            continue                                            # typedef'd callback etc.
        params = params.strip()
        n = 0 if params in ("", "void") else len([p for p in params.split(",") if p.strip()])
        assert n == len(_lib.PROTOTYPES[name][1]), f"{name}: header has {n} parameters, binding {len(_lib.PROTOTYPES[name][1])}"
        seen.add(name)
    assert seen == set(_lib.PROTOTYPES), sorted(set(_lib.PROTOTYPES) - seen)

    def fields(struct):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), text, flags=re.S).group(1)
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(",")
            out.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", names[0])[-1])
            out += [re.findall(r"[A-Za-z_][A-Za-z0-9_]*", x)[-1] for x in names[1:]]
        return out

    assert fields("sv_config") == [f[0] for f in _lib.SvConfig._fields_]
    assert fields("sv_sampling") == [f[0] for f in _lib.SvSampling._fields_]
    assert fields("sv_beam_config") == [f[0] for f in _lib.SvBeamConfig._fields_]


def test_struct_layouts():
    from starvector_amd._lib import SvConfig, SvSampling
    assert C.sizeof(SvConfig) == 24 * 4
    assert SvSampling.stop_ids.offset % 8 == 0 and SvSampling.seed.offset % 8 == 0


def test_default_config_is_starvector_1b(lib):
    from starvector_amd._lib import SvConfig
    c = SvConfig()
    lib.sv_config_default_1b(C.byref(c))
    assert (c.image_size, c.patch_size, c.vit_width, c.vit_layers, c.vit_heads) == (224, 14, 1024, 23, 16)
    assert (c.hidden, c.n_layer, c.n_head, c.n_inner, c.vocab, c.n_positions) == (2048, 24, 16, 8192, 49156, 8192)
    assert (c.arch, c.n_kv_head, c.exclusive_device) == (0, 1, 0)
    lib.sv_config_default_8b(C.byref(c))                   # siglip_384 + starcoder2-7b
    assert (c.image_size, c.patch_size, c.vit_layers, c.hidden, c.n_layer, c.n_head, c.n_kv_head, c.n_inner, c.vocab) == \
        (384, 16, 24, 4608, 32, 36, 4, 18432, 49157)
    assert c.arch == 1 and abs(c.rope_theta - 1e6) < 1 and abs(c.vit_eps - 1e-6) < 1e-9 and c.sliding_window == 4096


def test_errors_are_reported_not_crashes(lib):
    from starvector_amd._lib import SvConfig
    h = C.c_void_p()
    assert lib.sv_create(None, C.byref(h)) == -22
    assert b"null" in lib.sv_last_error()
    c = SvConfig()
    lib.sv_config_default_1b(C.byref(c))
    c.vit_width = 1000                       # head_dim 62.5 -> rejected before touching the device
    assert lib.sv_create(C.byref(c), C.byref(h)) == -22
    assert lib.sv_op_layernorm(None, None, None, None, 1, 8, 1e-5, None) == -22
    assert lib.sv_destroy(None) == 0


def test_product_path_has_no_cpu_fallback():
    import torch
    import starvector_amd as sva
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(sva.StarVectorHipError):
        sva.HipEngine(sva.EngineConfig())
    # and the package never imports the oracle
    import sys
    src_dir = os.path.join(ROOT, "star-vector_amd")
    for fn in os.listdir(src_dir):
        if fn.endswith(".py"):
            assert "oracle" not in open(os.path.join(src_dir, fn)).read().replace("oracle/", "").lower() or fn == "never"


def test_argument_validation_precedes_any_device_work(lib):
    """Every entry point rejects bad arguments with SV_EINVAL and a message before it touches the GPU (so the same checks
    run here, without one); a valid configuration fails loudly with SV_EHIP -- there is no CPU path to fall into."""
    import torch
    from starvector_amd._lib import SvBeamConfig, SvConfig

    def err():
        return lib.sv_last_error().decode()

    n, h = C.c_int32(0), C.c_void_p()
    assert lib.sv_generate(None, None, 1, 4, None, None, C.byref(n), None) == -22 and "null engine" in err()
    assert lib.sv_prefill(None, None, 1, 4, None, None) == -22
    assert lib.sv_decode_step(None, None, 1, None, None) == -22
    assert lib.sv_encode_image(None, None, 1, None, None) == -22
    assert lib.sv_weights_complete(None) == -22
    assert lib.sv_load_weight(None, b"x", None, 0, 1, None, None) == -22
    ok3, bad3 = (C.c_float * 3)(0.5, 0.5, 0.5), (C.c_float * 3)(0.5, 0.0, 0.5)
    p = C.c_void_p(16)                                         # never dereferenced: the checks come first
    assert lib.sv_preprocess_image(None, 10, 10, 3, 224, 0, ok3, ok3, None, None) == -22
    assert lib.sv_preprocess_image(p, 10, 10, 3, 224, 7, ok3, ok3, p, None) == -22 and "recipe" in err()
    assert lib.sv_preprocess_image(p, 10, 10, 3, 224, 0, ok3, bad3, p, None) == -22 and "std" in err()
    assert lib.sv_preprocess_image(p, 10, 10, 2, 224, 0, ok3, ok3, p, None) == -22 and "channels" in err()
    assert lib.sv_preprocess_image(p, 0, 10, 3, 224, 0, ok3, ok3, p, None) == -22
    assert lib.sv_beam_create(C.byref(SvBeamConfig()), C.byref(h)) == -22 and "bad shape" in err()
    assert lib.sv_beam_create(C.byref(SvBeamConfig(batch=2000, num_beams=2, vocab=100, max_new=4)), C.byref(h)) == -22
    for field, val, msg in [("patch_size", 15, "patch_size"), ("n_head", 15, "heads"), ("weight_dtype", 9, "weight_dtype"),
                            ("sliding_window", 5, "sliding_window"), ("max_seq_len", 1, "max_seq_len"),
                            ("max_seq_len", 9000, "max_seq_len"), ("arch", 4, "arch"), ("max_batch", 0, "max_batch")]:
        c = SvConfig()
        lib.sv_config_default_1b(C.byref(c))
        setattr(c, field, val)
        assert lib.sv_create(C.byref(c), C.byref(h)) == -22 and msg in err(), (field, err())
    # the measurement / A-B surfaces validate before they touch the device as well
    assert lib.sv_debug_set_gemm_form(3) == -22 and lib.sv_debug_set_gemm_form(-2) == -22
    assert lib.sv_debug_set_gemm_form(1) == 0 and lib.sv_debug_set_gemm_form(-1) == 0
    assert lib.sv_debug_set_linear_seq_rows(-70000) == -22 and lib.sv_debug_set_linear_seq_rows(259) == 0 and lib.sv_debug_set_linear_seq_rows(0) == 0
    buf = (C.c_int64 * 16)()
    assert lib.sv_debug_gemm_trace(100, 256, 256, 0, 1, buf, 1) == -22            # M < one tile
    assert lib.sv_debug_gemm_trace(256, 256, 256, 0, 2, buf, 1) == -22 and "form" in err()
    assert lib.sv_debug_gemm_trace(512, 512, 256, 0, 1, buf, 1) == -22 and "capacity" in err()
    assert lib.sv_debug_xcc_map(None, 8, 0, None) == -22
    assert lib.sv_debug_occupy_cus(None, 1, 1024, 1) == -22                       # the safety tests' tenant: no engine, no launch
    if not torch.cuda.is_available():
        c = SvConfig()
        lib.sv_config_default_1b(C.byref(c))
        assert lib.sv_create(C.byref(c), C.byref(h)) == -5 and "hipSetDevice" in err()
