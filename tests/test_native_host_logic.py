"""CPU: host-side decisions that live inside libstarvector_hip.so, through its C ABI (no GPU involved).

  * the fixed-point resampling table of `sv_preprocess_image` against the oracle's restatement of Pillow's
    precompute_coeffs / normalize_coeffs_8bpc (which `oracle/image_preprocess.py::pin` holds to Pillow itself);
  * the big-M GEMM dispatch (tile kernel / row-remainder peeling) at the shapes whose A/B measurements are committed in
    profiles/gemm_r01_dispatch_ab.log."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle.image_preprocess import resample_coeffs
from starvector_amd import _lib


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


@pytest.mark.parametrize("in_size,out_size", [(517, 224), (224, 224), (100, 224), (1024, 384), (50, 224), (3000, 224),
                                              (225, 224), (223, 224), (384, 384), (640, 384), (7, 5), (1, 3), (16384, 224)])
def test_resample_table_is_pillows(lib, in_size, out_size):
    cap = int(math.ceil(2.0 * max(in_size / out_size, 1.0))) * 2 + 1           # Resample.c: ksize
    b = (C.c_int32 * (2 * out_size))()
    t = (C.c_int32 * (out_size * cap))()
    k = lib.sv_debug_resample_coeffs(in_size, out_size, b, t, cap)
    rb, rt = resample_coeffs(in_size, out_size)
    assert k == rt.shape[1] == cap
    assert np.array_equal(np.array(b).reshape(out_size, 2), rb)
    assert np.array_equal(np.array(t).reshape(out_size, cap), rt)              # bit for bit: integer taps
    assert lib.sv_debug_resample_coeffs(in_size, out_size, b, t, cap - 1) == -22   # table would not fit: SV_EINVAL
    if in_size != out_size:                                                    # every row of taps sums to ~1.0 in 22-bit fixed point
        sums = np.array(t).reshape(out_size, cap).sum(axis=1)
        assert np.abs(sums - (1 << 22)).max() <= cap


def _plan(lib, M, N, K, act=0):
    out = (C.c_int32 * 5)()
    assert lib.sv_debug_gemm_plan(M, N, K, act, out) == 0
    return dict(peel=out[0], tail_rows=out[1], tail_by_tiles=out[2], main_256=out[3], est_us=out[4])


def test_gemm_dispatch_at_the_measured_shapes(lib):
    # StarVector-1B prefill, batch 32: 32 x 259 = 8288 rows = 32 tiles of 256 + 96 (profiles/gemm_r01_dispatch_ab.log:
    # "auto" tracks the faster of peel / no-peel at every one of these shapes)
    assert _plan(lib, 8288, 2304, 2048)["peel"] == 0                         # c_attn: no-peel 102.6 us vs peel 112.4 us
    p = _plan(lib, 8288, 2048, 2048)                                         # c_proj: peel 100.1 vs 108.3
    assert (p["peel"], p["tail_rows"], p["main_256"]) == (1, 96, 1)
    assert _plan(lib, 8288, 8192, 2048, act=3)["peel"] == 0                  # c_fc (GELU): 374.0 vs 376.3
    p = _plan(lib, 8288, 2048, 8192)                                         # down-proj: peel 275.2 vs 353.5
    assert (p["peel"], p["tail_rows"], p["main_256"]) == (1, 96, 1)
    # ViT, batch 32: 32 x 257 = 8224 rows = 32 tiles + 32 rows
    assert _plan(lib, 8224, 3072, 1024)["peel"] == 0
    assert _plan(lib, 8224, 1024, 1024)["peel"] == 1 and _plan(lib, 8224, 1024, 4096)["peel"] == 1
    # nothing to peel: exact multiples, small problems (a peeled main part must still fill the chip), large remainders
    for M, N, K in [(8192, 3072, 1024), (1036, 2048, 2048), (259, 2048, 2048), (8192 + 128, 2048, 2048)]:
        p = _plan(lib, M, N, K)
        assert (p["peel"], p["tail_rows"], p["tail_by_tiles"]) == (0, 0, 0), (M, N, K)
    # the 256^2 kernel only when its grid fills the chip (>= 200 tiles): B = 4 prefill stays on 128^2
    assert _plan(lib, 4 * 259, 8192, 2048)["main_256"] == 0
    # the modelled time is within 15 % of the measured one at the prefill shapes
    for (M, N, K, act), us in {(8288, 2304, 2048, 0): 104.8, (8288, 2048, 2048, 0): 100.1, (8288, 8192, 2048, 3): 378.1,
                               (8288, 2048, 8192, 0): 274.4}.items():
        assert abs(_plan(lib, M, N, K, act)["est_us"] - us) / us < 0.15
    out = (C.c_int32 * 5)()
    assert lib.sv_debug_gemm_plan(0, 8, 8, 0, out) == -22


def test_per_sequence_remainder_form_is_a_function_of_sequence_length_and_projection(lib):
    """gemm.hip gemm_seq_form: the rows a sequence leaves over its 256-row tiles take the split-K remainder kernel where the cost model
    peels the remainder of a 32-sequence batch -- decided from (S, N, K, act) alone, so the kernel (and the summation order) a row gets
    never depends on the batch around it."""
    f = lib.sv_debug_gemm_seq_form
    # StarVector-1B prompt rows (259 = 256 + 3): attention output projection and down projection yes; c_attn / c_fc no (not peeled)
    assert f(259, 2048, 2048, 0) == 1 and f(259, 2048, 8192, 0) == 1
    assert f(259, 2304, 2048, 0) == 0 and f(259, 8192, 2048, 3) == 0
    # ViT tokens (257 = 256 + 1): out_proj and the MLP's second projection yes, in_proj no
    assert f(257, 1024, 1024, 0) == 1 and f(257, 1024, 4096, 0) == 1 and f(257, 3072, 1024, 0) == 0
    # no full tile, nothing left over, or too many rows left over: never
    for S in (3, 33, 250, 256, 512, 260, 578, 729):
        assert f(S, 2048, 2048, 0) == 0, S
    assert f(515, 2048, 2048, 0) == 1                                        # two tiles + 3 rows
    assert f(0, 8, 8, 0) == -22


def _skinny(lib, rows, N, K, sk, fp8=0):
    out = (C.c_int32 * 2)()
    assert lib.sv_debug_skinny_plan(rows, N, K, sk, fp8, out) == 0
    return out[0], out[1]


def test_decode_gemm_wave_split_depends_on_the_gemm_only(lib):
    """How K is cut inside a block decides the order a row's partial sums are added in: it must not change with the batch
    (round 2 fixed a dependence on the number of row tiles), while 33..64 rows switch to two row tiles per block."""
    shapes = [(2304, 2048, 4), (2048, 2048, 4), (8192, 2048, 1), (2048, 8192, 4), (49156, 2048, 1),          # StarVector-1B decode
              (5632, 4608, 2), (4608, 4608, 2), (18432, 4608, 1), (4608, 18432, 2), (49157, 4608, 1),        # StarVector-8B decode
              (516, 128, 1), (512, 256, 2), (1024, 1024, 2)]                                                   # tiny / narrow outputs
    for N, K, sk in shapes:
        for fp8 in (0, 1):
            if fp8 and ((K // 16) // sk) % 4:
                continue
            w1, two1 = _skinny(lib, 1, N, K, sk, fp8)
            for rows in (7, 32, 33, 64):
                w, two = _skinny(lib, rows, N, K, sk, fp8)
                assert w == w1, (N, K, sk, fp8, rows)                      # the batch never changes the summation order
                assert two == (1 if rows > 32 and w in (4, 8) else 0)
            assert two1 == 0
    assert _skinny(lib, 32, 2304, 2048, 4)[0] == 8 and _skinny(lib, 32, 8192, 2048, 1)[0] == 8      # the 1B decode GEMMs: 8 waves
    assert _skinny(lib, 32, 1024, 1024, 2)[0] == 16                                                   # narrow output: 16 waves, one tile
    out = (C.c_int32 * 2)()
    assert lib.sv_debug_skinny_plan(32, 64, 100, 1, 0, out) == -22                                    # K % 16


def _decode_plan(lib, rows, N, K, fp8=0, whole_k=0, cus=256):
    out = (C.c_int32 * 2)()
    assert lib.sv_debug_decode_plan(rows, N, K, fp8, whole_k, cus, out) == 0
    return out[0], out[1]


def test_decode_plan_split_k_and_column_tiles(lib):
    """sv_create's per-Linear decode plan on a 256-CU GPU (DESIGN.md sections 3c / 3d): split-K by CU fill, and at 33..64 rows the
    column tiles per block picked together with it.  <= 32 rows never share activation fragments across column tiles (one tile per
    block); whole-K Linears (c_fc, lm_head) never split."""
    # StarVector-1B (hidden 2048): small GEMMs keep the round 1-2 rule (smallest power of two reaching one block per CU)
    assert _decode_plan(lib, 32, 2304, 2048) == (4, 1) and _decode_plan(lib, 32, 2048, 2048) == (4, 1)
    assert _decode_plan(lib, 32, 2048, 8192) == (4, 1)
    assert _decode_plan(lib, 32, 8192, 2048, whole_k=1) == (1, 1) and _decode_plan(lib, 32, 49156, 2048, whole_k=1) == (1, 1)
    # StarVector-8B at 16 rows (config 4), bf16: fill rule
    assert _decode_plan(lib, 16, 5632, 4608) == (4, 1)
    assert _decode_plan(lib, 16, 4608, 4608) == (3, 1) and _decode_plan(lib, 16, 4608, 18432) == (3, 1)
    assert _decode_plan(lib, 16, 18432, 4608, whole_k=1) == (1, 1)
    # StarVector-8B at 64 rows, fp8 weights (config 5): (split-K, column tiles) together
    assert _decode_plan(lib, 64, 5632, 4608, fp8=1) == (3, 3)
    assert _decode_plan(lib, 64, 4608, 4608, fp8=1) == (3, 2)
    assert _decode_plan(lib, 64, 18432, 4608, fp8=1, whole_k=1) == (1, 3)            # 576 tiles -> 192 blocks: one round
    assert _decode_plan(lib, 64, 4608, 18432, fp8=1) == (4, 3)
    assert _decode_plan(lib, 64, 49157, 4608, fp8=1, whole_k=1) == (1, 3)
    # the same at bf16
    assert _decode_plan(lib, 64, 5632, 4608) == (4, 3) and _decode_plan(lib, 64, 4608, 18432) == (4, 3)
    # every plan is launchable: K splits evenly, fp8 keeps pairs of k-steps per wave
    for rows in (16, 32, 64):
        for fp8 in (0, 1):
            for N, K in ((5632, 4608), (4608, 4608), (4608, 18432), (2304, 2048), (2048, 8192), (512, 256)):
                sk, ct = _decode_plan(lib, rows, N, K, fp8=fp8)
                assert 1 <= sk <= 8 and (K // 16) % sk == 0 and ct in (1, 2, 3)
                assert ct == 1 or rows > 32
                if fp8:
                    assert ((K // 16) // sk) % 4 == 0
    out = (C.c_int32 * 2)()
    assert lib.sv_debug_decode_plan(65, 512, 256, 0, 0, 256, out) == -22


def test_decode_attention_split_constants(lib):
    """Context splits of the decode attention are constants of the engine (DESIGN.md sections 3 / 3d): enough blocks to cover the
    256 CUs, never more than 8 per sequence; where rows x KV heads cover the chip on their own a block takes 8 key groups."""
    def plan(max_batch, nkv, cus=256):
        out = (C.c_int32 * 2)()
        assert lib.sv_debug_attn_plan(max_batch, nkv, cus, out) == 0
        return out[0], out[1]
    assert plan(32, 1) == (8, 4)          # StarVector-1B, batch 32 (multi-query): 32 rows x 8 splits = 256 blocks
    assert plan(1, 1) == (8, 4) and plan(4, 1) == (8, 4)
    assert plan(64, 1) == (8, 4)          # the split cap is sized for 32 rows: the same grouping as the 32-row engine
    assert plan(16, 4) == (4, 4)          # StarVector-8B, batch 16: 64 (row, KV head) pairs x 4
    assert plan(64, 4) == (2, 8)          # StarVector-8B, batch 64: 256 pairs cover the chip -> contexts <= 256 tokens stay in one block
    assert plan(32, 4) == (2, 4)
    out = (C.c_int32 * 2)()
    assert lib.sv_debug_attn_plan(0, 1, 256, out) == -22


def test_fused_row_update_c_attn_launch_shapes(lib):
    """Where an engine that owns its GPU runs the row update and the c_attn projection as one launch (host arithmetic of
    rowops.hip rowln_cattn_fits): a GEMM wave must hold its whole weight share in registers -- 4 k-steps (StarVector-1B: 2048 / 16 / 4 slices / 8
    waves) or 9 (StarVector-8B: 4608 / 16 / 4 / 8) -- narrow rows need all blocks resident at once (two per CU), the XCD-aware (tile, slice)
    assignment needs its divisibilities, the row update in front sums at most four slabs."""
    def plan(D, N, splitk, splitk_ru=4, cus=256):
        out = (C.c_int32 * 3)()
        assert lib.sv_debug_rowln_plan(D, N, splitk, splitk_ru, cus, out) == 0
        return out[0], out[1], out[2]
    assert plan(2048, 2304, 4) == (1, 4, 32 + 72 * 4)           # StarVector-1B: 320 blocks of 512 threads on 512 slots
    assert plan(4608, 5632, 4, splitk_ru=3) == (1, 9, 32 + 176 * 4)   # StarVector-8B: 736 blocks, three per CU
    assert plan(2048, 2304, 4, cus=128)[0] == 0                # 320 blocks do not fit 256 slots: the two launches
    assert plan(2048, 2304, 2)[0] == 0 and plan(2048, 2304, 8)[0] == 0      # 8 / 2 k-steps per wave: not an instantiation
    assert plan(4608, 5632, 3)[0] == 0                         # 8 % 3: no XCD-aware assignment
    assert plan(2048, 2304, 4, splitk_ru=5)[0] == 0            # the row role sums at most four slabs
    assert plan(1024, 1152, 2)[0] == 1 and plan(1024, 1152, 2)[1] == 4      # another narrow width with 4 k-steps per wave
    out = (C.c_int32 * 3)()
    assert lib.sv_debug_rowln_plan(0, 2304, 4, 4, 256, out) == -22
