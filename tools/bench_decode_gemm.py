#!/usr/bin/env python3
"""Micro-benchmark of the full-K decode GEMMs (csrc/decode_gemm.hip) next to round 1's split-K slab kernel, by shape.
Weight bytes / time per launch; `iters` back-to-back launches between one HIP event pair (random weights)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from starvector_amd import _lib  # noqa: E402

lib = _lib.load()
torch.cuda.init()
torch.zeros(1, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
IT = 200


def new(M, N, K, kind, cpb=0):
    us = C.c_double(0)
    rc = lib.sv_bench_decode_gemm(M, N, K, kind, cpb, IT, C.byref(us), st)
    return us.value if rc == 0 else float("nan")


def old(M, N, K, sk, ln, mode):
    us = C.c_double(0)
    rc = lib.sv_bench_decode_linear(M, N, K, sk, ln, mode, IT, C.byref(us), st)
    return us.value if rc == 0 else float("nan")


def row(name, us, N, K):
    mb = 2.0 * N * K / 1e6
    print(f"{name:44s} {us:8.2f} us  {mb:7.1f} MB  {mb / us / 1e6 * 1e6:8.1f} GB/s", flush=True)


batches = [32] if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]
for M in batches:
    print(f"---- M = {M} rows ----")
    for tag, D, F, QKV, V in (("1b", 2048, 8192, 2304, 49156), ("8b", 4608, 18432, 5632, 49157)):
        row(f"{tag} c_attn  old slabs sk4", old(M, QKV, D, 4, 0, 0), QKV, D)
        for w in ("16", "8"):
            os.environ["SV_COLS_WAVES"] = w
        os.environ.pop("SV_COLS_WAVES", None)
        row(f"{tag} c_attn  cols LN prologue", new(M, QKV, D, 1), QKV, D)
        row(f"{tag} c_attn  cols (no LN)", new(M, QKV, D, 0), QKV, D)
        row(f"{tag} c_proj  old slabs sk4", old(M, D, D, 4, 0, 0), D, D)
        row(f"{tag} c_proj  cols +residual", new(M, D, D, 2), D, D)
        row(f"{tag} c_proj  cols +residual cpb16", new(M, D, D, 2, 16), D, D)
        row(f"{tag} c_fc    old gelu", old(M, F, D, 1, 0, 1), F, D)
        row(f"{tag} c_fc    skinny_ln gelu", new(M, F, D, 3), F, D)
        row(f"{tag} c_proj2 old slabs sk4", old(M, D, F, 4, 0, 0), D, F)
        row(f"{tag} c_proj2 cols +residual", new(M, D, F, 2), D, F)
        row(f"{tag} c_proj2 cols +residual cpb16", new(M, D, F, 2, 16), D, F)
        row(f"{tag} lm_head old f32", old(M, V, D, 1, 0, 2), V, D)
        row(f"{tag} lm_head skinny_ln f32", new(M, V, D, 4), V, D)
