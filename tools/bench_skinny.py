#!/usr/bin/env python3
"""Micro-benchmark of the decode GEMM kernel by shape / prologue / epilogue (GB/s of weight bytes)."""
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
cases = [
    # name, M, N, K, splitk, ln, mode
    ("c_attn  slabs sk4", 32, 2304, 2048, 4, 0, 0),
    ("c_attn  slabs sk1", 32, 2304, 2048, 1, 0, 0),
    ("c_attn  rowmaj sk1", 32, 2304, 2048, 1, 0, 4),
    ("c_attn  LN rowmaj sk1", 32, 2304, 2048, 1, 1, 4),
    ("c_attn  LN rowmaj sk4(ticket)", 32, 2304, 2048, 4, 1, 4),
    ("c_proj  resid sk1", 32, 2048, 2048, 1, 0, 3),
    ("c_proj  resid sk4(ticket)", 32, 2048, 2048, 4, 0, 3),
    ("c_fc    slabs", 32, 8192, 2048, 1, 0, 0),
    ("c_fc    gelu", 32, 8192, 2048, 1, 0, 1),
    ("c_fc    LN gelu", 32, 8192, 2048, 1, 1, 1),
    ("c_proj2 slabs sk4", 32, 2048, 8192, 4, 0, 0),
    ("c_proj2 resid sk1", 32, 2048, 8192, 1, 0, 3),
    ("c_proj2 resid sk4(ticket)", 32, 2048, 8192, 4, 0, 3),
    ("lm_head f32", 32, 49156, 2048, 1, 0, 2),
    ("lm_head LN f32", 32, 49156, 2048, 1, 1, 2),
]
if len(sys.argv) > 1 and sys.argv[1] == "waves":        # in-process A/B of the block size (8 vs 16 waves)
    for name, M, N, K, sk, ln, mode in cases:
        res = []
        for rep in range(2):
            res = []
            for wv in ("8", "16"):
                os.environ["SV_SKINNY_WAVES"] = wv
                us = C.c_double(0)
                rc = lib.sv_bench_decode_linear(M, N, K, sk, ln, mode, 200, C.byref(us), st)
                res.append(us.value if rc == 0 else float("nan"))
        print(f"{name:32s} 8 waves {res[0]:7.2f} us   16 waves {res[1]:7.2f} us", flush=True)
    sys.exit(0)
for name, M, N, K, sk, ln, mode in cases:
    us = C.c_double(0)
    rc = lib.sv_bench_decode_linear(M, N, K, sk, ln, mode, 200, C.byref(us), st)
    if rc:
        print(name, "ERR", lib.sv_last_error().decode())
        continue
    mb = 2.0 * N * K / 1e6
    print(f"{name:32s} {us.value:8.2f} us/launch  {mb:7.1f} MB  {mb / us.value * 1e6 / 1e6:7.2f} TB/s" if False else
          f"{name:32s} {us.value:8.2f} us/launch  {mb:7.1f} MB  {mb / us.value / 1e6 * 1e6:9.1f} GB/s", flush=True)
