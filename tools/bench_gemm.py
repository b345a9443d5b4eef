#!/usr/bin/env python3
"""Micro-benchmark of the big-M MFMA GEMM at the ViT / prefill shapes (TFLOP/s, pseudo-random operands)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from starvector_amd import _lib  # noqa: E402

lib = _lib.load()
torch.zeros(1, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
cases = [("prefill c_attn", 8288, 2304, 2048, 0, 0), ("prefill c_proj+res", 8288, 2048, 2048, 0, 1),
         ("prefill c_fc gelu", 8288, 8192, 2048, 3, 0), ("prefill down+res", 8288, 2048, 8192, 0, 1),
         ("vit in_proj", 8224, 3072, 1024, 0, 0), ("vit out_proj+res", 8224, 1024, 1024, 0, 1),
         ("vit c_fc qgelu", 8224, 4096, 1024, 1, 0), ("vit c_proj+res", 8224, 1024, 4096, 0, 1),
         ("square 4096", 4096, 4096, 4096, 0, 0), ("square 8192", 8192, 8192, 8192, 0, 0)]
tot = 0.0
for name, M, N, K, act, res in cases:
    us = C.c_double(0)
    rc = lib.sv_bench_linear(M, N, K, act, res, 20, C.byref(us), st)
    if rc:
        print(name, "ERR", lib.sv_last_error().decode()); continue
    tf = 2.0 * M * N * K / us.value / 1e6
    print(f"{name:22s} M{M} N{N} K{K}: {us.value:9.1f} us  {tf:7.1f} TFLOP/s  ({tf / 2500 * 100:4.1f} % of 2.5 PF)", flush=True)
