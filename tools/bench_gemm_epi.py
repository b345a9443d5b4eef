#!/usr/bin/env python3
"""In-process A/B of the big-M GEMM epilogue (register row-per-lane stores vs LDS-transposed full-line stores), both tile
kernels, prefill / ViT shapes.  TFLOP/s on pseudo-random operands."""
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
         ("main c_fc gelu", 8192, 8192, 2048, 3, 0), ("square 8192", 8192, 8192, 8192, 0, 0),
         ("8b c_fc gelu", 9248, 18432, 4608, 3, 0), ("8b down+res", 9248, 4608, 18432, 0, 1)]
for name, M, N, K, act, res in cases:
    out = []
    for variant in ("-1", "0", "2"):          # dispatcher / 128^2 / 256^2
        for rep in range(2):
            vals = []
            for epi in ("regs", "lds"):
                os.environ["SV_GEMM_EPI"] = epi
                if variant == "-1":
                    os.environ.pop("SV_GEMM_VARIANT", None)
                else:
                    os.environ["SV_GEMM_VARIANT"] = variant
                us = C.c_double(0)
                rc = lib.sv_bench_linear(M, N, K, act, res, 20, C.byref(us), st)
                vals.append(us.value if rc == 0 else float("nan"))
        out.append(f"{ {'-1': 'auto', '0': '128', '2': '256'}[variant]}: regs {vals[0]:7.1f} lds {vals[1]:7.1f} us ({2.0 * M * N * K / vals[1] / 1e6:6.1f} TF)")
    print(f"{name:20s} M{M} N{N} K{K}  " + " | ".join(out), flush=True)
