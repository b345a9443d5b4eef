#!/usr/bin/env python3
"""In-process A/B of the 128^2 GEMM kernel's tile order: launch order vs XCD-aware bands (SV_GEMM_ORDER), prefill / ViT shapes."""
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
         ("patch embed", 8192, 1024, 640, 0, 0), ("square 8192", 8192, 8192, 8192, 0, 0)]
os.environ["SV_GEMM_VARIANT"] = "0"
for name, M, N, K, act, res in cases:
    out = []
    for order in ("plain", "xcd", "plain", "xcd"):
        if order == "plain":
            os.environ["SV_GEMM_ORDER"] = "plain"
        else:
            os.environ.pop("SV_GEMM_ORDER", None)
        us = C.c_double(0)
        rc = lib.sv_bench_linear(M, N, K, act, res, 20, C.byref(us), st)
        out.append(f"{order} {us.value if rc == 0 else float('nan'):7.1f}")
    print(f"{name:20s} M{M} N{N} K{K}  128^2: " + " | ".join(out) + f"  ({2.0 * M * N * K / float(out[-1].split()[-1]) / 1e6:6.1f} TF xcd)", flush=True)
