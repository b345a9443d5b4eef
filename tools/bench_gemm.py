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
         ("square 4096", 4096, 4096, 4096, 0, 0), ("square 8192", 8192, 8192, 8192, 0, 0),
         ("main c_attn", 8192, 2304, 2048, 0, 0), ("main c_proj+res", 8192, 2048, 2048, 0, 1),
         ("main c_fc gelu", 8192, 8192, 2048, 3, 0), ("main down+res", 8192, 2048, 8192, 0, 1),
         ("main vit in_proj", 8192, 3072, 1024, 0, 0), ("main vit out_proj", 8192, 1024, 1024, 0, 1),
         ("main vit c_fc", 8192, 4096, 1024, 1, 0), ("main vit c_proj", 8192, 1024, 4096, 0, 1)]
if len(sys.argv) > 1:
    cases = [c for c in cases if any(k in c[0] for k in sys.argv[1:])]
modes = [("auto", {}), ("no-peel", {"SV_GEMM_TAIL": "0"}), ("peel", {"SV_GEMM_TAIL": "2"}),
         ("128 no-peel", {"SV_GEMM_TAIL": "0", "SV_GEMM_VARIANT": "0"})]
if os.environ.get("BENCH_GEMM_SINGLE"):
    modes = [("env", {})]
for name, M, N, K, act, res in cases:
    line = f"{name:20s} M{M} N{N} K{K}:"
    for rep in range(2):                      # two interleaved passes: the second one is the one to read
        vals = []
        for mname, env in modes:
            for k in ("SV_GEMM_TAIL", "SV_GEMM_VARIANT"):
                if k in env:
                    os.environ[k] = env[k]
                elif not os.environ.get("BENCH_GEMM_SINGLE"):
                    os.environ.pop(k, None)
            us = C.c_double(0)
            rc = lib.sv_bench_linear(M, N, K, act, res, 20, C.byref(us), st)
            vals.append((mname, us.value if rc == 0 else float("nan")))
    print(line + "  ".join(f"{m} {u:7.1f} us ({2.0 * M * N * K / u / 1e6:6.1f} TF)" for m, u in vals), flush=True)
