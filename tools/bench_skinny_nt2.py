#!/usr/bin/env python3
"""In-process A/B of the two-column-tiles-per-wave decode GEMM (SV_SKINNY_NT2) and of the split-K factor it needs on the
narrow outputs (StarVector-1B decode shapes, 32 rows)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from starvector_amd import _lib  # noqa: E402

lib = _lib.load()
torch.zeros(1, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
cases = [("lm_head f32 sk1", 32, 49156, 2048, 1, 2), ("down slabs sk4", 32, 2048, 8192, 4, 0), ("down slabs sk8", 32, 2048, 8192, 8, 0),
         ("c_attn slabs sk4", 32, 2304, 2048, 4, 0), ("c_attn slabs sk8", 32, 2304, 2048, 8, 0),
         ("c_proj slabs sk4", 32, 2048, 2048, 4, 0), ("c_proj slabs sk8", 32, 2048, 2048, 8, 0), ("c_fc slabs sk1", 32, 8192, 2048, 1, 0),
         ("c_fc slabs sk2", 32, 8192, 2048, 2, 0), ("8b c_fc slabs sk1", 32, 18432, 4608, 1, 0), ("8b down sk2", 32, 4608, 18432, 2, 0),
         ("8b down sk4", 32, 4608, 18432, 4, 0), ("8b lm_head", 32, 49157, 4608, 1, 2)]
for name, M, N, K, sk, mode in cases:
    out = []
    for rep in range(2):
        for nt2 in ("1", "0"):
            os.environ["SV_SKINNY_NT2"] = nt2            # "1": pairs (the experiment), "0": one tile per wave (default)
            us = C.c_double(0)
            rc = lib.sv_bench_decode_linear(M, N, K, sk, 0, mode, 300, C.byref(us), st)
            out.append(f"{'pairs' if nt2 == '1' else 'single'} {us.value if rc == 0 else float('nan'):6.2f}")
    print(f"{name:22s} " + " | ".join(out), flush=True)
