#!/usr/bin/env python3
"""Micro-benchmark of the decode GEMM kernel by shape / epilogue (TB/s of weight bytes).  Back-to-back launches of ONE GEMM:
anything below ~200 MB is served by the Infinity Cache / L2 after the first launch -- an upper bound, not the in-situ time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from starvector_amd import engine as E  # noqa: E402

torch.zeros(1, device="cuda")
cases = [
    # name, M, N, K, splitk, mode (0 slabs, 1 bias + GELU, 2 fp32 logits)
    ("1b c_attn  slabs sk4", 32, 2304, 2048, 4, 0), ("1b c_proj  slabs sk4", 32, 2048, 2048, 4, 0),
    ("1b c_fc    gelu", 32, 8192, 2048, 1, 1), ("1b c_proj2 slabs sk4", 32, 2048, 8192, 4, 0),
    ("1b lm_head f32", 32, 49156, 2048, 1, 2),
    ("8b c_attn  slabs", 16, 5632, 4608, 2, 0), ("8b c_fc    gelu", 16, 18432, 4608, 1, 1),
    ("8b c_proj2 slabs sk2", 16, 4608, 18432, 2, 0), ("8b c_fc    gelu  64 rows", 64, 18432, 4608, 1, 1),
]
for name, M, N, K, sk, mode in cases:
    us = E.bench_decode_linear(M, N, K, sk, mode, 200)
    mb = 2.0 * N * K / 1e6
    print(f"{name:28s} {us:8.2f} us/launch  {mb:7.1f} MB  {mb / us:6.2f} TB/s", flush=True)
