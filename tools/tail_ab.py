#!/usr/bin/env python3
"""Remainder-row launches of the big-M GEMMs at the prefill / ViT shapes: the tuned choice against 256^2 tiles + the peeled rows through
gemm_tail_kernel (form 2) and against 256^2 tiles over all rows (form 1); microseconds per GEMM (sv_bench_linear: random operands, HIP
events), and -- round 6 -- against the per-sequence remainder (sv_debug_set_linear_seq_rows: tiles over the full 256-row tiles of every
sequence + gemm_tailk_kernel).  (Round 5 ran it with a third form, the four-wave tail kernel: profiles/gemm_tail4_r05_ab.log.)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from starvector_amd import engine as E  # noqa: E402

shapes = [("dec c_proj", 8288, 2048, 2048, "none", True), ("dec c_fc", 8288, 8192, 2048, "gelu_tanh", False),
          ("dec down", 8288, 2048, 8192, "none", True), ("dec c_attn", 8288, 2304, 2048, "none", False),
          ("vit qkv", 8224, 3072, 1024, "none", False), ("vit out", 8224, 1024, 1024, "none", True),
          ("vit fc1", 8224, 4096, 1024, "quickgelu", False), ("vit fc2", 8224, 1024, 4096, "none", True)]
seq = {8288: 259, 8224: 257}
print(f"{'shape':12s} {'M':>5s} {'N':>5s} {'K':>5s}   tuned   form2(256^2 + tail)  form1(256^2 whole)   per-sequence remainder (tuned tiles + gemm_tailk_kernel)   [us]")
for name, M, N, K, act, res in shapes:
    row = []
    for form in (-1, 2, 1):
        E.set_gemm_form(form)
        E.bench_linear(M, N, K, act=act, residual=res, iters=3)
        row.append(min(E.bench_linear(M, N, K, act=act, residual=res, iters=20) for _ in range(3)))
    E.set_gemm_form(-1)
    E.set_linear_seq_rows(seq[M])
    E.bench_linear(M, N, K, act=act, residual=res, iters=3)
    row.append(min(E.bench_linear(M, N, K, act=act, residual=res, iters=20) for _ in range(3)))
    E.set_linear_seq_rows(0)
    print(f"{name:12s} {M:5d} {N:5d} {K:5d}  {row[0]:7.1f}  {row[1]:17.1f}  {row[2]:18.1f}  {row[3]:18.1f}")
