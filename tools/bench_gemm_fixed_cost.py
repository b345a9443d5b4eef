#!/usr/bin/env python3
"""Where does a big-M GEMM tile's time go at short K?  K sweep at M = N = 8192 (1024 tiles of 256^2 = exactly 4 rounds of the
chip; 4096 tiles of 128^2 = 8 rounds at 2 blocks per CU): time = rounds x (fixed + slope x K).  The 256^2 kernel is also run
with its epilogue ablated (SV_GEMM_EPI=none: accumulators dropped; ldsonly: park + read back, no global loads / stores) --
timing only, those runs compute nothing useful."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from starvector_amd import _lib  # noqa: E402

lib = _lib.load()
torch.zeros(1, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = N = 8192


def run(K, act, res, variant, epi):
    os.environ["SV_GEMM_VARIANT"] = variant
    os.environ["SV_GEMM_EPI"] = epi
    best = 1e30
    for _ in range(2):
        us = C.c_double(0)
        rc = lib.sv_bench_linear(M, N, K, act, res, 20, C.byref(us), st)
        if rc == 0:
            best = min(best, us.value)
    return best


Ks = (64, 256, 1024, 2048, 4096)
for name, act, res in (("plain", 0, 0), ("gelu", 3, 0), ("residual", 0, 1)):
    for variant, epi in (("2", "lds"), ("2", "ldsonly"), ("2", "none"), ("2", "regs"), ("0", "lds")):
        ts = [run(K, act, res, variant, epi) for K in Ks]
        slope = (ts[-1] - ts[-2]) / (Ks[-1] - Ks[-2])                     # us per unit of K, whole launch
        fixed = ts[3] - slope * Ks[3]
        rounds = 4 if variant == "2" else 8
        print(f"{name:9s} {'256^2' if variant == '2' else '128^2'} epi {epi:8s}: " + "  ".join(f"K{K} {t:7.1f}" for K, t in zip(Ks, ts)) +
              f"  | per round: fixed {fixed / rounds:5.1f} us + {slope / rounds * 2048:5.1f} us per K=2048", flush=True)
