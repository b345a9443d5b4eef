#!/usr/bin/env python3
"""Evidence for the prefill-GEMM ceiling claim: the big-M GEMM dispatch on uniform-random vs zero-filled operands,
with the clock level / socket power rocm-smi reports WHILE the kernel loops (a background sampler), for the four decoder GEMMs of a
prefill layer at M = 32 x 259 and for 8192^3.  Same binary, same shapes: the TF/s gap between the fills is DVFS, not the kernel."""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from starvector_amd import engine as E  # noqa: E402

torch.zeros(1, device="cuda")


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.sclk, self.power = [], []

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
                w = re.search(r"Power \(W\): ([0-9.]+)", out)
                if m:
                    self.sclk.append(int(m.group(1)))
                if w:
                    self.power.append(float(w.group(1)))
            except Exception:
                pass


shapes = [("c_attn", 8288, 2304, 2048, "none", False), ("c_proj", 8288, 2048, 2048, "none", True), ("c_fc", 8288, 8192, 2048, "gelu_tanh", False),
          ("down_proj", 8288, 2048, 8192, "none", True), ("8192^3", 8192, 8192, 8192, "none", False)]
for fill in ("random", "zero"):
    if fill == "zero":
        os.environ["SV_BENCH_FILL"] = "zero"
    else:
        os.environ.pop("SV_BENCH_FILL", None)
    for name, M, N, K, act, res in shapes:
        E.bench_linear(M, N, K, act=act, residual=res, iters=5)                      # tune + warm
        s = Sampler()
        s.start()
        t0 = time.time()
        us = []
        while time.time() - t0 < 1.5:                                                 # ~1.5 s of back-to-back launches under the sampler
            us.append(E.bench_linear(M, N, K, act=act, residual=res, iters=200 if K < 8192 or N < 8192 else 20))
        s.stop = True
        s.join()
        u = sorted(us)[len(us) // 2]
        print(json.dumps({"gemm": name, "shape": [M, N, K], "fill": fill, "us": round(u, 1), "tflops": round(2.0 * M * N * K / u / 1e6, 1),
                          "sclk_mhz_samples": sorted(set(s.sclk)), "sclk_mhz_median": (sorted(s.sclk)[len(s.sclk) // 2] if s.sclk else None),
                          "power_w_median": (sorted(s.power)[len(s.power) // 2] if s.power else None), "n_samples": len(s.sclk)}), flush=True)
