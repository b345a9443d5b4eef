#!/usr/bin/env python3
"""Continuous batching (sv_cb_*): microseconds per decode step as the serve scheduler drives it -- `live` requests admitted into an engine of `max_batch` slots, then
`cb_step(n_steps)` call after call (one host round trip per call).    python tools/cb_step_bench.py [--live 4] [--max-batch 8] [--n-steps 8]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import starvector_amd as sva  # noqa: E402
from bench import synthetic_images  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--live", type=int, default=4)
ap.add_argument("--max-batch", type=int, default=8)
ap.add_argument("--n-steps", type=int, default=8)
ap.add_argument("--calls", type=int, default=40)
a = ap.parse_args()
dev = torch.device("cuda", 0)
budget = a.n_steps * (a.calls + 8) + 8
ec = sva.EngineConfig(max_batch=a.max_batch, max_seq_len=259 + budget + 8)
ec.exclusive_device = True
eng = sva.HipEngine(ec)
eng.load_random_weights(seed=1234)
img = synthetic_images(torch, a.live, 224, seed=0).to(dev)
prompt = torch.tensor([[7, 11]] * a.live, dtype=torch.long, device=dev)
emb = eng.prepare_inputs(eng.encode_image(img), prompt)
slots = eng.cb_admit(emb, [dict(max_new_tokens=budget, eos_token_id=-1, pad_token_id=49152) for _ in range(a.live)])
for _ in range(4):
    eng.cb_step(a.n_steps)                                   # graph capture, warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.calls):
    live = eng.cb_step(a.n_steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"live": a.live, "max_batch": a.max_batch, "n_steps_per_call": a.n_steps, "us_per_step": round(dt / (a.calls * a.n_steps) * 1e6, 1),
                  "steps_per_graph_launch": eng.last_timing()["graph_steps"], "still_live": live}))
eng.cb_reset()
eng.close()
