#!/usr/bin/env python3
"""In-process A/B of the experiment masks (SV_EXP bits) on BASELINE config 2's time to first token:
    python tools/ttft_ab.py [--reps 15] [--batch 32] 0 32768 ...
One engine, the masks interleaved; a request = encoder + adapter + prompt pass + first token (bench.py's `step(max_new=1)`), host wall
clock around a synchronised call.  Prints the median / min per mask and whether the first tokens equal mask 0's."""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import starvector_amd as sva  # noqa: E402
from bench import synthetic_images  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("masks", nargs="*", type=int, default=[0, 32768])
ap.add_argument("--reps", type=int, default=15)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--stages", action="store_true", help="also print sv_profile_ttft's stage table per mask")
a = ap.parse_args()
dev = torch.device("cuda", 0)
B = a.batch
ec = sva.EngineConfig(max_batch=B, max_seq_len=259 + 16)
ec.exclusive_device = True
eng = sva.HipEngine(ec)
eng.load_random_weights(seed=1234)
img = synthetic_images(torch, B, 224, seed=0).to(dev)
prompt = torch.tensor([[7, 11]] * B, dtype=torch.long, device=dev)


def request():
    emb = eng.prepare_inputs(eng.encode_image(img), prompt)
    return eng.generate(emb, max_length=emb.shape[1] + 1, eos_token_id=-1, pad_token_id=49152)


ms = {m: [] for m in a.masks}
first = {}
for rep in range(a.reps + 2):                      # reps 0-1 = warm-up (GEMM tuning, allocations)
    for m in a.masks:
        eng.set_exp(m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tok = request()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        if rep >= 2:
            ms[m].append(dt)
        first.setdefault(m, tok.cpu())
for m in a.masks:
    line = {"exp": m, "ttft_ms_median": round(statistics.median(ms[m]), 3), "ttft_ms_min": round(min(ms[m]), 3), "n": len(ms[m]),
            "first_tokens_equal_first_mask": bool(torch.equal(first[m], first[a.masks[0]]))}
    if a.stages:
        eng.set_exp(m)
        tp = eng.profile_ttft(img, prompt, iters=3)
        line["stages_ms"] = {k: round(tp[k], 3) for k in eng.TTFT_STAGES}
        line["launches"] = tp["launches"]
    print(json.dumps(line), flush=True)
eng.close()
