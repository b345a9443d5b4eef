#!/usr/bin/env python3
"""Decode step time, tokens/s and time to first token of StarVector-1B im2svg over the batch size (one engine per batch size, `exclusive_device` as in bench.py;
greedy, 256 new tokens, EOS disabled, synthetic images): what a caller with fewer than BASELINE config 2's 32 requests per GPU gets.
    python tools/batch_sweep.py [--batches 1 2 4 8 16 32 64] [--new-tokens 256]"""
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
ap.add_argument("--batches", type=int, nargs="*", default=[1, 2, 4, 8, 16, 32, 64])
ap.add_argument("--new-tokens", type=int, default=256)
ap.add_argument("--beams", type=int, default=1, help="num_beams (the reference's default decode: 2 with --sample)")
ap.add_argument("--sample", action="store_true")
ap.add_argument("--shared", action="store_true", help="exclusive_device = False (the library's default: no all-blocks-resident fused launches)")
ap.add_argument("--max-seq-len", type=int, default=0, help="engine max_seq_len (default: prompt + new tokens); the reference's eval configs generate up to max_length 8192")
a = ap.parse_args()
dev = torch.device("cuda", 0)
for B in a.batches:
    ec = sva.EngineConfig(max_batch=B * a.beams, max_seq_len=a.max_seq_len if a.max_seq_len > 0 else 259 + a.new_tokens)
    ec.exclusive_device = not a.shared
    eng = sva.HipEngine(ec)
    eng.load_random_weights(seed=1234)
    img = synthetic_images(torch, B, 224, seed=0).to(dev)
    prompt = torch.tensor([[7, 11]] * B, dtype=torch.long, device=dev)

    def request(n):
        emb = eng.prepare_inputs(eng.encode_image(img), prompt)
        if a.beams > 1:
            return eng.generate(emb, max_length=emb.shape[1] + n, eos_token_id=-1, pad_token_id=49152, num_beams=a.beams, do_sample=a.sample,
                                top_p=0.9 if a.sample else 1.0, seed=1)
        return eng.generate(emb, max_length=emb.shape[1] + n, eos_token_id=-1, pad_token_id=49152)

    request(a.new_tokens)                                   # warm-up: GEMM tuning, graph capture
    ttft = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter(); request(1); torch.cuda.synchronize()
        ttft.append((time.perf_counter() - t0) * 1e3)
    dec = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); request(a.new_tokens); torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        tm = eng.last_timing()
        dec.append((tm["decode_ms"] * 1e3 / max(tm["decode_steps"], 1), B * a.new_tokens / wall * 1e3))
    plan = eng.step_plan()
    print(json.dumps({"batch": B, "beams": a.beams, "sample": bool(a.sample), "decode_us_per_step": round(statistics.median(d[0] for d in dec), 1), "tokens_per_s": round(statistics.median(d[1] for d in dec), 1),
                      "tokens_per_s_per_sequence": round(statistics.median(d[1] for d in dec) / B, 1), "ttft_ms_p50": round(statistics.median(ttft), 2),
                      "launches_per_step": plan["graph_kernel_nodes"], "max_seq_len": ec.max_seq_len, "exclusive_device": bool(ec.exclusive_device), "new_tokens": a.new_tokens}), flush=True)
    eng.close()
    del eng
    torch.cuda.empty_cache()
