#!/usr/bin/env python3
"""In-process A/B of the experiment masks (SV_EXP bits) on BASELINE config 2's decode loop:
    python tools/ab_exp.py [--new-tokens 512] [--reps 2] 0 1 2 3 ...
One engine, the masks interleaved `reps` times; prints us per decode step, the per-class HIP-event profile and whether the
token stream equals mask 0's."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import starvector_amd as sva  # noqa: E402
from bench import synthetic_images  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("masks", nargs="*", type=int, default=[0, 1, 2, 3])
ap.add_argument("--new-tokens", type=int, default=512)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--shared", action="store_true", help="engine WITHOUT exclusive_device (none of the all-blocks-resident fused launches)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
B = a.batch
ec = sva.EngineConfig(max_batch=B, max_seq_len=259 + a.new_tokens + 8)
ec.exclusive_device = not a.shared          # bench.py's engine: one process per GPU
eng = sva.HipEngine(ec)
eng.load_random_weights(seed=1234)
img = synthetic_images(torch, B, 224, seed=0).to(dev)
prompt = torch.tensor([[7, 11]] * B, dtype=torch.long, device=dev)
emb = torch.cat([eng.adapter(eng.encode_image(img)), eng.embed_tokens(prompt)], 1)
S0 = emb.shape[1]
kw = dict(max_length=S0 + a.new_tokens, eos_token_id=-1, pad_token_id=49152)
ref = None
for rep in range(a.reps + 1):                      # rep 0 = warm-up (graph capture, tuning)
    for m in a.masks:
        eng.set_exp(m)
        toks = eng.generate(emb, **kw).cpu()
        tm = eng.last_timing()
        if m == a.masks[0] and ref is None:
            ref = toks
        if rep == 0:
            continue
        prof = eng.profile_decode_step(B, iters=3)
        print(json.dumps({"exp": m, "rep": rep, "us_per_step": round(tm["decode_ms"] / max(tm["decode_steps"], 1) * 1e3, 1),
                          "tokens_equal_first_mask": bool(torch.equal(toks, ref)),
                          "event_ms": {k: round(v["ms_per_step"], 4) for k, v in prof.items() if isinstance(v, dict)},
                          "gemm_chain_ms": round(prof["skinny_chain_ms_per_step"], 4),
                          "others_chain_ms": round(prof["others_chain_ms_per_step"], 4)}), flush=True)
eng.close()
