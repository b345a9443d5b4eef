#!/usr/bin/env python3
"""Decode-step time vs context length (StarVector-1B shapes, B=32): generate N tokens, then profile one
step by kernel class at the context the generation ended on."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import starvector_amd as sva  # noqa: E402
from bench import synthetic_images  # noqa: E402

dev = torch.device("cuda", 0)
B = 32
eng = sva.HipEngine(sva.EngineConfig(max_batch=B, max_seq_len=259 + 4100))
eng.load_random_weights(seed=1)
img = synthetic_images(torch, B, 224, seed=2).to(dev)
prompt = torch.tensor([[7, 11]] * B, dtype=torch.long, device=dev)
emb = torch.cat([eng.adapter(eng.encode_image(img)), eng.embed_tokens(prompt)], 1)
S0 = emb.shape[1]
for n_new in (2, 512, 1024, 2048, 4096):
    eng.generate(emb, max_length=S0 + n_new, eos_token_id=-1, pad_token_id=49152)          # first call of a length: graph capture
    eng.generate(emb, max_length=S0 + n_new, eos_token_id=-1, pad_token_id=49152)
    tm = eng.last_timing()
    # per-class HIP-event time at the context the generation ended on: the MEDIAN of three profile calls (one call's event deltas came out
    # wrong once, cause not established: profiles/ctx_sweep_r03.log's 1283 row)
    profs = [eng.profile_decode_step(B, iters=5) for _ in range(3)]
    med = {k: sorted(p[k]["ms_per_step"] for p in profs)[1] for k in profs[0] if isinstance(profs[0][k], dict)}
    print(json.dumps({"ctx_end": S0 + n_new, "avg_us_per_step": round(tm["decode_ms"] / max(tm["decode_steps"], 1) * 1e3, 1),
                      "ttft_ms": round(tm["ttft_ms"], 2), "at_ctx_end_ms": {k: round(v, 4) for k, v in med.items()},
                      "others_chain_ms": round(sorted(p["others_chain_ms_per_step"] for p in profs)[1], 4),
                      "gemm_chain_ms": round(sorted(p["skinny_chain_ms_per_step"] for p in profs)[1], 4)}), flush=True)
