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
from oracle import starvector_oracle as O  # noqa: E402

dev = torch.device("cuda", 0)
cfg = O.OracleConfig()
w = O.make_weights(cfg, seed=1, init="std002")
B = 32
eng = sva.HipEngine(sva.EngineConfig(max_batch=B, max_seq_len=259 + 2100))
eng.load_state_dict(w)
del w
img = O.synthetic_images(B, 224, seed=2).to(torch.bfloat16).to(dev)
prompt = torch.tensor([[7, 11]] * B, dtype=torch.long, device=dev)
emb = torch.cat([eng.adapter(eng.encode_image(img)), eng.embed_tokens(prompt)], 1)
S0 = emb.shape[1]
for n_new in (2, 128, 512, 1024, 2048):
    eng.generate(emb, max_length=S0 + n_new, eos_token_id=-1, pad_token_id=cfg.pad_token_id)
    tm = eng.last_timing()
    prof = eng.profile_decode_step(B, iters=5)
    print(json.dumps({"ctx_end": S0 + n_new, "avg_us_per_step": round(tm["decode_ms"] / max(tm["decode_steps"], 1) * 1e3, 1),
                      "ttft_ms": round(tm["ttft_ms"], 2),
                      "at_ctx_end_ms": {k: round(v["ms_per_step"], 4) for k, v in prof.items() if isinstance(v, dict)}}), flush=True)
