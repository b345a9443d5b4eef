#!/usr/bin/env python3
"""In-process A/B of the two-row-tiles-per-block decode GEMMs (SV_SKINNY_MT2, read per launch and part of the kept graph's
key) on BASELINE config 5's workload shape: StarVector-8B text2svg, batch 64, fp8 or bf16 decoder weights."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import starvector_amd as sva  # noqa: E402
from oracle import starvector_oracle as O  # noqa: E402  (weight factory only)

which = sys.argv[1:] or ["fp8", "bf16"]
cfg = O.OracleConfig.starvector_8b()
B, S0, NEW = 64, 33, 128
for wd in which:
    ec = sva.EngineConfig.starvector_8b(max_batch=B, max_seq_len=S0 + NEW)
    if wd == "fp8":
        ec.weight_dtype = "fp8_e4m3"
    t0 = time.time()
    eng = sva.HipEngine(ec)
    for name, t in O.iter_weights(cfg, seed=1234, init="std002"):
        if "image_encoder" in name or "image_projection" in name:
            pass
        eng.load_weight(name, t)
    eng.load_state_dict({})
    print(f"[{wd}] engine ready in {time.time() - t0:.0f} s", flush=True)
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(1, 49152, (B, S0), generator=g).cuda()
    emb = eng.embed_tokens(ids)
    kw = dict(max_length=S0 + NEW, eos_token_id=-1, pad_token_id=0, do_sample=True, temperature=1.0, top_p=0.95, top_k=50, seed=1)
    outs = {}
    for rep in range(2):
        for mode in ("1", "0"):
            os.environ["SV_SKINNY_MT2"] = mode
            toks = eng.generate(emb, **kw).cpu()
            tm = eng.last_timing()
            us = tm["decode_ms"] / max(tm["decode_steps"], 1) * 1e3
            outs[mode] = toks
            print(f"[{wd}] two-tile blocks {'on ' if mode == '1' else 'off'}: {us:8.1f} us/step  {B / us * 1e6:9.0f} tok/s", flush=True)
    os.environ.pop("SV_SKINNY_MT2", None)
    print(f"[{wd}] token streams identical on/off: {torch.equal(outs['1'], outs['0'])}", flush=True)
    prof = eng.profile_decode_step(B, iters=3)
    print(f"[{wd}] step profile (ms): " + ", ".join(f"{k} {v['ms_per_step']:.3f}" for k, v in prof.items() if isinstance(v, dict)), flush=True)
    eng.close()
    del eng
    torch.cuda.empty_cache()
