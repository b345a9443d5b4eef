#!/usr/bin/env python3
"""Bisect a NaN in the 64-row StarVector-8B text2svg decode (fp8 / bf16 weights): teacher-forced greedy steps, first step with a non-finite logit."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import starvector_amd as sva
dev = torch.device("cuda", 0)
wdt = sys.argv[1] if len(sys.argv) > 1 else "fp8_e4m3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ec = sva.EngineConfig.starvector_8b(max_batch=B, max_seq_len=33 + 320)
ec.weight_dtype = wdt
ec.n_layer = int(sys.argv[3]) if len(sys.argv) > 3 else 32
eng = sva.HipEngine(ec)
eng.load_random_weights(seed=1234)
g = torch.Generator().manual_seed(0)
caps = torch.randint(1, 49152, (B, 32), generator=g)
prompt = torch.cat([caps, torch.full((B, 1), 49153, dtype=torch.long)], 1).to(dev)
emb = eng.embed_tokens(prompt)
lg = eng.prefill(emb)
print("prefill finite:", bool(torch.isfinite(lg).all()), "max", float(lg.abs().max()))
tok = lg.argmax(-1)
for t in range(300):
    lg = eng.decode_step(tok)
    fin = torch.isfinite(lg).all(-1)
    if not bool(fin.all()):
        bad = (~fin).nonzero().flatten().tolist()
        print(f"step {t} (context {33 + t + 1}): rows with non-finite logits: {bad[:16]} ({len(bad)} rows); nan count {int(torch.isnan(lg).sum())}")
        break
    tok = lg.argmax(-1)
else:
    print("300 steps finite; max|logit|", float(lg.abs().max()))
eng.close()
