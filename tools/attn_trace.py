#!/usr/bin/env python3
"""Where the time of the decode attention launch goes: per-block wall-clock stamps of one launch in the middle of a decode step of
BASELINE config 2 (B = 32), at the contexts given on the command line (default 300 800 1283), in microseconds after the first block's start."""
import os
import sys

os.environ["SV_ATTN_TRACE"] = "1"
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import starvector_amd as sva  # noqa: E402
from bench import synthetic_images  # noqa: E402

dev = torch.device("cuda", 0)
B = 32
ctxs = [int(a) for a in sys.argv[1:]] or [300, 800, 1283]
eng = sva.HipEngine(sva.EngineConfig(max_batch=B, max_seq_len=max(ctxs) + 8))
eng.load_random_weights(seed=1234)
img = synthetic_images(torch, B, 224, seed=0).to(dev)
prompt = torch.tensor([[7, 11]] * B, dtype=torch.long, device=dev)
emb = torch.cat([eng.adapter(eng.encode_image(img)), eng.embed_tokens(prompt)], 1)
q = lambda x: [round(float(x.quantile(p)), 2) for p in (0.0, 0.5, 0.9, 1.0)]
for ctx in ctxs:
    for rep in range(2):
        eng.generate(emb, max_length=ctx, eos_token_id=-1, pad_token_id=49152)
    tr = eng.debug_attn_trace().double()
    live = tr[tr[:, 0] > 0]
    t0 = live[:, 0].min()
    us = (live[:, :7] - t0) / 100.0
    merger = live[:, 6] > 0
    act, groups = int(live[0, 7]), int(live[0, 8])
    print(f"--- context {ctx}: {live.shape[0]} active blocks ({act} splits per sequence, {groups} key groups), min / median / p90 / max over blocks")
    for k, n in enumerate(["start", "pos + table landed, KV requested", "q summed, in LDS", "key groups processed", "partial stored + drained", "ticket drawn"]):
        print(f"  {n:34s} {q(us[:, k])}")
    print(f"  {'merged + written (merging block)':34s} {q(us[merger, 6])}   launch span {float(us[merger, 6].max()):.2f} us")
    seg = lambda a, b2: float((us[:, b2] - us[:, a]).median())
    print(f"  segments (median): pos/table {seg(0, 1):.2f}  slab sum {seg(1, 2):.2f}  KV + MFMA {seg(2, 3):.2f}  block merge + partial store {seg(3, 4):.2f}  "
          f"ticket {seg(4, 5):.2f}  final merge {float((us[merger, 6] - us[merger, 5]).median()):.2f}")
eng.close()
