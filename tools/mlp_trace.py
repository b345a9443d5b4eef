#!/usr/bin/env python3
"""Where the time of the fused MLP launch goes (SV_EXP bit 128): per-block wall-clock stamps of one launch in the middle of a decode
step of BASELINE config 2 (B = 32, context ~400), printed as distributions in microseconds relative to the earliest block start."""
import os
import sys

os.environ["SV_MLP_TRACE"] = "1"
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import starvector_amd as sva  # noqa: E402
from bench import synthetic_images  # noqa: E402

dev = torch.device("cuda", 0)
B = 32
eng = sva.HipEngine(sva.EngineConfig(max_batch=B, max_seq_len=259 + 160))
eng.load_random_weights(seed=1234)
img = synthetic_images(torch, B, 224, seed=0).to(dev)
prompt = torch.tensor([[7, 11]] * B, dtype=torch.long, device=dev)
emb = torch.cat([eng.adapter(eng.encode_image(img)), eng.embed_tokens(prompt)], 1)
mask = int(sys.argv[1]) if len(sys.argv) > 1 else 128       # SV_EXP mask: 128 = the fused launch
eng.set_exp(mask)
print(f"=== SV_EXP {mask}")
for rep in range(3):
    eng.generate(emb, max_length=emb.shape[1] + 128, eos_token_id=-1, pad_token_id=49152)
    tr = eng.debug_mlp_trace().double()
    t0 = tr[:, 0].min()
    us = (tr[:, :5] - t0) / 100.0
    names = ["start", "c_fc loop done", "tile published", "slice complete", "end"]
    q = lambda x: [round(float(x.quantile(p)), 2) for p in (0.0, 0.5, 0.9, 1.0)]
    print(f"--- rep {rep}: {tr.shape[0]} blocks, launch span {float(us[:, 4].max()):.2f} us (min / median / p90 / max over blocks, us after the first block started)")
    for k, n in enumerate(names):
        print(f"  {n:16s} {q(us[:, k])}")
    print(f"  segments (median): loop {float((us[:,1]-us[:,0]).median()):.2f}  reduce+publish {float((us[:,2]-us[:,1]).median()):.2f}  "
          f"wait for the slice {float((us[:,3]-us[:,2]).median()):.2f} (max {float((us[:,3]-us[:,2]).max()):.2f})  phase 2 {float((us[:,4]-us[:,3]).median()):.2f}")
    xcc = tr[:, 5].long()
    print("  per XCC id: blocks", [int((xcc == x).sum()) for x in range(8)], " median end", [round(float(us[xcc == x, 4].median()), 2) if int((xcc == x).sum()) else None for x in range(8)])
eng.close()
