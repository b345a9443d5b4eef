#!/usr/bin/env python3
"""Round-5 diagnosis: the fused row-update + c_attn launch gives up (code 4) in the SECOND sv_generate call of bench.py, i.e. when the kept
hipGraph is replayed by a later call.  Two identical calls on one exclusive engine, graph and eager, short and long."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import starvector_amd as sva  # noqa: E402

dev = torch.device("cuda", 0)
B = 32
for n_new, nograph in ((1024, False), (1024, True)):
    if nograph:
        os.environ["SV_NO_GRAPH"] = "1"
    else:
        os.environ.pop("SV_NO_GRAPH", None)
    eng = sva.HipEngine(sva.EngineConfig(max_batch=B, max_seq_len=259 + n_new, exclusive_device=True))
    eng.load_random_weights(seed=1234)
    img = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).to(dev)
    prompt = torch.tensor([[7, 11]] * B, dtype=torch.long, device=dev)
    outs = []
    for call in range(4):
        emb = eng.prepare_inputs(eng.encode_image(img), prompt)
        try:
            outs.append(eng.generate(emb, max_length=emb.shape[1] + n_new, eos_token_id=-1, pad_token_id=49152).cpu())
            print(f"n_new {n_new} nograph {nograph} call {call}: ok, graph={eng.last_timing()['graph']}, equal to call 0: {torch.equal(outs[-1], outs[0])}", flush=True)
        except Exception as ex:
            print(f"n_new {n_new} nograph {nograph} call {call}: FAILED {str(ex)[:120]}", flush=True)
            break
    eng.close()
