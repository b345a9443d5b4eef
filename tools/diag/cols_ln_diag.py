#!/usr/bin/env python3
"""Diagnosis: the in-block LayerNorm GEMM (opt-in full-K decode pipeline) gives wrong rows when it runs late in the test file.
Runs the op before and after a warm-up made of the other tests' ops and reports which rows / columns are off and whether
repeated launches agree."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from starvector_amd import engine as E  # noqa: E402


def bf(t):
    return t.to(torch.bfloat16).cuda()


def ln_ref(h, g, b, eps=1e-5):
    mu = h.mean(-1, keepdim=True)
    var = ((h - mu) ** 2).mean(-1, keepdim=True)
    return ((h - mu) / torch.sqrt(var + eps) * g + b).bfloat16().float()


def case(M=32, N=2304, K=2048):
    g = torch.Generator().manual_seed(11 * M + N + K)
    h = (1.5 * torch.randn(M, K, generator=g) + 0.3).bfloat16().float()
    gam = (1 + 0.1 * torch.randn(K, generator=g)).bfloat16().float()
    bet = (0.1 * torch.randn(K, generator=g)).bfloat16().float()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().float()
    b = (0.1 * torch.randn(N, generator=g)).bfloat16().float()
    return h, gam, bet, W, b, ln_ref(h, gam, bet) @ W.T + b


def probe(tag, n=6):
    h, gam, bet, W, b, ref = case()
    outs = []
    for i in range(n):
        got = E.op_decode_cols(bf(h), bf(W), bf(b), gamma=bf(gam), beta=bf(bet), out_f32=True).cpu()
        outs.append(got)
    scale = float(ref.abs().max())
    for i, got in enumerate(outs):
        err = (got - ref).abs()
        rows = (err.max(dim=1).values > 1e-3 * scale).nonzero().flatten().tolist()
        cols = (err.max(dim=0).values > 1e-3 * scale).nonzero().flatten().tolist()
        print(f"[{tag}] launch {i}: max rel err {float(err.max()) / scale:.2e}; bad rows {rows[:40]}; bad cols {len(cols)} "
              f"(first {cols[:12]}); same bits as launch 0: {torch.equal(got, outs[0])}", flush=True)


probe("cold")
# an engine created and closed in the same process (what every other GPU test file does first)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import starvector_oracle as O  # noqa: E402  (diagnosis tool: weight factory)
from tests.gpu_util import build_engine  # noqa: E402
cfg = O.OracleConfig.tiny()
w = O.make_weights(cfg, seed=5)
eng = build_engine(cfg, w, 4, 96)
probe("engine alive, unused")
img = O.synthetic_images(2, cfg.image_size, seed=6).to(torch.bfloat16).cuda()
emb = torch.cat([eng.adapter(eng.encode_image(img)), eng.embed_tokens(torch.tensor([[7, 11]] * 2).cuda())], 1)
probe("after encode_image / adapter")
eng.generate(emb, max_length=emb.shape[1] + 8, eos_token_id=-1, pad_token_id=0)
probe("after generate")
eng.close()
probe("engine closed")
# warm-up: what the test file runs before this op (big GEMMs, LayerNorm rows, skinny GEMMs, attention)
x = torch.randn(8288, 2048).bfloat16().float()
Wb = (torch.randn(2048, 2048) / 45).bfloat16().float()
for _ in range(6):
    E.op_linear(bf(x), bf(Wb), None, None)
    E.op_linear_skinny(bf(x[:40]), bf(Wb), None, splitk=4)
probe("after big GEMMs")
os.environ["SV_SKINNY_MT2"] = "0"
E.op_linear_skinny(bf(x[:64]), bf(Wb), None, splitk=4)
os.environ.pop("SV_SKINNY_MT2")
probe("after env flip")
