"""GPU: fp8 (OCP e4m3) decoder weights -- BASELINE config 5's weight format.

Not a reference numerics mode (the reference reaches fp8 only through vLLM, which is not in tree), so the statement tested is
self-contained: the engine with `weight_dtype="fp8_e4m3"` computes exactly the bf16 path on dequant(quant(W)) -- one scale per
output row, RNE to e4m3 -- which is what `oracle.fake_quantize_fp8` builds with torch's own float8_e4m3fn cast."""
import os
import time

import pytest
import torch

from oracle import starvector_oracle as O
from tests.gpu_util import bf, build_engine, dev
from tests.test_gpu_e2e import LOGIT_TOL, _teacher_forced_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K,sk", [(32, 64, 256, 1), (32, 2304, 2048, 4), (7, 516, 256, 2), (40, 2048, 8192, 4),
                                       (32, 1024, 256, 1), (3, 256, 1024, 8), (32, 5632, 4608, 2), (16, 49157, 2048, 1)])
def test_fp8_skinny_gemm_matches_torch_fake_quant(M, N, K, sk):
    """The decode GEMM with e4m3 weights against torch: scales, quantised values (through the product) and the
    contraction.  Products x * q are exact in fp32, so only the summation order differs."""
    from starvector_amd import engine as E
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16()
    W[min(3, N - 1)] = 0                                              # an all-zero row: scale 1, q 0
    b = (0.1 * torch.randn(N, generator=g)).bfloat16()
    y, sc = E.op_linear_skinny_fp8(bf(x), bf(W), bf(b), splitk=sk)
    amax = W.float().abs().amax(1)
    ref_sc = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.equal(sc.cpu(), ref_sc)
    q = (W.float() / ref_sc[:, None]).to(torch.float8_e4m3fn).float()
    ref = x.float() @ (q * ref_sc[:, None]).T + b.float()
    err = float((y.cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err


@pytest.mark.parametrize("arch", ["v1", "v2"])
def test_fp8_weights_match_fake_quantised_oracle(arch):
    cfg = O.OracleConfig.tiny() if arch == "v1" else O.OracleConfig.tiny_v2()
    cfg = __import__("dataclasses").replace(cfg, eos_token_id=-1)
    w = O.make_weights(cfg, seed=55)
    wq = O.fake_quantize_fp8(w, cfg)
    eng = build_engine(cfg, w, max_batch=4, max_seq_len=96, weight_dtype="fp8_e4m3")
    img = bf(O.synthetic_images(3, cfg.image_size, seed=56))
    prompt = torch.tensor([[7, 11]] * 3, device=dev())
    emb = torch.cat([eng.adapter(eng.encode_image(img)), eng.embed_tokens(prompt)], 1)
    # every step (prefill GEMMs with bf16(q) + scale epilogue, decode GEMMs streaming fp8) against the quantised oracle
    worst, scale, checked, near, o_toks, _ = _teacher_forced_check(eng, emb, wq, cfg, 24)
    print(f"[fp8 {arch}] logits max|err| {worst:.3e} vs fake-quantised oracle (scale {scale:.3e}); {checked} exact, {near} near-tie flips")
    assert checked > 0
    # ... and it really is the quantised model: the unquantised oracle is further away than the tolerance somewhere
    lg_q = O.decoder_prefill(wq, cfg, emb.float().cpu(), "bf16")[0]
    lg_b = O.decoder_prefill(w, cfg, emb.float().cpu(), "bf16")[0]
    got = eng.prefill(emb).float().cpu()
    assert float((got - lg_q).abs().max()) < float((got - lg_b).abs().max())
    # generation: deterministic, graph == eager, same stream as teacher forcing predicts where margins allow
    kw = dict(max_length=emb.shape[1] + 70, eos_token_id=-1, pad_token_id=cfg.pad_token_id)
    a = eng.generate(emb, **kw).cpu()
    assert a.shape == (3, 70) and torch.equal(a, eng.generate(emb, **kw).cpu())
    os.environ["SV_NO_GRAPH"] = "1"
    try:
        assert torch.equal(a, eng.generate(emb, **kw).cpu())
    finally:
        os.environ.pop("SV_NO_GRAPH", None)
    eng.close()


def test_fp8_weights_full_size_speed():
    """StarVector-1B shapes, batch 32: the decode step with fp8 weights against the bf16 engine ON THE SAME PIPELINE (same box, same
    process).  fp8 weights run the 7-launch layer (the LayerNorm fold of the 6-launch layer would have to re-quantise gamma * W: another
    numerical contract than the fake-quant oracle), so the like-for-like comparison is bf16 with SV_EXP=2; the bf16 default (6 launches
    per layer) is printed next to it -- at this size the step is launch-bound and it is the faster of the three."""
    cfg = O.OracleConfig()
    w = O.make_weights(cfg, seed=7, init="std002")
    res = {}
    for tag, wd, exp in (("bf16 6-launch", "bf16", None), ("bf16 7-launch", "bf16", "2"), ("fp8 7-launch", "fp8_e4m3", None)):
        old_exp = os.environ.get("SV_EXP")
        if exp is not None:
            os.environ["SV_EXP"] = exp                                  # read once at sv_create
        try:
            eng = build_engine(cfg, w, max_batch=32, max_seq_len=259 + 130, weight_dtype=wd)
        finally:
            if exp is not None:
                os.environ.pop("SV_EXP", None) if old_exp is None else os.environ.__setitem__("SV_EXP", old_exp)
        img = bf(O.synthetic_images(32, 224, seed=8))
        emb = torch.cat([eng.adapter(eng.encode_image(img)), eng.embed_tokens(torch.tensor([[7, 11]] * 32, device=dev()))], 1)
        kw = dict(max_length=259 + 128, eos_token_id=-1, pad_token_id=cfg.pad_token_id)
        eng.generate(emb, **kw)
        t = eng.generate(emb, **kw).cpu()
        tm = eng.last_timing()
        assert t.shape == (32, 128) and tm["graph"]
        res[tag] = tm["decode_ms"] / tm["decode_steps"]
        eng.close()
    print("[fp8 1B] decode step: " + ", ".join(f"{k} {v * 1e3:.0f} us" for k, v in res.items()))
    assert res["fp8 7-launch"] < res["bf16 7-launch"], res
    assert res["bf16 6-launch"] < res["bf16 7-launch"], res
