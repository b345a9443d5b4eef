"""GPU: parity at the sizes BASELINE.json names, against the oracle run in float32 ON THE GPU (test infrastructure only: the
oracle's plain-torch ops on `cuda` tensors; the 288 GB part holds the 8B model in float32 next to the engine).

  * config 2 at its real size: StarVector-1B, B = 32, seed-1234 weights, 64 teacher-forced steps + a free run (the bench's own
    32-row workload next to the oracle, not a 2-row stand-in);
  * StarVector-8B at FULL depth (SigLIP-L/16-384 tower + 32 StarCoder2-7B layers), B = 2, prompt pass + 16 steps, bf16 and
    fp8-e4m3 decoder weights (fp8 against oracle.fake_quantize_fp8 of the same tensors).

Contract (as in tests/test_gpu_e2e.py): logits within LOGIT_TOL * max|logit| of the oracle's bf16-cast-point mode at every step;
token ids bit-exact wherever the oracle's top-1/top-2 margin is outside 2x that band; a free-running stream may leave the
oracle's only AT an in-band near-tie."""
import dataclasses
import gc

import pytest
import torch

import starvector_amd as sva
from oracle import starvector_oracle as O
from tests.gpu_util import bf, build_engine, dev, rel_err
from tests.test_gpu_e2e import LOGIT_TOL, LOGIT_TOL_FP8

pytestmark = pytest.mark.gpu


def _teacher_forced_gpu(eng, emb, w_dev, cfg, n_new, tag, min_checked, max_near=None, tol=LOGIT_TOL):
    """Oracle (float32 tensors on the GPU, bf16 cast points) greedy stream + logits; the engine is fed the oracle's tokens.
    `tol`: LOGIT_TOL (bf16 weights) or LOGIT_TOL_FP8 (fp8 weights against the fake-quantised oracle).
    Returns (o_toks cpu, margin cpu, band)."""
    S0 = emb.shape[1]
    with torch.no_grad():
        o_toks, o_lg = O.greedy_generate(w_dev, cfg, emb.float(), S0 + n_new, mode="bf16", return_logits=True)
    scale = float(o_lg.abs().max())
    band = 2 * tol * scale
    top2 = o_lg.topk(2, -1).values
    margin = top2[..., 0] - top2[..., 1]                       # [B, n]
    worst, checked, near = 0.0, 0, 0
    B = emb.shape[0]
    for t in range(n_new):
        lg = (eng.prefill(emb) if t == 0 else eng.decode_step(o_toks[:, t - 1].contiguous())).float()
        err = (lg - o_lg[:, t]).abs().amax(-1)                  # per row
        worst = max(worst, float(err.max()))
        assert float(err.max()) <= tol * scale, (f"[{tag}] step {t}: row {int(err.argmax())} logits off by {float(err.max()):.3e} "
                                                 f"(scale {scale:.3e}, tolerance {tol * scale:.3e})")
        am = lg.argmax(-1)
        safe = margin[:, t] > band
        bad = safe & (am != o_toks[:, t])
        assert not bool(bad.any()), (f"[{tag}] step {t}: rows {bad.nonzero().flatten().tolist()} differ from the oracle's token at a "
                                     f"margin outside the band ({margin[:, t][bad].tolist()} > {band:.3e})")
        checked += int(safe.sum())
        near += int((~safe & (am != o_toks[:, t])).sum())
    msg = (f"[{tag}] {n_new} steps x {B} rows: logits max|err| {worst:.3e} (scale {scale:.3e}, {worst / scale:.2e} relative; tolerance {tol:.1e}); {checked}/{B * n_new} positions "
           f"token-exact outside the band, {near} near-tie flips inside it; min margin {float(margin.min()):.3e}")
    print(msg)
    assert checked >= min_checked * B * n_new, msg
    if max_near is not None:
        assert near <= max_near, msg                                # in-band flips are legitimate near-ties; more of them = a regression
    return o_toks.cpu(), margin.cpu(), band


def _free_run_check(got, o_toks, margin, band, tag):
    """Every row either equals the oracle's stream to the end or leaves it AT an in-band near-tie (asserted, with the counts)."""
    B, n = got.shape
    lead = []
    for b in range(B):
        diff = (got[b] != o_toks[b]).nonzero()
        t = int(diff[0]) if diff.numel() else n
        lead.append(t)
        assert t == n or float(margin[b, t]) <= band, (f"[{tag}] row {b} leaves the oracle's stream at step {t}: token {int(got[b, t])} != "
                                                       f"{int(o_toks[b, t])} at margin {float(margin[b, t]):.3e} (band {band:.3e})")
    full = sum(1 for t in lead if t == n)
    print(f"[{tag}] free run: {full}/{B} rows identical to the oracle for all {n} tokens; first in-band departures {sorted(t for t in lead if t < n)}")
    return lead


@pytest.mark.parametrize("exclusive_device", [False, True])
def test_config2_batch32_against_gpu_oracle(exclusive_device):
    """BASELINE config 2 as the bench runs it: 32 images, StarVector-1B, bf16, greedy.  exclusive_device = True is the engine
    configuration bench.py builds (the MLP half of a layer as one launch, gemm.hip mlp_fused_kernel): it meets the oracle here
    first-hand, not only through its bit-identity with the two-launch layer (tests/test_gpu_e2e.py)."""
    cfg = dataclasses.replace(O.OracleConfig(), eos_token_id=-1)
    w = O.make_weights(cfg, seed=1234)
    B, n_new = 32, 64
    eng = build_engine(cfg, w, max_batch=B, max_seq_len=259 + n_new + 8, exclusive_device=exclusive_device)
    w_dev = {k: v.to(dev()) for k, v in w.items()}
    del w
    img = O.synthetic_images(B, 224, seed=1235)
    prompt = torch.tensor([[7, 11]] * B)
    enc = eng.encode_image(bf(img))
    vis = eng.adapter(enc)
    emb = torch.cat([vis, eng.embed_tokens(prompt.to(dev()))], 1)
    S0 = emb.shape[1]
    assert S0 == 259 and S0 + n_new > 320                      # the context crosses a KV page boundary
    with torch.no_grad():
        o_enc = O.image_encoder_forward(w_dev, cfg, img.to(dev()), "bf16")
        o_vis = O.adapter_forward(w_dev, cfg, o_enc, "bf16")
    e1, e2 = rel_err(enc, o_enc), rel_err(vis, o_vis)
    tag = f"config2 B=32{', exclusive_device' if exclusive_device else ''}"
    print(f"[{tag}] encoder rel err {e1:.3e}, adapter rel err {e2:.3e}")
    assert e1 <= 4e-2 and e2 <= 4e-2
    # coverage floor: with random-init weights ~80 % of the 2048 positions have a top-1/top-2 margin outside the band (measured:
    # 1634, min margin 0.0 -- exact ties exist); EVERY one of them must be token-exact and EVERY position's logits in tolerance
    # measured on the round-4 code (profiles/pytest_gpu_r04_final.log; the kernels are bit-deterministic, so these are properties of
    # the code, not of the box): 1634 / 2048 positions outside the band, 8 in-band near-tie flips, 26 / 32 free-running rows follow the
    # oracle to the end (the other six leave it AT an in-band near-tie: steps 1, 8, 15, 33, 42, 45).  Floors = measured minus a little.
    o_toks, margin, band = _teacher_forced_gpu(eng, emb, w_dev, cfg, n_new, tag, 0.78, max_near=16)
    got = eng.generate(emb, max_length=S0 + n_new, eos_token_id=-1, pad_token_id=cfg.pad_token_id).cpu()
    lead = _free_run_check(got, o_toks, margin, band, tag)
    assert sum(1 for t in lead if t == n_new) >= 24, f"only {sum(1 for t in lead if t == n_new)}/{B} rows follow the oracle to the end: {lead}"
    eng.close()
    del w_dev
    gc.collect(); torch.cuda.empty_cache()


@pytest.mark.parametrize("weights", ["bf16", "fp8_e4m3"])
def test_starvector_8b_full_depth_against_gpu_oracle(weights):
    """BASELINE configs 4 / 5 at full depth: SigLIP tower (24 layers) + 32 StarCoder2 layers, 7.6 B parameters.  The weights are
    drawn on the GPU (seconds instead of minutes of host RNG) and handed to BOTH the engine and the float32 oracle."""
    cfg = dataclasses.replace(O.OracleConfig.starvector_8b(), eos_token_id=-1)
    B, n_new = 2, 17
    ec = sva.EngineConfig.starvector_8b(max_batch=2, max_seq_len=578 + 72)
    ec.weight_dtype = weights
    eng = sva.HipEngine(ec)
    w_dev = {}
    for name, tns in O.iter_weights(cfg, seed=91, init="parity", device=dev()):
        eng.load_weight(name, tns.to(torch.bfloat16))
        w_dev[name] = tns                                       # float32, bf16-exact values
    eng.load_state_dict({})
    w_dev[O.K_LMH] = w_dev[O.embed_key(cfg)]
    if weights == "fp8_e4m3":
        w_dev = O.fake_quantize_fp8(w_dev, cfg)                 # dequant(quant(W)) with torch's own float8_e4m3fn cast
    img = O.synthetic_images(B, 384, seed=92)
    prompt = torch.tensor([[7, 11]] * B)
    enc = eng.encode_image(bf(img))
    vis = eng.adapter(enc)
    emb = torch.cat([vis, eng.embed_tokens(prompt.to(dev()))], 1)
    assert emb.shape == (B, 578, 4608)
    with torch.no_grad():
        o_enc = O.image_encoder_forward(w_dev, cfg, img.to(dev()), "bf16")
        o_vis = O.adapter_forward(w_dev, cfg, o_enc, "bf16")
    e1, e2 = rel_err(enc, o_enc), rel_err(vis, o_vis)
    print(f"[8b full depth, {weights}] siglip (24 layers) rel err {e1:.3e}, adapter rel err {e2:.3e}")
    assert e1 <= 4e-2 and e2 <= 4e-2
    tag = f"8b full depth, {weights}"
    o_toks, margin, band = _teacher_forced_gpu(eng, emb, w_dev, cfg, n_new, tag, 0.7, tol=LOGIT_TOL_FP8 if weights == "fp8_e4m3" else LOGIT_TOL)
    got = eng.generate(emb, max_length=578 + n_new, eos_token_id=-1, pad_token_id=0).cpu()
    _free_run_check(got, o_toks, margin, band, tag)
    eng.close()
    del w_dev
    gc.collect(); torch.cuda.empty_cache()
