"""GPU: BASELINE configs 4 and 5 at THEIR batch sizes against the oracle.

  * config 5: StarVector-8B text2svg, 64 rows, fp8-e4m3 decoder weights, full depth (32 StarCoder2 layers), caption prompt of
    33 ids as bench.py builds it -- the 64-row engine's decode path end to end: two-row-tile GEMMs with the engine's
    (column tiles, split-K) plan, the 512-block attention launch, rows 32..63 of every kernel.  Second case: a 250-id prompt, so
    that the 12 teacher-forced steps cross context 256 where the attention goes from one block per (row, KV head) to two
    context splits + ticket merge (the place where round 4's in-band hand-off produced NaNs for every row);
  * config 4: StarVector-8B im2svg, 16 rows, bf16, full depth, the 578-row image prompt, top-k 50 + top-p 0.95: teacher-forced
    logits, the warpers' kept set per step (the oracle's warp_scores = HF's TopK / TopP warpers bit for bit), the engine's
    sampler on those logits, and a sampled stream replayed through the oracle.

The reference side: starvector/model/models/starvector_v2.py:45-57 (StarCoder2 decoder behind the same generate),
starvector_base.py:223-241 (generate kwargs: top_p, temperature -- and transformers 4.49's implicit top_k = 50), :297-330
(generate_text2svg).  The oracle runs in float32 on the GPU (test infrastructure; the weights are drawn on the GPU and handed
to both sides); fp8 is compared with oracle.fake_quantize_fp8 of the same tensors."""
import dataclasses
import gc

import pytest
import torch

import starvector_amd as sva
from oracle import starvector_oracle as O
from tests.gpu_util import bf, dev, rel_err
from tests.test_gpu_e2e import LOGIT_TOL, LOGIT_TOL_FP8
from tests.test_gpu_parity_fullsize import _free_run_check, _teacher_forced_gpu

pytestmark = pytest.mark.gpu


def _engine_and_oracle_8b(max_batch, max_seq_len, weights, seed, exclusive_device=False):
    cfg = dataclasses.replace(O.OracleConfig.starvector_8b(), eos_token_id=-1)
    ec = sva.EngineConfig.starvector_8b(max_batch=max_batch, max_seq_len=max_seq_len)
    ec.weight_dtype = weights
    ec.exclusive_device = exclusive_device
    eng = sva.HipEngine(ec)
    w_dev = {}
    for name, tns in O.iter_weights(cfg, seed=seed, init="parity", device=dev()):
        eng.load_weight(name, tns.to(torch.bfloat16))
        w_dev[name] = tns                                       # float32, bf16-exact values
    eng.load_state_dict({})
    w_dev[O.K_LMH] = w_dev[O.embed_key(cfg)]
    if weights == "fp8_e4m3":
        w_dev = O.fake_quantize_fp8(w_dev, cfg)                 # dequant(quant(W)) with torch's own float8_e4m3fn cast
    return cfg, eng, w_dev


@pytest.mark.parametrize("S0", [33, 250])
def test_config5_batch64_fp8_text2svg_full_depth_against_gpu_oracle(S0):
    """64 live rows, fp8 weights, text-only prompt, full depth.  S0 = 33: bench.py's caption prompt (32 ids + <svg-start>);
    S0 = 250: contexts 250 -> 262 cross the one-block -> two-split switch of the 64-row attention launch at 256 keys."""
    B, n_new = 64, 12
    cfg, eng, w_dev = _engine_and_oracle_8b(B, S0 + n_new + 20, "fp8_e4m3", seed=93)
    ids = torch.randint(0, 49152, (B, S0), generator=torch.Generator().manual_seed(94 + S0))
    emb = eng.embed_tokens(ids.to(dev()))                       # generate_text2svg: no image encoder (starvector_base.py:297-330)
    assert emb.shape == (B, S0, 4608)
    tag = f"config5 B=64 fp8 text2svg S0={S0}"
    # coverage floor: measured 489 / 768 (S0 = 33) and 542 / 768 (S0 = 250) positions with a margin outside the fp8 band (random-init weights)
    o_toks, margin, band = _teacher_forced_gpu(eng, emb, w_dev, cfg, n_new, tag, 0.6, tol=LOGIT_TOL_FP8)
    got = eng.generate(emb, max_length=S0 + n_new, eos_token_id=-1, pad_token_id=0).cpu()
    lead = _free_run_check(got, o_toks, margin, band, tag)
    # every row is its own request: rows 0 / 31 / 32 / 63 alone (other row tile, other blocks) give the same stream
    for r in (0, 31, 32, 63):
        solo = eng.generate(emb[r:r + 1].contiguous(), max_length=S0 + n_new, eos_token_id=-1, pad_token_id=0).cpu()
        assert torch.equal(solo[0], got[r]), f"[{tag}] row {r} alone != row {r} inside the 64-row batch"
    assert sum(1 for t in lead if t == n_new) >= B // 2, f"[{tag}] only {sum(1 for t in lead if t == n_new)}/{B} rows follow the oracle"
    eng.close()
    del w_dev
    gc.collect(); torch.cuda.empty_cache()


def _kept_relaxed(lg_o, tok, top_k, top_p, slack):
    """Could `tok` be inside HF's TopK -> TopP kept set if every oracle score moved by at most `slack`?  lg_o: [V] scores after
    temperature.  TopK: tok's score is within `slack` of the k-th largest; TopP (HF removes the ascending-sorted tokens whose
    cumulative probability is <= 1 - top_p): the mass of the tokens that are more probable than tok by more than `slack`,
    renormalised over the top-k set, stays below top_p."""
    kth = torch.topk(lg_o, top_k).values[-1]
    if float(lg_o[tok]) < float(kth) - slack:
        return False
    keep = lg_o >= kth - slack
    p = torch.softmax(lg_o.masked_fill(~keep, float("-inf")), -1)
    above = float(p[lg_o > lg_o[tok] + slack].sum())
    return above < top_p + 1e-6


@pytest.mark.parametrize("exclusive_device", [False, True])
def test_config4_batch16_top_p_full_depth_against_gpu_oracle(exclusive_device):
    """16 rows, bf16, SigLIP tower + adapter + 32 layers, 578-row prompt, the reference's sampling (top_p 0.95 + the implicit
    top_k 50 of transformers 4.49, temperature 1).  exclusive_device: the configuration bench.py --model 8b measures -- its decode layers run
    the ln_1 row update inside the c_attn launch (rowln_cattn_kernel<9, true>); it meets the oracle first-hand, like config 2's."""
    from starvector_amd import engine as E
    B, n_new, TOP_K, TOP_P = 16, 12, 50, 0.95
    cfg, eng, w_dev = _engine_and_oracle_8b(B, 578 + 72, "bf16", seed=95, exclusive_device=exclusive_device)
    img = O.synthetic_images(B, 384, seed=96)
    prompt = torch.tensor([[7, 11]] * B)
    enc = eng.encode_image(bf(img))
    vis = eng.adapter(enc)
    emb = torch.cat([vis, eng.embed_tokens(prompt.to(dev()))], 1)
    assert emb.shape == (B, 578, 4608)
    with torch.no_grad():
        o_enc = O.image_encoder_forward(w_dev, cfg, img.to(dev()), "bf16")
        o_vis = O.adapter_forward(w_dev, cfg, o_enc, "bf16")
    e1, e2 = rel_err(enc, o_enc), rel_err(vis, o_vis)
    tag = "config4 B=16 bf16 im2svg"
    print(f"[{tag}] siglip (24 layers) rel err {e1:.3e}, adapter rel err {e2:.3e}")
    assert e1 <= 4e-2 and e2 <= 4e-2
    # (1) teacher-forced logits along the oracle's greedy stream, every step, all 16 rows
    o_toks, margin, band = _teacher_forced_gpu(eng, emb, w_dev, cfg, n_new, tag, 0.7)
    # (2) the kept set of TopK(50) -> TopP(0.95) per step: the engine's logits and the oracle's logits through the SAME warpers
    #     (oracle.warp_scores == HF's warper classes, oracle/make_golden.py::run_sampling_cases); a token may be in one set and
    #     not the other only where the oracle's scores put it within the logit band of a threshold; the engine's sampler on
    #     the engine's logits never draws outside the set those logits define (exact)
    S0 = emb.shape[1]
    with torch.no_grad():
        _, o_lg = O.greedy_generate(w_dev, cfg, emb.float(), S0 + n_new, mode="bf16", return_logits=True)
    o_lg = o_lg.cpu()
    slack = 2 * LOGIT_TOL * float(o_lg.abs().max())
    sym, outside, draws, tied_extra = 0, 0, 0, 0
    for t in range(n_new):
        lg = (eng.prefill(emb) if t == 0 else eng.decode_step(o_toks[:, t - 1].to(dev()).contiguous())).float()
        kept_e = O.warp_scores(lg.cpu(), 1.0, TOP_P, TOP_K) > float("-inf")
        kept_o = O.warp_scores(o_lg[:, t], 1.0, TOP_P, TOP_K) > float("-inf")
        for b in range(B):
            for tok in (kept_e[b] ^ kept_o[b]).nonzero().flatten().tolist():
                sym += 1
                assert _kept_relaxed(o_lg[b, t], tok, TOP_K, TOP_P, slack), (
                    f"[{tag}] step {t} row {b}: token {tok} is in one kept set only and NOT within the band of a warper threshold")
        rows = lg.repeat_interleave(8, 0).contiguous()                          # 8 draws per row and step, own random streams
        s = E.op_sample_top_p(rows, 1.0, TOP_P, seed=1000 + t, step=t, top_k=TOP_K).cpu().long().view(B, 8)
        draws += s.numel()
        # bf16-rounded logits tie often.  Scores tied ACROSS the top-p cut: HF drops them one by one in torch.sort's order (which of
        # the equal tokens survive is an accident of the sort), the engine keeps or drops equal scores together (sampling.hip rule 5;
        # tests/test_gpu_beam.py compares tie cases with top_p = 1 for that reason).  So the set the sampler may draw from is HF's kept
        # set closed under ties: every token whose score equals the smallest kept score.
        lg_c = lg.cpu()
        floor = torch.where(kept_e, lg_c, torch.full_like(lg_c, float("inf"))).amin(-1, keepdim=True)
        closed = lg_c >= floor
        tied_extra += int((closed & ~kept_e).sum())
        outside += int((~torch.gather(closed, 1, s)).sum())
    print(f"[{tag}] top-k {TOP_K} + top-p {TOP_P}: {n_new} steps x {B} rows, kept-set symmetric difference engine vs oracle {sym} tokens "
          f"(all within the band of a threshold); {draws} sampler draws, {outside} outside the tie-closed kept set "
          f"({tied_extra} tokens are tied with the smallest kept score)")
    assert outside == 0
    # (3) a sampled stream of the whole path (sv_generate, do_sample) replayed through the oracle: every sampled token lies in
    #     the oracle's kept set for ITS context (relaxed by the band), for all 16 rows
    n_s = 24
    got = eng.generate(emb, max_length=S0 + n_s, do_sample=True, temperature=1.0, top_p=TOP_P, top_k=TOP_K, seed=7,
                       eos_token_id=-1, pad_token_id=0).cpu()
    assert got.shape == (B, n_s)
    assert torch.equal(got, eng.generate(emb, max_length=S0 + n_s, do_sample=True, temperature=1.0, top_p=TOP_P, top_k=TOP_K, seed=7,
                                         eos_token_id=-1, pad_token_id=0).cpu())             # reproducible per seed
    strict = 0
    with torch.no_grad():
        logits, cache = O.decoder_prefill(w_dev, cfg, emb.float(), "bf16")
        for t in range(n_s):
            lg_t = logits.float().cpu()
            kept = O.warp_scores(lg_t, 1.0, TOP_P, TOP_K) > float("-inf")
            for b in range(B):
                tok = int(got[b, t])
                strict += int(kept[b, tok])
                assert _kept_relaxed(lg_t[b], tok, TOP_K, TOP_P, slack), (
                    f"[{tag}] sampled stream, step {t} row {b}: token {tok} is outside the oracle's kept set by more than the band")
            if t + 1 < n_s:
                logits, cache = O.decoder_decode_step(w_dev, cfg, got[:, t].to(dev()), cache, "bf16")
    print(f"[{tag}] sampled stream: {n_s} tokens x {B} rows, {strict}/{B * n_s} inside the oracle's strict kept set, the rest within the band; "
          f"{len(set(got.flatten().tolist()))} distinct tokens")
    assert strict >= 0.9 * B * n_s
    eng.close()
    del w_dev
    gc.collect(); torch.cuda.empty_cache()
