"""CPU: the oracle against the golden vectors minted from the reference (oracle/make_golden.py),
plus the generation semantics that define parity."""
import dataclasses
import os

import importlib.util

import pytest
import torch
from safetensors.torch import load_file

from oracle import starvector_oracle as O


def _load(golden_dir, name):
    return load_file(os.path.join(golden_dir, name + ".safetensors"))


@pytest.mark.parametrize("name,norm", [("tiny_b3", "layer_norm"), ("tiny_bn_b2", "batch_norm")])
def test_oracle_reproduces_reference_goldens(golden_dir, name, norm):
    g = _load(golden_dir, name)
    seed, B, n_new = [int(x) for x in g["meta"]]
    cfg = dataclasses.replace(O.OracleConfig.tiny(), adapter_norm=norm)
    w = O.apply_fixture_weights(O.make_weights(cfg, seed=seed), cfg, g)     # tiny_b3 carries a fitted embedding table
    assert torch.equal(O.synthetic_images(B, cfg.image_size, seed=seed + 1), g["image"])
    enc = O.image_encoder_forward(w, cfg, g["image"])
    vis = O.adapter_forward(w, cfg, enc)
    emb = O.prepare_generation_inputs(w, cfg, g["image"], g["prompt_ids"])
    logits0, _ = O.decoder_prefill(w, cfg, emb)
    # float32 restatement vs the reference's own modules: float32 round-off only
    torch.testing.assert_close(enc, g["enc"], rtol=0, atol=2e-5 * float(g["enc"].abs().max()))
    torch.testing.assert_close(vis, g["vis"], rtol=0, atol=2e-5 * float(g["vis"].abs().max()))
    torch.testing.assert_close(emb, g["emb"], rtol=0, atol=2e-5 * float(g["emb"].abs().max()))
    torch.testing.assert_close(logits0, g["logits0"], rtol=0, atol=5e-5 * max(1.0, float(g["logits0"].abs().max())))
    toks, lg = O.greedy_generate(w, cfg, emb, emb.shape[1] + n_new, return_logits=True)
    assert torch.equal(toks, g["tokens"])                  # integer token ids: bit-exact
    if "wte" in g:
        # the designed stream: diverse, and decided by margins far outside bf16 noise
        top2 = lg.topk(2, -1).values
        assert float(((top2[..., 0] - top2[..., 1]) / lg.abs().max()).min()) >= 0.1
        assert len(set(toks.flatten().tolist())) >= 20
        tb = O.greedy_generate(w, cfg, O.prepare_generation_inputs(w, cfg, g["image"], g["prompt_ids"], "bf16"),
                               emb.shape[1] + n_new, mode="bf16")
        assert torch.equal(tb, g["tokens"])                # the bf16 cast points do not move a single token
    full = O.generate_im2svg_tokens(w, cfg, g["image"], g["prompt_ids"], emb.shape[1] + n_new)
    assert torch.equal(full, torch.cat([g["prompt_ids"], g["tokens"]], 1))


def test_stop_eos_pad_semantics_match_hf(golden_dir):
    g = _load(golden_dir, "tiny_stop")
    seed, B, n_new, eos = [int(x) for x in g["meta"]]
    cfg = dataclasses.replace(O.OracleConfig.tiny(), eos_token_id=eos)
    w = O.apply_fixture_weights(O.make_weights(cfg, seed=seed), cfg, g)
    emb = O.prepare_generation_inputs(w, cfg, g["image"], g["prompt_ids"])
    toks = O.greedy_generate(w, cfg, emb, emb.shape[1] + n_new, stop_ids=g["stop_ids"].tolist())
    assert torch.equal(toks, g["tokens"])
    assert toks.shape[1] < n_new                           # row-0 stop ended the whole batch early
    assert (toks == cfg.pad_token_id).any()                # a finished row emitted pad afterwards
    assert toks[0, -len(g["stop_ids"]):].tolist() == g["stop_ids"].tolist()


def test_max_length_includes_prompt():
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=3)
    emb = O.prepare_generation_inputs(w, cfg, O.synthetic_images(1, cfg.image_size, 4), torch.tensor([[7, 11]]))
    S0 = emb.shape[1]
    assert O.greedy_generate(w, cfg, emb, S0 + 5).shape == (1, 5)
    with pytest.raises(ValueError):
        O.greedy_generate(w, cfg, emb, S0)


def test_kv_cache_equals_full_recompute():
    """decode with the cache == full forward over the grown sequence (the oracle's own consistency)."""
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=5)
    emb = O.prepare_generation_inputs(w, cfg, O.synthetic_images(2, cfg.image_size, 6), torch.tensor([[7, 11]] * 2))
    toks, lg = O.greedy_generate(w, cfg, emb, emb.shape[1] + 4, return_logits=True)
    cur = emb
    for t in range(4):
        full, _ = O.decoder_prefill(w, cfg, cur)
        torch.testing.assert_close(full, lg[:, t], rtol=0, atol=2e-4)
        cur = torch.cat([cur, w[O.P_DEC + "wte.weight"][toks[:, t]].unsqueeze(1)], 1)


def test_bf16_mode_tracks_fp32():
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=7)
    img = O.synthetic_images(2, cfg.image_size, 8)
    a = O.image_encoder_forward(w, cfg, img, "fp32")
    b = O.image_encoder_forward(w, cfg, img, "bf16")
    assert float((a - b).abs().max() / a.abs().max()) < 3e-2
    assert torch.equal(b, b.to(torch.bfloat16).float())    # bf16-exact values


def test_top_p_filter_properties():
    g = torch.Generator().manual_seed(0)
    lg = 3 * torch.randn(4, 100, generator=g)
    p = O.top_p_filtered_probs(lg, temperature=0.7, top_p=0.9)
    torch.testing.assert_close(p.sum(-1), torch.ones(4), rtol=0, atol=1e-5)
    full = torch.softmax(lg / 0.7, -1)
    for r in range(4):
        kept = p[r] > 0
        assert kept[full[r].argmax()]                       # the most probable token always survives
        assert full[r][kept].sum() >= 0.9 - 1e-6            # kept mass reaches top_p
        assert full[r][kept].min() >= full[r][~kept].max()  # a probability threshold separates them
    # top_p = 1 keeps everything
    assert (O.top_p_filtered_probs(lg, 1.0, 1.0) > 0).all()


def test_v2_oracle_reproduces_reference_goldens(golden_dir):
    """StarVector-8B op graph (SigLIP tower, StarCoder2: RoPE, GQA) at reduced shapes vs HF's own classes."""
    g = _load(golden_dir, "tiny_v2_b2")
    seed, B, n_new = [int(x) for x in g["meta"]]
    cfg = O.OracleConfig.tiny_v2()
    w = O.make_weights(cfg, seed=seed)
    enc = O.image_encoder_forward(w, cfg, g["image"])
    vis = O.adapter_forward(w, cfg, enc)
    emb = O.prepare_generation_inputs(w, cfg, g["image"], g["prompt_ids"])
    logits0, _ = O.decoder_prefill(w, cfg, emb)
    torch.testing.assert_close(enc, g["enc"], rtol=0, atol=2e-5 * float(g["enc"].abs().max()))
    torch.testing.assert_close(vis, g["vis"], rtol=0, atol=2e-5 * float(g["vis"].abs().max()))
    torch.testing.assert_close(logits0, g["logits0"], rtol=0, atol=5e-5 * max(1.0, float(g["logits0"].abs().max())))
    assert torch.equal(O.greedy_generate(w, cfg, emb, emb.shape[1] + n_new), g["tokens"])
    assert enc.shape == (B, cfg.query_length, cfg.vit_width) and cfg.query_length == cfg.n_patches     # no cls token


def test_repetition_penalty_matches_hf(golden_dir):
    g = _load(golden_dir, "tiny_reppen")
    seed, B, n_new = [int(x) for x in g["meta"]]
    pen = float(g["penalty"])
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=seed)
    emb = O.prepare_generation_inputs(w, cfg, g["image"], g["prompt_ids"])
    assert torch.equal(O.greedy_generate(w, cfg, emb, emb.shape[1] + n_new, repetition_penalty=pen), g["tokens"])


def test_min_length_matches_hf(golden_dir):
    """min_length above the prompt length: HF subtracts the prompt length and holds EOS at -inf for the remainder."""
    import dataclasses
    g = _load(golden_dir, "tiny_minlen")
    seed, B, n_new, eos, S0 = [int(x) for x in g["meta"]]
    cfg = dataclasses.replace(O.OracleConfig.tiny(), eos_token_id=eos)
    w = O.make_weights(cfg, seed=seed)
    emb = O.prepare_generation_inputs(w, cfg, g["image"], g["prompt_ids"])
    assert emb.shape[1] == S0
    for extra in (0, 3, 6):
        got = O.greedy_generate(w, cfg, emb, S0 + n_new, min_length=S0 + extra)
        assert torch.equal(got, g[f"tokens_{extra}"]), extra
        # under beam search HF applies the same processor to the log-probabilities
        gotb = O.beam_search_generate(w, cfg, emb, S0 + n_new, 2, early_stopping=True, min_length=S0 + extra)
        assert torch.equal(gotb, g[f"beam2_tokens_{extra}"]), extra
    assert not torch.equal(g["tokens_0"][:, :8], g["tokens_3"][:, :8])


def test_beam_search_matches_hf(golden_dir):
    """num_beams > 1 (the reference's default is 2): the restated _beam_search against HF generate on every case of
    tests/golden/tiny_beam (early_stopping True / False / "never", length penalties, EOS, the row-0 stop, the
    repetition penalty on log-probs)."""
    import dataclasses
    g = _load(golden_dir, "tiny_beam")
    seed, B, n_new = [int(x) for x in g["meta"]]
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=seed)
    emb = O.prepare_generation_inputs(w, cfg, g["image"], g["prompt_ids"])
    tags = sorted(k[:-len(".tokens")] for k in g if k.endswith(".tokens"))
    assert len(tags) == 5
    for tag in tags:
        nb, lp, es, eos, pen = g[tag + ".params"].tolist()
        stop = g[tag + ".stop"].tolist() or None
        cfg2 = dataclasses.replace(cfg, eos_token_id=int(eos))
        got = O.beam_search_generate(w, cfg2, emb, emb.shape[1] + n_new, int(nb), length_penalty=lp,
                                     early_stopping={0: False, 1: True, 2: "never"}[int(es)], stop_ids=stop,
                                     repetition_penalty=pen)
        assert got.shape == g[tag + ".tokens"].shape and torch.equal(got, g[tag + ".tokens"]), tag


def test_beam_search_single_beam_is_greedy():
    """num_beams = 1 through the beam bookkeeping degenerates to greedy decoding (no EOS in play)."""
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=5)
    emb = O.prepare_generation_inputs(w, cfg, O.synthetic_images(2, cfg.image_size, seed=6), torch.tensor([[7, 11]] * 2))
    import dataclasses
    cfg2 = dataclasses.replace(cfg, eos_token_id=-1)
    a = O.beam_search_generate(w, cfg2, emb, emb.shape[1] + 10, 1)
    b = O.greedy_generate(w, cfg2, emb, emb.shape[1] + 10)
    assert torch.equal(a, b)


def test_warpers_match_hf_classes():
    """temperature -> top-k -> top-p restated == HF's warper classes (transformers is importable on both boxes)."""
    lp = pytest.importorskip("transformers.generation.logits_process")
    g = torch.Generator().manual_seed(3)
    for (T, tp, tk, mk) in [(0.7, 0.9, 50, 1), (1.0, 0.9, 50, 2), (1.3, 0.5, 5, 2), (1.0, 1.0, 50, 1), (0.8, 0.95, 0, 1)]:
        lg = 3 * torch.randn(5, 257, generator=g)
        ref = lg.clone()
        if T != 1.0:
            ref = lp.TemperatureLogitsWarper(T)(None, ref)
        if tk:
            ref = lp.TopKLogitsWarper(top_k=tk, min_tokens_to_keep=mk)(None, ref)
        if tp < 1:
            ref = lp.TopPLogitsWarper(top_p=tp, min_tokens_to_keep=mk)(None, ref)
        assert torch.equal(O.warp_scores(lg, T, tp, tk, mk), ref)
    # top-k keeps ties with the k-th value; top_k = 0 is off
    tie = torch.tensor([[1.0, 3.0, 3.0, 2.0, 0.0]])
    assert (O.warp_scores(tie, top_k=1) > float("-inf")).sum() == 2
    assert torch.equal(O.warp_scores(tie, top_k=0), tie)


def test_beam_sample_matches_hf(golden_dir):
    """Beam-sample: with torch's RNG seeded as at minting time the restated loop reproduces HF generate's draws."""
    g = _load(golden_dir, "tiny_beam_sample")
    seed, B, n_new, rng = [int(x) for x in g["meta"]]
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=seed)
    emb = O.prepare_generation_inputs(w, cfg, g["image"], g["prompt_ids"])
    for i in range(3):
        nb, T, tp, tk, es, lp = g[f"case{i}.params"].tolist()
        torch.manual_seed(rng)
        got = O.beam_search_generate(w, cfg, emb, emb.shape[1] + n_new, int(nb), length_penalty=lp, early_stopping=bool(es),
                                     do_sample=True, temperature=T, top_p=tp, top_k=int(tk))
        assert got.shape == g[f"case{i}.tokens"].shape and torch.equal(got, g[f"case{i}.tokens"]), i


def test_starcoder2_sliding_window_matches_hf(golden_dir):
    """StarCoder2's sliding window (4096 in bigcode/starcoder2-7b; 24 here so the generation leaves it)."""
    import dataclasses
    g = _load(golden_dir, "tiny_v2_window")
    seed, B, n_new, W = [int(x) for x in g["meta"]]
    cfg = dataclasses.replace(O.OracleConfig.tiny_v2(), sliding_window=W, eos_token_id=-1)
    w = O.make_weights(cfg, seed=seed)
    emb = O.prepare_generation_inputs(w, cfg, g["image"], g["prompt_ids"])
    assert emb.shape[1] < W < emb.shape[1] + n_new
    assert torch.equal(O.greedy_generate(w, cfg, emb, emb.shape[1] + n_new), g["tokens"])
    # a prompt LONGER than the window: the mask applies inside the prompt pass as well (HF's streams for W = 8 < S0)
    cfg8 = dataclasses.replace(cfg, sliding_window=8)
    assert emb.shape[1] > 8
    assert torch.equal(O.greedy_generate(w, cfg8, emb, emb.shape[1] + g["tokens_w8"].shape[1]), g["tokens_w8"])


def test_image_preprocess_restatement_matches_pillow():
    """the numpy integer restatement of ImageTrainProcessor (composite on white, pad, Pillow's
    fixed-point bicubic resize, ToTensor, Normalize) equals Pillow + torch bit for bit (Pillow is the oracle's pin)."""
    pytest.importorskip("PIL")
    from oracle import image_preprocess as P
    P.pin(verbose=False)
    if importlib.util.find_spec("transformers") is not None:
        P.pin_siglip(verbose=False)                          # the v2 tower's HF image processor recipe
    b, t = P.resample_coeffs(448, 224)
    assert t.shape[1] == 9 and int(t[100, :b[100, 1]].sum()) in range((1 << 22) - 8, (1 << 22) + 9)   # taps sum to 1.0
    b, t = P.resample_coeffs(100, 224)                       # upscaling keeps the 2-pixel support
    assert t.shape[1] == 5 and int(b[:, 1].max()) <= 5


def test_scoring_forward_matches_hf(golden_dir):
    """StarVectorForCausalLM.forward: logits of the kept positions against HF's full-sequence logits (tiny_forward)."""
    g = _load(golden_dir, "tiny_forward")
    seed, B, n_ids = [int(x) for x in g["meta"]]
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=seed)
    emb = O.prepare_generation_inputs(w, cfg, g["image"], g["ids"])
    got = O.decoder_forward_logits(w, cfg, emb, 5)
    assert got.shape == g["logits_keep5"].shape and float((got - g["logits_keep5"]).abs().max()) <= 1e-5
    full = O.decoder_forward_logits(w, cfg, emb, 0)
    assert full.shape[1] == emb.shape[1] and float((full[:, -5:] - got).abs().max()) <= 1e-5
    assert float((full[:, -1] - O.decoder_prefill(w, cfg, emb)[0]).abs().max()) <= 1e-6     # last row == the prefill logits
    # HF's run of the batch with row 0 left-padded by 2 (mask 0 0 1 ... 1, positions cumsum(mask) - 1 as transformers 4.49 numbers
    # them): the row alone without its pads -- what the mirror's forward computes for left-padded rows
    alone = O.decoder_forward_logits(w, cfg, emb[:1, 2:], 5)
    assert float((alone[0] - g["logits_leftpad2_row0_keep5"]).abs().max()) <= 1e-5
