#!/usr/bin/env python3
"""Pin the oracle against the reference itself and mint the golden fixtures.

TEST INFRASTRUCTURE ONLY.  Runs in the BUILD container, where /root/reference exists:

    python oracle/make_golden.py            # validate + (re)write tests/golden/*.safetensors

It imports the reference's own modules
  * starvector.model.image_encoder.clip_model.VisionTransformer / LayerNorm  (clip_model.py:117,167)
  * starvector.model.adapters.adapter.Adapter                                 (adapter.py:12)
and the decoder the reference loads at run time (llm/starcoder.py:33), i.e.
``transformers.GPTBigCodeForCausalLM`` driven through ``GenerationMixin.generate`` exactly as
starvector_base.py:228-241,255 does (greedy: do_sample=False, num_beams=1), composes them the way
``_prepare_generation_inputs`` (starvector_base.py:203-221) does, and checks
``oracle.starvector_oracle`` against them in float32.  ``fairscale`` (train-only import,
clip_model.py:10) is stubbed.  The installed transformers is 5.x (the reference pins 4.49.0); the
greedy path is cross-checked with an independent no-cache argmax loop below.

The GPU box has no /root/reference, so the outputs are committed as small fixtures
(tiny shapes, same op graph and head dims as StarVector-1B).
"""
from __future__ import annotations

import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import starvector_oracle as O
from oracle.hostinfo import host_cores  # noqa: E402

REF = os.environ.get("STARVECTOR_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")


def _import_reference():
    for n in ("fairscale", "fairscale.nn", "fairscale.nn.checkpoint"):
        sys.modules.setdefault(n, types.ModuleType(n))
    m = types.ModuleType("fairscale.nn.checkpoint.checkpoint_activations")
    m.checkpoint_wrapper = lambda mod, *a, **k: mod
    sys.modules[m.__name__] = m
    sys.path.insert(0, REF)
    from starvector.model.image_encoder.clip_model import VisionTransformer, LayerNorm
    from starvector.model.adapters.adapter import Adapter
    from transformers import GPTBigCodeConfig, GPTBigCodeForCausalLM
    return VisionTransformer, LayerNorm, Adapter, GPTBigCodeConfig, GPTBigCodeForCausalLM


def build_reference(cfg: O.OracleConfig, w):
    VisionTransformer, LayerNorm, Adapter, GPTBigCodeConfig, GPTBigCodeForCausalLM = _import_reference()
    vit = VisionTransformer(cfg.image_size, cfg.patch_size, cfg.vit_width, cfg.vit_layers, cfg.vit_heads, False)
    lnv = LayerNorm(cfg.vit_width)
    adp = Adapter(cfg.vit_width, cfg.hidden, adapter_norm=cfg.adapter_norm, query_length=cfg.query_length)
    hf_cfg = GPTBigCodeConfig(
        vocab_size=cfg.vocab, n_positions=cfg.n_positions, n_embd=cfg.hidden, n_layer=cfg.n_layer,
        n_head=cfg.n_head, n_inner=cfg.n_inner, multi_query=True,
        activation_function="gelu_pytorch_tanh", resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0,
        layer_norm_epsilon=cfg.ln_eps, bos_token_id=cfg.eos_token_id, eos_token_id=cfg.eos_token_id,
        pad_token_id=cfg.pad_token_id)
    lm = GPTBigCodeForCausalLM(hf_cfg)

    def sub(prefix):
        return {k[len(prefix):]: v for k, v in w.items() if k.startswith(prefix)}

    vit.load_state_dict(sub(O.P_VIT), strict=True)
    lnv.load_state_dict(sub(O.P_LNV), strict=True)
    missing = adp.load_state_dict(sub(O.P_ADP), strict=False)
    assert not [k for k in missing.missing_keys if "num_batches_tracked" not in k], missing
    sd = sub("model.svg_transformer.transformer.")
    res = lm.load_state_dict(sd, strict=False)
    assert not [k for k in res.missing_keys if "attn.bias" not in k and "masked_bias" not in k], res
    for m in (vit, lnv, adp, lm):
        m.eval()
    return vit, lnv, adp, lm


@torch.no_grad()
def reference_outputs(cfg, w, image, prompt_ids, n_new, stop_ids=None, repetition_penalty=1.0, num_beams=1,
                      length_penalty=1.0, early_stopping=None, min_length=1):
    vit, lnv, adp, lm = build_reference(cfg, w)
    enc = lnv(vit(image))                                   # image_encoder.py:92-94
    vis = adp(enc)                                          # starvector_base.py:209
    emb = torch.cat([vis, lm.transformer.wte(prompt_ids)], dim=1)   # :217-218
    mask = torch.ones(emb.shape[:2], dtype=torch.long)
    S0 = emb.shape[1]
    kw = dict(inputs_embeds=emb, attention_mask=mask, do_sample=False, num_beams=num_beams, top_p=None,
              temperature=None, max_length=S0 + n_new, min_length=min_length, repetition_penalty=repetition_penalty,
              length_penalty=length_penalty, use_cache=True, pad_token_id=cfg.pad_token_id)
    if early_stopping is not None:
        kw["early_stopping"] = early_stopping
    if stop_ids:
        from transformers.generation.stopping_criteria import StoppingCriteria, StoppingCriteriaList

        class StoppingCriteriaSub(StoppingCriteria):        # starvector_base.py:9-20 (restated)
            def __call__(self, input_ids, scores, **kwargs):
                return input_ids[0][-len(stop_ids):].tolist() == list(stop_ids)

        kw["stopping_criteria"] = StoppingCriteriaList([StoppingCriteriaSub()])
    toks = lm.generate(**kw)
    if num_beams > 1:
        return dict(tokens=toks, emb=emb)
    # independent no-cache loop: full forward each step + argmax (checks HF 5.x == 4.49 semantics)
    cur = emb
    nocache = []
    for _ in range(toks.shape[1]):
        lg = lm(inputs_embeds=cur, attention_mask=torch.ones(cur.shape[:2], dtype=torch.long)).logits[:, -1].float()
        nx = lg.argmax(-1)
        nocache.append(nx)
        cur = torch.cat([cur, lm.transformer.wte(nx).unsqueeze(1)], dim=1)
    nocache = torch.stack(nocache, 1)
    logits0 = lm(inputs_embeds=emb, attention_mask=mask).logits[:, -1].float()
    return dict(enc=enc, vis=vis, emb=emb, logits0=logits0, tokens=toks, tokens_nocache=nocache)


def build_reference_v2(cfg, w):
    """v2: the reference builds HF SigLIP (image_encoder.py:32-48, keeps .vision_model) and HF
    Starcoder2ForCausalLM (llm/starcoder2.py:22-27); both un-vendored transformers classes."""
    from transformers import SiglipVisionConfig, SiglipVisionModel, Starcoder2Config, Starcoder2ForCausalLM
    sys.path.insert(0, REF)
    for n in ("fairscale", "fairscale.nn", "fairscale.nn.checkpoint"):
        sys.modules.setdefault(n, types.ModuleType(n))
    from starvector.model.adapters.adapter import Adapter
    vc = SiglipVisionConfig(hidden_size=cfg.vit_width, intermediate_size=cfg.vit_mlp, num_hidden_layers=cfg.vit_layers,
                            num_attention_heads=cfg.vit_heads, image_size=cfg.image_size, patch_size=cfg.patch_size,
                            layer_norm_eps=cfg.vit_eps, hidden_act="gelu_pytorch_tanh")
    vm = SiglipVisionModel(vc)
    vm = getattr(vm, "vision_model", vm)
    miss = vm.load_state_dict({k[len(O.P_VIT):]: v for k, v in w.items() if k.startswith(O.P_VIT)}, strict=False)
    assert not [k for k in miss.missing_keys if not k.startswith("head.")], miss
    adp = Adapter(cfg.vit_width, cfg.hidden, adapter_norm=cfg.adapter_norm, query_length=cfg.query_length)
    adp.load_state_dict({k[len(O.P_ADP):]: v for k, v in w.items() if k.startswith(O.P_ADP)}, strict=True)
    sc = Starcoder2Config(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.n_inner,
                          num_hidden_layers=cfg.n_layer, num_attention_heads=cfg.n_head,
                          num_key_value_heads=cfg.n_kv_head, hidden_act="gelu_pytorch_tanh",
                          max_position_embeddings=cfg.n_positions, norm_epsilon=cfg.ln_eps, rope_theta=cfg.rope_theta,
                          sliding_window=cfg.sliding_window or 4096, use_bias=True, tie_word_embeddings=True, residual_dropout=0.0,
                          embedding_dropout=0.0, attention_dropout=0.0, bos_token_id=cfg.eos_token_id,
                          eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id)
    lm = Starcoder2ForCausalLM(sc)
    pre = "model.svg_transformer.transformer."
    lm.load_state_dict({k[len(pre):]: v for k, v in w.items() if k.startswith(pre)}, strict=True)
    for m in (vm, adp, lm):
        m.eval()
    return vm, adp, lm


@torch.no_grad()
def run_case_v2(tag, cfg, seed, batch, n_new, write):
    print(f"[{tag}] cfg={cfg}")
    w = O.make_weights(cfg, seed=seed)
    image = O.synthetic_images(batch, cfg.image_size, seed=seed + 1)
    prompt_ids = torch.tensor([[7, 11]] * batch, dtype=torch.long)
    vm, adp, lm = build_reference_v2(cfg, w)
    r_enc = vm(image)["last_hidden_state"]                       # image_encoder.py:108-109
    r_vis = adp(r_enc)
    r_emb = torch.cat([r_vis, lm.model.embed_tokens(prompt_ids)], dim=1)      # starvector_v2.py:45-47
    mask = torch.ones(r_emb.shape[:2], dtype=torch.long)
    r_logits0 = lm(inputs_embeds=r_emb, attention_mask=mask).logits[:, -1].float()
    r_toks = lm.generate(inputs_embeds=r_emb, attention_mask=mask, do_sample=False, num_beams=1, top_p=None,
                         temperature=None, max_length=r_emb.shape[1] + n_new, use_cache=True,
                         pad_token_id=cfg.pad_token_id)
    enc = O.image_encoder_forward(w, cfg, image)
    vis = O.adapter_forward(w, cfg, enc)
    emb = O.prepare_generation_inputs(w, cfg, image, prompt_ids)
    logits0, _ = O.decoder_prefill(w, cfg, emb)
    toks = O.greedy_generate(w, cfg, emb, emb.shape[1] + n_new)
    check("siglip tower (a13)", enc, r_enc, 2e-5)
    check("adapter (a6)", vis, r_vis, 2e-5)
    check("inputs_embeds", emb, r_emb, 2e-5)
    check("starcoder2 prefill logits (a13)", logits0, r_logits0, 5e-5)
    n = r_toks.shape[1]
    print(f"  greedy tokens == HF generate: {torch.equal(toks[:, :n], r_toks)}; N={n}")
    assert torch.equal(toks[:, :n], r_toks) and toks.shape[1] == n
    if write:
        from safetensors.torch import save_file
        save_file({"image": image, "prompt_ids": prompt_ids, "enc": r_enc.contiguous(), "vis": r_vis.contiguous(),
                   "emb": r_emb.contiguous(), "logits0": r_logits0.contiguous(), "tokens": r_toks.contiguous(),
                   "meta": torch.tensor([seed, batch, n_new], dtype=torch.long)},
                  os.path.join(GOLD, f"{tag}.safetensors"))
        print(f"  wrote tests/golden/{tag}.safetensors")


def check(name, a, b, tol):
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    ok = err <= tol * max(1.0, ref)
    print(f"  {name:28s} max|diff|={err:.3e}  (ref max {ref:.3e})  {'OK' if ok else 'FAIL'}")
    assert ok, name


def run_case(tag, cfg, seed, batch, n_new, write):
    print(f"[{tag}] cfg={cfg}")
    w = O.make_weights(cfg, seed=seed)
    image = O.synthetic_images(batch, cfg.image_size, seed=seed + 1)
    prompt_ids = torch.tensor([[7, 11]] * batch, dtype=torch.long)
    ref = reference_outputs(cfg, w, image, prompt_ids, n_new)
    enc = O.image_encoder_forward(w, cfg, image)
    vis = O.adapter_forward(w, cfg, enc)
    emb = O.prepare_generation_inputs(w, cfg, image, prompt_ids)
    logits0, _ = O.decoder_prefill(w, cfg, emb)
    toks, step_logits = O.greedy_generate(w, cfg, emb, emb.shape[1] + n_new, return_logits=True)
    check("image_encoder (a2-a5)", enc, ref["enc"], 2e-5)
    check("adapter (a6)", vis, ref["vis"], 2e-5)
    check("inputs_embeds (a1,a7)", emb, ref["emb"], 2e-5)
    check("prefill logits (a8-a10)", logits0, ref["logits0"], 5e-5)
    # tokens: the reference may stop early on EOS; compare the common prefix HF returned
    n = ref["tokens"].shape[1]
    same = torch.equal(toks[:, :n], ref["tokens"])
    same_nc = torch.equal(ref["tokens"], ref["tokens_nocache"]) or bool((ref["tokens"] == cfg.pad_token_id).any())
    top2 = step_logits.topk(2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1]).min().item()
    print(f"  greedy tokens == HF generate: {same} ; HF generate == no-cache loop: {same_nc} ; "
          f"min top1-top2 margin {margin:.3e}; N={n}")
    assert same and toks.shape[1] == n
    if write:
        from safetensors.torch import save_file
        os.makedirs(GOLD, exist_ok=True)
        save_file({
            "image": image, "prompt_ids": prompt_ids, "enc": ref["enc"].contiguous(),
            "vis": ref["vis"].contiguous(), "emb": ref["emb"].contiguous(),
            "logits0": ref["logits0"].contiguous(), "tokens": ref["tokens"].contiguous(),
            "meta": torch.tensor([seed, batch, n_new], dtype=torch.long),
        }, os.path.join(GOLD, f"{tag}.safetensors"))
        print(f"  wrote tests/golden/{tag}.safetensors")


def designed_targets(batch, n_new, seed, lo=12, hi=500, palette=48, avoid=()):
    """A diverse token stream per row: ids from a seeded palette, no immediate repeats."""
    g = torch.Generator().manual_seed(seed)
    ids = [int(i) + lo for i in torch.randperm(hi - lo, generator=g) if int(i) + lo not in avoid][:palette]
    out = torch.zeros(batch, n_new, dtype=torch.long)
    for b in range(batch):
        prev = -1
        for t in range(n_new):
            c = prev
            while c == prev:
                c = ids[int(torch.randint(0, len(ids), (1,), generator=g))]
            out[b, t] = c
            prev = c
    return out


def fit_embedding(cfg, w, image, prompt_ids, targets, mask=None, steps=400, want=0.25, lr=3e-3):
    """Fit the tied embedding table (wte = lm_head) so that `targets` [B, n] is the GREEDY stream after the prompt with a
    top-1/top-2 margin of `want` x the logit scale (teacher-forced hinge loss through the oracle's own float32 forward; the
    other 99 % of the weights stay the seeded random init).  mask [B, n] (bool) selects the positions that are constrained.
    Why: a random-init transformer either repeats one token or decides by near-ties that no two bf16 implementations
    resolve alike; with a fitted table the integer token stream is a hard parity assertion.  Returns bf16-exact float32."""
    key = O.embed_key(cfg)
    base = w[key].clone()
    delta = torch.zeros_like(base, requires_grad=True)
    opt = torch.optim.Adam([delta], lr=lr)
    with torch.no_grad():
        vis = O.adapter_forward(w, cfg, O.image_encoder_forward(w, cfg, image))
    n = targets.shape[1]
    mk = torch.ones_like(targets, dtype=torch.bool) if mask is None else mask
    onehot = torch.nn.functional.one_hot(targets, cfg.vocab).bool()
    for _ in range(steps):
        wt = base + delta
        ww = dict(w)
        ww[key] = wt
        ww[O.K_LMH] = wt
        ids = torch.cat([prompt_ids, targets[:, :-1]], 1)
        lg = O.decoder_forward_logits(ww, cfg, torch.cat([vis, wt[ids]], 1), n)
        scale = lg.detach().abs().max()
        margin = lg.gather(-1, targets[..., None]).squeeze(-1) - lg.masked_fill(onehot, -1e9).max(-1).values
        loss = torch.relu(want * scale - margin)[mk].mean() + 1e-3 * (delta ** 2).sum()
        opt.zero_grad()
        loss.backward()
        opt.step()
    return (base + delta.detach()).to(torch.bfloat16).to(torch.float32)


def _margins(w, cfg, emb, n_new, mode, mask=None, **kw):
    toks, lg = O.greedy_generate(w, cfg, emb, emb.shape[1] + n_new, mode=mode, return_logits=True, **kw)
    top2 = lg.topk(2, -1).values
    rel = (top2[..., 0] - top2[..., 1]) / lg.abs().max()
    if mask is not None:                                     # positions of finished rows emit pad whatever their logits say
        rel = rel[mask[:, :rel.shape[1]]]
    return toks, float(rel.min())


def run_fitted_case(tag, cfg, seed, batch, n_new, write):
    """tiny_b3: encoder / adapter / prefill against the reference modules as before, and a DESIGNED greedy stream (fitted
    embedding table, see fit_embedding) that HF generate, the no-cache loop and the oracle all reproduce token for token."""
    import dataclasses
    print(f"[{tag}] cfg={cfg}")
    w = O.make_weights(cfg, seed=seed)
    image = O.synthetic_images(batch, cfg.image_size, seed=seed + 1)
    prompt_ids = torch.tensor([[7, 11]] * batch, dtype=torch.long)
    targets = designed_targets(batch, n_new, seed + 7, avoid=(cfg.eos_token_id, cfg.pad_token_id, 7, 11))
    wte = fit_embedding(cfg, w, image, prompt_ids, targets)
    w = O.apply_fixture_weights(w, cfg, {"wte": wte})
    ref = reference_outputs(cfg, w, image, prompt_ids, n_new)
    enc = O.image_encoder_forward(w, cfg, image)
    vis = O.adapter_forward(w, cfg, enc)
    emb = O.prepare_generation_inputs(w, cfg, image, prompt_ids)
    logits0, _ = O.decoder_prefill(w, cfg, emb)
    check("image_encoder (a2-a5)", enc, ref["enc"], 2e-5)
    check("adapter (a6)", vis, ref["vis"], 2e-5)
    check("inputs_embeds (a1,a7)", emb, ref["emb"], 2e-5)
    check("prefill logits (a8-a10)", logits0, ref["logits0"], 5e-5)
    toks, m32 = _margins(w, cfg, emb, n_new, "fp32")
    tb, m16 = _margins(w, cfg, O.prepare_generation_inputs(w, cfg, image, prompt_ids, "bf16"), n_new, "bf16")
    same = torch.equal(toks, ref["tokens"]) and torch.equal(toks, targets) and torch.equal(tb, targets)
    print(f"  greedy tokens == HF generate == the designed stream (fp32 and bf16 oracle): {same}; HF generate == no-cache loop: "
          f"{torch.equal(ref['tokens'], ref['tokens_nocache'])}; min top1-top2 margin / logit scale: fp32 {m32:.3f}, bf16 {m16:.3f}; "
          f"{len(set(toks.flatten().tolist()))} distinct tokens in {toks.numel()} positions")
    assert same and torch.equal(ref["tokens"], ref["tokens_nocache"]) and min(m32, m16) >= 0.1
    if write:
        from safetensors.torch import save_file
        os.makedirs(GOLD, exist_ok=True)
        save_file({
            "image": image, "prompt_ids": prompt_ids, "enc": ref["enc"].contiguous(),
            "vis": ref["vis"].contiguous(), "emb": ref["emb"].contiguous(),
            "logits0": ref["logits0"].contiguous(), "tokens": ref["tokens"].contiguous(),
            "wte": wte.to(torch.bfloat16).contiguous(),
            "meta": torch.tensor([seed, batch, n_new], dtype=torch.long),
        }, os.path.join(GOLD, f"{tag}.safetensors"))
        print(f"  wrote tests/golden/{tag}.safetensors")


def run_stop_case(write):
    """EOS / pad / row-0 stop-sequence semantics (a11) on the tiny config, with a designed stream (fit_embedding): row 0 emits
    the stop pair at steps 9-10 (first occurrence), row 1 emits EOS at step 4 and is padded afterwards, row 2 runs on; HF
    generate + the reference's StoppingCriteriaSub end the WHOLE batch after step 10 of a budget of 24."""
    import dataclasses
    cfg = O.OracleConfig.tiny()
    seed, B, budget, s_end, e_at = 77, 3, 24, 10, 4
    eos, stop_ids = 401, [402, 403]
    cfg2 = dataclasses.replace(cfg, eos_token_id=eos)
    w = O.make_weights(cfg2, seed=seed)
    image = O.synthetic_images(B, cfg.image_size, seed=seed + 1)
    prompt_ids = torch.tensor([[7, 11]] * B, dtype=torch.long)
    n = s_end + 1
    targets = designed_targets(B, n, seed + 7, avoid=(eos, cfg.pad_token_id, 7, 11, *stop_ids))
    targets[0, s_end - 1], targets[0, s_end] = stop_ids[0], stop_ids[1]
    targets[1, e_at] = eos
    targets[1, e_at + 1:] = cfg.pad_token_id                 # what HF feeds a finished row; unconstrained outputs
    mask = torch.ones(B, n, dtype=torch.bool)
    mask[1, e_at + 1:] = False
    wte = fit_embedding(cfg2, w, image, prompt_ids, targets, mask)
    w = O.apply_fixture_weights(w, cfg2, {"wte": wte})
    emb = O.prepare_generation_inputs(w, cfg2, image, prompt_ids)
    ref = reference_outputs(cfg2, w, image, prompt_ids, budget, stop_ids=stop_ids)
    mine = O.greedy_generate(w, cfg2, emb, emb.shape[1] + budget, stop_ids=stop_ids)
    mine16, m16 = _margins(w, cfg2, O.prepare_generation_inputs(w, cfg2, image, prompt_ids, "bf16"), budget, "bf16", mask=mask,
                           stop_ids=stop_ids)
    print(f"[tiny_stop] eos={eos} stop={stop_ids} ref shape {tuple(ref['tokens'].shape)} mine {tuple(mine.shape)}; "
          f"min margin / scale over the bf16 run {m16:.3f}")
    assert torch.equal(mine, ref["tokens"]) and torch.equal(mine16, mine), (mine, ref["tokens"])
    assert torch.equal(mine, targets)
    assert (mine == cfg.pad_token_id).any(), "case must exercise pad-after-EOS"
    assert mine.shape[1] == s_end + 1 < budget, "case must exercise the row-0 stop"
    assert m16 >= 0.1
    if write:
        from safetensors.torch import save_file
        save_file({"image": image, "prompt_ids": prompt_ids, "tokens": ref["tokens"].contiguous(),
                   "stop_ids": torch.tensor(stop_ids), "wte": wte.to(torch.bfloat16).contiguous(),
                   "meta": torch.tensor([seed, B, budget, eos])},
                  os.path.join(GOLD, "tiny_stop.safetensors"))
        print("  wrote tests/golden/tiny_stop.safetensors")


def run_reppen_case(write):
    """repetition_penalty (starvector_base.py:237; quickstart uses 3.1) through HF's RepetitionPenaltyLogitsProcessor."""
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=88)
    B, n_new, pen = 2, 20, 1.7
    image = O.synthetic_images(B, cfg.image_size, seed=89)
    prompt_ids = torch.tensor([[7, 11]] * B, dtype=torch.long)
    ref = reference_outputs(cfg, w, image, prompt_ids, n_new, repetition_penalty=pen)
    emb = O.prepare_generation_inputs(w, cfg, image, prompt_ids)
    mine = O.greedy_generate(w, cfg, emb, emb.shape[1] + n_new, repetition_penalty=pen)
    free = O.greedy_generate(w, cfg, emb, emb.shape[1] + n_new)
    print(f"[tiny_reppen] tokens == HF: {torch.equal(mine, ref['tokens'])}; differs from penalty-free run: {not torch.equal(mine, free)}")
    assert torch.equal(mine, ref["tokens"]) and not torch.equal(mine, free)
    if write:
        from safetensors.torch import save_file
        save_file({"image": image, "prompt_ids": prompt_ids, "tokens": ref["tokens"].contiguous(),
                   "meta": torch.tensor([88, B, n_new]), "penalty": torch.tensor([pen])},
                  os.path.join(GOLD, "tiny_reppen.safetensors"))
        print("  wrote tests/golden/tiny_reppen.safetensors")


def run_minlen_case(write):
    """min_length (starvector_base.py:236) beyond the prompt length: HF subtracts the prompt length, then keeps EOS at -inf
    for the first min_length - S0 generated tokens.  EOS is a token row 1 reaches at step 2 of the free run."""
    import dataclasses
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=77)
    B, n_new = 3, 16
    image = O.synthetic_images(B, cfg.image_size, seed=78)
    prompt_ids = torch.tensor([[7, 11]] * B, dtype=torch.long)
    emb = O.prepare_generation_inputs(w, cfg, image, prompt_ids)
    S0 = emb.shape[1]
    free = O.greedy_generate(w, dataclasses.replace(cfg, eos_token_id=-1), emb, S0 + n_new)
    eos = int(free[1, 2])
    cfg2 = dataclasses.replace(cfg, eos_token_id=eos)
    out = {"image": image, "prompt_ids": prompt_ids}
    seen = []
    for extra in (0, 3, 6):                                 # min_length = S0 + extra
        ref = reference_outputs(cfg2, w, image, prompt_ids, n_new, min_length=S0 + extra)
        mine = O.greedy_generate(w, cfg2, emb, S0 + n_new, min_length=S0 + extra)
        assert torch.equal(mine, ref["tokens"]), (extra, mine, ref["tokens"])
        first_eos = [int((r == eos).nonzero()[0]) if bool((r == eos).any()) else -1 for r in mine]
        assert all(f < 0 or f >= extra for f in first_eos)
        print(f"[tiny_minlen] min_length = S0 + {extra}: tokens == HF, first EOS per row {first_eos}")
        out[f"tokens_{extra}"] = ref["tokens"].contiguous()
        seen.append(mine)
        # the same remainder under beam search (HF applies the processor to the log-probabilities there)
        refb = reference_outputs(cfg2, w, image, prompt_ids, n_new, min_length=S0 + extra, num_beams=2, early_stopping=True)
        mineb = O.beam_search_generate(w, cfg2, emb, S0 + n_new, 2, early_stopping=True, min_length=S0 + extra)
        assert torch.equal(mineb, refb["tokens"]), (extra, mineb, refb["tokens"])
        print(f"[tiny_minlen] min_length = S0 + {extra}, num_beams 2: tokens == HF, shape {tuple(mineb.shape)}")
        out[f"beam2_tokens_{extra}"] = refb["tokens"].contiguous()
    assert not torch.equal(seen[0][:, :8], seen[1][:, :8]), "the case must exercise the suppression"
    if write:
        from safetensors.torch import save_file
        out["meta"] = torch.tensor([77, B, n_new, eos, S0])
        save_file(out, os.path.join(GOLD, "tiny_minlen.safetensors"))
        print("  wrote tests/golden/tiny_minlen.safetensors")


def run_beam_cases(write):
    """num_beams > 1 (the reference's default is 2, starvector_base.py:234) through HF's beam search: early_stopping
    True (v1 im2svg, :293) and False (v2), length penalties, an EOS some hypotheses reach, the row-0 stop sequence and
    the repetition penalty on log-probs."""
    import dataclasses
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=99)
    B, n_new = 3, 16
    image = O.synthetic_images(B, cfg.image_size, seed=98)
    prompt_ids = torch.tensor([[7, 11]] * B, dtype=torch.long)
    emb = O.prepare_generation_inputs(w, cfg, image, prompt_ids)
    free = O.greedy_generate(w, cfg, emb, emb.shape[1] + n_new)
    eos = int(free[1, 5])                                  # a token the greedy path of request 1 reaches
    r0 = free[0].tolist()
    cases = [
        dict(tag="nb2_es", nb=2, lp=1.0, es=True, eos=eos, stop=None, pen=1.0),
        dict(tag="nb3_noes", nb=3, lp=0.6, es=False, eos=eos, stop=None, pen=1.0),
        dict(tag="nb2_stop", nb=2, lp=1.0, es=True, eos=cfg.eos_token_id, stop=None, pen=1.0),   # stop filled below
        dict(tag="nb4_pen", nb=4, lp=1.3, es=True, eos=eos, stop=None, pen=1.5),
        dict(tag="nb2_never", nb=2, lp=1.0, es="never", eos=eos, stop=None, pen=1.0),
    ]
    out = {"image": image, "prompt_ids": prompt_ids, "meta": torch.tensor([99, B, n_new])}
    for c in cases:
        cfg2 = dataclasses.replace(cfg, eos_token_id=c["eos"])
        if c["tag"] == "nb2_stop":
            base = O.beam_search_generate(w, cfg2, emb, emb.shape[1] + n_new, 2)
            c["stop"] = base[0, 6:8].tolist()              # a pair the best beam of request 0 emits
        ref = reference_outputs(cfg2, w, image, prompt_ids, n_new, stop_ids=c["stop"], repetition_penalty=c["pen"],
                                num_beams=c["nb"], length_penalty=c["lp"], early_stopping=c["es"])["tokens"]
        mine = O.beam_search_generate(w, cfg2, emb, emb.shape[1] + n_new, c["nb"], length_penalty=c["lp"],
                                      early_stopping=c["es"], stop_ids=c["stop"], repetition_penalty=c["pen"])
        same = mine.shape == ref.shape and torch.equal(mine, ref)
        print(f"[tiny_beam/{c['tag']}] HF {tuple(ref.shape)} mine {tuple(mine.shape)} equal: {same}; "
              f"differs from greedy: {not torch.equal(mine[:, :n_new], free[:, :mine.shape[1]])}")
        assert same, (mine, ref)
        es_code = {True: 1, False: 0, "never": 2}[c["es"]]
        out[c["tag"] + ".tokens"] = ref.contiguous()
        out[c["tag"] + ".params"] = torch.tensor([c["nb"], c["lp"], es_code, c["eos"], c["pen"]], dtype=torch.float64)
        out[c["tag"] + ".stop"] = torch.tensor(c["stop"] if c["stop"] else [], dtype=torch.long)
    if write:
        from safetensors.torch import save_file
        save_file(out, os.path.join(GOLD, "tiny_beam.safetensors"))
        print("  wrote tests/golden/tiny_beam.safetensors")


def run_sampling_cases(write):
    """do_sample paths.  (1) The warpers: temperature -> top-k -> top-p against HF's own warper classes (the reference
    never passes top_k, but its pinned transformers==4.49.0 defaults it to 50).  (2) beam-sample (the reference's
    validation-time generation: num_beams 3 + nucleus sampling, configs/models/starvector-8b/im2svg-stack.yaml:75-81):
    with the same torch seed the restated loop draws exactly what HF generate draws."""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    g = torch.Generator().manual_seed(0)
    for (T, tp, tk, mk) in [(0.7, 0.9, 50, 1), (1.0, 0.9, 50, 2), (1.3, 0.5, 5, 2), (1.0, 1.0, 50, 1), (0.8, 0.95, 0, 1),
                            (1.0, 0.01, 50, 2)]:
        lg = 3 * torch.randn(6, 300, generator=g)
        ref = lg.clone()
        if T != 1.0:
            ref = TemperatureLogitsWarper(T)(None, ref)
        if tk:
            ref = TopKLogitsWarper(top_k=tk, min_tokens_to_keep=mk)(None, ref)
        if tp < 1:
            ref = TopPLogitsWarper(top_p=tp, min_tokens_to_keep=mk)(None, ref)
        assert torch.equal(O.warp_scores(lg, T, tp, tk, mk), ref), (T, tp, tk, mk)
    print("[warpers] temperature/top-k/top-p == HF warper classes on 6 settings")
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=99)
    B, n_new = 3, 12
    image = O.synthetic_images(B, cfg.image_size, seed=98)
    prompt_ids = torch.tensor([[7, 11]] * B, dtype=torch.long)
    emb = O.prepare_generation_inputs(w, cfg, image, prompt_ids)
    _, _, _, lm = build_reference(cfg, w)
    out = {"image": image, "prompt_ids": prompt_ids, "meta": torch.tensor([99, B, n_new, 1234])}
    for i, (nb, T, tp, tk, es, lp) in enumerate([(3, 1.0, 0.9, 50, False, 0.5), (2, 0.7, 0.95, 50, True, 1.0),
                                                 (3, 1.0, 1.0, 0, True, 1.0)]):
        kw = dict(inputs_embeds=emb, attention_mask=torch.ones(emb.shape[:2], dtype=torch.long), do_sample=True,
                  num_beams=nb, top_p=tp, temperature=T, top_k=tk if tk else None, max_length=emb.shape[1] + n_new,
                  min_length=1, repetition_penalty=1.0, length_penalty=lp, use_cache=True,
                  pad_token_id=cfg.pad_token_id, early_stopping=es)
        torch.manual_seed(1234)
        ref = lm.generate(**kw)
        torch.manual_seed(1234)
        mine = O.beam_search_generate(w, cfg, emb, emb.shape[1] + n_new, nb, length_penalty=lp, early_stopping=es,
                                      do_sample=True, temperature=T, top_p=tp, top_k=tk)
        same = ref.shape == mine.shape and torch.equal(ref, mine)
        print(f"[tiny_beam_sample/{i}] nb={nb} T={T} top_p={tp} top_k={tk}: HF {tuple(ref.shape)} equal: {same}")
        assert same
        out[f"case{i}.tokens"] = ref.contiguous()
        out[f"case{i}.params"] = torch.tensor([nb, T, tp, tk, {True: 1, False: 0}[es], lp], dtype=torch.float64)
    if write:
        from safetensors.torch import save_file
        save_file(out, os.path.join(GOLD, "tiny_beam_sample.safetensors"))
        print("  wrote tests/golden/tiny_beam_sample.safetensors")


def run_window_case(write):
    """StarCoder2 sliding window (bigcode/starcoder2-7b: 4096; here 24 so that 40 new tokens leave it): HF's mask
    `kv > q - W` on the un-cropped cache."""
    import dataclasses
    cfg = dataclasses.replace(O.OracleConfig.tiny_v2(), sliding_window=24)
    w = O.make_weights(cfg, seed=2024)
    _, _, lm = build_reference_v2(cfg, w)
    B, n_new = 2, 40
    image = O.synthetic_images(B, cfg.image_size, seed=5)
    prompt_ids = torch.tensor([[7, 11]] * B, dtype=torch.long)
    emb = O.prepare_generation_inputs(w, cfg, image, prompt_ids)
    ref = lm.generate(inputs_embeds=emb, attention_mask=torch.ones(emb.shape[:2], dtype=torch.long), do_sample=False,
                      num_beams=1, max_length=emb.shape[1] + n_new, use_cache=True, pad_token_id=cfg.pad_token_id,
                      eos_token_id=None)
    free = dataclasses.replace(cfg, eos_token_id=-1)
    mine = O.greedy_generate(w, free, emb, emb.shape[1] + n_new)
    nowin = O.greedy_generate(w, dataclasses.replace(free, sliding_window=0), emb, emb.shape[1] + n_new)
    print(f"[tiny_v2_window] W=24, S0={emb.shape[1]}, {n_new} new tokens: tokens == HF {torch.equal(ref, mine)}; "
          f"differs from full attention: {not torch.equal(mine, nowin)}")
    assert torch.equal(ref, mine) and not torch.equal(mine, nowin)
    # a prompt LONGER than the window (W = 8 < S0): HF applies the same mask inside the prompt pass; the engine's windowed
    # prompt pass (attention.hip: masked + tile-skipping flash prefill) is checked against this oracle path
    W8, n8 = 8, 24
    cfg8 = dataclasses.replace(O.OracleConfig.tiny_v2(), sliding_window=W8)
    _, _, lm8 = build_reference_v2(cfg8, w)
    ref8 = lm8.generate(inputs_embeds=emb, attention_mask=torch.ones(emb.shape[:2], dtype=torch.long), do_sample=False,
                        num_beams=1, max_length=emb.shape[1] + n8, use_cache=True, pad_token_id=cfg.pad_token_id,
                        eos_token_id=None)
    mine8 = O.greedy_generate(w, dataclasses.replace(cfg8, eos_token_id=-1), emb, emb.shape[1] + n8)
    lg_ref8 = lm8(inputs_embeds=emb, attention_mask=torch.ones(emb.shape[:2], dtype=torch.long)).logits[:, -1].detach()
    lg_mine8 = O.greedy_generate(w, dataclasses.replace(cfg8, eos_token_id=-1), emb, emb.shape[1] + 1, return_logits=True)[1][:, 0]
    d8 = float((lg_ref8 - lg_mine8).abs().max())
    print(f"[tiny_v2_window] W={W8} < S0={emb.shape[1]} (windowed prompt pass): prefill logits max|diff| {d8:.2e}; "
          f"{n8} tokens == HF {torch.equal(ref8, mine8)}; differs from W=24: {not torch.equal(mine8, mine[:, :n8])}")
    assert emb.shape[1] > W8 and d8 < 1e-4 and torch.equal(ref8, mine8)
    if write:
        from safetensors.torch import save_file
        save_file({"image": image, "prompt_ids": prompt_ids, "tokens": ref.contiguous(), "tokens_w8": ref8.contiguous(),
                   "meta": torch.tensor([2024, B, n_new, 24])}, os.path.join(GOLD, "tiny_v2_window.safetensors"))
        print("  wrote tests/golden/tiny_v2_window.safetensors")


def run_forward_case(write):
    """StarVectorForCausalLM.forward (starvector_arch.py:161-184): logits of the last num_logits_to_keep positions of
    [visual prefix | completion ids] -- against HF GPTBigCodeForCausalLM(inputs_embeds).logits."""
    cfg = O.OracleConfig.tiny()
    w = O.make_weights(cfg, seed=1234)
    _, _, _, lm = build_reference(cfg, w)
    image = O.synthetic_images(2, cfg.image_size, seed=3)
    ids = torch.randint(0, 500, (2, 9), generator=torch.Generator().manual_seed(4))
    emb = O.prepare_generation_inputs(w, cfg, image, ids)
    ref = lm(inputs_embeds=emb, attention_mask=torch.ones(emb.shape[:2], dtype=torch.long)).logits
    for n in (0, 5, 1):
        mine = O.decoder_forward_logits(w, cfg, emb, n)
        check(f"forward logits, keep {n or 'all'}", mine, ref if n == 0 else ref[:, -n:], 1e-5)
    # right-padded mask (completions padded after EOS, a GRPO trainer's call): the real positions of HF's masked run equal the
    # unmasked run's -- what the mirror's forward relies on (model.py StarVectorForCausalLM.forward)
    mask = torch.ones(emb.shape[:2], dtype=torch.long)
    mask[1, -3:] = 0
    masked = lm(inputs_embeds=emb, attention_mask=mask).logits.detach()
    d_real = float((masked - ref.detach())[mask.bool()].abs().max())
    d_pad = float((masked - ref.detach())[~mask.bool()].abs().max())
    print(f"[tiny_forward] right-padded mask: max|diff| at real positions {d_real:.2e} (at padded ones {d_pad:.2e}: unspecified)")
    assert d_real < 1e-5
    # left-padded mask: the reference's pinned transformers (4.49; the vendored snapshot gpt_bigcode/modeling_gpt_bigcode.py:980-983)
    # masks the padded keys and numbers positions by cumsum(mask) - 1 inside the model's forward -> a row's real positions equal the
    # row scored ALONE without its pads (what the mirror's forward does for such rows), and an unpadded row is untouched.  The
    # transformers installed here (5.x) builds that numbering in generate() only, so the 4.49 rule is passed in as position_ids.
    lmask = torch.ones(emb.shape[:2], dtype=torch.long)
    lmask[0, :2] = 0
    pos = lmask.cumsum(-1) - 1
    pos.masked_fill_(lmask == 0, 1)
    lpad = lm(inputs_embeds=emb, attention_mask=lmask, position_ids=pos).logits.detach()
    alone = O.decoder_forward_logits(w, cfg, emb[:1, 2:], 0)
    d_left = float((lpad[0, 2:] - alone[0]).abs().max())
    d_other = float((lpad[1] - ref.detach()[1]).abs().max())
    print(f"[tiny_forward] left-padded mask: max|HF masked row - oracle on the row without pads| {d_left:.2e}; "
          f"unpadded row vs unmasked run {d_other:.2e}")
    assert d_left < 1e-5 and d_other < 1e-5
    if write:
        from safetensors.torch import save_file
        save_file({"image": image, "ids": ids, "logits_keep5": ref[:, -5:].contiguous(), "meta": torch.tensor([1234, 2, 9]),
                   "logits_leftpad2_row0_keep5": lpad[0, -5:].contiguous()},
                  os.path.join(GOLD, "tiny_forward.safetensors"))
        print("  wrote tests/golden/tiny_forward.safetensors")


def run_padding_case():
    """Padded prompts (text2svg with captions of different lengths): HF masks the padded keys and numbers positions by
    cumsum(attention_mask), so generating a padded batch equals generating every row alone with its padding removed --
    for left padding (the v2 tokenizer, llm/starcoder2.py:53), for right padding before the trigger token (the v1 default),
    GPTBigCode and StarCoder2.  The mirror relies on this to serve padded batches group by group (model.py::_generate_padded)."""
    def run(lm, wte, cfg, side):
        lens, n_new = [5, 9, 7], 10
        S, B = max(lens) + 1, len(lens)
        g = torch.Generator().manual_seed(3)
        ids = [torch.randint(1, 500, (n,), generator=g) for n in lens]
        trig = torch.tensor([501])
        full = torch.zeros(B, S, dtype=torch.long)
        mask = torch.zeros(B, S, dtype=torch.long)
        for b, t in enumerate(ids):
            n = len(t)
            if side == "left":
                full[b, S - 1 - n:S - 1] = t
                mask[b, S - 1 - n:] = 1
            else:
                full[b, :n] = t
                mask[b, :n] = 1
                mask[b, S - 1] = 1
            full[b, S - 1] = trig
        kw = dict(do_sample=False, num_beams=1, max_length=S + n_new, use_cache=True, pad_token_id=0, eos_token_id=None)
        ref = lm.generate(inputs_embeds=wte(full), attention_mask=mask, **kw)
        for b, t in enumerate(ids):
            e = wte(torch.cat([t, trig])[None])
            solo = lm.generate(inputs_embeds=e, attention_mask=torch.ones(1, e.shape[1], dtype=torch.long), **kw)
            assert torch.equal(ref[b, :n_new], solo[0, :n_new]), (side, b)
    cfg = O.OracleConfig.tiny()
    _, _, _, lm = build_reference(cfg, O.make_weights(cfg, seed=5))
    run(lm, lm.transformer.wte, cfg, "left")
    run(lm, lm.transformer.wte, cfg, "right")
    cfg2 = O.OracleConfig.tiny_v2()
    _, _, lm2 = build_reference_v2(cfg2, O.make_weights(cfg2, seed=5))
    run(lm2, lm2.model.embed_tokens, cfg2, "left")
    print("[padding] HF generate on a padded batch == every row alone without its padding (GPTBigCode left/right, StarCoder2 left)")


def main():
    write = "--no-write" not in sys.argv
    torch.manual_seed(0)
    torch.set_num_threads(host_cores())
    if "--only-window" in sys.argv:                 # re-mint tests/golden/tiny_v2_window.safetensors alone
        run_window_case(write)
        return
    if "--only-forward" in sys.argv:                # re-mint tests/golden/tiny_forward.safetensors alone
        run_forward_case(write)
        return
    run_fitted_case("tiny_b3", O.OracleConfig.tiny(), seed=1234, batch=3, n_new=24, write=write)
    import dataclasses
    run_case("tiny_bn_b2", dataclasses.replace(O.OracleConfig.tiny(), adapter_norm="batch_norm"),
             seed=4321, batch=2, n_new=8, write=write)
    run_stop_case(write)
    run_reppen_case(write)
    run_minlen_case(write)
    run_beam_cases(write)
    run_sampling_cases(write)
    run_forward_case(write)
    run_padding_case()
    run_case_v2("tiny_v2_b2", O.OracleConfig.tiny_v2(), seed=2024, batch=2, n_new=12, write=write)
    run_window_case(write)
    if "--full" in sys.argv:
        # StarVector-1B shapes, 1 image, a few tokens: validates the restatement at BASELINE
        # config 1 (too large to commit; run on demand)
        run_case("full_1b_b1", O.OracleConfig(), seed=1234, batch=1, n_new=4, write=False)
    print("oracle pinned against the reference: OK")


if __name__ == "__main__":
    main()
