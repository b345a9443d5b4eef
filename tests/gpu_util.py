"""Helpers shared by the -m gpu tests (engine construction from an oracle config)."""
import torch

import starvector_amd as sva
from oracle import starvector_oracle as O


def dev():
    return torch.device("cuda", 0)


def bf(x):
    return x.to(torch.bfloat16).to(dev())


def build_engine(cfg: O.OracleConfig, w, max_batch, max_seq_len, weight_dtype="bf16", exclusive_device=False):
    ec = sva.EngineConfig(image_size=cfg.image_size, patch_size=cfg.patch_size, vit_width=cfg.vit_width,
                          vit_layers=cfg.vit_layers, vit_heads=cfg.vit_heads, adapter_norm=cfg.adapter_norm,
                          hidden=cfg.hidden, n_layer=cfg.n_layer, n_head=cfg.n_head, n_inner=cfg.n_inner,
                          vocab=cfg.vocab, n_positions=cfg.n_positions, max_batch=max_batch, max_seq_len=max_seq_len,
                          arch=cfg.arch, n_kv_head=cfg.n_kv_head, rope_theta=cfg.rope_theta, vit_mlp=cfg.vit_mlp,
                          vit_eps=cfg.vit_eps, sliding_window=cfg.sliding_window if cfg.arch == "v2" else 0,
                          weight_dtype=weight_dtype, exclusive_device=exclusive_device)
    eng = sva.HipEngine(ec)
    eng.load_state_dict({k: v.to(torch.bfloat16) for k, v in w.items()})
    return eng


def rel_err(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return float((got - ref).abs().max() / ref.abs().max())


def mean_err(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return float((got - ref).abs().mean() / ref.abs().max())


def hf_decoder_bf16(cfg: O.OracleConfig, w):
    """The decoder class the reference instantiates (llm/starcoder.py:33 -> transformers GPTBigCodeForCausalLM; v2:
    Starcoder2ForCausalLM, llm/starcoder2.py:22-27) with the seeded weights, as REAL torch.bfloat16 modules on the GPU
    (eager attention: the restated gpt_bigcode `_attn`, softmax upcast to float32): the second comparator of the parity
    tests.  transformers is in the image; /root/reference is not needed."""
    if cfg.arch == "v2":
        from transformers import Starcoder2Config, Starcoder2ForCausalLM
        hc = Starcoder2Config(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.n_inner,
                              num_hidden_layers=cfg.n_layer, num_attention_heads=cfg.n_head, num_key_value_heads=cfg.n_kv_head,
                              hidden_act="gelu_pytorch_tanh", max_position_embeddings=cfg.n_positions, norm_epsilon=cfg.ln_eps,
                              rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window or 4096, use_bias=True,
                              tie_word_embeddings=True, residual_dropout=0.0, embedding_dropout=0.0, attention_dropout=0.0,
                              bos_token_id=0, eos_token_id=0, pad_token_id=cfg.pad_token_id)
        cls = Starcoder2ForCausalLM
    else:
        from transformers import GPTBigCodeConfig, GPTBigCodeForCausalLM
        hc = GPTBigCodeConfig(vocab_size=cfg.vocab, n_positions=cfg.n_positions, n_embd=cfg.hidden, n_layer=cfg.n_layer,
                              n_head=cfg.n_head, n_inner=cfg.n_inner, multi_query=True, activation_function="gelu_pytorch_tanh",
                              resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0, layer_norm_epsilon=cfg.ln_eps,
                              bos_token_id=0, eos_token_id=0, pad_token_id=cfg.pad_token_id)
        cls = GPTBigCodeForCausalLM
    hc._attn_implementation = "eager"
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev()):
            lm = cls(hc)
    finally:
        torch.set_default_dtype(prev)
    pre = "model.svg_transformer.transformer."
    res = lm.load_state_dict({k[len(pre):]: v.to(torch.bfloat16) for k, v in w.items() if k.startswith(pre)}, strict=False)
    assert not [k for k in res.missing_keys if "attn.bias" not in k and "masked_bias" not in k and "rotary" not in k], res
    return lm.eval()


def hf_teacher_forced_logits(lm, emb_bf16, token_emb, toks):
    """ONE full forward over [prompt rows | embeddings of toks[:, :-1]]: the logits that predict toks[:, t], t = 0..n-1."""
    import torch as _t
    with _t.no_grad():
        full = _t.cat([emb_bf16, token_emb(toks[:, :-1].to(emb_bf16.device))], 1) if toks.shape[1] > 1 else emb_bf16
        mask = _t.ones(full.shape[:2], dtype=_t.long, device=full.device)
        lg = lm(inputs_embeds=full, attention_mask=mask).logits
    S0 = emb_bf16.shape[1]
    return lg[:, S0 - 1:].float().cpu()
