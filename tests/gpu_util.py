"""Helpers shared by the -m gpu tests (engine construction from an oracle config)."""
import torch

import starvector_amd as sva
from oracle import starvector_oracle as O


def dev():
    return torch.device("cuda", 0)


def bf(x):
    return x.to(torch.bfloat16).to(dev())


def build_engine(cfg: O.OracleConfig, w, max_batch, max_seq_len, weight_dtype="bf16"):
    ec = sva.EngineConfig(image_size=cfg.image_size, patch_size=cfg.patch_size, vit_width=cfg.vit_width,
                          vit_layers=cfg.vit_layers, vit_heads=cfg.vit_heads, adapter_norm=cfg.adapter_norm,
                          hidden=cfg.hidden, n_layer=cfg.n_layer, n_head=cfg.n_head, n_inner=cfg.n_inner,
                          vocab=cfg.vocab, n_positions=cfg.n_positions, max_batch=max_batch, max_seq_len=max_seq_len,
                          arch=cfg.arch, n_kv_head=cfg.n_kv_head, rope_theta=cfg.rope_theta, vit_mlp=cfg.vit_mlp,
                          vit_eps=cfg.vit_eps, sliding_window=cfg.sliding_window if cfg.arch == "v2" else 0,
                          weight_dtype=weight_dtype)
    eng = sva.HipEngine(ec)
    eng.load_state_dict({k: v.to(torch.bfloat16) for k, v in w.items()})
    return eng


def rel_err(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return float((got - ref).abs().max() / ref.abs().max())


def mean_err(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return float((got - ref).abs().mean() / ref.abs().max())
