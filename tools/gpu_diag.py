#!/usr/bin/env python3
"""GPU bring-up diagnostics: every operator and the end-to-end path against the CPU oracle, printing
error metrics (no asserts) so one gpurun call shows everything.  Writes gpurun_out/diag.json."""
from __future__ import annotations

import json
import os
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import starvector_amd as sva  # noqa: E402
from starvector_amd import engine as E  # noqa: E402
from oracle import starvector_oracle as O
from oracle.hostinfo import host_cores  # noqa: E402

OUT = {}
dev = torch.device("cuda", 0)


def bf(x):
    return x.to(torch.bfloat16).to(dev)


def metric(name, got, ref):
    got = got.float().cpu()
    ref = ref.float().cpu()
    d = (got - ref).abs()
    m = dict(max_abs=float(d.max()), mean_abs=float(d.mean()), ref_max=float(ref.abs().max()),
             rel=float(d.max() / (ref.abs().max() + 1e-12)), nan=int(torch.isnan(got).sum()))
    OUT[name] = m
    print(f"{name:44s} max_abs={m['max_abs']:.4e} mean_abs={m['mean_abs']:.4e} ref_max={m['ref_max']:.3e} "
          f"rel={m['rel']:.3e} nan={m['nan']}", flush=True)
    return m


def section(fn):
    try:
        t = time.time()
        fn()
        print(f"-- {fn.__name__} done in {time.time() - t:.1f}s", flush=True)
    except Exception:
        OUT[fn.__name__ + "_error"] = traceback.format_exc()
        print(f"!! {fn.__name__} FAILED:\n{traceback.format_exc()}", flush=True)


def ops_layernorm():
    g = torch.Generator().manual_seed(1)
    for (M, D) in [(5, 128), (257, 1024), (64, 2048)]:
        x = torch.randn(M, D, generator=g).bfloat16().float()
        w = (1 + 0.1 * torch.randn(D, generator=g)).bfloat16().float()
        b = (0.1 * torch.randn(D, generator=g)).bfloat16().float()
        ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-5)
        got = E.op_layernorm(bf(x), bf(w), bf(b))
        metric(f"layernorm M{M} D{D}", got, ref)


def ops_linear():
    g = torch.Generator().manual_seed(2)
    for (M, N, K, act, res) in [(32, 128, 64, "none", False), (300, 384, 128, "none", False),
                                (257, 1024, 588, "none", False), (200, 512, 1024, "gelu_tanh", False),
                                (130, 256, 256, "quickgelu", True), (1000, 3072, 1024, "swish", True)]:
        x = torch.randn(M, K, generator=g).bfloat16().float()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().float()
        b = (0.1 * torch.randn(N, generator=g)).bfloat16().float()
        r = torch.randn(M, N, generator=g).bfloat16().float()
        y = x @ W.T + b
        if act == "gelu_tanh":
            y = torch.nn.functional.gelu(y.bfloat16().float(), approximate="tanh")
        elif act == "quickgelu":
            yy = y.bfloat16().float(); y = yy * torch.sigmoid(1.702 * yy)
        elif act == "swish":
            yy = y.bfloat16().float(); y = yy * torch.sigmoid(yy)
        if res:
            y = y.bfloat16().float() + r
        got = E.op_linear(bf(x), bf(W), bf(b), bf(r) if res else None, act=act)
        metric(f"linear M{M} N{N} K{K} {act} res{int(res)}", got, y)
    # fp32 output
    x = torch.randn(70, 256, generator=g).bfloat16().float()
    W = (torch.randn(516, 256, generator=g) / 16).bfloat16().float()
    got = E.op_linear(bf(x), bf(W), None, None, out_f32=True)
    metric("linear f32out M70 N516 K256", got, x @ W.T)


def ops_skinny():
    g = torch.Generator().manual_seed(3)
    for (M, N, K, sk) in [(32, 64, 256, 1), (32, 2304, 2048, 4), (7, 516, 256, 2), (40, 2048, 8192, 4), (32, 96, 64, 1)]:
        x = torch.randn(M, K, generator=g).bfloat16().float()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).bfloat16().float()
        b = (0.1 * torch.randn(N, generator=g)).bfloat16().float()
        got = E.op_linear_skinny(bf(x), bf(W), bf(b), splitk=sk)
        metric(f"skinny M{M} N{N} K{K} splitk{sk}", got, x @ W.T + b)


def ops_attention():
    g = torch.Generator().manual_seed(4)
    for (B, S, H, Hkv, hd, causal) in [(2, 17, 2, 2, 64, 0), (2, 257, 16, 16, 64, 0), (3, 19, 2, 1, 128, 1),
                                       (2, 259, 16, 1, 128, 1), (1, 130, 4, 4, 128, 1), (1, 70, 8, 2, 64, 1)]:
        q = torch.randn(B, S, H * hd, generator=g).bfloat16().float()
        k = torch.randn(B, S, Hkv * hd, generator=g).bfloat16().float()
        v = torch.randn(B, S, Hkv * hd, generator=g).bfloat16().float()
        qq = q.view(B, S, H, hd).transpose(1, 2)
        kk = k.view(B, S, Hkv, hd).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
        vv = v.view(B, S, Hkv, hd).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
        s = qq @ kk.transpose(-1, -2) * hd ** -0.5
        if causal:
            s = s.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
        ref = (torch.softmax(s, -1) @ vv).transpose(1, 2).reshape(B, S, H * hd)
        got = E.op_attention(bf(q), bf(k), bf(v), H, Hkv, causal)
        metric(f"attention B{B} S{S} H{H}/{Hkv} d{hd} causal{causal}", got, ref)


def ops_misc():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 17, 256, generator=g).bfloat16().float()
    w = (1 + 0.1 * torch.randn(17, 256, generator=g)).bfloat16().float()
    b = (0.1 * torch.randn(17, 256, generator=g)).bfloat16().float()
    ref = torch.nn.functional.layer_norm(x, (17, 256), w, b, 1e-5)
    metric("plane_layernorm", E.op_plane_layernorm(bf(x), bf(w), bf(b)), ref)
    lg = torch.randn(5, 516, generator=g)
    lg[2, 100] = lg[2, 400] = 9.0          # tie -> lowest index
    got = E.op_argmax(lg.to(dev))
    print("argmax", got.tolist(), lg.argmax(-1).tolist(), flush=True)
    OUT["argmax_ok"] = bool(got.cpu().tolist() == lg.argmax(-1).tolist())
    # top-p sampler: empirical distribution vs the oracle's filtered distribution
    V = 64
    lg = (2.0 * torch.randn(1, V, generator=g))
    probs = O.top_p_filtered_probs(lg, 0.8, 0.9)[0]
    n = 4000
    rows = lg.repeat(n, 1).to(dev).contiguous()
    s = E.op_sample_top_p(rows, 0.8, 0.9, seed=123, step=7).cpu().long()
    emp = torch.bincount(s, minlength=V).float() / n
    OUT["top_p_l1"] = float((emp - probs).abs().sum())
    OUT["top_p_outside_support"] = float(emp[probs == 0].sum())
    print(f"top_p sampler: L1(emp, oracle)={OUT['top_p_l1']:.4f} mass outside support={OUT['top_p_outside_support']:.4f} "
          f"support={int((probs > 0).sum())}/{V}", flush=True)


def build_engine(cfg: O.OracleConfig, w, max_batch, max_seq_len):
    ec = sva.EngineConfig(image_size=cfg.image_size, patch_size=cfg.patch_size, vit_width=cfg.vit_width,
                          vit_layers=cfg.vit_layers, vit_heads=cfg.vit_heads, adapter_norm=cfg.adapter_norm,
                          hidden=cfg.hidden, n_layer=cfg.n_layer, n_head=cfg.n_head, n_inner=cfg.n_inner,
                          vocab=cfg.vocab, n_positions=cfg.n_positions, max_batch=max_batch, max_seq_len=max_seq_len)
    eng = sva.HipEngine(ec)
    eng.load_state_dict({k: v.to(torch.bfloat16) for k, v in w.items()})
    return eng


def e2e_case(tag, cfg, seed, B, n_new, modes=("fp32", "bf16")):
    w = O.make_weights(cfg, seed=seed)
    image = O.synthetic_images(B, cfg.image_size, seed=seed + 1)
    prompt = torch.tensor([[7, 11]] * B, dtype=torch.long)
    eng = build_engine(cfg, w, max_batch=max(B, 2), max_seq_len=min(cfg.n_positions, cfg.query_length + 2 + n_new + 8))
    enc = eng.encode_image(bf(image))
    vis = eng.adapter(enc)
    tok = eng.embed_tokens(prompt.to(dev))
    emb = torch.cat([vis, tok], 1)
    logits0 = eng.prefill(emb)
    for mode in modes:
        o_enc = O.image_encoder_forward(w, cfg, image, mode)
        o_vis = O.adapter_forward(w, cfg, o_enc, mode)
        o_emb = O.prepare_generation_inputs(w, cfg, image, prompt, mode)
        o_logits0, _ = O.decoder_prefill(w, cfg, o_emb, mode)
        metric(f"{tag} encoder vs oracle-{mode}", enc, o_enc)
        metric(f"{tag} adapter vs oracle-{mode}", vis, o_vis)
        metric(f"{tag} prefill logits vs oracle-{mode}", logits0, o_logits0)
    # decode-step logits: feed the oracle's greedy tokens
    S0 = emb.shape[1]
    o_emb = O.prepare_generation_inputs(w, cfg, image, prompt, "bf16")
    o_toks, o_lg = O.greedy_generate(w, cfg, o_emb, S0 + n_new, mode="bf16", return_logits=True)
    worst = 0.0
    for t in range(1, min(n_new, o_toks.shape[1])):
        lg = eng.decode_step(o_toks[:, t - 1].to(dev))
        worst = max(worst, float((lg.float().cpu() - o_lg[:, t]).abs().max()))
    OUT[f"{tag} decode logits max_abs vs oracle-bf16"] = worst
    print(f"{tag} decode-step logits (teacher-forced, {n_new - 1} steps) max_abs vs oracle-bf16 = {worst:.4e}", flush=True)
    # greedy generate
    for graph in (False, True):
        if graph:
            os.environ.pop("SV_NO_GRAPH", None)
        else:
            os.environ["SV_NO_GRAPH"] = "1"
        toks = eng.generate(emb, max_length=S0 + n_new, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id).cpu()
        n = min(toks.shape[1], o_toks.shape[1])
        same = bool(torch.equal(toks[:, :n], o_toks[:, :n])) and toks.shape == o_toks.shape
        top2 = o_lg.topk(2, -1).values
        margin = (top2[..., 0] - top2[..., 1])
        first_div = None
        if not same:
            neq = (toks[:, :n] != o_toks[:, :n]).nonzero()
            if len(neq):
                b0, t0 = int(neq[0, 0]), int(neq[:, 1].min())
                first_div = dict(step=t0, margin_at_step=float(margin[:, t0].min()))
        OUT[f"{tag} greedy graph={graph}"] = dict(identical=same, shape=list(toks.shape), first_div=first_div,
                                                 min_margin=float(margin.min()), timing=eng.last_timing())
        print(f"{tag} greedy graph={graph}: identical={same} shape={tuple(toks.shape)} first_div={first_div} "
              f"min_margin={float(margin.min()):.3e} timing={eng.last_timing()}", flush=True)
    eng.close()


def e2e_tiny():
    e2e_case("tiny", O.OracleConfig.tiny(), 1234, 3, 24)


def e2e_tiny_bn():
    import dataclasses
    e2e_case("tiny_bn", dataclasses.replace(O.OracleConfig.tiny(), adapter_norm="batch_norm"), 4321, 2, 8)


def e2e_golden_stop():
    from safetensors.torch import load_file
    gold = load_file(os.path.join(ROOT, "tests", "golden", "tiny_stop.safetensors"))
    seed, B, n_new, eos = [int(x) for x in gold["meta"]]
    import dataclasses
    cfg = dataclasses.replace(O.OracleConfig.tiny(), eos_token_id=eos)
    w = O.apply_fixture_weights(O.make_weights(cfg, seed=seed), cfg, gold)
    eng = build_engine(cfg, w, 4, 64)
    emb = torch.cat([eng.adapter(eng.encode_image(bf(gold["image"]))), eng.embed_tokens(gold["prompt_ids"].to(dev))], 1)
    toks = eng.generate(emb, max_length=emb.shape[1] + n_new, eos_token_id=eos, pad_token_id=cfg.pad_token_id,
                        stop_ids=gold["stop_ids"].tolist()).cpu()
    ok = toks.shape == gold["tokens"].shape and bool(torch.equal(toks, gold["tokens"]))
    OUT["golden_stop_identical"] = ok
    print("golden stop case identical:", ok, tuple(toks.shape), tuple(gold["tokens"].shape), flush=True)
    if not ok:
        print(toks.tolist(), gold["tokens"].tolist(), flush=True)
    eng.close()


def e2e_full_1b():
    # StarVector-1B shapes, B=2, few tokens (oracle on CPU takes a while: keep it small)
    torch.set_num_threads(host_cores())
    e2e_case("1b", O.OracleConfig(), 1234, 2, 6, modes=("bf16",))


def perf_1b():
    """first performance numbers at BASELINE config 2 shapes (B=32), short decode"""
    cfg = O.OracleConfig()
    t = time.time()
    w = O.make_weights(cfg, seed=1, init="std002")
    print(f"weights generated in {time.time() - t:.1f}s", flush=True)
    B, n_new = 32, 128
    t = time.time()
    eng = build_engine(cfg, w, max_batch=B, max_seq_len=259 + n_new + 8)
    del w
    print(f"engine loaded in {time.time() - t:.1f}s", flush=True)
    img = bf(O.synthetic_images(B, 224, seed=2))
    prompt = torch.tensor([[7, 11]] * B, dtype=torch.long, device=dev)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        enc = eng.encode_image(img); torch.cuda.synchronize(); t1 = time.time()
        vis = eng.adapter(enc); torch.cuda.synchronize(); t2 = time.time()
        emb = torch.cat([vis, eng.embed_tokens(prompt)], 1); torch.cuda.synchronize(); t3 = time.time()
        print(f"iter {it}: encoder {1e3*(t1-t0):.2f} ms, adapter {1e3*(t2-t1):.2f} ms, embed+cat {1e3*(t3-t2):.2f} ms", flush=True)
    res = {}
    for graph in (False, True, True):
        if graph:
            os.environ.pop("SV_NO_GRAPH", None)
        else:
            os.environ["SV_NO_GRAPH"] = "1"
        torch.cuda.synchronize(); t0 = time.time()
        toks = eng.generate(emb, max_length=emb.shape[1] + n_new, eos_token_id=-1, pad_token_id=cfg.pad_token_id)
        torch.cuda.synchronize(); dt = time.time() - t0
        tm = eng.last_timing()
        per_step = tm["decode_ms"] / max(tm["decode_steps"], 1)
        print(f"generate graph={graph}: total {dt*1e3:.1f} ms, {tm}, {per_step*1e3:.1f} us/step, "
              f"{B * 1e3 / per_step:.0f} tok/s decode-only, tokens {tuple(toks.shape)} uniq {toks.unique().numel()}", flush=True)
        res[f"graph={graph}"] = dict(total_ms=dt * 1e3, **tm)
    prof = eng.profile_decode_step(B, iters=5)
    print("decode-step profile (ctx ~%d):" % (emb.shape[1] + n_new), json.dumps(prof), flush=True)
    W = 2 * (24 * 42490112 + 4096 + 100671488)
    sk = prof["skinny_gemm"]["ms_per_step"]
    print(f"skinny GEMMs: {W/1e6:.0f} MB weights / {sk*1e3:.1f} us = {W / (sk * 1e-3) / 1e12:.2f} TB/s", flush=True)
    OUT["perf_1b"] = dict(gen=res, prof=prof)
    eng.close()


def main():
    print("device:", torch.cuda.get_device_name(0), flush=True)
    which = sys.argv[1:] or ["ops", "tiny", "full"]
    if "ops" in which:
        for f in (ops_layernorm, ops_linear, ops_skinny, ops_attention, ops_misc):
            section(f)
    if "tiny" in which:
        for f in (e2e_tiny, e2e_tiny_bn, e2e_golden_stop):
            section(f)
    if "full" in which:
        section(e2e_full_1b)
    if "perf" in which:
        section(perf_1b)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w") as f:
        json.dump(OUT, f, indent=1)


if __name__ == "__main__":
    main()
