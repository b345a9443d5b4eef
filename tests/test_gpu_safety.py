"""GPU: the safety contract of the decode launches whose blocks wait for each other (mlp_fused_kernel, rowln_cattn_kernel; enabled by
sv_config.exclusive_device).  include/starvector_hip.h and INTEGRATION.md promise: when the engine does NOT own its GPU after all, a call
ends with an error within milliseconds -- never a hang, never tokens -- and the engine is usable again afterwards.  The reference has no
such launches (HF generate is one kernel per op: starvector_base.py:228-241); this is the price of fusing them and it is tested like one."""
import time

import pytest
import torch

import starvector_amd as sva
from tests.gpu_util import dev

pytestmark = pytest.mark.gpu


def _inputs(eng, B):
    g = torch.Generator().manual_seed(9)
    img = torch.randn(B, 3, 224, 224, generator=g).to(torch.bfloat16).to(dev())
    prompt = torch.tensor([[7, 11]] * B, dtype=torch.long, device=dev())
    return torch.cat([eng.adapter(eng.encode_image(img)), eng.embed_tokens(prompt)], 1)


def test_foreign_tenant_on_half_the_cus_ends_the_call_with_an_error_and_the_engine_recovers():
    """A tenant pins 144 KiB of LDS on HALF the CUs for 400 ms (sv_debug_occupy_cus: what a second process's kernels would do to an engine that
    was told it owns the GPU).  The fused MLP launch needs its 256 blocks resident at once, one per CU: half of them are, wait for tiles the
    other half would produce, give up after the wall-clock bound (5 ms) and raise code 3; every later launch of the step sees the flag and
    falls through.  sv_generate (greedy and beam), sv_decode_step: an error each, well inside a second.  Afterwards: the reference tokens."""
    B = 32
    eng = sva.HipEngine(sva.EngineConfig(max_batch=B, max_seq_len=259 + 64, exclusive_device=True))
    eng.load_random_weights(seed=13)
    emb = _inputs(eng, B)
    S0 = emb.shape[1]
    kw = dict(max_length=S0 + 24, eos_token_id=-1, pad_token_id=49152)
    ref = eng.generate(emb, **kw).cpu()
    assert ref.unique().numel() > 4
    cus = torch.cuda.get_device_properties(0).multi_processor_count

    def tenant():
        torch.cuda.synchronize()
        eng.debug_occupy_cus(cus // 2, 144 * 1024, 400)
        time.sleep(0.02)                                   # let its blocks take their CUs

    # greedy sv_generate
    tenant()
    t0 = time.time()
    with pytest.raises(sva.StarVectorHipError, match="gave up waiting"):
        eng.generate(emb, **kw)
    t_greedy = time.time() - t0
    torch.cuda.synchronize()                               # the tenant leaves
    assert torch.equal(eng.generate(emb, **kw).cpu(), ref), "the engine did not recover after a give-up"

    # sv_decode_step (this entry point used to return rc 0 with logits computed from the pattern)
    lg = eng.prefill(emb)
    tok = lg.argmax(-1)
    good = eng.decode_step(tok).float().cpu()
    eng.prefill(emb)
    tenant()
    t0 = time.time()
    with pytest.raises(sva.StarVectorHipError, match="gave up waiting"):
        eng.decode_step(tok)
    t_step = time.time() - t0
    torch.cuda.synchronize()
    eng.prefill(emb)
    assert torch.equal(eng.decode_step(tok).float().cpu(), good)

    # beam search (the reference's default num_beams = 2)
    kwb = dict(max_length=S0 + 12, eos_token_id=-1, pad_token_id=49152, num_beams=2)
    refb = eng.generate(emb[:8].contiguous(), **kwb).cpu()
    tenant()
    t0 = time.time()
    with pytest.raises(sva.StarVectorHipError, match="gave up waiting"):
        eng.generate(emb[:8].contiguous(), **kwb)
    t_beam = time.time() - t0
    torch.cuda.synchronize()
    assert torch.equal(eng.generate(emb[:8].contiguous(), **kwb).cpu(), refb)
    print(f"[safety] error after {t_greedy * 1e3:.0f} ms (greedy, 24 tokens incl. the prompt pass on half the CUs), {t_step * 1e3:.0f} ms (one decode step), "
          f"{t_beam * 1e3:.0f} ms (beam); the tenant holds its CUs for 400 ms")
    assert max(t_greedy, t_step, t_beam) < 2.0
    eng.close()


def test_optimistic_ownership_falls_back_for_good_and_reruns_the_call():
    """sv_config.exclusive_device = 2 (EngineConfig(exclusive_device="auto"), what the drop-in model wrapper passes by default): the fused launches are on
    until one gives up; the engine then switches them off for the rest of its life and sv_generate runs the failed call AGAIN -- the caller gets the
    tokens (the kernels are bit-identical with and without the fused launches), late by the failed attempt; a streaming callback sees every column
    exactly once; later calls run unfused without an error even while the tenant is still there."""
    B = 32
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    eng = sva.HipEngine(sva.EngineConfig(max_batch=B, max_seq_len=259 + 64, exclusive_device="auto"))
    eng.load_random_weights(seed=13)
    emb = _inputs(eng, B)
    S0 = emb.shape[1]
    kw = dict(max_length=S0 + 24, eos_token_id=-1, pad_token_id=49152)
    ref = eng.generate(emb, **kw).cpu()
    plan = eng.step_plan()
    assert plan["mlp_fused"] and plan["rowln_cattn_fused"], "the optimistic engine did not start with the fused launches"
    seen = []
    torch.cuda.synchronize()
    eng.debug_occupy_cus(cus // 2, 144 * 1024, 400)
    time.sleep(0.02)
    t0 = time.time()
    got = eng.generate(emb, sync_every=4, on_tokens=lambda t, first: seen.append((first, t.clone())), **kw).cpu()      # no exception
    dt = time.time() - t0
    assert torch.equal(got, ref), "the re-run of an optimistic call did not reproduce the tokens"
    plan = eng.step_plan()
    assert not plan["mlp_fused"] and not plan["rowln_cattn_fused"], "the engine kept the fused launches after a give-up"
    cols = torch.cat([t for _, t in sorted(seen, key=lambda x: x[0])], 1)
    firsts = sorted(f for f, _ in seen)
    assert firsts == sorted(set(firsts)) and cols.shape[1] == ref.shape[1] and torch.equal(cols.to(ref.dtype), ref), "a column was streamed twice, never, or wrong"
    # the tenant is still there (400 ms): the engine now decodes beside it
    assert torch.equal(eng.generate(emb, **kw).cpu(), ref)
    torch.cuda.synchronize()
    assert torch.equal(eng.generate(emb, **kw).cpu(), ref) and not eng.step_plan()["mlp_fused"]
    print(f"[safety] optimistic engine: the call that met the tenant took {dt * 1e3:.0f} ms (failed attempt + re-run), tokens identical, every column streamed once")
    eng.close()
    # continuous batching on an optimistic engine: the batch that meets the tenant fails ONCE (its step graphs carried the fused launches and are dropped
    # with them), the next batch -- the tenant still there -- decodes
    eng = sva.HipEngine(sva.EngineConfig(max_batch=8, max_seq_len=259 + 64, exclusive_device="auto"))
    eng.load_random_weights(seed=13)
    emb4 = _inputs(eng, 4)
    req = dict(max_new_tokens=16, eos_token_id=-1, pad_token_id=49152)
    slots = eng.cb_admit(emb4, [req] * 4)
    while eng.cb_step(8) > 0:
        pass
    want = [eng.cb_read(s, 0, 16) for s in slots]
    eng.cb_reset()
    torch.cuda.synchronize()
    eng.debug_occupy_cus(cus // 2, 144 * 1024, 400)
    time.sleep(0.02)
    slots = eng.cb_admit(emb4, [req] * 4)
    with pytest.raises(sva.StarVectorHipError, match="gave up waiting"):
        while eng.cb_step(8) > 0:
            pass
    eng.cb_reset()
    slots = eng.cb_admit(emb4, [req] * 4)                      # the tenant holds its CUs for 400 ms: still there
    while eng.cb_step(8) > 0:
        pass
    assert all(torch.equal(eng.cb_read(s, 0, 16), w) for s, w in zip(slots, want)), "continuous batching after the fall-back"
    eng.cb_reset()
    torch.cuda.synchronize()
    eng.close()


def test_tenant_that_leaves_room_changes_nothing():
    """Control: a tenant on EVERY CU that pins only 16 KiB leaves room for every block of the engine (the fused launches take 37 / 2 x 32 KiB, the
    256-row-tile prompt-pass GEMM 128 KiB of a CU's 160) -- concurrent work as such is not what fails: same tokens, no error."""
    B = 32
    eng = sva.HipEngine(sva.EngineConfig(max_batch=B, max_seq_len=259 + 64, exclusive_device=True))
    eng.load_random_weights(seed=13)
    emb = _inputs(eng, B)
    kw = dict(max_length=emb.shape[1] + 24, eos_token_id=-1, pad_token_id=49152)
    ref = eng.generate(emb, **kw).cpu()
    torch.cuda.synchronize()
    eng.debug_occupy_cus(torch.cuda.get_device_properties(0).multi_processor_count, 16 * 1024, 400)
    time.sleep(0.02)
    assert torch.equal(eng.generate(emb, **kw).cpu(), ref)
    torch.cuda.synchronize()
    eng.close()


def test_a_poisoned_request_fails_its_batch_once_and_nothing_after_it():
    """A request whose prompt embeddings carry NaN writes NaN K / V rows into its pages.  The decode attention masks a key behind a sequence's
    position by its score (a select: NaN-safe), but the key group's stale V rows still passed through the P.V MFMA with P = 0, and
    0 x NaN = NaN: before round 5 the NEXT owners of those pages failed too -- call after call, until every stale row had been overwritten
    (measured with this test: 5 more failing steps through sv_decode_step, one more failing sv_generate / continuous batch).  The kernel now
    clears the stale V columns of the one key group that has any (attention.hip, process()).  Three entry points: the step-wise pair
    (sv_prefill / sv_decode_step hand the logits to the caller: no flag), continuous batching, the classic sv_generate."""
    import os
    from safetensors.torch import load_file
    from oracle import starvector_oracle as O
    from tests.gpu_util import bf, build_engine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = load_file(os.path.join(root, "tests", "golden", "tiny_b3.safetensors"))
    cfg = O.OracleConfig.tiny()
    w = O.apply_fixture_weights(O.make_weights(cfg, seed=int(g["meta"][0])), cfg, g)
    eng = build_engine(cfg, w, max_batch=8, max_seq_len=120)
    img = bf(g["image"])
    prompt = torch.tensor([[7, 11]] * 3, device=dev())
    emb = torch.cat([eng.adapter(eng.encode_image(img)), eng.embed_tokens(prompt)], 1).contiguous()
    gold = g["tokens"]
    req = dict(max_new_tokens=24, eos_token_id=-1)
    bad = emb.clone()
    bad[1, 2, 5] = float("nan")
    poisoned = torch.cat([emb[0:1], bad[1:2], emb[2:3]], 0).contiguous()

    # (1) step by step: prompt 7 rows + 8 steps, every sequence stays inside its first 32-key group, so the next owner of the page has the
    # stale rows behind its position in the group it reads at every step
    def walk(x, n=8):
        lg = [eng.prefill(x).float().cpu()]
        for _ in range(n):
            lg.append(eng.decode_step(torch.nan_to_num(lg[-1]).argmax(-1).to(dev())).float().cpu())
        return torch.stack(lg)
    clean = walk(emb)
    assert torch.isfinite(clean).all()
    dirty = walk(poisoned)
    assert torch.isnan(dirty[:, 1]).all() and torch.isfinite(dirty[:, 0]).all() and torch.isfinite(dirty[:, 2]).all()   # rows do not mix
    assert torch.equal(walk(emb), clean), "stale non-finite K / V rows of the previous owner reached the next one"

    # (2) continuous batching: the poisoned request next to two good ones -- the batch fails once (as the scheduler documents) ...
    eng.cb_admit(poisoned, [req, req, req])
    with pytest.raises(sva.StarVectorHipError, match="no finite value"):
        for _ in range(4):
            eng.cb_step(8)
    eng.cb_reset()
    slots = eng.cb_admit(emb, [req, req, req])               # ... and the same slots, the same pages serve the next requests
    while eng.cb_step(8) > 0:
        pass
    for i, s in enumerate(slots):
        assert torch.equal(eng.cb_read(s, 0, 24), gold[i]), f"request {i} after the poisoned batch differs from the golden stream"
    eng.cb_reset()

    # (3) the classic entry point: a poisoned call, then a clean one on the same pages
    kw = dict(max_length=emb.shape[1] + 24, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id)
    with pytest.raises(sva.StarVectorHipError, match="no finite value"):
        eng.generate(poisoned, max_length=emb.shape[1] + 24, eos_token_id=-1)
    assert torch.equal(eng.generate(emb, **kw).cpu(), gold)
    eng.close()


def test_continuous_batching_on_an_engine_that_owns_its_gpu_equals_solo_runs():
    """The serving deployment with SV_EXCLUSIVE_DEVICE=1: the slots of a continuous batch run through the same decode_forward as sv_generate, so
    on an exclusive engine their steps take BOTH fused launches (rows >= 10: the attention launch has room for the patterns).  StarVector-1B
    widths, 6 decoder layers: twelve requests admitted in two groups (bucket 16), greedy and sampled -- every slot's stream equals its solo
    sv_generate run, and equals what an engine WITHOUT the fused launches (SV_EXP 512 + 8192) gives."""
    B = 12
    eng = sva.HipEngine(sva.EngineConfig(max_batch=16, max_seq_len=259 + 80, n_layer=6, vit_layers=2, exclusive_device=True))
    eng.load_random_weights(seed=21)
    g = torch.Generator().manual_seed(5)
    img = torch.randn(B, 3, 224, 224, generator=g).to(torch.bfloat16).to(dev())
    prompt = torch.tensor([[7, 11]] * B, dtype=torch.long, device=dev())
    emb = torch.cat([eng.adapter(eng.encode_image(img)), eng.embed_tokens(prompt)], 1).contiguous()
    S0 = emb.shape[1]
    reqs = [dict(max_new_tokens=40 + 3 * i, eos_token_id=-1) if i % 3 else
            dict(max_new_tokens=40 + 3 * i, eos_token_id=-1, do_sample=True, temperature=0.9, top_p=0.9, top_k=50, seed=100 + i) for i in range(B)]

    def solo(i):
        r = reqs[i]
        return eng.generate(emb[i:i + 1].contiguous(), max_length=S0 + r["max_new_tokens"], eos_token_id=-1, do_sample=r.get("do_sample", False),
                            temperature=r.get("temperature", 1.0), top_p=r.get("top_p", 1.0), top_k=r.get("top_k", 0), seed=r.get("seed", 0)).cpu()[0]

    def batched():
        slots = eng.cb_admit(emb[:7].contiguous(), reqs[:7])
        eng.cb_step(5)
        slots += eng.cb_admit(emb[7:].contiguous(), reqs[7:])
        while eng.cb_step(8) > 0:
            pass
        out = [eng.cb_read(s, 0, reqs[i]["max_new_tokens"]) for i, s in enumerate(slots)]
        eng.cb_reset()
        return out

    alone = [solo(i) for i in range(B)]
    assert len({tuple(a.tolist()[:8]) for a in alone}) >= 4                     # distinct streams (a 6-layer random model repeats itself under greedy)
    fused = batched()
    eng.set_exp(512 + 8192)                                                     # neither fused launch
    plain = batched()
    eng.set_exp(0)
    for i in range(B):
        assert torch.equal(fused[i], alone[i]), f"slot {i}: continuous batch (fused launches) differs from the solo run"
        assert torch.equal(plain[i], alone[i]), f"slot {i}: continuous batch (plain launches) differs from the solo run"
    eng.close()
